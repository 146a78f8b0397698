// Bandwidth-bound kernels of the recurrent decoder on CHANNEL-BLOCKED bf16 tensors (gfx950, `-dtype bf16`; the convs are in
// conv_blk_dec.hip).  A blk tensor is the logical [B][C][H][W] tensor stored as bf16 [B][C/8][H][W][8]: one thread works on one
// 16-byte cell (8 channels of a pixel), computes in fp32 and rounds ONCE at the store.  Reference ops:
//   * the align-corners bilinear upsample between pyramid levels and its transpose (nn.UpsamplingBilinear2d, model.py:149-150,163-164),
//     the transpose with the side max-pool's gradient (model.py:143) added at the arg-max pixel,
//   * the ConvLSTM pointwise backward (derivative of clstm.py:47-58, formulas as in pointwise.hip lstm_bwd_kernel),
//   * conv_out (model.py:109,167: 8 -> 1 channels, 3x3) over ALL T timesteps: forward, data gradient, weight + bias gradient.
// The first three come as GROUPED launches: the cells of one diagonal of the decoder's (level, timestep) wavefront are independent,
// so their jobs share a grid (job table by value in the kernel arguments, block ranges per job).
#include "common.h"

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned kd_pack2(float lo, float hi) {
  const f32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ void kd_unpack(const u32x4 c, float (&v)[8]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(c[k] << 16); v[2 * k + 1] = __uint_as_float(c[k] & 0xFFFF0000u); }
}
__device__ __forceinline__ u32x4 kd_pack(const float (&v)[8]) {
  u32x4 c;
#pragma unroll
  for (int k = 0; k < 4; ++k) c[k] = kd_pack2(v[2 * k], v[2 * k + 1]);
  return c;
}

#define RSIS_KD_MAXJ 8
template <typename J>
struct KdGroup {
  int n;
  long begin[RSIS_KD_MAXJ + 1];      // first thread (cell) of every job; begin[n] = total
  J job[RSIS_KD_MAXJ];
};
// job of thread e (begin[] ascending); returns the job index and the thread's index inside it
template <typename G>
__device__ __forceinline__ int kd_find(const G& g, long e, long& local) {
  int j = 0;
#pragma unroll
  for (int k = 1; k < RSIS_KD_MAXJ; ++k) j += (k < g.n && g.begin[k] <= e) ? 1 : 0;
  local = e - g.begin[j];
  return j;
}

// ------------------------------------------------------------------------------------------------
// upsample forward: one thread = one output cell, 4 input cells (same arithmetic per element as upsample_fwd_kernel)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blk_upsample_fwd_group_kernel(const KdGroup<BlkResizeJob> g) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= g.begin[g.n]) return;
  long loc;
  const BlkResizeJob& p = g.job[kd_find(g, e, loc)];
  const int Hi = p.Hi, Wi = p.Wi, Ho = p.Ho, Wo = p.Wo;
  const int wo = (int)(loc % Wo);
  const long t = loc / Wo;
  const int ho = (int)(t % Ho);
  const long pl = t / Ho;
  const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  int h0, h1, w0, w1; float lh, lw;
  ac_coord(ho, sh, Hi, h0, h1, lh);
  ac_coord(wo, sw, Wi, w0, w1, lw);
  const u32x4* xb = (const u32x4*)p.src + pl * Hi * Wi;
  float v00[8], v01[8], v10[8], v11[8], o[8];
  kd_unpack(xb[h0 * Wi + w0], v00); kd_unpack(xb[h0 * Wi + w1], v01);
  kd_unpack(xb[h1 * Wi + w0], v10); kd_unpack(xb[h1 * Wi + w1], v11);
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = (1.f - lh) * ((1.f - lw) * v00[k] + lw * v01[k]) + lh * ((1.f - lw) * v10[k] + lw * v11[k]);
  ((u32x4*)p.dst)[loc] = kd_pack(o);
}

// ------------------------------------------------------------------------------------------------
// upsample backward (gather form, as upsample_bwd_generic_kernel: the candidate outputs of an input pixel are scanned with the SAME
// ac_coord the forward uses, so forward and backward agree on which outputs touch which input) + the pooled gradient at the arg-max
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blk_upsample_bwd_group_kernel(const KdGroup<BlkResizeJob> g) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= g.begin[g.n]) return;
  long loc;
  const BlkResizeJob& p = g.job[kd_find(g, e, loc)];
  const int Hi = p.Hi, Wi = p.Wi, Ho = p.Ho, Wo = p.Wo;
  const int wi = (int)(loc % Wi);
  const long t = loc / Wi;
  const int hi = (int)(t % Hi);
  const long pl = t / Hi;
  const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  int ho_lo = 0, ho_hi = Ho - 1, wo_lo = 0, wo_hi = Wo - 1;
  if (sh > 0.f) { ho_lo = max(0, (int)floorf((hi - 1) / sh) - 1); ho_hi = min(Ho - 1, (int)ceilf((hi + 1) / sh) + 1); }
  if (sw > 0.f) { wo_lo = max(0, (int)floorf((wi - 1) / sw) - 1); wo_hi = min(Wo - 1, (int)ceilf((wi + 1) / sw) + 1); }
  const u32x4* yb = (const u32x4*)p.src + pl * Ho * Wo;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // the interpolation is separable: the column weights of the <= KD_MAXC candidate outputs are computed ONCE (a scan that recomputes
  // them for every candidate row spends its time in ~60 ac_coord evaluations per cell, 9-16 of which contribute: 30 us per diagonal
  // for a few MB); wider candidate ranges (upsampling by more than ~4x) take the plain scan
  constexpr int KD_MAXC = 10;
  if (wo_hi - wo_lo < KD_MAXC) {
    float ww[KD_MAXC];
#pragma unroll
    for (int k = 0; k < KD_MAXC; ++k) {
      const int wo = wo_lo + k;
      int w0, w1; float lw;
      ac_coord(wo <= wo_hi ? wo : wo_hi, sw, Wi, w0, w1, lw);
      ww[k] = wo <= wo_hi ? (w0 == wi ? 1.f - lw : 0.f) + (w1 == wi ? lw : 0.f) : 0.f;
    }
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
      int h0, h1; float lh;
      ac_coord(ho, sh, Hi, h0, h1, lh);
      const float wh = (h0 == hi ? 1.f - lh : 0.f) + (h1 == hi ? lh : 0.f);
      if (wh == 0.f) continue;
      float row[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KD_MAXC; ++k) {
        if (ww[k] != 0.f) {
          float v[8];
          kd_unpack(yb[ho * Wo + wo_lo + k], v);
#pragma unroll
          for (int c = 0; c < 8; ++c) row[c] += ww[k] * v[c];
        }
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] += wh * row[c];
    }
  } else {
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
      int h0, h1; float lh;
      ac_coord(ho, sh, Hi, h0, h1, lh);
      const float wh = (h0 == hi ? 1.f - lh : 0.f) + (h1 == hi ? lh : 0.f);
      if (wh == 0.f) continue;
      float row[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        int w0, w1; float lw;
        ac_coord(wo, sw, Wi, w0, w1, lw);
        const float ww = (w0 == wi ? 1.f - lw : 0.f) + (w1 == wi ? lw : 0.f);
        if (ww != 0.f) {
          float v[8];
          kd_unpack(yb[ho * Wo + wo], v);
#pragma unroll
          for (int k = 0; k < 8; ++k) row[k] += ww * v[k];
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += wh * row[k];
    }
  }
  if (p.dpool) {        // plane pl = (b, channel block): channels 8 pl .. 8 pl + 7 of the [B][C] arrays
    const int sp = hi * Wi + wi;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (p.arg[pl * 8 + k] == sp) acc[k] += p.dpool[pl * 8 + k];
  }
  ((u32x4*)p.dst)[loc] = kd_pack(acc);
}

// ------------------------------------------------------------------------------------------------
// ConvLSTM pointwise backward: one thread = the 8 hidden channels of one pixel (one dh cell, four act / da cells)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blk_lstm_bwd_group_kernel(const KdGroup<BlkLstmBwdJob> g) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= g.begin[g.n]) return;
  long loc;
  const BlkLstmBwdJob& p = g.job[kd_find(g, e, loc)];
  const int HW = p.HW, hid = p.hid;
  const int sp = (int)(loc % HW);
  const long bc = loc / HW;                 // b * (hid / 8) + cbh
  const int Cbh = hid >> 3;
  const long b = bc / Cbh;
  const int cbh = (int)(bc - b * Cbh);
  float dh[8], d2[8];
  kd_unpack(((const u32x4*)p.dh)[loc], dh);
  if (p.dh2) {
    kd_unpack(((const u32x4*)p.dh2)[loc], d2);
#pragma unroll
    for (int k = 0; k < 8; ++k) dh[k] += d2[k];
  }
  const size_t s0 = ((size_t)b * hid + cbh * 8) * HW + sp;       // fp32 [B][hid][HW] index of channel 8 cbh
  float cv[8], cp[8], dcn[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    cv[k] = p.c[s0 + (size_t)k * HW];
    cp[k] = p.c_prev ? p.c_prev[s0 + (size_t)k * HW] : 0.f;
    dcn[k] = p.dc_next ? p.dc_next[s0 + (size_t)k * HW] : 0.f;
  }
  const size_t a0 = ((size_t)b * 4 * Cbh + 4 * cbh) * HW + sp;   // act cell of hidden channels (8 cbh, 8 cbh + 1); the next pairs are HW cells apart
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float a[8], da[8];
    kd_unpack(((const u32x4*)p.act)[a0 + (size_t)q * HW], a);
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int k = 2 * q + h2;
      const float gi = a[4 * h2], gf = a[4 * h2 + 1], go = a[4 * h2 + 2], gg = a[4 * h2 + 3];
      const float tc = rsis_tanh_fast(cv[k]);
      const float dcv = dh[k] * go * (1.f - tc * tc) + dcn[k];
      da[4 * h2] = dcv * gg * gi * (1.f - gi);
      da[4 * h2 + 1] = dcv * cp[k] * gf * (1.f - gf);
      da[4 * h2 + 2] = dh[k] * tc * go * (1.f - go);
      da[4 * h2 + 3] = dcv * gi * (1.f - gg * gg);
      if (p.dc_prev) p.dc_prev[s0 + (size_t)k * HW] = dcv * gf;
    }
    ((u32x4*)p.da)[a0 + (size_t)q * HW] = kd_pack(da);
  }
}

template <typename J, typename K>
static int kd_launch(K kernel, const J* jobs, const long* cells, int n, hipStream_t st) {
  for (int j0 = 0; j0 < n; j0 += RSIS_KD_MAXJ) {
    const int m = n - j0 < RSIS_KD_MAXJ ? n - j0 : RSIS_KD_MAXJ;
    KdGroup<J> g;
    g.n = m;
    long tot = 0;
    for (int k = 0; k < m; ++k) { g.begin[k] = tot; g.job[k] = jobs[j0 + k]; tot += cells[j0 + k]; }
    for (int k = m; k <= RSIS_KD_MAXJ; ++k) g.begin[k] = tot;
    if (tot < 1) continue;
    if ((tot + 255) / 256 > 0x7FFFFFFFL) return RSIS_ERR_ARG;
    hipLaunchKernelGGL(kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, g);
    if (rsis_check_launch() != RSIS_OK) return RSIS_ERR_LAUNCH;
  }
  return RSIS_OK;
}

int rsis_l_blk_upsample_fwd(const BlkResizeJob* jobs, int n, hipStream_t st) {
  long cells[64];
  if (n > 64) return RSIS_ERR_ARG;
  for (int j = 0; j < n; ++j) cells[j] = (long)jobs[j].planes * jobs[j].Ho * jobs[j].Wo;
  return kd_launch(blk_upsample_fwd_group_kernel, jobs, cells, n, st);
}
int rsis_l_blk_upsample_bwd(const BlkResizeJob* jobs, int n, hipStream_t st) {
  long cells[64];
  if (n > 64) return RSIS_ERR_ARG;
  for (int j = 0; j < n; ++j) cells[j] = (long)jobs[j].planes * jobs[j].Hi * jobs[j].Wi;
  return kd_launch(blk_upsample_bwd_group_kernel, jobs, cells, n, st);
}
int rsis_l_blk_lstm_bwd(const BlkLstmBwdJob* jobs, int n, hipStream_t st) {
  long cells[64];
  if (n > 64) return RSIS_ERR_ARG;
  for (int j = 0; j < n; ++j) cells[j] = (long)jobs[j].B * (jobs[j].hid >> 3) * jobs[j].HW;
  return kd_launch(blk_lstm_bwd_group_kernel, jobs, cells, n, st);
}

// ------------------------------------------------------------------------------------------------
// y[cell] = sum_t x[t][cell] (t ascending, fp32 accumulation, one rounding): the gradient of a level's time-invariant gate term
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blk_sum_leading_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, int T, long n) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float acc[8], v[8];
  kd_unpack(x[e], acc);
  for (int t = 1; t < T; ++t) {
    kd_unpack(x[(size_t)t * n + e], v);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += v[k];
  }
  y[e] = kd_pack(acc);
}
int rsis_l_blk_sum_leading(const void* x, void* y, int T, long n, hipStream_t st) {
  if ((n + 255) / 256 > 0x7FFFFFFFL) return RSIS_ERR_ARG;
  hipLaunchKernelGGL(blk_sum_leading_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const u32x4*)x, (u32x4*)y, T, n);
  return rsis_check_launch();
}

// db[row(c)] += sum over (b, pixel) of dy[b][c][pixel] for a blk dy [B][C/8][HW][8]; hid > 0: blk channel c = 4 j + gate goes to
// db[gate * hid + j] (the reference row order of a ConvLSTM bias, clstm.py:47).  Block (cb, split) reduces its slice of the
// B * HW cells of channel block cb and issues 8 atomics (deterministic mode: one split).
__global__ __launch_bounds__(256) void blk_channel_sum_kernel(const u32x4* __restrict__ dy, float* __restrict__ db, int B, int Cb, int HW, int hid,
                                                              int splits) {
  const int cb = blockIdx.x / splits, sp = blockIdx.x % splits;
  const long n = (long)B * HW;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, v[8];
  for (long i = (long)sp * 256 + threadIdx.x; i < n; i += (long)splits * 256) {
    const long b = i / HW;
    kd_unpack(dy[(b * Cb + cb) * HW + (i - b * HW)], v);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += v[k];
  }
  __shared__ float red[4][8];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float s = acc[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) red[wv][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int c = cb * 8 + threadIdx.x;
    const int row = hid > 0 ? (c & 3) * hid + (c >> 2) : c;
    atomicAdd(db + row, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}
int rsis_l_blk_channel_sum(const void* dy, float* db, int B, int C, int HW, int hid, hipStream_t st) {
  const int Cb = C >> 3;
  const long n = (long)B * HW;
  int splits = 1;
  if (!rsis_deterministic()) {
    splits = (int)((n + 2047) / 2048);
    const int cap = (1024 + Cb - 1) / Cb;
    if (splits > cap) splits = cap;
    if (splits < 1) splits = 1;
  }
  hipLaunchKernelGGL(blk_channel_sum_kernel, dim3(Cb * splits), dim3(256), 0, st, (const u32x4*)dy, db, B, Cb, HW, hid, splits);
  return rsis_check_launch();
}

// ------------------------------------------------------------------------------------------------
// conv_out on the blk hidden state of the last level (8 channels = ONE cell per pixel), all T timesteps in one launch.
// x / dx: [T * B][H][W][8] bf16 (image t * B + b); y / dy: fp32 [B][T][H * W] (image b * T + t, the (B, T, N) layout of train.py:118).
// w[72] = W[0][c][r][s] (reference layout), fp32.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ long c1b_ymap(long img, int B, int T) { return (img % B) * T + img / B; }

// forward: one thread = 4 consecutive pixels of a row (3 x 6 cells in, one float4 out)
__global__ __launch_bounds__(256) void blk_c1_fwd_kernel(const u32x4* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ y, int T, int B, int H, int W, long total4) {
  __shared__ float wl[72];
  if (threadIdx.x < 72) wl[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total4) return;
  const int W4 = W >> 2;
  const int xq = (int)(e % W4);
  const long t = e / W4;
  const int yy = (int)(t % H);
  const long img = t / H;
  const u32x4* xb = x + img * H * W;
  const float b0 = bias ? bias[0] : 0.f;
  float acc[4] = {b0, b0, b0, b0};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int iy = yy + r - 1;
    if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
    for (int c6 = 0; c6 < 6; ++c6) {
      const int ix = xq * 4 + c6 - 1;
      if ((unsigned)ix >= (unsigned)W) continue;
      float v[8];
      kd_unpack(xb[iy * W + ix], v);
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int s = c6 - o;               // tap column of output o that reads input column c6
        if (s < 0 || s > 2) continue;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[o] = fmaf(wl[c * 9 + r * 3 + s], v[c], acc[o]);
      }
    }
  }
  *reinterpret_cast<f32x4*>(y + c1b_ymap(img, B, T) * H * W + (long)yy * W + xq * 4) = f32x4{acc[0], acc[1], acc[2], acc[3]};
}

// data gradient: one thread = one pixel (9 dy floats in, one cell out)
__global__ __launch_bounds__(256) void blk_c1_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, u32x4* __restrict__ dx,
                                                           int T, int B, int H, int W, long total) {
  __shared__ float wl[72];
  if (threadIdx.x < 72) wl[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int xx = (int)(e % W);
  const long t = e / W;
  const int yy = (int)(t % H);
  const long img = t / H;
  const float* gb = dy + c1b_ymap(img, B, T) * H * W;
  float g[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int oy = yy + 1 - r, ox = xx + 1 - s;       // output pixel whose tap (r, s) reads this input pixel
      g[r * 3 + s] = ((unsigned)oy < (unsigned)H && (unsigned)ox < (unsigned)W) ? gb[oy * W + ox] : 0.f;
    }
  float o[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) a = fmaf(wl[c * 9 + k], g[k], a);
    o[c] = a;
  }
  dx[e] = kd_pack(o);
}

// weight + bias gradient: persistent blocks; a thread walks pixels, keeps the 72 + 1 sums, block reduction, one atomic per sum
__global__ __launch_bounds__(256) void blk_c1_wgrad_kernel(const float* __restrict__ dy, const u32x4* __restrict__ x, float* __restrict__ dw,
                                                           float* __restrict__ db, int T, int B, int H, int W, long total) {
  float acc[72];
#pragma unroll
  for (int i = 0; i < 72; ++i) acc[i] = 0.f;
  float gsum = 0.f;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int xx = (int)(e % W);
    const long t = e / W;
    const int yy = (int)(t % H);
    const long img = t / H;
    const float* gb = dy + c1b_ymap(img, B, T) * H * W;
    // dW[c][r][s] += sum over output pixels (oy, ox) of dy[oy][ox] * x[oy + r - 1][ox + s - 1][c]: this thread owns INPUT pixel (yy, xx)
    float v[8];
    kd_unpack(x[e], v);
    gsum += gb[yy * W + xx];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int oy = yy + 1 - r, ox = xx + 1 - s;
        const float gv = ((unsigned)oy < (unsigned)H && (unsigned)ox < (unsigned)W) ? gb[oy * W + ox] : 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c * 9 + r * 3 + s] = fmaf(gv, v[c], acc[c * 9 + r * 3 + s]);
      }
  }
  __shared__ float red[4][73];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i <= 72; ++i) {
    float v = i < 72 ? acc[i] : gsum;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[wv][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 72) atomicAdd(dw + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
  if (threadIdx.x == 72 && db) atomicAdd(db, red[0][72] + red[1][72] + red[2][72] + red[3][72]);
}

int rsis_l_blk_c1_fwd(const void* x, const float* w, const float* bias, float* y, int T, int B, int H, int W, hipStream_t st) {
  const long total4 = (long)T * B * H * (W >> 2);
  if ((total4 + 255) / 256 > 0x7FFFFFFFL) return RSIS_ERR_ARG;
  hipLaunchKernelGGL(blk_c1_fwd_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, (const u32x4*)x, w, bias, y, T, B, H, W, total4);
  return rsis_check_launch();
}
int rsis_l_blk_c1_dgrad(const float* dy, const float* w, void* dx, int T, int B, int H, int W, hipStream_t st) {
  const long total = (long)T * B * H * W;
  if ((total + 255) / 256 > 0x7FFFFFFFL) return RSIS_ERR_ARG;
  hipLaunchKernelGGL(blk_c1_dgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dy, w, (u32x4*)dx, T, B, H, W, total);
  return rsis_check_launch();
}
int rsis_l_blk_c1_wgrad(const float* dy, const void* x, float* dw, float* db, int T, int B, int H, int W, hipStream_t st) {
  const long total = (long)T * B * H * W;
  const int grid = rsis_deterministic() ? 1 : (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  hipLaunchKernelGGL(blk_c1_wgrad_kernel, dim3(grid), dim3(256), 0, st, dy, (const u32x4*)x, dw, db, T, B, H, W, total);
  return rsis_check_launch();
}
