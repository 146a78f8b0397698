// Bandwidth-bound kernels of the RSIS hot path for gfx950 (NCHW fp32): ConvLSTM pointwise backward, bilinear
// align-corners upsample fwd/bwd, global max-pool fwd/bwd, train/eval BatchNorm(+residual)(+ReLU) fwd/bwd,
// 3x3/2 max-pool fwd/bwd, per-channel bias-grad reduction and the flat fused Adam step (weight repacking: pack.hip).
// All are HBM-roofline kernels: coalesced along W, grid-stride, float4 where the row length allows it.
#include "common.h"
#include <cstdlib>

// ------------------------------------------------------------------------------------------------
// ConvLSTM pointwise backward (derivative of reference clstm.py:47-58; formulas in SURVEY.md 8(a))
//   do = dh*tanh(c); dc = dc_next + dh*o*(1-tanh(c)^2); di = dc*g; dg = dc*i; df = dc*c_prev; dc_prev = dc*f
//   da_i = di*i(1-i), da_f = df*f(1-f), da_o = do*o(1-o), da_g = dg*(1-g^2)
// act / da use gate-interleaved rows (4*j+gate).  Optionally accumulates da into da_sum (for the hoisted,
// time-invariant skip channels: sum_t da_t feeds ONE dgrad/wgrad per iteration).
// ------------------------------------------------------------------------------------------------
__global__ void lstm_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dh2, const float* __restrict__ dc_next,
                                const float* __restrict__ act, const float* __restrict__ c_prev,
                                const float* __restrict__ c, float* __restrict__ da, float* __restrict__ dc_prev,
                                float* __restrict__ da_sum, int hid, int HW, long total) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long bj = e / HW;               // b*hid + j
    const int sp = (int)(e - bj * HW);
    const long g0 = bj * 4 * HW + sp;     // (b*4*hid + 4*j)*HW + sp
    const float gi = act[g0], gf = act[g0 + HW], go = act[g0 + 2L * HW], gg = act[g0 + 3L * HW];
    const float tc = tanhf(c[e]);
    const float dhv = (dh ? dh[e] : 0.f) + (dh2 ? dh2[e] : 0.f);   // dh2: a second consumer's gradient (the recurrence)
    float dcv = dhv * go * (1.f - tc * tc);
    if (dc_next) dcv += dc_next[e];
    const float cp = c_prev ? c_prev[e] : 0.f;
    const float dai = dcv * gg * gi * (1.f - gi);
    const float daf = dcv * cp * gf * (1.f - gf);
    const float dao = dhv * tc * go * (1.f - go);
    const float dag = dcv * gi * (1.f - gg * gg);
    da[g0] = dai; da[g0 + HW] = daf; da[g0 + 2L * HW] = dao; da[g0 + 3L * HW] = dag;
    if (da_sum) { da_sum[g0] += dai; da_sum[g0 + HW] += daf; da_sum[g0 + 2L * HW] += dao; da_sum[g0 + 3L * HW] += dag; }
    if (dc_prev) dc_prev[e] = dcv * gf;
  }
}

// several independent cells in ONE grid (the cells of a reverse diagonal of the decoder's (level, timestep) wavefront): thread e of
// the grid belongs to the job whose [begin, begin + total) range holds it; same arithmetic per element as lstm_bwd_kernel
#define RSIS_LB_MAXJ 8
struct LstmBwdJobF {
  const float* dh; const float* dh2; const float* dc_next; const float* act; const float* c_prev; const float* c;
  float* da; float* dc_prev;
  int hid, HW;
};
struct LstmBwdGroupF {
  int n;
  long begin[RSIS_LB_MAXJ + 1];
  LstmBwdJobF job[RSIS_LB_MAXJ];
};
__global__ __launch_bounds__(256) void lstm_bwd_group_kernel(const LstmBwdGroupF g) {
  const long e0 = (long)blockIdx.x * 256 + threadIdx.x;
  if (e0 >= g.begin[g.n]) return;
  int j = 0;
#pragma unroll
  for (int k = 1; k < RSIS_LB_MAXJ; ++k) j += (k < g.n && g.begin[k] <= e0) ? 1 : 0;
  const LstmBwdJobF& p = g.job[j];
  const long e = e0 - g.begin[j];
  const int HW = p.HW;
  const long bj = e / HW;
  const int sp = (int)(e - bj * HW);
  const long g0 = bj * 4 * HW + sp;
  const float gi = p.act[g0], gf = p.act[g0 + HW], go = p.act[g0 + 2L * HW], gg = p.act[g0 + 3L * HW];
  const float tc = tanhf(p.c[e]);
  const float dhv = (p.dh ? p.dh[e] : 0.f) + (p.dh2 ? p.dh2[e] : 0.f);
  float dcv = dhv * go * (1.f - tc * tc);
  if (p.dc_next) dcv += p.dc_next[e];
  const float cp = p.c_prev ? p.c_prev[e] : 0.f;
  p.da[g0] = dcv * gg * gi * (1.f - gi);
  p.da[g0 + HW] = dcv * cp * gf * (1.f - gf);
  p.da[g0 + 2L * HW] = dhv * tc * go * (1.f - go);
  p.da[g0 + 3L * HW] = dcv * gi * (1.f - gg * gg);
  if (p.dc_prev) p.dc_prev[e] = dcv * gf;
}

// The same, four consecutive pixels per thread (every job's HW % 4 == 0 and 16-byte aligned tensors: the launcher checks) and NO load
// under a condition: an absent operand (dh, dh2, dc_next, c_prev) reads `c` instead and is selected away.  The scalar kernel's
// `p.dh ? p.dh[e] : 0` chain compiles to four branch + load + vmcnt(0) groups -- five dependent round trips per thread, one
// float each in flight: 3.9 TB/s on 4.2 GB per fp32 step.
__global__ __launch_bounds__(256) void lstm_bwd_group_v4_kernel(const LstmBwdGroupF g) {
  const long e0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e0 >= g.begin[g.n]) return;
  int j = 0;
#pragma unroll
  for (int k = 1; k < RSIS_LB_MAXJ; ++k) j += (k < g.n && g.begin[k] <= e0) ? 1 : 0;
  const LstmBwdJobF& p = g.job[j];
  const long e = e0 - g.begin[j];
  const int HW = p.HW;
  const long bj = e / HW;
  const int sp = (int)(e - bj * HW);
  const long g0 = bj * 4 * HW + sp;
  const f32x4 gi = *(const f32x4*)(p.act + g0), gf = *(const f32x4*)(p.act + g0 + HW);
  const f32x4 go = *(const f32x4*)(p.act + g0 + 2L * HW), gg = *(const f32x4*)(p.act + g0 + 3L * HW);
  const f32x4 cv = *(const f32x4*)(p.c + e);
  const f32x4 d1 = *(const f32x4*)((p.dh ? p.dh : p.c) + e), d2 = *(const f32x4*)((p.dh2 ? p.dh2 : p.c) + e);
  const f32x4 dn = *(const f32x4*)((p.dc_next ? p.dc_next : p.c) + e), cpv = *(const f32x4*)((p.c_prev ? p.c_prev : p.c) + e);
  // absent operands are SELECTED away (not multiplied by 0: 0 * Inf of a diverged cell state would poison terms the scalar kernel keeps
  // finite); the loads above stay unconditional
  const bool m1 = p.dh != nullptr, m2 = p.dh2 != nullptr, mn = p.dc_next != nullptr, mp = p.c_prev != nullptr;
  f32x4 dai, daf, dao, dag, dcp;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float tc = tanhf(cv[i]);
    const float dhv = (m1 ? d1[i] : 0.f) + (m2 ? d2[i] : 0.f);
    const float dcv = dhv * go[i] * (1.f - tc * tc) + (mn ? dn[i] : 0.f);
    const float cp = mp ? cpv[i] : 0.f;
    dai[i] = dcv * gg[i] * gi[i] * (1.f - gi[i]);
    daf[i] = dcv * cp * gf[i] * (1.f - gf[i]);
    dao[i] = dhv * tc * go[i] * (1.f - go[i]);
    dag[i] = dcv * gi[i] * (1.f - gg[i] * gg[i]);
    dcp[i] = dcv * gf[i];
  }
  *(f32x4*)(p.da + g0) = dai; *(f32x4*)(p.da + g0 + HW) = daf; *(f32x4*)(p.da + g0 + 2L * HW) = dao; *(f32x4*)(p.da + g0 + 3L * HW) = dag;
  if (p.dc_prev) *(f32x4*)(p.dc_prev + e) = dcp;
}

// ------------------------------------------------------------------------------------------------
// bilinear upsample, align_corners=True  (nn.UpsamplingBilinear2d: model.py:149,163; train.py:96; test.py:39)
// ------------------------------------------------------------------------------------------------
// (ac_coord / ac_scale: common.h)

__global__ void upsample_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int Hi, int Wi, int Ho, int Wo,
                                    float sh, float sw, long total) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int wo = (int)(e % Wo);
    const long t = e / Wo;
    const int ho = (int)(t % Ho);
    const long bc = t / Ho;
    int h0, h1, w0, w1; float lh, lw;
    ac_coord(ho, sh, Hi, h0, h1, lh);
    ac_coord(wo, sw, Wi, w0, w1, lw);
    const float* xb = x + bc * Hi * Wi;
    const float v00 = xb[h0 * Wi + w0], v01 = xb[h0 * Wi + w1], v10 = xb[h1 * Wi + w0], v11 = xb[h1 * Wi + w1];
    y[e] = (1.f - lh) * ((1.f - lw) * v00 + lw * v01) + lh * ((1.f - lw) * v10 + lw * v11);
  }
}

// Wo % 4 == 0: one thread = 4 consecutive outputs of a row (float4 store).  A block covers 256 / (Wo/4) consecutive output
// rows of the flattened (bc, ho) row space (or a 256-thread slice of one long row); the only division left is one per thread
// (row -> bc, ho).  Same arithmetic per element as upsample_fwd_kernel (bit-identical results).
__global__ __launch_bounds__(256) void upsample_fwd_v4_kernel(const float* __restrict__ x, float* __restrict__ y, int Hi, int Wi,
                                                              int Ho, int Wo, float sh, float sw, long rows, int Wq, int wq_shift) {
  // wq_shift >= 0: Wq is a power of two <= 256 and a block spans 256 >> wq_shift rows; -2: any Wq <= 256, the (row, float4)
  // space is walked flat (one division per thread: the 7 / 14 / 28-float4 rows of the 224 x 224 pyramid had one 256-thread block
  // per ROW before, 3-11 % of the lanes active: 22 us per launch instead of 6); otherwise blockIdx.y walks the row
  long row;
  int wq;
  if (wq_shift >= 0) {
    row = (long)blockIdx.x * (256 >> wq_shift) + (threadIdx.x >> wq_shift);
    wq = threadIdx.x & (Wq - 1);
  } else if (wq_shift == -2) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    row = e / Wq;
    wq = (int)(e - row * Wq);
  } else {
    row = blockIdx.x;
    wq = blockIdx.y * 256 + threadIdx.x;
  }
  if (row >= rows || wq >= Wq) return;
  const long bc = row / Ho;
  const int ho = (int)(row - bc * Ho);
  int h0, h1; float lh;
  ac_coord(ho, sh, Hi, h0, h1, lh);
  const float* r0 = x + (size_t)bc * Hi * Wi + h0 * Wi;
  const float* r1 = x + (size_t)bc * Hi * Wi + h1 * Wi;
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int w0, w1; float lw;
    ac_coord(wq * 4 + k, sw, Wi, w0, w1, lw);
    o[k] = (1.f - lh) * ((1.f - lw) * r0[w0] + lw * r0[w1]) + lh * ((1.f - lw) * r1[w0] + lw * r1[w1]);
  }
  *reinterpret_cast<f32x4*>(y + ((size_t)row * Wq + wq) * 4) = o;
}

// Large maps (Ho >= 32, Wo >= 64, Wi % 4 == 0, scale <~ 0.56): a block owns a 32 x 64 output tile of one plane, stages the
// <= 20 x 48 input pixels it interpolates from in LDS with (at most) one aligned float4 load per thread, and every thread
// writes two float4s.  The v4 kernel issues 16 scalar loads per float4 of output and is bound by the vector-memory pipe
// (34 us for the 8 x 256^2 x 32 level, a 67 MB write); this one is bound by the write.  Same arithmetic, bit-identical results.
// Tile shapes: 16 x 128 outputs (512-byte output rows; <= 12 x 69 inputs) on maps at least 128 wide, 32 x 64 (<= 20 x 45) below.
template <int UF_TOH, int UF_TOW, int UF_RH, int UF_RQ>
__global__ __launch_bounds__(256) void upsample_fwd_lds_kernel(const float* __restrict__ x, float* __restrict__ y, int Hi, int Wi,
                                                               int Ho, int Wo, float sh, float sw, int tiles_x, int tiles_y) {
  __shared__ __attribute__((aligned(16))) float src[UF_RH][UF_RQ * 4];
  const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y;
  const long bc = blockIdx.x / (tiles_x * tiles_y);
  const int oh0 = ty * UF_TOH, ow0 = tx * UF_TOW;
  int h_lo, w_lo, t0; float tl;
  ac_coord(oh0, sh, Hi, h_lo, t0, tl);
  ac_coord(ow0, sw, Wi, w_lo, t0, tl);
  const int wa = w_lo & ~3;
  {
    const int r = threadIdx.x / UF_RQ, q = threadIdx.x - r * UF_RQ;
    const int h = h_lo + r, col = wa + q * 4;
    if (r < UF_RH) {
      const bool ok = h < Hi && col < Wi;          // Wi % 4 == 0: a float4 is all in or all out
      f32x4 v = *reinterpret_cast<const f32x4*>(x + bc * Hi * Wi + (ok ? (size_t)h * Wi + col : 0));
      *reinterpret_cast<f32x4*>(&src[r][q * 4]) = v;
    }
  }
  __syncthreads();
  constexpr int QW = UF_TOW / 4, RPP = 256 / QW;          // float4 per tile row, tile rows per pass
  const int lr = threadIdx.x / QW, lq = threadIdx.x % QW;
#pragma unroll
  for (int p = 0; p < UF_TOH / RPP; ++p) {
    const int ho = oh0 + p * RPP + lr, wo = ow0 + lq * 4;
    if (ho >= Ho || wo >= Wo) continue;
    int h0, h1; float lh;
    ac_coord(ho, sh, Hi, h0, h1, lh);
    const float* r0 = src[h0 - h_lo];
    const float* r1 = src[h1 - h_lo];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int w0, w1; float lw;
      ac_coord(wo + k, sw, Wi, w0, w1, lw);
      w0 -= wa;
      w1 -= wa;
      o[k] = (1.f - lh) * ((1.f - lw) * r0[w0] + lw * r0[w1]) + lh * ((1.f - lw) * r1[w0] + lw * r1[w1]);
    }
    *reinterpret_cast<f32x4*>(y + (bc * Ho + ho) * (size_t)Wo + wo) = o;
  }
}

// backward as a GATHER (no atomics, deterministic): dx[hi][wi] = sum over the few output pixels whose 2x2 stencil touches
// (hi, wi).  A block owns a UT x UT tile of input pixels of one (b, c) plane.  The 1-D interpolation weights of the (<= UK)
// consecutive output rows / columns that touch each input row / column are evaluated ONCE per block (with the forward's own
// ac_coord, so membership is bit-consistent with it); the block then stages the dy region the tile needs in LDS with
// row-coalesced loads and reduces it separably: rows first (tmp[hi][wo]), then columns.
#define UK 6   // consecutive candidate outputs per input index: covers scale factors >= ~0.46 (checked by the launcher)
template <int UT, int RM>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int Hi, int Wi,
                                                           int Ho, int Wo, float sh, float sw, int tiles_x, int tiles_y, long BC,
                                                           int planes_per_block, int vec, const float* __restrict__ pool_g,
                                                           const int* __restrict__ pool_arg) {
  constexpr int RQ = RM / 4 + 1, RP = RQ * 4;      // staged row: RQ aligned float4s = RM + 4 columns (the region starts 0..3 columns in)
  __shared__ float wgt[2][UT][UK];
  __shared__ int st[2][UT];          // first candidate of each input index (absolute output index)
  __shared__ int org[2], ext[2];     // region origin / extent (rows, cols)
  __shared__ __attribute__((aligned(16))) float reg[RM * RP];
  __shared__ float tmp[UT * RP];
  const int tid = threadIdx.x;
  const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y;
  const long bc0 = (long)(blockIdx.x / (tiles_x * tiles_y)) * planes_per_block;
  // ---- per-block tables (shared by all planes of the block): threads -> (axis, local index) ----
  if (tid < 2 * UT) {
    const int axis = tid / UT, li = tid % UT;
    const int in = axis ? Wi : Hi, out = axis ? Wo : Ho;
    const float sc = axis ? sw : sh;
    const int i = (axis ? tx : ty) * UT + li;
    int l = max(0, (int)floorf((i - 1) / sc) - 1);
    int first = -1;
    float w[UK];
#pragma unroll
    for (int k = 0; k < UK; ++k) w[k] = 0.f;
    if (i < in) {
      for (int o = l; o < out && o < l + 4 + UK; ++o) {
        int i0, i1; float l1;
        ac_coord(o, sc, in, i0, i1, l1);
        const float wv = (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
        if (first < 0 && (i0 == i || i1 == i)) first = o;
        if (first >= 0 && o - first < UK) w[o - first] = wv;
      }
    }
    if (first < 0) first = min(l, out - 1);
#pragma unroll
    for (int k = 0; k < UK; ++k) wgt[axis][li][k] = w[k];
    st[axis][li] = first;
  }
  __syncthreads();
  if (tid < 2) {
    const int o0 = st[tid][0];
    int last = 0;
    for (int li = 0; li < UT; ++li) last = max(last, st[tid][li] + UK - 1);
    const int out = tid ? Wo : Ho;
    org[tid] = o0;
    ext[tid] = min(min(last, out - 1) - o0 + 1, RM);
  }
  __syncthreads();
  const int oy = org[0], ox = org[1], ey = ext[0], ex = ext[1];   // ey, ex <= RM (launcher)
  for (int pl = 0; pl < planes_per_block; ++pl) {
    const long bc = bc0 + pl;
    if (bc >= BC) break;
    const float* yb = dy + bc * Ho * Wo;
    int cs = 0;                                   // column of `reg` that holds region column 0
    if (vec) {
      // aligned float4 window [ox & ~3, ...): all loads of a thread are independent and issued together (the scalar loop below
      // waits for memory once per iteration: 50 us instead of 20 on the 256^2 level); columns outside [0, ex) are never read
      constexpr int NV = RM * RQ, ITER = (NV + 255) / 256;
      const int oxa = ox & ~3;
      cs = ox - oxa;
      f32x4 v[ITER];
#pragma unroll
      for (int k = 0; k < ITER; ++k) {
        const int i = tid + k * 256;
        const int r = i / RQ, q = i - r * RQ;
        const int col = oxa + q * 4;
        const bool ok = i < NV && r < ey && col < Wo;        // Wo % 4 == 0: a float4 is all in or all out
        v[k] = *reinterpret_cast<const f32x4*>(ok ? yb + (size_t)(oy + r) * Wo + col : yb);
      }
#pragma unroll
      for (int k = 0; k < ITER; ++k) {
        const int i = tid + k * 256;
        if (i < NV) *reinterpret_cast<f32x4*>(reg + i * 4) = v[k];
      }
    } else {
      for (int e = tid; e < ey * RM; e += 256) {
        const int r = e / RM, c = e - r * RM;
        reg[r * RP + c] = c < ex ? yb[(size_t)(oy + r) * Wo + ox + c] : 0.f;
      }
    }
    __syncthreads();
    // rows: tmp[ly][c] = sum_k wgt_y[ly][k] * reg[st_y[ly] - oy + k][c]
    for (int e = tid; e < UT * RM; e += 256) {
      const int ly = e / RM, c = e - ly * RM;
      const int r0 = st[0][ly] - oy;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < UK; ++k) acc = fmaf(wgt[0][ly][k], reg[min(r0 + k, ey - 1) * RP + cs + c], acc);   // weights beyond the region are zero
      tmp[ly * RP + c] = acc;
    }
    __syncthreads();
    for (int e = tid; e < UT * UT; e += 256) {
      const int ly = e / UT, lx = e - ly * UT;
      const int hi = ty * UT + ly, wi = tx * UT + lx;
      if (hi >= Hi || wi >= Wi) continue;
      const int c0 = st[1][lx] - ox;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < UK; ++k) acc = fmaf(wgt[1][lx][k], tmp[ly * RP + min(c0 + k, ex - 1)], acc);
      if (pool_g && hi * Wi + wi == pool_arg[bc]) acc += pool_g[bc];     // + the global max-pool's gradient at its arg-max pixel
      dx[bc * Hi * Wi + (size_t)hi * Wi + wi] = acc;
    }
    // (the next plane's `reg` fill is ordered after this plane's last `reg` read by the barrier above; `tmp` is rewritten only
    //  after the next barrier)
  }
}

// general fallback (any scale): per-thread candidate scan
__global__ void upsample_bwd_generic_kernel(const float* __restrict__ dy, float* __restrict__ dx, int Hi, int Wi, int Ho, int Wo,
                                            float sh, float sw, long total, const float* __restrict__ pool_g,
                                            const int* __restrict__ pool_arg) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int wi = (int)(e % Wi);
    const long t = e / Wi;
    const int hi = (int)(t % Hi);
    const long bc = t / Hi;
    int ho_lo = 0, ho_hi = Ho - 1, wo_lo = 0, wo_hi = Wo - 1;
    if (sh > 0.f) { ho_lo = max(0, (int)floorf((hi - 1) / sh) - 1); ho_hi = min(Ho - 1, (int)ceilf((hi + 1) / sh) + 1); }
    if (sw > 0.f) { wo_lo = max(0, (int)floorf((wi - 1) / sw) - 1); wo_hi = min(Wo - 1, (int)ceilf((wi + 1) / sw) + 1); }
    const float* yb = dy + bc * Ho * Wo;
    float acc = 0.f;
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
      int h0, h1; float lh;
      ac_coord(ho, sh, Hi, h0, h1, lh);
      const float wh = (h0 == hi ? 1.f - lh : 0.f) + (h1 == hi ? lh : 0.f);
      if (wh == 0.f) continue;
      float row = 0.f;
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        int w0, w1; float lw;
        ac_coord(wo, sw, Wi, w0, w1, lw);
        const float ww = (w0 == wi ? 1.f - lw : 0.f) + (w1 == wi ? lw : 0.f);
        if (ww != 0.f) row += ww * yb[ho * Wo + wo];
      }
      acc += wh * row;
    }
    if (pool_g && hi * Wi + wi == pool_arg[bc]) acc += pool_g[bc];
    dx[e] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// global spatial max-pool (nn.MaxPool2d(full map): model.py:143) + its backward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void global_maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 int* __restrict__ arg, int HW) {
  const long bc = blockIdx.x;
  const float* xb = x + bc * HW;
  float best = -INFINITY; int bi = 0x7fffffff;
  if ((HW & 3) == 0) {
    // float4 loads, 8 in flight per thread (one plane = one block: a loop with one load per iteration is pure memory latency,
    // 21 us for a 128 x 128 plane); a thread visits its pixels in increasing order, so `>` keeps the first maximum
    const f32x4* xv = reinterpret_cast<const f32x4*>(xb);
    const int n4 = HW >> 2;
    for (int g0 = threadIdx.x; g0 < n4; g0 += 256 * 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int g = g0 + u * 256;
        v[u] = xv[g < n4 ? g : g0];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int g = g0 + u * 256;
        if (g < n4) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (v[u][k] > best) { best = v[u][k]; bi = g * 4 + k; }
        }
      }
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += 256) {
      const float v = xb[i];
      if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
  }
  __shared__ float sv[256]; __shared__ int si[256];
  sv[threadIdx.x] = best; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float v = sv[threadIdx.x + s]; const int i = si[threadIdx.x + s];
      if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && i < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = i; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { y[bc] = sv[0]; arg[bc] = si[0] == 0x7fffffff ? 0 : si[0]; }
}

__global__ void global_maxpool_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ arg, float* __restrict__ dx,
                                          int HW, long total) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long bc = e / HW;
    const int sp = (int)(e - bc * HW);
    dx[e] = (sp == arg[bc]) ? dy[bc] : 0.f;
  }
}

// dx[bc][argmax[bc]] += dy[bc]: the max-pool gradient added into a gradient that already holds another path's contribution
__global__ void global_maxpool_bwd_add_kernel(const float* __restrict__ dy, const int* __restrict__ arg, float* __restrict__ dx,
                                              int HW, long BC) {
  const long bc = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (bc < BC) dx[bc * HW + arg[bc]] += dy[bc];
}

// ------------------------------------------------------------------------------------------------
// BatchNorm2d (stock semantics: eps, momentum, biased var for normalisation, unbiased for running_var)
//   fused with the optional residual add and ReLU of the ResNet bottleneck.
// stats[c] = {sum, sumsq} (train) or {sum g, sum g*xhat} (backward) accumulated in fp64.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_reduce2_atomic(double a, double b, double* dst) {
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); }
  __shared__ double sa[4], sb[4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sa[w] = a; sb[w] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(dst, sa[0] + sa[1] + sa[2] + sa[3]);
    atomicAdd(dst + 1, sb[0] + sb[1] + sb[2] + sb[3]);
  }
}

// Iteration scheme of the BN kernels: grid (C, S); block (c, s) walks channel c's B*HW elements in float4 groups
// (H*W % 4 == 0: a group never straddles an image) g = s*256 + tid, stride S*256, or element-wise otherwise.
// VEC = 4 or 1 elements per step; `body(idx, k)` sees the flat NCHW index of element k of the group.
#define BN_FOREACH(VEC, ...)                                                                    \
  {                                                                                             \
    const long ng = N / (VEC);                                                                  \
    const int hwg = HW / (VEC);                                                                 \
    for (long g = blockIdx.y * 256L + threadIdx.x; g < ng; g += gridDim.y * 256L) {            \
      const long b = g / hwg;                                                                   \
      const int sp = (int)(g - b * hwg) * (VEC);                                                \
      const long idx = (b * C + c) * HW + sp;                                                   \
      __VA_ARGS__                                                                               \
    }                                                                                           \
  }

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, double* __restrict__ stats, int C, int HW,
                                                       long N) {
  const int c = blockIdx.x;
  double s = 0.0, ss = 0.0;
  if ((HW & 3) == 0) {
    BN_FOREACH(4, {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + idx);
      const float ps = (v[0] + v[1]) + (v[2] + v[3]);
      const float pq = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      s += ps; ss += pq;
    })
  } else {
    BN_FOREACH(1, { const float v = x[idx]; s += v; ss += (double)v * v; })
  }
  block_reduce2_atomic(s, ss, stats + 2 * c);
}

// mode 0: train (batch stats from `stats`, updates running stats, saves mean/rstd); mode 1: eval (running stats)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                       float* __restrict__ y, const double* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ run_mean, float* __restrict__ run_var,
                                                       float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                       int C, int HW, long N, float eps, float momentum, int relu, int mode) {
  const int c = blockIdx.x;
  float mean, rstd;
  if (mode == 0) {
    const double m = stats[2 * c] / (double)N;
    double var = stats[2 * c + 1] / (double)N - m * m;
    if (var < 0) var = 0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (blockIdx.y == 0 && threadIdx.x == 0) {
      save_mean[c] = mean; save_rstd[c] = rstd;
      const double unb = N > 1 ? var * (double)N / (double)(N - 1) : var;
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
    }
  } else {
    mean = run_mean[c];
    rstd = rsis_bn_eval_rstd(run_var[c], eps);
  }
  float sc, sh;
  rsis_bn_affine(rstd, gamma[c], beta[c], mean, sc, sh);
  if ((HW & 3) == 0) {
    BN_FOREACH(4, {
      f32x4 v = *reinterpret_cast<const f32x4*>(x + idx);
      f32x4 r = {0.f, 0.f, 0.f, 0.f};
      if (res) r = *reinterpret_cast<const f32x4*>(res + idx);
      for (int k = 0; k < 4; ++k) v[k] = rsis_bn_apply(v[k], sc, sh, r[k], relu);
      *reinterpret_cast<f32x4*>(y + idx) = v;
    })
  } else {
    BN_FOREACH(1, { y[idx] = rsis_bn_apply(x[idx], sc, sh, res ? res[idx] : 0.f, relu); })
  }
}

// backward pass 1: stats[c] = {sum g, sum g*xhat},  g = dy * (y > 0 if relu)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, double* __restrict__ stats,
                                                            int C, int HW, long N, int relu, float eval_eps) {
  const int c = blockIdx.x;
  // eval_eps >= 0: eval-mode BatchNorm -- `mean` / `rstd` hold the running mean / running VARIANCE
  const float m = mean[c], r = eval_eps >= 0.f ? 1.f / sqrtf(rstd[c] + eval_eps) : rstd[c];
  double s1 = 0.0, s2 = 0.0;
  if ((HW & 3) == 0) {
    BN_FOREACH(4, {
      f32x4 g = *reinterpret_cast<const f32x4*>(dy + idx);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + idx);
      if (relu) {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + idx);
        for (int k = 0; k < 4; ++k) if (!(yv[k] > 0.f)) g[k] = 0.f;
      }
      float p1 = 0.f, p2 = 0.f;
      for (int k = 0; k < 4; ++k) { p1 += g[k]; p2 += g[k] * ((xv[k] - m) * r); }
      s1 += p1; s2 += p2;
    })
  } else {
    BN_FOREACH(1, {
      float g = dy[idx];
      if (relu && !(y[idx] > 0.f)) g = 0.f;
      s1 += g; s2 += (double)g * ((x[idx] - m) * r);
    })
  }
  block_reduce2_atomic(s1, s2, stats + 2 * c);
}

// backward pass 2: dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)); dres = g; dgamma/dbeta from the sums
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const double* __restrict__ stats, float* __restrict__ dx,
                                                           float* __restrict__ dres, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int C, int HW, long N, int relu,
                                                           int accum, float eval_eps) {
  const int c = blockIdx.x;
  const bool ev = eval_eps >= 0.f;      // eval mode: the statistics are constants, dx = gamma * rstd * g (no mean terms)
  const float m = mean[c], r = ev ? 1.f / sqrtf(rstd[c] + eval_eps) : rstd[c];
  const float mg = ev ? 0.f : (float)(stats[2 * c] / (double)N), mgx = ev ? 0.f : (float)(stats[2 * c + 1] / (double)N);
  const float kk = gamma[c] * r;
  if (blockIdx.y == 0 && threadIdx.x == 0) {
    const float dg = (float)stats[2 * c + 1], db = (float)stats[2 * c];
    dgamma[c] = accum ? dgamma[c] + dg : dg;
    dbeta[c] = accum ? dbeta[c] + db : db;
  }
  if ((HW & 3) == 0) {
    BN_FOREACH(4, {
      f32x4 g = *reinterpret_cast<const f32x4*>(dy + idx);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + idx);
      if (relu) {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + idx);
        for (int k = 0; k < 4; ++k) if (!(yv[k] > 0.f)) g[k] = 0.f;
      }
      f32x4 o;
      for (int k = 0; k < 4; ++k) o[k] = kk * (g[k] - mg - ((xv[k] - m) * r) * mgx);
      *reinterpret_cast<f32x4*>(dx + idx) = o;
      if (dres) *reinterpret_cast<f32x4*>(dres + idx) = g;
    })
  } else {
    BN_FOREACH(1, {
      float g = dy[idx];
      if (relu && !(y[idx] > 0.f)) g = 0.f;
      const float xh = (x[idx] - m) * r;
      dx[idx] = kk * (g - mg - xh * mgx);
      if (dres) dres[idx] = g;
    })
  }
}
#undef BN_FOREACH

// ------------------------------------------------------------------------------------------------
// Register-resident BatchNorm for the layers whose channel plane set fits one block (B*HW/4 <= NT*VPT float4 groups: layers
// 2-4 of the trunk and the deep skip BNs at batch 32): ONE launch and ONE read of each input per pass instead of a reduce
// launch + an apply launch that re-reads everything.  Block c holds channel c's values in registers, reduces them
// (fp64 partial sums), then normalises / back-propagates from the registers.  Same arithmetic as the split kernels above.
// ------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void block_allreduce2(double& a, double& b) {
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); }
  __shared__ double sa[NT / 64], sb[NT / 64];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sa[w] = a; sb[w] = b; }
  __syncthreads();
  a = 0.0; b = 0.0;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) { a += sa[i]; b += sb[i]; }
}

template <int NT, int VPT>
__global__ __launch_bounds__(NT) void bn_fwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                          float* __restrict__ y, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ run_mean,
                                                          float* __restrict__ run_var, float* __restrict__ save_mean,
                                                          float* __restrict__ save_rstd, int C, int HW, long N, float eps,
                                                          float momentum, int relu) {
  const int c = blockIdx.x;
  const int ng = (int)(N / 4), hwg = HW / 4;
  f32x4 v[VPT];
  double s = 0.0, ss = 0.0;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int g = threadIdx.x + i * NT;
    if (g < ng) {
      const int b = g / hwg, sp = (g - b * hwg) * 4;
      v[i] = *reinterpret_cast<const f32x4*>(x + ((size_t)b * C + c) * HW + sp);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      ss += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
    }
  }
  block_allreduce2<NT>(s, ss);
  const double m = s / (double)N;
  double var = ss / (double)N - m * m;
  if (var < 0) var = 0;
  const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) {
    save_mean[c] = mean; save_rstd[c] = rstd;
    const double unb = N > 1 ? var * (double)N / (double)(N - 1) : var;
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
  }
  const float sc = rstd * gamma[c], sh = beta[c] - mean * sc;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int g = threadIdx.x + i * NT;
    if (g < ng) {
      const int b = g / hwg, sp = (g - b * hwg) * 4;
      const size_t idx = ((size_t)b * C + c) * HW + sp;
      f32x4 r = {0.f, 0.f, 0.f, 0.f};
      if (res) r = *reinterpret_cast<const f32x4*>(res + idx);
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float t = v[i][k] * sc + sh + r[k]; o[k] = relu ? fmaxf(t, 0.f) : t; }
      *reinterpret_cast<f32x4*>(y + idx) = o;
    }
  }
}

template <int NT, int VPT>
__global__ __launch_bounds__(NT) void bn_bwd_fused_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ y, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          float* __restrict__ dx, float* __restrict__ dres,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int C, int HW,
                                                          long N, int relu, int accum) {
  const int c = blockIdx.x;
  const int ng = (int)(N / 4), hwg = HW / 4;
  const float m = mean[c], r = rstd[c];
  f32x4 g[VPT], xh[VPT];
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int gi = threadIdx.x + i * NT;
    if (gi < ng) {
      const int b = gi / hwg, sp = (gi - b * hwg) * 4;
      const size_t idx = ((size_t)b * C + c) * HW + sp;
      g[i] = *reinterpret_cast<const f32x4*>(dy + idx);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + idx);
      if (relu) {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + idx);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (!(yv[k] > 0.f)) g[i][k] = 0.f;
      }
      float p1 = 0.f, p2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { xh[i][k] = (xv[k] - m) * r; p1 += g[i][k]; p2 += g[i][k] * xh[i][k]; }
      s1 += p1; s2 += p2;
    }
  }
  block_allreduce2<NT>(s1, s2);
  const float mg = (float)(s1 / (double)N), mgx = (float)(s2 / (double)N);
  const float kk = gamma[c] * r;
  if (threadIdx.x == 0) {
    const float dg = (float)s2, db = (float)s1;
    dgamma[c] = accum ? dgamma[c] + dg : dg;
    dbeta[c] = accum ? dbeta[c] + db : db;
  }
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int gi = threadIdx.x + i * NT;
    if (gi < ng) {
      const int b = gi / hwg, sp = (gi - b * hwg) * 4;
      const size_t idx = ((size_t)b * C + c) * HW + sp;
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = kk * (g[i][k] - mg - xh[i][k] * mgx);
      *reinterpret_cast<f32x4*>(dx + idx) = o;
      if (dres) *reinterpret_cast<f32x4*>(dres + idx) = g[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// y[bc][ho][wo] = x[bc][ho * s][wo * s]: the input of a 1x1 / stride-s conv (the downsample convs of ResNet layers 2-4, torchvision
// Bottleneck; reference vision.py:16-19) as a dense map, so that the conv itself runs as a stride-1 GEMM on the LDS-DMA path and
// the same copy serves its weight gradient.  One thread per 4 output pixels of a row: s = 2 reads two float4 (every other element
// of 8 consecutive inputs), writes one float4.
// ------------------------------------------------------------------------------------------------
__global__ void subsample_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int Ho, int Wo, int s, long total4) {
  const int wq = (Wo + 3) / 4;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total4; e += (long)gridDim.x * blockDim.x) {
    const int q = (int)(e % wq);
    const long t = e / wq;
    const int ho = (int)(t % Ho);
    const long bc = t / Ho;
    const float* xr = x + (bc * H + (long)ho * s) * W;
    float* yr = y + (bc * Ho + ho) * (long)Wo + 4 * q;
    if (s == 2 && (W & 3) == 0 && 4 * q + 3 < Wo && 8 * q + 7 < W) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 8 * q), b = *reinterpret_cast<const f32x4*>(xr + 8 * q + 4);
      const f32x4 o = {a[0], a[2], b[0], b[2]};
      if ((Wo & 3) == 0) *reinterpret_cast<f32x4*>(yr) = o;
      else { yr[0] = o[0]; yr[1] = o[1]; yr[2] = o[2]; yr[3] = o[3]; }
    } else {
      for (int k = 0; k < 4 && 4 * q + k < Wo; ++k) yr[k] = xr[(long)(4 * q + k) * s];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(3, stride 2, pad 1) of the ResNet stem (torchvision; reference vision.py:15)
// ------------------------------------------------------------------------------------------------
__global__ void maxpool3x3s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ arg,
                                        int H, int W, int Ho, int Wo, long total) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int wo = (int)(e % Wo);
    const long t = e / Wo;
    const int ho = (int)(t % Ho);
    const long bc = t / Ho;
    const float* xb = x + bc * H * W;
    float v[9];
    bool ok[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {            // branch-free: all 9 loads are issued together (clamped address, masked value)
        const int h = ho * 2 - 1 + r, w = wo * 2 - 1 + s;
        ok[r * 3 + s] = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        v[r * 3 + s] = xb[min(max(h, 0), H - 1) * W + min(max(w, 0), W - 1)];
      }
    float best = -INFINITY; int bi = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k)
      if (ok[k] && (v[k] > best || v[k] != v[k])) { best = v[k]; bi = k; }
    y[e] = best; arg[e] = (unsigned char)bi;
  }
}

__global__ void maxpool3x3s2_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg,
                                        float* __restrict__ dx, int H, int W, int Ho, int Wo, long total, int accumulate) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int w = (int)(e % W);
    const long t = e / W;
    const int h = (int)(t % H);
    const long bc = t / H;
    float g = 0.f;
    const int ho_lo = h >> 1, wo_lo = w >> 1;            // windows (2*ho-1 .. 2*ho+1) containing h: ho in {h/2, (h+1)/2}
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {                      // branch-free: the 4 (arg, dy) pairs are loaded together
        const int ho = ho_lo + a, wo = wo_lo + b;
        const int r = h - (ho * 2 - 1), s = w - (wo * 2 - 1);
        const bool ok = ho < Ho && r >= 0 && r <= 2 && wo < Wo && s >= 0 && s <= 2;
        const long o = (bc * Ho + min(ho, Ho - 1)) * Wo + min(wo, Wo - 1);
        const int ar = arg[o];
        const float d = dy[o];
        g += (ok && ar == r * 3 + s) ? d : 0.f;
      }
    }
    if (accumulate) g += dx[e];
    dx[e] = g;
  }
}

// H even, W % 4 == 0: one thread = a 2 x 4 block of dx.  The 6 pooling windows that can point into it (2 rows x 3 columns of
// the pooled map) are loaded together and scattered in registers: 12 loads + 2 float4 stores per 8 pixels instead of 8 + 1 per
// pixel (the per-pixel kernel is bound by the number of memory instructions, 134 us on the stem's 128^2 x 64 x 32 map).
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_v8_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg,
                                                                  float* __restrict__ dx, int H, int W, int Ho, int Wo, long items,
                                                                  int accumulate) {
  const int Wq = W >> 2, Hp = H >> 1;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < items; e += (long)gridDim.x * blockDim.x) {
    const int wq = (int)(e % Wq);
    const long t = e / Wq;
    const int hp = (int)(t % Hp);
    const long bc = t / Hp;
    const int h0 = hp * 2, w0 = wq * 4;
    float d[6];
    int th[6], tw[6];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int ho = hp + a, wo = wq * 2 + b;
        const bool ok = ho < Ho && wo < Wo;
        const long o = (bc * Ho + min(ho, Ho - 1)) * Wo + min(wo, Wo - 1);
        const int ar = arg[o];
        d[a * 3 + b] = dy[o];
        th[a * 3 + b] = ok ? ho * 2 - 1 + ar / 3 - h0 : -1;       // row / column of the arg-max pixel inside the 2 x 4 block
        tw[a * 3 + b] = wo * 2 - 1 + ar % 3 - w0;
      }
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o0[j] += (th[k] == 0 && tw[k] == j) ? d[k] : 0.f;
        o1[j] += (th[k] == 1 && tw[k] == j) ? d[k] : 0.f;
      }
    float* p = dx + (bc * H + h0) * W + w0;
    if (accumulate) {                          // dx already holds the gradient of another consumer of the pooled tensor's input
      o0 += *reinterpret_cast<const f32x4*>(p);
      o1 += *reinterpret_cast<const f32x4*>(p + W);
    }
    *reinterpret_cast<f32x4*>(p) = o0;
    *reinterpret_cast<f32x4*>(p + W) = o1;
  }
}

// ------------------------------------------------------------------------------------------------
// per-channel sum over (B, HW): conv bias gradient.  hid>0: interleaved rows -> reference rows. Accumulates.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ dy, float* __restrict__ db, int C, int HW,
                                                          long N, int hid) {
  const int c = blockIdx.x;
  double s = 0.0;
  if ((HW & 3) == 0) {                       // float4 groups never straddle an image
    const long ng = N / 4;
    const int hwg = HW / 4;
    for (long g = blockIdx.y * 256L + threadIdx.x; g < ng; g += gridDim.y * 256L) {
      const long b = g / hwg; const int sp = (int)(g - b * hwg) * 4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(dy + (b * C + c) * HW + sp);
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
  } else {
    for (long e = blockIdx.y * 256L + threadIdx.x; e < N; e += gridDim.y * 256L) {
      const long b = e / HW; const int sp = (int)(e - b * HW);
      s += dy[(b * C + c) * HW + sp];
    }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  __shared__ double sa[4];
  if ((threadIdx.x & 63) == 0) sa[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int row = hid > 0 ? (c & 3) * hid + (c >> 2) : c;
    atomicAdd(db + row, (float)(sa[0] + sa[1] + sa[2] + sa[3]));
  }
}

// ------------------------------------------------------------------------------------------------
// flat fused Adam (torch.optim.Adam semantics incl. L2 weight decay: reference utils/utils.py:83-84)
// ------------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float lr, float b1, float b2, float eps, float wd, int step, float gscale,
                            const int* __restrict__ step_dev) {
  // step count from the argument, or kept on the device (a captured hipGraph replays with the live count, not the captured one);
  // the bias corrections are computed HERE in both cases, so that an eager step and a replayed one are the same arithmetic
  const float st = (float)(step_dev ? *step_dev : step);
  const float bc1 = 1.f - powf(b1, st);
  const float bc2_sqrt = sqrtf(1.f - powf(b2, st));
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    float gv = g[e] * gscale + wd * p[e];
    const float mv = b1 * m[e] + (1.f - b1) * gv;
    const float vv = b2 * v[e] + (1.f - b2) * gv * gv;
    m[e] = mv; v[e] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[e] -= (lr / bc1) * (mv / denom);
  }
}

// ------------------------------------------------------------------------------------------------
// Minimum-cost assignment of predictions to ground-truth slots (reference src/utils/hungarian.py:91-125: Munkres per
// sample on the host, with a D2H copy of the scores and a host sync in the middle of every iteration).  One thread per
// sample runs the O(T^2 G) shortest-augmenting-path Hungarian algorithm with potentials in fp64 on the device, so the
// training step has no host synchronisation at all.  scores[b][g][t]: rows = GT slots, columns = predictions (G >= T);
// perm[b][t] = GT slot assigned to prediction t, perm[b][t >= T] = 0 (what hungarian.py leaves in unassigned columns).
// ------------------------------------------------------------------------------------------------
#define RSIS_ASSIGN_MAX 64
// One 64-lane wave per sample: lane j owns GT slot ("column") j+1 of the classical algorithm -- its potential v, the
// running minimum minv, used / way / p -- and lane i owns the potential u of prediction ("row") i+1; the inner scan over
// columns is a wave-wide arg-min (first minimum wins, as in the sequential scan), row lookups are lane shuffles.
__global__ __launch_bounds__(64) void assign_kernel(const float* __restrict__ scores, long long* __restrict__ perm, int B, int G,
                                                    int T) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const float* a = scores + (size_t)b * G * T;   // a[g*T + t]
  const int n = T, m = G;
  const bool col = lane < m;
  double u = 0.0, v = 0.0;
  int p = 0;                                       // row assigned to my column (0 = free)
  for (int i = 1; i <= n; ++i) {
    const int p0 = i;                              // row of the virtual column 0
    int j0 = 0;
    double minv = 1e300;
    bool used = false, in_tree = false;
    int way = 0;
    // (at most m + 1 columns can enter the tree; the cap only matters for non-finite scores -- a NaN cost would otherwise cycle
    // forever between used columns -- which are mapped to a large finite cost below)
    for (int it = 0; it <= m; ++it) {
      if (j0 > 0 && lane == j0 - 1) used = true;
      const int i0 = j0 == 0 ? p0 : __shfl(p, j0 - 1, 64);
      if (lane == i0 - 1) in_tree = true;
      const double ui0 = __shfl(u, i0 - 1, 64);
      double cand = 1e300;
      if (col && !used) {
        const float sc = a[(size_t)lane * T + (i0 - 1)];
        const double cur = (sc == sc && fabsf(sc) < 1e30f ? (double)sc : 1e30) - ui0 - v;
        if (cur < minv) { minv = cur; way = j0; }
        cand = minv;
      }
      // wave arg-min, smallest lane on ties
      double best = cand;
      int bj = lane;
      for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64);
        const int oj = __shfl_xor(bj, o, 64);
        if (ob < best || (ob == best && oj < bj)) { best = ob; bj = oj; }
      }
      const double delta = best;
      if (in_tree) u += delta;
      if (col) { if (used) v -= delta; else minv -= delta; }
      j0 = bj + 1;
      if (__shfl(p, j0 - 1, 64) == 0) break;
    }
    // augment along the alternating path
    while (j0 != 0) {
      const int j1 = __shfl(way, j0 - 1, 64);
      const int pj1 = j1 == 0 ? p0 : __shfl(p, j1 - 1, 64);
      if (lane == j0 - 1) p = pj1;
      j0 = j1;
    }
  }
  // perm[b][t] = column of row t+1; zeros elsewhere
  __shared__ long long outp[RSIS_ASSIGN_MAX];
  if (lane < G) outp[lane] = 0;
  __syncthreads();
  if (col && p != 0) outp[p - 1] = lane;
  __syncthreads();
  if (lane < G) perm[(size_t)b * G + lane] = outp[lane];
}

// zero fill (see common.h: rsis_zero_async)
__global__ void zero_fill_kernel(unsigned* __restrict__ p, size_t n_words) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
int rsis_zero_async(void* p, size_t bytes, hipStream_t st) {
  if (bytes == 0) return RSIS_OK;
  if ((bytes & 3) || ((size_t)p & 3)) return hipMemsetAsync(p, 0, bytes, st) == hipSuccess ? RSIS_OK : RSIS_ERR_LAUNCH;
  const size_t n = bytes >> 2;
  size_t g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)g), dim3(256), 0, st, (unsigned*)p, n);
  return rsis_check_launch();
}

// ------------------------------------------------------------------------------------------------
// launchers (called from api.hip)
// ------------------------------------------------------------------------------------------------
static inline int ew_grid(long total) {
  long g = (total + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}
// register-resident single-launch BatchNorm: float4-able planes, the whole channel (B*HW values) in one block's registers, and
// enough channels to fill the chip (RSIS_BN_FUSED=0 forces the split kernels)
static inline bool bn_fused_ok(int C, int HW, long N) {
  static const bool on = !(getenv("RSIS_BN_FUSED") && getenv("RSIS_BN_FUSED")[0] == '0');
  return on && (HW & 3) == 0 && N / 4 <= 1024 * 8 && C >= 64;
}
static inline int chan_splits(int C, long N) {
  if (rsis_deterministic()) return 1;        // all of a channel in one block: no cross-block atomics
  long s = (2048 + C - 1) / C;
  const long maxs = (N + 1023) / 1024;
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  return (int)s;
}

int rsis_l_lstm_bwd(const float* dh, const float* dh2, const float* dc_next, const float* act, const float* c_prev, const float* c, float* da,
                    float* dc_prev, float* da_sum, int B, int hid, int HW, hipStream_t st) {
  const long total = (long)B * hid * HW;
  hipLaunchKernelGGL(lstm_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, st, dh, dh2, dc_next, act, c_prev, c, da, dc_prev,
                     da_sum, hid, HW, total);
  return rsis_check_launch();
}
// jobs[j] = {dh, dh2, dc_next, act, c_prev, c, da, dc_prev} pointers, dims[j] = {B, hid, HW}
int rsis_l_lstm_bwd_group(const void* const* ptrs, const int* dims, int n, hipStream_t st) {
  for (int j0 = 0; j0 < n; j0 += RSIS_LB_MAXJ) {
    const int m = n - j0 < RSIS_LB_MAXJ ? n - j0 : RSIS_LB_MAXJ;
    LstmBwdGroupF g;
    g.n = m;
    long tot = 0;
    for (int k = 0; k < m; ++k) {
      const void* const* q = ptrs + (size_t)(j0 + k) * 8;
      const int* d = dims + (size_t)(j0 + k) * 3;
      LstmBwdJobF& a = g.job[k];
      a.dh = (const float*)q[0]; a.dh2 = (const float*)q[1]; a.dc_next = (const float*)q[2]; a.act = (const float*)q[3];
      a.c_prev = (const float*)q[4]; a.c = (const float*)q[5]; a.da = (float*)q[6]; a.dc_prev = (float*)q[7];
      a.hid = d[1]; a.HW = d[2];
      g.begin[k] = tot;
      tot += (long)d[0] * d[1] * d[2];
    }
    for (int k = m; k <= RSIS_LB_MAXJ; ++k) g.begin[k] = tot;
    if ((tot + 255) / 256 > 0x7FFFFFFFL) return RSIS_ERR_ARG;
    bool v4 = true;            // four pixels per thread: whole float4s of every tensor of every job
    for (int k = 0; k < m; ++k) {
      const LstmBwdJobF& a = g.job[k];
      const void* q[8] = {a.dh, a.dh2, a.dc_next, a.act, a.c_prev, a.c, a.da, a.dc_prev};
      if (a.HW & 3) v4 = false;
      for (int i = 0; i < 8; ++i) if (q[i] && ((size_t)q[i] & 15)) v4 = false;
    }
    if (v4) {
      hipLaunchKernelGGL(lstm_bwd_group_v4_kernel, dim3((unsigned)((tot / 4 + 255) / 256)), dim3(256), 0, st, g);
      if (rsis_check_launch() != RSIS_OK) return RSIS_ERR_LAUNCH;
      continue;
    }
    hipLaunchKernelGGL(lstm_bwd_group_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, g);
    if (rsis_check_launch() != RSIS_OK) return RSIS_ERR_LAUNCH;
  }
  return RSIS_OK;
}
int rsis_l_upsample_fwd(const float* x, float* y, long BC, int Hi, int Wi, int Ho, int Wo, hipStream_t st) {
  const long total = BC * Ho * Wo;
  {
    // the LDS-tiled kernel needs its output tile to read at most RH x (RQ * 4 - 3) input pixels
    const float fh = ac_scale(Hi, Ho), fw = ac_scale(Wi, Wo);
    const bool al = Wo % 4 == 0 && Wi % 4 == 0;
    if (al && Ho >= 16 && Wo >= 128 && fh * 15 + 3.f <= 12 && fw * 127 + 6.f <= 72 && BC * ((Wo + 127) / 128) * ((Ho + 15) / 16) < (1L << 31)) {
      const int tiles_x = (Wo + 127) / 128, tiles_y = (Ho + 15) / 16;
      hipLaunchKernelGGL((upsample_fwd_lds_kernel<16, 128, 12, 18>), dim3((unsigned)(BC * tiles_x * tiles_y)), dim3(256), 0, st, x, y, Hi,
                         Wi, Ho, Wo, fh, fw, tiles_x, tiles_y);
      return rsis_check_launch();
    }
    if (al && Ho >= 32 && Wo >= 64 && fh * 31 + 3.f <= 20 && fw * 63 + 6.f <= 48 && BC * ((Wo + 63) / 64) * ((Ho + 31) / 32) < (1L << 31)) {
      const int tiles_x = (Wo + 63) / 64, tiles_y = (Ho + 31) / 32;
      hipLaunchKernelGGL((upsample_fwd_lds_kernel<32, 64, 20, 12>), dim3((unsigned)(BC * tiles_x * tiles_y)), dim3(256), 0, st, x, y, Hi,
                         Wi, Ho, Wo, fh, fw, tiles_x, tiles_y);
      return rsis_check_launch();
    }
  }
  if (Wo % 4 == 0 && BC * Ho < (1L << 31)) {
    const int Wq = Wo / 4;
    const long rows = BC * Ho;
    int shift = -1;
    if (Wq <= 256 && (Wq & (Wq - 1)) == 0) { shift = 0; while ((1 << shift) < Wq) ++shift; }
    if (shift < 0 && Wq <= 256) shift = -2;
    const dim3 grid = shift >= 0 ? dim3((unsigned)((rows + (256 >> shift) - 1) / (256 >> shift)))
                                 : (shift == -2 ? dim3((unsigned)((rows * Wq + 255) / 256)) : dim3((unsigned)rows, (Wq + 255) / 256));
    hipLaunchKernelGGL(upsample_fwd_v4_kernel, grid, dim3(256), 0, st, x, y, Hi, Wi, Ho, Wo, ac_scale(Hi, Ho), ac_scale(Wi, Wo), rows,
                       Wq, shift);
  } else {
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, st, x, y, Hi, Wi, Ho, Wo, ac_scale(Hi, Ho),
                       ac_scale(Wi, Wo), total);
  }
  return rsis_check_launch();
}
int rsis_l_upsample_bwd(const float* dy, float* dx, long BC, int Hi, int Wi, int Ho, int Wo, const float* pool_g, const int* pool_arg,
                        hipStream_t st) {
  const float sh = ac_scale(Hi, Ho), sw = ac_scale(Wi, Wo);
  // the tiled kernel keeps UK = 6 consecutive candidate outputs per input index (< 2/scale + 1 touch it) and a dy region of
  // at most (UT + 1)/scale + UK outputs per axis in LDS
  const float smin = sh < sw ? sh : sw;
  const bool tiled = smin > 0.f && (2.f / smin + 1.f <= UK);
  const int ut = (Hi > 16 || Wi > 16) ? 32 : 16;
  const bool fits = tiled && ((ut - 1) / smin + UK + 3 <= (ut == 32 ? 80 : 44));
  if (fits) {
    const int tiles_x = (Wi + ut - 1) / ut, tiles_y = (Hi + ut - 1) / ut;
    // several planes per block (the interpolation tables depend on the tile only), keeping >= ~2048 blocks
    long ppb = BC * tiles_x * tiles_y / 2048;
    if (ppb < 1) ppb = 1;
    if (ppb > 16) ppb = 16;
    const long blocks = (BC + ppb - 1) / ppb * tiles_x * tiles_y;
    const int vec = Wo % 4 == 0;
    if (ut == 32)
      hipLaunchKernelGGL((upsample_bwd_kernel<32, 80>), dim3((unsigned)blocks), dim3(256), 0, st, dy, dx, Hi, Wi, Ho, Wo, sh, sw, tiles_x,
                         tiles_y, BC, (int)ppb, vec, pool_g, pool_arg);
    else
      hipLaunchKernelGGL((upsample_bwd_kernel<16, 44>), dim3((unsigned)blocks), dim3(256), 0, st, dy, dx, Hi, Wi, Ho, Wo, sh, sw, tiles_x,
                         tiles_y, BC, (int)ppb, vec, pool_g, pool_arg);
  } else {
    const long total = BC * Hi * Wi;
    hipLaunchKernelGGL(upsample_bwd_generic_kernel, dim3(ew_grid(total)), dim3(256), 0, st, dy, dx, Hi, Wi, Ho, Wo, sh, sw, total, pool_g,
                       pool_arg);
  }
  return rsis_check_launch();
}
int rsis_l_gmax_fwd(const float* x, float* y, int* arg, long BC, int HW, hipStream_t st) {
  hipLaunchKernelGGL(global_maxpool_fwd_kernel, dim3((unsigned)BC), dim3(256), 0, st, x, y, arg, HW);
  return rsis_check_launch();
}
int rsis_l_gmax_bwd(const float* dy, const int* arg, float* dx, long BC, int HW, hipStream_t st) {
  const long total = BC * HW;
  hipLaunchKernelGGL(global_maxpool_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, st, dy, arg, dx, HW, total);
  return rsis_check_launch();
}
int rsis_l_bn_fwd(const float* x, const float* res, float* y, double* stats, const float* gamma, const float* beta,
                  float* run_mean, float* run_var, float* save_mean, float* save_rstd, int B, int C, int HW, float eps,
                  float momentum, int relu, int train_flags, hipStream_t st) {
  const long N = (long)B * HW;
  const int S = chan_splits(C, N);
  const int train = train_flags & 1;
  if (train && bn_fused_ok(C, HW, N)) {      // channel fits one block: one launch, one read of x
    // few channels = few blocks (one per channel): 512 threads x 4 float4 instead of 256 x 8 puts twice the waves on a CU
    // (measured +0.5 % on the training step; 1024 x 2 is no better)
    if (N / 4 <= 256 * 8 && C <= 512)
      hipLaunchKernelGGL((bn_fwd_fused_kernel<512, 4>), dim3(C), dim3(512), 0, st, x, res, y, gamma, beta, run_mean, run_var, save_mean,
                         save_rstd, C, HW, N, eps, momentum, relu);
    else if (N / 4 <= 256 * 8)
      hipLaunchKernelGGL((bn_fwd_fused_kernel<256, 8>), dim3(C), dim3(256), 0, st, x, res, y, gamma, beta, run_mean, run_var, save_mean,
                         save_rstd, C, HW, N, eps, momentum, relu);
    else
      hipLaunchKernelGGL((bn_fwd_fused_kernel<1024, 8>), dim3(C), dim3(1024), 0, st, x, res, y, gamma, beta, run_mean, run_var,
                         save_mean, save_rstd, C, HW, N, eps, momentum, relu);
    return rsis_check_launch();
  }
  if (train) {
    if (!(train_flags & 2) && rsis_zero_async(stats, sizeof(double) * 2 * C, st) != RSIS_OK) return RSIS_ERR_LAUNCH;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, S), dim3(256), 0, st, x, stats, C, HW, N);
  }
  hipLaunchKernelGGL(bn_apply_kernel, dim3(C, S), dim3(256), 0, st, x, res, y, stats, gamma, beta, run_mean, run_var, save_mean,
                     save_rstd, C, HW, N, eps, momentum, relu, train ? 0 : 1);
  return rsis_check_launch();
}
int rsis_l_bn_bwd(const float* dy, const float* x, const float* y, const float* mean, const float* rstd, const float* gamma,
                  double* stats, float* dx, float* dres, float* dgamma, float* dbeta, int B, int C, int HW, int relu_flags,
                  float eval_eps, hipStream_t st) {
  const long N = (long)B * HW;
  const int S = chan_splits(C, N);
  const int relu = relu_flags & 1, accum = (relu_flags >> 2) & 1;
  if (eval_eps < 0.f && bn_fused_ok(C, HW, N)) {
    if (N / 4 <= 256 * 8 && C <= 512)          // (as in the forward)
      hipLaunchKernelGGL((bn_bwd_fused_kernel<512, 4>), dim3(C), dim3(512), 0, st, dy, x, y, mean, rstd, gamma, dx, dres, dgamma, dbeta,
                         C, HW, N, relu, accum);
    else if (N / 4 <= 256 * 8)
      hipLaunchKernelGGL((bn_bwd_fused_kernel<256, 8>), dim3(C), dim3(256), 0, st, dy, x, y, mean, rstd, gamma, dx, dres, dgamma, dbeta,
                         C, HW, N, relu, accum);
    else
      hipLaunchKernelGGL((bn_bwd_fused_kernel<1024, 8>), dim3(C), dim3(1024), 0, st, dy, x, y, mean, rstd, gamma, dx, dres, dgamma,
                         dbeta, C, HW, N, relu, accum);
    return rsis_check_launch();
  }
  if (!(relu_flags & 2) && rsis_zero_async(stats, sizeof(double) * 2 * C, st) != RSIS_OK) return RSIS_ERR_LAUNCH;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, S), dim3(256), 0, st, dy, x, y, mean, rstd, stats, C, HW, N, relu, eval_eps);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(C, S), dim3(256), 0, st, dy, x, y, mean, rstd, gamma, stats, dx, dres, dgamma,
                     dbeta, C, HW, N, relu, accum, eval_eps);
  return rsis_check_launch();
}
int rsis_l_gmax_bwd_add(const float* dy, const int* arg, float* dx, long BC, int HW, hipStream_t st) {
  hipLaunchKernelGGL(global_maxpool_bwd_add_kernel, dim3((unsigned)((BC + 255) / 256)), dim3(256), 0, st, dy, arg, dx, HW, BC);
  return rsis_check_launch();
}
int rsis_l_subsample(const float* x, float* y, long BC, int H, int W, int Ho, int Wo, int s, hipStream_t st) {
  const long total4 = BC * Ho * ((Wo + 3) / 4);
  hipLaunchKernelGGL(subsample_kernel, dim3(ew_grid(total4)), dim3(256), 0, st, x, y, H, W, Ho, Wo, s, total4);
  return rsis_check_launch();
}
int rsis_l_maxpool_fwd(const float* x, float* y, unsigned char* arg, long BC, int H, int W, int Ho, int Wo, hipStream_t st) {
  const long total = BC * Ho * Wo;
  hipLaunchKernelGGL(maxpool3x3s2_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, st, x, y, arg, H, W, Ho, Wo, total);
  return rsis_check_launch();
}
int rsis_l_maxpool_bwd(const float* dy, const unsigned char* arg, float* dx, long BC, int H, int W, int Ho, int Wo, int accumulate,
                       hipStream_t st) {
  const long total = BC * H * W;
  if ((H & 1) == 0 && (W & 3) == 0) {
    hipLaunchKernelGGL(maxpool3x3s2_bwd_v8_kernel, dim3(ew_grid(total / 8)), dim3(256), 0, st, dy, arg, dx, H, W, Ho, Wo, total / 8,
                       accumulate);
    return rsis_check_launch();
  }
  hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, st, dy, arg, dx, H, W, Ho, Wo, total, accumulate);
  return rsis_check_launch();
}
// y[e] = sum_t x[t][e] (t ascending, one thread per float4 / float: a fixed order, bit-reproducible): the sum over the timesteps of
// the stacked gate gradients d(gates_t) that the time-invariant skip term of a ConvLSTM level receives (decoder_seq.py)
__global__ __launch_bounds__(256) void sum_leading_kernel(const float* __restrict__ x, float* __restrict__ y, int T, long n, long n4) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < n4) {
    const f32x4* p = reinterpret_cast<const f32x4*>(x) + e;
    f32x4 acc = p[0];
    for (int t = 1; t < T; ++t) { const f32x4 v = p[(size_t)t * (n / 4)]; acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3]; }
    reinterpret_cast<f32x4*>(y)[e] = acc;
  } else {
    const long i = 4 * n4 + (e - n4);
    if (i < n) { float acc = x[i]; for (int t = 1; t < T; ++t) acc += x[(size_t)t * n + i]; y[i] = acc; }
  }
}
int rsis_l_sum_leading(const float* x, float* y, int T, long n, hipStream_t st) {
  const bool v4 = (n % 4 == 0) && ((((size_t)x | (size_t)y) & 15) == 0);
  const long n4 = v4 ? n / 4 : 0, threads = n4 + (n - 4 * n4);
  hipLaunchKernelGGL(sum_leading_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, x, y, T, n, n4);
  return rsis_check_launch();
}

int rsis_l_channel_sum(const float* dy, float* db, int B, int C, int HW, int hid, hipStream_t st) {
  const long N = (long)B * HW;
  int S = chan_splits(C, N);
  if (S > 512) S = 512;                      // (all splits of a channel end in one atomic on the same address)
  hipLaunchKernelGGL(channel_sum_kernel, dim3(C, S), dim3(256), 0, st, dy, db, C, HW, N, hid);
  return rsis_check_launch();
}
int rsis_l_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd,
                int step, float gscale, const int* step_dev, hipStream_t st) {
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, step, gscale, step_dev);
  return rsis_check_launch();
}

int rsis_l_assign(const float* scores, long long* perm, int B, int G, int T, hipStream_t st) {
  hipLaunchKernelGGL(assign_kernel, dim3(B), dim3(64), 0, st, scores, perm, B, G, T);
  return rsis_check_launch();
}
