// The tail of the RSIS decoder as ONE pass per direction: the x2 align-corners upsample of the last ConvLSTM level's hidden state
// (reference src/modules/model.py:163-164, nn.UpsamplingBilinear2d) followed by conv_out (model.py:109,167: Conv2d(hidden/16 -> 1,
// 3x3, pad 1)), over the stacked images of all T timesteps.
//
// Run as two ops the upsampled tensor U -- 8 channels at the full output resolution, 4x the hidden state -- is written once and read
// twice forward and backward: at 256^2 / batch 32 / T = 10 in fp32 that is 671 MB per pass, five passes, ~1.1 ms of a 38 ms step
// (bf16 224^2: 257 MB per pass, ~0.9 ms of 13.5).  Both maps are linear and the channel contraction commutes with everything
// spatial, so with 9 tap maps at the LOW resolution
//     q_t = sum_c w[c][t] h_c                       (t = (r, s): 72 MACs per low-resolution pixel)
//     out(Y, X) = b + sum_t  up(q_t)(Y + r - 1, X + s - 1)          (zero outside the upsampled map: the conv's padding)
// the forward reads h once and writes the logits once (fp32: 168 + 84 MB instead of 168 + 671 + 671 + 84), and the backward
//     dq_t = up^T(shift_t^T(dout)),   dh_c = sum_t w[c][t] dq_t,   dW[c][t] = sum h_c dq_t,   db = sum dout
// reads dout and h once and writes dh once.  U is never formed; nothing is rounded between the two maps (the blk path used to round U
// to bf16).  The forward samples the tap maps from LDS; the backward's transpose interpolation is separable: columns (E), then rows.
//
// Layouts: h / dh are fp32 planes [T*B][8][Hs][Ws] (HBLK = 0) or bf16 blk cells [T*B][1][Hs][Ws][8] (HBLK = 1), images in [t][b]
// order; the logits and their gradient are fp32 [B][T][Ho*Wo] (image t * B + b of the former pairs with image b * T + t of the
// latter, as rsis_conv_out_seq_*).  w is the reference layout [1][8][3][3].
// HBM-bound: algorithmic bytes per launch = the three tensors once (fwd: h + out; bwd: dout + h + dh).
#include "common.h"

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

struct UpconvArgs {
  const void* h;
  const float* w;
  const float* bias;
  float* out;
  const float* dout;
  void* dh;
  float* partial;            // bwd: [gridDim.x][80] per-block sums of dW (72) and db (1)
  const float* dside;        // bwd: gradient of the side max-pool feature [T*B][8] (or null) ...
  const int* arg;            // ... and its arg-max pixel: added to dh at that pixel (model.py:143's max over the map)
  int T, Bn, Hs, Ws, Ho, Wo;
  float sh, sw;
  int tiles_x, tiles_y;
  int ntiles;
};

__device__ __forceinline__ float bf_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __uint_as_float(v & 0xFFFF0000u); }
__device__ __forceinline__ unsigned bf_pack2(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
}
template <int HBLK>
__device__ __forceinline__ void load_h8(const void* h, long img, int HsWs, int sp, float* v) {
  if constexpr (HBLK) {
    const u32x4_t c = ((const u32x4_t*)h)[img * HsWs + sp];
    v[0] = bf_lo(c[0]); v[1] = bf_hi(c[0]); v[2] = bf_lo(c[1]); v[3] = bf_hi(c[1]);
    v[4] = bf_lo(c[2]); v[5] = bf_hi(c[2]); v[6] = bf_lo(c[3]); v[7] = bf_hi(c[3]);
  } else {
    const float* p = (const float*)h + img * 8 * HsWs + sp;
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = p[(long)c * HsWs];
  }
}

// ------------------------------------------------------------------------------------------------
// forward: block = 32 x 64 logits of one image
// ------------------------------------------------------------------------------------------------
#define UF_OY 32
#define UF_OX 64
#define UF_MR 20          // low-resolution rows / columns under a tile and its 1-pixel halo (scale <= 0.52: the launcher checks)
#define UF_MC 36
template <int HBLK>
__global__ __launch_bounds__(256) void upconv_fwd_kernel(const UpconvArgs a) {
  __shared__ float q[9][UF_MR][UF_MC + 1];
  __shared__ int ry0[UF_OY + 2], ry1[UF_OY + 2], cx0[UF_OX + 2], cx1[UF_OX + 2];
  __shared__ float rfy[UF_OY + 2], rv[UF_OY + 2], cfx[UF_OX + 2], cv[UF_OX + 2];
  const int tid = threadIdx.x;
  const int per_img = a.tiles_x * a.tiles_y;
  const int m = blockIdx.x / per_img, trem = blockIdx.x - m * per_img;
  const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
  const int Y0 = ty * UF_OY, X0 = tx * UF_OX;
  const int Hs = a.Hs, Ws = a.Ws, Ho = a.Ho, Wo = a.Wo, HsWs = Hs * Ws;

  // ---- the low-resolution footprint of the tile and its one-pixel ring, by every thread for itself: the loads of h go out first ----
  int rlo, rhi, clo, chi, tmp; float tl;
  ac_coord(max(Y0 - 1, 0), a.sh, Hs, rlo, tmp, tl);
  ac_coord(min(Y0 + UF_OY, Ho - 1), a.sh, Hs, tmp, rhi, tl);
  ac_coord(max(X0 - 1, 0), a.sw, Ws, clo, tmp, tl);
  ac_coord(min(X0 + UF_OX, Wo - 1), a.sw, Ws, tmp, chi, tl);
  const int nr = rhi - rlo + 1, nc = chi - clo + 1, npx = nr * nc;       // (nr <= UF_MR, nc <= UF_MC: the launcher checks the scale)
  constexpr int NPT = (UF_MR * UF_MC + 255) / 256;
  float hv[NPT][8];
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int e = min(tid + i * 256, npx - 1);        // (clamped, not skipped: a load under a per-lane branch is waited for at once)
    const int lr = e / nc, lc = e - lr * nc;
    load_h8<HBLK>(a.h, m, HsWs, (rlo + lr) * Ws + clo + lc, hv[i]);
  }
  // ---- source rows / columns of the outputs and of the ring around them (the 3x3 taps) ----
  if (tid < UF_OY + 2) {
    const int Y = Y0 - 1 + tid;
    const bool ok = Y >= 0 && Y < Ho;
    int i0, i1; float l;
    ac_coord(ok ? Y : (Y < 0 ? 0 : Ho - 1), a.sh, Hs, i0, i1, l);
    ry0[tid] = i0 - rlo; ry1[tid] = i1 - rlo; rfy[tid] = l; rv[tid] = ok ? 1.f : 0.f;
  } else if (tid >= 128 && tid < 128 + UF_OX + 2) {
    const int j = tid - 128, X = X0 - 1 + j;
    const bool ok = X >= 0 && X < Wo;
    int i0, i1; float l;
    ac_coord(ok ? X : (X < 0 ? 0 : Wo - 1), a.sw, Ws, i0, i1, l);
    cx0[j] = i0 - clo; cx1[j] = i1 - clo; cfx[j] = l; cv[j] = ok ? 1.f : 0.f;
  }
  // ---- the 9 tap maps on the footprint ----
  float wv[72];
#pragma unroll
  for (int i = 0; i < 72; ++i) wv[i] = a.w[i];
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int e = tid + i * 256;
    if (e < npx) {
      const int lr = e / nc, lc = e - lr * nc;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) s += wv[c * 9 + t] * hv[i][c];
        q[t][lr][lc] = s;
      }
    }
  }
  __syncthreads();
  // ---- out(Y, X) = b + sum over the taps of the bilinear sample of q_t at (Y + r - 1, X + s - 1) (0 outside the upsampled map) ----
  {
    const int x = tid & 63, X = X0 + x;
    int l0s[3], l1s[3];
    float fxs[3], oks[3];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) { l0s[s3] = cx0[x + s3]; l1s[s3] = cx1[x + s3]; fxs[s3] = cfx[x + s3]; oks[s3] = cv[x + s3]; }
    const float b = a.bias ? a.bias[0] : 0.f;
    const int bimg = m % a.Bn, timg = m / a.Bn;
    float* const o = a.out + ((long)bimg * a.T + timg) * Ho * Wo;
#pragma unroll
    for (int k = 0; k < UF_OY / 4; ++k) {
      const int y = (tid >> 6) + 4 * k, Y = Y0 + y;
      float acc = b;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int i = y + r;
        const int l0 = ry0[i], l1 = ry1[i];
        const float f = rfy[i], ok = rv[i];
        float sr = 0.f;
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
          const float (*qt)[UF_MC + 1] = q[r * 3 + s3];
          const float a0 = (1.f - fxs[s3]) * qt[l0][l0s[s3]] + fxs[s3] * qt[l0][l1s[s3]];
          const float a1 = (1.f - fxs[s3]) * qt[l1][l0s[s3]] + fxs[s3] * qt[l1][l1s[s3]];
          sr += oks[s3] * ((1.f - f) * a0 + f * a1);
        }
        acc += ok * sr;
      }
      if (Y < Ho && X < Wo) o[Y * Wo + X] = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward: persistent blocks over 16 x 32 low-resolution tiles
// ------------------------------------------------------------------------------------------------
#define UB_RY 16
#define UB_RX 32
#define UB_UK 6           // candidate outputs per input index and axis (scale >= 0.46: the launcher checks)
#define UB_FY 46          // dout footprint of a tile: (conservative) candidate range + the taps' ring
#define UB_FX 80
template <int HBLK>
__global__ __launch_bounds__(256, 2) void upconv_bwd_kernel(const UpconvArgs a) {
  __shared__ float D[UB_FY][UB_FX + 1];
  __shared__ float E[3][UB_FY][UB_RX + 1];
  __shared__ float wy[UB_RY][UB_UK], wx[UB_RX][UB_UK];
  __shared__ int sty[UB_RY], stx[UB_RX];
  const int tid = threadIdx.x;
  const int Hs = a.Hs, Ws = a.Ws, Ho = a.Ho, Wo = a.Wo, HsWs = Hs * Ws;
  const int per_img = a.tiles_x * a.tiles_y;
  float wv[72];
#pragma unroll
  for (int i = 0; i < 72; ++i) wv[i] = a.w[i];
  float accw[72], accb = 0.f;
#pragma unroll
  for (int i = 0; i < 72; ++i) accw[i] = 0.f;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int m = tile / per_img, trem = tile - m * per_img;
    const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
    const int R0 = ty * UB_RY, C0 = tx * UB_RX;
    // ---- phase 0, nothing depends on anything: the hidden state of this thread's two pixels, the footprint of dout (its extent by a
    //      conservative closed form: an output o touches input i only if i - 1 < scale * o < i + 1), the per-axis tables ----
    float hv[2][8];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int px = tid + 256 * half;
      const int ys = min(R0 + (px >> 5), Hs - 1), xs = min(C0 + (px & 31), Ws - 1);      // (clamped, not skipped: no branch around a load)
      load_h8<HBLK>(a.h, m, HsWs, ys * Ws + xs, hv[half]);
    }
    const int oy = max(0, (int)floorf((R0 - 1) / a.sh) - 1) - 1, ox = max(0, (int)floorf((C0 - 1) / a.sw) - 1) - 1;
    const int ey = min((int)floorf(min(R0 + UB_RY, Hs) / a.sh) + 1 - oy + 1, UB_FY);
    const int ex = min((int)floorf(min(C0 + UB_RX, Ws) / a.sw) + 1 - ox + 1, UB_FX);
    {
      const int bimg = m % a.Bn, timg = m / a.Bn;
      const float* const d = a.dout + ((long)bimg * a.T + timg) * Ho * Wo;
      // element e = tid + 256 i of the MAXIMAL footprint (a constant divisor), 8 loads in flight at a time: every load of a chunk goes
      // out before the first use, at clamped addresses (no branch around a load)
      constexpr int NIT = (UB_FY * UB_FX + 255) / 256, CH = 8;
#pragma unroll
      for (int i0 = 0; i0 < NIT; i0 += CH) {
        float dv_[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int e = tid + 256 * (i0 + i);
          const int fr = e / UB_FX, fc = e - fr * UB_FX;
          const int Y = min(max(oy + fr, 0), Ho - 1), X = min(max(ox + fc, 0), Wo - 1);
          if (i0 + i < NIT) dv_[i] = d[Y * Wo + X];
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          if (i0 + i >= NIT) continue;
          const int e = tid + 256 * (i0 + i);
          const int fr = e / UB_FX, fc = e - fr * UB_FX;
          const int Y = oy + fr, X = ox + fc;
          const bool in = fr < ey && fc < ex && Y >= 0 && Y < Ho && X >= 0 && X < Wo;
          const float v = in ? dv_[i] : 0.f;
          if (fr < UB_FY) D[fr][fc] = v;
          // (bias gradient: an output belongs to the tile that holds its (i0, i0) source corner -- every output has exactly one)
          const int iy = min((int)(a.sh * Y), Hs - 1), ix = min((int)(a.sw * X), Ws - 1);
          accb += (in && iy >= R0 && iy < R0 + UB_RY && ix >= C0 && ix < C0 + UB_RX) ? v : 0.f;
        }
      }
    }
    if (tid < UB_RY || (tid >= 64 && tid < 64 + UB_RX)) {      // first candidate output of every input index and its UB_UK transpose weights
      const int axis = tid >= 64, li = axis ? tid - 64 : tid;
      const int in = axis ? Ws : Hs, out = axis ? Wo : Ho;
      const float sc = axis ? a.sw : a.sh;
      const int i = (axis ? C0 : R0) + li;
      const int l = max(0, (int)floorf((i - 1) / sc) - 1);
      int first = -1;
      float w[UB_UK];
#pragma unroll
      for (int k = 0; k < UB_UK; ++k) w[k] = 0.f;
      if (i < in) {
        for (int o = l; o < out && o < l + 4 + UB_UK; ++o) {
          int i0, i1; float l1;
          ac_coord(o, sc, in, i0, i1, l1);
          const float wo = (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
          if (first < 0 && (i0 == i || i1 == i)) first = o;
          if (first >= 0 && o - first < UB_UK) w[o - first] = wo;
        }
      }
      if (first < 0) first = l;                  // (no candidate: all weights 0; any in-footprint start will do)
#pragma unroll
      for (int k = 0; k < UB_UK; ++k) { if (axis) wx[li][k] = w[k]; else wy[li][k] = w[k]; }
      if (axis) stx[li] = first; else sty[li] = first;
    }
    __syncthreads();
    // ---- columns: E[s][row][xs] = sum_k wx[xs][k] dout[row][stx[xs] + k - (s - 1)] ----
    {
      const int xs = tid & 31;
      const int c0 = stx[xs] - ox;               // >= 1
      float wk[UB_UK];
#pragma unroll
      for (int k = 0; k < UB_UK; ++k) wk[k] = wx[xs][k];
      for (int fr = tid >> 5; fr < ey; fr += 8) {
        float v[UB_UK + 2];
#pragma unroll
        for (int k = 0; k < UB_UK + 2; ++k) { const int fc = c0 - 1 + k; v[k] = fc < ex ? D[fr][fc] : 0.f; }
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
          float e = 0.f;
#pragma unroll
          for (int k = 0; k < UB_UK; ++k) e += wk[k] * v[k + 2 - s3];      // column stx + k - (s3 - 1) = c0 - 1 + (k + 2 - s3)
          E[s3][fr][xs] = e;
        }
      }
    }
    __syncthreads();
    // ---- rows, then the channel side: dh, dW ----
    int argv[8];
    float sidev[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { argv[c] = -1; sidev[c] = 0.f; }
    if (a.arg) {                       // (uniform, once per tile: a per-pixel `if (arg) load` would serialise 8 round trips)
#pragma unroll
      for (int c = 0; c < 8; ++c) { argv[c] = a.arg[m * 8 + c]; sidev[c] = a.dside[m * 8 + c]; }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int px = tid + 256 * half;
      const int lys = px >> 5, lxs = px & 31;
      const int ys = R0 + lys, xs = C0 + lxs;
      if (ys < Hs && xs < Ws) {
        float dq[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) dq[t] = 0.f;
        const int r0 = sty[lys] - oy;            // >= 1
        float ev[3][UB_UK + 2];
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
          for (int k = 0; k < UB_UK + 2; ++k) { const int fr = r0 - 1 + k; ev[s3][k] = fr < ey ? E[s3][fr][lxs] : 0.f; }
#pragma unroll
        for (int k = 0; k < UB_UK; ++k) {
          const float wk = wy[lys][k];
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) dq[r * 3 + s3] += wk * ev[s3][k + 2 - r];      // row sty + k - (r - 1)
        }
        float dv[8];
        const int sp = ys * Ws + xs;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          float s = 0.f;
#pragma unroll
          for (int t = 0; t < 9; ++t) { s += wv[c * 9 + t] * dq[t]; accw[c * 9 + t] += hv[half][c] * dq[t]; }
          s += argv[c] == sp ? sidev[c] : 0.f;
          dv[c] = s;
        }
        if constexpr (HBLK) {
          const u32x4_t cell = {bf_pack2(dv[0], dv[1]), bf_pack2(dv[2], dv[3]), bf_pack2(dv[4], dv[5]), bf_pack2(dv[6], dv[7])};
          ((u32x4_t*)a.dh)[(long)m * HsWs + sp] = cell;
        } else {
          float* const p = (float*)a.dh + (long)m * 8 * HsWs + sp;
#pragma unroll
          for (int c = 0; c < 8; ++c) p[(long)c * HsWs] = dv[c];
        }
      }
    }
    __syncthreads();
  }
  // ---- the block's 73 sums: waves by shuffles, then through LDS, one row of `partial` per block (no atomics: the finalize kernel
  //      adds the rows in order, so dW / db are reproducible run to run) ----
  float* red = &E[0][0][0];
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < 73; ++i) {
    float v = i < 72 ? accw[i] : accb;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) red[wave * 80 + i] = v;
  }
  __syncthreads();
  if (tid < 73) a.partial[(long)blockIdx.x * 80 + tid] = red[tid] + red[80 + tid] + red[160 + tid] + red[240 + tid];
}

// dW / db += the blocks' rows, in a fixed order: 12 row groups x 80 columns sum their strided rows, then the groups in order
__global__ __launch_bounds__(960) void upconv_finalize_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float red[12][80];
  const int i = threadIdx.x % 80, g = threadIdx.x / 80;
  float s = 0.f;
  if (i < 73)
    for (int b = g; b < nblk; b += 12) s += partial[(long)b * 80 + i];
  red[g][i] = s;
  __syncthreads();
  if (g == 0 && i < 73) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 12; ++k) t += red[k][i];
    if (i < 72) { if (dW) dW[i] += t; }
    else if (db) db[0] += t;
  }
}

// ---- host side ----
bool rsis_upconv_supported(int C, int Hs, int Ws, int Ho, int Wo) {
  if (C != 8 || Hs < 2 || Ws < 2 || Ho < Hs || Wo < Ws) return false;
  const float sh = ac_scale(Hs, Ho), sw = ac_scale(Ws, Wo);
  if (sh < 0.46f || sw < 0.46f) return false;                                   // backward: UB_UK candidates per index
  if ((UF_OY + 1) * sh + 3.f > UF_MR || (UF_OX + 1) * sw + 3.f > UF_MC) return false;     // forward: the q footprint
  if ((UB_RY + 1) / sh + 7.f > UB_FY || (UB_RX + 1) / sw + 7.f > UB_FX) return false;           // backward: the dout footprint
  return (long)Ho * Wo < (1L << 30) && (long)Hs * Ws * 8 < (1L << 30);
}
int rsis_upconv_bwd_blocks(int T, int B, int Hs, int Ws) {
  const long ntiles = (long)T * B * rsis_cdiv(Hs, UB_RY) * rsis_cdiv(Ws, UB_RX);
  return (int)(ntiles < 512 ? ntiles : 512);          // two blocks per CU are resident (registers): one round of persistent blocks
}

int rsis_l_upconv_fwd(const void* h, int h_blk, const float* w, const float* bias, float* out, int T, int B, int Hs, int Ws, int Ho, int Wo,
                      hipStream_t st) {
  UpconvArgs a = {};
  a.h = h; a.w = w; a.bias = bias; a.out = out;
  a.T = T; a.Bn = B; a.Hs = Hs; a.Ws = Ws; a.Ho = Ho; a.Wo = Wo; a.sh = ac_scale(Hs, Ho); a.sw = ac_scale(Ws, Wo);
  a.tiles_x = rsis_cdiv(Wo, UF_OX); a.tiles_y = rsis_cdiv(Ho, UF_OY);
  const long grid = (long)T * B * a.tiles_x * a.tiles_y;
  if (grid <= 0 || grid > 0x7FFFFFFFL) return RSIS_ERR_ARG;
  if (h_blk) hipLaunchKernelGGL(upconv_fwd_kernel<1>, dim3((unsigned)grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(upconv_fwd_kernel<0>, dim3((unsigned)grid), dim3(256), 0, st, a);
  return rsis_check_launch();
}

int rsis_l_upconv_bwd(const float* dout, const void* h, int h_blk, const float* w, void* dh, float* dW, float* db, const float* dside,
                      const int* arg, float* partial, int T, int B, int Hs, int Ws, int Ho, int Wo, hipStream_t st) {
  UpconvArgs a = {};
  a.dout = dout; a.h = h; a.w = w; a.dh = dh; a.partial = partial; a.dside = dside; a.arg = dside ? arg : nullptr;
  a.T = T; a.Bn = B; a.Hs = Hs; a.Ws = Ws; a.Ho = Ho; a.Wo = Wo; a.sh = ac_scale(Hs, Ho); a.sw = ac_scale(Ws, Wo);
  a.tiles_x = rsis_cdiv(Ws, UB_RX); a.tiles_y = rsis_cdiv(Hs, UB_RY);
  const long ntiles = (long)T * B * a.tiles_x * a.tiles_y;
  if (ntiles <= 0 || ntiles > 0x7FFFFFFFL) return RSIS_ERR_ARG;
  a.ntiles = (int)ntiles;
  const int grid = rsis_upconv_bwd_blocks(T, B, Hs, Ws);
  if (h_blk) hipLaunchKernelGGL(upconv_bwd_kernel<1>, dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(upconv_bwd_kernel<0>, dim3(grid), dim3(256), 0, st, a);
  if (rsis_check_launch() != RSIS_OK) return RSIS_ERR_LAUNCH;
  if (dW || db) {
    hipLaunchKernelGGL(upconv_finalize_kernel, dim3(1), dim3(960), 0, st, (const float*)partial, grid, dW, db);
    return rsis_check_launch();
  }
  return RSIS_OK;
}
