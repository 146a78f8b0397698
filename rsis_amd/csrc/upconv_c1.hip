// conv_out fused with the decoder's final x2 bilinear upsample (reference src/modules/model.py:163-167:
// hidden = UpsamplingBilinear2d(2H, 2W)(hidden); out_mask = conv_out(hidden)), forward and weight gradient.
//
// Unfused, the up-sampled hidden state (8 channels at the full output resolution: 67 MB per timestep at 256^2 x 32) is written
// by the upsample, read by the conv, read again by the conv's weight gradient, and its gradient is written by the conv's data
// gradient and read by the upsample backward: 5 passes over a tensor that is a bilinear re-sampling of 17 MB.  Here the block
// that owns an 8 x 256 tile of the output stages the <= 8 x 136 hidden pixels its patch interpolates from (all channels, float4
// loads), builds the (8 + 2) x (256 + 2) up-sampled patch of a few channels in LDS -- zero outside the up-sampled image: the
// conv's padding -- and runs conv_c1.hip's LDS-tile convolution on it.  Same interpolation arithmetic as
// upsample_fwd_*_kernel (ac_coord, common.h) and same tap order as conv_c1_fwd_kernel: results equal the two-kernel sequence
// to a few ulp (the compiler contracts the interpolation into FMAs differently per kernel).  Bound: HBM at (C + 4) * B * H * W * 4 bytes (input at 1/4 resolution + output) instead of 2 * C + 1 ...
#include "common.h"

typedef const float __attribute__((address_space(1)))* gcf_t;
typedef float __attribute__((address_space(1)))* gf_t;

#define UC_TH 8                 // output tile rows
#define UC_TW 256               // output tile columns
#define UC_PH (UC_TH + 2)
#define UC_PW (UC_TW + 8)       // 4 floats each side: halo column, float4 aligned (as in conv_c1.hip)
#define UC_HR 8                 // staged hidden rows   (>= scale * (TH + 1) + 3)
#define UC_HC 136               // staged hidden columns (>= scale * (TW + 1) + 2 + 3 of float4 alignment), multiple of 4

#define UC_OUT (-2.f)
struct UpconvTables {           // per-block interpolation tables of the patch rows / columns
  int r0[UC_PH], r1[UC_PH];
  float rl[UC_PH];              // UC_OUT: row outside the up-sampled image (zero padding).  (A valid weight can be a tiny
                                // NEGATIVE number: src - i0 contracts to fma(scale, o, -i0) on the unrounded product.)
  int c0[UC_PW], c1[UC_PW];
  float cl[UC_PW];
};

// hidden region rows [r_lo, r_lo + UC_HR), columns [c_lo, c_lo + UC_HC) of NC channels -> LDS (float4, branch-free)
template <int NC>
__device__ __forceinline__ void uc_stage_hidden(float* __restrict__ hreg, gcf_t hb, int Hi, int Wi, int r_lo, int c_lo) {
  constexpr int NV = NC * UC_HR * (UC_HC / 4), ITER = (NV + 255) / 256;
  f32x4 v[ITER];
#pragma unroll
  for (int k = 0; k < ITER; ++k) {
    const int i = threadIdx.x + k * 256;
    const int q = i % (UC_HC / 4), t = i / (UC_HC / 4);
    const int r = t % UC_HR, ci = t / UC_HR;
    const int hy = r_lo + r, hx = c_lo + q * 4;
    const bool ok = i < NV && hy < Hi && hx < Wi;                  // Wi % 4 == 0, c_lo % 4 == 0: a float4 is all in or all out
    v[k] = *(const f32x4 __attribute__((address_space(1)))*)(ok ? hb + (size_t)ci * Hi * Wi + (size_t)hy * Wi + hx : hb);
  }
#pragma unroll
  for (int k = 0; k < ITER; ++k) {
    const int i = threadIdx.x + k * 256;
    if (i < NV) *reinterpret_cast<f32x4*>(hreg + i * 4) = v[k];
  }
}

// tables for the patch of the tile at (y0, x0): patch row pr <-> up-sampled row y0 - 1 + pr, column pc <-> x0 - 4 + pc
__device__ __forceinline__ void uc_make_tables(UpconvTables& T, int y0, int x0, int Hi, int Wi, int Ho, int Wo, float sh, float sw,
                                               int r_lo, int c_lo) {
  for (int i = threadIdx.x; i < UC_PH + UC_PW; i += 256) {
    if (i < UC_PH) {
      const int uy = y0 - 1 + i;
      int a = 0, b = 0; float l = UC_OUT;
      if (uy >= 0 && uy < Ho) { ac_coord(uy, sh, Hi, a, b, l); a -= r_lo; b -= r_lo; }
      T.r0[i] = a; T.r1[i] = b; T.rl[i] = l;
    } else {
      const int pc = i - UC_PH, ux = x0 - 4 + pc;
      int a = 0, b = 0; float l = UC_OUT;
      if (ux >= 0 && ux < Wo) { ac_coord(ux, sw, Wi, a, b, l); a -= c_lo; b -= c_lo; }
      T.c0[pc] = a; T.c1[pc] = b; T.cl[pc] = l;
    }
  }
}

// the up-sampled patch of NC channels from the staged hidden region (same expression as upsample_fwd_*_kernel)
template <int NC>
__device__ __forceinline__ void uc_make_patch(float* __restrict__ patch, const float* __restrict__ hreg, const UpconvTables& T) {
  constexpr int PE = UC_PH * UC_PW;
  for (int e = threadIdx.x; e < NC * PE; e += 256) {
    const int ci = e / PE, rem = e - ci * PE;
    const int pr = rem / UC_PW, pc = rem - pr * UC_PW;
    const float lh = T.rl[pr], lw = T.cl[pc];
    float v = 0.f;
    if (lh > -1.f && lw > -1.f) {
      const float* r0 = hreg + (ci * UC_HR + T.r0[pr]) * UC_HC;
      const float* r1 = hreg + (ci * UC_HR + T.r1[pr]) * UC_HC;
      const int w0 = T.c0[pc], w1 = T.c1[pc];
      v = (1.f - lh) * ((1.f - lw) * r0[w0] + lw * r0[w1]) + lh * ((1.f - lw) * r1[w0] + lw * r1[w1]);
    }
    patch[e] = v;
  }
}

// first hidden row / column (float4 aligned) the tile's patch reads
__device__ __forceinline__ void uc_region(int y0, int x0, int Hi, int Wi, float sh, float sw, int& r_lo, int& c_lo) {
  int a, b; float l;
  ac_coord(max(y0 - 1, 0), sh, Hi, a, b, l);
  r_lo = a;
  ac_coord(max(x0 - 1, 0), sw, Wi, a, b, l);
  c_lo = a & ~3;
}

template <int CIN>
__global__ __launch_bounds__(256) void upconv_c1_fwd_kernel(const float* __restrict__ h_, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y_, int B, int Hi,
                                                            int Wi, int Ho, int Wo, float sh, float sw) {
  constexpr int CG = CIN < 4 ? CIN : 4;                  // channels whose patch is in LDS at a time
  constexpr int QW = UC_TW / 4, RPP = 256 / QW, NOUT = UC_TH / RPP;
  __shared__ __attribute__((aligned(16))) float hreg[CIN * UC_HR * UC_HC];
  __shared__ __attribute__((aligned(16))) float patch[CG * UC_PH * UC_PW];
  __shared__ __attribute__((aligned(16))) float wl[CIN * 3 * 4];
  __shared__ UpconvTables T;
  const gcf_t hsrc = (gcf_t)h_;
  const gf_t y = (gf_t)y_;
  const int ntx = (Wo + UC_TW - 1) / UC_TW, nty = (Ho + UC_TH - 1) / UC_TH;
  int tile = blockIdx.x;
  const int txi = tile % ntx;
  tile /= ntx;
  const int tyi = tile % nty, b = tile / nty;
  const int y0 = tyi * UC_TH, x0 = txi * UC_TW;
  const int ty = threadIdx.x / QW, tx = threadIdx.x % QW;
  int r_lo, c_lo;
  uc_region(y0, x0, Hi, Wi, sh, sw, r_lo, c_lo);
  if (threadIdx.x < CIN * 9) {                           // reference-layout weight [1][CIN][3][3]
    const int ci = threadIdx.x / 9, rs = threadIdx.x % 9;
    wl[(ci * 3 + rs / 3) * 4 + rs % 3] = w[threadIdx.x];
  }
  uc_make_tables(T, y0, x0, Hi, Wi, Ho, Wo, sh, sw, r_lo, c_lo);
  uc_stage_hidden<CIN>(hreg, hsrc + (size_t)b * CIN * Hi * Wi, Hi, Wi, r_lo, c_lo);
  const float b0 = bias ? bias[0] : 0.f;
  f32x4 acc[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) acc[o] = f32x4{b0, b0, b0, b0};
#pragma unroll 1
  for (int cb = 0; cb < CIN; cb += CG) {
    __syncthreads();                                     // hidden region + tables ready / previous patch consumed
    uc_make_patch<CG>(patch, hreg + cb * UC_HR * UC_HC, T);
    __syncthreads();
#pragma unroll
    for (int ci = 0; ci < CG; ++ci) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + ((cb + ci) * 3 + r) * 4);
        const float w0 = wv[0], w1 = wv[1], w2 = wv[2];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
          const float* p = patch + (ci * UC_PH + ty + o * RPP + r) * UC_PW + 4 + tx * 4;
          const f32x4 m = *reinterpret_cast<const f32x4*>(p);
          const float l = p[-1], rr = p[4];
          acc[o][0] = fmaf(w0, l, fmaf(w1, m[0], fmaf(w2, m[1], acc[o][0])));
          acc[o][1] = fmaf(w0, m[0], fmaf(w1, m[1], fmaf(w2, m[2], acc[o][1])));
          acc[o][2] = fmaf(w0, m[1], fmaf(w1, m[2], fmaf(w2, m[3], acc[o][2])));
          acc[o][3] = fmaf(w0, m[2], fmaf(w1, m[3], fmaf(w2, rr, acc[o][3])));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    const int oy = y0 + ty + o * RPP, ox = x0 + tx * 4;
    if (oy < Ho && ox < Wo) *(f32x4 __attribute__((address_space(1)))*)(y + (size_t)b * Ho * Wo + (size_t)oy * Wo + ox) = acc[o];
  }
}

// Weight (and bias) gradient: dW[ci][r][s] += sum dy[y][x] * up(h)[ci][y + r - 1][x + s - 1], db += sum dy.  Persistent blocks own
// one channel pair (conv_c1_wgrad_kernel's scheme: 18 + 1 same-address atomics per block), the patch is rebuilt from the hidden
// state instead of being read back.
template <int CIN>
__global__ __launch_bounds__(256) void upconv_c1_wgrad_kernel(const float* __restrict__ dy_, const float* __restrict__ h_,
                                                              float* __restrict__ dw, float* __restrict__ db, int B, int Hi, int Wi,
                                                              int Ho, int Wo, float sh, float sw) {
  constexpr int CGW = 2, S = CIN / CGW;
  constexpr int QW = UC_TW / 4, RPP = 256 / QW, NOUT = UC_TH / RPP;
  __shared__ __attribute__((aligned(16))) float hreg[CGW * UC_HR * UC_HC];
  __shared__ __attribute__((aligned(16))) float patch[CGW * UC_PH * UC_PW];
  __shared__ UpconvTables T;
  const gcf_t dy = (gcf_t)dy_, hsrc = (gcf_t)h_;
  const int ntx = (Wo + UC_TW - 1) / UC_TW, nty = (Ho + UC_TH - 1) / UC_TH;
  const int ntiles = B * nty * ntx;
  const int ty = threadIdx.x / QW, tx = threadIdx.x % QW;
  const int cb = (blockIdx.x % S) * CGW;
  float acc[CGW * 9];
#pragma unroll
  for (int i = 0; i < CGW * 9; ++i) acc[i] = 0.f;
  float gsum = 0.f;
#pragma unroll 1
  for (int tile = blockIdx.x / S; tile < ntiles; tile += gridDim.x / S) {
    const int txi = tile % ntx, t2 = tile / ntx;
    const int tyi = t2 % nty, b = t2 / nty;
    const int y0 = tyi * UC_TH, x0 = txi * UC_TW;
    int r_lo, c_lo;
    uc_region(y0, x0, Hi, Wi, sh, sw, r_lo, c_lo);
    f32x4 g[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      const int oy = y0 + ty + o * RPP, ox = x0 + tx * 4;
      g[o] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (oy < Ho && ox < Wo) g[o] = *(const f32x4 __attribute__((address_space(1)))*)(dy + (size_t)b * Ho * Wo + (size_t)oy * Wo + ox);
      gsum += (g[o][0] + g[o][1]) + (g[o][2] + g[o][3]);
    }
    __syncthreads();                                     // previous tile's patch / tables consumed
    uc_make_tables(T, y0, x0, Hi, Wi, Ho, Wo, sh, sw, r_lo, c_lo);
    uc_stage_hidden<CGW>(hreg, hsrc + ((size_t)b * CIN + cb) * Hi * Wi, Hi, Wi, r_lo, c_lo);
    __syncthreads();
    uc_make_patch<CGW>(patch, hreg, T);
    __syncthreads();
#pragma unroll
    for (int ci = 0; ci < CGW; ++ci) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float* a = acc + ci * 9 + r * 3;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
          const float* p = patch + (ci * UC_PH + ty + o * RPP + r) * UC_PW + 4 + tx * 4;
          const f32x4 m = *reinterpret_cast<const f32x4*>(p);
          const float l = p[-1], rr = p[4];
          a[0] = fmaf(g[o][0], l, fmaf(g[o][1], m[0], fmaf(g[o][2], m[1], fmaf(g[o][3], m[2], a[0]))));
          a[1] = fmaf(g[o][0], m[0], fmaf(g[o][1], m[1], fmaf(g[o][2], m[2], fmaf(g[o][3], m[3], a[1]))));
          a[2] = fmaf(g[o][0], m[1], fmaf(g[o][1], m[2], fmaf(g[o][2], m[3], fmaf(g[o][3], rr, a[2]))));
        }
      }
    }
  }
  __shared__ float red[4][CGW * 9 + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i <= CGW * 9; ++i) {
    float v = i < CGW * 9 ? acc[i] : gsum;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[wv][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < CGW * 9) atomicAdd(dw + cb * 9 + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
  if (threadIdx.x == CGW * 9 && db && cb == 0) atomicAdd(db, red[0][CGW * 9] + red[1][CGW * 9] + red[2][CGW * 9] + red[3][CGW * 9]);
}

// Data gradient: dh = upsample_backward(conv_out_dgrad(dy)) without materialising the up-sampled gradient.  Structure and
// arithmetic of upsample_bwd_kernel<32, 80> (pointwise.hip: a block owns a 32 x 32 tile of hidden pixels, per-block tables of
// the <= 6 consecutive up-sampled rows / columns that touch each hidden row / column, separable gather through LDS) -- but the
// region of the up-sampled gradient it gathers from is not loaded: the block stages the dy region (+ 1-pixel halo, zero outside
// the image) ONCE and, for each of the CIN channels in turn, computes that channel's 3x3 data gradient of the region into LDS
// (same tap order as conv_c1_dgrad_kernel).  dy is read once per tile instead of d(up) being written and read per channel.
#define UB_UT 32
#define UB_RM 80
#define UB_UK 6
#define UB_RP (UB_RM + 4)
#define UB_DQ ((UB_RM + 8) / 4)      // float4 per staged dy row: RM + 2 halo columns + <= 3 of alignment
#define UB_DP (UB_DQ * 4)
template <int CIN>
__global__ __launch_bounds__(256) void upconv_c1_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                 float* __restrict__ dh, int Hi, int Wi, int Ho, int Wo, float sh,
                                                                 float sw, int tiles_x, int tiles_y) {
  __shared__ float wgt[2][UB_UT][UB_UK];
  __shared__ int st[2][UB_UT];
  __shared__ int org[2], ext[2];
  __shared__ __attribute__((aligned(16))) float dyr[(UB_RM + 2) * UB_DP];
  __shared__ __attribute__((aligned(16))) float reg[UB_RM * UB_RP];
  __shared__ float tmp[UB_UT * UB_RP];
  __shared__ float wl[CIN * 9];
  const int tid = threadIdx.x;
  const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y;
  const long b = blockIdx.x / (tiles_x * tiles_y);
  if (tid < CIN * 9) wl[tid] = w[tid];
  if (tid < 2 * UB_UT) {                       // (identical to upsample_bwd_kernel)
    const int axis = tid / UB_UT, li = tid % UB_UT;
    const int in = axis ? Wi : Hi, out = axis ? Wo : Ho;
    const float sc = axis ? sw : sh;
    const int i = (axis ? tx : ty) * UB_UT + li;
    int l = max(0, (int)floorf((i - 1) / sc) - 1);
    int first = -1;
    float wv6[UB_UK];
#pragma unroll
    for (int k = 0; k < UB_UK; ++k) wv6[k] = 0.f;
    if (i < in) {
      for (int o = l; o < out && o < l + 4 + UB_UK; ++o) {
        int i0, i1; float l1;
        ac_coord(o, sc, in, i0, i1, l1);
        const float wv = (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
        if (first < 0 && (i0 == i || i1 == i)) first = o;
        if (first >= 0 && o - first < UB_UK) wv6[o - first] = wv;
      }
    }
    if (first < 0) first = min(l, out - 1);
#pragma unroll
    for (int k = 0; k < UB_UK; ++k) wgt[axis][li][k] = wv6[k];
    st[axis][li] = first;
  }
  __syncthreads();
  if (tid < 2) {
    const int o0 = st[tid][0];
    int last = 0;
    for (int li = 0; li < UB_UT; ++li) last = max(last, st[tid][li] + UB_UK - 1);
    const int out = tid ? Wo : Ho;
    org[tid] = o0;
    ext[tid] = min(min(last, out - 1) - o0 + 1, UB_RM);
  }
  __syncthreads();
  const int oy = org[0], ox = org[1], ey = ext[0], ex = ext[1];
  // ---- dy region rows oy - 1 .. oy + ey, columns from wa = floor4(ox - 1): aligned float4s, zero outside the image ----
  const int wa = (ox - 1) & ~3;                // (two's complement: -1 -> -4)
  const int cso = ox - wa;                     // dyr column of up-sampled column ox
  {
    constexpr int NV = (UB_RM + 2) * UB_DQ, ITER = (NV + 255) / 256;
    const float* yb = dy + b * Ho * Wo;
    f32x4 v[ITER];
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
      const int i = tid + k * 256;
      const int r = i / UB_DQ, q = i - r * UB_DQ;
      const int row = oy - 1 + r, col = wa + q * 4;
      const bool ok = i < NV && r < ey + 2 && row >= 0 && row < Ho && col >= 0 && col < Wo;   // Wo % 4 == 0
      v[k] = *reinterpret_cast<const f32x4*>(ok ? yb + (size_t)row * Wo + col : yb);
      if (!ok) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
      const int i = tid + k * 256;
      if (i < NV) *reinterpret_cast<f32x4*>(dyr + i * 4) = v[k];
    }
  }
  __syncthreads();
  for (int ci = 0; ci < CIN; ++ci) {
    // d(up)[ci] on the region: reg[r][c] = sum_{r', s'} w[ci][r'][s'] * dy[oy + r + 1 - r'][ox + c + 1 - s'], 4 columns per thread
    float wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = wl[ci * 9 + t];
    for (int e = tid; e < ey * (UB_RM / 4); e += 256) {
      const int r = e / (UB_RM / 4), c = (e - r * (UB_RM / 4)) * 4;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int rp = 0; rp < 3; ++rp) {
        const float* d = dyr + (r + 2 - rp) * UB_DP + c + cso;          // dy[.][ox + c] of this tap row
        const float dm1 = d[-1], d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4];
        const float w0 = wt[rp * 3], w1 = wt[rp * 3 + 1], w2 = wt[rp * 3 + 2];   // s' = 0, 1, 2 read column + 1, 0, -1
        a0 = fmaf(w0, d1, a0); a0 = fmaf(w1, d0, a0); a0 = fmaf(w2, dm1, a0);
        a1 = fmaf(w0, d2, a1); a1 = fmaf(w1, d1, a1); a1 = fmaf(w2, d0, a1);
        a2 = fmaf(w0, d3, a2); a2 = fmaf(w1, d2, a2); a2 = fmaf(w2, d1, a2);
        a3 = fmaf(w0, d4, a3); a3 = fmaf(w1, d3, a3); a3 = fmaf(w2, d2, a3);
      }
      float* o = reg + r * UB_RP + c;
      o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
    }
    __syncthreads();
    for (int e = tid; e < UB_UT * UB_RM; e += 256) {
      const int ly = e / UB_RM, c = e - ly * UB_RM;
      const int r0 = st[0][ly] - oy;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < UB_UK; ++k) acc = fmaf(wgt[0][ly][k], reg[min(r0 + k, ey - 1) * UB_RP + c], acc);
      tmp[ly * UB_RP + c] = acc;
    }
    __syncthreads();
    for (int e = tid; e < UB_UT * UB_UT; e += 256) {
      const int ly = e / UB_UT, lx = e - ly * UB_UT;
      const int hi = ty * UB_UT + ly, wi = tx * UB_UT + lx;
      if (hi >= Hi || wi >= Wi) continue;
      const int c0 = st[1][lx] - ox;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < UB_UK; ++k) acc = fmaf(wgt[1][lx][k], tmp[ly * UB_RP + min(c0 + k, ex - 1)], acc);
      dh[((size_t)b * CIN + ci) * Hi * Wi + (size_t)hi * Wi + wi] = acc;
    }
    // (reg is rewritten only after the barrier that follows the row phase of this channel has been passed by every thread --
    //  the column phase above reads tmp only; tmp is rewritten after the next channel's first barrier)
  }
}

// the fused kernels cover x2-like up-samplings whose tile reads fit the staged hidden region
bool rsis_upconv_c1_supported(int Cin, int Hi, int Wi, int Ho, int Wo) {
  if (!(Cin == 4 || Cin == 8 || Cin == 16) || Wi % 4 || Wo % 4 || Hi < 2 || Wi < 2) return false;
  const float sh = ac_scale(Hi, Ho), sw = ac_scale(Wi, Wo);
  if (!(sh * (UC_TH + 1) + 3.f <= UC_HR && sw * (UC_TW + 1) + 6.f <= UC_HC)) return false;
  // data gradient (upsample_bwd_kernel<32, 80>'s conditions): <= UB_UK candidates per hidden index, region <= UB_RM per axis
  const float smin = sh < sw ? sh : sw;
  return smin > 0.f && 2.f / smin + 1.f <= UB_UK && (UB_UT - 1) / smin + UB_UK + 3 <= UB_RM;
}

#define UC_DISPATCH(KERNEL, GRID, ...)                                                                    \
  switch (Cin) {                                                                                          \
    case 4: hipLaunchKernelGGL((KERNEL<4>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__); break;            \
    case 8: hipLaunchKernelGGL((KERNEL<8>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__); break;            \
    case 16: hipLaunchKernelGGL((KERNEL<16>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__); break;          \
    default: return RSIS_ERR_UNSUPPORTED;                                                                 \
  }

int rsis_l_upconv_c1_fwd(const float* h, const float* w, const float* bias, float* y, int B, int Cin, int Hi, int Wi, int Ho, int Wo,
                         hipStream_t st) {
  const long tiles = (long)B * ((Ho + UC_TH - 1) / UC_TH) * ((Wo + UC_TW - 1) / UC_TW);
  if (tiles > 0x7fffffffL) return RSIS_ERR_ARG;
  UC_DISPATCH(upconv_c1_fwd_kernel, (int)tiles, h, w, bias, y, B, Hi, Wi, Ho, Wo, ac_scale(Hi, Ho), ac_scale(Wi, Wo))
  return rsis_check_launch();
}
int rsis_l_upconv_c1_wgrad(const float* dy, const float* h, float* dw, float* db, int B, int Cin, int Hi, int Wi, int Ho, int Wo,
                           hipStream_t st) {
  const long tiles = (long)B * ((Ho + UC_TH - 1) / UC_TH) * ((Wo + UC_TW - 1) / UC_TW);
  const long want = tiles * (Cin / 2);
  const int grid = (int)(want > 512 ? 512 : want);
  UC_DISPATCH(upconv_c1_wgrad_kernel, grid, dy, h, dw, db, B, Hi, Wi, Ho, Wo, ac_scale(Hi, Ho), ac_scale(Wi, Wo))
  return rsis_check_launch();
}
int rsis_l_upconv_c1_bwd_data(const float* dy, const float* w, float* dh, int B, int Cin, int Hi, int Wi, int Ho, int Wo,
                              hipStream_t st) {
  const int tiles_x = (Wi + UB_UT - 1) / UB_UT, tiles_y = (Hi + UB_UT - 1) / UB_UT;
  const long blocks = (long)B * tiles_x * tiles_y;
  if (blocks > 0x7fffffffL) return RSIS_ERR_ARG;
  UC_DISPATCH(upconv_c1_bwd_data_kernel, (int)blocks, dy, w, dh, Hi, Wi, Ho, Wo, ac_scale(Hi, Ho), ac_scale(Wi, Wo), tiles_x, tiles_y)
  return rsis_check_launch();
}
