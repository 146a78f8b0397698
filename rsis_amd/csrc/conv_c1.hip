// conv_out of the RSIS decoder: nn.Conv2d(hidden/16 -> 1, 3x3, pad 1) at the full output resolution (reference
// src/modules/model.py:109,167).  With ONE output channel there is no GEMM to speak of (AI ~4 FLOP/B): forward, data
// gradient and weight gradient are memory-bound kernels on the vector ALUs instead of 1/32-utilised MFMA tiles.
// Algorithmic bytes per launch: fwd/dgrad (Cin+1)*B*H*W*4, wgrad the same reads + 9*Cin floats out.  Measured at
// 8 x 256^2 x 32 (75.5 MB): fwd 22.5 us, dgrad 18.7 us, wgrad 28.8 us (a streaming read of the input alone: 10.4 us).
#include "common.h"

typedef const float __attribute__((address_space(1)))* gcf_t;
typedef float __attribute__((address_space(1)))* gf_t;

// weight accessors into the library's packed layouts (pack.hip, direct-3x3 layout, Cout == 1 -> column 0 / one chunk)
__device__ __forceinline__ float w_fwd(const float* __restrict__ wp, int ldw, int ci, int rs) {
  return wp[(size_t)((ci >> 3) * (RSIS_CK * 9) + (((ci & 7) >> 1) * 9 + rs) * 2 + (ci & 1)) * ldw];
}
__device__ __forceinline__ float w_dgrad(const float* __restrict__ wd, int ldw, int ci, int rs) {   // = W[0][ci][rs]
  return wd[(size_t)(2 * (8 - rs)) * ldw + ci];
}

// ---- LDS-tiled forward / weight gradient.  Measured (tools/exp/c1_exp.hip, 8 channels, 256^2, B = 32; a plain streaming read
// of the input takes 10 us): with operands straight from global memory every input row is requested 3x (once per filter row)
// plus two single-float neighbour loads per float4, the working set of a block exceeds the L1, and the kernel sits at 32 us;
// a first version of this file also paid ~0.5 us for EACH of 72 dependent scalar loads of the filter taps (40 us).  Here a
// block stages the input patch of a TH x TW pixel tile (+ halo) in LDS once, all float4, the taps go through LDS too, and a
// thread owns 4 consecutive pixels of a tile row: one ds_read_b128 + two ds_read_b32 per (channel, filter row).
// The tile is as wide as the image when W <= 256: a halo COLUMN costs a whole extra cache line per row and side, so narrow
// tiles double the fetched bytes (measured: 16 x 64 tiles stage at half the streaming rate); halo columns outside the image
// are never fetched.
template <int TW, int TH> struct C1Tile {
  static constexpr int QW = TW / 4;              // float4 per tile row
  static constexpr int RPP = 256 / QW;           // tile rows covered by the 256 threads at once
  static constexpr int NOUT = TH / RPP;          // float4 outputs per thread
  static constexpr int PH = TH + 2, PW = TW + 8; // patch: 1 halo row each side; 4 floats each side (halo column, float4 aligned)
  static constexpr int CGMAX = TW == 64 ? 8 : 4; // channels staged per pass (~42 KB of LDS)
};

// time-batched launches (rsis_conv_out_seq_*): the Cin-channel tensor holds the images of seqT timesteps in [t][b] order, the
// one-channel tensor (logits / their gradient) is [b][t] -- image t * (B / seqT) + b of the former pairs with image b * seqT + t
__device__ __forceinline__ int c1_ymap(int img, int B, int seqT) {
  if (seqT <= 1) return img;
  const int Bn = B / seqT;
  return (img % Bn) * seqT + img / Bn;
}

template <int CG, int TW, int TH>
__device__ __forceinline__ void c1_stage_patch(float* __restrict__ patch, gcf_t xb, int H, int W, int y0, int x0) {
  using T = C1Tile<TW, TH>;
  constexpr int NV = CG * T::PH * (T::PW / 4), ITER = (NV + 255) / 256;
  const int HW = H * W;
  f32x4 v[ITER];
#pragma unroll
  for (int k = 0; k < ITER; ++k) {                 // branch-free so that the loads are issued back to back
    const int i = threadIdx.x + k * 256;
    const int q = i % (T::PW / 4), t = i / (T::PW / 4);
    const int pr = t % T::PH, ci = t / T::PH;
    const int iy = y0 - 1 + pr, ix = x0 - 4 + q * 4;
    const bool ok = i < NV && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;   // W % 4 == 0: a float4 is all in or all out
    const gcf_t pc = ok ? xb + (size_t)ci * HW + iy * W + ix : xb;
    v[k] = *(const f32x4 __attribute__((address_space(1)))*)pc;
    if (!ok) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int k = 0; k < ITER; ++k) {
    const int i = threadIdx.x + k * 256;
    if (i < NV) *reinterpret_cast<f32x4*>(patch + i * 4) = v[k];
  }
}

template <int CIN, int TW, int TH>
__global__ __launch_bounds__(256) void conv_c1_fwd_kernel(const float* __restrict__ x_, const float* __restrict__ wp, int ldw,
                                                          const float* __restrict__ bias, float* __restrict__ y_, int B, int H,
                                                          int W, int seqT) {
  using T = C1Tile<TW, TH>;
  constexpr int CG = CIN < T::CGMAX ? CIN : T::CGMAX;
  __shared__ __attribute__((aligned(16))) float patch[CG * T::PH * T::PW];
  __shared__ __attribute__((aligned(16))) float wl[CIN * 3 * 4];
  const gcf_t x = (gcf_t)x_;
  const gf_t y = (gf_t)y_;
  const int HW = H * W, ntx = (W + TW - 1) / TW, nty = (H + TH - 1) / TH;
  int tile = blockIdx.x;
  const int txi = tile % ntx;
  tile /= ntx;
  const int tyi = tile % nty, b = tile / nty;
  const int y0 = tyi * TH, x0 = txi * TW;
  const int ty = threadIdx.x / T::QW, tx = threadIdx.x % T::QW;
  if (threadIdx.x < CIN * 9) {                     // filter taps: one parallel vector load, read back as broadcast float4s
    const int ci = threadIdx.x / 9, rs = threadIdx.x % 9;
    wl[(ci * 3 + rs / 3) * 4 + rs % 3] = w_fwd(wp, ldw, ci, rs);
  }
  const float b0 = bias ? bias[0] : 0.f;
  f32x4 acc[T::NOUT];
#pragma unroll
  for (int o = 0; o < T::NOUT; ++o) acc[o] = f32x4{b0, b0, b0, b0};
#pragma unroll 1
  for (int c0 = 0; c0 < CIN; c0 += CG) {
    if (c0) __syncthreads();
    c1_stage_patch<CG, TW, TH>(patch, x + ((size_t)b * CIN + c0) * HW, H, W, y0, x0);
    __syncthreads();
#pragma unroll
    for (int ci = 0; ci < CG; ++ci) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + ((c0 + ci) * 3 + r) * 4);
        const float w0 = wv[0], w1 = wv[1], w2 = wv[2];
#pragma unroll
        for (int o = 0; o < T::NOUT; ++o) {
          const float* p = patch + (ci * T::PH + ty + o * T::RPP + r) * T::PW + 4 + tx * 4;
          const f32x4 m = *reinterpret_cast<const f32x4*>(p);
          const float l = p[-1], rr = p[4];
          acc[o][0] = fmaf(w0, l, fmaf(w1, m[0], fmaf(w2, m[1], acc[o][0])));
          acc[o][1] = fmaf(w0, m[0], fmaf(w1, m[1], fmaf(w2, m[2], acc[o][1])));
          acc[o][2] = fmaf(w0, m[1], fmaf(w1, m[2], fmaf(w2, m[3], acc[o][2])));
          acc[o][3] = fmaf(w0, m[2], fmaf(w1, m[3], fmaf(w2, rr, acc[o][3])));
        }
      }
      __builtin_amdgcn_sched_barrier(0);     // keep the LDS reads of later channels from being hoisted (register pressure)
    }
  }
#pragma unroll
  for (int o = 0; o < T::NOUT; ++o) {
    const int oy = y0 + ty + o * T::RPP, ox = x0 + tx * 4;
    if (oy < H && ox < W) *(f32x4 __attribute__((address_space(1)))*)(y + (size_t)c1_ymap(b, B, seqT) * HW + (size_t)oy * W + ox) = acc[o];
  }
}

template <int CIN>
__global__ __launch_bounds__(256) void conv_c1_dgrad_kernel(const float* __restrict__ dy_, const float* __restrict__ wd, int ldw,
                                                            float* __restrict__ dx_, int B, int H, int W, int seqT) {
  const gcf_t dy = (gcf_t)dy_;
  const gf_t dx = (gf_t)dx_;
  const int HW = H * W;
  const long total = (long)B * HW;
  float w[CIN * 9];
#pragma unroll
  for (int i = 0; i < CIN * 9; ++i) w[i] = w_dgrad(wd, ldw, i / 9, i % 9);
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int b = (int)(e / HW), sp = (int)(e - (long)b * HW);
    const int yy = sp / W, xx = sp - yy * W;
    const gcf_t gb = dy + (size_t)c1_ymap(b, B, seqT) * HW;
    float g[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int oy = yy + 1 - r, ox = xx + 1 - s;   // output pixel whose tap (r, s) reads this input pixel
        g[r * 3 + s] = ((unsigned)oy < (unsigned)H && (unsigned)ox < (unsigned)W) ? gb[oy * W + ox] : 0.f;
      }
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) acc = fmaf(w[ci * 9 + t], g[t], acc);
      dx[((size_t)b * CIN + ci) * HW + sp] = acc;
    }
  }
}

// Weight gradient on the same LDS tiles.  The 9 * CIN sums end in same-address atomics, which the L2 serialises (measured ~0.5 ns
// each: 2048 blocks x 72 atomics = 80 us): persistent blocks (grid <= 512), and each block owns only CGW = 2 input channels (18
// accumulators per thread, 18 atomics per block; the dy tile is re-read once per channel pair).
template <int CIN, int TW, int TH>
__global__ __launch_bounds__(256) void conv_c1_wgrad_kernel(const float* __restrict__ dy_, const float* __restrict__ x_,
                                                            float* __restrict__ dw, float* __restrict__ db, int B, int H, int W,
                                                            int seqT) {
  using T = C1Tile<TW, TH>;
  constexpr int CGW = 2, S = CIN / CGW;
  __shared__ __attribute__((aligned(16))) float patch[CGW * T::PH * T::PW];
  const gcf_t dy = (gcf_t)dy_, x = (gcf_t)x_;
  const int HW = H * W, ntx = (W + TW - 1) / TW, nty = (H + TH - 1) / TH;
  const int ntiles = B * nty * ntx;
  const int ty = threadIdx.x / T::QW, tx = threadIdx.x % T::QW;
  const int cb = (blockIdx.x % S) * CGW;                   // this block's channel pair
  float acc[CGW * 9];
#pragma unroll
  for (int i = 0; i < CGW * 9; ++i) acc[i] = 0.f;
  float gsum = 0.f;                                          // bias gradient (sum of dy), taken by the blocks of channel pair 0
#pragma unroll 1
  for (int tile = blockIdx.x / S; tile < ntiles; tile += gridDim.x / S) {
    const int txi = tile % ntx, t2 = tile / ntx;
    const int tyi = t2 % nty, b = t2 / nty;
    const int y0 = tyi * TH, x0 = txi * TW;
    f32x4 g[T::NOUT];
#pragma unroll
    for (int o = 0; o < T::NOUT; ++o) {
      const int oy = y0 + ty + o * T::RPP, ox = x0 + tx * 4;
      g[o] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (oy < H && ox < W) g[o] = *(const f32x4 __attribute__((address_space(1)))*)(dy + (size_t)c1_ymap(b, B, seqT) * HW + (size_t)oy * W + ox);
      gsum += (g[o][0] + g[o][1]) + (g[o][2] + g[o][3]);
    }
    __syncthreads();
    c1_stage_patch<CGW, TW, TH>(patch, x + ((size_t)b * CIN + cb) * HW, H, W, y0, x0);
    __syncthreads();
#pragma unroll
    for (int ci = 0; ci < CGW; ++ci) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float* a = acc + ci * 9 + r * 3;
#pragma unroll
        for (int o = 0; o < T::NOUT; ++o) {
          const float* p = patch + (ci * T::PH + ty + o * T::RPP + r) * T::PW + 4 + tx * 4;
          const f32x4 m = *reinterpret_cast<const f32x4*>(p);
          const float l = p[-1], rr = p[4];
          a[0] = fmaf(g[o][0], l, fmaf(g[o][1], m[0], fmaf(g[o][2], m[1], fmaf(g[o][3], m[2], a[0]))));
          a[1] = fmaf(g[o][0], m[0], fmaf(g[o][1], m[1], fmaf(g[o][2], m[2], fmaf(g[o][3], m[3], a[1]))));
          a[2] = fmaf(g[o][0], m[1], fmaf(g[o][1], m[2], fmaf(g[o][2], m[3], fmaf(g[o][3], rr, a[2]))));
        }
      }
    }
  }
  // block reduction: wave shuffle, then LDS across the 4 waves, then one atomic per filter tap
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

static inline int c1_grid(long total, int per_cu) {
  long g = (total + 255) / 256;
  const long cap = 256L * per_cu;
  if (g > cap) g = cap;
  return (int)(g < 1 ? 1 : g);
}

bool rsis_c1_supported(int Cin) { return Cin == 4 || Cin == 8 || Cin == 16; }   // (+ W % 4 == 0, checked by the caller)

#define C1_DISPATCH(KERNEL, GRID, ...)                                                                       \
  switch (Cin) {                                                                                             \
    case 4: hipLaunchKernelGGL((KERNEL<4>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__); break;               \
    case 8: hipLaunchKernelGGL((KERNEL<8>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__); break;               \
    case 16: hipLaunchKernelGGL((KERNEL<16>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__); break;             \
    default: return RSIS_ERR_UNSUPPORTED;                                                                    \
  }
// tile shape by image width (full-width tiles up to W = 256); the kernels handle partial tiles
#define C1_TILED(KERNEL, TW, TH, GRIDCAP, PER_TILE, ...)                                                               \
  {                                                                                                          \
    const long tiles = (long)B * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);                                  \
    if (tiles > 0x7fffffffL) return RSIS_ERR_ARG;                                                            \
    if (tiles * (PER_TILE) > 0x7fffffffL) return RSIS_ERR_ARG;                                               \
    const int grid = (int)(GRIDCAP > 0 && tiles * (PER_TILE) > GRIDCAP ? GRIDCAP : tiles * (PER_TILE));      \
    switch (Cin) {                                                                                           \
      case 4: hipLaunchKernelGGL((KERNEL<4, TW, TH>), dim3(grid), dim3(256), 0, st, __VA_ARGS__); break;     \
      case 8: hipLaunchKernelGGL((KERNEL<8, TW, TH>), dim3(grid), dim3(256), 0, st, __VA_ARGS__); break;     \
      case 16: hipLaunchKernelGGL((KERNEL<16, TW, TH>), dim3(grid), dim3(256), 0, st, __VA_ARGS__); break;   \
      default: return RSIS_ERR_UNSUPPORTED;                                                                  \
    }                                                                                                        \
  }
#define C1_BY_WIDTH(KERNEL, GRIDCAP, PER_TILE, ...)                                                          \
  if (W > 128) C1_TILED(KERNEL, 256, 8, GRIDCAP, PER_TILE, __VA_ARGS__)                                      \
  else if (W > 64) C1_TILED(KERNEL, 128, 16, GRIDCAP, PER_TILE, __VA_ARGS__)                                 \
  else C1_TILED(KERNEL, 64, 16, GRIDCAP, PER_TILE, __VA_ARGS__)

// seqT > 1: B = seqT * batch images, the Cin-channel tensor in [t][b] order, the one-channel tensor in [b][t] order (c1_ymap)
int rsis_l_c1_fwd(const float* x, const float* wp, int ldw, const float* bias, float* y, int B, int Cin, int H, int W, int seqT,
                  hipStream_t st) {
  C1_BY_WIDTH(conv_c1_fwd_kernel, 0, 1, x, wp, ldw, bias, y, B, H, W, seqT)
  return rsis_check_launch();
}
int rsis_l_c1_dgrad(const float* dy, const float* wd, int ldw, float* dx, int B, int Cin, int H, int W, int seqT, hipStream_t st) {
  const int grid = c1_grid((long)B * H * W, 16);
  C1_DISPATCH(conv_c1_dgrad_kernel, grid, dy, wd, ldw, dx, B, H, W, seqT)
  return rsis_check_launch();
}
int rsis_l_c1_wgrad(const float* dy, const float* x, float* dw, float* db, int B, int Cin, int H, int W, int seqT, hipStream_t st) {
  // one block per (tile, channel pair) up to the cap (a multiple of every Cin / 2), persistent beyond it
  // (deterministic mode: ONE persistent block per channel pair, so every dW / db address has a single contributor)
  const int cap = rsis_deterministic() ? Cin / 2 : 512;
  C1_BY_WIDTH(conv_c1_wgrad_kernel, cap, Cin / 2, dy, x, dw, db, B, H, W, seqT)     // measured: 256 -> 39 us, 512 -> 29 us, 1024 -> 30 us, 2048 -> 43 us
  return rsis_check_launch();
}
