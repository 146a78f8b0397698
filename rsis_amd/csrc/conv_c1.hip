// conv_out of the RSIS decoder: nn.Conv2d(hidden/16 -> 1, 3x3, pad 1) at the full output resolution (reference
// src/modules/model.py:109,167).  With ONE output channel there is no GEMM to speak of (AI ~4 FLOP/B): forward, data
// gradient and weight gradient are HBM-bandwidth kernels on the vector ALUs -- coalesced along W, filter taps in scalar
// registers -- instead of 1/32-utilised MFMA tiles.  Bound: HBM (~6.3 TB/s achievable); algorithmic bytes per launch:
// fwd/dgrad (Cin+1)*B*H*W*4, wgrad the same reads + 9*Cin floats out.
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

// One thread = 4 consecutive output pixels of a row (W % 4 == 0): per channel and filter row it loads the aligned float4 and
// its two neighbours (6 inputs feed 12 taps), 4x fewer load instructions than one pixel per thread.
template <int CIN>
__global__ __launch_bounds__(256) void conv_c1_fwd_kernel(const float* __restrict__ x_, const float* __restrict__ wp, int ldw,
                                                          const float* __restrict__ bias, float* __restrict__ y_, int B, int H,
                                                          int W) {
  const gcf_t x = (gcf_t)x_;
  const gf_t y = (gf_t)y_;
  const int HW = H * W, Wq = W >> 2;
  const int items = B * H * Wq;
  float w[CIN * 9];
#pragma unroll
  for (int i = 0; i < CIN * 9; ++i) w[i] = w_fwd(wp, ldw, i / 9, i % 9);   // uniform -> scalar loads
  const float b0 = bias ? bias[0] : 0.f;
  for (int it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) {
    const int xq = it % Wq, t = it / Wq;
    const int yy = t % H, b = t / H;
    const int x0 = xq * 4;
    const gcf_t xb = x + (size_t)b * CIN * HW;
    f32x4 acc = {b0, b0, b0, b0};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int iy = yy + r - 1;
      if ((unsigned)iy >= (unsigned)H) continue;
      const gcf_t row = xb + iy * W + x0;
      const bool hl = x0 > 0, hr = x0 + 4 < W;
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        const gcf_t pc = row + (size_t)ci * HW;
        const f32x4 m = *(const f32x4 __attribute__((address_space(1)))*)pc;
        const float l = hl ? pc[-1] : 0.f, rr = hr ? pc[4] : 0.f;
        const float w0 = w[ci * 9 + r * 3], w1 = w[ci * 9 + r * 3 + 1], w2 = w[ci * 9 + r * 3 + 2];
        acc[0] = fmaf(w0, l, fmaf(w1, m[0], fmaf(w2, m[1], acc[0])));
        acc[1] = fmaf(w0, m[0], fmaf(w1, m[1], fmaf(w2, m[2], acc[1])));
        acc[2] = fmaf(w0, m[1], fmaf(w1, m[2], fmaf(w2, m[3], acc[2])));
        acc[3] = fmaf(w0, m[2], fmaf(w1, m[3], fmaf(w2, rr, acc[3])));
      }
    }
    *(f32x4 __attribute__((address_space(1)))*)(y + (size_t)it * 4) = acc;
  }
}

template <int CIN>
__global__ __launch_bounds__(256) void conv_c1_dgrad_kernel(const float* __restrict__ dy_, const float* __restrict__ wd, int ldw,
                                                            float* __restrict__ dx_, int B, int H, int W) {
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
    const gcf_t gb = dy + (size_t)b * HW;
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

// One thread = 4 consecutive pixels of a row (W % 4 == 0), same 6-inputs-per-row reuse as the forward.
template <int CIN>
__global__ __launch_bounds__(256) void conv_c1_wgrad_kernel(const float* __restrict__ dy_, const float* __restrict__ x_,
                                                            float* __restrict__ dw, int B, int H, int W) {
  const gcf_t dy = (gcf_t)dy_, x = (gcf_t)x_;
  const int HW = H * W, Wq = W >> 2;
  const int items = B * H * Wq;
  float acc[CIN * 9];
#pragma unroll
  for (int i = 0; i < CIN * 9; ++i) acc[i] = 0.f;
  for (int it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) {
    const int xq = it % Wq, t = it / Wq;
    const int yy = t % H, b = t / H;
    const int x0 = xq * 4;
    const f32x4 g = *(const f32x4 __attribute__((address_space(1)))*)(dy + (size_t)it * 4);
    const gcf_t xb = x + (size_t)b * CIN * HW;
    const bool hl = x0 > 0, hr = x0 + 4 < W;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int iy = yy + r - 1;
      if ((unsigned)iy >= (unsigned)H) continue;
      const gcf_t row = xb + iy * W + x0;
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        const gcf_t pc = row + (size_t)ci * HW;
        const f32x4 m = *(const f32x4 __attribute__((address_space(1)))*)pc;
        const float l = hl ? pc[-1] : 0.f, rr = hr ? pc[4] : 0.f;
        float* a = acc + ci * 9 + r * 3;
        a[0] = fmaf(g[0], l, fmaf(g[1], m[0], fmaf(g[2], m[1], fmaf(g[3], m[2], a[0]))));
        a[1] = fmaf(g[0], m[0], fmaf(g[1], m[1], fmaf(g[2], m[2], fmaf(g[3], m[3], a[1]))));
        a[2] = fmaf(g[0], m[1], fmaf(g[1], m[2], fmaf(g[2], m[3], fmaf(g[3], rr, a[2]))));
      }
    }
  }
  // block reduction: wave shuffle, then LDS across the 4 waves, then one atomic per filter tap
  __shared__ float red[4][CIN * 9];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < CIN * 9; ++i) {
    float v = acc[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[wv][i] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CIN * 9; i += 256) atomicAdd(dw + i, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
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

int rsis_l_c1_fwd(const float* x, const float* wp, int ldw, const float* bias, float* y, int B, int Cin, int H, int W,
                  hipStream_t st) {
  const int grid = c1_grid((long)B * H * W / 4, 16);
  C1_DISPATCH(conv_c1_fwd_kernel, grid, x, wp, ldw, bias, y, B, H, W)
  return rsis_check_launch();
}
int rsis_l_c1_dgrad(const float* dy, const float* wd, int ldw, float* dx, int B, int Cin, int H, int W, hipStream_t st) {
  const int grid = c1_grid((long)B * H * W, 16);
  C1_DISPATCH(conv_c1_dgrad_kernel, grid, dy, wd, ldw, dx, B, H, W)
  return rsis_check_launch();
}
int rsis_l_c1_wgrad(const float* dy, const float* x, float* dw, int B, int Cin, int H, int W, hipStream_t st) {
  const int grid = c1_grid((long)B * H * W / 4, 4);
  C1_DISPATCH(conv_c1_wgrad_kernel, grid, dy, x, dw, B, H, W)
  return rsis_check_launch();
}
