// Scratch experiment (not part of the library): what bounds the conv_out forward (8 -> 1 channels, 3x3, 256x256, B = 32)?
//   hipcc --offload-arch=gfx950 -O3 tools/exp/c1_exp.hip -o /tmp/c1_exp && /tmp/c1_exp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const float __attribute__((address_space(1)))* gcf_t;
typedef float __attribute__((address_space(1)))* gf_t;
#define CIN 8

// MODE 0: as the library kernel; 1: no neighbour loads; 2: centre row only
template <int MODE>
__global__ __launch_bounds__(256) void k_base(const float* __restrict__ x_, const float* __restrict__ wp, float* __restrict__ y_, int B, int H, int W) {
  const gcf_t x = (gcf_t)x_;
  const gf_t y = (gf_t)y_;
  const int HW = H * W, Wq = W >> 2;
  const int items = B * H * Wq;
  float w[CIN * 9];
#pragma unroll
  for (int i = 0; i < CIN * 9; ++i) w[i] = wp[i];
  for (int it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) {
    const int xq = it % Wq, t = it / Wq;
    const int yy = t % H, b = t / H;
    const int x0 = xq * 4;
    const gcf_t xb = x + (size_t)b * CIN * HW;
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      if (MODE == 2 && r != 1) continue;
      const int iy = yy + r - 1;
      if ((unsigned)iy >= (unsigned)H) continue;
      const gcf_t row = xb + iy * W + x0;
      const bool hl = x0 > 0, hr = x0 + 4 < W;
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        const gcf_t pc = row + (size_t)ci * HW;
        const f32x4 m = *(const f32x4 __attribute__((address_space(1)))*)pc;
        float l = 0.f, rr = 0.f;
        if (MODE == 0 || MODE == 2) { l = hl ? pc[-1] : 0.f; rr = hr ? pc[4] : 0.f; }
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

// R output rows x 4 px per thread, rolling over R + 2 input rows; neighbours via wave shuffles (a wave = 64 consecutive float4 of a row)
template <int R>
__global__ __launch_bounds__(256) void k_rows(const float* __restrict__ x_, const float* __restrict__ wp, float* __restrict__ y_, int B, int H, int W) {
  const gcf_t x = (gcf_t)x_;
  const gf_t y = (gf_t)y_;
  const int HW = H * W, Wq = W >> 2, Hr = H / R;
  const int items = B * Hr * Wq;
  float w[CIN * 9];
#pragma unroll
  for (int i = 0; i < CIN * 9; ++i) w[i] = wp[i];
  for (int it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) {
    const int xq = it % Wq, t = it / Wq;
    const int yr = t % Hr, b = t / Hr;
    const int x0 = xq * 4, y0 = yr * R;
    const gcf_t xb = x + (size_t)b * CIN * HW + x0;
    f32x4 acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const bool hl = x0 > 0, hr = x0 + 4 < W;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll
      for (int j = 0; j < R + 2; ++j) {
        const int iy = y0 + j - 1;
        f32x4 m = {0, 0, 0, 0};
        float l = 0.f, rr = 0.f;
        if ((unsigned)iy < (unsigned)H) {
          const gcf_t pc = xb + (size_t)ci * HW + iy * W;
          m = *(const f32x4 __attribute__((address_space(1)))*)pc;
          l = hl ? pc[-1] : 0.f;
          rr = hr ? pc[4] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int o = j - r;      // output row (local) fed by input row j through tap row r
          if (o < 0 || o >= R) continue;
          const float w0 = w[ci * 9 + r * 3], w1 = w[ci * 9 + r * 3 + 1], w2 = w[ci * 9 + r * 3 + 2];
          acc[o][0] = fmaf(w0, l, fmaf(w1, m[0], fmaf(w2, m[1], acc[o][0])));
          acc[o][1] = fmaf(w0, m[0], fmaf(w1, m[1], fmaf(w2, m[2], acc[o][1])));
          acc[o][2] = fmaf(w0, m[1], fmaf(w1, m[2], fmaf(w2, m[3], acc[o][2])));
          acc[o][3] = fmaf(w0, m[2], fmaf(w1, m[3], fmaf(w2, rr, acc[o][3])));
        }
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) *(f32x4 __attribute__((address_space(1)))*)(y + ((size_t)b * HW + (size_t)(y0 + i) * W + x0)) = acc[i];
  }
}

// as k_rows, neighbours from the adjacent lanes (a wave = 64 consecutive float4 of the flattened [row][W/4] index space); only the
// two end lanes of a wave load their outer neighbour from memory (one 2-lane load per row and channel)
template <int R, int DPP, int EDGE = 0>
__global__ __launch_bounds__(256) void k_rows_x(const float* __restrict__ x_, const float* __restrict__ wp, float* __restrict__ y_, int B, int H, int W) {
  const gcf_t x = (gcf_t)x_;
  const gf_t y = (gf_t)y_;
  const int HW = H * W, Wq = W >> 2, Hr = H / R;
  const int items = B * Hr * Wq;
  const int lane = threadIdx.x & 63;
  float w[CIN * 9];
#pragma unroll
  for (int i = 0; i < CIN * 9; ++i) w[i] = wp[i];
  for (int it0 = blockIdx.x * 256; it0 < items; it0 += gridDim.x * 256) {
    const int it = it0 + threadIdx.x;          // items % 256 == 0 assumed here
    const int xq = it % Wq, t = it / Wq;
    const int yr = t % Hr, b = t / Hr;
    const int x0 = xq * 4, y0 = yr * R;
    const gcf_t xb = x + (size_t)b * CIN * HW + x0;
    f32x4 acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const bool hl = x0 > 0, hr = x0 + 4 < W;
    const bool edge_l = lane == 0 && hl, edge_r = lane == 63 && hr;   // neighbour lives in another wave
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll
      for (int j = 0; j < R + 2; ++j) {
        const int iy = y0 + j - 1;
        f32x4 m = {0, 0, 0, 0};
        float e = 0.f;
        if (EDGE == 1 || EDGE == 2) {      // branch-free: clamped row, zeroed afterwards
          const bool ok = (unsigned)iy < (unsigned)H;
          const gcf_t pc = xb + (size_t)ci * HW + (ok ? iy : y0) * W;
          m = *(const f32x4 __attribute__((address_space(1)))*)pc;
          if (EDGE == 2) e = pc[edge_l ? -1 : (edge_r ? 4 : 0)];
          if (!ok) { m = f32x4{0, 0, 0, 0}; e = 0.f; }
        } else if ((unsigned)iy < (unsigned)H) {
          const gcf_t pc = xb + (size_t)ci * HW + iy * W;
          m = *(const f32x4 __attribute__((address_space(1)))*)pc;
          if (edge_l || edge_r) e = pc[edge_l ? -1 : 4];
        }
        float l, rr;
        if (DPP) {
          l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m[3]), 0x138, 0xf, 0xf, false));   // wave_shr:1
          rr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m[0]), 0x130, 0xf, 0xf, false));  // wave_shl:1
        } else {
          l = __shfl_up(m[3], 1, 64);
          rr = __shfl_down(m[0], 1, 64);
        }
        l = lane == 0 ? e : (hl ? l : 0.f);
        rr = lane == 63 ? e : (hr ? rr : 0.f);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int o = j - r;
          if (o < 0 || o >= R) continue;
          const float w0 = w[ci * 9 + r * 3], w1 = w[ci * 9 + r * 3 + 1], w2 = w[ci * 9 + r * 3 + 2];
          acc[o][0] = fmaf(w0, l, fmaf(w1, m[0], fmaf(w2, m[1], acc[o][0])));
          acc[o][1] = fmaf(w0, m[0], fmaf(w1, m[1], fmaf(w2, m[2], acc[o][1])));
          acc[o][2] = fmaf(w0, m[1], fmaf(w1, m[2], fmaf(w2, m[3], acc[o][2])));
          acc[o][3] = fmaf(w0, m[2], fmaf(w1, m[3], fmaf(w2, rr, acc[o][3])));
        }
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) *(f32x4 __attribute__((address_space(1)))*)(y + ((size_t)b * HW + (size_t)(y0 + i) * W + x0)) = acc[i];
  }
}

// ---- LDS-tiled variant (16 x 64 tile + halo staged once; neighbours by DPP row shifts (NB = 1), shuffles (NB = 0) or LDS (NB = 2))
#define TH 16
#define TW 64
#define PH (TH + 2)
#define PW (TW + 8)
template <int NB, int STAGE_ONLY>
__global__ __launch_bounds__(256) void k_lds(const float* __restrict__ x_, const float* __restrict__ wp, float* __restrict__ y_, int B, int H, int W) {
  __shared__ __attribute__((aligned(16))) float patch[CIN * PH * PW];
  __shared__ __attribute__((aligned(16))) float wl[CIN * 3 * 4];
  const gcf_t x = (gcf_t)x_;
  const gf_t y = (gf_t)y_;
  const int HW = H * W, ntx = W / TW, nty = H / TH;
  int tile = blockIdx.x;
  const int txi = tile % ntx; tile /= ntx;
  const int tyi = tile % nty, b = tile / nty;
  const int y0 = tyi * TH, x0 = txi * TW;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  if (threadIdx.x < CIN * 9) wl[(threadIdx.x / 9 * 3 + threadIdx.x % 9 / 3) * 4 + threadIdx.x % 3] = wp[threadIdx.x];
  constexpr int NV = CIN * PH * (PW / 4), ITER = (NV + 255) / 256;
  const gcf_t xb = x + (size_t)b * CIN * HW;
  f32x4 v[ITER];
#pragma unroll
  for (int k = 0; k < ITER; ++k) {
    const int i = threadIdx.x + k * 256;
    const int q = i % (PW / 4), t = i / (PW / 4);
    const int pr = t % PH, ci = t / PH;
    const int iy = y0 - 1 + pr, ix = x0 - 4 + q * 4;
    const bool ok = i < NV && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const gcf_t pc = ok ? xb + (size_t)ci * HW + iy * W + ix : xb;
    v[k] = *(const f32x4 __attribute__((address_space(1)))*)pc;
    if (!ok) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int k = 0; k < ITER; ++k) {
    const int i = threadIdx.x + k * 256;
    if (i < NV) *reinterpret_cast<f32x4*>(patch + i * 4) = v[k];
  }
  __syncthreads();
  f32x4 acc = {0, 0, 0, 0};
  if (STAGE_ONLY) {
    acc = *reinterpret_cast<const f32x4*>(patch + (ty * PW) + 4 + tx * 4);
  } else {
    const int eoff = tx == 0 ? 3 : (tx == 15 ? TW + 4 : 4 + tx * 4);
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float* row = patch + (ci * PH + ty + r) * PW;
        const f32x4 m = *reinterpret_cast<const f32x4*>(row + 4 + tx * 4);
        float l, rr;
        if (NB == 2) { l = row[3 + tx * 4]; rr = row[8 + tx * 4]; }
        else {
          const float e = row[eoff];
          if (NB == 1) {
            l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m[3]), 0x111, 0xf, 0xf, false));
            rr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m[0]), 0x101, 0xf, 0xf, false));
          } else { l = __shfl_up(m[3], 1, 64); rr = __shfl_down(m[0], 1, 64); }
          l = tx == 0 ? e : l;
          rr = tx == 15 ? e : rr;
        }
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + (ci * 3 + r) * 4);
        const float w0 = wv[0], w1 = wv[1], w2 = wv[2];
        acc[0] = fmaf(w0, l, fmaf(w1, m[0], fmaf(w2, m[1], acc[0])));
        acc[1] = fmaf(w0, m[0], fmaf(w1, m[1], fmaf(w2, m[2], acc[1])));
        acc[2] = fmaf(w0, m[1], fmaf(w1, m[2], fmaf(w2, m[3], acc[2])));
        acc[3] = fmaf(w0, m[2], fmaf(w1, m[3], fmaf(w2, rr, acc[3])));
      }
    }
  }
  *(f32x4 __attribute__((address_space(1)))*)(y + (size_t)b * HW + (size_t)(y0 + ty) * W + x0 + tx * 4) = acc;
}

// pure streaming read of x (float4 per lane, sum) -- the bandwidth ceiling for this buffer
__global__ __launch_bounds__(256) void k_read(const float* __restrict__ x_, float* __restrict__ y, long n4) {
  const f32x4 __attribute__((address_space(1)))* x = (const f32x4 __attribute__((address_space(1)))*)x_;
  f32x4 s = {0, 0, 0, 0};
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += gridDim.x * 256L) s += x[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.f) y[0] = 1.f;
}

template <typename F>
static float timeit(F f, int iters = 20) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f();
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

int main() {
  const int B = 32, H = 256, W = 256;
  const size_t nx = (size_t)B * CIN * H * W, ny = (size_t)B * H * W;
  float *x, *y, *w, *big;
  hipMalloc(&x, nx * 4); hipMalloc(&y, ny * 4); hipMalloc(&w, 72 * 4);
  hipMalloc(&big, 1024ul << 20);          // 1 GiB scrub buffer: evicts L2 / MALL between launches
  {
    std::vector<float> hx(nx), hw(72);
    unsigned s = 1;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / (1 << 24) - 0.5f; }
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / (1 << 24) - 0.5f; }
    hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), 72 * 4, hipMemcpyHostToDevice);
  }
  hipMemset(big, 0, 1024ul << 20);
  std::vector<float> ref(ny), got(ny);
  auto check = [&](const char* name) {
    hipMemcpy(got.data(), y, ny * 4, hipMemcpyDeviceToHost);
    double md = 0;
    for (size_t i = 0; i < ny; ++i) md = fmax(md, fabs((double)got[i] - ref[i]));
    printf("  check %-12s max |diff| vs base = %.3g\n", name, md);
    hipMemset(y, 0, ny * 4);
  };
  hipLaunchKernelGGL(k_base<0>, dim3(B * H * W / 4 / 256), dim3(256), 0, 0, x, w, y, B, H, W);
  hipMemcpy(ref.data(), y, ny * 4, hipMemcpyDeviceToHost);
  hipMemset(y, 0, ny * 4);
  hipLaunchKernelGGL((k_rows_x<4, 0>), dim3(B * H * W / 16 / 256), dim3(256), 0, 0, x, w, y, B, H, W); check("rows_x<4,shfl>");
  hipLaunchKernelGGL((k_rows_x<4, 1>), dim3(B * H * W / 16 / 256), dim3(256), 0, 0, x, w, y, B, H, W); check("rows_x<4,dpp>");
  hipLaunchKernelGGL((k_lds<0, 0>), dim3(B * (H / TH) * (W / TW)), dim3(256), 0, 0, x, w, y, B, H, W); check("lds shfl");
  hipLaunchKernelGGL((k_lds<1, 0>), dim3(B * (H / TH) * (W / TW)), dim3(256), 0, 0, x, w, y, B, H, W); check("lds dpp");
  hipLaunchKernelGGL((k_lds<2, 0>), dim3(B * (H / TH) * (W / TW)), dim3(256), 0, 0, x, w, y, B, H, W); check("lds lds");
  hipLaunchKernelGGL((k_rows_x<2, 0, 1>), dim3(B * H * W / 8 / 256), dim3(256), 0, 0, x, w, y, B, H, W); check("aligned<2>");
  hipLaunchKernelGGL((k_rows_x<2, 0, 2>), dim3(B * H * W / 8 / 256), dim3(256), 0, 0, x, w, y, B, H, W); check("dummy<2>");
  const int items = B * H * W / 4;
  for (int cold = 0; cold < 2; ++cold) {
    auto wrap = [&](auto launch) {
      if (!cold) return timeit(launch);
      float tot = 0;                       // cold: scrub, then time ONE launch with events
      for (int i = 0; i < 5; ++i) {
        hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, big, y, (long)(1024ul << 20) / 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms * 1e3f;
      }
      return tot / 5;
    };
    printf("---- %s\n", cold ? "cold (cache scrubbed before each launch)" : "hot (back-to-back launches)");
    for (int g : {1024, 2048, 4096})
      printf("stream read of x, grid %4d   %7.1f us\n", g, wrap([&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, x, y, (long)nx / 4); }));
    printf("base (library kernel)          %7.1f us\n", wrap([&] { hipLaunchKernelGGL(k_base<0>, dim3(items / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("base, no neighbour loads       %7.1f us\n", wrap([&] { hipLaunchKernelGGL(k_base<1>, dim3(items / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("base, centre row only          %7.1f us\n", wrap([&] { hipLaunchKernelGGL(k_base<2>, dim3(items / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("rows R=2                       %7.1f us\n", wrap([&] { hipLaunchKernelGGL(k_rows<2>, dim3(items / 2 / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("rows R=4                       %7.1f us\n", wrap([&] { hipLaunchKernelGGL(k_rows<4>, dim3(items / 4 / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("rows R=8                       %7.1f us\n", wrap([&] { hipLaunchKernelGGL(k_rows<8>, dim3(items / 8 / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("rows_x R=1 shfl                %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_rows_x<1, 0>), dim3(items / 1 / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("rows_x R=1 dpp                 %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_rows_x<1, 1>), dim3(items / 1 / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("rows_x R=2 shfl                %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_rows_x<2, 0>), dim3(items / 2 / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("rows_x R=2 dpp                 %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_rows_x<2, 1>), dim3(items / 2 / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("rows_x R=4 shfl                %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_rows_x<4, 0>), dim3(items / 4 / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("rows_x R=4 dpp                 %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_rows_x<4, 1>), dim3(items / 4 / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("aligned (no edge loads) R=1    %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_rows_x<1, 0, 1>), dim3(items / 1 / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("aligned (no edge loads) R=2    %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_rows_x<2, 0, 1>), dim3(items / 2 / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("aligned (no edge loads) R=4    %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_rows_x<4, 0, 1>), dim3(items / 4 / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("aligned (no edge loads) R=8    %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_rows_x<8, 0, 1>), dim3(items / 8 / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("dummy-address edge load R=1    %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_rows_x<1, 0, 2>), dim3(items / 1 / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("dummy-address edge load R=2    %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_rows_x<2, 0, 2>), dim3(items / 2 / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("dummy-address edge load R=4    %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_rows_x<4, 0, 2>), dim3(items / 4 / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("lds tile, shfl neighbours      %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_lds<0, 0>), dim3(B * (H / TH) * (W / TW)), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("lds tile, dpp neighbours       %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_lds<1, 0>), dim3(B * (H / TH) * (W / TW)), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("lds tile, lds neighbours       %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_lds<2, 0>), dim3(B * (H / TH) * (W / TW)), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("lds tile, staging only         %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_lds<2, 1>), dim3(B * (H / TH) * (W / TW)), dim3(256), 0, 0, x, w, y, B, H, W); }));
    printf("rows_x R=8 dpp                 %7.1f us\n", wrap([&] { hipLaunchKernelGGL((k_rows_x<8, 1>), dim3(items / 8 / 256), dim3(256), 0, 0, x, w, y, B, H, W); }));
  }
  return 0;
}
