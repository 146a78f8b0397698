// Micro-benchmark: what the exact-f32 MFMA (v_mfma_f32_32x32x2_f32) of gfx950 sustains under the operand-feed patterns the
// library's kernels use.  Build + run on the GPU box:   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// NACC independent accumulators per wave, operands in registers (no memory traffic at all)
template <int NACC>
__global__ __launch_bounds__(256) void k_pure(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// TM x TN accumulators per wave; per K-step TM + TN ds_read_b32 from a padded LDS tile (the wgrad / igemm feed)
template <int TM, int TN>
__global__ __launch_bounds__(256) void k_lds(float* out, int iters) {
  constexpr int LDK = 33;
  __shared__ float As[128 * LDK], Bs[128 * LDK];
  for (int i = threadIdx.x; i < 128 * LDK; i += 256) { As[i] = i * 1e-4f; Bs[i] = i * 2e-4f; }
  __syncthreads();
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5, wave = threadIdx.x >> 6;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const float* Ap = As + ((wave & 1) * TM * 32 % 128 + l31) * LDK;
  const float* Bp = Bs + ((wave >> 1) * TN * 32 % 128 + l31) * LDK;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int k = kk * 2 + hi;
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = Ap[(i * 32 % 96) * LDK + k];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bp[(j * 32 % 96) * LDK + k];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// same, plus a __syncthreads() every 16 K-steps (the per-K-tile barrier of a double-buffered main loop)
template <int TM, int TN>
__global__ __launch_bounds__(256) void k_lds_sync(float* out, int iters) {
  constexpr int LDK = 33;
  __shared__ float As[128 * LDK], Bs[128 * LDK];
  for (int i = threadIdx.x; i < 128 * LDK; i += 256) { As[i] = i * 1e-4f; Bs[i] = i * 2e-4f; }
  __syncthreads();
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5, wave = threadIdx.x >> 6;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const float* Ap = As + ((wave & 1) * TM * 32 % 128 + l31) * LDK;
  const float* Bp = Bs + ((wave >> 1) * TN * 32 % 128 + l31) * LDK;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int k = kk * 2 + hi;
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = Ap[(i * 32 % 96) * LDK + k];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bp[(j * 32 % 96) * LDK + k];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// shader clock during a kernel: clock64() ticks (shader clock) per wall_clock64() tick (constant 100 MHz)
__global__ void k_clock_probe(long long* out, int iters) {
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)s; }
}

template <typename F>
static void run(const char* name, F launch, double mfma_per_wave_iter, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch(blocks, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) launch(blocks, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 5.0 * blocks * 4 * (double)iters * mfma_per_wave_iter * 32 * 32 * 2 * 2;
  printf("%-34s blocks %5d  %8.3f ms  %7.1f TF/s\n", name, blocks, ms / 5, flops / (ms * 1e-3) / 1e12);
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * sizeof(float));
  {
    long long* d; long long h[3];
    hipMalloc(&d, 3 * sizeof(long long));
    for (int blocks : {1, 256, 1024}) {
      hipLaunchKernelGGL(k_clock_probe, dim3(blocks), dim3(256), 0, 0, d, 20000);
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("clock probe, %4d blocks of MFMA work: %.0f MHz shader clock (clock64 / wall_clock64 @100 MHz), 16 MFMA per %.1f clk\n", blocks,
             100.0 * (double)h[0] / (double)h[1], (double)h[0] / 20000.0);
    }
  }
  const int it = 2000;
  for (int bpc : {1, 2, 3, 4}) {
    const int blocks = 256 * bpc;
    printf("-- %d block(s) of 4 waves per CU\n", bpc);
    run("pure NACC=1", [&](int b, int n) { hipLaunchKernelGGL(k_pure<1>, dim3(b), dim3(256), 0, 0, out, n); }, 8, blocks, it);
    run("pure NACC=2", [&](int b, int n) { hipLaunchKernelGGL(k_pure<2>, dim3(b), dim3(256), 0, 0, out, n); }, 16, blocks, it);
    run("pure NACC=4", [&](int b, int n) { hipLaunchKernelGGL(k_pure<4>, dim3(b), dim3(256), 0, 0, out, n); }, 32, blocks, it);
    run("lds 1x1", [&](int b, int n) { hipLaunchKernelGGL((k_lds<1, 1>), dim3(b), dim3(256), 0, 0, out, n); }, 16, blocks, it);
    run("lds 1x2", [&](int b, int n) { hipLaunchKernelGGL((k_lds<1, 2>), dim3(b), dim3(256), 0, 0, out, n); }, 32, blocks, it);
    run("lds 2x2", [&](int b, int n) { hipLaunchKernelGGL((k_lds<2, 2>), dim3(b), dim3(256), 0, 0, out, n); }, 64, blocks, it / 2);
    run("lds+sync 1x1", [&](int b, int n) { hipLaunchKernelGGL((k_lds_sync<1, 1>), dim3(b), dim3(256), 0, 0, out, n); }, 16, blocks, it);
    run("lds+sync 2x2", [&](int b, int n) { hipLaunchKernelGGL((k_lds_sync<2, 2>), dim3(b), dim3(256), 0, 0, out, n); }, 64, blocks, it / 2);
  }
  return 0;
}
