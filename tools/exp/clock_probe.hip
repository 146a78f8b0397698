// What shader clock does the MI355X hold under the loads of this library?  The 157.3 TFLOP/s exact-f32 MFMA peak every roofline fraction
// here is quoted against assumes 2.4 GHz.  Each block reads the shader-clock counter (s_memtime, clock64) and the constant 100 MHz
// reference counter (s_memrealtime, wall_clock64) at its start and end: d(clock64) / d(wall_clock64) x 100 MHz = the clock it ran at.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MEM>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ src, long n, int iters, long long* t) {
  const long long c0 = clock64(), w0 = wall_clock64();
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  long idx = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  for (int it = 0; it < iters; ++it) {
    if (MEM) {                                   // stream 16 bytes per lane per 16 MFMAs (~ the byte / flop ratio of the widest gate level)
      const float4 v = *(const float4*)(src + (idx % n));
      a += v.x * 1e-9f; b += v.y * 1e-9f;
      idx += (long)gridDim.x * 1024;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = clock64() - c0; t[2 * blockIdx.x + 1] = wall_clock64() - w0; }
}

int main() {
  const int blocks = 256 * 8;
  float *out, *src; long long* t;
  const long n = 1L << 28;                       // 1 GiB of floats to stream through
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&src, n * 4); hipMalloc(&t, blocks * 16);
  hipMemset(src, 0, n * 4);
  std::vector<long long> h(2 * blocks);
  for (int mem = 0; mem < 1; ++mem)           // (the streaming variant's loads feed the MFMA operands: it measures load latency, not a clock)
    for (int rep = 0; rep < 6; ++rep) {
      const int iters = mem ? 60000 : 200000;    // ~100-150 ms per launch: long enough for the power management to settle
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mem) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, src, n, iters, t);
      else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, src, n, iters, t);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h.data(), t, blocks * 16, hipMemcpyDeviceToHost);
      std::vector<double> f;
      for (int b = 0; b < blocks; ++b) f.push_back(100.0 * (double)h[2 * b] / (double)h[2 * b + 1]);
      std::sort(f.begin(), f.end());
      const double flops = (double)blocks * 4 * iters * 16 * 4096.0;      // v_mfma_f32_32x32x2_f32 = 32 x 32 x 2 MACs
      printf("%s rep %d: %.1f ms, %.1f TFLOP/s; shader clock (MHz) from clock64 / wall_clock64: min %.0f median %.0f max %.0f -> peak at that clock %.1f TFLOP/s\n",
             mem ? "MFMA + 16 B/lane stream" : "pure MFMA             ", rep, ms, flops / ms / 1e9, f.front(), f[f.size() / 2], f.back(),
             f[f.size() / 2] * 1e6 * 65536 / 1e12);
    }
  return 0;
}
