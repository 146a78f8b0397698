#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_empty(float* out) { __shared__ float l[7552]; if (out == nullptr) l[threadIdx.x] = 1.f; }
__global__ __launch_bounds__(256) void k_store(float* out) {
  __shared__ float l[7552]; if (out == nullptr) l[threadIdx.x] = 1.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* o = out + (size_t)blockIdx.x * 4096 + wave * 1024 + lane % 32 + (lane / 32) * 4 * 32;   // 16 rows x 32 px pattern-ish
  for (int r = 0; r < 16; ++r) o[(r % 4) * 32 + (r / 4) * 256] = r;
}
__global__ __launch_bounds__(256) void k_load_store(const float* in, float* out) {
  __shared__ float l[7552];
  for (int i = threadIdx.x; i < 3744; i += 256) l[i] = in[(size_t)(blockIdx.x % 64) * 3744 + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* o = out + (size_t)blockIdx.x * 4096 + wave * 1024 + lane % 32 + (lane / 32) * 4 * 32;
  const float v = l[threadIdx.x];
  for (int r = 0; r < 16; ++r) o[(r % 4) * 32 + (r / 4) * 256] = v + r;
}
int main() {
  float *in, *out; hipMalloc(&in, 64 * 3744 * 4); hipMalloc(&out, 2048 * 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {512, 2048}) {
    for (int which = 0; which < 3; ++which) {
      auto launch = [&]() {
        if (which == 0) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, 0, out);
        if (which == 1) hipLaunchKernelGGL(k_store, dim3(blocks), dim3(256), 0, 0, out);
        if (which == 2) hipLaunchKernelGGL(k_load_store, dim3(blocks), dim3(256), 0, 0, in, out);
      };
      for (int i = 0; i < 10; ++i) launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 200; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("blocks %4d  %-12s %.2f us per back-to-back launch\n", blocks, which == 0 ? "empty" : which == 1 ? "store 16/thr" : "load+store", ms * 1000 / 200);
    }
  }
  return 0;
}
