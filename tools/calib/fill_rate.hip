// Per-CU ingest rate microbenchmark (gfx950): how fast can one CU pull L2-resident data
//   (a) into LDS with the LDS-DMA (buffer_load_dwordx4 ... lds), (b) into VGPRs with buffer_load_dwordx4,
// with 1 / 2 / 4 blocks of 256 threads per CU and a working set small enough to stay in L2 (every block re-reads the same 64 KB).
// build: hipcc --offload-arch=gfx950 -O3 -o fill_rate fill_rate.hip ; run: ./fill_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_vp_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void fill_kernel(const float* __restrict__ src, float* __restrict__ sink, int iters, int bytes_per_block) {
#if __HIP_DEVICE_COMPILE__
  __shared__ __attribute__((aligned(16))) float lds[8192];     // 32 KB
  const int tid = threadIdx.x, wave = tid >> 6;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes_per_block, 0x00020000);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nchunk = bytes_per_block / (256 * 16 * 8);        // 8 x b128 per thread per chunk = 32 KB per block-chunk
  for (int it = 0; it < iters; ++it) {
    for (int c = 0; c < nchunk; ++c) {
      const unsigned base = (unsigned)c * 32768u + (unsigned)tid * 16u;
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vp_t)(lds + k * 1024 + wave * 256), 16, base + k * 4096u, 0, 0, 0);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, base + k * 4096u, 0, 0));
          acc += v;
        }
      }
    }
  }
  if (MODE == 0) {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    acc[0] = lds[tid];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[tid] = acc[0];
#endif
}

int main() {
  const int bytes = 65536;
  float *src, *sink;
  hipMalloc(&src, bytes); hipMalloc(&sink, 4096);
  hipMemset(src, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
      const int blocks = 256 * bpc, iters = 200;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(fill_kernel<0>, dim3(blocks), dim3(256), 0, 0, src, sink, iters, bytes);
        else hipLaunchKernelGGL(fill_kernel<1>, dim3(blocks), dim3(256), 0, 0, src, sink, iters, bytes);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double total = (double)blocks * iters * bytes;
      printf("%s  %d block(s)/CU: %.1f GB/s per CU, %.2f TB/s chip\n", mode == 0 ? "LDS-DMA b128" : "VGPR   b128", bpc,
             total / (ms * 1e-3) / 256 / 1e9, total / (ms * 1e-3) / 1e12);
    }
  return 0;
}
