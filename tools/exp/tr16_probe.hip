// Probe of ds_read_b64_tr_b16 (gfx950): every lane supplies the address of its own 8-byte chunk (lane * 8 bytes); prints, per lane,
// which source elements (chunk lane, element) the 4 result elements came from.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;       // element e of chunk c holds 4 c + e
  __syncthreads();
  typedef __attribute__((address_space(3))) s16x4* lp;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds + threadIdx.x * 4));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf("  (c%2d,e%d)", h[l * 4 + j] / 4, h[l * 4 + j] % 4);
    printf("\n");
  }
  return 0;
}
