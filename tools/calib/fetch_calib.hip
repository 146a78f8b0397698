// Known-byte-count micro-kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the RSIS
// kernels use (MI355X_MICROARCH.md calibrates only the 16 B/lane streaming read: "FETCH_SIZE reports exactly 1/2 of the bytes";
// other widths and WRITE_SIZE are "uncalibrated").  Every kernel touches each byte of an n-byte buffer exactly once.
#include <hip/hip_runtime.h>
typedef __attribute__((address_space(3))) void* lds_vp_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: buffer_load_dword ... lds (4 B/lane LDS-DMA, the direct 3x3 kernel's patch fetch)     1: buffer_load_dwordx4 ... lds
// mode 2: global_load_dword to VGPR                                                             3: global_load_dwordx4 to VGPR
template <int MODE>
__global__ __launch_bounds__(256) void read_kernel(const float* __restrict__ x, float* __restrict__ sink, long n_floats) {
#if __HIP_DEVICE_COMPILE__
  __shared__ __attribute__((aligned(16))) float lds[4096];
  constexpr int VEC = (MODE & 1) ? 4 : 1;
  const long per_block = 256L * VEC * 16;                     // 16 loads per thread per block-iteration
  float acc = 0.f;
  for (long base = blockIdx.x * per_block; base < n_floats; base += (long)gridDim.x * per_block) {
    if constexpr (MODE < 2) {
      const long rem = n_floats - base;
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(x + base), 0, (int)(rem > per_block ? per_block : rem) * 4, 0x00020000);
      const int wave = threadIdx.x >> 6;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const unsigned off = (unsigned)((i * 256 + threadIdx.x) * VEC) * 4u;
        float* dst = lds + ((i & 3) * 256 + wave * 64) * VEC;
        if constexpr (VEC == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vp_t)dst, 4, off, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vp_t)dst, 16, off, 0, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);
      acc += lds[threadIdx.x];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const long e = base + (long)(i * 256 + threadIdx.x) * VEC;
        if (e + VEC <= n_floats) {
          if constexpr (VEC == 1) acc += x[e];
          else { const f32x4 v = *reinterpret_cast<const f32x4*>(x + e); acc += v[0] + v[1] + v[2] + v[3]; }
        }
      }
    }
  }
  if (acc == 123456.789f) sink[0] = acc;                      // keep the loads alive
#endif
}

// mode 0: global_store_dword   1: global_store_dwordx4   2: dword stores in 128-byte runs separated by a stride (the NCHW epilogue
// pattern of the MFMA kernels: 32 lanes x 4 B per (channel, row) segment)
template <int MODE>
__global__ __launch_bounds__(256) void write_kernel(float* __restrict__ y, long n_floats) {
  constexpr int VEC = MODE == 1 ? 4 : 1;
  for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * VEC; e + VEC <= n_floats; e += (long)gridDim.x * 256 * VEC) {
    long a = e;
    if (MODE == 2) {       // permute 32-float runs: run r goes to run (r * 17) mod nruns -- same bytes, scattered 128-byte segments
      const long nruns = n_floats / 32, r = e / 32;
      a = ((r * 17) % nruns) * 32 + (e & 31);
    }
    if constexpr (VEC == 1) y[a] = 1.0f;
    else *reinterpret_cast<f32x4*>(y + a) = f32x4{1.f, 2.f, 3.f, 4.f};
  }
}

extern "C" int calib_read(int mode, const float* x, float* sink, long n_floats, void* stream) {
  const dim3 g(2048), b(256);
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case 0: hipLaunchKernelGGL(read_kernel<0>, g, b, 0, st, x, sink, n_floats); break;
    case 1: hipLaunchKernelGGL(read_kernel<1>, g, b, 0, st, x, sink, n_floats); break;
    case 2: hipLaunchKernelGGL(read_kernel<2>, g, b, 0, st, x, sink, n_floats); break;
    case 3: hipLaunchKernelGGL(read_kernel<3>, g, b, 0, st, x, sink, n_floats); break;
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
extern "C" int calib_write(int mode, float* y, long n_floats, void* stream) {
  const dim3 g(4096), b(256);
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case 0: hipLaunchKernelGGL(write_kernel<0>, g, b, 0, st, y, n_floats); break;
    case 1: hipLaunchKernelGGL(write_kernel<1>, g, b, 0, st, y, n_floats); break;
    case 2: hipLaunchKernelGGL(write_kernel<2>, g, b, 0, st, y, n_floats); break;
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
