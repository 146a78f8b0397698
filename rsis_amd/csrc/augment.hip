// Data-layer augmentation on the device: the nearest-neighbour affine warp the reference applies to the image,
// the instance map and the class map of every training sample (reference src/dataloader/transforms/utils.py:67-147
// th_affine2d(mode='nearest', center=True) + th_nearest_interp2d; called from dataloader/pascal.py:47-51 through
// transforms.py:23-142 RandomAffine -> Affine).  Pure gather: output pixel (i, j) of every channel of sample n reads the input
// pixel at  round_half_even(clamp(A_n (i - ci, j - cj) + b_n + (ci, cj))),  ci = H/2 - 0.5, cj = W/2 - 0.5, in float32 with
// the reference's operation order (separately rounded products and sums: no FMA contraction, the rounding decides which pixel
// is read).  Bound: HBM (one read + one write of the tensor); one thread per output pixel, all channels.
#include "common.h"

__global__ __launch_bounds__(256) void affine_nearest_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             const float* __restrict__ mat, int C, int H, int W, int mat_stride) {
  const int n = blockIdx.y;
  const int HW = H * W;
  const float* m = mat + (size_t)n * mat_stride;          // row-major 3x3 (or 2x3): [a00 a01 b0; a10 a11 b1; ...]
  const float a00 = m[0], a01 = m[1], b0 = m[2], a10 = m[3], a11 = m[4], b1 = m[5];
  const float ci = (float)(H / 2.0 - 0.5), cj = (float)(W / 2.0 - 0.5);
  for (int e = blockIdx.x * 256 + threadIdx.x; e < HW; e += gridDim.x * 256) {
#pragma clang fp contract(off)      // the reference rounds every product and sum: a fused multiply-add can flip a .5 tie
    const int i = e / W, j = e - i * W;
    const float fi = __fsub_rn((float)i, ci), fj = __fsub_rn((float)j, cj);
    float ni = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(fi, a00), __fmul_rn(fj, a01)), b0), ci);
    float nj = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(fi, a10), __fmul_rn(fj, a11)), b1), cj);
    ni = fminf(fmaxf(ni, 0.f), (float)(H - 1));
    nj = fminf(fmaxf(nj, 0.f), (float)(W - 1));
    const int src = (int)rintf(ni) * W + (int)rintf(nj);
    const float* xs = x + (size_t)n * C * HW + src;
    float* yd = y + (size_t)n * C * HW + e;
    for (int c = 0; c < C; ++c) yd[(size_t)c * HW] = xs[(size_t)c * HW];
  }
}

int rsis_l_affine_nearest(const float* x, float* y, const float* mat, int N, int C, int H, int W, int mat_stride, hipStream_t st) {
  long gx = ((long)H * W + 255) / 256;
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(affine_nearest_kernel, dim3((unsigned)gx, (unsigned)N), dim3(256), 0, st, x, y, mat, C, H, W, mat_stride);
  return rsis_check_launch();
}
