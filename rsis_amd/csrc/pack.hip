// Weight repacking for the gfx950 conv kernels: private MFMA-friendly copies of an nn.Conv2d weight kept in the
// reference layout W[Cout][Ctot][ks][ks] (reference src/modules/clstm.py:17, model.py:43-47,109; torchvision trunk).
// Never serialised; rebuilt whenever the weight changes.
//
// A packed copy covers a list of up to 3 input-channel SEGMENTS (channel offset + count) of the weight: the channel
// concat of the tensors the conv will gather from (torch.cat folded into the kernel) -- or any subset of the input
// channels (the ConvLSTM path packs the time-invariant skip channels and the recurrent channels separately).
//   implicit-GEMM layout:  FWD  Wp[cg*KK + rs][co_p],  DGRAD  Wd[co_p*KK + rs][cg]           (cg = index in the concat)
//   direct-3x3 layout   :  8-channel chunks per segment (zero padded); inside a chunk row (c2*9 + rs)*2 + h is channel
//                          c0 + 2*c2 + h, tap rs.  FWD columns = co_p; DGRAD rows = chunks of co_p, columns = cg, tap 8-rs.
// co_p -> reference row: (co_p&3)*hid + (co_p>>2) for gate-interleaved ConvLSTM rows (clstm.py:47), identity otherwise.
#include "common.h"

struct SegMap { int n; int C[3]; int off[3]; };

__device__ __forceinline__ int seg_channel(const SegMap& m, int cg) {   // concat index -> weight input channel (-1: none)
  int base = 0;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    if (s < m.n) {
      if (cg < base + m.C[s]) return m.off[s] + (cg - base);
      base += m.C[s];
    }
  }
  return -1;
}
__device__ __forceinline__ int ref_row(int cop, int hid) { return hid > 0 ? (cop & 3) * hid + (cop >> 2) : cop; }

__global__ void pack_fwd_kernel(const float* __restrict__ W, float* __restrict__ Wp, int Cout, int Ctot, int KK, SegMap m,
                                int ldw, int krows, int hid) {
  const long total = (long)krows * ldw;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / ldw), cop = (int)(e - (long)k * ldw);
    float v = 0.f;
    if (cop < Cout) {
      const int cg = k / KK, rs = k - cg * KK;
      const int ci = seg_channel(m, cg);
      if (ci >= 0) v = W[((long)ref_row(cop, hid) * Ctot + ci) * KK + rs];
    }
    Wp[e] = v;
  }
}

__global__ void pack_dgrad_kernel(const float* __restrict__ W, float* __restrict__ Wd, int Cout, int Ctot, int KK, SegMap m,
                                  int ldw, int krows, int hid) {
  const long total = (long)krows * ldw;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e / ldw), cg = (int)(e - (long)k * ldw);
    float v = 0.f;
    const int cop = k / KK, rs = k - cop * KK;
    const int ci = seg_channel(m, cg);
    if (cop < Cout && ci >= 0) v = W[((long)ref_row(cop, hid) * Ctot + ci) * KK + rs];
    Wd[e] = v;
  }
}

__global__ void pack_direct_fwd_kernel(const float* __restrict__ W, float* __restrict__ Wp, int Cout, int Ctot, SegMap m, int ldw,
                                       int krows, int hid) {
  const long total = (long)krows * ldw;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int row = (int)(e / ldw), col = (int)(e - (long)row * ldw);
    float v = 0.f;
    if (col < Cout) {
      const int qg = row / (RSIS_CK * 9), kin = row - qg * (RSIS_CK * 9);
      const int pair = kin >> 1, h = kin & 1;
      const int cc = pair / 9, rs = pair - cc * 9;
      int qs = 0;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        if (s < m.n) {
          const int nq = (m.C[s] + RSIS_CK - 1) / RSIS_CK;
          if (qg >= qs && qg < qs + nq) {
            const int c = (qg - qs) * RSIS_CK + 2 * cc + h;
            if (c < m.C[s]) v = W[((long)ref_row(col, hid) * Ctot + m.off[s] + c) * 9 + rs];
          }
          qs += nq;
        }
      }
    }
    Wp[e] = v;
  }
}

// flip: stride-1 dgrad reads tap 8-rs (the transposed conv); the stride-2 dgrad (EPI_S2 of conv3x3_direct.hip) keeps the
// original tap order (each tap is routed to its parity class by the kernel)
__global__ void pack_direct_dgrad_kernel(const float* __restrict__ W, float* __restrict__ Wd, int Cout, int Ctot, SegMap m, int ldw,
                                         int krows, int hid, int flip) {
  const long total = (long)krows * ldw;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int row = (int)(e / ldw), cg = (int)(e - (long)row * ldw);
    float v = 0.f;
    const int qg = row / (RSIS_CK * 9), kin = row - qg * (RSIS_CK * 9);
    const int pair = kin >> 1, h = kin & 1;
    const int cc = pair / 9, rs = pair - cc * 9;
    const int c = qg * RSIS_CK + 2 * cc + h;      // channel of dy (packed row order for ConvLSTM)
    const int ci = seg_channel(m, cg);
    if (c < Cout && ci >= 0) v = W[((long)ref_row(c, hid) * Ctot + ci) * 9 + (flip ? 8 - rs : rs)];
    Wd[e] = v;
  }
}

static inline int pack_grid(long total) {
  long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  return (int)(g < 1 ? 1 : g);
}

// mode: 0 igemm fwd, 1 igemm dgrad, 2 direct fwd, 3 direct dgrad (stride 1), 4 direct dgrad (stride 2: taps not flipped)
int rsis_l_pack(int mode, const float* W, float* out, int Cout, int Ctot, int ks, int nseg, const int* Cseg, const int* Coff,
                int ldw, int krows, int hid, hipStream_t st) {
  SegMap m = {};
  m.n = nseg;
  int base = 0;
  for (int s = 0; s < nseg; ++s) { m.C[s] = Cseg[s]; m.off[s] = Coff ? Coff[s] : base; base += Cseg[s]; }
  const long total = (long)krows * ldw;
  const dim3 g(pack_grid(total)), b(256);
  switch (mode) {
    case 0: hipLaunchKernelGGL(pack_fwd_kernel, g, b, 0, st, W, out, Cout, Ctot, ks * ks, m, ldw, krows, hid); break;
    case 1: hipLaunchKernelGGL(pack_dgrad_kernel, g, b, 0, st, W, out, Cout, Ctot, ks * ks, m, ldw, krows, hid); break;
    case 2: hipLaunchKernelGGL(pack_direct_fwd_kernel, g, b, 0, st, W, out, Cout, Ctot, m, ldw, krows, hid); break;
    case 3: hipLaunchKernelGGL(pack_direct_dgrad_kernel, g, b, 0, st, W, out, Cout, Ctot, m, ldw, krows, hid, 1); break;
    case 4: hipLaunchKernelGGL(pack_direct_dgrad_kernel, g, b, 0, st, W, out, Cout, Ctot, m, ldw, krows, hid, 0); break;
    default: return RSIS_ERR_ARG;
  }
  return rsis_check_launch();
}
