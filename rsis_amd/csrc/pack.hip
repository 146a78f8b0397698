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
#include "../../include/rsis_hip.h"

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

// ---- one element of each packed layout (shared by the per-conv kernels and the batched repack) ----
__device__ __forceinline__ float pack_elem(int mode, const float* __restrict__ W, long e, int Cout, int Ctot, int KK, const SegMap& m,
                                           int ldw, int hid) {
  const int row = (int)(e / ldw), col = (int)(e - (long)row * ldw);
  if (mode == 0) {                 // igemm fwd: Wp[cg*KK + rs][co_p]
    if (col >= Cout) return 0.f;
    const int cg = row / KK, rs = row - cg * KK;
    const int ci = seg_channel(m, cg);
    return ci >= 0 ? W[((long)ref_row(col, hid) * Ctot + ci) * KK + rs] : 0.f;
  }
  if (mode == 1) {                 // igemm dgrad: Wd[co_p*KK + rs][cg]
    const int cop = row / KK, rs = row - cop * KK;
    const int ci = seg_channel(m, col);
    return (cop < Cout && ci >= 0) ? W[((long)ref_row(cop, hid) * Ctot + ci) * KK + rs] : 0.f;
  }
  const int qg = row / (RSIS_CK * 9), kin = row - qg * (RSIS_CK * 9);
  const int pair = kin >> 1, h = kin & 1;
  const int cc = pair / 9, rs = pair - cc * 9;
  if (mode == 2) {                 // direct fwd: 8-channel chunks per segment, columns = co_p
    if (col >= Cout) return 0.f;
    int qs = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (s < m.n) {
        const int nq = (m.C[s] + RSIS_CK - 1) / RSIS_CK;
        if (qg >= qs && qg < qs + nq) {
          const int c = (qg - qs) * RSIS_CK + 2 * cc + h;
          return c < m.C[s] ? W[((long)ref_row(col, hid) * Ctot + m.off[s] + c) * 9 + rs] : 0.f;
        }
        qs += nq;
      }
    }
    return 0.f;
  }
  // direct dgrad: rows = chunks of co_p, columns = cg.  mode 3 (stride 1) reads tap 8-rs (the transposed conv); mode 4 (stride 2,
  // EPI_S2 of conv3x3_direct.hip) keeps the original tap order (each tap is routed to its parity class by the kernel)
  const int c = qg * RSIS_CK + 2 * cc + h;      // channel of dy (packed row order for ConvLSTM)
  const int ci = seg_channel(m, col);
  return (c < Cout && ci >= 0) ? W[((long)ref_row(c, hid) * Ctot + ci) * 9 + (mode == 3 ? 8 - rs : rs)] : 0.f;
}

__global__ void pack_kernel(int mode, const float* __restrict__ W, float* __restrict__ out, int Cout, int Ctot, int KK, SegMap m, int ldw,
                            int krows, int hid) {
  const long total = (long)krows * ldw;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    out[e] = pack_elem(mode, W, e, Cout, Ctot, KK, m, ldw, hid);
}

// ---- batched repack: every packed copy of every conv weight in ONE launch (after an optimizer step ~240 tiny pack launches
// per training step otherwise).  jobs[] lives in device memory; job i owns the blocks [block_begin_i, block_begin_{i+1}). ----
#define PACK_CHUNK 4096   // elements per block
__global__ __launch_bounds__(256) void pack_batch_kernel(const rsis_pack_job* __restrict__ jobs, int njobs) {
  int lo = 0, hi = njobs - 1;
  const int b = blockIdx.x;
  while (lo < hi) {                       // last job whose block_begin <= b
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_begin <= b) lo = mid; else hi = mid - 1;
  }
  const rsis_pack_job j = jobs[lo];
  SegMap m;
  m.n = j.nseg;
#pragma unroll
  for (int s = 0; s < 3; ++s) { m.C[s] = j.Cseg[s]; m.off[s] = j.Coff[s]; }
  const long total = (long)j.krows * j.ldw;
  const long base = (long)(b - j.block_begin) * PACK_CHUNK;
#pragma unroll 4
  for (int i = 0; i < PACK_CHUNK / 256; ++i) {
    const long e = base + i * 256 + threadIdx.x;
    if (e < total) j.out[e] = pack_elem(j.imode, j.W, e, j.Cout, j.Ctot, j.ks * j.ks, m, j.ldw, j.lstm_hid);
  }
}

int rsis_l_pack_batch(const rsis_pack_job* jobs, int njobs, int total_blocks, hipStream_t st) {
  hipLaunchKernelGGL(pack_batch_kernel, dim3(total_blocks), dim3(256), 0, st, jobs, njobs);
  return rsis_check_launch();
}
int rsis_l_pack_chunk() { return PACK_CHUNK; }

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
  if (mode < 0 || mode > 4) return RSIS_ERR_ARG;
  hipLaunchKernelGGL(pack_kernel, g, b, 0, st, mode, W, out, Cout, Ctot, ks * ks, m, ldw, krows, hid);
  return rsis_check_launch();
}
