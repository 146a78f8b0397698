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
// (the element functions return the INDEX into the reference weight, or -1 for a zero of the padding: the batched repack fetches a
//  handful of elements per thread at clamped indices before it uses any of them -- `cond ? W[i] : 0` element by element compiles to
//  a branch around every load and a wait behind it, one memory round trip per element)
template <int KKC = 0>    // KKC: compile-time ks*ks (0 = use the run-time value) -- makes the row / KK divisions cheap
__device__ __forceinline__ int pack_elem_idx(int mode, int row, int col, int Cout, int Ctot, int KKr, const SegMap& m, int hid) {
  const int KK = KKC ? KKC : KKr;
  if (mode == 0) {                 // igemm fwd: Wp[cg*KK + rs][co_p]
    if (col >= Cout) return -1;
    const int cg = row / KK, rs = row - cg * KK;
    const int ci = seg_channel(m, cg);
    return ci >= 0 ? (ref_row(col, hid) * Ctot + ci) * KK + rs : -1;
  }
  if (mode == 1) {                 // igemm dgrad: Wd[co_p*KK + rs][cg]
    const int cop = row / KK, rs = row - cop * KK;
    const int ci = seg_channel(m, col);
    return (cop < Cout && ci >= 0) ? (ref_row(cop, hid) * Ctot + ci) * KK + rs : -1;
  }
  const int qg = row / (RSIS_CK * 9), kin = row - qg * (RSIS_CK * 9);
  const int pair = kin >> 1, h = kin & 1;
  const int cc = pair / 9, rs = pair - cc * 9;
  if (mode == 2) {                 // direct fwd: 8-channel chunks per segment, columns = co_p
    if (col >= Cout) return -1;
    int qs = 0, r = -1;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (s < m.n) {
        const int nq = (m.C[s] + RSIS_CK - 1) / RSIS_CK;
        if (qg >= qs && qg < qs + nq) {
          const int c = (qg - qs) * RSIS_CK + 2 * cc + h;
          if (c < m.C[s]) r = (ref_row(col, hid) * Ctot + m.off[s] + c) * 9 + rs;
        }
        qs += nq;
      }
    }
    return r;
  }
  // direct dgrad: rows = chunks of co_p, columns = cg.  mode 3 (stride 1) reads tap 8-rs (the transposed conv); mode 4 (stride 2,
  // EPI_S2 of conv3x3_direct.hip) keeps the original tap order (each tap is routed to its parity class by the kernel)
  const int c = qg * RSIS_CK + 2 * cc + h;      // channel of dy (packed row order for ConvLSTM)
  const int ci = seg_channel(m, col);
  return (c < Cout && ci >= 0) ? (ref_row(c, hid) * Ctot + ci) * 9 + (mode == 3 ? 8 - rs : rs) : -1;
}
template <int KKC = 0>
__device__ __forceinline__ float pack_elem(int mode, const float* __restrict__ W, int row, int col, int Cout, int Ctot, int KKr,
                                           const SegMap& m, int hid) {
  const int i = pack_elem_idx<KKC>(mode, row, col, Cout, Ctot, KKr, m, hid);
  const float v = W[i < 0 ? 0 : i];
  return i < 0 ? 0.f : v;
}

__global__ void pack_kernel(int mode, const float* __restrict__ W, float* __restrict__ out, int Cout, int Ctot, int KK, SegMap m, int ldw,
                            int krows, int hid) {
  const long total = (long)krows * ldw;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int row = (int)(e / ldw), col = (int)(e - (long)row * ldw);
    out[e] = pack_elem(mode, W, row, col, Cout, Ctot, KK, m, hid);
  }
}

// ---- bf16 layouts (conv_bf16.hip): 16-byte cells of 8 consecutive reduction channels, out[(kb * ldw + col) * 8 + e] ----
//   kb = (q * KK + rs) * NCB + cb: chunk q of CKB channels (CKB = RSIS_CKB3 for 3x3, RSIS_CKB1 for 1x1; chunks never straddle
//   a concat segment, the tail of a segment is zero), tap rs, 8-channel block cb of the chunk; NCB = CKB / 8.
//   mode 5 (forward): reduction channel = input channel of the concat, col = co_p.
//   mode 6 (data gradient): reduction channel = dy channel co_p, col = cg (index in the concat), 3x3 taps flipped.
typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned pk_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned short to_bf16(float v) {
  const pk_f32x2 f = {v, 0.f};
  return (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(f, pk_bf16x2)) & 0xFFFFu);
}
__host__ __device__ __forceinline__ int bf16_ckb(int KK) { return KK == 1 ? RSIS_CKB1 : RSIS_CKB3; }

__device__ __forceinline__ int pack_bf16_idx(int mode, int kb, int col, int e, int Cout, int Ctot, int KK, const SegMap& m, int hid) {
  const int CKB = bf16_ckb(KK), NCB = CKB / 8;
  const int q = kb / (KK * NCB), rem = kb - q * (KK * NCB);
  const int rs = rem / NCB, cb = rem - rs * NCB;
  const int cl = cb * 8 + e;                         // channel inside the chunk
  if (mode == 5) {
    if (col >= Cout) return -1;
    int qs = 0, r = -1;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (s < m.n) {
        const int nq = (m.C[s] + CKB - 1) / CKB;
        if (q >= qs && q < qs + nq) {
          const int c = (q - qs) * CKB + cl;
          if (c < m.C[s]) r = (ref_row(col, hid) * Ctot + m.off[s] + c) * KK + rs;
        }
        qs += nq;
      }
    }
    return r;
  }
  const int c = q * CKB + cl;                        // dy channel (packed row order for ConvLSTM)
  const int ci = seg_channel(m, col);
  return (c < Cout && ci >= 0) ? (ref_row(c, hid) * Ctot + ci) * KK + (KK == 9 ? 8 - rs : rs) : -1;
}
__device__ __forceinline__ float pack_bf16_src(int mode, const float* __restrict__ W, int kb, int col, int e, int Cout, int Ctot, int KK,
                                               const SegMap& m, int hid) {
  const int i = pack_bf16_idx(mode, kb, col, e, Cout, Ctot, KK, m, hid);
  const float v = W[i < 0 ? 0 : i];
  return i < 0 ? 0.f : v;
}

__global__ void pack_bf16_kernel(int mode, const float* __restrict__ W, unsigned short* __restrict__ out, int Cout, int Ctot, int KK,
                                 SegMap m, int ldw, int nkb, int hid) {
  const long total = (long)nkb * ldw * 8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7);
    const long cell = i >> 3;
    const int kb = (int)(cell / ldw), col = (int)(cell - (long)kb * ldw);
    out[i] = to_bf16(pack_bf16_src(mode, W, kb, col, e, Cout, Ctot, KK, m, hid));
  }
}

// batched repack of the bf16 layouts.  Forward (mode 5): one block = one channel chunk (KK * NCB cell rows) x 64 columns; for a
// column the chunk's KK * CKB source weights are one contiguous run of the reference weight, read with the lanes running along
// it, converted, parked in LDS in cell order and written out as full 1 KB rows.  Data gradient (mode 6): 16 cell rows x 64
// columns per block; the lanes run along the columns (= the reference weight's input channels), reads and writes coalesce.
__device__ __forceinline__ void pack_tile_bf16(const rsis_pack_job& j, int tb, unsigned short* lds16) {
  SegMap m;
  m.n = j.nseg;
#pragma unroll
  for (int s = 0; s < 3; ++s) { m.C[s] = j.Cseg[s]; m.off[s] = j.Coff[s]; }
  const int KK = j.ks * j.ks, CKB = bf16_ckb(KK), NCB = CKB / 8;
  const int nct = j.ldw / 64;
  const int c0 = (tb % nct) * 64;
  unsigned short* out = (unsigned short*)j.out;
  if (j.imode == 5) {
    const int R = KK * NCB;                          // cell rows of one chunk (18 or 8)
    const int q = tb / nct;
    const int run = KK * CKB;                        // source floats per column
    for (int i0 = threadIdx.x; i0 < 64 * run; i0 += 256 * 4) {      // four elements per thread in flight
      int src[4], dst[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 256 * u;
        const int cl = i / run, idx = i - cl * run;    // column, position in the run = (channel in chunk, tap)
        const int ch = idx / KK, rs = idx - ch * KK;
        const int kb = q * R + rs * NCB + (ch >> 3);
        dst[u] = i < 64 * run ? ((rs * NCB + (ch >> 3)) * 64 + cl) * 8 + (ch & 7) : -1;
        src[u] = i < 64 * run ? pack_bf16_idx(5, kb, c0 + cl, ch & 7, j.Cout, j.Ctot, KK, m, j.lstm_hid) : -1;
      }
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = j.W[src[u] < 0 ? 0 : src[u]];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (dst[u] >= 0) lds16[dst[u]] = to_bf16(src[u] < 0 ? 0.f : v[u]);
    }
    __syncthreads();
    const pk_u32x4* src = (const pk_u32x4*)lds16;
    for (int i = threadIdx.x; i < R * 64; i += 256) {
      const int rr = i >> 6, cl = i & 63;
      *(pk_u32x4*)(out + ((long)(q * R + rr) * j.ldw + c0 + cl) * 8) = src[i];
    }
  } else {
    const int r0 = (tb / nct) * 16;
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
      const int rr = i >> 6, cl = i & 63;
      if (r0 + rr >= j.krows) continue;
      int src[8];
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) src[e] = pack_bf16_idx(6, r0 + rr, c0 + cl, e, j.Cout, j.Ctot, KK, m, j.lstm_hid);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = j.W[src[e] < 0 ? 0 : src[e]];
      unsigned v[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const unsigned lo = to_bf16(src[2 * e2] < 0 ? 0.f : f[2 * e2]), hi = to_bf16(src[2 * e2 + 1] < 0 ? 0.f : f[2 * e2 + 1]);
        v[e2] = lo | (hi << 16);
      }
      const pk_u32x4 cell = {v[0], v[1], v[2], v[3]};
      *(pk_u32x4*)(out + ((long)(r0 + rr) * j.ldw + c0 + cl) * 8) = cell;
    }
  }
}

// ---- Winograd F(2x2, 3x3) copies (conv_wino.hip): U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], stored as
//   out[((ot * nq + q) * 16 + xi) * 256 + cl * 32 + col]   ot = output-channel tile of 32, q = chunk of 8 reduction channels,
//   xi = 4 i + j the position in the 4 x 4 Winograd domain -- one (ot, q) block of 4096 floats is what a block of the conv kernel
//   DMAs per chunk.  mode 7 (forward): output channel = the weight's row co, reduction channel = its input channel ci, g = W[co][ci];
//   mode 8 (data gradient): output channel = ci, reduction channel = co, g = W[co][ci] rotated by 180 degrees (the transposed conv);
//   here the output channels may be the concat of up to 3 channel segments and the rows gate-interleaved (the ConvLSTM gate convs).
//   One tile = one (ot, q) block, one thread = one channel pair (9 loads, 16 coalesced stores).
__device__ __forceinline__ void pack_tile_wino(int mode, const float* __restrict__ W, float* __restrict__ out, int Cout, int Ctot, int tb,
                                               const SegMap& m, int hid) {
  // channel maps as in the other layouts: the conv's input channels are the CONCAT of the segments (index cg -> weight input channel
  // seg_channel), its rows may be gate-interleaved ConvLSTM rows (packed row co_p -> weight row ref_row): the gate convs' data gradient
  int csum = 0;
#pragma unroll
  for (int s_ = 0; s_ < 3; ++s_) csum += s_ < m.n ? m.C[s_] : 0;
  const int n_red = mode == 7 ? csum : Cout, nq = n_red >> 3;
  const int ot = tb / nq, q = tb - ot * nq;
  const int cl = threadIdx.x >> 5, col = threadIdx.x & 31;
  const int o = ot * 32 + col, rch = q * 8 + cl;                 // output / reduction channel of the conv this copy serves
  const int cop = mode == 7 ? o : rch, cg = mode == 7 ? rch : o; // packed row / concat index
  const int ci = seg_channel(m, cg);
  const bool ok = ci >= 0 && cop < Cout;
  const float* g_ = W + ((size_t)ref_row(ok ? cop : 0, hid) * Ctot + (ok ? ci : 0)) * 9;
  float g[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) g[k] = g_[mode == 7 ? k : 8 - k];
#pragma unroll
  for (int k = 0; k < 9; ++k) g[k] = ok ? g[k] : 0.f;
  float t[4][3];
#pragma unroll
  for (int s_ = 0; s_ < 3; ++s_) {
    t[0][s_] = g[s_];
    t[1][s_] = 0.5f * (g[s_] + g[3 + s_] + g[6 + s_]);
    t[2][s_] = 0.5f * (g[s_] - g[3 + s_] + g[6 + s_]);
    t[3][s_] = g[6 + s_];
  }
  float* o_ = out + ((size_t)tb * 16) * 256 + cl * 32 + col;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o_[(4 * i + 0) * 256] = t[i][0];
    o_[(4 * i + 1) * 256] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
    o_[(4 * i + 2) * 256] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
    o_[(4 * i + 3) * 256] = t[i][2];
  }
}
__global__ __launch_bounds__(256) void pack_wino_kernel(int mode, const float* __restrict__ W, float* __restrict__ out, int Cout, int Ctot, SegMap m,
                                                        int hid) {
  pack_tile_wino(mode, W, out, Cout, Ctot, blockIdx.x, m, hid);
}

// ---- batched repack: every packed copy of every conv weight in ONE launch (after an optimizer step ~240 tiny pack launches
// per training step otherwise).  jobs[] lives in device memory; job i owns the blocks [block_begin_i, block_begin_{i+1}), one block
// per PACK_T x PACK_T tile of the packed matrix.  The forward layouts (columns = output channel) are transposes of the reference
// weight: their tiles are gathered with the lanes running down the packed ROWS (= along the contiguous input-channel / tap axis
// of the reference weight), parked in LDS and written out row-major, so both the global reads and the global writes coalesce. ----
#define PACK_T 64        // tile rows
#define PACK_TC 64       // tile columns (divides RSIS_LDW_ALIGN).  Measured: 128 columns with float2 stores is 8 % slower
template <int KKC>
__device__ __forceinline__ void pack_tile(const rsis_pack_job& j, int tb, float (*tile)[PACK_TC + 1]) {
  SegMap m;
  m.n = j.nseg;
#pragma unroll
  for (int s = 0; s < 3; ++s) { m.C[s] = j.Cseg[s]; m.off[s] = j.Coff[s]; }
  const int nct = j.ldw / PACK_TC;
  const int l = threadIdx.x & 63, q = threadIdx.x >> 6, KK = j.ks * j.ks;
  const int c0 = (tb % nct) * PACK_TC;
  int r0, nrow;
  if (j.imode >= 3) {
    // direct dgrad layouts: tile = one 72-row chunk (8 dy channels x 9 taps) x 128 columns (input channels).  For one dy channel the
    // reference weight is contiguous over (input channel, tap): the lanes walk that run, LDS re-orders it into packed rows.
    constexpr int R = RSIS_CK * 9;
    r0 = (tb / nct) * R;
    nrow = R;
    constexpr int PER = (PACK_TC * 9 + 255) / 256;                     // elements per thread and dy channel (3)
    for (int c = 0; c < RSIS_CK; c += 2) {                             // two dy channels = 2 * PER loads in flight per thread
      int src[2 * PER], kin_[2 * PER], cl_[2 * PER];
#pragma unroll
      for (int u = 0; u < 2 * PER; ++u) {
        const int cu = c + u / PER, i = threadIdx.x + 256 * (u % PER);
        const int cl = i / 9, tap = i - cl * 9;                      // tap of the reference weight
        const int rs = j.imode == 3 ? 8 - tap : tap;                 // its packed position
        const int kin = ((cu >> 1) * 9 + rs) * 2 + (cu & 1);
        const bool ok = i < PACK_TC * 9;
        kin_[u] = ok ? kin : -1; cl_[u] = cl;
        src[u] = ok ? pack_elem_idx<KKC>(j.imode, r0 + kin, c0 + cl, j.Cout, j.Ctot, KK, m, j.lstm_hid) : -1;
      }
      float v[2 * PER];
#pragma unroll
      for (int u = 0; u < 2 * PER; ++u) v[u] = j.W[src[u] < 0 ? 0 : src[u]];
#pragma unroll
      for (int u = 0; u < 2 * PER; ++u)
        if (kin_[u] >= 0) tile[kin_[u]][cl_[u]] = src[u] < 0 ? 0.f : v[u];
    }
  } else {
    r0 = (tb / nct) * PACK_T;
    nrow = min(PACK_T, j.krows - r0);
    if (j.imode == 1) {          // igemm dgrad: columns already run along the reference weight's input channels
      static_assert(PACK_TC == 64, "one column per lane");
      for (int rr0 = q; rr0 < nrow; rr0 += 16) {            // four rows per thread in flight
        int src[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) src[u] = rr0 + 4 * u < nrow ? pack_elem_idx<KKC>(1, r0 + rr0 + 4 * u, c0 + l, j.Cout, j.Ctot, KK, m, j.lstm_hid) : -1;
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = j.W[src[u] < 0 ? 0 : src[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (rr0 + 4 * u < nrow) tile[rr0 + 4 * u][l] = src[u] < 0 ? 0.f : v[u];
      }
    } else {                     // forward layouts: lanes run down the rows
      const int lr = l < nrow ? l : 0;
      for (int k0 = 0; k0 < PACK_TC / 4; k0 += 4) {        // four columns per thread in flight
        int src[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) src[u] = l < nrow ? pack_elem_idx<KKC>(j.imode, r0 + lr, c0 + (k0 + u) * 4 + q, j.Cout, j.Ctot, KK, m, j.lstm_hid) : -1;
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = j.W[src[u] < 0 ? 0 : src[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (l < nrow) tile[l][(k0 + u) * 4 + q] = src[u] < 0 ? 0.f : v[u];
      }
    }
  }
  __syncthreads();
  for (int rr = q; rr < nrow; rr += 4) {
#pragma unroll
    for (int cc = l; cc < PACK_TC; cc += 64) ((float*)j.out)[(long)(r0 + rr) * j.ldw + c0 + cc] = tile[rr][cc];
  }
}

#define PACK_TPB 1     // consecutive tiles per block (measured: 4 is slower than 1 -- the job lookup is not the bottleneck)
__global__ __launch_bounds__(256) void pack_batch_kernel(const rsis_pack_job* __restrict__ jobs, int njobs, int total_tiles) {
  __shared__ __attribute__((aligned(16))) float tile[RSIS_CK * 9][PACK_TC + 1];     // 72 rows: one channel chunk of the direct-dgrad layouts (>= PACK_T)
  int lo = 0, hi = njobs - 1;
  int b = blockIdx.x * PACK_TPB;
  while (lo < hi) {                       // last job whose block_begin <= b
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_begin <= b) lo = mid; else hi = mid - 1;
  }
  rsis_pack_job j = jobs[lo];
  int next_begin = lo + 1 < njobs ? jobs[lo + 1].block_begin : total_tiles;
  for (int t = 0; t < PACK_TPB && b < total_tiles; ++t, ++b) {
    if (b >= next_begin) {                // tiles of a job are consecutive: at most a step to the next job
      ++lo;
      j = jobs[lo];
      next_begin = lo + 1 < njobs ? jobs[lo + 1].block_begin : total_tiles;
    }
    const int tb = b - j.block_begin;
    if (j.imode >= 7) {
      SegMap wm;
      wm.n = j.nseg;
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) { wm.C[s_] = j.Cseg[s_]; wm.off[s_] = j.Coff[s_]; }
      pack_tile_wino(j.imode, j.W, (float*)j.out, j.Cout, j.Ctot, tb, wm, j.lstm_hid);
    }
    else if (j.imode >= 5) pack_tile_bf16(j, tb, (unsigned short*)&tile[0][0]);
    else if (j.ks == 1) pack_tile<1>(j, tb, tile);
    else if (j.ks == 3) pack_tile<9>(j, tb, tile);
    else pack_tile<0>(j, tb, tile);
    __syncthreads();                      // the LDS tile is reused
  }
}

int rsis_l_pack_batch(const rsis_pack_job* jobs, int njobs, int total_blocks, hipStream_t st) {
  hipLaunchKernelGGL(pack_batch_kernel, dim3((total_blocks + PACK_TPB - 1) / PACK_TPB), dim3(256), 0, st, jobs, njobs, total_blocks);
  return rsis_check_launch();
}
int rsis_l_pack_blocks(int mode, int krows, int ldw, int ks) {
  if (mode >= 7) return krows;                              // Winograd copies: krows carries the number of (ot, q) blocks
  if (mode == 5) return krows / (ks * ks * (bf16_ckb(ks * ks) / 8)) * (ldw / 64);     // one chunk x 64 columns per block
  if (mode == 6) return (krows + 15) / 16 * (ldw / 64);
  return (mode >= 3 ? krows / (RSIS_CK * 9) : (krows + PACK_T - 1) / PACK_T) * (ldw / PACK_TC);
}

static inline int pack_grid(long total) {
  long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  return (int)(g < 1 ? 1 : g);
}

// mode: 0 igemm fwd, 1 igemm dgrad, 2 direct fwd, 3 direct dgrad (stride 1), 4 direct dgrad (stride 2: taps not flipped),
// 5 bf16 fwd, 6 bf16 dgrad (cell layouts of conv_bf16.hip), 7 Winograd fwd, 8 Winograd dgrad (conv_wino.hip)
int rsis_l_pack(int mode, const float* W, void* out, int Cout, int Ctot, int ks, int nseg, const int* Cseg, const int* Coff,
                int ldw, int krows, int hid, hipStream_t st) {
  SegMap m = {};
  m.n = nseg;
  int base = 0;
  for (int s = 0; s < nseg; ++s) { m.C[s] = Cseg[s]; m.off[s] = Coff ? Coff[s] : base; base += Cseg[s]; }
  const long total = (long)krows * ldw;
  const dim3 g(pack_grid(total)), b(256);
  if (mode == 7 || mode == 8) {      // Winograd copies (one source covering every input channel; krows = (ot, q) blocks)
    hipLaunchKernelGGL(pack_wino_kernel, dim3(krows), b, 0, st, mode, W, (float*)out, Cout, Ctot, m, hid);
    return rsis_check_launch();
  }
  if (mode < 0 || mode > 6) return RSIS_ERR_ARG;
  if (mode >= 5) {     // bf16 cell layouts: krows = cell rows, out = bf16
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(pack_grid(total * 8)), b, 0, st, mode, W, (unsigned short*)out, Cout, Ctot, ks * ks, m, ldw,
                       krows, hid);
    return rsis_check_launch();
  }
  hipLaunchKernelGGL(pack_kernel, g, b, 0, st, mode, W, (float*)out, Cout, Ctot, ks * ks, m, ldw, krows, hid);
  return rsis_check_launch();
}
