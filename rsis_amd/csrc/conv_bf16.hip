// bf16-operand / fp32-accumulate convolutions for gfx950 on v_mfma_f32_32x32x16_bf16 (the `-dtype bf16` path; BASELINE.json
// configs[2..4]).  Activations stay fp32 NCHW in HBM -- module boundaries, BatchNorm, the LSTM cell update and every gradient
// buffer are unchanged -- and are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) while they are staged into LDS;
// the weights come from a bf16 packed copy of the fp32 master weights (pack.hip, modes 5-7).
//
// Covers, behind the same C-ABI entry points as the exact-f32 kernels (dtype argument of include/rsis_hip.h):
//   * KS = 3 / stride 1 / pad 1: ConvLSTM gates with the fused cell epilogue (reference src/modules/clstm.py:43-58), skip convs,
//     the hoisted gate term, the 3x3 convs of the ResNet bottlenecks (model.py:43-47,59-63; vision.py:12-19), and the data
//     gradient of all of them (same conv, flipped taps / swapped channel roles: only the packing differs);
//   * KS = 1 / stride 1: the bottleneck 1x1 convs and their data gradients (a plain GEMM over the flattened H*W axis, optionally
//     scattered to every `ostride`-th pixel: the data gradient of the strided downsample convs).
//
// Why a different structure from conv3x3_direct.hip: a bf16 MFMA wants 8 consecutive K values per lane, K = input channels,
// and NCHW has them HW floats apart.  So the transposition happens ONCE per staged element instead of per MFMA operand: a
// thread fetches the 8 channels of one patch pixel (8 coalesced dword loads, or 8 dwordx4 loads for 4 pixels on the 1x1 path),
// packs them to one 16-byte cell and writes it with ONE ds_write_b128 into Xs[c8 block][py][px][8 ch].  The MFMA loop is then
// the same shape as the f32 kernel's: per MFMA one conflict-free `ds_read_b128 v, base offset:imm` per operand, no address
// arithmetic, no masks (halo / channel tail / ragged tiles were zero-filled by the buffer descriptor's range check at load time).
// The weights need no conversion and are copied global -> LDS by the LDS-DMA (`buffer_load_dwordx4 ... lds`).
// With 16x the f32 MFMA rate these layers are HBM-bound (the f32 kernels are MFMA-bound): what matters here is bytes in
// flight per CU and one pass over the input, not MFMA utilisation.
#include "common.h"
#include <stdlib.h>
#ifndef XCD_CHUNKED
#define XCD_CHUNKED 1
#endif

enum { EPI_PLAIN = 0, EPI_LSTM = 1 };
typedef __attribute__((address_space(3))) void* lds_vp_t;
typedef const float __attribute__((address_space(1)))* gcf_t;
typedef float __attribute__((address_space(1)))* gf_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define RSIS_OOB 0x7FFFFFF0u

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const f32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}

// KS: 1 or 3.  BM: output rows per block (32 / 64 / 128 -> 1 / 2 / 4 waves along M).  TW x TH: output pixels per block (KS = 1:
// TH = 1 and TW consecutive pixels of the flattened map).  CKB: input channels per LDS stage.  V4 (KS = 1, H*W % 4 == 0): the
// activation tile is fetched as dwordx4 along the pixels.
template <int KS, int BM, int TW, int TH, int EPI, int CKB, bool V4>
__global__ __launch_bounds__(256) void conv_bf16_kernel(const ConvArgs p) {
#if __HIP_DEVICE_COMPILE__
  constexpr int KK = KS * KS, HALO = KS / 2;
  constexpr int BN = TW * TH;
  constexpr int WGM = BM / 32, WGN = 4 / WGM;
  constexpr int TN = BN / WGN / 32;
  constexpr int NCB = CKB / 8;                       // 8-channel blocks per stage
  constexpr int PW = TW + 2 * HALO, PH = TH + 2 * HALO, IMS = PH * PW;
  constexpr int XC = NCB * IMS;                      // 16-byte cells of the activation stage
  constexpr int WC = KK * NCB * BM;                  // ... of the weight stage
  constexpr int NXT = V4 ? (NCB * (TW / 4) + 255) / 256 : (XC + 255) / 256;   // staging tasks per thread per chunk
  constexpr int NW = (WC + 255) / 256;
  static_assert(TN >= 1 && BN % (WGN * 32) == 0 && WGM * WGN == 4 && CKB % 16 == 0, "tile");
  static_assert(!V4 || (KS == 1 && TW % 4 == 0), "V4 is the 1x1 path");

  __shared__ __attribute__((aligned(16))) u32x4 lds[2 * (XC + WC)];
  u32x4* const Xs0 = lds;
  u32x4* const Ws0 = lds + 2 * XC;

  const gcf_t src0 = (gcf_t)p.src[0], src1 = (gcf_t)p.src[1], src2 = (gcf_t)p.src[2];
  const int C0 = p.C[0], C1 = p.C[1], C2 = p.C[2];
  const int q0 = (C0 + CKB - 1) / CKB, q1 = (C1 + CKB - 1) / CKB, q2 = (C2 + CKB - 1) / CKB;   // chunks per source
  const int nq_all = q0 + q1 + q2;
  const int ksplit = gridDim.y, kz = blockIdx.y;
  const int q_begin = (int)((long)nq_all * kz / ksplit), q_end = (int)((long)nq_all * (kz + 1) / ksplit);
  const int nq = q_end - q_begin;
  // KS = 1 walks the flattened map: one "row" of H*W pixels
  const int H = KS == 1 ? 1 : p.H, W = KS == 1 ? p.H * p.W : p.W, HW = p.H * p.W;
  const int ldw = p.ldw;

  // ---- block -> (co tile, spatial tile); blocks b, b+8, ... share an XCD: a tile's co tiles stay on one L2 ----
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = bid >> 3;
  const int co_t = q % p.n_co_tiles;
  // (XCD x owns the contiguous range [x * chunk, (x + 1) * chunk) of the spatial tiles: blocks b, b + 8, ... -- the ones this XCD
  //  runs one after the other -- are NEIGHBOURING tiles, so the halo rows / columns two tiles share are L2 hits instead of a
  //  second fetch from HBM by another XCD.  Measured against the round-robin map `(q / n_co_tiles) * 8 + xcd` (a build with
  //  -DXCD_CHUNKED=0): the bf16 gate launch of the 112 x 112 level 48.4 -> 33.6 us, the fp32 128 x 128 level 83.7 -> 79.3 us.)
  const int sp_t = XCD_CHUNKED ? xcd * ((p.n_px_tiles + 7) >> 3) + q / p.n_co_tiles : (q / p.n_co_tiles) * 8 + xcd;
  if (sp_t >= p.n_px_tiles) return;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int tx = sp_t % tiles_x;
  const int ty = (sp_t / tiles_x) % tiles_y;
  const int b0 = sp_t / (tiles_x * tiles_y);
  const int x0 = tx * TW, y0 = ty * TH;

  const int tid = threadIdx.x;
  const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  // ---- loop-invariant byte offsets of this thread's staging tasks inside the [CKB][H][W] slab of one chunk of image b0 ----
  unsigned xvo[NXT];
#pragma unroll
  for (int i = 0; i < NXT; ++i) {
    const int e = tid + i * 256;
    if constexpr (V4) {
      const int cb = e / (TW / 4), x4 = e - cb * (TW / 4);
      const int gx = x0 + x4 * 4;
      xvo[i] = (cb < NCB && gx < W) ? (unsigned)(cb * 8 * HW + gx) * 4u : RSIS_OOB;
    } else {
      const int cb = e / IMS, rem = e - cb * IMS;
      const int py = rem / PW, pxx = rem - py * PW;
      const int gy = y0 + py - HALO, gx = x0 + pxx - HALO;
      const bool ok = (e < XC) && ((unsigned)gy < (unsigned)H) && ((unsigned)gx < (unsigned)W);
      xvo[i] = ok ? (unsigned)(cb * 8 * HW + gy * W + gx) * 4u : RSIS_OOB;
    }
  }
  unsigned wvo[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int idx = tid + i * 256;
    wvo[i] = (unsigned)((idx / BM) * ldw + idx % BM) * 16u;
  }
  const unsigned chs = (unsigned)HW * 4u;            // byte stride between channels

  // ---- per-lane LDS read bases (cells; the rest are immediates in the unrolled loop) ----
  int xoff[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int pp = (wn * TN + j) * 32 + l31;
    const int x = pp % TW, y = pp / TW;
    xoff[j] = hi * IMS + y * PW + x;
  }
  const int woff = hi * BM + wm * 32 + l31;

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const char* const wbase = (const char*)p.wp + (size_t)co_t * BM * 16;

  // scalar chunk cursor, positioned on this block's first chunk
  int cs = 0, cq = q_begin;
  if (cs == 0 && cq >= q0 && q0 < nq_all) { cq -= q0; cs = 1; }
  if (cs == 1 && cq >= q1 && q0 + q1 < nq_all) { cq -= q1; cs = 2; }

  float rx[NXT][V4 ? 32 : 8];
  // fetch chunk QG: activations -> registers (8 channels per task), weights -> LDS stage BUF by the LDS-DMA
#define BF_ISSUE(QG, BUF)                                                                                          \
  {                                                                                                                \
    gcf_t src = src0; int Cs = C0;                                                                                 \
    if (cs == 1) { src = src1; Cs = C1; }                                                                          \
    if (cs == 2) { src = src2; Cs = C2; }                                                                          \
    const int c0 = cq * CKB;                                                                                       \
    const int cn = min(CKB, Cs - c0);                                                                              \
    const float* xb = (const float*)src + ((size_t)b0 * Cs + c0) * HW;                                             \
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, cn * HW * 4, 0x00020000);   \
    _Pragma("unroll") for (int i = 0; i < NXT; ++i) {                                                              \
      _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                              \
        if constexpr (V4) {                                                                                        \
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx_, xvo[i] + c * chs, 0, 0)); \
          rx[i][c * 4 + 0] = v[0]; rx[i][c * 4 + 1] = v[1]; rx[i][c * 4 + 2] = v[2]; rx[i][c * 4 + 3] = v[3];     \
        } else {                                                                                                   \
          rx[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx_, xvo[i] + c * chs, 0, 0)); \
        }                                                                                                          \
      }                                                                                                            \
    }                                                                                                              \
    const char* wrow = wbase + (size_t)(QG) * (KK * NCB) * ldw * 16;                                               \
    const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc((void*)wrow, 0, KK * NCB * ldw * 16, 0x00020000); \
    u32x4* Ws = Ws0 + (BUF) * WC + wave * 64;                                                                      \
    _Pragma("unroll") for (int i = 0; i < NW; ++i)                                                                 \
      if (WC % 256 == 0 || i * 256 + wave * 64 < WC)                                                               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_, (lds_vp_t)(Ws + i * 256), 16, wvo[i], 0, 0, 0);              \
    if (++cq == (cs == 0 ? q0 : (cs == 1 ? q1 : q2))) { cq = 0; ++cs; if (cs == 1 && q1 == 0) ++cs; }             \
  }
  // registers -> bf16 cells of LDS stage BUF
#define BF_STORE(BUF)                                                                                              \
  {                                                                                                                \
    u32x4* Xs = Xs0 + (BUF) * XC;                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NXT; ++i) {                                                              \
      const int e = tid + i * 256;                                                                                 \
      if constexpr (V4) {                                                                                          \
        const int cb = e / (TW / 4), x4 = e - cb * (TW / 4);                                                       \
        if (NCB * (TW / 4) % 256 == 0 || cb < NCB) {                                                               \
          _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                          \
            u32x4 cell;                                                                                            \
            _Pragma("unroll") for (int c2 = 0; c2 < 4; ++c2) cell[c2] = pack_bf16x2(rx[i][(2 * c2) * 4 + k], rx[i][(2 * c2 + 1) * 4 + k]); \
            Xs[cb * TW + x4 * 4 + k] = cell;                                                                       \
          }                                                                                                        \
        }                                                                                                          \
      } else {                                                                                                     \
        if (XC % 256 == 0 || e < XC) {                                                                             \
          u32x4 cell;                                                                                              \
          _Pragma("unroll") for (int c2 = 0; c2 < 4; ++c2) cell[c2] = pack_bf16x2(rx[i][2 * c2], rx[i][2 * c2 + 1]); \
          Xs[e] = cell;                                                                                            \
        }                                                                                                          \
      }                                                                                                            \
    }                                                                                                              \
  }
#define BF_LAND() __builtin_amdgcn_s_waitcnt(0x0F70);   /* vmcnt(0): loads returned, this wave's DMA has landed in LDS */

  if (nq > 0) {
    BF_ISSUE(q_begin, 0)
    BF_LAND()
    BF_STORE(0)
  }
  __syncthreads();
  for (int t = 0; t < nq; ++t) {
    const int cur = t & 1;
    const bool more = t + 1 < nq;
    if (more) BF_ISSUE(q_begin + t + 1, cur ^ 1)   // (weight stage cur^1 was last read before the barrier that ended step t-1)
    {
      const u32x4* Xs = Xs0 + cur * XC;
      const u32x4* Ws = Ws0 + cur * WC + woff;
#pragma unroll
      for (int kk = 0; kk < NCB / 2; ++kk)
#pragma unroll
        for (int r = 0; r < KS; ++r)
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, Ws[((r * KS + s) * NCB + 2 * kk) * BM]);
            bf16x8 b[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = __builtin_bit_cast(bf16x8, Xs[xoff[j] + 2 * kk * IMS + r * PW + s]);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[j], acc[j], 0, 0, 0);
          }
    }
    BF_LAND()
    if (more) BF_STORE(cur ^ 1)
    __syncthreads();
  }
#undef BF_ISSUE
#undef BF_STORE
#undef BF_LAND

  // ---- epilogue (fp32; same accumulator layout as the f32 MFMA kernels: column = pixel l31, rows (r&3) + 8 (r>>2) + 4 hi) ----
  const int co_base = co_t * BM + wm * 32;
  const gcf_t bias = (gcf_t)p.bias, addend = (gcf_t)p.addend;
  unsigned long long best[4] = {0ull, 0ull, 0ull, 0ull};      // EPI_LSTM + side_key: this lane's best (h, pixel) per hidden channel
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int pp = (wn * TN + j) * 32 + l31;
    const int ox = x0 + pp % TW, oy = y0 + pp / TW;
    if (oy >= H || ox >= W) continue;
    int osp = oy * W + ox;
    if (EPI == EPI_PLAIN) {
      const gf_t d0 = (gf_t)p.dst[0], d1 = (gf_t)p.dst[1], d2 = (gf_t)p.dst[2];
      const int Cd0 = p.Cd[0], Cd1 = p.Cd[1], Cd2 = p.Cd[2], Cout = p.Cout;
      const int e1 = Cd0, e2 = Cd0 + Cd1;
      int oHW = HW;
      if (KS == 1 && p.ostride > 1) {      // strided 1x1 data gradient: the GEMM walks the dy grid, rows go to every ostride-th pixel
        const int oho = osp / p.W;
        osp = (oho * p.ostride) * p.oW + (osp - oho * p.W) * p.ostride;
        oHW = p.oH * p.oW;
      }
      if (p.ndst == 1 && ksplit == 1) {
        // single destination, no split-K (every trunk / skip / hoisted conv and data gradient): the rows of the tile are oHW floats
        // apart inside image b0's [Cout][oHW] slab, so a row is ONE buffer store whose offset is a per-lane base plus a scalar
        // row term, and rows >= Cout fall outside the descriptor (dropped) -- ~2 instructions per stored value instead of ~25 of
        // 64-bit index arithmetic and destination selection.  (The general path below costs 8 of a 26 us launch.)
        const size_t slab = (size_t)b0 * Cout * oHW;
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dst[0] + slab), 0, Cout * oHW * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(addend ? p.addend + slab : p.dst[0]), 0, Cout * oHW * 4, 0x00020000);
        const unsigned vo = (unsigned)((co_base + 4 * hi) * oHW + osp) * 4u;
        const unsigned rowb = (unsigned)oHW * 4u;
        float av[16];
        if (addend) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            av[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, vo + (unsigned)((r & 3) + 8 * (r >> 2)) * rowb, 0, 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[j][r];
          if (bias) { const int co = co_base + (r & 3) + 8 * (r >> 2) + 4 * hi; v += co < Cout ? bias[co] : 0.f; }
          if (addend) v += av[r];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, vo + (unsigned)((r & 3) + 8 * (r >> 2)) * rowb, 0, 0);
        }
        continue;
      }
      float av[16];
      if (addend) {                        // all addend loads of the tile first, then the stores (no load -> add -> store chains)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co_base + (r & 3) + 8 * (r >> 2) + 4 * hi;
          av[r] = (co < Cout && kz == 0) ? addend[((size_t)b0 * Cd0 + co) * oHW + osp] : 0.f;     // (addend: single destination only)
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_base + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (co >= Cout) continue;
        float v = acc[j][r];
        if (bias && kz == 0) v += bias[co];
        if (addend) v += av[r];
        gf_t d = d0;
        int cl = co, Cd = Cd0;
        if (co >= e1) { d = d1; cl = co - e1; Cd = Cd1; }
        if (co >= e2) { d = d2; cl = co - e2; Cd = Cd2; }
        const size_t idx = ((size_t)b0 * Cd + cl) * oHW + osp;
        if (ksplit > 1) atomicAdd((float*)(d + idx), v);
        else d[idx] = v;
      }
    } else {
      const int hid = p.hid;
      const gcf_t c_prev = (gcf_t)p.c_prev;
      const gf_t c_out = (gf_t)p.c_out, h_out = (gf_t)p.h_out, act_out = (gf_t)p.act_out;
      if ((size_t)p.B * 4 * hid * HW * 4 < (1ull << 31)) {
        // the cell update through buffer descriptors: per hidden channel 4 addend loads, c_prev, and the c / h / 4 gate stores are
        // per-lane bases + scalar multiples of HW (no 64-bit index arithmetic per access); channels >= hid get an out-of-range
        // offset (loads return 0, stores are dropped)
        const unsigned gspan = (unsigned)((size_t)p.B * 4 * hid * HW * 4), sspan = gspan / 4;
        const __amdgpu_buffer_rsrc_t r_add = __builtin_amdgcn_make_buffer_rsrc((void*)(addend ? p.addend : p.h_out), 0, addend ? gspan : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_act = __builtin_amdgcn_make_buffer_rsrc((void*)(act_out ? p.act_out : p.h_out), 0, act_out ? gspan : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_cp = __builtin_amdgcn_make_buffer_rsrc((void*)(c_prev ? p.c_prev : p.h_out), 0, c_prev ? sspan : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_c = __builtin_amdgcn_make_buffer_rsrc((void*)p.c_out, 0, sspan, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_h = __builtin_amdgcn_make_buffer_rsrc((void*)p.h_out, 0, sspan, 0x00020000);
        const unsigned rowb = (unsigned)HW * 4u;
        const int jh0 = (co_base >> 2) + hi;
        const unsigned vg = (unsigned)((b0 * 4 * hid + 4 * jh0) * HW + osp) * 4u;
        const unsigned vs = (unsigned)((b0 * hid + jh0) * HW + osp) * 4u;
        float ga[4][4], cpv[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {          // all loads of the tile first
          const bool ok = jh0 + 2 * r4 < hid;
#pragma unroll
          for (int g = 0; g < 4; ++g)
            ga[r4][g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_add, ok ? vg + (8 * r4 + g) * rowb : 0x7FFFFFF0u, 0, 0));
          cpv[r4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_cp, ok ? vs + 2 * r4 * rowb : 0x7FFFFFF0u, 0, 0));
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const bool ok = jh0 + 2 * r4 < hid;
          const int cop = 4 * (jh0 + 2 * r4);
          float ai = acc[j][4 * r4 + 0] + ga[r4][0], af = acc[j][4 * r4 + 1] + ga[r4][1];
          float ao = acc[j][4 * r4 + 2] + ga[r4][2], ag = acc[j][4 * r4 + 3] + ga[r4][3];
          if (bias && ok) { ai += bias[cop]; af += bias[cop + 1]; ao += bias[cop + 2]; ag += bias[cop + 3]; }
          const float gi = rsis_sigmoid(ai), gf = rsis_sigmoid(af), go = rsis_sigmoid(ao), gg = tanhf(ag);
          const float c = gf * cpv[r4] + gi * gg;  // clstm.py:57
          const float h = go * tanhf(c);           // clstm.py:58
          if (p.side_key && ok) { const unsigned long long k = rsis_side_key(h, osp); best[r4] = k > best[r4] ? k : best[r4]; }
          const unsigned os = ok ? vs + 2 * r4 * rowb : 0x7FFFFFF0u, og = ok ? vg + 8 * r4 * rowb : 0x7FFFFFF0u;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c), r_c, os, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h), r_h, os, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gi), r_act, og, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gf), r_act, og + rowb, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, go), r_act, og + 2 * rowb, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gg), r_act, og + 3 * rowb, 0, 0);
        }
        continue;
      }
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int jh = (co_base >> 2) + 2 * r4 + hi;   // hidden channel (rows are gate-interleaved: 4 * jh + gate)
        if (jh >= hid) continue;
        const int cop = jh * 4;
        float ai = acc[j][4 * r4 + 0], af = acc[j][4 * r4 + 1], ao = acc[j][4 * r4 + 2], ag = acc[j][4 * r4 + 3];
        if (bias) { ai += bias[cop]; af += bias[cop + 1]; ao += bias[cop + 2]; ag += bias[cop + 3]; }
        const size_t gidx = ((size_t)b0 * 4 * hid + cop) * HW + osp;
        if (addend) {
          ai += addend[gidx]; af += addend[gidx + HW];
          ao += addend[gidx + 2 * (size_t)HW]; ag += addend[gidx + 3 * (size_t)HW];
        }
        const float gi = rsis_sigmoid(ai), gf = rsis_sigmoid(af), go = rsis_sigmoid(ao), gg = tanhf(ag);
        const size_t sidx = ((size_t)b0 * hid + jh) * HW + osp;
        const float cp = c_prev ? c_prev[sidx] : 0.f;
        const float c = gf * cp + gi * gg;       // clstm.py:57
        const float h = go * tanhf(c);           // clstm.py:58
        if (p.side_key) { const unsigned long long k = rsis_side_key(h, osp); best[r4] = k > best[r4] ? k : best[r4]; }
        c_out[sidx] = c;
        h_out[sidx] = h;
        if (act_out) {
          act_out[gidx] = gi; act_out[gidx + HW] = gf;
          act_out[gidx + 2 * (size_t)HW] = go; act_out[gidx + 3 * (size_t)HW] = gg;
        }
      }
    }
  }
  if constexpr (EPI == EPI_LSTM) {
    // the side feature of model.py:143 (as in conv3x3_direct.hip): one 64-bit atomic max per hidden channel and half wave
    if (p.side_key) {
      unsigned long long kk[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) kk[r4] = rsis_key_max32(best[r4]);
      rsis_side_key_max4(p.side_key + (size_t)b0 * p.hid, (co_base >> 2) + hi, p.hid, kk, l31 == 0, tiles_x * tiles_y >= RSIS_SIDE_CHECK_TILES);
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// 1x1 / stride 1, single source, H*W % 4 == 0: the same GEMM with a DEEP load pipeline.  The register-staged kernel above keeps
// ONE chunk of loads in flight per block, and with a bf16 MFMA loop of a few hundred cycles per chunk every chunk exposes a full
// memory round trip: 0.1-0.17 of the HBM roofline on the trunk's 1x1 layers.  Here the fp32 activations are copied global -> LDS
// RAW by the LDS-DMA into a ring of NR stages (no registers: NR - 1 chunks, 50-100 KB per CU, stay in flight, counted `vmcnt`
// waits, raw `s_barrier`), a conversion pass turns the landed stage into bf16 cells (8 ds_read_b32 + 4 v_cvt_pk + 1 ds_write_b128
// per cell, conflict-free both ways), and the MFMA loop reads cells exactly as before.  The weights ride the same ring.
// ------------------------------------------------------------------------------------------------
#define RSIS_VMCNT(N) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 15) | (((N) >> 4) << 14))

template <int BM, int BN, int CKB, int NR>
__global__ __launch_bounds__(256) void conv1x1_bf16_ring_kernel(const ConvArgs p) {
#if __HIP_DEVICE_COMPILE__
  constexpr int WGM = BM / 32, WGN = 4 / WGM;
  constexpr int TN = BN / WGN / 32;
  constexpr int NCB = CKB / 8;
  constexpr int RAW_F = CKB * BN;                    // floats per raw stage
  constexpr int WC = NCB * BM, XC = NCB * BN;        // cells per weight / activation stage
  constexpr int NXD = RAW_F / 4 / 256;               // dwordx4 DMA per thread per chunk (activations)
  constexpr int NWD = (WC + 255) / 256;              // ... (weights; stage padded to whole 256-lane rows)
  constexpr int WCP = NWD * 256;
  constexpr int NCV = XC / 256;                      // cells converted per thread per chunk
  static_assert(TN >= 1 && WGM * WGN == 4 && (RAW_F / 4) % 256 == 0 && XC % 256 == 0 && CKB % 16 == 0 && NR >= 3, "tile");
  constexpr int C_DMA = NXD + NWD;                   // DMA instructions per wave per chunk (vmcnt bookkeeping)
  static_assert((NR - 2) * C_DMA < 64, "vmcnt is 6 bits");

  // ONE shared array (a second __shared__ object makes hipcc drain the LDS-DMA queue before every ds_read)
  __shared__ __attribute__((aligned(16))) float lds[NR * RAW_F + (NR * WCP + 2 * XC) * 4];
  float* const raw0 = lds;
  u32x4* const ws0 = reinterpret_cast<u32x4*>(lds + NR * RAW_F);
  u32x4* const xb0 = ws0 + NR * WCP;

  const int HW = p.H * p.W, C = p.C[0];
  const int nq = (C + CKB - 1) / CKB;
  const int ldw = p.ldw;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = bid >> 3;
  const int co_t = q % p.n_co_tiles;
  // (XCD x owns the contiguous range [x * chunk, (x + 1) * chunk) of the spatial tiles: blocks b, b + 8, ... -- the ones this XCD
  //  runs one after the other -- are NEIGHBOURING tiles, so the halo rows / columns two tiles share are L2 hits instead of a
  //  second fetch from HBM by another XCD.  Measured against the round-robin map `(q / n_co_tiles) * 8 + xcd` (a build with
  //  -DXCD_CHUNKED=0): the bf16 gate launch of the 112 x 112 level 48.4 -> 33.6 us, the fp32 128 x 128 level 83.7 -> 79.3 us.)
  const int sp_t = XCD_CHUNKED ? xcd * ((p.n_px_tiles + 7) >> 3) + q / p.n_co_tiles : (q / p.n_co_tiles) * 8 + xcd;
  if (sp_t >= p.n_px_tiles) return;
  const int tiles_x = (HW + BN - 1) / BN;
  const int tx = sp_t % tiles_x, b0 = sp_t / tiles_x;
  const int x0 = tx * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  // loop-invariant DMA offsets
  unsigned xvo[NXD], wvo[NWD];
#pragma unroll
  for (int i = 0; i < NXD; ++i) {
    const int e = tid + i * 256;
    const int ch = e / (BN / 4), c4 = e - ch * (BN / 4);
    const int gx = x0 + c4 * 4;
    xvo[i] = gx < HW ? (unsigned)(ch * HW + gx) * 4u : RSIS_OOB;
  }
#pragma unroll
  for (int i = 0; i < NWD; ++i) {
    const int idx = tid + i * 256;
    wvo[i] = idx < WC ? (unsigned)((idx / BM) * ldw + idx % BM) * 16u : RSIS_OOB;
  }
  const float* const xbase = p.src[0] + (size_t)b0 * C * HW;
  const char* const wbase = (const char*)p.wp + (size_t)co_t * BM * 16;

#define RING_ISSUE(QG)                                                                                            \
  {                                                                                                               \
    const int slot = (QG) % NR;                                                                                   \
    const int c0 = (QG) * CKB;                                                                                    \
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)(xbase + (size_t)c0 * HW), 0, min(CKB, C - c0) * HW * 4, 0x00020000); \
    float* dst = raw0 + slot * RAW_F + wave * 256;                                                                \
    _Pragma("unroll") for (int i = 0; i < NXD; ++i)                                                               \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_, (lds_vp_t)(dst + i * 1024), 16, xvo[i], 0, 0, 0);              \
    const char* wrow = wbase + (size_t)(QG) * NCB * ldw * 16;                                                     \
    const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc((void*)wrow, 0, NCB * ldw * 16, 0x00020000); \
    u32x4* wd = ws0 + slot * WCP + wave * 64;                                                                     \
    _Pragma("unroll") for (int i = 0; i < NWD; ++i)                                                               \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_, (lds_vp_t)(wd + i * 256), 16, wvo[i], 0, 0, 0);                \
  }

  int xoff[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) xoff[j] = hi * BN + (wn * TN + j) * 32 + l31;
  const int woff = hi * BM + wm * 32 + l31;

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // prologue: NR - 1 chunks in flight
#pragma unroll
  for (int i = 0; i < NR - 1; ++i)
    if (i < nq) RING_ISSUE(i)

  for (int t = 0; t < nq; ++t) {
    // chunk t has landed once at most `ahead` younger chunks of this wave's DMA are still outstanding
    const int ahead = min(NR - 2, nq - 1 - t);
    if (ahead >= 2) { RSIS_VMCNT(2 * C_DMA); }
    else if (ahead == 1) { RSIS_VMCNT(C_DMA); }
    else { RSIS_VMCNT(0); }
    __builtin_amdgcn_s_barrier();                     // every wave's share of chunk t is in LDS; slot (t-1) % NR is free
    if (t + NR - 1 < nq) RING_ISSUE(t + NR - 1)
    {                                                 // raw fp32 [ch][px] -> bf16 cells [c8][px]
      const float* raw = raw0 + (t % NR) * RAW_F;
      u32x4* xb = xb0 + (t & 1) * XC;
#pragma unroll
      for (int j = 0; j < NCV; ++j) {
        const int c = tid + j * 256;
        const int cb = c / BN, px = c - cb * BN;
        const float* r8 = raw + cb * 8 * BN + px;
        u32x4 cell;
#pragma unroll
        for (int k = 0; k < 4; ++k) cell[k] = pack_bf16x2(r8[(2 * k) * BN], r8[(2 * k + 1) * BN]);
        xb[c] = cell;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): my cells are written
    __builtin_amdgcn_s_barrier();
    {
      const u32x4* Xs = xb0 + (t & 1) * XC;
      const u32x4* Ws = ws0 + (t % NR) * WCP + woff;
#pragma unroll
      for (int kk = 0; kk < NCB / 2; ++kk) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, Ws[2 * kk * BM]);
        bf16x8 b[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = __builtin_bit_cast(bf16x8, Xs[xoff[j] + 2 * kk * BN]);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[j], acc[j], 0, 0, 0);
      }
    }
  }
#undef RING_ISSUE

  // ---- epilogue (as conv_bf16_kernel, EPI_PLAIN, KS = 1) ----
  const int co_base = co_t * BM + wm * 32;
  const gcf_t bias = (gcf_t)p.bias, addend = (gcf_t)p.addend;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int ox = x0 + (wn * TN + j) * 32 + l31;
    if (ox >= HW) continue;
    int osp = ox, oHW = HW;
    if (p.ostride > 1) {
      const int oho = osp / p.W;
      osp = (oho * p.ostride) * p.oW + (osp - oho * p.W) * p.ostride;
      oHW = p.oH * p.oW;
    }
    const gf_t d0 = (gf_t)p.dst[0], d1 = (gf_t)p.dst[1], d2 = (gf_t)p.dst[2];
    const int Cd0 = p.Cd[0], Cd1 = p.Cd[1], Cd2 = p.Cd[2], Cout = p.Cout;
    const int e1 = Cd0, e2 = Cd0 + Cd1;
    if (p.ndst == 1) {      // (the fast single-destination epilogue of conv_bf16_kernel)
      const size_t slab = (size_t)b0 * Cout * oHW;
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dst[0] + slab), 0, Cout * oHW * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(addend ? p.addend + slab : p.dst[0]), 0, Cout * oHW * 4, 0x00020000);
      const unsigned vo = (unsigned)((co_base + 4 * hi) * oHW + osp) * 4u;
      const unsigned rowb = (unsigned)oHW * 4u;
      float avf[16];
      if (addend) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          avf[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, vo + (unsigned)((r & 3) + 8 * (r >> 2)) * rowb, 0, 0));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[j][r];
        if (bias) { const int co = co_base + (r & 3) + 8 * (r >> 2) + 4 * hi; v += co < Cout ? bias[co] : 0.f; }
        if (addend) v += avf[r];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, vo + (unsigned)((r & 3) + 8 * (r >> 2)) * rowb, 0, 0);
      }
      continue;
    }
    float av[16];
    if (addend) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_base + (r & 3) + 8 * (r >> 2) + 4 * hi;
        av[r] = co < Cout ? addend[((size_t)b0 * Cd0 + co) * oHW + osp] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co_base + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (co >= Cout) continue;
      float v = acc[j][r];
      if (bias) v += bias[co];
      if (addend) v += av[r];
      gf_t d = d0;
      int cl = co, Cd = Cd0;
      if (co >= e1) { d = d1; cl = co - e1; Cd = Cd1; }
      if (co >= e2) { d = d2; cl = co - e2; Cd = Cd2; }
      d[((size_t)b0 * Cd + cl) * oHW + osp] = v;
    }
  }
#endif
}

template <int BM, int BN>
static int launch_ring(ConvArgs& a, hipStream_t st) {
  a.n_co_tiles = rsis_cdiv(a.Cout, BM);
  a.n_px_tiles = rsis_cdiv(a.H * a.W, BN) * a.B;
  const int grid = a.n_co_tiles * 8 * rsis_cdiv(a.n_px_tiles, 8);
  hipLaunchKernelGGL((conv1x1_bf16_ring_kernel<BM, BN, 32, 4>), dim3(grid), dim3(256), 0, st, a);
  return rsis_check_launch();
}

// ------------------------------------------------------------------------------------------------
template <int KS, int BM, int TW, int TH, int EPI, int CKB>
static int launch_bf16_cfg(ConvArgs& a, hipStream_t st) {
  a.n_co_tiles = rsis_cdiv(a.Cout, BM);
  const int gw = KS == 1 ? a.H * a.W : a.W, gh = KS == 1 ? 1 : a.H;
  a.n_px_tiles = rsis_cdiv(gw, TW) * rsis_cdiv(gh, TH) * a.B;
  const int grid = a.n_co_tiles * 8 * rsis_cdiv(a.n_px_tiles, 8);
  int ksplit = 1;
  if (EPI == EPI_PLAIN && a.ksplit == 0) {    // ksplit == 0: the caller zeroed the output and allows split-K
    int nq = 0;
    for (int s = 0; s < a.nsrc; ++s) nq += (a.C[s] + CKB - 1) / CKB;
    const int blocks = a.n_co_tiles * a.n_px_tiles;
    if (blocks < 160 && nq >= 16) {
      ksplit = rsis_cdiv(512, blocks);
      if (ksplit > nq / 4) ksplit = nq / 4;
      if (ksplit > 16) ksplit = 16;
      if (ksplit < 1) ksplit = 1;
    }
  }
  if constexpr (KS == 1) {
    if ((a.H * a.W) % 4 == 0) {
      hipLaunchKernelGGL((conv_bf16_kernel<KS, BM, TW, TH, EPI, CKB, true>), dim3(grid, ksplit), dim3(256), 0, st, a);
      return rsis_check_launch();
    }
  }
  hipLaunchKernelGGL((conv_bf16_kernel<KS, BM, TW, TH, EPI, CKB, false>), dim3(grid, ksplit), dim3(256), 0, st, a);
  return rsis_check_launch();
}

// 3x3 variants: 1 = BM64 8x8 (64 px), 2 = BM64 16x8, 3 = BM64 32x8, 4 = BM32 16x8, 5 = BM32 32x8, 6 = BM128 8x8
template <int EPI>
static int launch_bf16_k3(ConvArgs& a, hipStream_t st, int force) {
  int v = force;
  if (v <= 0) {
    // measured on the trunk / decoder shapes at batch 32, 224^2 and 256^2 (tools/exp/bf16_shape_sweep.py): these launches are
    // latency-bound, so the rule is "enough blocks for two per CU", then the widest tile
    const bool small_co = a.Cout <= 32;
    if (a.W <= 8 && a.H <= 8) v = (a.Cout >= 256 && (long)rsis_cdiv(a.Cout, 128) * a.B >= 256) ? 6 : 1;
    else if (a.W <= 16) {
      v = small_co ? 4 : 2;
      const long blocks = (long)rsis_cdiv(a.Cout, 64) * rsis_cdiv(a.W, 16) * rsis_cdiv(a.H, 8) * a.B;
      if (v == 2 && blocks < 512) v = 1;       // 256 -> 256 @14^2 / 16^2: 29.5 -> 26 us
    } else v = a.Cout <= 64 ? 5 : 3;           // <= 64 rows: 32-row tiles (64 -> 64 @56^2: 36 -> 26 us; 16 -> 64 @112^2: 72 -> 52 us)
    if (v == 3) {      // keep >= ~2 blocks per CU
      const long blocks = (long)rsis_cdiv(a.Cout, 64) * rsis_cdiv(a.W, 32) * rsis_cdiv(a.H, 8) * a.B;
      if (blocks < 512) v = 2;
    }
  }
  switch (v) {
    case 1: return launch_bf16_cfg<3, 64, 8, 8, EPI, RSIS_CKB3>(a, st);
    case 2: return launch_bf16_cfg<3, 64, 16, 8, EPI, RSIS_CKB3>(a, st);
    case 3: return launch_bf16_cfg<3, 64, 32, 8, EPI, RSIS_CKB3>(a, st);
    case 4: return launch_bf16_cfg<3, 32, 16, 8, EPI, RSIS_CKB3>(a, st);
    case 5: return launch_bf16_cfg<3, 32, 32, 8, EPI, RSIS_CKB3>(a, st);
    case 6: return launch_bf16_cfg<3, 128, 8, 8, EPI, RSIS_CKB3>(a, st);
    default: return RSIS_ERR_ARG;
  }
}

// 1x1 variants: 1 = BM128 x 128 px, 2 = BM64 x 128 px, 3 = BM32 x 256 px (register staged); 4-7 = LDS-DMA ring kernel (BM128x128,
// BM128x64, BM64x128, BM64x64): the default wherever it applies (H*W % 4 == 0, one source, 32-bit offsets)
static int launch_bf16_k1(ConvArgs& a, hipStream_t st, int force) {
  int v = force;
  static const bool ring_ok = !(getenv("RSIS_BF16_RING") && getenv("RSIS_BF16_RING")[0] == '0');
  const bool ring_geom = (a.H * a.W) % 4 == 0 && a.nsrc == 1 && a.Cout > 32 && (long)a.C[0] * a.H * a.W * 4 < (1L << 31);
  // measured (B = 32 trunk shapes): the ring wins where K is deep and the output small (1024 -> 256 @16^2: 27 -> 22 us, 512 -> 128
  // @32^2: 25.5 -> 21.4 us); with many output rows or a short K loop its one-or-two-blocks-per-CU footprint loses to the
  // register-staged kernel's 2-3 co-resident blocks (256 -> 1024 @16^2: 27 vs 41 us).  Both sit near 25 GB/s of LDS fill per CU:
  // the weight tile every block re-reads weighs as much as the fp32 activations on these small maps.
  if (v <= 0 && ring_ok && ring_geom && a.C[0] >= 512 && 2 * a.Cout <= a.C[0]) {
    // enough blocks to cover the chip twice with the widest tile that allows it
    const long px = (long)a.H * a.W;
    const long b128 = (long)rsis_cdiv(a.Cout, 128) * rsis_cdiv(px, 128) * a.B;
    if (a.Cout <= 64) v = ((long)rsis_cdiv(px, 128) * a.B >= 512) ? 6 : 7;
    else v = b128 >= 512 ? 4 : 5;
  }
  if (v >= 4 && v <= 7) {
    if (!ring_geom) return RSIS_ERR_UNSUPPORTED;
    switch (v) {
      case 4: return launch_ring<128, 128>(a, st);
      case 5: return launch_ring<128, 64>(a, st);
      case 6: return launch_ring<64, 128>(a, st);
      default: return launch_ring<64, 64>(a, st);
    }
  }
  if (v <= 0) {
    const long px = (long)a.H * a.W;
    if (a.Cout <= 32) v = 3;
    else if (a.Cout <= 64 || a.C[0] <= 128 || (long)rsis_cdiv(a.Cout, 128) * rsis_cdiv(px, 128) * a.B < 512) v = 2;   // (K <= 128: 1-2 chunks, 64 -> 256 @56^2: 63 -> 49 us)
    else v = 1;
  }
  switch (v) {
    case 1: return launch_bf16_cfg<1, 128, 128, 1, EPI_PLAIN, RSIS_CKB1>(a, st);
    case 2: return launch_bf16_cfg<1, 64, 128, 1, EPI_PLAIN, RSIS_CKB1>(a, st);
    case 3: return launch_bf16_cfg<1, 32, 256, 1, EPI_PLAIN, RSIS_CKB1>(a, st);
    default: return RSIS_ERR_ARG;
  }
}

// ks in {1, 3}, stride 1 "same" geometry (the caller checked); epi: 0 plain, 1 fused ConvLSTM cell (ks == 3 only)
int rsis_launch_conv_bf16(ConvArgs& a, int ks, int epi, int force_variant, hipStream_t st) {
  if (a.nsrc < 0 || a.nsrc > RSIS_MAX_SRC) return RSIS_ERR_ARG;
  if (ks == 3) return epi == EPI_LSTM ? launch_bf16_k3<EPI_LSTM>(a, st, force_variant) : launch_bf16_k3<EPI_PLAIN>(a, st, force_variant);
  if (ks == 1 && epi == EPI_PLAIN) return launch_bf16_k1(a, st, force_variant);
  return RSIS_ERR_UNSUPPORTED;
}
