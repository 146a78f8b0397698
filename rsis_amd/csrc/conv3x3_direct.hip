// Direct 3x3 / stride 1 / pad 1 convolution for gfx950 on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32), NCHW fp32.
//
// This is the kernel behind the ConvLSTM gates (reference src/modules/clstm.py:43-58, fused EPI_LSTM epilogue), the skip
// convs and conv_out (model.py:43-47,109), the 3x3 convs of the ResNet-101 bottlenecks, and the data-gradient of all of
// them (a 3x3/s1/p1 dgrad is the same conv with flipped taps and swapped channel roles: only the weight packing differs).
//
// Structure (why it exists next to the generic implicit-GEMM kernel): a block owns a TWxTHxNI spatial patch of the
// output and a BM-wide slice of the output channels.  For each chunk of CK=8 input channels it stages
//   * the input patch WITH its 1-pixel halo, zero-filled outside the image:  Xs[CK][NI][TH+2][TW+2]   (row-coalesced
//     global loads, each input element is fetched once per block instead of 9 times), and
//   * the chunk's weights Ws[CK*9][BM]
// in LDS, and the MFMA loop then needs NO address arithmetic and NO masks at all: with the K order (channel pair, tap)
// the B-operand of lane l is Xs[lane_base + compile-time offset] (lanes 0-31 take the even channel of the pair, lanes
// 32-63 the odd one), i.e. a bare `ds_read_b32 v, v_base offset:imm`; the A-operand likewise.  Per MFMA (64 cycles) the
// wave issues 1.25-2 LDS reads and nothing else, which is what makes the exact-f32 MFMA the bound.
// Channel concat (torch.cat at clstm.py:43 / model.py:153) is by pointer: chunks never straddle a source.
//
// EPI_S2: data gradient of the 3x3 / stride 2 / pad 1 convs (the first block of ResNet layers 2-4).  The four parity classes
// of the input pixel (2y+ph, 2x+pw) each receive the taps with r = 1 (ph = 0) or r in {0, 2} (ph = 1), likewise s: every tap
// belongs to exactly one class, reads dy at (y + (r == 0), x + (s == 0)) and accumulates into that class's accumulator.  So
// the block walks the dy grid exactly like a stride-1 conv (9 taps per channel pair, no wasted MFMAs -- the gather
// formulation multiplies 3/4 zeros) and the epilogue scatters the 4 accumulators to the 2x larger dx tile.
//
// EPI_F2: the FORWARD of those 3x3 / stride 2 / pad 1 convs.  Same kernel with a (2 TH + 1) x (2 TW + 1) input patch per
// TW x TH output tile: lane (y, x) reads patch pixel (2y + r, 2x + s) for tap (r, s) -- still one `ds_read_b32` with an
// immediate offset per MFMA (2-way LDS bank conflicts from the pixel stride of 2), plain epilogue on the output grid.
#include "common.h"
#include <stdlib.h>
#ifndef XCD_CHUNKED
#define XCD_CHUNKED 1
#endif

enum { EPI_PLAIN = 0, EPI_LSTM = 1, EPI_S2 = 2, EPI_F2 = 3 };
typedef __attribute__((address_space(3))) void* lds_vp_t;
#define CK RSIS_CK
#ifndef RSIS_ACC_FLUSH
#define RSIS_ACC_FLUSH 4      // chunks (of 8 channels x 9 taps) per accumulation segment; a power of two
#endif
#define RSIS_FLUSH_MIN_CHUNKS 48    // reductions at least this many chunks long (K >= 3456) are summed in segments

typedef const float __attribute__((address_space(1)))* gcf_t;
typedef float __attribute__((address_space(1)))* gf_t;
typedef const f32x4 __attribute__((address_space(1)))* gcf4_t;

// KSP = 2: the 4 waves are 2 K-halves x (BM/32 x BN/32/TN) tiles -- each wave walks every other channel pair of a chunk and the
// two partial accumulators are summed through LDS before the epilogue.  Used for the 8x8 maps (64 pixels = 2 column tiles):
// twice the blocks (32 instead of 64 output rows each) and twice the waves per SIMD on a launch that otherwise fills only
// one wave per SIMD.
// NWV = 8: a 512-thread block -- twice the pixels against ONE weight stage (the weight chunk is ~60 % of what a 64-row block
// copies into LDS per chunk, and the LDS-DMA fill rate of a CU, ~25 GB/s, is what two or three co-resident 4-wave blocks run into).
// floats of LDS one block of a variant needs (two stages of patch + weight chunk)
template <int BM, int TW, int TH, int NI, int EPI, int NWV>
constexpr int direct_lds_floats() {
  constexpr int S = EPI == EPI_F2 ? 2 : 1;
  constexpr int PW = TW * S + 3 - S, PH = TH * S + 3 - S;
  constexpr int XS = CK * NI * PH * PW, WS = CK * 9 * BM, NT = NWV * 64;
  return 2 * ((XS + NT - 1) / NT * NT + WS);
}

// The block program.  bid: the block's index inside its launch (or inside its job of a grouped launch); kz / ksplit: its slice of
// the channel chunks (grid split-K); lds: the block's LDS, direct_lds_floats<...>() floats.
template <int BM, int TW, int TH, int NI, int EPI, int KSP = 1, int NWV = 4, bool FLUSH = false, bool BNE = false>
__device__ __forceinline__ void conv3x3_direct_body(const ConvArgs& p, const int bid, const int kz, const int ksplit, float* const lds) {
#if __HIP_DEVICE_COMPILE__   // (the host pass only needs the launch stub; the buffer-resource builtins do not exist there)
  constexpr int BN = TW * TH * NI;
  constexpr int NT = NWV * 64;                           // threads per block
  constexpr int WGM = BM / 32, WGN = NWV / WGM / KSP;    // BM=64: 2x2 waves, BM=32: 1x4 (KSP = 2: 1x2 tiles x 2 K-halves)
  constexpr int TN = BN / WGN / 32;                 // 32-pixel MFMA column tiles per wave (TM == 1)
  constexpr int S = EPI == EPI_F2 ? 2 : 1;          // pixel stride of the patch reads (forward stride)
  constexpr int PW = TW * S + 3 - S, PH = TH * S + 3 - S;   // S = 1: tile + 1-pixel halo; S = 2: 2T + 1
  constexpr int IMS = PH * PW;                      // one image of the patch
  constexpr int CHS = NI * IMS;                     // channel stride of the patch
  constexpr int XS = CK * CHS, WS = CK * 9 * BM;    // floats per LDS stage
  constexpr int XSP = (XS + NT - 1) / NT * NT;      // the DMA writes whole 64-lane rows: pad the stage
  constexpr int NX = (XS + NT - 1) / NT;            // patch loads per thread per chunk
  constexpr int W_F4 = WS / 4;
  constexpr int NW = (W_F4 + NT - 1) / NT;          // weight float4 loads per thread per chunk
  static_assert(TN >= 1 && BN % (WGN * 32) == 0 && WGM * WGN * KSP == NWV && (CK / 2) % KSP == 0 && (KSP == 1 || EPI != EPI_S2), "tile");

  static_assert(2 * (XSP + WS) == direct_lds_floats<BM, TW, TH, NI, EPI, NWV>(), "LDS size helper out of sync");
  float* const Xs0 = lds;
  float* const Ws0 = lds + 2 * XSP;

  const gcf_t src0 = (gcf_t)p.src[0], src1 = (gcf_t)p.src[1], src2 = (gcf_t)p.src[2];
  const int C0 = p.C[0], C1 = p.C[1], C2 = p.C[2];
  const int q0 = (C0 + CK - 1) / CK, q1 = (C1 + CK - 1) / CK, q2 = (C2 + CK - 1) / CK;   // chunks per source
  const int nq_all = q0 + q1 + q2;
  // split-K over the channel chunks (deep-K / few-pixel layers such as sk5: 2048 channels on an 8x8 map): gridDim.y blocks
  // each take a contiguous chunk range and finish with fp32 atomics into the (zeroed) output
  const int q_begin = (int)((long)nq_all * kz / ksplit), q_end = (int)((long)nq_all * (kz + 1) / ksplit);
  const int nq = q_end - q_begin;
  // H x W: the grid the block tiles walk and the epilogue writes (the output map; == the gathered map except for EPI_F2);
  // Hs x Ws: the gathered (source) map the patch is read from
  const int Hs = p.H, Ws = p.W, HWs = Hs * Ws;
  const int H = EPI == EPI_F2 ? p.Ho : p.H, W = EPI == EPI_F2 ? p.Wo : p.W, HW = H * W, B = p.B;
  const int ldw = p.ldw;

  // ---- block -> (co tile, spatial tile); blocks b, b+8, ... share an XCD: a tile's co tiles stay on one L2 ----
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
  const int b0 = (sp_t / (tiles_x * tiles_y)) * NI;
  const int x0 = tx * TW, y0 = ty * TH;

  const int tid = threadIdx.x;
  const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave / (WGM * WGN);                 // K-half of this wave (0 when KSP == 1)
  const int wm = (wave / WGN) % WGM, wn = wave % WGN;

  // ---- loop-invariant byte offset of this thread's patch elements inside the [CK][H][W] slab of one chunk of image b0;
  // halo / out-of-image elements get an offset beyond the buffer range, which the buffer load turns into a zero ----
  static_assert(NI == 1, "LDS-DMA staging addresses one image per block");
  unsigned xvo[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int e = tid + i * NT;
    const int cl = e / CHS, rem2 = e - cl * CHS;
    const int py = rem2 / PW, pxx = rem2 - py * PW;
    const int gy = y0 * S + py - 1, gx = x0 * S + pxx - 1;
    const bool ok = (e < XS) && ((unsigned)gy < (unsigned)Hs) && ((unsigned)gx < (unsigned)Ws);
    xvo[i] = ok ? (unsigned)(cl * HWs + gy * Ws + gx) * 4u : 0x7FFFFFF0u;
  }
  unsigned wvo[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int idx = tid + i * NT;
    const int row = idx / (BM / 4), c4 = idx % (BM / 4);
    wvo[i] = (unsigned)(row * ldw + c4 * 4) * 4u;
  }

  // ---- per-lane LDS read bases (bytes are immediates in the unrolled loop) ----
  int xoff[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int pp = (wn * TN + j) * 32 + l31;
    const int x = pp % TW, y = (pp / TW) % TH, img = pp / (TW * TH);
    xoff[j] = hi * CHS + img * IMS + y * S * PW + x * S;
  }
  // K-split: wave half wk starts at channel pair wk * (CK/2/KSP) of every chunk (folded into the two read bases)
  constexpr int C2W = CK / 2 / KSP;                 // channel pairs per wave per chunk
  const int woff = hi * BM + wm * 32 + l31 + wk * (C2W * 9 * 2) * BM;
  const int xk = wk * (2 * C2W) * CHS;

  constexpr int NACC = EPI == EPI_S2 ? 4 : 1;   // EPI_S2: one accumulator set per parity class of the input pixel
  f32x16 accs[NACC][TN];
#pragma unroll
  for (int c = 0; c < NACC; ++c)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) accs[c][j][r] = 0.f;
  f32x16 (&acc)[TN] = accs[0];

  const gcf_t wbase = (gcf_t)p.wp + co_t * BM;

  // scalar chunk cursor, positioned on this block's first chunk
  int cs = 0, cq = q_begin;   // source index, chunk index inside the source
  if (cs == 0 && cq >= q0 && q0 < nq_all) { cq -= q0; cs = 1; }
  if (cs == 1 && cq >= q1 && q0 + q1 < nq_all) { cq -= q1; cs = 2; }

  // One chunk = NX dword + NW dwordx4 `buffer_load ... lds` per thread: no staging registers, no ds_write pass, no per-lane
  // 64-bit addresses or predicates (the descriptor's range check zero-fills the halo and the channel tail).
#define DIRECT_ISSUE(QG, BUF)                                                                             \
  {                                                                                                       \
    gcf_t src = src0; int Cs = C0;                                                                        \
    if (cs == 1) { src = src1; Cs = C1; }                                                                 \
    if (cs == 2) { src = src2; Cs = C2; }                                                                 \
    const int c0 = cq * CK;                                                                               \
    const int cn = min(CK, Cs - c0);                                                                      \
    const float* xb = (const float*)src + ((size_t)b0 * Cs + c0) * HWs;                                   \
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, cn * HWs * 4, 0x00020000); \
    float* Xs = Xs0 + (BUF) * XSP + wave * 64;                                                            \
    _Pragma("unroll") for (int i = 0; i < NX; ++i)                                                        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_, (lds_vp_t)(Xs + i * NT), 4, xvo[i], 0, 0, 0);        \
    const float* wrow = (const float*)wbase + (size_t)(QG) * (CK * 9) * ldw;                              \
    const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc((void*)wrow, 0, CK * 9 * ldw * 4, 0x00020000); \
    float* Ws = Ws0 + (BUF) * WS + wave * 256;                                                            \
    _Pragma("unroll") for (int i = 0; i < NW; ++i)                                                        \
      if (W_F4 % NT == 0 || i * NT + wave * 64 < W_F4)                                                    \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_, (lds_vp_t)(Ws + i * NT * 4), 16, wvo[i], 0, 0, 0);  \
    if (++cq == (cs == 0 ? q0 : (cs == 1 ? q1 : q2))) { cq = 0; ++cs; if (cs == 1 && q1 == 0) ++cs; }     \
  }
#define DIRECT_LAND() __builtin_amdgcn_s_waitcnt(0x0F70);   /* vmcnt(0): this wave's DMA has landed in LDS */
  if (nq > 0) DIRECT_ISSUE(q_begin, 0)
  DIRECT_LAND()
  __syncthreads();
  // Long reductions are summed in SEGMENTS: every RSIS_ACC_FLUSH chunks (288 products per output) the MFMA accumulators are added to a
  // second register set and cleared.  An MFMA accumulates its k-steps one after the other, so a K = 18432 dot product (the deepest
  // skip conv) is a chain of 9216 fp32 additions whose rounding error grows like sqrt(9216) eps -- three times what a CPU's blocked
  // sums leave (tools/exp/stop_logit_diag.py: 5.6e-4 against 1.9e-4 on the 1/32-scale skip features, the source of the stop logit's
  // 1.1e-4).  Segments make it sqrt(288) eps per segment plus a short chain over the segments: ~4x less, for 32 VALU instructions
  // per 144 MFMAs.  Same values up to summation order; deterministic.  Its own instantiation (FLUSH), launched for reductions of
  // >= RSIS_FLUSH_MIN_CHUNKS chunks only: compiled into every variant the extra register set cost the training step 3 % (39.6 vs 38.3 ms).
  static_assert(!FLUSH || EPI == EPI_PLAIN || EPI == EPI_LSTM, "segmented accumulation: plain and LSTM epilogues");
  f32x16 tot[FLUSH ? TN : 1];
  if constexpr (FLUSH) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) tot[j][r] = 0.f;
  }
  for (int t = 0; t < nq; ++t) {
    const int cur = t & 1;
    const bool more = t + 1 < nq;
    if constexpr (FLUSH) {
      if (t > 0 && (t & (RSIS_ACC_FLUSH - 1)) == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) { tot[j][r] += acc[j][r]; acc[j][r] = 0.f; }
      }
    }
    if (more) DIRECT_ISSUE(q_begin + t + 1, cur ^ 1)   // stage cur^1 was last read before the barrier that ended step t-1
    {
      const float* Xs = Xs0 + cur * XSP;
      const float* Ws = Ws0 + cur * WS + woff;
#pragma unroll
      for (int c2 = 0; c2 < C2W; ++c2)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const float a = Ws[((c2 * 9 + r * 3 + s) * 2) * BM];
            float b[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j)
              b[j] = Xs[xk + xoff[j] + (2 * c2) * CHS + (EPI == EPI_S2 ? ((r == 0) + 1) * PW + (s == 0) + 1 : r * PW + s)];
            const int cls = EPI == EPI_S2 ? (r != 1) * 2 + (s != 1) : 0;   // compile time after unrolling
#pragma unroll
            for (int j = 0; j < TN; ++j) accs[cls][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[j], accs[cls][j], 0, 0, 0);
          }
    }
    DIRECT_LAND()
    __syncthreads();
  }
#undef DIRECT_ISSUE
#undef DIRECT_LAND
  if constexpr (FLUSH) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] += tot[j][r];
  }

  if constexpr (BNE) {
    // inference, eval-mode BatchNorm folded into the epilogue (rsis_conv2d_fwd_bn_eval): the block's BM (scale, shift) pairs, computed
    // once by its first BM threads into the tail of the (now dead) weight stage -- the epilogue reads them back as float4s, so the fold
    // holds no registers across the tiles (as 2 x 16 registers per lane it took the 32-row variants from 92 to 140 VGPRs)
    float* tab = lds + direct_lds_floats<BM, TW, TH, NI, EPI, NWV>() - 2 * BM;
    if (tid < BM) {
      const int row = co_t * BM + tid;
      float sc = 0.f, sh = 0.f;
      if (row < p.Cout) rsis_bn_affine(rsis_bn_eval_rstd(p.ep_var[row], p.ep_eps), p.ep_gamma[row], p.ep_beta[row], p.ep_mean[row], sc, sh);
      tab[tid] = sc; tab[BM + tid] = sh;
    }
    if constexpr (KSP == 1) __syncthreads();      // (KSP > 1: the barrier of the reduction below)
  }
  if constexpr (KSP > 1) {
    // sum the K-halves: waves with wk > 0 park their accumulators in LDS (the staging buffers are dead after the last barrier)
    float* red = lds + ((wm * WGN + wn) * TN) * 16 * 64;
    if (wk > 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(j * 16 + r) * 64 + lane] = acc[j][r];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] += red[(j * 16 + r) * 64 + lane];
  }
  // ---- epilogue ----
  const int co_base = co_t * BM + wm * 32;
  const gcf_t bias = (gcf_t)p.bias, addend = (gcf_t)p.addend;
  unsigned long long best[4] = {0ull, 0ull, 0ull, 0ull};      // EPI_LSTM + side_key: this lane's best (h, pixel) per hidden channel
  // The bias of this lane's 16 rows, fetched ONCE, before the tiles, through a descriptor (no bias: zero range, loads return 0).
  // Inside the tile loop and under `if (bias)` hipcc branched around every load and waited vmcnt(0) behind it: 4 dependent
  // round trips per tile in the LSTM epilogue, each one also draining the stores of the tile before.
  float bv[16];
  {
    const int nrows = EPI == EPI_LSTM ? 4 * p.hid : p.Cout;
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(bias ? (const void*)p.bias : (const void*)p.wp), 0, bias ? nrows * 4 : 0, 0x00020000);
#pragma unroll
    for (int r = 0; r < 16; ++r)
      bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (unsigned)(co_base + 4 * hi + (r & 3) + 8 * (r >> 2)) * 4u, 0, 0));
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int pp = (wn * TN + j) * 32 + l31;
    const int x = pp % TW, y = (pp / TW) % TH, img = pp / (TW * TH);
    const int ob = b0 + img, oy = y0 + y, ox = x0 + x;
    // (the descriptor paths below run out-of-map lanes through with out-of-range offsets instead of skipping them: without the
    //  branch the loads of the next tile can be issued under the arithmetic of this one)
    const bool inb = ob < B && oy < H && ox < W;
    const bool fast_plain = (EPI == EPI_PLAIN || EPI == EPI_F2) && p.ndst == 1 && ksplit == 1 && (size_t)B * p.Cout * HW * 4 < (1ull << 31);
    const bool fast_lstm = EPI == EPI_LSTM && (size_t)p.B * 4 * p.hid * HW * 4 < (1ull << 31);
    if (!inb && !fast_plain && !fast_lstm) continue;
    const int osp = oy * W + ox;
    if (EPI == EPI_S2) {
      const gf_t d0 = (gf_t)p.dst[0];
      const int Cd0 = p.Cd[0], oH = p.oH, oW = p.oW;
#pragma unroll
      for (int c = 0; c < NACC; ++c) {
        const int py = 2 * oy + (c >> 1), px = 2 * ox + (c & 1);
        if (py >= oH || px >= oW) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co_base + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (co < Cd0) d0[((size_t)ob * Cd0 + co) * ((size_t)oH * oW) + py * oW + px] = accs[c][j][r];
        }
      }
    } else if (EPI == EPI_PLAIN || EPI == EPI_F2) {
      const gf_t d0 = (gf_t)p.dst[0], d1 = (gf_t)p.dst[1], d2 = (gf_t)p.dst[2];
      const int Cd0 = p.Cd[0], Cd1 = p.Cd[1], Cd2 = p.Cd[2], Cout = p.Cout;
      const int e1 = Cd0, e2 = Cd0 + Cd1;
      if (fast_plain) {
        // single destination, no split-K (every trunk / skip / hoisted conv and most data gradients): a row of the tile is ONE
        // buffer store at a per-lane base + row * HW floats; rows >= Cout get an out-of-range offset (dropped by the descriptor).
        // ~3 instructions per stored value instead of ~25 of 64-bit index arithmetic and destination selection: the general
        // path below costs ~8 us of a launch whatever its size (measured by ablation on the bf16 twin of this epilogue).
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)p.dst[0], 0, (unsigned)((size_t)B * Cout * HW * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(addend ? p.addend : p.dst[0]), 0, (unsigned)((size_t)B * Cout * HW * 4), 0x00020000);
        const unsigned vo = (unsigned)((ob * Cout + co_base + 4 * hi) * HW + osp) * 4u;
        const unsigned rowb = (unsigned)HW * 4u;
        const int rows_left = inb ? Cout - (co_base + 4 * hi) : 0;        // rows k of this lane are valid while k < rows_left
        float av[16];
        if (addend) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int k = (r & 3) + 8 * (r >> 2);
            av[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, k < rows_left ? vo + k * rowb : 0x7FFFFFF0u, 0, 0));
          }
        }
        if constexpr (BNE) {
          // inference: y = relu?(bn_eval(conv + bias) + addend) -- the arithmetic of bn_apply_kernel (common.h: rsis_bn_apply)
          const float* tab = lds + direct_lds_floats<BM, TW, TH, NI, EPI, NWV>() - 2 * BM + wm * 32 + 4 * hi;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(tab + 8 * g), h4 = *reinterpret_cast<const f32x4*>(tab + BM + 8 * g);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              const int r = 4 * g + kk, k = kk + 8 * g;
              const float v = rsis_bn_apply(acc[j][r] + bv[r], s4[kk], h4[kk], addend ? av[r] : 0.f, p.ep_relu);
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, k < rows_left ? vo + k * rowb : 0x7FFFFFF0u, 0, 0);
            }
          }
          continue;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = (r & 3) + 8 * (r >> 2);
          float v = acc[j][r] + bv[r];
          if (addend) v += av[r];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, k < rows_left ? vo + k * rowb : 0x7FFFFFF0u, 0, 0);
        }
        continue;
      }
      if (!inb) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_base + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (co >= Cout) continue;
        float v = acc[j][r];
        if (bias && kz == 0) v += bias[co];
        gf_t d = d0;
        int cl = co, Cd = Cd0;
        if (co >= e1) { d = d1; cl = co - e1; Cd = Cd1; }
        if (co >= e2) { d = d2; cl = co - e2; Cd = Cd2; }
        const size_t idx = ((size_t)ob * Cd + cl) * HW + osp;
        if (addend && kz == 0) v += addend[idx];
        if (ksplit > 1) atomicAdd((float*)(d + idx), v);
        else d[idx] = v;
      }
    } else {
      const int hid = p.hid;
      const gcf_t c_prev = (gcf_t)p.c_prev;
      const gf_t c_out = (gf_t)p.c_out, h_out = (gf_t)p.h_out, act_out = (gf_t)p.act_out;
      if (fast_lstm) {
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
        const unsigned vg = (unsigned)((ob * 4 * hid + 4 * jh0) * HW + osp) * 4u;
        const unsigned vs = (unsigned)((ob * hid + jh0) * HW + osp) * 4u;
        float ga[4][4], cpv[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {          // all loads of the tile first
          const bool ok = inb && jh0 + 2 * r4 < hid;
#pragma unroll
          for (int g = 0; g < 4; ++g)
            ga[r4][g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_add, ok ? vg + (8 * r4 + g) * rowb : 0x7FFFFFF0u, 0, 0));
          cpv[r4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_cp, ok ? vs + 2 * r4 * rowb : 0x7FFFFFF0u, 0, 0));
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const bool ok = inb && jh0 + 2 * r4 < hid;
          const float ai = acc[j][4 * r4 + 0] + ga[r4][0] + bv[4 * r4 + 0], af = acc[j][4 * r4 + 1] + ga[r4][1] + bv[4 * r4 + 1];
          const float ao = acc[j][4 * r4 + 2] + ga[r4][2] + bv[4 * r4 + 2], ag = acc[j][4 * r4 + 3] + ga[r4][3] + bv[4 * r4 + 3];
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
      if (!inb) continue;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int jh = (co_base >> 2) + 2 * r4 + hi;   // hidden channel
        if (jh >= hid) continue;
        const int cop = jh * 4;
        float ai = acc[j][4 * r4 + 0], af = acc[j][4 * r4 + 1], ao = acc[j][4 * r4 + 2], ag = acc[j][4 * r4 + 3];
        if (bias) { ai += bias[cop]; af += bias[cop + 1]; ao += bias[cop + 2]; ag += bias[cop + 3]; }
        const size_t gidx = ((size_t)ob * 4 * hid + cop) * HW + osp;
        if (addend) {
          ai += addend[gidx]; af += addend[gidx + HW];
          ao += addend[gidx + 2 * (size_t)HW]; ag += addend[gidx + 3 * (size_t)HW];
        }
        const float gi = rsis_sigmoid(ai), gf = rsis_sigmoid(af), go = rsis_sigmoid(ao), gg = tanhf(ag);
        const size_t sidx = ((size_t)ob * hid + jh) * HW + osp;
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
    // the side feature of model.py:143: max of h over the whole map.  The lanes of a half wave hold 32 pixels of the same four hidden
    // channels: shuffle-reduce their keys, one 64-bit atomic max per channel and half wave (uniform branch, all lanes shuffle)
    if (p.side_key) {
      unsigned long long kk[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) kk[r4] = rsis_key_max32(best[r4]);
      if constexpr (KSP == 1)
        // (parked in the staging buffers: dead since the barrier that ended the last chunk; no LDS of its own -- the 40 KB variants fit
        //  four blocks on a CU exactly)
        rsis_side_key_block((unsigned long long*)lds, wave, WGM, WGN, hi, kk, l31 == 0, p.side_key + (size_t)b0 * p.hid, (co_t * BM) >> 2, p.hid,
                            tiles_x * tiles_y >= RSIS_SIDE_CHECK_TILES);
      else      // (the waves of the second K half have left the kernel: no block barrier here)
        rsis_side_key_max4(p.side_key + (size_t)b0 * p.hid, (co_base >> 2) + hi, p.hid, kk, l31 == 0, tiles_x * tiles_y >= RSIS_SIDE_CHECK_TILES);
    }
  }
#endif
}

template <int BM, int TW, int TH, int NI, int EPI, int KSP = 1, int NWV = 4, bool FLUSH = false, bool BNE = false>
__global__ __launch_bounds__(NWV * 64) void conv3x3_direct_kernel(const ConvArgs p) {
  __shared__ __attribute__((aligned(16))) float lds[direct_lds_floats<BM, TW, TH, NI, EPI, NWV>()];
  conv3x3_direct_body<BM, TW, TH, NI, EPI, KSP, NWV, FLUSH, BNE>(p, blockIdx.x, blockIdx.y, gridDim.y, lds);
}

// ---- grouped launch: several INDEPENDENT convs in ONE grid (rsis_convlstm_fwd_batch: the ConvLSTM levels of one diagonal of the
// decoder's (level, timestep) wavefront -- level i at step t depends on level i-1 at step t and on itself at step t-1, so the cells
// (i, d - i) of a diagonal d are independent).  Alone, each gate launch pays its ramp, prologue, per-chunk barrier stalls and
// epilogue with every block of the grid in the same phase (0.53-0.67 of the f32 MFMA peak per level); together the jobs' blocks
// have different lengths and phases, so one block's prologue / epilogue hides behind another's MFMA loop.  The jobs travel by value
// in the kernel arguments; block b belongs to the job whose [begin, end) range holds it.  Every job runs one of the 32-row, 256-thread
// variants {4, 5, 6} (one kernel cannot mix block sizes, and its LDS footprint is the largest variant's: 40 KB, four blocks per CU).
#define RSIS_DG_MAXJ 8
constexpr int cmax(int a, int b) { return a > b ? a : b; }
struct DirectGroup {
  int n;
  int begin[RSIS_DG_MAXJ + 1];
  int variant[RSIS_DG_MAXJ];
  ConvArgs job[RSIS_DG_MAXJ];
};
static_assert(sizeof(DirectGroup) <= 4000, "kernel arguments are limited to 4 KB");

template <int EPI, bool FLUSH = false>
__global__ __launch_bounds__(256) void conv3x3_direct_group_kernel(const DirectGroup g) {
  constexpr int LMAX = cmax(cmax(direct_lds_floats<32, 16, 8, 1, EPI, 4>(), direct_lds_floats<32, 32, 8, 1, EPI, 4>()),
                            direct_lds_floats<32, 8, 8, 1, EPI, 4>());
  __shared__ __attribute__((aligned(16))) float lds[LMAX];
  const int b = blockIdx.x;
  int j = 0;
#pragma unroll
  for (int k = 1; k < RSIS_DG_MAXJ; ++k) j += (k < g.n && g.begin[k] <= b) ? 1 : 0;     // (begin[] ascending: the last job whose begin <= b)
  const ConvArgs& p = g.job[j];
  const int local = b - g.begin[j];
  switch (g.variant[j]) {
    case 4: conv3x3_direct_body<32, 16, 8, 1, EPI, 1, 4, FLUSH>(p, local, 0, 1, lds); break;
    case 5: conv3x3_direct_body<32, 32, 8, 1, EPI, 1, 4, FLUSH>(p, local, 0, 1, lds); break;
    default: conv3x3_direct_body<32, 8, 8, 1, EPI, 2, 4, FLUSH>(p, local, 0, 1, lds); break;
  }
}

// ------------------------------------------------------------------------------------------------
template <int BM, int TW, int TH, int NI, int EPI, int KSP = 1, int NWV = 4>
static int launch_direct_cfg(ConvArgs& a, hipStream_t st) {
  a.n_co_tiles = rsis_cdiv(a.Cout, BM);
  const int gw = EPI == EPI_F2 ? a.Wo : a.W, gh = EPI == EPI_F2 ? a.Ho : a.H;      // the grid the tiles walk
  a.n_px_tiles = rsis_cdiv(gw, TW) * rsis_cdiv(gh, TH) * rsis_cdiv(a.B, NI);
  const int grid = a.n_co_tiles * 8 * rsis_cdiv(a.n_px_tiles, 8);
  int ksplit = 1;
  if (EPI == EPI_PLAIN && a.ksplit == 0) {    // ksplit == 0: the caller zeroed the output and allows split-K
    int nq = 0;
    for (int s = 0; s < a.nsrc; ++s) nq += (a.C[s] + RSIS_CK - 1) / RSIS_CK;
    const int blocks = a.n_co_tiles * a.n_px_tiles;
    if (blocks < 160 && nq >= 32) {
      ksplit = rsis_cdiv(512, blocks);
      if (ksplit > nq / 8) ksplit = nq / 8;
      if (ksplit > 16) ksplit = 16;
      if (ksplit < 1) ksplit = 1;
    }
  }
  if (a.ep_gamma) {
    // inference with the eval-mode BatchNorm folded into the epilogue (rsis_conv2d_fwd_bn_eval): the BNE instantiations -- segmented
    // accumulation whatever the depth on the 256-thread variants (no segment boundary is crossed below RSIS_ACC_FLUSH chunks: same bits)
    if constexpr (EPI == EPI_PLAIN || EPI == EPI_F2) {     // (EPI_F2, the stride-2 forward: no segmented instantiation, as its plain calls)
      if (ksplit != 1 || a.ndst != 1 || a.Cout % 4 != 0 || (size_t)a.B * a.Cout * a.H * a.W * 4 >= (1ull << 31)) return RSIS_ERR_UNSUPPORTED;
      hipLaunchKernelGGL((conv3x3_direct_kernel<BM, TW, TH, NI, EPI, KSP, NWV, NWV == 4 && EPI == EPI_PLAIN, true>), dim3(grid, 1), dim3(NWV * 64), 0, st, a);
      return rsis_check_launch();
    } else {
      return RSIS_ERR_UNSUPPORTED;       // (nothing launched: the caller runs conv and BatchNorm as two launches)
    }
  }
  if constexpr ((EPI == EPI_PLAIN || EPI == EPI_LSTM) && NWV == 4) {
    // deep reductions (>= RSIS_FLUSH_MIN_CHUNKS chunks of 8 channels per block: the skip convs of the 1/16 and 1/32 scales, the 3x3
    // convs of ResNet layer 4) through the segmented-accumulation instantiation; inference calls (a.precise: nothing is being
    // trained, the call is the reference-parity path of test()) whenever there is more than one segment
    int nq = 0;
    for (int s = 0; s < a.nsrc; ++s) nq += (a.C[s] + RSIS_CK - 1) / RSIS_CK;
    if (nq / ksplit >= RSIS_FLUSH_MIN_CHUNKS || (a.precise && nq / ksplit > RSIS_ACC_FLUSH)) {
      hipLaunchKernelGGL((conv3x3_direct_kernel<BM, TW, TH, NI, EPI, KSP, NWV, true>), dim3(grid, ksplit), dim3(NWV * 64), 0, st, a);
      return rsis_check_launch();
    }
  }
  hipLaunchKernelGGL((conv3x3_direct_kernel<BM, TW, TH, NI, EPI, KSP, NWV>), dim3(grid, ksplit), dim3(NWV * 64), 0, st, a);
  return rsis_check_launch();
}

// variant codes: 1 = BM64 8x8x1 (64 px), 2 = BM64 16x8 (128 px), 3 = BM64 32x8 (256 px), 4 = BM32 16x8, 5 = BM32 32x8,
// 6 = BM32 8x8 with the K range split over two wave pairs; 512-thread blocks: 7 = BM64 32x16 (512 px), 8 = BM32 32x16, 9 = BM64 16x16
template <int EPI>
static int pick_direct_variant(const ConvArgs& a, int force) {
  int v = force;
  if (v <= 0) {
    const bool small_co = a.Cout <= 32;
    if (a.W <= 8 && a.H <= 8) {
      // 8x8 maps: 64-row blocks give at most one wave per SIMD below 1024 blocks; the 32-row K-split variant doubles both
      const long b64 = (long)rsis_cdiv(a.Cout, 64) * a.B;
      const bool gsplit = EPI == EPI_PLAIN && a.ksplit == 0 && b64 < 160;   // the grid-level split-K schedule is tuned for v = 1
      v = (EPI != EPI_S2 && b64 < 1024 && !gsplit) ? 6 : 1;
    }
    else if (a.W <= 16) v = small_co ? 4 : 2;
    else v = small_co ? 5 : 3;
    // keep >= ~1 block per CU: fall back to the smaller pixel tile when the big one leaves CUs idle
    if (v == 3) {
      const long blocks = (long)rsis_cdiv(a.Cout, 64) * rsis_cdiv(a.W, 32) * rsis_cdiv(a.H, 8) * a.B;
      if (blocks < 200) v = 2;
    }
    // measured (B=32 trunk / gate shapes): when the 64-row variant gives fewer than 2 blocks per CU, the 32-row variant
    // (twice the blocks, 3-5 resident per CU) is 8-10 % faster (256ch@16^2: 83 -> 90 TF/s, 128ch@32^2: 91 -> 100 TF/s)
    if (v == 2 || v == 3) {
      const int tw = v == 2 ? 16 : 32;
      const long blocks = (long)rsis_cdiv(a.Cout, 64) * rsis_cdiv(a.W, tw) * rsis_cdiv(a.H, 8) * a.B;
      if (blocks < 512) v = v == 2 ? 4 : 5;
    }
    // 32-pixel-wide maps: the 16 x 8 tile beats the 32 x 8 one on the 32-row variant (gate level 2, product launch: 68.8 -> 66.0 us;
    // its data gradient 80.6 -> 62.4 us)
    if (v == 5 && a.W <= 32 && a.Cout > 32) v = 4;
    // wide maps, few output rows: a 512-thread block (32 rows x 32x16 pixels, 8 waves on one weight stage) -- 64->64 @64^2: 87.7 ->
    // 81.7 us, 64->16 @128^2: 168 -> 151 us, gate level 3 (product launch): 73.0 -> 69.5 us
    if (EPI != EPI_S2 && (v == 3 || v == 5) && a.W >= 64 && a.H >= 16 && a.Cout <= 64) v = 8;
  }
  if (EPI == EPI_S2 && v == 3) v = 5;   // 4 accumulator sets: the 256-pixel x 64-row tile would need 256 accumulator registers
  if (EPI == EPI_S2 && v == 6) v = 1;   // (no K-split variant of the 4-accumulator epilogue)
  return v;
}

template <int EPI>
static int launch_direct_epi(ConvArgs& a, hipStream_t st, int force) {
  const int v = pick_direct_variant<EPI>(a, force);
  switch (v) {
    case 1: return launch_direct_cfg<64, 8, 8, 1, EPI>(a, st);
    case 2: return launch_direct_cfg<64, 16, 8, 1, EPI>(a, st);
    case 3: if constexpr (EPI != EPI_S2) return launch_direct_cfg<64, 32, 8, 1, EPI>(a, st); else return RSIS_ERR_ARG;
    case 4: return launch_direct_cfg<32, 16, 8, 1, EPI>(a, st);
    case 5: return launch_direct_cfg<32, 32, 8, 1, EPI>(a, st);
    case 6: if constexpr (EPI != EPI_S2) return launch_direct_cfg<32, 8, 8, 1, EPI, 2>(a, st); else return RSIS_ERR_ARG;
    case 7: if constexpr (EPI != EPI_S2) return launch_direct_cfg<64, 32, 16, 1, EPI, 1, 8>(a, st); else return RSIS_ERR_ARG;
    case 8: if constexpr (EPI != EPI_S2) return launch_direct_cfg<32, 32, 16, 1, EPI, 1, 8>(a, st); else return RSIS_ERR_ARG;
    case 9: if constexpr (EPI != EPI_S2) return launch_direct_cfg<64, 16, 16, 1, EPI, 1, 8>(a, st); else return RSIS_ERR_ARG;
    default: return RSIS_ERR_ARG;
  }
}

// stride-2 forward: the (2T + 1)^2 patch limits the tile to 16 x 8 (32 rows) or 8 x 8 (64 rows) inside 64 KB of LDS
static int launch_direct_f2(ConvArgs& a, hipStream_t st, int force) {
  int v = force;
  if (v != 1 && v != 4) v = (a.Wo <= 8 || (long)rsis_cdiv(a.Cout, 32) * rsis_cdiv(a.Wo, 16) * rsis_cdiv(a.Ho, 8) * a.B < 256) ? 1 : 4;
  if (v == 1) return launch_direct_cfg<64, 8, 8, 1, EPI_F2>(a, st);
  return launch_direct_cfg<32, 16, 8, 1, EPI_F2>(a, st);
}

int rsis_launch_conv3x3_direct(ConvArgs& a, int epi, int force_variant, hipStream_t st) {
  if (a.nsrc < 0 || a.nsrc > RSIS_MAX_SRC) return RSIS_ERR_ARG;
  if (epi == EPI_F2) return launch_direct_f2(a, st, force_variant);
  if (epi == EPI_LSTM) return launch_direct_epi<EPI_LSTM>(a, st, force_variant);
  if (epi == EPI_S2) return launch_direct_epi<EPI_S2>(a, st, force_variant);
  return launch_direct_epi<EPI_PLAIN>(a, st, force_variant);
}

// n independent 3x3 / stride 1 / pad 1 convs with the fused ConvLSTM cell epilogue in one grid (see conv3x3_direct_group_kernel).
// Jobs are ordered by decreasing work per block (the long blocks start first: longest-processing-time scheduling of the tail).
template <int EPI>
static int launch_direct_group(ConvArgs* jobs, int n, const int* force_variant, hipStream_t st);
int rsis_launch_convlstm_direct_group(ConvArgs* jobs, int n, const int* force_variant, hipStream_t st) {
  return launch_direct_group<EPI_LSTM>(jobs, n, force_variant, st);
}
// ... and with the plain epilogue: the data gradients of the gate convs of one REVERSE diagonal of the wavefront (multi-destination:
// d(up) | d(h_prev)), rsis_conv2d_dgrad_batch
int rsis_launch_conv3x3_direct_group_plain(ConvArgs* jobs, int n, const int* force_variant, hipStream_t st) {
  return launch_direct_group<EPI_PLAIN>(jobs, n, force_variant, st);
}
template <int EPI>
static int launch_direct_group(ConvArgs* jobs, int n, const int* force_variant, hipStream_t st) {
  if (n < 1) return RSIS_OK;
  for (int j0 = 0; j0 < n; j0 += RSIS_DG_MAXJ) {
    const int m = n - j0 < RSIS_DG_MAXJ ? n - j0 : RSIS_DG_MAXJ;
    int order[RSIS_DG_MAXJ], var[RSIS_DG_MAXJ];
    long work[RSIS_DG_MAXJ];
    for (int j = 0; j < m; ++j) {
      ConvArgs& a = jobs[j0 + j];
      if (a.nsrc < 0 || a.nsrc > RSIS_MAX_SRC) return RSIS_ERR_ARG;
      int v = pick_direct_variant<EPI>(a, force_variant ? force_variant[j0 + j] : 0);
      // the grouped kernel's variants: 6 (8x8 maps), 4 (16 x 8 tiles), 5 (32 x 8 tiles), all 32 rows x 256 threads
      if (v == 1) v = 6; else if (v == 2 || v == 9) v = 4; else if (v == 3 || v == 7 || v == 8) v = 5;
      var[j] = v;
      int nq = 0;
      for (int s = 0; s < a.nsrc; ++s) nq += (a.C[s] + RSIS_CK - 1) / RSIS_CK;
      work[j] = (long)nq * (v == 6 ? 64 : (v == 5 ? 256 : 128));
      order[j] = j;
    }
    for (int x = 1; x < m; ++x)                        // insertion sort, descending work per block
      for (int y = x; y > 0 && work[order[y]] > work[order[y - 1]]; --y) { const int t = order[y]; order[y] = order[y - 1]; order[y - 1] = t; }
    DirectGroup g;
    g.n = m;
    int blocks = 0;
    for (int k = 0; k < m; ++k) {
      const int j = order[k];
      ConvArgs a = jobs[j0 + j];
      const int v = var[j];
      const int bm = 32, tw = v == 6 ? 8 : (v == 5 ? 32 : 16);
      a.n_co_tiles = rsis_cdiv(a.Cout, bm);
      a.n_px_tiles = rsis_cdiv(a.W, tw) * rsis_cdiv(a.H, 8) * a.B;
      a.ksplit = 1;
      g.begin[k] = blocks;
      g.variant[k] = v;
      g.job[k] = a;
      blocks += a.n_co_tiles * 8 * rsis_cdiv(a.n_px_tiles, 8);
    }
    for (int k = m; k <= RSIS_DG_MAXJ; ++k) g.begin[k] = blocks;
    bool precise = false;
    for (int k = 0; k < m; ++k) precise = precise || g.job[k].precise != 0;
    if (EPI == EPI_LSTM && precise) hipLaunchKernelGGL((conv3x3_direct_group_kernel<EPI_LSTM, true>), dim3(blocks), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((conv3x3_direct_group_kernel<EPI>), dim3(blocks), dim3(256), 0, st, g);
    if (rsis_check_launch() != RSIS_OK) return RSIS_ERR_LAUNCH;
  }
  return RSIS_OK;
}
