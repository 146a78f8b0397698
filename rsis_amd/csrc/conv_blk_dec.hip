// The recurrent decoder's convolutions on CHANNEL-BLOCKED bf16 tensors for gfx950 (v_mfma_f32_32x32x16_bf16) -- `-dtype bf16`,
// BASELINE.json configs[2..4].  Reference ops: the ConvLSTM gates of src/modules/clstm.py:43-58 inside the decoder loop of
// src/modules/model.py:129-165 (forward, fused cell epilogue), their data gradient, and the time-invariant skip term of the gates.
//
// conv_blk.hip is the trunk's kernel: one source, plain epilogue.  The decoder needs, on the same LDS-DMA ring and MFMA loop,
//   * the channel concat of up to three blk sources ([up(h) | h_prev], torch.cat of clstm.py:43 folded into the chunk cursor),
//   * the fused LSTM cell epilogue: gates = acc + G (blk addend, the hoisted skip term) -> sigma / tanh -> c (fp32 NCHW: the cell
//     state keeps its precision over the T steps), h (blk, rounded ONCE), the saved gates (blk, rows 4 j + gate) and the packed
//     (value, pixel) keys of the global max-pool side feature (model.py:143) -- taken from the fp32 h BEFORE it is rounded for
//     storage: the heads see the unrounded maximum and the arg-max (which routes the pooled gradient) is decided on fp32 values;
//     deciding it on the rounded tensor makes every pixel within 2^-9 of the maximum a tie for "first", and on the 7 / 14-pixel
//     levels that moved the gradients of levels 0-1 from 2 % to 5 % of their float64 value (tools/exp/diag_blk_decoder.py),
//   * a plain epilogue with fp32 bias and up to two blk destinations splitting the output channels (the data gradient's
//     d(up) | d(h_prev), the inverse of the concat),
//   * several independent convs in ONE grid: the cells (level i, step d - i) of a diagonal of the decoder's (level, timestep)
//     wavefront are independent (forward), and so are the cells of a reverse diagonal (backward).  At batch 32 the three coarse
//     levels are 7..28-pixel maps: alone each of their launches is all latency (8-25 us for a few MB); in one grid they hide
//     behind the two fine levels, which run at the memory system's rate.
// Everything a lane stores is an 8-byte half cell (4 consecutive channels of one pixel): the accumulator rows of a lane are
// (r & 3) + 8 (r >> 2) + 4 hi, i.e. four groups of 4 consecutive rows.  For h the rows are gate-interleaved, so a lane holds all four
// gates of hidden channels 2 r4 + hi: the two lanes (l31, hi = 0 / 1) exchange two values each (one wave shuffle pair) to own
// channels 0-3 / 4-7 of the cell.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lds_vp_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define RSIS_OOB 0x7FFFFFF0u
#define RSIS_VMCNT(N) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 15) | (((N) >> 4) << 14))
#ifndef XCD_CHUNKED
#define XCD_CHUNKED 1
#endif

enum { BEPI_PLAIN = 0, BEPI_LSTM = 1 };

__device__ __forceinline__ unsigned bd_pack2(float lo, float hi) {
  const f32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ float bd_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bd_hi(unsigned v) { return __uint_as_float(v & 0xFFFF0000u); }
__device__ __forceinline__ float bd_round(float v) { return bd_lo(bd_pack2(v, 0.f)); }      // the value a bf16 store keeps

constexpr int bd_rows64(int n) { return (n + 63) / 64; }
// 16-byte cells of LDS one block of a tile variant needs: a ring of NR stages of (activation patch + weight chunk), each padded to
// whole 64-lane wave rows only (round 5: padded to 256-lane rows the 32-row variants needed 48 KB and three blocks fitted a CU; the
// 1280-cell stage of the widest variant makes it 40 KB exactly: four)
template <int BM, int TW, int TH, int NR>
constexpr int bd_lds_cells() { return NR * 64 * (bd_rows64(2 * (TH + 2) * (TW + 2)) + bd_rows64(9 * 2 * BM)); }

// The block program.  bid: the block's index inside its job.
template <int BM, int TW, int TH, int EPI, int NR>
__device__ __forceinline__ void bd_body(const BlkConvJob& p, const int bid, u32x4* const lds) {
#if __HIP_DEVICE_COMPILE__
  constexpr int NCB = 2, KK = 9;
  constexpr int BN = TW * TH;
  constexpr int WGM = BM / 32, WGN = 4 / WGM;
  constexpr int TN = BN / WGN / 32;
  constexpr int PW = TW + 2, PH = TH + 2, IMS = PH * PW;
  constexpr int XC = NCB * IMS, WC = KK * NCB * BM;
  // DMA at wave-row granularity: the stage is XR wave rows (64 cells) of activations followed by WR of weights; wave w issues rows w, w + 4, ...
  // (a row is all-x or all-w: the descriptor is a wave-uniform choice), NI instructions per wave and chunk -- one fewer for the waves
  // beyond TR % 4 (their `vmcnt` immediate follows)
  constexpr int XR = bd_rows64(XC), WR = bd_rows64(WC), TR = XR + WR;
  constexpr int NI = (TR + 3) / 4, STG = TR * 64;
  static_assert(TN >= 1 && BN % (WGN * 32) == 0 && WGM * WGN == 4, "tile");
  static_assert(NR * STG == bd_lds_cells<BM, TW, TH, NR>(), "LDS size helper out of sync");
  static_assert(NR >= 2 && NR <= 3 && (NR - 2) * NI < 64, "vmcnt is 6 bits");

  const int H = p.H, W = p.W, HW = H * W;
  const int ldw = p.ldw;
  const int Cb0 = p.C[0] >> 3, Cb1 = p.C[1] >> 3, Cb2 = p.C[2] >> 3;
  const int q0 = (Cb0 + NCB - 1) / NCB, q1 = (Cb1 + NCB - 1) / NCB, q2 = (Cb2 + NCB - 1) / NCB;     // 16-channel chunks per source
  const int nq = q0 + q1 + q2;

  const int xcd = bid & 7, q = bid >> 3;
  const int co_t = q % p.n_co_tiles;
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

  unsigned vo[NI];                 // byte offset of this lane's cell of its wave's i-th row (inside the x chunk or inside the weight chunk)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int g = i * 4 + wave;
    if (g < XR) {
      const int e = g * 64 + lane;
      const int cb = e / IMS, rem = e - cb * IMS;
      const int py = rem / PW, pxx = rem - py * PW;
      const int gy = y0 + py - 1, gx = x0 + pxx - 1;
      const bool ok = (e < XC) && ((unsigned)gy < (unsigned)H) && ((unsigned)gx < (unsigned)W);
      vo[i] = ok ? (unsigned)(cb * HW + gy * W + gx) * 16u : RSIS_OOB;
    } else {
      const int idx = (g - XR) * 64 + lane;
      vo[i] = (g < TR && idx < WC) ? (unsigned)((idx / BM) * ldw + idx % BM) * 16u : RSIS_OOB;
    }
  }
  const bool full_rows = (TR % 4 == 0) || wave < (TR % 4);       // this wave issues NI (else NI - 1) DMA instructions per chunk
  const char* const wbase = (const char*)p.wp + (size_t)co_t * BM * 16;

  // chunk QG of the concatenated K axis = chunk cq of source s (the pack pads every source to whole 16-channel chunks)
#define BD_ISSUE(QG)                                                                                               \
  {                                                                                                                \
    const int slot = (QG) % NR;                                                                                    \
    int cq = (QG);                                                                                                 \
    const char* sb = (const char*)p.src[0];                                                                        \
    int Cbs = Cb0;                                                                                                 \
    if (cq >= q0) {                                                                                                \
      cq -= q0; sb = (const char*)p.src[1]; Cbs = Cb1;                                                             \
      if (cq >= q1) { cq -= q1; sb = (const char*)p.src[2]; Cbs = Cb2; }                                           \
    }                                                                                                              \
    const int cb0 = cq * NCB;                                                                                      \
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc(                                          \
        (void*)(sb + ((size_t)b0 * Cbs + cb0) * HW * 16), 0, min(NCB, Cbs - cb0) * HW * 16, 0x00020000);            \
    const char* wrow = wbase + (size_t)(QG) * (KK * NCB) * ldw * 16;                                               \
    const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc((void*)wrow, 0, KK * NCB * ldw * 16, 0x00020000); \
    u32x4* sd = lds + slot * STG + wave * 64;                                                                      \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                               \
      const int g = i * 4 + wave;                                                                                  \
      if (g < XR) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_, (lds_vp_t)(sd + i * 256), 16, vo[i], 0, 0, 0);     \
      else if (g < TR) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_, (lds_vp_t)(sd + i * 256), 16, vo[i], 0, 0, 0); \
    }                                                                                                              \
  }

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

#pragma unroll
  for (int i = 0; i < NR - 1; ++i)
    if (i < nq) BD_ISSUE(i)

  for (int t = 0; t < nq; ++t) {
    const int ahead = min(NR - 2, nq - 1 - t);
    if (ahead == 1) { if (full_rows) { RSIS_VMCNT(NI); } else { RSIS_VMCNT(NI - 1); } }
    else { RSIS_VMCNT(0); }
    __builtin_amdgcn_s_barrier();
    if (t + NR - 1 < nq) BD_ISSUE(t + NR - 1)
    {
      const u32x4* Xs = lds + (t % NR) * STG;
      const u32x4* Ws = Xs + XR * 64 + woff;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const bf16x8 a = __builtin_bit_cast(bf16x8, Ws[((r * 3 + s) * NCB) * BM]);
          bf16x8 b[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) b[j] = __builtin_bit_cast(bf16x8, Xs[xoff[j] + r * PW + s]);
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[j], acc[j], 0, 0, 0);
        }
    }
  }
#undef BD_ISSUE

  // ---- epilogue ----
  const int co_base = co_t * BM + wm * 32;
  const float* const bias = p.bias;
  if constexpr (EPI == BEPI_PLAIN) {
    // accumulator rows 4 g + i (i = 0..3) of a lane = channels co_base + 8 g + 4 hi + i: the half cell [4 hi, 4 hi + 4) of channel
    // block co_base / 8 + g.  Destination d0 holds the first Cd[0] / 8 blocks of the output channels, d1 the rest.
    const int Cbo = p.Cout >> 3, Cbd0 = p.Cd[0] >> 3, Cbd1 = p.ndst > 1 ? (p.Cd[1] >> 3) : 0;
    const __amdgpu_buffer_rsrc_t ro0 = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.dst[0] + (size_t)b0 * Cbd0 * HW * 16), 0, Cbd0 * HW * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(Cbd1 ? (char*)p.dst[1] + (size_t)b0 * Cbd1 * HW * 16 : (char*)p.dst[0]), 0, Cbd1 * HW * 16, 0x00020000);
    const bool has_add = p.addend != nullptr;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(has_add ? (const char*)p.addend + (size_t)b0 * Cbo * HW * 16 : (const char*)p.dst[0]), 0, has_add ? Cbo * HW * 16 : 0, 0x00020000);
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co_base + (r & 3) + 8 * (r >> 2) + 4 * hi;
      bv[r] = (bias && co < p.Cout) ? bias[co] : 0.f;
    }
    // (addend cells of the whole tile first, outside any per-element condition: see conv_blk.hip)
    auto epilogue = [&](auto with_addend) {
      constexpr bool ADD = decltype(with_addend)::value;
      u32x2 av[TN][4];
      if constexpr (ADD) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int pp = (wn * TN + j) * 32 + l31;
          const int ox = x0 + pp % TW, oy = y0 + pp / TW;
          const int osp = oy * W + ox;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int cbo = (co_base >> 3) + g;
            av[j][g] = __builtin_amdgcn_raw_buffer_load_b64(ra, (oy < H && ox < W && cbo < Cbo) ? (unsigned)(cbo * HW + osp) * 16u + 8u * hi : RSIS_OOB, 0, 0);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int pp = (wn * TN + j) * 32 + l31;
        const int ox = x0 + pp % TW, oy = y0 + pp / TW;
        const bool in = oy < H && ox < W;
        const int osp = oy * W + ox;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int cbo = (co_base >> 3) + g;
          const bool ok = in && cbo < Cbo;
          float o0 = acc[j][4 * g] + bv[4 * g], o1 = acc[j][4 * g + 1] + bv[4 * g + 1];
          float o2 = acc[j][4 * g + 2] + bv[4 * g + 2], o3 = acc[j][4 * g + 3] + bv[4 * g + 3];
          if constexpr (ADD) { o0 += bd_lo(av[j][g][0]); o1 += bd_hi(av[j][g][0]); o2 += bd_lo(av[j][g][1]); o3 += bd_hi(av[j][g][1]); }
          const u32x2 v = {bd_pack2(o0, o1), bd_pack2(o2, o3)};
          const bool first = cbo < Cbd0;
          const unsigned off0 = (ok && first) ? (unsigned)(cbo * HW + osp) * 16u + 8u * hi : RSIS_OOB;
          const unsigned off1 = (ok && !first) ? (unsigned)((cbo - Cbd0) * HW + osp) * 16u + 8u * hi : RSIS_OOB;
          __builtin_amdgcn_raw_buffer_store_b64(v, ro0, off0, 0, 0);
          if (Cbd1) __builtin_amdgcn_raw_buffer_store_b64(v, ro1, off1, 0, 0);
        }
      }
    };
    if (has_add) epilogue(std::true_type{}); else epilogue(std::false_type{});
  } else {
    // LSTM cell (clstm.py:47-58).  Rows are gate-interleaved: register 4 r4 + gate of a lane = gate `gate` of hidden channel
    // jl = 2 r4 + hi of the 8 hidden channels this wave's 32 rows cover (hidden cell block cbh = co_base / 32).
    const int hid = p.hid;
    const int Cbg = (4 * hid) >> 3, Cbh = hid >> 3;
    const int cbh = co_base >> 5;
    const __amdgpu_buffer_rsrc_t r_add = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.addend ? (const char*)p.addend + (size_t)b0 * Cbg * HW * 16 : (const char*)p.h_out), 0, p.addend ? Cbg * HW * 16 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_act = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.act_out ? (char*)p.act_out + (size_t)b0 * Cbg * HW * 16 : (char*)p.h_out), 0, p.act_out ? Cbg * HW * 16 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_h = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.h_out + (size_t)b0 * Cbh * HW * 16), 0, Cbh * HW * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_cp = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.c_prev ? (const char*)(p.c_prev + (size_t)b0 * hid * HW) : (const char*)p.h_out), 0, p.c_prev ? hid * HW * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_c = __builtin_amdgcn_make_buffer_rsrc((void*)(p.c_out + (size_t)b0 * hid * HW), 0, hid * HW * 4, 0x00020000);
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co_base + (r & 3) + 8 * (r >> 2) + 4 * hi;
      bv[r] = (bias && co < 4 * hid) ? bias[co] : 0.f;
    }
    unsigned long long best[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int pp = (wn * TN + j) * 32 + l31;
      const int ox = x0 + pp % TW, oy = y0 + pp / TW;
      const bool in = oy < H && ox < W;
      const int osp = oy * W + ox;
      u32x2 ga[4];
      float cpv[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {            // all loads of the tile first
        const int jh = 8 * cbh + 2 * r4 + hi;
        const bool ok = in && jh < hid;
        ga[r4] = __builtin_amdgcn_raw_buffer_load_b64(r_add, ok ? (unsigned)((4 * cbh + r4) * HW + osp) * 16u + 8u * hi : RSIS_OOB, 0, 0);
        cpv[r4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_cp, ok ? (unsigned)(jh * HW + osp) * 4u : RSIS_OOB, 0, 0));
      }
      float hv[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int jh = 8 * cbh + 2 * r4 + hi;
        const bool ok = in && jh < hid;
        const float ai = acc[j][4 * r4 + 0] + bv[4 * r4 + 0] + bd_lo(ga[r4][0]), af = acc[j][4 * r4 + 1] + bv[4 * r4 + 1] + bd_hi(ga[r4][0]);
        const float ao = acc[j][4 * r4 + 2] + bv[4 * r4 + 2] + bd_lo(ga[r4][1]), ag = acc[j][4 * r4 + 3] + bv[4 * r4 + 3] + bd_hi(ga[r4][1]);
        const float gi = rsis_sigmoid_fast(ai), gf = rsis_sigmoid_fast(af), go = rsis_sigmoid_fast(ao), gg = rsis_tanh_fast(ag);
        const float c = gf * cpv[r4] + gi * gg;     // clstm.py:57
        const float h = go * rsis_tanh_fast(c);      // clstm.py:58
        hv[r4] = h;
        if (p.side_key && ok) { const unsigned long long k = rsis_side_key(h, osp); best[r4] = k > best[r4] ? k : best[r4]; }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c), r_c, ok ? (unsigned)(jh * HW + osp) * 4u : RSIS_OOB, 0, 0);
        const u32x2 av = {bd_pack2(gi, gf), bd_pack2(go, gg)};
        __builtin_amdgcn_raw_buffer_store_b64(av, r_act, ok ? (unsigned)((4 * cbh + r4) * HW + osp) * 16u + 8u * hi : RSIS_OOB, 0, 0);
      }
      // h cell: this lane holds channels 2 r4 + hi; lane (l31, 0) takes channels 0-3, lane (l31, 1) channels 4-7
      const float s0 = hi ? hv[0] : hv[2], s1 = hi ? hv[1] : hv[3];
      const float g0 = __shfl_xor(s0, 32, 64), g1 = __shfl_xor(s1, 32, 64);
      const u32x2 hc = hi ? u32x2{bd_pack2(g0, hv[2]), bd_pack2(g1, hv[3])} : u32x2{bd_pack2(hv[0], g0), bd_pack2(hv[1], g1)};
      __builtin_amdgcn_raw_buffer_store_b64(hc, r_h, (in && cbh < Cbh) ? (unsigned)(cbh * HW + osp) * 16u + 8u * hi : RSIS_OOB, 0, 0);
    }
    if (p.side_key) {
      unsigned long long kk[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) kk[r4] = rsis_key_max32(best[r4]);
      // (parked in the ring slot nobody reads any more: the last chunk sits in slot (nq - 1) % NR, slot nq % NR was last read before the
      //  barrier every wave has passed and nothing was issued into it since -- no LDS of its own: 40 KB is four blocks per CU exactly)
      rsis_side_key_block((unsigned long long*)(lds + (nq % NR) * STG), wave, WGM, WGN, hi, kk, l31 == 0, p.side_key + (size_t)b0 * hid, (co_t * BM) >> 2, hid,
                          tiles_x * tiles_y >= RSIS_SIDE_CHECK_TILES);
    }
  }
#endif
}

// ---- the grouped launch: jobs by value in the kernel arguments, job after job in the grid ----
#define RSIS_BD_MAXJ 8
struct BlkConvGroup {
  int n;
  int begin[RSIS_BD_MAXJ + 1];
  int variant[RSIS_BD_MAXJ];
  BlkConvJob job[RSIS_BD_MAXJ];
};
static_assert(sizeof(BlkConvGroup) <= 4000, "kernel arguments are limited to 4 KB");
constexpr int bd_max(int a, int b) { return a > b ? a : b; }

// tile variants of the group kernel: 4 = 32 rows x 16x8 px, 5 = 32 rows x 32x8 px (wide maps).  40 KB of LDS: four blocks per CU.
// (Until round 5 the 7 / 14-pixel levels ran a 64-row x 8x8 variant whose 18 wave rows of weights per stage set the kernel's LDS to
//  48 KB -- three blocks per CU for EVERY job of the grid; they run variant 4 now: more, shorter blocks of a latency-bound job.)
template <int EPI, int NR>
__global__ __launch_bounds__(256) void conv_blk_dec_group_kernel(const BlkConvGroup g) {
  constexpr int LMAX = bd_max(bd_lds_cells<32, 16, 8, NR>(), bd_lds_cells<32, 32, 8, NR>());
  __shared__ __attribute__((aligned(16))) u32x4 lds[LMAX];
  const int b = blockIdx.x;
  int j = 0;
#pragma unroll
  for (int k = 1; k < RSIS_BD_MAXJ; ++k) j += (k < g.n && g.begin[k] <= b) ? 1 : 0;
  const BlkConvJob& p = g.job[j];
  const int local = b - g.begin[j];
  switch (g.variant[j]) {
    case 4: bd_body<32, 16, 8, EPI, NR>(p, local, lds); break;
    default: bd_body<32, 32, 8, EPI, NR>(p, local, lds); break;
  }
}

static int bd_pick_variant(const BlkConvJob& a, int force) {
  if (force == 4 || force == 5) return force;
  if (force == 1) return 4;            // (the retired 64-row variant: callers that still ask for it get the 16x8 tile)
  return a.W <= 32 ? 4 : 5;
}

// n independent jobs with the same epilogue kind (epi: 0 plain, 1 LSTM) as grouped launches of <= RSIS_BD_MAXJ jobs, longest blocks first
int rsis_launch_conv_blk_dec(BlkConvJob* jobs, int n, int epi, const int* force_variant, hipStream_t st) {
  for (int j0 = 0; j0 < n; j0 += RSIS_BD_MAXJ) {
    const int m = n - j0 < RSIS_BD_MAXJ ? n - j0 : RSIS_BD_MAXJ;
    int order[RSIS_BD_MAXJ], var[RSIS_BD_MAXJ];
    long work[RSIS_BD_MAXJ];
    for (int j = 0; j < m; ++j) {
      const BlkConvJob& a = jobs[j0 + j];
      var[j] = bd_pick_variant(a, force_variant ? force_variant[j0 + j] : 0);
      int nq = 0;
      for (int s = 0; s < a.nsrc; ++s) nq += rsis_cdiv(a.C[s], 16);
      work[j] = (long)(nq + 2) * (var[j] == 5 ? 256 : 128);
      order[j] = j;
    }
    for (int x = 1; x < m; ++x)
      for (int y = x; y > 0 && work[order[y]] > work[order[y - 1]]; --y) { const int t = order[y]; order[y] = order[y - 1]; order[y - 1] = t; }
    BlkConvGroup g;
    g.n = m;
    int blocks = 0;
    for (int k = 0; k < m; ++k) {
      const int j = order[k];
      BlkConvJob a = jobs[j0 + j];
      const int v = var[j];
      const int bm = 32, tw = v == 5 ? 32 : 16;
      a.n_co_tiles = rsis_cdiv(a.Cout, bm);
      a.n_px_tiles = rsis_cdiv(a.W, tw) * rsis_cdiv(a.H, 8) * a.B;
      g.begin[k] = blocks;
      g.variant[k] = v;
      g.job[k] = a;
      blocks += a.n_co_tiles * 8 * rsis_cdiv(a.n_px_tiles, 8);
    }
    for (int k = m; k <= RSIS_BD_MAXJ; ++k) g.begin[k] = blocks;
    static const int nr = (getenv("RSIS_BLKDEC_NR") && getenv("RSIS_BLKDEC_NR")[0] == '3') ? 3 : 2;     // ring depth (A/B switch)
    if (epi == BEPI_LSTM) {
      if (nr == 3) hipLaunchKernelGGL((conv_blk_dec_group_kernel<BEPI_LSTM, 3>), dim3(blocks), dim3(256), 0, st, g);
      else hipLaunchKernelGGL((conv_blk_dec_group_kernel<BEPI_LSTM, 2>), dim3(blocks), dim3(256), 0, st, g);
    } else {
      if (nr == 3) hipLaunchKernelGGL((conv_blk_dec_group_kernel<BEPI_PLAIN, 3>), dim3(blocks), dim3(256), 0, st, g);
      else hipLaunchKernelGGL((conv_blk_dec_group_kernel<BEPI_PLAIN, 2>), dim3(blocks), dim3(256), 0, st, g);
    }
    if (rsis_check_launch() != RSIS_OK) return RSIS_ERR_LAUNCH;
  }
  return RSIS_OK;
}
