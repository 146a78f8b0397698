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

// RSIS_DTYPE_F32X3: the three bf16 limbs of a pair of fp32 values (v = l0 + l1 + l2 exactly: every residual is exact in fp32)
__device__ __forceinline__ void limbs3(float lo, float hi, unsigned& l0, unsigned& l1, unsigned& l2) {
  l0 = pack_bf16x2(lo, hi);
  const float r0 = lo - __builtin_bit_cast(float, l0 << 16), r1 = hi - __builtin_bit_cast(float, l0 & 0xFFFF0000u);
  l1 = pack_bf16x2(r0, r1);
  l2 = pack_bf16x2(r0 - __builtin_bit_cast(float, l1 << 16), r1 - __builtin_bit_cast(float, l1 & 0xFFFF0000u));
}

// KS: 1 or 3.  BM: output rows per block (32 / 64 / 128 -> 1 / 2 / 4 waves along M).  TW x TH: output pixels per block (KS = 1:
// TH = 1 and TW consecutive pixels of the flattened map).  CKB: input channels per LDS stage.  V4 (KS = 1, H*W % 4 == 0): the
// activation tile is fetched as dwordx4 along the pixels.
// SP = 1: bf16 operands.  SP = 3 (RSIS_DTYPE_F32X3): fp32 arithmetic -- every operand value travels as three bf16 limbs (weights:
// three cell rows per packed row; activations: three LDS planes, split when they are staged) and a k-step issues the six limb
// products of weight >= 2^-16, smallest first, into the same fp32 accumulator.
template <int KS, int BM, int TW, int TH, int EPI, int CKB, bool V4, int SP = 1>
__global__ __launch_bounds__(256) void conv_x3_kernel(const ConvArgs p) {
#if __HIP_DEVICE_COMPILE__
  constexpr int KK = KS * KS, HALO = KS / 2;
  constexpr int BN = TW * TH;
  constexpr int WGM = BM / 32, WGN = 4 / WGM;
  constexpr int TN = BN / WGN / 32;
  constexpr int NCB = CKB / 8;                       // 8-channel blocks per stage
  constexpr int PW = TW + 2 * HALO, PH = TH + 2 * HALO, IMS = PH * PW;
  constexpr int XC = NCB * IMS;                      // 16-byte cells of the activation stage (per limb plane)
  constexpr int WROWS = KK * NCB * SP;               // cell rows of the weight stage (limbs of a row are consecutive rows)
  constexpr int WC = WROWS * BM;                     // ... cells
  constexpr int NXT = V4 ? (NCB * (TW / 4) + 255) / 256 : (XC + 255) / 256;   // staging tasks per thread per chunk
  constexpr int NW = (WC + 255) / 256;
  static_assert(TN >= 1 && BN % (WGN * 32) == 0 && WGM * WGN == 4 && CKB % 16 == 0, "tile");
  static_assert(!V4 || (KS == 1 && TW % 4 == 0), "V4 is the 1x1 path");

  __shared__ __attribute__((aligned(16))) u32x4 lds[2 * (SP * XC + WC)];
  u32x4* const Xs0 = lds;
  u32x4* const Ws0 = lds + 2 * SP * XC;

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
  const int sp_t = (q / p.n_co_tiles) * 8 + xcd;
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
  const int woff = hi * SP * BM + wm * 32 + l31;

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
    const char* wrow = wbase + (size_t)(QG) * WROWS * ldw * 16;                                                    \
    const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc((void*)wrow, 0, WROWS * ldw * 16, 0x00020000); \
    u32x4* Ws = Ws0 + (BUF) * WC + wave * 64;                                                                      \
    _Pragma("unroll") for (int i = 0; i < NW; ++i)                                                                 \
      if (WC % 256 == 0 || i * 256 + wave * 64 < WC)                                                               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_, (lds_vp_t)(Ws + i * 256), 16, wvo[i], 0, 0, 0);              \
    if (++cq == (cs == 0 ? q0 : (cs == 1 ? q1 : q2))) { cq = 0; ++cs; if (cs == 1 && q1 == 0) ++cs; }             \
  }
  // registers -> bf16 cells of LDS stage BUF
#define BF_STORE(BUF)                                                                                              \
  {                                                                                                                \
    u32x4* Xs = Xs0 + (BUF) * (SP * XC);                                                                           \
    _Pragma("unroll") for (int i = 0; i < NXT; ++i) {                                                              \
      const int e = tid + i * 256;                                                                                 \
      if constexpr (V4) {                                                                                          \
        const int cb = e / (TW / 4), x4 = e - cb * (TW / 4);                                                       \
        if (NCB * (TW / 4) % 256 == 0 || cb < NCB) {                                                               \
          _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                          \
            u32x4 c0, c1, c2;                                                                                      \
            _Pragma("unroll") for (int h = 0; h < 4; ++h) {                                                        \
              if constexpr (SP == 1) c0[h] = pack_bf16x2(rx[i][(2 * h) * 4 + k], rx[i][(2 * h + 1) * 4 + k]);      \
              else { unsigned u0, u1, u2; limbs3(rx[i][(2 * h) * 4 + k], rx[i][(2 * h + 1) * 4 + k], u0, u1, u2); c0[h] = u0; c1[h] = u1; c2[h] = u2; } \
            }                                                                                                      \
            Xs[cb * TW + x4 * 4 + k] = c0;                                                                         \
            if constexpr (SP == 3) { Xs[XC + cb * TW + x4 * 4 + k] = c1; Xs[2 * XC + cb * TW + x4 * 4 + k] = c2; } \
          }                                                                                                        \
        }                                                                                                          \
      } else {                                                                                                     \
        if (XC % 256 == 0 || e < XC) {                                                                             \
          u32x4 c0, c1, c2;                                                                                        \
          _Pragma("unroll") for (int h = 0; h < 4; ++h) {                                                          \
            if constexpr (SP == 1) c0[h] = pack_bf16x2(rx[i][2 * h], rx[i][2 * h + 1]);                            \
            else { unsigned u0, u1, u2; limbs3(rx[i][2 * h], rx[i][2 * h + 1], u0, u1, u2); c0[h] = u0; c1[h] = u1; c2[h] = u2; } \
          }                                                                                                        \
          Xs[e] = c0;                                                                                              \
          if constexpr (SP == 3) { Xs[XC + e] = c1; Xs[2 * XC + e] = c2; }                                         \
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
  for (int t = 0; t < ((p.dbg & 16) ? 1 : nq); ++t) {
    const int cur = t & 1;
    const bool more = t + 1 < nq;
    if (more && !(p.dbg & 1)) BF_ISSUE(q_begin + t + 1, cur ^ 1)   // (weight stage cur^1 was last read before the barrier that ended step t-1)
    {
      const u32x4* Xs = Xs0 + cur * (SP * XC);
      const u32x4* Ws = Ws0 + cur * WC + woff;
#pragma unroll
      for (int kk = 0; kk < NCB / 2; ++kk)
#pragma unroll
        for (int r = 0; r < KS; ++r)
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            if ((p.dbg & 4) && (r + s) > 0) continue;
            if constexpr (SP == 1) {
              const bf16x8 a = __builtin_bit_cast(bf16x8, Ws[((r * KS + s) * NCB + 2 * kk) * BM]);
              bf16x8 b[TN];
#pragma unroll
              for (int j = 0; j < TN; ++j) b[j] = __builtin_bit_cast(bf16x8, Xs[xoff[j] + 2 * kk * IMS + r * PW + s]);
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[j], acc[j], 0, 0, 0);
            } else {
              bf16x8 a[3], b[TN][3];
#pragma unroll
              for (int l = 0; l < 3; ++l) a[l] = __builtin_bit_cast(bf16x8, Ws[(((r * KS + s) * NCB + 2 * kk) * 3 + l) * BM]);
#pragma unroll
              for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int l = 0; l < 3; ++l) b[j][l] = __builtin_bit_cast(bf16x8, Xs[l * XC + xoff[j] + 2 * kk * IMS + r * PW + s]);
              // limb products of weight 2^-16 first, then 2^-8, then the leading one
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][2], acc[j], 0, 0, 0);
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[j][1], acc[j], 0, 0, 0);
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[j][0], acc[j], 0, 0, 0);
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][1], acc[j], 0, 0, 0);
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[j][0], acc[j], 0, 0, 0);
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][0], acc[j], 0, 0, 0);
            }
          }
    }
    BF_LAND()
    if (more && !(p.dbg & 2)) BF_STORE(cur ^ 1)
    __syncthreads();
  }
#undef BF_ISSUE
#undef BF_STORE
#undef BF_LAND

  if (p.dbg & 8) return;
  // ---- epilogue (fp32; same accumulator layout as the f32 MFMA kernels: column = pixel l31, rows (r&3) + 8 (r>>2) + 4 hi) ----
  const int co_base = co_t * BM + wm * 32;
  const gcf_t bias = (gcf_t)p.bias, addend = (gcf_t)p.addend;
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
        c_out[sidx] = c;
        h_out[sidx] = h;
        if (act_out) {
          act_out[gidx] = gi; act_out[gidx + HW] = gf;
          act_out[gidx + 2 * (size_t)HW] = go; act_out[gidx + 3 * (size_t)HW] = gg;
        }
      }
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------------
template <int KS, int BM, int TW, int TH, int EPI, int CKB, int SP = 1>
static int launch_x3_cfg(ConvArgs& a, hipStream_t st) {
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
      hipLaunchKernelGGL((conv_x3_kernel<KS, BM, TW, TH, EPI, CKB, true, SP>), dim3(grid, ksplit), dim3(256), 0, st, a);
      return rsis_check_launch();
    }
  }
  hipLaunchKernelGGL((conv_x3_kernel<KS, BM, TW, TH, EPI, CKB, false, SP>), dim3(grid, ksplit), dim3(256), 0, st, a);
  return rsis_check_launch();
}

// RSIS_DTYPE_F32X3 (three limbs: 3x the LDS of a stage, one block per CU, MFMA-bound like the exact-f32 kernels): 3x3 variants
// 1 = BM64 8x8, 2 = BM64 16x8, 4 = BM32 16x8, 5 = BM32 32x8; 1x1 (32-channel stages) 1 = BM128 x 128 px, 2 = BM64 x 128 px,
// 3 = BM32 x 256 px.
template <int EPI>
static int launch_x3_k3(ConvArgs& a, hipStream_t st, int force) {
  int v = force;
  if (v <= 0) {
    if (a.W <= 8 && a.H <= 8) v = 1;
    else if (a.Cout <= 32) v = a.W <= 16 ? 4 : 5;
    else v = 2;
  }
  switch (v) {
    case 1: return launch_x3_cfg<3, 64, 8, 8, EPI, RSIS_CKB3, 3>(a, st);
    case 2: case 3: case 6: return launch_x3_cfg<3, 64, 16, 8, EPI, RSIS_CKB3, 3>(a, st);
    case 4: return launch_x3_cfg<3, 32, 16, 8, EPI, RSIS_CKB3, 3>(a, st);
    case 5: return launch_x3_cfg<3, 32, 32, 8, EPI, RSIS_CKB3, 3>(a, st);
    default: return RSIS_ERR_ARG;
  }
}
static int launch_x3_k1(ConvArgs& a, hipStream_t st, int force) {
  int v = force;
  if (v <= 0) {
    const long px = (long)a.H * a.W;
    if (a.Cout <= 32) v = 3;
    else if (a.Cout <= 64 || (long)rsis_cdiv(a.Cout, 128) * rsis_cdiv(px, 128) * a.B < 256) v = 2;
    else v = 1;
  }
  switch (v) {
    case 1: return launch_x3_cfg<1, 128, 128, 1, EPI_PLAIN, 32, 3>(a, st);
    case 2: return launch_x3_cfg<1, 64, 128, 1, EPI_PLAIN, 32, 3>(a, st);
    case 3: return launch_x3_cfg<1, 32, 256, 1, EPI_PLAIN, 32, 3>(a, st);
    default: return RSIS_ERR_ARG;
  }
}


int rsis_launch_conv_x3(ConvArgs& a, int ks, int epi, int force_variant, hipStream_t st) {
  a.dbg = getenv("RSIS_X3_DBG") ? atoi(getenv("RSIS_X3_DBG")) : 0;
  if (a.nsrc < 0 || a.nsrc > RSIS_MAX_SRC) return RSIS_ERR_ARG;
  if (ks == 3) return epi == EPI_LSTM ? launch_x3_k3<EPI_LSTM>(a, st, force_variant) : launch_x3_k3<EPI_PLAIN>(a, st, force_variant);
  if (ks == 1 && epi == EPI_PLAIN) return launch_x3_k1(a, st, force_variant);
  return RSIS_ERR_UNSUPPORTED;
}
