// Convolutions on CHANNEL-BLOCKED bf16 activations for gfx950 (v_mfma_f32_32x32x16_bf16): the storage half of the `-dtype bf16` path
// (BASELINE.json configs[2..4]; reference layers: the ResNet-101 bottleneck convs of src/modules/vision.py:12-19 via torchvision).
//
// Layout ("blk"): a logical [B][C][H][W] tensor is stored as bf16 [B][C/8][H][W][8] -- the 8 channels of one pixel form one 16-byte
// CELL.  That is exactly the cell conv_bf16.hip builds in LDS out of eight fp32 loads, a conversion and a ds_write_b128 per cell
// (a bf16 MFMA wants 8 consecutive K = input channels per lane).  With the cells already in HBM
//   * staging is `buffer_load_dwordx4 ... lds` for activations AND weights: no staging registers, no conversion pass, 1/8 of the
//     load instructions, half the bytes -- and a ring of NR stages keeps NR - 1 chunks of DMA in flight per block;
//   * halo pixels, the channel tail and ragged tiles are out-of-range offsets that the buffer descriptor turns into zero cells;
//   * the epilogue packs the 4 consecutive output channels a lane holds per 8-row group into one 8-byte store (two lanes = a cell).
// The MFMA loop, the LDS cell layout and the packed bf16 weights (pack.hip modes 5-7) are those of conv_bf16_kernel; the data
// gradient is the same kernel on the flipped / transposed pack.  Stride 1, "same" padding, one source, C % 8 == 0, Cout % 8 == 0.
#include "common.h"
#include <type_traits>

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

__device__ __forceinline__ unsigned blk_pack2(float lo, float hi) {
  const f32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}

// KS: 1 or 3.  BM: output channels per block (32 / 64 / 128 -> 1 / 2 / 4 waves along M).  TW x TH output pixels per block (KS = 1:
// TH = 1 and TW consecutive pixels of the flattened map).  CKB: input channels per ring stage (KS = 3: the pack's chunk, 16).
template <int KS, int BM, int TW, int TH, int CKB, int NR>
__global__ __launch_bounds__(256) void conv_blk_kernel(const ConvArgs p) {
#if __HIP_DEVICE_COMPILE__
  constexpr int KK = KS * KS, HALO = KS / 2;
  constexpr int BN = TW * TH;
  constexpr int WGM = BM / 32, WGN = 4 / WGM;
  constexpr int TN = BN / WGN / 32;
  constexpr int NCB = CKB / 8;
  constexpr int PW = TW + 2 * HALO, PH = TH + 2 * HALO, IMS = PH * PW;
  constexpr int XC = NCB * IMS, WC = KK * NCB * BM;          // cells per activation / weight stage
  constexpr int NXD = (XC + 255) / 256, NWD = (WC + 255) / 256;
  constexpr int XCP = NXD * 256, WCP = NWD * 256;            // stages padded to whole 256-lane DMA rows
  constexpr int C_DMA = NXD + NWD;                           // DMA instructions per wave per chunk (vmcnt bookkeeping)
  static_assert(TN >= 1 && BN % (WGN * 32) == 0 && WGM * WGN == 4 && CKB % 16 == 0 && NR >= 2 && NR <= 4, "tile");
  static_assert((NR - 2) * C_DMA < 64, "vmcnt is 6 bits");

  __shared__ __attribute__((aligned(16))) u32x4 lds[NR * (XCP + WCP)];
  u32x4* const xs0 = lds;
  u32x4* const ws0 = lds + NR * XCP;

  const int Cb = p.C[0] >> 3;
  const int nq = (p.C[0] + CKB - 1) / CKB;
  const int H = KS == 1 ? 1 : p.H, W = KS == 1 ? p.H * p.W : p.W, HW = p.H * p.W;
  const int ldw = p.ldw;

  // block -> (co tile, spatial tile): XCD x owns a contiguous range of the spatial tiles (conv_bf16.hip)
  const int bid = blockIdx.x;
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

  // loop-invariant DMA offsets (bytes inside the [NCB][H][W] cells of one chunk of image b0 / inside one chunk of the pack)
  unsigned xvo[NXD], wvo[NWD];
#pragma unroll
  for (int i = 0; i < NXD; ++i) {
    const int e = tid + i * 256;
    const int cb = e / IMS, rem = e - cb * IMS;
    const int py = rem / PW, pxx = rem - py * PW;
    const int gy = y0 + py - HALO, gx = x0 + pxx - HALO;
    const bool ok = (e < XC) && ((unsigned)gy < (unsigned)H) && ((unsigned)gx < (unsigned)W);
    xvo[i] = ok ? (unsigned)(cb * HW + gy * W + gx) * 16u : RSIS_OOB;
  }
#pragma unroll
  for (int i = 0; i < NWD; ++i) {
    const int idx = tid + i * 256;
    wvo[i] = idx < WC ? (unsigned)((idx / BM) * ldw + idx % BM) * 16u : RSIS_OOB;
  }
  const char* const xbase = (const char*)p.src[0] + (size_t)b0 * Cb * HW * 16;
  const char* const wbase = (const char*)p.wp + (size_t)co_t * BM * 16;

#define BLK_ISSUE(QG)                                                                                              \
  {                                                                                                                \
    const int slot = (QG) % NR;                                                                                    \
    const int cb0 = (QG) * NCB;                                                                                    \
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)(xbase + (size_t)cb0 * HW * 16), 0, \
                                                                         min(NCB, Cb - cb0) * HW * 16, 0x00020000); \
    u32x4* xd = xs0 + slot * XCP + wave * 64;                                                                      \
    _Pragma("unroll") for (int i = 0; i < NXD; ++i)                                                                \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_, (lds_vp_t)(xd + i * 256), 16, xvo[i], 0, 0, 0);                \
    const char* wrow = wbase + (size_t)(QG) * (KK * NCB) * ldw * 16;                                               \
    const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc((void*)wrow, 0, KK * NCB * ldw * 16, 0x00020000); \
    u32x4* wd = ws0 + slot * WCP + wave * 64;                                                                      \
    _Pragma("unroll") for (int i = 0; i < NWD; ++i)                                                                \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_, (lds_vp_t)(wd + i * 256), 16, wvo[i], 0, 0, 0);                \
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

  // prologue: NR - 1 chunks in flight
#pragma unroll
  for (int i = 0; i < NR - 1; ++i)
    if (i < nq) BLK_ISSUE(i)

  for (int t = 0; t < nq; ++t) {
    // chunk t has landed once at most `ahead` younger chunks of this wave's DMA are still outstanding
    const int ahead = min(NR - 2, nq - 1 - t);
    if (NR >= 4 && ahead >= 2) { RSIS_VMCNT(2 * C_DMA); }
    else if (NR >= 3 && ahead == 1) { RSIS_VMCNT(C_DMA); }
    else { RSIS_VMCNT(0); }
    __builtin_amdgcn_s_barrier();                     // every wave's share of chunk t is in LDS; slot (t - 1) % NR is free
    if (t + NR - 1 < nq) BLK_ISSUE(t + NR - 1)
    {
      const u32x4* Xs = xs0 + (t % NR) * XCP;
      const u32x4* Ws = ws0 + (t % NR) * WCP + woff;
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
  }
#undef BLK_ISSUE

  // ---- epilogue: accumulator column = pixel l31, rows i + 8 g + 4 hi (r = 4 g + i) -> the half cell [4 hi, 4 hi + 4) of channel
  //      block co_base / 8 + g: one 8-byte store; blocks >= Cout / 8 fall outside the descriptor ----
  const int co_base = co_t * BM + wm * 32;
  const int Cbo = p.Cout >> 3;
  char* const obase = (char*)p.dst[0] + (size_t)b0 * Cbo * HW * 16;
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)obase, 0, Cbo * HW * 16, 0x00020000);
  // optional blk addend of the output's shape (the data gradient of a residual block's first conv + the gradient of the identity
  // branch): added in fp32 before the one rounding
  const bool has_add = p.addend != nullptr;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(has_add ? (const char*)p.addend + (size_t)b0 * Cbo * HW * 16 : obase), 0, has_add ? Cbo * HW * 16 : 0, 0x00020000);
  // (the addend cells are fetched for the whole tile BEFORE the first store and outside any per-element condition: with the load
  //  inside `if (has_add)` hipcc branches around each one and waits vmcnt(0) behind it -- 4 TN dependent L2 round trips per wave,
  //  each also draining the stores issued so far)
  auto epilogue = [&](auto with_addend, auto with_affine) {
    constexpr bool ADD = decltype(with_addend)::value, AFF = decltype(with_affine)::value;
    // AFF (inference): the eval-mode BatchNorm (+ residual) (+ ReLU) of blk_bn_apply_kernel in the epilogue of the conv that feeds it,
    // in that kernel's own arithmetic (rstd = rsqrtf(var + eps), g = gamma * rstd, y = x * g + (beta - mean * g), then the residual, then
    // the ReLU); this lane's 16 channels are co_base + 8 g + 4 hi + i
    float sc[16], sh[16];
    if constexpr (AFF) {
      const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.ep_gamma, 0, p.Cout * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.ep_beta, 0, p.Cout * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)p.ep_mean, 0, p.Cout * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)p.ep_var, 0, p.Cout * 4, 0x00020000);
      float gv[16], bv[16], mv[16], vv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned o = (unsigned)(co_base + 8 * (r >> 2) + 4 * hi + (r & 3)) * 4u;
        gv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, o, 0, 0));
        bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, o, 0, 0));
        mv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, o, 0, 0));
        vv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, o, 0, 0));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float rstd = rsqrtf(vv[r] + p.ep_eps);
        sc[r] = gv[r] * rstd;
        sh[r] = bv[r] - mv[r] * sc[r];
      }
    }
    const bool pre_round = p.ep_round != 0;
    const bool relu = p.ep_relu != 0;
    unsigned off[TN][4];
    u32x2 av[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int pp = (wn * TN + j) * 32 + l31;
      const int ox = x0 + pp % TW, oy = y0 + pp / TW;
      const bool in = oy < H && ox < W;
      const int osp = oy * W + ox;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cbo = (co_base >> 3) + g;
        off[j][g] = (in && cbo < Cbo) ? (unsigned)(cbo * HW + osp) * 16u + 8u * hi : RSIS_OOB;
        if constexpr (ADD) av[j][g] = __builtin_amdgcn_raw_buffer_load_b64(ra, off[j][g], 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float o0 = acc[j][4 * g], o1 = acc[j][4 * g + 1], o2 = acc[j][4 * g + 2], o3 = acc[j][4 * g + 3];
        if constexpr (AFF) {
          if (pre_round) {        // (the conv's own bf16 store of the unfused path, then its BatchNorm)
            const unsigned q0 = blk_pack2(o0, o1), q1 = blk_pack2(o2, o3);
            o0 = __uint_as_float(q0 << 16); o1 = __uint_as_float(q0 & 0xFFFF0000u); o2 = __uint_as_float(q1 << 16); o3 = __uint_as_float(q1 & 0xFFFF0000u);
          }
          o0 = o0 * sc[4 * g] + sh[4 * g]; o1 = o1 * sc[4 * g + 1] + sh[4 * g + 1];
          o2 = o2 * sc[4 * g + 2] + sh[4 * g + 2]; o3 = o3 * sc[4 * g + 3] + sh[4 * g + 3];
        }
        if constexpr (ADD) {
          o0 += __uint_as_float(av[j][g][0] << 16); o1 += __uint_as_float(av[j][g][0] & 0xFFFF0000u);
          o2 += __uint_as_float(av[j][g][1] << 16); o3 += __uint_as_float(av[j][g][1] & 0xFFFF0000u);
        }
        if constexpr (AFF) {
          if (relu) { o0 = fmaxf(o0, 0.f); o1 = fmaxf(o1, 0.f); o2 = fmaxf(o2, 0.f); o3 = fmaxf(o3, 0.f); }
        }
        const u32x2 v = {blk_pack2(o0, o1), blk_pack2(o2, o3)};
        __builtin_amdgcn_raw_buffer_store_b64(v, ro, off[j][g], 0, 0);
      }
  };
  if (p.ep_gamma) {
    if (has_add) epilogue(std::true_type{}, std::true_type{}); else epilogue(std::false_type{}, std::true_type{});
  } else if (has_add) epilogue(std::true_type{}, std::false_type{});
  else epilogue(std::false_type{}, std::false_type{});
#endif
}

template <int KS, int BM, int TW, int TH, int CKB, int NR>
static int launch_blk(ConvArgs& a, hipStream_t st) {
  a.n_co_tiles = rsis_cdiv(a.Cout, BM);
  const int gw = KS == 1 ? a.H * a.W : a.W, gh = KS == 1 ? 1 : a.H;
  a.n_px_tiles = rsis_cdiv(gw, TW) * rsis_cdiv(gh, TH) * a.B;
  const int grid = a.n_co_tiles * 8 * rsis_cdiv(a.n_px_tiles, 8);
  hipLaunchKernelGGL((conv_blk_kernel<KS, BM, TW, TH, CKB, NR>), dim3(grid), dim3(256), 0, st, a);
  return rsis_check_launch();
}

// variant: 0 = pick; 3x3: 1 = BM64 8x8, 2 = BM64 16x8, 3 = BM64 32x8, 4 = BM32 16x8, 5 = BM32 32x8, 6 = BM128 8x8;
// 1x1: 1 = BM128 x 128 px, 2 = BM64 x 128 px, 3 = BM128 x 64 px, 4 = BM64 x 64 px, 5 = BM32 x 128 px
int rsis_launch_conv_blk(ConvArgs& a, int ks, int variant, hipStream_t st) {
  if (a.nsrc != 1 || (a.C[0] & 7) || (a.Cout & 7) || a.C[0] < 8) return RSIS_ERR_UNSUPPORTED;
  if ((size_t)(a.C[0] >> 3) * a.H * a.W * 16 >= (1ull << 31) || (size_t)(a.Cout >> 3) * a.H * a.W * 16 >= (1ull << 31)) return RSIS_ERR_UNSUPPORTED;
  const long px = (long)a.H * a.W;
  int v = variant;
  if (ks == 3) {
    if (v <= 0) {
      const long t64_8 = (long)a.B * rsis_cdiv(a.H, 8) * rsis_cdiv(a.W, 8);
      if (a.W <= 8) v = a.Cout >= 128 && t64_8 * rsis_cdiv(a.Cout, 128) >= 256 ? 6 : 1;
      else if (a.W <= 16) v = (long)a.B * rsis_cdiv(a.H, 8) * rsis_cdiv(a.Cout, 64) >= 512 ? 2 : 4;
      else v = a.Cout <= 128 ? 5 : 3;       // (tools/blk_bench.py --variants, batch 32: 64 / 128 rows fill the chip better as 32-row tiles)
    }
    switch (v) {
      case 1: return launch_blk<3, 64, 8, 8, 16, 3>(a, st);
      case 2: return launch_blk<3, 64, 16, 8, 16, 3>(a, st);
      case 3: return launch_blk<3, 64, 32, 8, 16, 3>(a, st);
      case 4: return launch_blk<3, 32, 16, 8, 16, 3>(a, st);
      case 5: return launch_blk<3, 32, 32, 8, 16, 3>(a, st);
      case 6: return launch_blk<3, 128, 8, 8, 16, 3>(a, st);
      default: return RSIS_ERR_ARG;
    }
  }
  if (ks == 1) {
    if (v <= 0) {      // measured on the trunk's shapes at batch 32, 224^2 and 256^2 inputs (tools/blk_bench.py --variants)
      const long t128 = (long)a.B * rsis_cdiv(px, 128);
      if (px <= 64) v = a.Cout >= 1024 ? 3 : 4;                                   // 7x7 / 8x8 maps: 64-pixel tiles
      else if (a.Cout >= 128 && t128 * rsis_cdiv(a.Cout, 128) >= 256) v = 1;       // one block per CU or more with the big tile
      else if (t128 * rsis_cdiv(a.Cout, 64) >= 256) v = 2;
      else v = 4;
    }
    switch (v) {
      case 1: return launch_blk<1, 128, 128, 1, 32, 4>(a, st);
      case 2: return launch_blk<1, 64, 128, 1, 32, 4>(a, st);
      case 3: return launch_blk<1, 128, 64, 1, 32, 4>(a, st);
      case 4: return launch_blk<1, 64, 64, 1, 32, 4>(a, st);
      case 5: return launch_blk<1, 32, 128, 1, 32, 4>(a, st);
      default: return RSIS_ERR_ARG;
    }
  }
  return RSIS_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// layout converters at the boundaries of the blocked region: fp32 [B][C][H][W] <-> bf16 [B][C/8][H][W][8]
// (one thread = one cell; the 8 channel reads / writes of a wave are 8 coalesced 256-byte rows)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blk_from_nchw_kernel(const float* __restrict__ x, u32x4* __restrict__ y, int C, int HW, long cells) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= cells) return;
  const long bc = e / HW;                 // b * (C/8) + cb
  const int sp = (int)(e - bc * HW);
  const int Cb = C >> 3;
  const long b = bc / Cb;
  const int cb = (int)(bc - b * Cb);
  const float* s = x + ((size_t)b * C + cb * 8) * HW + sp;
  u32x4 cell;
#pragma unroll
  for (int k = 0; k < 4; ++k) cell[k] = blk_pack2(s[(size_t)(2 * k) * HW], s[(size_t)(2 * k + 1) * HW]);
  y[e] = cell;
}
__global__ __launch_bounds__(256) void blk_to_nchw_kernel(const u32x4* __restrict__ x, float* __restrict__ y, int C, int HW, long cells) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= cells) return;
  const long bc = e / HW;
  const int sp = (int)(e - bc * HW);
  const int Cb = C >> 3;
  const long b = bc / Cb;
  const int cb = (int)(bc - b * Cb);
  float* d = y + ((size_t)b * C + cb * 8) * HW + sp;
  const u32x4 cell = x[e];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    d[(size_t)(2 * k) * HW] = __uint_as_float(cell[k] << 16);
    d[(size_t)(2 * k + 1) * HW] = __uint_as_float(cell[k] & 0xFFFF0000u);
  }
}
int rsis_l_blk_from_nchw(const float* x, void* y, int B, int C, int HW, hipStream_t st) {
  const long cells = (long)B * (C >> 3) * HW;
  hipLaunchKernelGGL(blk_from_nchw_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, x, (u32x4*)y, C, HW, cells);
  return rsis_check_launch();
}
int rsis_l_blk_to_nchw(const void* x, float* y, int B, int C, int HW, hipStream_t st) {
  const long cells = (long)B * (C >> 3) * HW;
  hipLaunchKernelGGL(blk_to_nchw_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, (const u32x4*)x, y, C, HW, cells);
  return rsis_check_launch();
}
