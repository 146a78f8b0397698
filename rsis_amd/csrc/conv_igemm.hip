// Implicit-GEMM convolution for gfx950 (CDNA4), NCHW fp32, exact-f32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Replaces, on the RSIS hot path, every nn.Conv2d the reference dispatches to cuDNN:
//   * ConvLSTMCell.Gates + cat + chunk + sigmoid/tanh + cell update  (reference src/modules/clstm.py:43-58)  -> EPI_LSTM
//   * skip convs, conv_out, ResNet-101 trunk convs                     (src/modules/model.py:43-47,109; vision.py:12-19) -> EPI_PLAIN
//   * the data-gradient of all of them (DGRAD gather mode)
//
// GEMM view:  D[co][px] = sum_k Wp[k][co] * Xcol[k][px],  k = (ci over the channel concat, r, s)
//   MFMA "A" operand = packed weights (rows = output channels), "B" operand = gathered pixels, so that the
//   accumulator tile is [co][px] with px = lane&31: stores are coalesced along W in NCHW, and (with
//   gate-interleaved weight rows 4*j+gate) one lane holds i,f,o,g of the same hidden channel/pixel in
//   acc[4*r4 .. 4*r4+3], so the whole LSTM cell update happens in registers.
//   Channel concat (torch.cat at clstm.py:43 / model.py:153) is "by pointer": up to 3 source tensors; the source of a
//   K row is picked with scalar compares on the (wave-uniform) row index.
//   Gather cost: all per-pixel work (validity bit per filter tap, centre address) is hoisted out of the K loop; per K
//   row only SCALAR work remains, so a gathered element costs ~6 VALU + 1 global_load.
//   LDS tiles are k-major ([BK][BM] / [BK][BN]) so every ds_read_b32 of an MFMA operand is conflict-free.
#include "common.h"
#include <type_traits>

enum { EPI_PLAIN = 0, EPI_LSTM = 1, EPI_BN = 2 };     // EPI_BN: EPI_PLAIN + the folded eval-mode BatchNorm (inference; its own instantiations)
#ifndef RSIS_GEMM_NST_DEFAULT
#define RSIS_GEMM_NST_DEFAULT 2
#endif

typedef const float __attribute__((address_space(1)))* gcf_t;   // explicit global pointers: global_load, never flat_load
typedef const char __attribute__((address_space(1)))* gcc_t;
typedef float __attribute__((address_space(1)))* gf_t;
typedef const f32x4 __attribute__((address_space(1)))* gcf4_t;

// V4: 1x1 / stride 1 / single source with H*W % 4 == 0 and C % BK == 0 (the bottleneck 1x1 convs and their data gradients):
// a plain GEMM whose two operand tiles are copied global -> LDS by the LDS-DMA (`buffer_load_dwordx4 ... lds`): the k-major
// tiles are lane-linear images of what the threads fetch, so there are no staging registers, no ds_write pass and no
// per-lane predicates (pixels beyond the tensor get an out-of-range offset, which the descriptor turns into zeros).
typedef __attribute__((address_space(3))) void* lds_vp_t;
// NST (V4 only): depth of the LDS-DMA ring.  NST = 3 keeps two K-tiles in flight and waits with a COUNTED vmcnt + a bare s_barrier --
// `__syncthreads()` carries a fence that hipcc turns into s_waitcnt vmcnt(0), which drains the ring at every K-tile (what made the
// ring-depth experiment of round 5, NOTES (41), a no-op with extra LDS; found with the Winograd kernel in round 6).
template <int BM, int BN, int BK, int WGM, int WGN, int KS, bool DGRAD, int EPI, bool V4, int NST = 2>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
#if __HIP_DEVICE_COMPILE__   // (the host pass only needs the launch stub; the buffer-resource builtins do not exist there)
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  constexpr int KK = KS * KS;
  constexpr int B_ROWS = 256 / BN;       // k rows covered by one load pass of the pixel operand
  constexpr int B_LOADS = BK / B_ROWS;   // dword loads per thread per K-tile
  constexpr int A_F4 = BK * BM / 4;      // float4 per weight tile
  constexpr int A_LOADS = (A_F4 + 255) / 256;
  static_assert(WGM * WGN == 4, "4 waves per block");
  static_assert(BN == 64 || BN == 128 || BN == 256, "BN");
  static_assert(RSIS_KPAD % BK == 0, "BK must divide the packed K padding");
  static_assert(!V4 || (KS == 1 && !DGRAD), "V4 is the 1x1 GEMM path");
  constexpr int BV_COLS = BN / 4;            // float4 columns of the pixel tile
  constexpr int BV_ROWS = 256 / BV_COLS;     // k rows per load pass
  constexpr int BV_LOADS = V4 ? BK / BV_ROWS : 1;
  typedef typename std::conditional<(KK > 32), unsigned long long, unsigned>::type mask_t;

  static_assert(NST == 2 || V4, "deeper rings: the LDS-DMA path only");
  __shared__ __attribute__((aligned(16))) float lds[NST * BK * (BM + BN)];
  float* const As0 = lds;                   // [NST][BK][BM]
  float* const Bs0 = lds + NST * BK * BM;   // [NST][BK][BN]

  // ---- scalar copies of the arguments (no dynamic indexing of the by-value struct: that would spill it to scratch) ----
  const gcc_t src0 = (gcc_t)p.src[0], src1 = (gcc_t)p.src[1], src2 = (gcc_t)p.src[2];
  const int C0 = p.C[0], C1 = p.C[1], C2 = p.C[2];
  const int cb1 = C0, cb2 = C0 + C1, cb3 = C0 + C1 + C2;
  const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo, pad = p.pad, stride = p.stride, sshift = p.sshift;
  const int HoWo = Ho * Wo, HW = H * W;
  const int Npx = p.B * HoWo;
  const int ldw = p.ldw;

  // ---- block -> (co tile, px tile); blocks b, b+8, ... share an XCD (L2): keep a pixel tile's co tiles there ----
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = bid >> 3;
  const int co_t = q % p.n_co_tiles;
  const int px_t = (q / p.n_co_tiles) * 8 + xcd;
  if (px_t >= p.n_px_tiles) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  // ---- per-thread pixel of the gathered operand (fixed for the whole K loop) ----
  const int px_local = tid % BN;
  const int krow0 = __builtin_amdgcn_readfirstlane(tid / BN);
  int px = px_t * BN + px_local;
  const bool pxv = px < Npx;
  if (!pxv) px = 0;
  const int pb = px / HoWo;
  const int psp = px - pb * HoWo;
  const int pho = psp / Wo, pwo = psp - pho * Wo;
  // loop-invariant part of the gather: a validity bit per filter tap (zero padding / stride parity / image border) and the
  // byte offset of the tap-independent "centre" address inside each concat source
  mask_t vmask = 0;
  int center;
  {
    const int hi0 = DGRAD ? pho + pad : pho * stride - pad;
    const int wi0 = DGRAD ? pwo + pad : pwo * stride - pad;
    const int smask = (1 << sshift) - 1;
#pragma unroll
    for (int r = 0; r < KS; ++r)
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        bool ok;
        if (!DGRAD) {
          ok = ((unsigned)(hi0 + r) < (unsigned)H) && ((unsigned)(wi0 + s) < (unsigned)W);
        } else {
          const int th = hi0 - r, tw = wi0 - s;
          ok = (th >= 0) && (tw >= 0) && (((th | tw) & smask) == 0) && ((th >> sshift) < H) && ((tw >> sshift) < W);
        }
        if (ok && pxv) vmask |= (mask_t)1 << (r * KS + s);
      }
    center = DGRAD ? (hi0 >> sshift) * W + (wi0 >> sshift) : (pho * stride) * W + pwo * stride;
  }
  const unsigned voff0 = (unsigned)(pb * C0 * HW + center) * 4u;
  const unsigned voff1 = (unsigned)(pb * C1 * HW + center) * 4u;
  const unsigned voff2 = (unsigned)(pb * C2 * HW + center) * 4u;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float rb[V4 ? 1 : B_LOADS];
  f32x4 ra[V4 ? 1 : A_LOADS];
  // V4 mapping: thread -> (float4 column, first k row); the 4 pixels share an image because H*W % 4 == 0
  const int v_col = tid % BV_COLS, v_row0 = tid / BV_COLS;
  const int v_px = px_t * BN + v_col * 4;
  const int v_b = v_px / HoWo;
  unsigned v_bo[BV_LOADS], v_ao[A_LOADS];   // loop-invariant byte offsets of this thread's float4s inside a K-tile
#pragma unroll
  for (int i = 0; i < BV_LOADS; ++i)       // V4 => H == Ho, W == Wo, one source
    v_bo[i] = v_px < Npx ? (unsigned)(v_b * C0 * HW + (v_px - v_b * HoWo) + (v_row0 + i * BV_ROWS) * HW) * 4u : 0x7FFFFFF0u;
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int idx = tid + i * 256;
    v_ao[i] = (unsigned)((idx / (BM / 4)) * ldw + (idx % (BM / 4)) * 4) * 4u;
  }
  const int ntiles = (p.K + BK - 1) / BK;
  const gcf_t wbase = (gcf_t)p.wp + co_t * BM;
  const unsigned x_bytes = (unsigned)p.B * C0 * HW * 4u;   // V4 only (the host routes tensors >= 2 GiB to the generic path)

  // V4: issue the LDS-DMA of K-tile T into LDS stage BUF (stage BUF was last read before the previous barrier)
#define RSIS_DMA_TILE(T, BUF)                                                                                      \
  {                                                                                                                \
    const unsigned xo = (unsigned)(T) * BK * HW * 4u;                                                              \
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)src0 + xo), 0, x_bytes - xo, 0x00020000); \
    float* Bs = Bs0 + (BUF) * BK * BN + wave * 256;                                                                \
    _Pragma("unroll") for (int i = 0; i < BV_LOADS; ++i)                                                           \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_, (lds_vp_t)(Bs + i * 1024), 16, v_bo[i], 0, 0, 0);              \
    const float* wrow = (const float*)wbase + (size_t)(T) * BK * ldw;                                              \
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc((void*)wrow, 0, BK * ldw * 4, 0x00020000); \
    float* As = As0 + (BUF) * BK * BM + wave * 256;                                                                \
    _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i)                                                            \
      if (A_F4 % 256 == 0 || i * 256 + wave * 64 < A_F4)                                                           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (lds_vp_t)(As + i * 1024), 16, v_ao[i], 0, 0, 0);             \
  }

#define RSIS_LOAD_TILE(T)                                                                                          \
  {                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i) {                                                          \
      const int kl = __builtin_amdgcn_readfirstlane((T) * BK + krow0 + i * B_ROWS); /* scalar from here on */      \
      const int cg = kl / KK;                                                                                      \
      const int rs = kl - cg * KK;                                                                                 \
      const int r = rs / KS, s = rs - r * KS;                                                                      \
      gcc_t src = src0;                                                                                            \
      int ci = cg;                                                                                                 \
      unsigned voff = voff0;                                                                                       \
      if (cg >= cb1) { src = src1; ci = cg - cb1; voff = voff1; }                                                  \
      if (cg >= cb2) { src = src2; ci = cg - cb2; voff = voff2; }                                                  \
      const int soff = DGRAD ? -((r >> sshift) * W + (s >> sshift)) : (r - pad) * W + (s - pad);                   \
      const gcc_t sbase = src + ((long)ci * HW + soff) * 4;                                                        \
      const bool ok = (cg < cb3) && ((vmask >> rs) & 1);                                                           \
      float v = 0.f;                                                                                               \
      if (ok) v = *(gcf_t)(sbase + voff);                                                                          \
      rb[i] = v;                                                                                                   \
    }                                                                                                              \
    const gcf_t wrow = wbase + (size_t)(T) * BK * ldw;                                                             \
    _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i) {                                                          \
      const int idx = tid + i * 256;                                                                               \
      if (A_F4 % 256 == 0 || idx < A_F4) {                                                                         \
        const int row = idx / (BM / 4), c4 = idx % (BM / 4);                                                       \
        ra[i] = *(gcf4_t)(wrow + (size_t)row * ldw + c4 * 4);                                                      \
      }                                                                                                            \
    }                                                                                                              \
  }

#define RSIS_STORE_TILE(BUF)                                                                                       \
  {                                                                                                                \
    float* As = As0 + (BUF) * BK * BM;                                                                             \
    float* Bs = Bs0 + (BUF) * BK * BN;                                                                             \
    _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i) Bs[(krow0 + i * B_ROWS) * BN + px_local] = rb[i];          \
    _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i) {                                                          \
      const int idx = tid + i * 256;                                                                               \
      if (A_F4 % 256 == 0 || idx < A_F4) {                                                                         \
        const int row = idx / (BM / 4), c4 = idx % (BM / 4);                                                       \
        *reinterpret_cast<f32x4*>(As + row * BM + c4 * 4) = ra[i];                                                 \
      }                                                                                                            \
    }                                                                                                              \
  }

  if constexpr (V4 && NST > 2) {
    // ---- ring of NST stages: tile t + NST - 1 is issued while tile t computes; one barrier per tile, at the top ----
    constexpr int C_DMA = BV_LOADS + A_LOADS;          // DMA instructions per wave and K-tile (A_F4 % 256 == 0 for every V4 tile shape)
    static_assert(A_F4 % 256 == 0 && (NST - 2) * C_DMA < 64, "vmcnt bookkeeping");
#pragma unroll
    for (int i = 0; i < NST - 1; ++i)
      if (i < ntiles) RSIS_DMA_TILE(i, i)
    int slot = 0;
    for (int t = 0; t < ntiles; ++t) {
      const int ahead = min(NST - 2, ntiles - 1 - t);  // younger tiles of this wave's DMA that may still be in flight
      if (ahead >= 2) __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * C_DMA) & 15) | (((2 * C_DMA) >> 4) << 14));
      else if (ahead == 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (C_DMA & 15) | ((C_DMA >> 4) << 14));
      else __builtin_amdgcn_s_waitcnt(0x0F70);
      __builtin_amdgcn_s_barrier();                     // every wave's share of tile t is in LDS; the stage of tile t - 1 is free
      const int nslot = slot == 0 ? NST - 1 : slot - 1; // (t + NST - 1) % NST
      if (t + NST - 1 < ntiles) RSIS_DMA_TILE(t + NST - 1, nslot)
      const float* As = As0 + slot * BK * BM + wm * TM * 32 + l31;
      const float* Bs = Bs0 + slot * BK * BN + wn * TN * 32 + l31;
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const int krow = kk * 2 + hi;
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[krow * BM + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs[krow * BN + j * 32];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      slot = slot + 1 == NST ? 0 : slot + 1;
    }
  } else {
  // ---- software pipeline: global->regs for tile t+1 overlaps MFMA on tile t; one barrier per tile ----
  if (ntiles > 0) {
    if constexpr (V4) {
      RSIS_DMA_TILE(0, 0)
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's DMA has landed
    } else {
      RSIS_LOAD_TILE(0)
      RSIS_STORE_TILE(0)
    }
  }
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    const bool more = t + 1 < ntiles;
    if constexpr (V4) {
      if (more) RSIS_DMA_TILE(t + 1, cur ^ 1)
    } else {
      if (more) RSIS_LOAD_TILE(t + 1)
    }
    {
      const float* As = As0 + cur * BK * BM + wm * TM * 32 + l31;
      const float* Bs = Bs0 + cur * BK * BN + wn * TN * 32 + l31;
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const int krow = kk * 2 + hi;
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[krow * BM + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs[krow * BN + j * 32];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if constexpr (V4) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
    } else {
      if (more) RSIS_STORE_TILE(cur ^ 1)
    }
    __syncthreads();
  }
  }
#undef RSIS_LOAD_TILE
#undef RSIS_STORE_TILE
#undef RSIS_DMA_TILE

  if constexpr (EPI == EPI_BN) {
    // the block's BM (scale, shift) pairs of the folded eval-mode BatchNorm, once, into the operand stages (dead: the K loop ends on a
    // barrier) -- registers instead (2 x 16 per lane) took this tile from 92 to 132 VGPRs
    if (tid < BM) {
      const int row = co_t * BM + tid;
      float sc = 0.f, sh = 0.f;
      if (row < p.Cout) rsis_bn_affine(rsis_bn_eval_rstd(p.ep_var[row], p.ep_eps), p.ep_gamma[row], p.ep_beta[row], p.ep_mean[row], sc, sh);
      lds[tid] = sc; lds[BM + tid] = sh;
    }
    __syncthreads();
  }
  // ---- epilogue ----
  const int co_base = co_t * BM + wm * TM * 32;
  const gcf_t bias = (gcf_t)p.bias, addend = (gcf_t)p.addend;
  if (EPI == EPI_PLAIN || EPI == EPI_BN) {
    const gf_t d0 = (gf_t)p.dst[0], d1 = (gf_t)p.dst[1], d2 = (gf_t)p.dst[2];
    const int Cd0 = p.Cd[0], Cd1 = p.Cd[1], Cd2 = p.Cd[2], Cout = p.Cout;
    const int e1 = Cd0, e2 = Cd0 + Cd1;
    const int ostride = p.ostride, oHW = p.oH * p.oW;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int opx = px_t * BN + wn * TN * 32 + j * 32 + l31;
      if (opx >= Npx) continue;
      const int ob = opx / HoWo;
      int osp = opx - ob * HoWo;
      if (ostride > 1) { const int oho = osp / Wo; osp = (oho * ostride) * p.oW + (osp - oho * Wo) * ostride; }
      if (p.ndst == 1 && (size_t)p.B * Cout * oHW * 4 < (1ull << 31)) {
        // single destination (every 1x1 conv of the trunk and its data gradient): a row of the tile is ONE buffer store at a
        // per-lane base + row * oHW floats, rows >= Cout get an out-of-range offset -- ~3 instructions per stored value instead
        // of ~25 of 64-bit index arithmetic and destination selection (the general path costs ~8 us of a 48 us launch)
        const unsigned span = (unsigned)((size_t)p.B * Cout * oHW * 4);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)p.dst[0], 0, span, 0x00020000);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(addend ? p.addend : p.dst[0]), 0, span, 0x00020000);
        const unsigned rowb = (unsigned)oHW * 4u;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row0 = co_base + i * 32 + 4 * hi;
          const unsigned vo = (unsigned)((ob * Cout + row0) * oHW + osp) * 4u;
          const int rows_left = Cout - row0;
          float av[16];
          if (addend) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int k = (r & 3) + 8 * (r >> 2);
              av[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, k < rows_left ? vo + k * rowb : 0x7FFFFFF0u, 0, 0));
            }
          }
          if constexpr (EPI == EPI_BN) {
            // inference: y = relu?(bn_eval(conv + bias) + addend) -- the arithmetic of bn_apply_kernel (common.h: rsis_bn_apply); the
            // block's (scale, shift) table sits in the dead operand stages (filled below the K loop)
            const float* tab = lds + wm * TM * 32 + i * 32 + 4 * hi;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 s4 = *reinterpret_cast<const f32x4*>(tab + 8 * g), h4 = *reinterpret_cast<const f32x4*>(tab + BM + 8 * g);
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) {
                const int r = 4 * g + kk, k = kk + 8 * g;
                float v = acc[i][j][r];
                if (bias) v += k < rows_left ? bias[row0 + k] : 0.f;
                v = rsis_bn_apply(v, s4[kk], h4[kk], addend ? av[r] : 0.f, p.ep_relu);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, k < rows_left ? vo + k * rowb : 0x7FFFFFF0u, 0, 0);
              }
            }
            continue;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int k = (r & 3) + 8 * (r >> 2);
            float v = acc[i][j][r];
            if (bias) v += k < rows_left ? bias[row0 + k] : 0.f;
            if (addend) v += av[r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, k < rows_left ? vo + k * rowb : 0x7FFFFFF0u, 0, 0);
          }
        }
        continue;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (addend) {
          // gradient hand-over (rsis_conv2d_dgrad addend) / fused add: all addend loads of the tile first, then the stores --
          // a load -> add -> store chain per element would expose one memory round trip per accumulator register
          float av[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            av[r] = co < Cout ? addend[((size_t)ob * Cd0 + co) * oHW + osp] : 0.f;     // (addend: single destination only)
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (co < Cout) d0[((size_t)ob * Cd0 + co) * oHW + osp] = acc[i][j][r] + av[r] + (bias ? bias[co] : 0.f);
          }
          continue;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (co >= Cout) continue;
          float v = acc[i][j][r];
          if (bias) v += bias[co];
          gf_t d = d0;
          int cl = co, Cd = Cd0;
          if (co >= e1) { d = d1; cl = co - e1; Cd = Cd1; }
          if (co >= e2) { d = d2; cl = co - e2; Cd = Cd2; }
          d[((size_t)ob * Cd + cl) * oHW + osp] = v;
        }
      }
    }
  } else {
    const int hid = p.hid;
    const gcf_t c_prev = (gcf_t)p.c_prev;
    const gf_t c_out = (gf_t)p.c_out, h_out = (gf_t)p.h_out, act_out = (gf_t)p.act_out;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int opx = px_t * BN + wn * TN * 32 + j * 32 + l31;
      if (opx >= Npx) continue;
      const int ob = opx / HoWo;
      const int osp = opx - ob * HoWo;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int jh = ((co_base + i * 32) >> 2) + 2 * r4 + hi;   // hidden channel
          if (jh >= hid) continue;
          const int cop = jh * 4;                                      // packed gate row of gate i
          float ai = acc[i][j][4 * r4 + 0], af = acc[i][j][4 * r4 + 1];
          float ao = acc[i][j][4 * r4 + 2], ag = acc[i][j][4 * r4 + 3];
          if (bias) { ai += bias[cop]; af += bias[cop + 1]; ao += bias[cop + 2]; ag += bias[cop + 3]; }
          const size_t gidx = ((size_t)ob * 4 * hid + cop) * HoWo + osp;
          if (addend) {
            ai += addend[gidx]; af += addend[gidx + HoWo];
            ao += addend[gidx + 2 * (size_t)HoWo]; ag += addend[gidx + 3 * (size_t)HoWo];
          }
          const float gi = rsis_sigmoid(ai), gf = rsis_sigmoid(af), go = rsis_sigmoid(ao), gg = tanhf(ag);
          const size_t sidx = ((size_t)ob * hid + jh) * HoWo + osp;
          const float cp = c_prev ? c_prev[sidx] : 0.f;
          const float c = gf * cp + gi * gg;       // clstm.py:57
          const float h = go * tanhf(c);           // clstm.py:58
          c_out[sidx] = c;
          h_out[sidx] = h;
          if (act_out) {
            act_out[gidx] = gi; act_out[gidx + HoWo] = gf;
            act_out[gidx + 2 * (size_t)HoWo] = go; act_out[gidx + 3 * (size_t)HoWo] = gg;
          }
        }
      }
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int BK, int WGM, int WGN, int KS, bool DGRAD, int EPI>
static int launch_cfg(ConvArgs& a, hipStream_t st) {
  const long Npx = (long)a.B * a.Ho * a.Wo;
  a.n_co_tiles = rsis_cdiv(a.Cout, BM);
  a.n_px_tiles = rsis_cdiv(Npx, BN);
  const int grid = a.n_co_tiles * 8 * rsis_cdiv(a.n_px_tiles, 8);
  if constexpr (KS == 1 && !DGRAD && (EPI == EPI_PLAIN || EPI == EPI_BN)) {
    if (a.stride == 1 && a.pad == 0 && a.nsrc == 1 && (a.H * a.W) % 4 == 0 && a.C[0] % BK == 0 &&
        (long)a.B * a.C[0] * a.H * a.W * 4 < (1L << 31)) {
      if constexpr (BM == 64 && BN == 64 && BK == 32 && EPI == EPI_PLAIN) {      // the trunk's 1x1 GEMM tile: ring depth by RSIS_GEMM_NST (2, 3 or 4)
        static const int nst = getenv("RSIS_GEMM_NST") ? atoi(getenv("RSIS_GEMM_NST")) : RSIS_GEMM_NST_DEFAULT;
        if (nst == 3) { hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WGM, WGN, KS, DGRAD, EPI, true, 3>), dim3(grid), dim3(256), 0, st, a); return rsis_check_launch(); }
        if (nst == 4) { hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WGM, WGN, KS, DGRAD, EPI, true, 4>), dim3(grid), dim3(256), 0, st, a); return rsis_check_launch(); }
      }
      hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WGM, WGN, KS, DGRAD, EPI, true>), dim3(grid), dim3(256), 0, st, a);
      return rsis_check_launch();
    }
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WGM, WGN, KS, DGRAD, EPI, false>), dim3(grid), dim3(256), 0, st, a);
  return rsis_check_launch();
}

// tile codes: 1 = 32x256, 2 = 32x128, 3 = 64x128, 4 = 128x128, 5 = 64x64, 6 = 128x64 (co x px); +10 selects BK=32
template <int KS, bool DGRAD, int EPI>
static int launch_ks(ConvArgs& a, hipStream_t st, int force_tile) {
  const long Npx = (long)a.B * a.Ho * a.Wo;
  int tile = force_tile;
  if (tile <= 0) {
    if (a.Cout <= 32) tile = (Npx >= 256L * 256) ? 1 : 2;
    else if (KS == 1) tile = 5;   // 1x1 GEMMs of the trunk (8k..131k px, 64..2048 ch): 64x64 tiles measured best (79-95 TF/s)
    else if (a.Cout <= 64) tile = 3;
    else {
      const long b128 = (long)rsis_cdiv(a.Cout, 128) * rsis_cdiv(Npx, 128);
      tile = (b128 >= 200) ? 4 : ((b128 >= 100) ? 6 : 5);
    }
    tile += 10;   // BK = 32 by default
  }
  switch (tile) {
    case 1: return launch_cfg<32, 256, 16, 1, 4, KS, DGRAD, EPI>(a, st);
    case 2: return launch_cfg<32, 128, 16, 1, 4, KS, DGRAD, EPI>(a, st);
    case 3: return launch_cfg<64, 128, 16, 2, 2, KS, DGRAD, EPI>(a, st);
    case 4: return launch_cfg<128, 128, 16, 2, 2, KS, DGRAD, EPI>(a, st);
    case 5: return launch_cfg<64, 64, 16, 2, 2, KS, DGRAD, EPI>(a, st);
    case 6: return launch_cfg<128, 64, 16, 2, 2, KS, DGRAD, EPI>(a, st);
    case 11: return launch_cfg<32, 256, 32, 1, 4, KS, DGRAD, EPI>(a, st);
    case 12: return launch_cfg<32, 128, 32, 1, 4, KS, DGRAD, EPI>(a, st);
    case 13: return launch_cfg<64, 128, 32, 2, 2, KS, DGRAD, EPI>(a, st);
    case 14: return launch_cfg<128, 128, 32, 2, 2, KS, DGRAD, EPI>(a, st);
    case 15: return launch_cfg<64, 64, 32, 2, 2, KS, DGRAD, EPI>(a, st);
    case 16: return launch_cfg<128, 64, 32, 2, 2, KS, DGRAD, EPI>(a, st);
    default: return RSIS_ERR_ARG;
  }
}

int rsis_launch_conv_igemm(ConvArgs& a, int ks, bool dgrad, int epi, int force_tile, hipStream_t st) {
  if (a.nsrc < 0 || a.nsrc > RSIS_MAX_SRC) return RSIS_ERR_ARG;
  if (epi == EPI_LSTM) {
    if (dgrad) return RSIS_ERR_UNSUPPORTED;
    if (ks == 3) return launch_ks<3, false, EPI_LSTM>(a, st, force_tile);
    if (ks == 1) return launch_ks<1, false, EPI_LSTM>(a, st, force_tile);
    return RSIS_ERR_UNSUPPORTED;
  }
  if (a.ep_gamma) {
    // rsis_conv2d_fwd_bn_eval: the 1x1 GEMMs of the trunk (64 x 64 tile) and the 7x7 stem (64 x 128), single destination
    if (dgrad || epi != EPI_PLAIN || a.ndst != 1 || a.Cout <= 32 || a.Cout % 4 != 0 || (size_t)a.B * a.Cout * a.oH * a.oW * 4 >= (1ull << 31))
      return RSIS_ERR_UNSUPPORTED;       // (nothing launched)
    if (ks == 1) return launch_cfg<64, 64, 32, 2, 2, 1, false, EPI_BN>(a, st);
    if (ks == 7 && a.Cout <= 64) return launch_cfg<64, 128, 32, 2, 2, 7, false, EPI_BN>(a, st);
    return RSIS_ERR_UNSUPPORTED;
  }
  if (!dgrad) {
    if (ks == 1) return launch_ks<1, false, EPI_PLAIN>(a, st, force_tile);
    if (ks == 3) return launch_ks<3, false, EPI_PLAIN>(a, st, force_tile);
    if (ks == 7) return launch_ks<7, false, EPI_PLAIN>(a, st, force_tile);
  } else {
    if (ks == 1 && a.stride == 1 && a.pad == 0) return launch_ks<1, false, EPI_PLAIN>(a, st, force_tile);  // 1x1/s1 dgrad == a 1x1 conv
    if (ks == 1) return launch_ks<1, true, EPI_PLAIN>(a, st, force_tile);
    if (ks == 3) return launch_ks<3, true, EPI_PLAIN>(a, st, force_tile);
    if (ks == 7) return launch_ks<7, true, EPI_PLAIN>(a, st, force_tile);
  }
  return RSIS_ERR_UNSUPPORTED;
}
