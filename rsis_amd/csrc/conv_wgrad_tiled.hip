// Weight gradient of the stride-1 "same" convolutions (3x3/p1 and 1x1/p0) for gfx950 on the exact-f32 MFMA
// (v_mfma_f32_32x32x2_f32), NCHW fp32, fed entirely by the LDS-DMA:
//     dW[co][ci][r][s] += sum_{b,y,x} dy[b][co][y][x] * x[b][ci][y+r-pad][x+s-pad]
// (autograd of nn.Conv2d in reference src/modules/clstm.py:17,44 -- the ConvLSTM gates, time-batched over T*B images --
//  model.py:43-47 and the 1x1 / 3x3 convs of the torchvision bottlenecks).
//
// GEMM view D[co][n], n = ci*ks*ks + rs (the reference's weight layout, so the result is accumulated straight into dW),
// the PIXELS are the reduction axis.  A block owns a BM x BN tile of D and walks a range of 8 x TH spatial tiles
// (split-K over tiles and images, fp32 atomics at the end).  Per spatial tile it copies into LDS, with
// `buffer_load ... lds` only (no staging registers, no ds_write pass, double buffered):
//   * the dy tile  As[BM][TP]  as float4 runs along W.  Row r keeps its 16-byte group G at slot G ^ (r & 7): the LDS image
//     stays lane-linear for the DMA (the swizzle is applied to the SOURCE offset) and the MFMA A operand of lane (r, h) is
//     ONE conflict-free ds_read_b128 per tile row y: group 2y+h = pixels (y, 4h..4h+3).  The K index of the MFMA is only
//     a summation index, so "lanes 0-31 take pixels x=0..3, lanes 32-63 take x=4..7 of row y" is as good as any order,
//     provided B uses the same one;
//   * 3x3: the input patch WITH halo  Xs[CI_P][TH+2][10]  as dwords (each input element once instead of 9 times; halo
//     outside the image gets an out-of-range offset, which the buffer descriptor turns into zeros).  The B operand of lane
//     n = (ci, r, s) is Xs[base(n) + 4h + y*10 + e]: a bare ds_read_b32 with an immediate offset;
//   * 1x1: the x tile Bs[BN][TP], same layout / same reads as dy.
// Per MFMA the wave issues <= 1 ds_read_b32 and 1/8..1/4 ds_read_b128; the DMA costs ~15 VMEM + ~25 VALU per thread per
// tile (128 MFMAs per wave for 3x3).  The spatial tile is as wide as the map allows (32x2 / 16x4 / 8x8 pixels for 3x3,
// 32x1 / 16x2 / 8x4 for 1x1) so that dy / x rows are read as full 128-byte lines; maps that no tile shape divides use
// conv_wgrad.hip.
#include "common.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lds_vp_t;

struct WgradTiledArgs {
  const float* dy;   // [B][CoutDy][H][W]
  const float* x;    // [B][Cs][H][W]
  float* dw;         // [Cout][ldo]
  int B, Cs, H, W, Cout;
  int ldo, n_off, interleave_hid;
  int n_co_tiles, n_n_tiles, n_sp_tiles, tiles_per_split;
};

#define RSIS_OOB 0x7FFFFFF0u

// KSP = 2: the block's four waves form a WGM x WGN grid TWICE; the two copies take alternate halves of every stage's reduction
// depth and both add their partial tile with the (already atomic) epilogue -- lets a 32-row tile be only 64 columns wide.
// RAG: maps that no tile shape divides (the 7 / 14 / 28-pixel pyramid of 224 x 224 inputs).  The dy tile (and the x tile of a 1x1) then
// goes dword by dword -- four DMA instructions where the aligned map needs one dwordx4 -- so that every element carries its own
// "beyond the map" test: an element of the last tile column / row whose pixel lies outside gets an out-of-range offset and lands as
// 0.  The MFMA loop is unchanged (zeros add nothing); the tile grid rounds up.  RAG = 2: the map width is a multiple of 4 (28-pixel maps;
// every 1x1 whose H * W is, since a 1x1 may walk the FLATTENED map): a dwordx4 group is inside or outside as a whole, so the one
// DMA per group stays and only carries the two flag bits.
template <int BM, int BN, int WGM, int WGN, int KS, int TW, int KSP = 1, int RAG = 0>
__device__ __forceinline__ void wgrad_tiled_body(const WgradTiledArgs& p, const int bx, const int by) {
#if __HIP_DEVICE_COMPILE__   // (the host pass only needs the launch stub; the buffer-resource builtins do not exist there)
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  constexpr int KK = KS * KS, HALO = KS / 2;
  constexpr int TP = KS == 1 ? 32 : 64;       // pixels per spatial tile = reduction depth per LDS stage
  constexpr int TH = TP / TW;                 // tile = TW x TH pixels: 32x2 / 16x4 / 8x8 (3x3), 32x1 / 16x2 / 8x4 (1x1)
  constexpr int NG = TP / 4;                  // 16-byte groups per dy row
  constexpr int GPR = TW / 4;                 // groups per tile row
  constexpr int PW = TW + 2 * HALO, PH = TH + 2 * HALO, IMS = PW * PH;
  constexpr int CI_P = KS == 1 ? BN : (BN + KK - 2) / KK + 1;   // input channels whose taps cover BN consecutive n
  constexpr int AS = BM * TP;                                   // floats per dy stage
  constexpr int XS = KS == 1 ? BN * TP : (CI_P * IMS + 255) / 256 * 256;
  constexpr int NA = BM * NG / 256;           // dwordx4 DMA per thread per tile (dy)
  constexpr int NB4 = KS == 1 ? BN * NG / 256 : 0;   // dwordx4 DMA per thread per tile (x, 1x1)
  constexpr int NB1 = KS == 1 ? 0 : XS / 256;        // dword DMA per thread per tile (x patch, 3x3)
  static_assert(WGM * WGN * KSP == 4 && (BM * NG) % 256 == 0 && (KS == 3 || (BN * NG) % 256 == 0) && (NG / 2) % KSP == 0, "config");

  __shared__ __attribute__((aligned(16))) float lds[2 * (AS + XS)];
  float* const As0 = lds;
  float* const Xs0 = lds + 2 * AS;

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave / (WGM * WGN), wm = (wave % (WGM * WGN)) / WGN, wn = wave % WGN;
  const int H = p.H, W = p.W, HW = H * W, Cs = p.Cs, Cout = p.Cout;
  const int co_t = bx % p.n_co_tiles, n_t = bx / p.n_co_tiles;
  const int co0 = co_t * BM, n0 = n_t * BN;
  const int ci0 = n0 / KK;                     // first input channel of this block's patch / x tile
  const int Nn = Cs * KK;
  const int tiles_x = RAG ? (W + TW - 1) / TW : W / TW, tiles_y = RAG ? (H + TH - 1) / TH : H / TH;
  const int rx = W - (tiles_x - 1) * TW, ry = H - (tiles_y - 1) * TH;      // valid columns / rows of the last tile column / row
  const int t_begin = by * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.n_sp_tiles);
  if (t_begin >= t_end) return;

  // ---- loop-invariant DMA offsets (bytes, relative to the tile's scalar base) ----
  unsigned voa[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int idx = tid + i * 256;
    const int row = idx / NG, sl = idx % NG;
    const int G = (sl & ~7) | ((sl & 7) ^ (row & 7));     // group held by this slot
    const int y = G / GPR, x4 = G % GPR;
    voa[i] = co0 + row < Cout ? ((unsigned)((co0 + row) * HW + y * W + x4 * 4) * 4u) | (RAG == 2 && x4 * 4 >= rx ? 1u : 0u) | (RAG == 2 && y >= ry ? 2u : 0u) : RSIS_OOB;
  }
  // RAG: dword DMA k of the dy tile moves LDS floats k * 256 + tid = element e of the dwordx4 slot (k * 256 + tid) / 4; bits 0 / 1 of
  // the offset (a multiple of 4) flag "column >= rx" / "row >= ry": outside the map in the last tile column / row
  unsigned voa1[RAG == 1 ? 4 * NA : 1];
  unsigned vob1r[RAG == 1 && NB4 ? 4 * NB4 : 1];
  if constexpr (RAG == 1) {
#pragma unroll
    for (int k = 0; k < 4 * NA; ++k) {
      const int P = k * 256 + tid, idx = P >> 2, e = P & 3;
      const int row = idx / NG, sl = idx % NG;
      const int G = (sl & ~7) | ((sl & 7) ^ (row & 7));
      const int y = G / GPR, x = (G % GPR) * 4 + e;
      voa1[k] = co0 + row < Cout ? ((unsigned)((co0 + row) * HW + y * W + x) * 4u) | (x >= rx ? 1u : 0u) | (y >= ry ? 2u : 0u) : RSIS_OOB;
    }
    if constexpr (KS == 1) {
#pragma unroll
      for (int k = 0; k < 4 * NB4; ++k) {
        const int P = k * 256 + tid, idx = P >> 2, e = P & 3;
        const int row = idx / NG, sl = idx % NG;
        const int G = (sl & ~7) | ((sl & 7) ^ (row & 7));
        const int y = G / GPR, x = (G % GPR) * 4 + e;
        vob1r[k] = n0 + row < Cs ? ((unsigned)((n0 + row) * HW + y * W + x) * 4u) | (x >= rx ? 1u : 0u) | (y >= ry ? 2u : 0u) : RSIS_OOB;
      }
    }
  }
  unsigned vob4[NB4 ? NB4 : 1];
  unsigned vob1[NB1 ? NB1 : 1];
  unsigned cls1[NB1 ? NB1 : 1];   // halo class of a patch element: 1 top row, 2 bottom row, 4 left column, 8 right column
  if constexpr (KS == 1) {
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / NG, sl = idx % NG;
      const int G = (sl & ~7) | ((sl & 7) ^ (row & 7));
      const int y = G / GPR, x4 = G % GPR;
      vob4[i] = n0 + row < Cs ? ((unsigned)((n0 + row) * HW + y * W + x4 * 4) * 4u) | (RAG == 2 && x4 * 4 >= rx ? 1u : 0u) | (RAG == 2 && y >= ry ? 2u : 0u) : RSIS_OOB;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NB1; ++i) {
      const int e = tid + i * 256;
      const int cl = e / IMS, rem = e - cl * IMS;
      const int py = rem / PW, pxx = rem - py * PW;
      const bool ok = cl < CI_P && ci0 + cl < Cs;
      // relative to the element one row above / one column left of the tile origin in channel ci0 (+ (W+1)*4 so that it is >= 0)
      vob1[i] = ok ? (unsigned)(cl * HW + py * W + pxx) * 4u : RSIS_OOB;
      cls1[i] = (py == 0 ? 1u : 0u) | (py - 1 >= ry ? 2u : 0u) | (pxx == 0 ? 4u : 0u) | (pxx - 1 >= rx ? 8u : 0u);      // (aligned maps: ry = TH, rx = TW -- the halo ring)
    }
  }

  // ---- per-lane LDS read offsets ----
  // A (and 1x1 B): lanes 0-31 take group 2g, lanes 32-63 group 2g+1 of MFMA step block g; the group's slot in this lane's
  // row is (2g+h) ^ (row & 7), and row & 7 == l31 & 7 for every 32-row tile
  // (KSP == 2: this wave copy handles the step blocks g = 2 * gg + wk; their offsets live in per-wave registers)
  int sg[NG / 2 / KSP];
  int bo[NG / 2 / KSP];                        // 3x3: patch offset of the first pixel of step block g
#pragma unroll
  for (int gg = 0; gg < NG / 2 / KSP; ++gg) {
    const int g = gg * KSP + (KSP == 1 ? 0 : wk);
    sg[gg] = ((2 * g + hi) ^ (l31 & 7)) * 4;
    bo[gg] = KS == 1 ? 0 : (g / (TW / 8)) * PW + (g % (TW / 8)) * 8;
  }
  const int arow = (wm * TM * 32 + l31) * TP;
  int xb[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nl = wn * TN * 32 + j * 32 + l31;          // column inside the block tile
    if constexpr (KS == 1) {
      xb[j] = nl * TP;
    } else {
      const int n = n0 + nl;
      const int cl = n / KK - ci0, rs = n - (n / KK) * KK;
      xb[j] = cl * IMS + (rs / KS) * PW + (rs % KS) + 4 * hi;
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // scalar tile cursor
  int tb = t_begin / (tiles_x * tiles_y);
  int trem = t_begin - tb * (tiles_x * tiles_y);
  int ty = trem / tiles_x, tx = trem - ty * tiles_x;

#define WG_ISSUE(BUF)                                                                                          \
  {                                                                                                            \
    const int y0 = ty * TH, x0 = tx * TW;                                                                      \
    const float* ab = p.dy + ((size_t)tb * Cout * HW + y0 * W + x0);                                           \
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc((void*)ab, 0, 0x7FFFFFF0, 0x00020000); \
    const unsigned lastm = (tx == tiles_x - 1 ? 1u : 0u) | (ty == tiles_y - 1 ? 2u : 0u);                      \
    if constexpr (RAG == 1) {                                                                                  \
      float* As = As0 + (BUF) * AS + wave * 64;                                                                \
      _Pragma("unroll") for (int k = 0; k < 4 * NA; ++k)                                                       \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (lds_vp_t)(As + k * 256), 4, (voa1[k] & lastm) ? RSIS_OOB : (voa1[k] & ~3u), 0, 0, 0); \
    } else {                                                                                                   \
      float* As = As0 + (BUF) * AS + wave * 256;                                                               \
      _Pragma("unroll") for (int i = 0; i < NA; ++i)                                                           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (lds_vp_t)(As + i * 1024), 16,                           \
                                                 RAG == 2 ? ((voa[i] & lastm) ? RSIS_OOB : (voa[i] & ~3u)) : voa[i], 0, 0, 0); \
    }                                                                                                          \
    if constexpr (KS == 1) {                                                                                   \
      const float* bb = p.x + ((size_t)tb * Cs * HW + y0 * W + x0);                                            \
      const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc((void*)bb, 0, 0x7FFFFFF0, 0x00020000); \
      if constexpr (RAG == 1) {                                                                                \
        float* Xs = Xs0 + (BUF) * XS + wave * 64;                                                              \
        _Pragma("unroll") for (int k = 0; k < 4 * NB4; ++k)                                                    \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_, (lds_vp_t)(Xs + k * 256), 4, (vob1r[k] & lastm) ? RSIS_OOB : (vob1r[k] & ~3u), 0, 0, 0); \
      } else {                                                                                                 \
        float* Xs = Xs0 + (BUF) * XS + wave * 256;                                                             \
        _Pragma("unroll") for (int i = 0; i < NB4; ++i)                                                        \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_, (lds_vp_t)(Xs + i * 1024), 16,                         \
                                                   RAG == 2 ? ((vob4[i] & lastm) ? RSIS_OOB : (vob4[i] & ~3u)) : vob4[i], 0, 0, 0); \
      }                                                                                                        \
    } else {                                                                                                   \
      const float* bb = p.x + (((size_t)tb * Cs + ci0) * HW + (y0 - 1) * W + (x0 - 1));                        \
      const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc((void*)bb, 0, 0x7FFFFFF0, 0x00020000); \
      const unsigned edge = (y0 == 0 ? 1u : 0u) | (ty == tiles_y - 1 ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (tx == tiles_x - 1 ? 8u : 0u); \
      float* Xs = Xs0 + (BUF) * XS + wave * 64;                                                                \
      _Pragma("unroll") for (int i = 0; i < NB1; ++i) {                                                        \
        const unsigned vo = (cls1[i] & edge) ? RSIS_OOB : vob1[i];                                             \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_, (lds_vp_t)(Xs + i * 256), 4, vo, 0, 0, 0);               \
      }                                                                                                        \
    }                                                                                                          \
    if (++tx == tiles_x) { tx = 0; if (++ty == tiles_y) { ty = 0; ++tb; } }                                    \
  }
#define WG_LAND() __builtin_amdgcn_s_waitcnt(0x0F70);   /* vmcnt(0): this wave's DMA has landed in LDS */

  WG_ISSUE(0)
  WG_LAND()
  __syncthreads();
  const int ntl = t_end - t_begin;
  for (int t = 0; t < ntl; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntl) WG_ISSUE(cur ^ 1)   // stage cur^1 was last read before the barrier that ended step t-1
    {
      const float* As = As0 + cur * AS + arow;
      const float* Xs = Xs0 + cur * XS;
#pragma unroll
      for (int g = 0; g < NG / 2 / KSP; ++g) {
        f32x4 a4[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) a4[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * TP + sg[g]);
        if constexpr (KS == 1) {
          f32x4 b4[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) b4[j] = *reinterpret_cast<const f32x4*>(Xs + xb[j] + sg[g]);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][e], b4[j][e], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float b[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Xs[xb[j] + (KSP == 1 ? (g / (TW / 8)) * PW + (g % (TW / 8)) * 8 : bo[g]) + e];   // pixel (y, x) of group 2g+h, element e
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][e], b[j], acc[i][j], 0, 0, 0);
          }
        }
      }
    }
    WG_LAND()
    __syncthreads();
  }
#undef WG_ISSUE
#undef WG_LAND

  // ---- epilogue: fp32 atomics into dW (reference layout) ----
  // (buffer atomics: a row of the tile is a per-lane base + a scalar multiple of ldo; the gate-interleaved rows 4 j + g of the
  //  ConvLSTM weights go to row g * hid + j; rows >= Cout get an out-of-range offset and are dropped)
  const __amdgpu_buffer_rsrc_t rdw = __builtin_amdgcn_make_buffer_rsrc((void*)p.dw, 0, (unsigned)((size_t)Cout * p.ldo * 4), 0x00020000);
  const int ihid = p.interleave_hid;
  const unsigned ldb = (unsigned)p.ldo * 4u;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * TN * 32 + j * 32 + l31;
    if (n >= Nn) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row0 = co0 + wm * TM * 32 + i * 32 + 4 * hi;       // (a multiple of 4)
      const int rows_left = Cout - row0;
      const unsigned vo = (unsigned)((ihid > 0 ? row0 >> 2 : row0) * p.ldo + p.n_off + n) * 4u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2);
        const unsigned ro = (unsigned)(ihid > 0 ? (r & 3) * ihid + 2 * (r >> 2) : k) * ldb;
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[i][j][r], rdw, k < rows_left ? vo + ro : 0x7FFFFFF0u, 0, 0);
      }
    }
  }
#endif
}

template <int BM, int BN, int WGM, int WGN, int KS, int TW, int KSP = 1>
__global__ __launch_bounds__(256) void conv_wgrad_tiled_kernel(const WgradTiledArgs p) {
  wgrad_tiled_body<BM, BN, WGM, WGN, KS, TW, KSP>(p, blockIdx.x, blockIdx.y);
}

// ---- grouped launch: the weight gradients of many layers in ONE grid (rsis_conv2d_wgrad_batch).  A weight gradient is off the
// backward pass's critical path (nothing reads it before the optimizer step), so the training driver parks them and flushes
// them together: every launch costs ~15 us of ramp, prologue and drain whatever its size, and alone a layer has to split its
// pixel axis 8-32 ways to fill the chip -- each split a dW-sized pass of fp32 atomics -- while forty layers together fill it with
// one or two.  The jobs travel by value in the kernel arguments (no device-side table, nothing to keep alive under graph replay);
// block b belongs to the job whose [begin, end) range holds it and plays block (x, y) = (b' % tiles, b' / tiles) of that job. ----
#define RSIS_WG_MAXJ 48
struct WgradTiledGroup {
  int n;
  int begin[RSIS_WG_MAXJ + 1];
  WgradTiledArgs job[RSIS_WG_MAXJ];
};
static_assert(sizeof(WgradTiledGroup) <= 4000, "kernel arguments are limited to 4 KB");

template <int BM, int BN, int WGM, int WGN, int KS, int TW, int KSP = 1, int RAG = 0>
__global__ __launch_bounds__(256) void conv_wgrad_tiled_group_kernel(const WgradTiledGroup g) {
  const int b = blockIdx.x;
  int lo = 0, hi = g.n - 1;
  while (lo < hi) {                       // last job whose begin <= b (uniform: scalar code)
    const int mid = (lo + hi + 1) >> 1;
    if (g.begin[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const WgradTiledArgs& p = g.job[lo];
  const int local = b - g.begin[lo];
  const int ntile = p.n_co_tiles * p.n_n_tiles;
  wgrad_tiled_body<BM, BN, WGM, WGN, KS, TW, KSP, RAG>(p, local % ntile, local / ntile);
}

template <int BM, int BN, int WGM, int WGN, int KS, int TW, int KSP = 1>
static int launch_tiled_cfg(WgradTiledArgs& a, hipStream_t st) {
  constexpr int TH = (KS == 1 ? 32 : 64) / TW;
  a.n_co_tiles = rsis_cdiv(a.Cout, BM);
  a.n_n_tiles = rsis_cdiv((long)a.Cs * KS * KS, BN);
  a.n_sp_tiles = a.B * (a.H / TH) * (a.W / TW);
  const int ntile = a.n_co_tiles * a.n_n_tiles;
  // split-K over the spatial tiles: fill the resident block slots (2 blocks per CU for the 128-wide tiles, 3 otherwise) in one
  // round, keeping >= 2 spatial tiles per split
  // Every split adds a full dW-sized pass of atomics: with the slots filled that is slots * BM * BN atomics per launch whatever
  // the layer (8.4 M for 128 x 128 tiles, ~20 us) -- a third of a 1x1 weight gradient's time but < 10 % of a 3x3's.  Measured:
  // 1x1 layers +15 % with one block per CU (half the splits), 3x3 layers lose up to 20 % (they need the second block to hide
  // the patch staging).
  const int slots = KS == 1 ? (BM * BN >= 128 * 128 ? 256 : 512) : 256 * (BM * BN >= 128 * 128 ? 2 : 3);
  int nsplit = ntile >= slots ? 1 : slots / ntile;
  if (nsplit > a.n_sp_tiles / 2) nsplit = a.n_sp_tiles / 2;
  if (nsplit < 1 || rsis_deterministic()) nsplit = 1;     // deterministic mode: one block walks every spatial tile of its dW tile
  a.tiles_per_split = rsis_cdiv(a.n_sp_tiles, nsplit);
  nsplit = rsis_cdiv(a.n_sp_tiles, a.tiles_per_split);
  hipLaunchKernelGGL((conv_wgrad_tiled_kernel<BM, BN, WGM, WGN, KS, TW, KSP>), dim3(ntile, nsplit), dim3(256), 0, st, a);
  return rsis_check_launch();
}

template <int KS, int TW>
static int launch_tiled_tw(WgradTiledArgs& a, hipStream_t st) {
  // N = Cs * KS * KS columns: a 64-wide tile when that pads N less than the 128-wide one (N mod 128 in 1..64), e.g. the decoder's
  // 16- and 32-channel sources (N = 144 / 288: 56 % / 75 % -> 75 % / 90 % useful MFMA columns)
  const int nmod = (a.Cs * KS * KS) % 128;
  const bool narrow = nmod != 0 && nmod <= 64;
  // (deterministic mode: not the KSP = 2 tile -- its two wave copies both add into dW, in either order)
  if (rsis_deterministic() && a.Cout <= 32) return launch_tiled_cfg<32, 128, 1, 4, KS, TW>(a, st);
  // 1x1: 64 x 64 tiles with two blocks per CU.  The split count -- and with it the dW-sized passes of fp32 atomics, which run at
  // ~0.3 T atomics/s and were a third of these launches -- goes with slots / tiles: a quarter of the 128 x 128 tile's at twice
  // its slots.  Measured on every 1x1 shape of the trunk at batch 32 (tools/exp/bf16_shape_sweep.py --dtype fp32): 47-54 -> 38-44 us.
  if (KS == 1 && a.Cout > 32) return launch_tiled_cfg<64, 64, 2, 2, KS, TW>(a, st);
  if (a.Cout <= 32) return narrow ? launch_tiled_cfg<32, 64, 1, 2, KS, TW, 2>(a, st) : launch_tiled_cfg<32, 128, 1, 4, KS, TW>(a, st);
  if (a.Cout <= 64) return narrow ? launch_tiled_cfg<64, 64, 2, 2, KS, TW>(a, st) : launch_tiled_cfg<64, 128, 2, 2, KS, TW>(a, st);
  return narrow ? launch_tiled_cfg<128, 64, 2, 2, KS, TW>(a, st) : launch_tiled_cfg<128, 128, 2, 2, KS, TW>(a, st);
}

// widest tile the map allows: full 128-byte lines of dy / x per tile row on the wide maps, whole rows on the narrow ones
static int tiled_tw(int H, int W, int ks) {
  const int tp = ks == 1 ? 32 : 64;
  // 3x3: the 32x2 tile's halo patch pushes the 128x128 configuration past 80 KB of LDS (1 block per CU); 16x4 measured equal
  // or better on every trunk / gate shape.  1x1 has no patch and prefers full 128-byte rows.
  for (int tw = ks == 1 ? 32 : 16; tw >= 8; tw >>= 1)
    if (W % tw == 0 && H % (tp / tw) == 0) return tw;
  return 0;
}
// ragged maps (RAG instantiations): the narrowest tile that is at least as wide as the map, or the widest one
static int tiled_tw_ragged(int W, int ks) {
  const int wide = ks == 1 ? 32 : 16;
  return W > 16 ? wide : (W > 8 ? 16 : 8);
}

// ---- grouped launch (host side) ----
// tile configuration of a job, the same rule as launch_tiled_tw: 0 = 32x64 (KSP 2), 1 = 32x128, 2 = 64x64, 3 = 64x128, 4 = 128x64, 5 = 128x128
static int tiled_cfg_code(const WgradTiledArgs& a, int ks) {
  const int nmod = (a.Cs * ks * ks) % 128;
  const bool narrow = nmod != 0 && nmod <= 64;
  if (ks == 1 && a.Cout > 32) return 2;
  if (a.Cout <= 32) return narrow && !rsis_deterministic() ? 0 : 1;
  if (a.Cout <= 64) return narrow ? 2 : 3;
  return narrow ? 4 : 5;
}

template <int BM, int BN, int WGM, int WGN, int KS, int TW, int KSP = 1, int RAG = 0>
static int launch_group_cfg(WgradTiledArgs* jobs, int n, hipStream_t st) {
  constexpr int TH = (KS == 1 ? 32 : 64) / TW;
  long total_iters = 0;
  for (int j = 0; j < n; ++j) {
    WgradTiledArgs& a = jobs[j];
    a.n_co_tiles = rsis_cdiv(a.Cout, BM);
    a.n_n_tiles = rsis_cdiv((long)a.Cs * KS * KS, BN);
    a.n_sp_tiles = RAG ? a.B * rsis_cdiv(a.H, TH) * rsis_cdiv(a.W, TW) : a.B * (a.H / TH) * (a.W / TW);
    total_iters += (long)a.n_co_tiles * a.n_n_tiles * a.n_sp_tiles;
  }
  // equal work per block: every block walks ~L spatial tiles of its job; ~8 blocks per CU over the whole group keeps the tail short,
  // and a job is split as little as that allows (each split is a dW-sized pass of atomics)
  static const int env_tb = getenv("RSIS_WG_GROUP_BLOCKS") ? atoi(getenv("RSIS_WG_GROUP_BLOCKS")) : 0;     // tuning knob
  const long target_blocks = env_tb > 0 ? env_tb : 2048;
  long L = (total_iters + target_blocks - 1) / target_blocks;
  if (L < 2) L = 2;
  if (rsis_deterministic()) L = 1L << 40;            // no split: every dW tile has one contributor
  for (int j0 = 0; j0 < n; j0 += RSIS_WG_MAXJ) {
    WgradTiledGroup g;
    g.n = n - j0 < RSIS_WG_MAXJ ? n - j0 : RSIS_WG_MAXJ;
    int blocks = 0;
    for (int j = 0; j < g.n; ++j) {
      WgradTiledArgs a = jobs[j0 + j];
      int nsplit = rsis_cdiv(a.n_sp_tiles, L);
      if (nsplit < 1) nsplit = 1;
      a.tiles_per_split = rsis_cdiv(a.n_sp_tiles, nsplit);
      nsplit = rsis_cdiv(a.n_sp_tiles, a.tiles_per_split);
      g.begin[j] = blocks;
      g.job[j] = a;
      blocks += a.n_co_tiles * a.n_n_tiles * nsplit;
    }
    g.begin[g.n] = blocks;
    hipLaunchKernelGGL((conv_wgrad_tiled_group_kernel<BM, BN, WGM, WGN, KS, TW, KSP, RAG>), dim3(blocks), dim3(256), 0, st, g);
    if (rsis_check_launch() != RSIS_OK) return RSIS_ERR_LAUNCH;
  }
  return RSIS_OK;
}

template <int KS, int TW, int RAG = 0>
static int launch_group_tw(int code, WgradTiledArgs* jobs, int n, hipStream_t st) {
  switch (code) {
    case 0: return launch_group_cfg<32, 64, 1, 2, KS, TW, 2, RAG>(jobs, n, st);
    case 1: return launch_group_cfg<32, 128, 1, 4, KS, TW, 1, RAG>(jobs, n, st);
    case 2: return launch_group_cfg<64, 64, 2, 2, KS, TW, 1, RAG>(jobs, n, st);
    case 3: return launch_group_cfg<64, 128, 2, 2, KS, TW, 1, RAG>(jobs, n, st);
    case 4: return launch_group_cfg<128, 64, 2, 2, KS, TW, 1, RAG>(jobs, n, st);
    default: return launch_group_cfg<128, 128, 2, 2, KS, TW, 1, RAG>(jobs, n, st);
  }
}

// n weight gradients that rsis_wgrad_tiled_supported accepts, all with the same kernel size: bucketed by (tile width, tile
// configuration), one grouped launch per bucket (per RSIS_WG_MAXJ jobs of a bucket)
int rsis_launch_conv_wgrad_tiled_group(const WgradArgs* w, int n, int ks, hipStream_t st) {
  if (n < 1) return RSIS_OK;
  WgradTiledArgs* all = (WgradTiledArgs*)malloc(sizeof(WgradTiledArgs) * n * 2);
  int* key = (int*)malloc(sizeof(int) * n);
  if (!all || !key) { free(all); free(key); return RSIS_ERR_LAUNCH; }
  WgradTiledArgs* bucket = all + n;
  for (int j = 0; j < n; ++j) {
    WgradTiledArgs a = {};
    a.dy = w[j].dy; a.x = w[j].x; a.dw = w[j].dw; a.B = w[j].B; a.Cs = w[j].Cs; a.H = w[j].H; a.W = w[j].W; a.Cout = w[j].Cout;
    a.ldo = w[j].ldo; a.n_off = w[j].n_off; a.interleave_hid = w[j].interleave_hid;
    all[j] = a;
    const int twa = tiled_tw(a.H, a.W, ks);
    if (twa) key[j] = twa * 8 + tiled_cfg_code(a, ks);
    else {                      // ragged map: 1024 = dword DMA (RAG 1), 2048 = whole dwordx4 groups (RAG 2)
      if (ks == 1 && (a.H * a.W) % 4 == 0) { all[j].W = a.W = a.H * a.W; all[j].H = a.H = 1; }      // a 1x1 walks the flattened map
      key[j] = (a.W % 4 == 0 ? 2048 : 1024) + tiled_tw_ragged(a.W, ks) * 8 + tiled_cfg_code(a, ks);
    }
  }
  int rc = RSIS_OK;
  for (int j = 0; j < n && rc == RSIS_OK; ++j) {
    if (key[j] < 0) continue;
    const int k = key[j];
    int m = 0;
    for (int i = j; i < n; ++i)
      if (key[i] == k) { bucket[m++] = all[i]; key[i] = -1; }
    const int rag = k >> 10;
    const int tw = (k & 1023) / 8, code = k % 8;
    if (rag == 1) {
      if (ks == 1) rc = tw == 32 ? launch_group_tw<1, 32, 1>(code, bucket, m, st) : (tw == 16 ? launch_group_tw<1, 16, 1>(code, bucket, m, st) : launch_group_tw<1, 8, 1>(code, bucket, m, st));
      else rc = tw == 16 ? launch_group_tw<3, 16, 1>(code, bucket, m, st) : launch_group_tw<3, 8, 1>(code, bucket, m, st);
    } else if (rag == 2) {
      if (ks == 1) rc = tw == 32 ? launch_group_tw<1, 32, 2>(code, bucket, m, st) : (tw == 16 ? launch_group_tw<1, 16, 2>(code, bucket, m, st) : launch_group_tw<1, 8, 2>(code, bucket, m, st));
      else rc = tw == 16 ? launch_group_tw<3, 16, 2>(code, bucket, m, st) : launch_group_tw<3, 8, 2>(code, bucket, m, st);
    } else if (ks == 1) rc = tw == 32 ? launch_group_tw<1, 32>(code, bucket, m, st) : (tw == 16 ? launch_group_tw<1, 16>(code, bucket, m, st) : launch_group_tw<1, 8>(code, bucket, m, st));
    else rc = tw == 16 ? launch_group_tw<3, 16>(code, bucket, m, st) : launch_group_tw<3, 8>(code, bucket, m, st);
  }
  free(all); free(key);
  return rc;
}

// true when the LDS-DMA tiled kernel covers this weight gradient (stride 1, "same" padding, tile-aligned map, 32-bit offsets)
bool rsis_wgrad_tiled_supported(const WgradArgs& w, int ks) {
  if (!(ks == 1 || ks == 3) || w.stride != 1 || w.pad != ks / 2 || w.H != w.Ho || w.W != w.Wo) return false;
  static const bool ragged_on = !(getenv("RSIS_WGRAD_RAGGED") && getenv("RSIS_WGRAD_RAGGED")[0] == '0');
  if (tiled_tw(w.H, w.W, ks) == 0 && !ragged_on) return false;      // (ragged maps: the RAG instantiations, grouped launch)
  const long img = (long)w.H * w.W * 4;
  return (long)w.Cout * img < (1L << 30) && (long)w.Cs * img < (1L << 30);
}

int rsis_launch_conv_wgrad_tiled(const WgradArgs& w, int ks, hipStream_t st) {
  WgradTiledArgs a = {};
  a.dy = w.dy; a.x = w.x; a.dw = w.dw; a.B = w.B; a.Cs = w.Cs; a.H = w.H; a.W = w.W; a.Cout = w.Cout;
  a.ldo = w.ldo; a.n_off = w.n_off; a.interleave_hid = w.interleave_hid;
  const int tw = tiled_tw(w.H, w.W, ks);
  if (tw == 0) return rsis_launch_conv_wgrad_tiled_group(&w, 1, ks, st);       // ragged map: the grouped kernel with one job
  if (ks == 1) {
    if (tw == 32) return launch_tiled_tw<1, 32>(a, st);
    if (tw == 16) return launch_tiled_tw<1, 16>(a, st);
    return launch_tiled_tw<1, 8>(a, st);
  }
  if (tw == 16) return launch_tiled_tw<3, 16>(a, st);
  return launch_tiled_tw<3, 8>(a, st);
}
