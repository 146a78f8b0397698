// Weight gradient of the stride-1 "same" convolutions (3x3/p1 and 1x1/p0) with bf16 operands / fp32 accumulation on
// v_mfma_f32_32x32x16_bf16 (the `-dtype bf16` path):
//     dW[co][ci][r][s] += sum_{b,y,x} dy[b][co][y][x] * x[b][ci][y+r-pad][x+s-pad]
// (autograd of nn.Conv2d in reference src/modules/clstm.py:17,44 -- the ConvLSTM gates, time-batched over T*B images --
//  model.py:43-47 and the 1x1 / 3x3 convs of the torchvision bottlenecks).  dy, x and dW are fp32 (NCHW / reference layout);
//  the operands are rounded to bf16 while they are staged into LDS (IN = 1 / 0: rows of whole / ragged float4s).  Channel-blocked bf16
//  operands (conv_blk.hip: cells of 8 channels x 1 pixel) take the DMA / transposing-read kernels further down (wgrad3_tr_body,
//  wgrad1_tr_body): no staging pass at all.
//
// The PIXELS are the reduction axis, and NCHW has them contiguous: a lane's 8 K values are 8 consecutive pixels of one row,
// i.e. one 16-byte LDS cell, for dy (A operand, rows = co) and for x (B operand) alike.
//   * 1x1: a plain GEMM D[co][ci] over BM x BN tiles, both operand tiles staged as [row][64 px] (+16 B row padding: the 32 rows
//     a wave reads land on distinct banks).
//   * 3x3: the lanes of the B operand are 32 INPUT CHANNELS and the 9 taps are 9 accumulator tiles of the wave, so the tap
//     (and with it the 1-pixel shift of the window) is a compile-time constant: per patch row the wave reads the aligned 16-byte
//     window plus one dword on either side and builds the s = 0 / s = 2 windows with 4 v_alignbit_b32 each -- 3 LDS reads and
//     8 VALU per 3 MFMAs.  The input patch (with halo) is staged once per tile, not 9 times.  The 32 x 288 accumulator slab of
//     a wave is transposed through LDS before the fp32 atomics so that consecutive lanes hit consecutive dW addresses.
// Ragged maps (the 7/14/28/56/112-pixel pyramid of 224 x 224 inputs) need no special case: out-of-map pixels are zero-filled
// when the tile is staged.  Split-K over spatial tiles / images with fp32 atomics into dW, as conv_wgrad_tiled.hip.
#include "common.h"

typedef __attribute__((address_space(3))) void* lds_vp_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define RSIS_OOB 0x7FFFFFF0u

struct WgradBf16Args {
  const float* dy;   // [B][CoutDy][H][W]
  const float* x;    // [B][Cs][H][W]
  float* dw;         // [Cout][ldo]
  int B, Cs, H, W, Cout;
  int ldo, n_off;
  short interleave_hid;
  short blk;         // dy and x are channel-blocked bf16 tensors (wgrad3_tr_body / wgrad1_tr_body)
  int n_co_tiles, n_n_tiles, n_sp_tiles, tiles_per_split;
};

__device__ __forceinline__ unsigned wg_pack2(float lo, float hi) {
  const f32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}

// One staging task = 8 consecutive pixels of one row of one channel -> one 16-byte bf16 cell.
// V4: two dwordx4 loads (W % 4 == 0: each is entirely inside or outside the row); otherwise 8 dword loads + a validity mask.
template <bool V4>
struct Task8 {
  float v[8];
  __device__ __forceinline__ void load(const __amdgpu_buffer_rsrc_t r, int off_elems, bool ok_row, int gx, int W) {
    if constexpr (V4) {
      const unsigned o0 = (ok_row && gx >= 0 && gx + 3 < W) ? (unsigned)off_elems * 4u : RSIS_OOB;
      const unsigned o1 = (ok_row && gx + 4 >= 0 && gx + 7 < W) ? (unsigned)(off_elems + 4) * 4u : RSIS_OOB;
      const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, o0, 0, 0));
      const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, o1, 0, 0));
      v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool ok = ok_row && (unsigned)(gx + i) < (unsigned)W;
        v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, ok ? (unsigned)(off_elems + i) * 4u : RSIS_OOB, 0, 0));
      }
    }
  }
  __device__ __forceinline__ u32x4 cell() const {
    u32x4 c;
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = wg_pack2(v[2 * k], v[2 * k + 1]);
    return c;
  }
};

// ------------------------------------------------------------------------------------------------
// 3x3: block = BM dy rows x 32 input channels x 9 taps; wave = 32 rows x 32 channels x 9 taps, the 4 / (BM / 32) wave copies of a
// row group take alternate 16-pixel reduction steps.
// ------------------------------------------------------------------------------------------------
template <int BM, int TW, int IN>
__device__ __forceinline__ void wgrad3_bf16_body(const WgradBf16Args& p, const int bx, const int by) {
#if __HIP_DEVICE_COMPILE__
  constexpr int TP = 64, TH = TP / TW, GPR = TW / 8, NG = TP / 8, KSTEPS = NG / 2;
  constexpr int WGM = BM / 32, KSPW = 4 / WGM, NGG = KSTEPS / KSPW;
  constexpr int PH = TH + 2, XG = GPR + 2, PWP = XG * 8;          // patch: PH rows of XG 8-pixel groups (tile columns start at 8)
  constexpr int ARS = (TP + 8) * 2;                               // bytes per dy row (16 B padding)
  constexpr int CHSB = (PH * PWP * 2 + 31) / 32 * 32 + 16;        // bytes per patch channel: an odd number of 16-byte slots
  constexpr int A_BYTES = BM * ARS, X_BYTES = 32 * CHSB, ST_BYTES = A_BYTES + X_BYTES;
  constexpr int NTA = BM * NG / 256;                              // dy tasks per thread
  constexpr int XT = 32 * PH * XG, NTX = (XT + 255) / 256;        // patch tasks (per thread)
  constexpr int RED_BYTES = 4 * 8 * 288 * 4;                      // epilogue transposition buffer: 8 rows x 288 n per wave
  static_assert(WGM * KSPW == 4 && KSTEPS % KSPW == 0 && (BM * NG) % 256 == 0, "config");
  constexpr int LDS_BYTES = 2 * ST_BYTES > RED_BYTES ? 2 * ST_BYTES : RED_BYTES;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WGM, wk = wave / WGM;
  const int H = p.H, W = p.W, HW = H * W, Cs = p.Cs, Cout = p.Cout;
  const int co_t = bx % p.n_co_tiles, n_t = bx / p.n_co_tiles;
  const int co0 = co_t * BM, ci0 = n_t * 32;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int t_begin = by * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.n_sp_tiles);
  if (t_begin >= t_end) return;

  // ---- loop-invariant parts of the staging tasks ----
  int a_off[NTA], a_y[NTA], a_x[NTA], a_lds[NTA];
#pragma unroll
  for (int i = 0; i < NTA; ++i) {
    const int e = tid + i * 256;
    const int row = e / NG, G = e % NG;
    a_y[i] = G / GPR; a_x[i] = (G % GPR) * 8;
    a_off[i] = co0 + row < Cout ? row * HW + a_y[i] * W + a_x[i] : -1;
    a_lds[i] = row * ARS + G * 16;
  }
  int x_off[NTX], x_y[NTX], x_x[NTX], x_lds[NTX];
#pragma unroll
  for (int i = 0; i < NTX; ++i) {
    const int e = tid + i * 256;
    const int cl = e / (PH * XG), rem = e - cl * (PH * XG);
    const int py = rem / XG, xg = rem - py * XG;
    x_y[i] = py - 1; x_x[i] = (xg - 1) * 8;
    x_off[i] = (e < XT && ci0 + cl < Cs) ? cl * HW + x_y[i] * W + x_x[i] : (int)0x40000000;   // (flag: never valid)
    x_lds[i] = cl * CHSB + (py * PWP + xg * 8) * 2;
  }

  // ---- per-lane LDS read bases (bytes) ----
  const int a_base = (wm * 32 + l31) * ARS + hi * 16;
  // step g covers groups 2g (lanes 0-31) and 2g+1 (lanes 32-63): pixel (y, x8*8) of the tile, patch column 8 + x8*8
  int bo[NGG];
#pragma unroll
  for (int gg = 0; gg < NGG; ++gg) {
    const int G = 2 * (gg * KSPW + wk) + hi;
    bo[gg] = l31 * CHSB + ((G / GPR) * PWP + (G % GPR) * 8 + 8) * 2;
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // scalar tile cursor
  int tb = t_begin / (tiles_x * tiles_y);
  int trem = t_begin - tb * (tiles_x * tiles_y);
  int ty = trem / tiles_x, tx = trem - ty * tiles_x;

  Task8<IN == 1> ra[NTA], rxp[NTX];
#define W3_LOAD()                                                                                                  \
  {                                                                                                                \
    const int y0 = ty * TH, x0 = tx * TW;                                                                          \
    const float* ab = p.dy + ((size_t)tb * Cout + co0) * HW;                                                       \
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc((void*)ab, 0, (Cout - co0) * HW * 4, 0x00020000); \
    const int tsc = y0 * W + x0;                                                                                   \
    _Pragma("unroll") for (int i = 0; i < NTA; ++i)                                                                \
      ra[i].load(ra_, a_off[i] + tsc, a_off[i] >= 0 && y0 + a_y[i] < H, x0 + a_x[i], W);                           \
    const float* xb = p.x + ((size_t)tb * Cs + ci0) * HW;                                                          \
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (Cs - ci0) * HW * 4, 0x00020000); \
    _Pragma("unroll") for (int i = 0; i < NTX; ++i)                                                                \
      rxp[i].load(rx_, x_off[i] + tsc, x_off[i] < 0x20000000 && (unsigned)(y0 + x_y[i]) < (unsigned)H, x0 + x_x[i], W); \
    if (++tx == tiles_x) { tx = 0; if (++ty == tiles_y) { ty = 0; ++tb; } }                                        \
  }
#define W3_STORE(BUF)                                                                                              \
  {                                                                                                                \
    char* st = lds + (BUF) * ST_BYTES;                                                                             \
    _Pragma("unroll") for (int i = 0; i < NTA; ++i) *(u32x4*)(st + a_lds[i]) = ra[i].cell();                       \
    _Pragma("unroll") for (int i = 0; i < NTX; ++i)                                                                \
      if (XT % 256 == 0 || tid + i * 256 < XT) *(u32x4*)(st + A_BYTES + x_lds[i]) = rxp[i].cell();                 \
  }

  W3_LOAD() W3_STORE(0)
  __syncthreads();
  const int ntl = t_end - t_begin;
  for (int t = 0; t < ntl; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntl) W3_LOAD()
    {
      const char* As = lds + cur * ST_BYTES + a_base;
      const char* Xs = lds + cur * ST_BYTES + A_BYTES;
#pragma unroll
      for (int gg = 0; gg < NGG; ++gg) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, *(const u32x4*)(As + (gg * KSPW + wk) * 32));
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const char* row = Xs + bo[gg] + r * PWP * 2;
          const unsigned d0 = *(const unsigned*)(row - 4);
          const u32x4 m = *(const u32x4*)row;
          const unsigned d5 = *(const unsigned*)(row + 16);
          const u32x4 w0 = {__builtin_amdgcn_alignbit(m[0], d0, 16), __builtin_amdgcn_alignbit(m[1], m[0], 16),
                            __builtin_amdgcn_alignbit(m[2], m[1], 16), __builtin_amdgcn_alignbit(m[3], m[2], 16)};
          const u32x4 w2 = {__builtin_amdgcn_alignbit(m[1], m[0], 16), __builtin_amdgcn_alignbit(m[2], m[1], 16),
                            __builtin_amdgcn_alignbit(m[3], m[2], 16), __builtin_amdgcn_alignbit(d5, m[3], 16)};
          acc[r * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, w0), acc[r * 3 + 0], 0, 0, 0);
          acc[r * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, m), acc[r * 3 + 1], 0, 0, 0);
          acc[r * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, w2), acc[r * 3 + 2], 0, 0, 0);
        }
      }
    }
    if (t + 1 < ntl) W3_STORE(cur ^ 1)
    __syncthreads();
  }
#undef W3_LOAD
#undef W3_STORE

  // ---- epilogue: the KSPW wave copies of a row group first sum their partial tiles in LDS (fewer same-address atomics: they are
  // serialised by the L2), 8 rows x 288 columns at a time; the rows are then split over the copies and added to dW with the lanes
  // running along n (consecutive addresses) ----
  float* red = (float*)lds + wm * (8 * 288);
  const int nvalid = min(32, Cs - ci0) * 9;                        // valid columns of this block's 288
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int k = 0; k < KSPW; ++k) {
      if (wk == k) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* d = red + (r + 4 * hi) * 288 + l31 * 9 + t;
            if (k == 0) *d = acc[t][4 * q + r];
            else *d += acc[t][4 * q + r];
          }
      }
      __syncthreads();
    }
#pragma unroll
    for (int rr = wk; rr < 8; rr += KSPW) {
      const int co = co0 + wm * 32 + 8 * q + rr;
      if (co >= Cout) continue;
      const int orow = p.interleave_hid > 0 ? (co & 3) * p.interleave_hid + (co >> 2) : co;
      float* dst = p.dw + (size_t)orow * p.ldo + p.n_off + ci0 * 9;
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const int n = c * 64 + lane;
        if (n < nvalid) atomicAdd(dst + n, red[rr * 288 + n]);
      }
    }
    __syncthreads();
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// 3x3 on blk operands, no staging pass (gfx950: LDS-DMA + ds_read_b64_tr_b16).  The register-staged version of round 3 -- per thread
// and 64-pixel tile 8 x 16-byte loads, an 8 x 8 transposition (~100 VALU) and 8 LDS writes, a full memory round trip per tile behind
// one stage of prefetch -- bound those launches at ~1.4 us per tile and CU whatever the channel counts (a decoder
// level with 32 gate rows over 4 M pixels ran 10x over its HBM time).  Here the blk cells go from HBM to LDS as they are, by DMA
// (1 KB per wave instruction: 16 pixels x the 4 channel blocks of a 32-channel slab, [pixel][cb][8 ch] = 64 bytes per pixel), NR
// tiles deep, and the MFMA operands -- 8 consecutive pixels of one channel per lane -- are gathered by the transposing LDS read:
// within a 16-lane group, result element j of lane l is element l % 4 of the 8-byte chunk lane 4 j + l / 4 pointed at
// (tools/exp/tr16_probe.hip), so lane q of a group points at pixel q / 4, channels 4 (q % 4) .. + 3 of the group's 16 channels: the
// 32 lanes of a K half read 4 pixels x 64 bytes = 256 consecutive bytes (no bank conflict), a second read 4 pixels on completes
// the operand, and the 9 taps are 9 immediate offsets ((r PW + s) 64 bytes) off one per-lane base -- no VALU in the loop at all.
// Tile = TW x TH pixels (TW x TH / 16 K steps, dealt over the 4 / (BM / 32) wave copies of a row group); accumulators, split-K and
// the epilogue are those of wgrad3_bf16_body.
// ------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;
#define RSIS_VMCNT(N) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 15) | (((N) >> 4) << 14))
// The transposing reads are inline asm: behind the intrinsic (__builtin_amdgcn_ds_read_tr16_b64_*) hipcc waits vmcnt(0) before the first
// read of every tile -- for the DMA of the NEXT tile, issued a few instructions earlier -- and the ring degenerates to load-then-compute
// (measured: 2.9 us per 128-pixel tile and block whatever NR; plain ds_read_b64 in the same place gets no such wait).  The asm reads are
// invisible to the waitcnt pass, so their lgkmcnt is counted by hand: LDS returns in order, `s_waitcnt lgkmcnt(N)` = all but the N
// youngest reads have landed (scalar loads in flight can only make the wait longer).  The waits name the registers they guard as
// in/out operands so that no consumer is scheduled above them.
#define TR_READ(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory")
#define TR_WAIT8(N, a0, a1, b0, b1, b2, b3, b4, b5)                                                                \
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5) : "n"(N))
#define TR_WAITN(N) asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N) : "memory")
#define TR_PIN(reg) asm volatile("" : "+v"(reg))        // (orders the consumers of `reg` behind the wait above: volatile asm keeps its order)
__device__ __forceinline__ bf16x8 tr_cat(const s16x4 lo, const s16x4 hi) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int BM, int TW, int TH, int NR>
__device__ __forceinline__ void wgrad3_tr_body(const WgradBf16Args& p, const int bx, const int by) {
#if __HIP_DEVICE_COMPILE__
  constexpr int TP = TW * TH, NK = TP / 16;
  constexpr int WGM = BM / 32, KSPW = 4 / WGM, NGG = NK / KSPW;
  constexpr int PW = TW + 2, PH = TH + 2, XPX = PH * PW;
  constexpr int NA = WGM * NK;                       // DMA instructions of the dy tile (slab-major, 16 pixels each)
  constexpr int NX = (XPX + 15) / 16;                // ... of the input patch (row-major with halo, PW pixels per row)
  constexpr int C_DMA = (NA + NX + 3) / 4;           // per wave and stage (short waves issue all-OOB fillers: one vmcnt rule)
  constexpr int ST_BYTES = 4 * C_DMA * 1024, X_OFF = NA * 1024;
  constexpr int RED_BYTES = 4 * 8 * 288 * 4;
  constexpr int LDS_BYTES = NR * ST_BYTES > RED_BYTES ? NR * ST_BYTES : RED_BYTES;
  static_assert(WGM * KSPW == 4 && NK % KSPW == 0 && TW % 8 == 0 && NR >= 2 && NR <= 5, "config");
  static_assert((NR - 2) * C_DMA < 64 && LDS_BYTES <= 160 * 1024, "vmcnt is 6 bits; LDS of a CU");
  extern __shared__ __attribute__((aligned(1024))) char lds[];      // w3t_lds_bytes<BM, TW, TH, NR>() at launch

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WGM, wk = wave / WGM;
  const int H = p.H, W = p.W, HW = H * W, Cs = p.Cs, Cout = p.Cout;
  const int co_t = bx % p.n_co_tiles, n_t = bx / p.n_co_tiles;
  const int co0 = co_t * BM, ci0 = n_t * 32;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int t_begin = by * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.n_sp_tiles);
  if (t_begin >= t_end) return;

  // ---- loop-invariant part of this lane's share of a stage: instruction d = wave + 4 i moves pixels 16 (d % ..) + lane / 4 ----
  int d_y[C_DMA], d_x[C_DMA], d_cb[C_DMA];           // tile-local pixel (halo: -1), channel-block offset in cells; d_cb < 0: filler
  bool d_a[C_DMA];
#pragma unroll
  for (int i = 0; i < C_DMA; ++i) {
    const int d = wave + 4 * i;
    d_a[i] = d < NA;
    if (d < NA) {
      const int pa = 16 * (d % NK) + (lane >> 2);
      d_y[i] = pa / TW; d_x[i] = pa % TW;
      d_cb[i] = ((d / NK) * 4 + (lane & 3)) * HW;
    } else {
      const int pi = 16 * (d - NA) + (lane >> 2);
      d_y[i] = pi / PW - 1; d_x[i] = pi % PW - 1;
      d_cb[i] = (d < NA + NX && pi < XPX) ? (lane & 3) * HW : -1;
    }
  }
  const size_t a_img = (size_t)(Cout >> 3) * HW * 16, x_img = (size_t)(Cs >> 3) * HW * 16;
  const char* const a_base = (const char*)p.dy + (size_t)(co0 >> 3) * HW * 16;
  const char* const x_base = (const char*)p.x + (size_t)(ci0 >> 3) * HW * 16;
  const int a_len = ((Cout - co0) >> 3) * HW * 16, x_len = ((Cs - ci0) >> 3) * HW * 16;

  // scalar tile cursor (of the NEXT tile to fetch)
  int tb = t_begin / (tiles_x * tiles_y);
  int trem = t_begin - tb * (tiles_x * tiles_y);
  int ty = trem / tiles_x, tx = trem - ty * tiles_x;
#define W3T_ISSUE(SLOT)                                                                                            \
  {                                                                                                                \
    const int y0 = ty * TH, x0 = tx * TW;                                                                          \
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc((void*)(a_base + tb * a_img), 0, a_len, 0x00020000); \
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)(x_base + tb * x_img), 0, x_len, 0x00020000); \
    char* const sd = lds + (SLOT) * ST_BYTES + wave * 1024;                                                        \
    _Pragma("unroll") for (int i = 0; i < C_DMA; ++i) {                                                            \
      const int gy = y0 + d_y[i], gx = x0 + d_x[i];                                                                \
      const bool ok = d_cb[i] >= 0 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;                    \
      const unsigned off = ok ? (unsigned)(d_cb[i] + gy * W + gx) * 16u : RSIS_OOB;                                \
      if (d_a[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (lds_vp_t)(sd + i * 4096), 16, off, 0, 0, 0);      \
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_, (lds_vp_t)(sd + i * 4096), 16, off, 0, 0, 0);            \
    }                                                                                                              \
    if (++tx == tiles_x) { tx = 0; if (++ty == tiles_y) { ty = 0; ++tb; } }                                        \
  }

  // ---- per-lane operand bases (bytes inside a stage) ----
  const int q16 = lane & 15, grp = lane >> 4;
  const int lane_c = (grp & 1) * 32 + (q16 & 3) * 8;               // byte of this lane's 4 channels inside the 64-byte pixel
  int aoff[NGG], xoff[NGG];
#pragma unroll
  for (int gg = 0; gg < NGG; ++gg) {
    const int pa = 16 * (gg * KSPW + wk) + 8 * hi + (q16 >> 2);
    aoff[gg] = wm * NK * 1024 + pa * 64 + lane_c;
    xoff[gg] = X_OFF + ((pa / TW) * PW + pa % TW) * 64 + lane_c;
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  const int ntl = t_end - t_begin;
#pragma unroll
  for (int i = 0; i < NR - 1; ++i)
    if (i < ntl) W3T_ISSUE(i)
  for (int t = 0; t < ntl; ++t) {
    {   // tile t has landed once at most `ahead` younger tiles of this wave's DMA are still in flight
      const int ahead = min(NR - 2, ntl - 1 - t);
      if (NR >= 5 && ahead == 3) { RSIS_VMCNT(3 * C_DMA); }
      else if (NR >= 4 && ahead == 2) { RSIS_VMCNT(2 * C_DMA); }
      else if (NR >= 3 && ahead == 1) { RSIS_VMCNT(C_DMA); }
      else { RSIS_VMCNT(0); }
    }
    __builtin_amdgcn_s_barrier();                     // tile t is in LDS (every wave's share); slot (t - 1) % NR is free
    if (t + NR - 1 < ntl) W3T_ISSUE((t + NR - 1) % NR)
    // K steps of this wave, software-pipelined over the LDS queue: the reads of (step, patch row r) are re-issued for the next step
    // right behind the 3 MFMAs that consumed row r, so that 14 reads (two rows + the next dy operand) are in flight under them
    const unsigned sb = lds0 + (t % NR) * ST_BYTES;
    s16x4 A[2][2], X[3][3][2];
#define W3T_READ_A(GG) { const unsigned aa = sb + aoff[GG]; TR_READ(A[(GG) & 1][0], aa, 0); TR_READ(A[(GG) & 1][1], aa, 256); }
#define W3T_READ_ROW(GG, R)                                                                                        \
  {                                                                                                                \
    const unsigned xa = sb + xoff[GG];                                                                             \
    TR_READ(X[R][0][0], xa, ((R) * PW + 0) * 64); TR_READ(X[R][0][1], xa, ((R) * PW + 0) * 64 + 256);              \
    TR_READ(X[R][1][0], xa, ((R) * PW + 1) * 64); TR_READ(X[R][1][1], xa, ((R) * PW + 1) * 64 + 256);              \
    TR_READ(X[R][2][0], xa, ((R) * PW + 2) * 64); TR_READ(X[R][2][1], xa, ((R) * PW + 2) * 64 + 256);              \
  }
#define W3T_ROW(GG, R, N)                                                                                          \
  {                                                                                                                \
    TR_WAIT8(N, A[(GG) & 1][0], A[(GG) & 1][1], X[R][0][0], X[R][0][1], X[R][1][0], X[R][1][1], X[R][2][0], X[R][2][1]); \
    const bf16x8 a = tr_cat(A[(GG) & 1][0], A[(GG) & 1][1]);                                                       \
    acc[(R) * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, tr_cat(X[R][0][0], X[R][0][1]), acc[(R) * 3 + 0], 0, 0, 0); \
    acc[(R) * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, tr_cat(X[R][1][0], X[R][1][1]), acc[(R) * 3 + 1], 0, 0, 0); \
    acc[(R) * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, tr_cat(X[R][2][0], X[R][2][1]), acc[(R) * 3 + 2], 0, 0, 0); \
  }
    W3T_READ_A(0) W3T_READ_ROW(0, 0) W3T_READ_ROW(0, 1) W3T_READ_ROW(0, 2)
#pragma unroll
    for (int gg = 0; gg < NGG; ++gg) {
      if (gg + 1 < NGG) {
        W3T_ROW(gg, 0, 12) W3T_READ_A(gg + 1) W3T_READ_ROW(gg + 1, 0)
        W3T_ROW(gg, 1, 14) W3T_READ_ROW(gg + 1, 1)
        W3T_ROW(gg, 2, 14) W3T_READ_ROW(gg + 1, 2)
      } else {
        W3T_ROW(gg, 0, 12) W3T_ROW(gg, 1, 6) W3T_ROW(gg, 2, 0)
      }
    }
#undef W3T_READ_A
#undef W3T_READ_ROW
#undef W3T_ROW
  }
#undef W3T_ISSUE
  __syncthreads();

  // ---- epilogue (as wgrad3_bf16_body) ----
  float* red = (float*)lds + wm * (8 * 288);
  const int nvalid = min(32, Cs - ci0) * 9;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int k = 0; k < KSPW; ++k) {
      if (wk == k) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* d = red + (r + 4 * hi) * 288 + l31 * 9 + t;
            if (k == 0) *d = acc[t][4 * q + r];
            else *d += acc[t][4 * q + r];
          }
      }
      __syncthreads();
    }
#pragma unroll
    for (int rr = wk; rr < 8; rr += KSPW) {
      const int co = co0 + wm * 32 + 8 * q + rr;
      if (co >= Cout) continue;
      const int orow = p.interleave_hid > 0 ? (co & 3) * p.interleave_hid + (co >> 2) : co;
      float* dst = p.dw + (size_t)orow * p.ldo + p.n_off + ci0 * 9;
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const int n = c * 64 + lane;
        if (n < nvalid) atomicAdd(dst + n, red[rr * 288 + n]);
      }
    }
    __syncthreads();
  }
#endif
}
// ------------------------------------------------------------------------------------------------
// 1x1 on blk operands, the same way (DMA ring of raw cells, transposing reads): D[co][ci] over BM x BN with 2 x 2 waves, a stage =
// TP consecutive pixels of the flattened map for the BM / 32 + BN / 32 channel slabs of the two operands; every wave walks all TP / 16
// K steps, the reads of step g + 1 in flight under the MFMAs of step g (two register sets, lgkmcnt counted by hand).
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int TP, int NR>
__device__ __forceinline__ void wgrad1_tr_body(const WgradBf16Args& p, const int bx, const int by) {
#if __HIP_DEVICE_COMPILE__
  constexpr int WGM = 2, WGN = 2, TM = BM / WGM / 32, TN = BN / WGN / 32, NK = TP / 16;
  constexpr int SA = BM / 32, SB = BN / 32;            // channel slabs of the dy / x tiles
  constexpr int ND = (SA + SB) * NK;                   // DMA instructions per stage (16 pixels x 4 channel blocks each)
  constexpr int C_DMA = (ND + 3) / 4;
  constexpr int ST_BYTES = 4 * C_DMA * 1024, B_OFF = SA * NK * 1024;
  constexpr int NRD = 2 * (TM + TN);                   // LDS reads per K step
  static_assert(TM >= 1 && TN >= 1 && NR >= 2 && NR <= 4 && (NR - 2) * C_DMA < 64 && NRD <= 15 && NR * ST_BYTES <= 160 * 1024, "config");
  extern __shared__ __attribute__((aligned(1024))) char lds[];      // NR * ST_BYTES at launch

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int HW = p.W, Cs = p.Cs, Cout = p.Cout;        // (1x1: H = 1, W = the flattened map)
  const int co_t = bx % p.n_co_tiles, n_t = bx / p.n_co_tiles;
  const int co0 = co_t * BM, n0 = n_t * BN;
  const int tiles_x = (HW + TP - 1) / TP;
  const int t_begin = by * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.n_sp_tiles);
  if (t_begin >= t_end) return;

  int d_px[C_DMA], d_cb[C_DMA];                        // pixel inside the tile; channel-block offset in cells (< 0: filler)
  bool d_a[C_DMA];
#pragma unroll
  for (int i = 0; i < C_DMA; ++i) {
    const int d = wave + 4 * i;
    d_a[i] = d < SA * NK;
    const int slab = d_a[i] ? d / NK : (d - SA * NK) / NK;
    d_px[i] = 16 * (d % NK) + (lane >> 2);
    d_cb[i] = d < ND ? (slab * 4 + (lane & 3)) * HW : -1;
  }
  const size_t a_img = (size_t)(Cout >> 3) * HW * 16, x_img = (size_t)(Cs >> 3) * HW * 16;
  const char* const a_base = (const char*)p.dy + (size_t)(co0 >> 3) * HW * 16;
  const char* const x_base = (const char*)p.x + (size_t)(n0 >> 3) * HW * 16;
  const int a_len = ((Cout - co0) >> 3) * HW * 16, x_len = ((Cs - n0) >> 3) * HW * 16;

  int tb = t_begin / tiles_x, tx = t_begin - tb * tiles_x;
#define W1T_ISSUE(SLOT)                                                                                            \
  {                                                                                                                \
    const int x0 = tx * TP;                                                                                        \
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc((void*)(a_base + tb * a_img), 0, a_len, 0x00020000); \
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)(x_base + tb * x_img), 0, x_len, 0x00020000); \
    char* const sd = lds + (SLOT) * ST_BYTES + wave * 1024;                                                        \
    _Pragma("unroll") for (int i = 0; i < C_DMA; ++i) {                                                            \
      const int gx = x0 + d_px[i];                                                                                 \
      const unsigned off = (d_cb[i] >= 0 && gx < HW) ? (unsigned)(d_cb[i] + gx) * 16u : RSIS_OOB;                  \
      if (d_a[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (lds_vp_t)(sd + i * 4096), 16, off, 0, 0, 0);      \
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_, (lds_vp_t)(sd + i * 4096), 16, off, 0, 0, 0);            \
    }                                                                                                              \
    if (++tx == tiles_x) { tx = 0; ++tb; }                                                                         \
  }

  const int q16 = lane & 15, grp = lane >> 4;
  const unsigned lane_o = (8 * hi + (q16 >> 2)) * 64 + (grp & 1) * 32 + (q16 & 3) * 8;     // K half, pixel in the quad, channel quad
  const unsigned a_lane = wm * TM * NK * 1024 + lane_o, b_lane = B_OFF + wn * TN * NK * 1024 + lane_o;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  const int ntl = t_end - t_begin;
#pragma unroll
  for (int i = 0; i < NR - 1; ++i)
    if (i < ntl) W1T_ISSUE(i)
  for (int t = 0; t < ntl; ++t) {
    {
      const int ahead = min(NR - 2, ntl - 1 - t);
      if (NR >= 4 && ahead == 2) { RSIS_VMCNT(2 * C_DMA); }
      else if (NR >= 3 && ahead == 1) { RSIS_VMCNT(C_DMA); }
      else { RSIS_VMCNT(0); }
    }
    __builtin_amdgcn_s_barrier();
    if (t + NR - 1 < ntl) W1T_ISSUE((t + NR - 1) % NR)
    const unsigned sa = lds0 + (t % NR) * ST_BYTES + a_lane, sbb = lds0 + (t % NR) * ST_BYTES + b_lane;
    s16x4 A[2][TM][2], B[2][TN][2];
#define W1T_READ(G)                                                                                                \
  {                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                               \
      const unsigned aa = sa + i * NK * 1024;                                                                      \
      TR_READ(A[(G) & 1][i][0], aa, (G) * 1024); TR_READ(A[(G) & 1][i][1], aa, (G) * 1024 + 256);                  \
    }                                                                                                              \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                               \
      const unsigned ba = sbb + j * NK * 1024;                                                                     \
      TR_READ(B[(G) & 1][j][0], ba, (G) * 1024); TR_READ(B[(G) & 1][j][1], ba, (G) * 1024 + 256);                  \
    }                                                                                                              \
  }
    W1T_READ(0)
#pragma unroll
    for (int g = 0; g < NK; ++g) {
      if (g + 1 < NK) {
        if (g == 0) W1T_READ(1) else if (g == 1) W1T_READ(2) else if (g == 2) W1T_READ(3) else if (g == 3) W1T_READ(4)
        else if (g == 4) W1T_READ(5) else if (g == 5) W1T_READ(6) else W1T_READ(7)
      }
      // the reads of step g have landed once at most the NRD reads of step g + 1 are outstanding
      if (g + 1 < NK) { TR_WAITN(NRD); } else { TR_WAITN(0); }
#pragma unroll
      for (int i = 0; i < TM; ++i) { TR_PIN(A[g & 1][i][0]); TR_PIN(A[g & 1][i][1]); }
#pragma unroll
      for (int j = 0; j < TN; ++j) { TR_PIN(B[g & 1][j][0]); TR_PIN(B[g & 1][j][1]); }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_cat(A[g & 1][i][0], A[g & 1][i][1]), tr_cat(B[g & 1][j][0], B[g & 1][j][1]), acc[i][j], 0, 0, 0);
    }
#undef W1T_READ
  }
#undef W1T_ISSUE

  // (buffer atomics, as wgrad1_bf16_body)
  const __amdgpu_buffer_rsrc_t rdw = __builtin_amdgcn_make_buffer_rsrc((void*)p.dw, 0, (unsigned)((size_t)Cout * p.ldo * 4), 0x00020000);
  const int ihid = p.interleave_hid;
  const unsigned ldb = (unsigned)p.ldo * 4u;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * TN * 32 + j * 32 + l31;
    if (n >= Cs) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row0 = co0 + wm * TM * 32 + i * 32 + 4 * hi;
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
template <int BM, int BN, int TP, int NR>
constexpr int w1t_lds_bytes() { return NR * 4 * (((BM / 32 + BN / 32) * (TP / 16) + 3) / 4) * 1024; }
#ifndef W1T_TP
#define W1T_TP 64
#endif
constexpr int w1t_nr(int bm, int bn) { return (bm + bn) >= 256 ? 2 : 3; }

template <int BM, int TW, int TH, int NR>
constexpr int w3t_lds_bytes() {
  constexpr int st = 4 * ((BM / 32 * (TW * TH / 16) + ((TH + 2) * (TW + 2) + 15) / 16 + 3) / 4) * 1024;
  return NR * st > 4 * 8 * 288 * 4 ? NR * st : 4 * 8 * 288 * 4;
}
// tile height / ring depth of the (BM, TW) instantiations: the deepest tile whose ring fits 64 KB of static LDS, two blocks per CU
#ifndef W3T_TH_32_32      // (tuning builds override the three configurations that carry the step: -DW3T_TH_32_32=.. -DW3T_NR_32_32=.. ...)
#define W3T_TH_32_32 4
#define W3T_NR_32_32 2
#endif
#ifndef W3T_TH_64_32
#define W3T_TH_64_32 4
#define W3T_NR_64_32 2
#endif
#ifndef W3T_TH_64_16
#define W3T_TH_64_16 8
#define W3T_NR_64_16 2
#endif
constexpr int w3t_th(int bm, int tw) {
  return tw == 32 ? (bm == 128 ? 2 : (bm == 64 ? W3T_TH_64_32 : W3T_TH_32_32)) : (tw == 16 ? (bm == 128 ? 4 : (bm == 64 ? W3T_TH_64_16 : 8)) : 8);
}
constexpr int w3t_nr(int bm, int tw) {
  return tw == 32 ? (bm == 128 ? 2 : (bm == 64 ? W3T_NR_64_32 : W3T_NR_32_32)) : (tw == 16 ? (bm == 32 ? 3 : (bm == 64 ? W3T_NR_64_16 : 2)) : (bm == 128 ? 2 : 3));
}

// ------------------------------------------------------------------------------------------------
// 1x1: D[co][ci] = sum_px dy[co][px] x[ci][px]; block tile BM x BN, waves WGM x WGN, TM x TN MFMA tiles per wave.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WGM, int WGN, int TW, int IN>
__device__ __forceinline__ void wgrad1_bf16_body(const WgradBf16Args& p, const int bx, const int by) {
#if __HIP_DEVICE_COMPILE__
  constexpr int TP = 64, TH = TP / TW, GPR = TW / 8, NG = TP / 8, KSTEPS = NG / 2;
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  constexpr int ARS = (TP + 8) * 2;
  constexpr int A_BYTES = BM * ARS, B_BYTES = BN * ARS, ST_BYTES = A_BYTES + B_BYTES;
  constexpr int NTA = BM * NG / 256, NTB = BN * NG / 256;
  static_assert(WGM * WGN == 4 && (BM * NG) % 256 == 0 && (BN * NG) % 256 == 0, "config");
  __shared__ __attribute__((aligned(16))) char lds[2 * ST_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int H = p.H, W = p.W, HW = H * W, Cs = p.Cs, Cout = p.Cout;
  const int co_t = bx % p.n_co_tiles, n_t = bx / p.n_co_tiles;
  const int co0 = co_t * BM, n0 = n_t * BN;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int t_begin = by * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.n_sp_tiles);
  if (t_begin >= t_end) return;

  int a_off[NTA], a_y[NTA], a_x[NTA], a_lds[NTA];
#pragma unroll
  for (int i = 0; i < NTA; ++i) {
    const int e = tid + i * 256;
    const int row = e / NG, G = e % NG;
    a_y[i] = G / GPR; a_x[i] = (G % GPR) * 8;
    a_off[i] = co0 + row < Cout ? row * HW + a_y[i] * W + a_x[i] : -1;
    a_lds[i] = row * ARS + G * 16;
  }
  int b_off[NTB], b_y[NTB], b_x[NTB], b_lds[NTB];
#pragma unroll
  for (int i = 0; i < NTB; ++i) {
    const int e = tid + i * 256;
    const int row = e / NG, G = e % NG;
    b_y[i] = G / GPR; b_x[i] = (G % GPR) * 8;
    b_off[i] = n0 + row < Cs ? row * HW + b_y[i] * W + b_x[i] : -1;
    b_lds[i] = row * ARS + G * 16;
  }
  const int a_base = (wm * TM * 32 + l31) * ARS + hi * 16;
  const int b_base = (wn * TN * 32 + l31) * ARS + hi * 16;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int tb = t_begin / (tiles_x * tiles_y);
  int trem = t_begin - tb * (tiles_x * tiles_y);
  int ty = trem / tiles_x, tx = trem - ty * tiles_x;

  Task8<IN == 1> ra[NTA], rb[NTB];
#define W1_LOAD()                                                                                                  \
  {                                                                                                                \
    const int y0 = ty * TH, x0 = tx * TW;                                                                          \
    const int tsc = y0 * W + x0;                                                                                   \
    const float* ab = p.dy + ((size_t)tb * Cout + co0) * HW;                                                       \
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc((void*)ab, 0, (Cout - co0) * HW * 4, 0x00020000); \
    _Pragma("unroll") for (int i = 0; i < NTA; ++i)                                                                \
      ra[i].load(ra_, a_off[i] + tsc, a_off[i] >= 0 && y0 + a_y[i] < H, x0 + a_x[i], W);                           \
    const float* xb = p.x + ((size_t)tb * Cs + n0) * HW;                                                           \
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (Cs - n0) * HW * 4, 0x00020000); \
    _Pragma("unroll") for (int i = 0; i < NTB; ++i)                                                                \
      rb[i].load(rb_, b_off[i] + tsc, b_off[i] >= 0 && y0 + b_y[i] < H, x0 + b_x[i], W);                           \
    if (++tx == tiles_x) { tx = 0; if (++ty == tiles_y) { ty = 0; ++tb; } }                                        \
  }
#define W1_STORE(BUF)                                                                                              \
  {                                                                                                                \
    char* st = lds + (BUF) * ST_BYTES;                                                                             \
    _Pragma("unroll") for (int i = 0; i < NTA; ++i) *(u32x4*)(st + a_lds[i]) = ra[i].cell();                       \
    _Pragma("unroll") for (int i = 0; i < NTB; ++i) *(u32x4*)(st + A_BYTES + b_lds[i]) = rb[i].cell();             \
  }

  W1_LOAD() W1_STORE(0)
  __syncthreads();
  const int ntl = t_end - t_begin;
  for (int t = 0; t < ntl; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntl) W1_LOAD()
    {
      const char* As = lds + cur * ST_BYTES + a_base;
      const char* Bs = lds + cur * ST_BYTES + A_BYTES + b_base;
#pragma unroll
      for (int g = 0; g < KSTEPS; ++g) {
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = __builtin_bit_cast(bf16x8, *(const u32x4*)(As + i * 32 * ARS + g * 32));
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = __builtin_bit_cast(bf16x8, *(const u32x4*)(Bs + j * 32 * ARS + g * 32));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (t + 1 < ntl) W1_STORE(cur ^ 1)
    __syncthreads();
  }
#undef W1_LOAD
#undef W1_STORE

  // (buffer atomics, as in conv_wgrad_tiled.hip)
  const __amdgpu_buffer_rsrc_t rdw = __builtin_amdgcn_make_buffer_rsrc((void*)p.dw, 0, (unsigned)((size_t)Cout * p.ldo * 4), 0x00020000);
  const int ihid = p.interleave_hid;
  const unsigned ldb = (unsigned)p.ldo * 4u;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * TN * 32 + j * 32 + l31;
    if (n >= Cs) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row0 = co0 + wm * TM * 32 + i * 32 + 4 * hi;
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

// ------------------------------------------------------------------------------------------------
// (the 3x3 kernels are compiled for TWO waves per SIMD -- `__launch_bounds__(256, 2)`: 9 accumulator tiles + staging registers came
//  to 150-190 VGPRs + 144 AGPRs, one wave per SIMD, and a loop whose every s_waitcnt / barrier stalls the whole SIMD; held to 256
//  registers (no AGPRs, at most 9 spilled in three fp32-operand variants) two blocks share a CU and the bf16 224^2 step went
//  19.66 -> 18.80 ms.  LDS allows two blocks in every variant.)
// single launches, and grouped launches: the weight gradients of many layers in one grid (rsis_conv2d_wgrad_batch; see
// conv_wgrad_tiled.hip).  The jobs travel by value in the kernel arguments.
template <int BM, int TW, int IN>
__global__ __launch_bounds__(256, 2) void wgrad3_bf16_kernel(const WgradBf16Args p) { wgrad3_bf16_body<BM, TW, IN>(p, blockIdx.x, blockIdx.y); }
template <int BM, int TW, int TH, int NR>
__global__ __launch_bounds__(256, 2) void wgrad3_tr_kernel(const WgradBf16Args p) {
  wgrad3_tr_body<BM, TW, TH, NR>(p, blockIdx.x, blockIdx.y);
}
template <int BM, int BN, int WGM, int WGN, int TW, int IN>
__global__ __launch_bounds__(256) void wgrad1_bf16_kernel(const WgradBf16Args p) {
  wgrad1_bf16_body<BM, BN, WGM, WGN, TW, IN>(p, blockIdx.x, blockIdx.y);
}

// XCD placement of a grouped launch.  A (job, range of consecutive splits) ITEM -- all dW tiles of one job over one range of its
// spatial tiles -- runs on ONE XCD: its blocks re-read the same dy / x pixels once per dW tile (4-16 times each), and only inside
// one L2 are those re-reads hits.  With consecutive block indices spread round-robin over the eight XCDs every XCD fetched the pixels
// again: the bf16 1x1 group read 6.1 GB where 1.5 GB are needed and ran AT the fabric's rate (6 TB/s; rocprofv3 --pmc FETCH_SIZE per
// kernel, tools/step_traffic.py) -- with the MFMA 16x faster than in fp32 these launches are what the memory system lets them be.
// Hardware sends block b to XCD b % 8, so the grid is eight interleaved LANES: lane x is the blocks b = 8 k + x and holds a list of
// items (k ranges); the host balances the lanes by work (longest item first).  (The exact-f32 weight gradients are MFMA-bound: the
// same placement halved their reads and changed their time by nothing, NOTES.md (21).)
#define RSIS_WGB_MAXJ 32
#define RSIS_WGB_LANE_ITEMS 28
struct WgradBf16Group {
  int n;
  int lane_n[8];
  int lane_start[8][RSIS_WGB_LANE_ITEMS + 1];     // first k of each item of a lane, ascending; [lane_n] = the lane's block count
  unsigned short lane_split0[8][RSIS_WGB_LANE_ITEMS];   // an item covers the consecutive splits split0, split0 + 1, ... of its job
  unsigned char lane_job[8][RSIS_WGB_LANE_ITEMS];
  WgradBf16Args job[RSIS_WGB_MAXJ];
};
static_assert(sizeof(WgradBf16Group) <= 4000, "kernel arguments are limited to 4 KB");

// block -> (job, tile, split); false: a padding block of a short lane
__device__ __forceinline__ bool wgb_find(const WgradBf16Group& g, int& job, int& tile, int& split) {
  const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int nl = g.lane_n[x];
  if (k >= g.lane_start[x][nl]) return false;
  int i = 0;
  while (i + 1 < nl && g.lane_start[x][i + 1] <= k) ++i;        // (uniform: scalar code)
  job = g.lane_job[x][i];
  const int kl = k - g.lane_start[x][i], ntile = g.job[job].n_co_tiles * g.job[job].n_n_tiles;
  tile = kl % ntile;
  split = g.lane_split0[x][i] + kl / ntile;
  return true;
}
template <int BM, int TW, int IN>
__global__ __launch_bounds__(256, 2) void wgrad3_bf16_group_kernel(const WgradBf16Group g) {
  int j, tile, split;
  if (!wgb_find(g, j, tile, split)) return;
  wgrad3_bf16_body<BM, TW, IN>(g.job[j], tile, split);
}
template <int BM, int TW, int TH, int NR>
__global__ __launch_bounds__(256, 2) void wgrad3_tr_group_kernel(const WgradBf16Group g) {
  int j, tile, split;
  if (!wgb_find(g, j, tile, split)) return;
  wgrad3_tr_body<BM, TW, TH, NR>(g.job[j], tile, split);
}
// the ring is dynamic LDS (up to 80 KB: two blocks per CU); sizes above 64 KB are an opt-in per kernel
template <int BM, int TW, int TH, int NR>
static void w3t_launch(const WgradBf16Args& a, dim3 grid, hipStream_t st) {
  constexpr int lds = w3t_lds_bytes<BM, TW, TH, NR>();
  static const hipError_t once = hipFuncSetAttribute((const void*)wgrad3_tr_kernel<BM, TW, TH, NR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)once;
  hipLaunchKernelGGL((wgrad3_tr_kernel<BM, TW, TH, NR>), grid, dim3(256), lds, st, a);
}
template <int BM, int TW, int TH, int NR>
static void w3t_launch_group(const WgradBf16Group& g, int blocks, hipStream_t st) {
  constexpr int lds = w3t_lds_bytes<BM, TW, TH, NR>();
  static const hipError_t once = hipFuncSetAttribute((const void*)wgrad3_tr_group_kernel<BM, TW, TH, NR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)once;
  hipLaunchKernelGGL((wgrad3_tr_group_kernel<BM, TW, TH, NR>), dim3(blocks), dim3(256), lds, st, g);
}
template <int BM, int BN, int TP, int NR>
__global__ __launch_bounds__(256, 2) void wgrad1_tr_kernel(const WgradBf16Args p) { wgrad1_tr_body<BM, BN, TP, NR>(p, blockIdx.x, blockIdx.y); }
template <int BM, int BN, int TP, int NR>
__global__ __launch_bounds__(256, 2) void wgrad1_tr_group_kernel(const WgradBf16Group g) {
  int j, tile, split;
  if (!wgb_find(g, j, tile, split)) return;
  wgrad1_tr_body<BM, BN, TP, NR>(g.job[j], tile, split);
}
template <int BM, int BN, int TP, int NR>
static void w1t_launch(const WgradBf16Args& a, dim3 grid, hipStream_t st) {
  constexpr int lds = w1t_lds_bytes<BM, BN, TP, NR>();
  static const hipError_t once = hipFuncSetAttribute((const void*)wgrad1_tr_kernel<BM, BN, TP, NR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)once;
  hipLaunchKernelGGL((wgrad1_tr_kernel<BM, BN, TP, NR>), grid, dim3(256), lds, st, a);
}
template <int BM, int BN, int TP, int NR>
static void w1t_launch_group(const WgradBf16Group& g, int blocks, hipStream_t st) {
  constexpr int lds = w1t_lds_bytes<BM, BN, TP, NR>();
  static const hipError_t once = hipFuncSetAttribute((const void*)wgrad1_tr_group_kernel<BM, BN, TP, NR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)once;
  hipLaunchKernelGGL((wgrad1_tr_group_kernel<BM, BN, TP, NR>), dim3(blocks), dim3(256), lds, st, g);
}
#ifdef RSIS_W3T_SWEEP      // tuning build: tile height / ring depth of the single launches from the environment (tools/exp/wgrad_blk_bench.py)
template <int BM, int TW, int TH, int NR>
static int w3t_try(WgradBf16Args& a, int ntile, hipStream_t st) {
  constexpr int st_b = w3t_lds_bytes<BM, TW, TH, NR>() / NR;
  if constexpr (w3t_lds_bytes<BM, TW, TH, NR>() <= 81920 && (NR - 2) * (st_b / 4096) < 64 && (TW * TH / 16) % (4 / (BM / 32)) == 0) {
    void split_plan(WgradBf16Args&, int, int, int, int);
    split_plan(a, TW, TH, ntile, 256);
    w3t_launch<BM, TW, TH, NR>(a, dim3(ntile, rsis_cdiv(a.n_sp_tiles, a.tiles_per_split)), st);
    return rsis_check_launch();
  }
  return RSIS_ERR_UNSUPPORTED;
}
template <int BM, int TW, int TH>
static int w3t_sweep_nr(WgradBf16Args& a, int ntile, int nr, hipStream_t st) {
  if (nr == 2) return w3t_try<BM, TW, TH, 2>(a, ntile, st);
  if (nr == 3) return w3t_try<BM, TW, TH, 3>(a, ntile, st);
  if (nr == 4) return w3t_try<BM, TW, TH, 4>(a, ntile, st);
  if (nr == 5) return w3t_try<BM, TW, TH, 5>(a, ntile, st);
  return RSIS_ERR_UNSUPPORTED;
}
template <int BM, int TW>
static int w3t_sweep(WgradBf16Args& a, int ntile, hipStream_t st) {
  const int th = atoi(getenv("RSIS_W3T_TH")), nr = atoi(getenv("RSIS_W3T_NR"));
  if (th == 2) return w3t_sweep_nr<BM, TW, 2>(a, ntile, nr, st);
  if (th == 4) return w3t_sweep_nr<BM, TW, 4>(a, ntile, nr, st);
  if (th == 8) return w3t_sweep_nr<BM, TW, 8>(a, ntile, nr, st);
  if (th == 16) return w3t_sweep_nr<BM, TW, 16>(a, ntile, nr, st);
  return RSIS_ERR_UNSUPPORTED;
}
#endif
template <int BM, int BN, int WGM, int WGN, int TW, int IN>
__global__ __launch_bounds__(256) void wgrad1_bf16_group_kernel(const WgradBf16Group g) {
  int j, tile, split;
  if (!wgb_find(g, j, tile, split)) return;
  wgrad1_bf16_body<BM, BN, WGM, WGN, TW, IN>(g.job[j], tile, split);
}

void split_plan(WgradBf16Args& a, int TW, int TH, int ntile, int slots) {
  a.n_sp_tiles = a.B * rsis_cdiv(a.H, TH) * rsis_cdiv(a.W, TW);
  // Every split adds a dW-sized pass of fp32 atomics, and those run at ~0.3 T atomics/s whatever the layer: measured on the
  // trunk shapes at batch 32 (tools/exp/bf16_shape_sweep.py), one block per CU (256 slots) beats two (512) on every layer --
  // 4.4 vs 5.7 ms per step -- although the main loop alone is no faster (2.7 vs 2.6 ms): the second block only buys atomics.
  int nsplit = ntile >= slots ? 1 : slots / ntile;
  if (nsplit > 256) nsplit = 256;
  if (nsplit > a.n_sp_tiles / 2) nsplit = a.n_sp_tiles / 2;
  if (nsplit < 1 || rsis_deterministic()) nsplit = 1;
  a.tiles_per_split = rsis_cdiv(a.n_sp_tiles, nsplit);
}

template <int BM, int TW>
static int launch_w3(WgradBf16Args& a, hipStream_t st) {
  a.n_co_tiles = rsis_cdiv(a.Cout, BM);
  a.n_n_tiles = rsis_cdiv(a.Cs, 32);
  const int ntile = a.n_co_tiles * a.n_n_tiles;
#ifdef RSIS_W3T_SWEEP
  if (a.blk && getenv("RSIS_W3T_TH")) return w3t_sweep<BM, TW>(a, ntile, st);
#endif
  split_plan(a, TW, a.blk ? w3t_th(BM, TW) : 64 / TW, ntile, 256);
  const dim3 grid(ntile, rsis_cdiv(a.n_sp_tiles, a.tiles_per_split));
  if (a.blk) w3t_launch<BM, TW, w3t_th(BM, TW), w3t_nr(BM, TW)>(a, grid, st);
  else if (a.W % 4 == 0) hipLaunchKernelGGL((wgrad3_bf16_kernel<BM, TW, 1>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((wgrad3_bf16_kernel<BM, TW, 0>), grid, dim3(256), 0, st, a);
  return rsis_check_launch();
}

template <int TW>
static int launch_w3_tw(WgradBf16Args& a, hipStream_t st) {
  // 64 dy rows per block unless the patch side is deep: half the splits (= half the atomics) of the 128-row tile for 20 % more
  // operand reads (128 -> 128 @28^2: 51 -> 36 us, 256 -> 256 @14^2: 62 -> 59 us; 1024 -> 128 @14^2: 92 -> 101 us, kept at 128)
  if (a.Cout <= 32) return launch_w3<32, TW>(a, st);
  if (a.Cout <= 64 || a.Cs <= 512) return launch_w3<64, TW>(a, st);
  return launch_w3<128, TW>(a, st);
}

template <int BM, int BN, int WGM, int WGN, int TW>
static int launch_w1(WgradBf16Args& a, hipStream_t st) {
  a.n_co_tiles = rsis_cdiv(a.Cout, BM);
  a.n_n_tiles = rsis_cdiv(a.Cs, BN);
  const int ntile = a.n_co_tiles * a.n_n_tiles;
  split_plan(a, a.blk ? W1T_TP : TW, a.blk ? 1 : 64 / TW, ntile, 256);
  const dim3 grid(ntile, rsis_cdiv(a.n_sp_tiles, a.tiles_per_split));
  if (a.blk) w1t_launch<BM, BN, W1T_TP, w1t_nr(BM, BN)>(a, grid, st);
  else if (a.W % 4 == 0) hipLaunchKernelGGL((wgrad1_bf16_kernel<BM, BN, WGM, WGN, TW, 1>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((wgrad1_bf16_kernel<BM, BN, WGM, WGN, TW, 0>), grid, dim3(256), 0, st, a);
  return rsis_check_launch();
}

template <int TW>
static int launch_w1_tw(WgradBf16Args& a, hipStream_t st) {
  if (a.Cout <= 64 && a.Cs <= 64) return launch_w1<64, 64, 2, 2, TW>(a, st);
  if (a.Cout <= 64) return launch_w1<64, 128, 2, 2, TW>(a, st);
  if (a.Cs <= 64) return launch_w1<128, 64, 2, 2, TW>(a, st);
  return launch_w1<128, 128, 2, 2, TW>(a, st);
}

// stride 1, "same" padding, 32-bit offsets inside one image slab
bool rsis_wgrad_bf16_supported(const WgradArgs& w, int ks) {
  if (!(ks == 1 || ks == 3) || w.stride != 1 || w.pad != ks / 2 || w.H != w.Ho || w.W != w.Wo) return false;
  if (w.Cout == 1) return false;          // conv_out: HBM-bound VALU kernel (conv_c1.hip)
  if (w.blk && ((w.Cout & 7) || (w.Cs & 7))) return false;
  const long img = (long)w.H * w.W * 4;
  return (long)w.Cout * img < (1L << 30) && (long)w.Cs * img < (1L << 30);
}

int rsis_launch_conv_wgrad_bf16(const WgradArgs& w, int ks, hipStream_t st) {
  WgradBf16Args a = {};
  a.dy = w.dy; a.x = w.x; a.dw = w.dw; a.B = w.B; a.Cs = w.Cs; a.H = w.H; a.W = w.W; a.Cout = w.Cout;
  a.ldo = w.ldo; a.n_off = w.n_off; a.interleave_hid = (short)w.interleave_hid; a.blk = (short)w.blk;
  const int tw = w.W > 16 ? 32 : (w.W > 8 ? 16 : 8);     // widest 64-pixel tile the map fills
  if (ks == 3) {
    if (tw == 32) return launch_w3_tw<32>(a, st);
    if (tw == 16) return launch_w3_tw<16>(a, st);
    return launch_w3_tw<8>(a, st);
  }
  // 1x1: the reduction runs over the flattened map (the H*W pixels of a channel are contiguous), 64 pixels per tile: whole
  // 16-byte loads whenever H*W % 4 == 0.  Tiled in 2-D the 14-pixel rows of the 224 x 224 pyramid went dword by dword, 4x the
  // VMEM instructions, and the texture-address rate -- not HBM -- bound the loop (256 -> 1024 @14^2: 43 -> 29 us).
  a.W = w.H * w.W; a.H = 1;
  return launch_w1_tw<64>(a, st);
}

// ---- grouped launch (host side): jobs bucketed by kernel instantiation, every block of a bucket walks ~L spatial tiles ----
struct WgbKey { int ks, bm, bn, tw, v4, th; };
static WgbKey wgb_key(const WgradBf16Args& a, int ks) {      // the rules of launch_w3_tw / launch_w1_tw / rsis_launch_conv_wgrad_bf16
  WgbKey k = {ks, 0, 0, 0, 0, 0};
  if (ks == 3) {
    k.tw = a.W > 16 ? 32 : (a.W > 8 ? 16 : 8);
    k.bm = a.Cout <= 32 ? 32 : ((a.Cout <= 64 || a.Cs <= 512) ? 64 : 128);
    k.bn = 32;
  } else {
    k.tw = a.blk ? W1T_TP : 64;
    k.bm = a.Cout <= 64 ? 64 : 128;
    k.bn = a.Cs <= 64 ? 64 : 128;
  }
  k.v4 = a.blk ? 2 : (a.W % 4 == 0 ? 1 : 0);      // the IN template argument (2: blk operands -> the DMA kernels)
  k.th = ks == 3 ? (a.blk ? w3t_th(k.bm, k.tw) : 64 / k.tw) : (a.blk ? 1 : 64 / k.tw);      // tile height (blk 3x3: the DMA kernel's tile)
  return k;
}
static inline bool wgb_same(const WgbKey& a, const WgbKey& b) { return a.ks == b.ks && a.bm == b.bm && a.bn == b.bn && a.tw == b.tw && a.v4 == b.v4; }

template <typename LaunchFn>
static int wgb_launch_bucket(WgradBf16Args* jobs, int n, const WgbKey& k, LaunchFn launch) {
  const int TH = k.th;
  long total = 0;
  for (int j = 0; j < n; ++j) {
    WgradBf16Args& a = jobs[j];
    a.n_co_tiles = rsis_cdiv(a.Cout, k.bm);
    a.n_n_tiles = rsis_cdiv(a.Cs, k.bn);
    a.n_sp_tiles = a.B * rsis_cdiv(a.H, TH) * rsis_cdiv(a.W, k.tw);
    total += (long)a.n_co_tiles * a.n_n_tiles * a.n_sp_tiles;
  }
  static const int env_tb = getenv("RSIS_WGB_GROUP_BLOCKS") ? atoi(getenv("RSIS_WGB_GROUP_BLOCKS")) : 0;     // tuning knob
  // (swept at the bench geometry, 224^2 / batch 32, with two blocks per CU resident: 640 18.77 ms per step, 768 18.51, 896 18.43,
  //  1024 18.79, 1280 18.53, 1536 18.52, 1792 18.52 -- twice each, reproducible to 0.02; 1024 happens to cut the layer-3 jobs badly)
  const long target_blocks = env_tb > 0 ? env_tb : 896;
  long L = (total + target_blocks - 1) / target_blocks;
  if (L < 2) L = 2;
  if (rsis_deterministic()) L = 1L << 40;
  int j0 = 0;
  while (j0 < n) {
    WgradBf16Group g = {};
    // items of this launch: a job with up to 8 splits is one item per split, a job with more is 8 items of consecutive splits
    struct Item { int job, split, blocks; long work; };
    Item items[8 * RSIS_WGB_LANE_ITEMS];
    int ni = 0, nj = 0;
    while (j0 + nj < n && nj < RSIS_WGB_MAXJ) {
      WgradBf16Args a = jobs[j0 + nj];
      int nsplit = rsis_cdiv(a.n_sp_tiles, L);
      if (nsplit < 1) nsplit = 1;
      if (nsplit > 60000) nsplit = 60000;
      a.tiles_per_split = rsis_cdiv(a.n_sp_tiles, nsplit);
      nsplit = rsis_cdiv(a.n_sp_tiles, a.tiles_per_split);
      const int nit = nsplit < 8 ? nsplit : 8;
      if (ni + nit > 8 * RSIS_WGB_LANE_ITEMS) break;            // (nj >= 1 here: one job is at most 8 items)
      const int ntile = a.n_co_tiles * a.n_n_tiles;
      for (int it = 0; it < nit; ++it) {
        const int s0 = (int)((long)nsplit * it / nit), s1 = (int)((long)nsplit * (it + 1) / nit);
        const long tiles = (long)(s1 < nsplit ? (long)s1 * a.tiles_per_split : a.n_sp_tiles) - (long)s0 * a.tiles_per_split;
        items[ni++] = Item{nj, s0, ntile * (s1 - s0), (long)ntile * tiles};
      }
      g.job[nj++] = a;
    }
    g.n = nj;
    for (int x = 1; x < ni; ++x)           // longest item first onto the least loaded lane that still has a free slot
      for (int y = x; y > 0 && items[y].work > items[y - 1].work; --y) { const Item t = items[y]; items[y] = items[y - 1]; items[y - 1] = t; }
    long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int blocks[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ni; ++i) {
      int best = -1;
      for (int x = 0; x < 8; ++x)
        if (g.lane_n[x] < RSIS_WGB_LANE_ITEMS && (best < 0 || load[x] < load[best])) best = x;
      const int e = g.lane_n[best]++;
      g.lane_start[best][e] = blocks[best];
      g.lane_job[best][e] = (unsigned char)items[i].job;
      g.lane_split0[best][e] = (unsigned short)items[i].split;
      blocks[best] += items[i].blocks;
      load[best] += items[i].work;
    }
    int maxb = 0;
    for (int x = 0; x < 8; ++x) { g.lane_start[x][g.lane_n[x]] = blocks[x]; if (blocks[x] > maxb) maxb = blocks[x]; }
    launch(g, 8 * maxb);
    if (rsis_check_launch() != RSIS_OK) return RSIS_ERR_LAUNCH;
    j0 += nj;
  }
  return RSIS_OK;
}

#define WGB3(BMv, TWv)                                                                                             \
  if (k.bm == BMv && k.tw == TWv) {                                                                                \
    if (k.v4 == 2) return wgb_launch_bucket(jobs, n, k, [&](const WgradBf16Group& g, int blocks) {                 \
      w3t_launch_group<BMv, TWv, w3t_th(BMv, TWv), w3t_nr(BMv, TWv)>(g, blocks, st); });                           \
    if (k.v4 == 1) return wgb_launch_bucket(jobs, n, k, [&](const WgradBf16Group& g, int blocks) {                 \
      hipLaunchKernelGGL((wgrad3_bf16_group_kernel<BMv, TWv, 1>), dim3(blocks), dim3(256), 0, st, g); });          \
    return wgb_launch_bucket(jobs, n, k, [&](const WgradBf16Group& g, int blocks) {                                \
      hipLaunchKernelGGL((wgrad3_bf16_group_kernel<BMv, TWv, 0>), dim3(blocks), dim3(256), 0, st, g); });          \
  }
#define WGB1(BMv, BNv)                                                                                             \
  if (k.bm == BMv && k.bn == BNv) {                                                                                \
    if (k.v4 == 2) return wgb_launch_bucket(jobs, n, k, [&](const WgradBf16Group& g, int blocks) {                 \
      w1t_launch_group<BMv, BNv, W1T_TP, w1t_nr(BMv, BNv)>(g, blocks, st); });                                     \
    if (k.v4 == 1) return wgb_launch_bucket(jobs, n, k, [&](const WgradBf16Group& g, int blocks) {                 \
      hipLaunchKernelGGL((wgrad1_bf16_group_kernel<BMv, BNv, 2, 2, 64, 1>), dim3(blocks), dim3(256), 0, st, g); });      \
    return wgb_launch_bucket(jobs, n, k, [&](const WgradBf16Group& g, int blocks) {                                \
      hipLaunchKernelGGL((wgrad1_bf16_group_kernel<BMv, BNv, 2, 2, 64, 0>), dim3(blocks), dim3(256), 0, st, g); });      \
  }
static int wgb_dispatch(WgradBf16Args* jobs, int n, const WgbKey& k, hipStream_t st) {
  if (k.ks == 3) {
    WGB3(32, 32) WGB3(64, 32) WGB3(128, 32) WGB3(32, 16) WGB3(64, 16) WGB3(128, 16) WGB3(32, 8) WGB3(64, 8) WGB3(128, 8)
  } else {
    WGB1(64, 64) WGB1(64, 128) WGB1(128, 64) WGB1(128, 128)
  }
  return RSIS_ERR_ARG;
}
#undef WGB3
#undef WGB1

// n weight gradients that rsis_wgrad_bf16_supported accepts, all with the same kernel size
int rsis_launch_conv_wgrad_bf16_group(const WgradArgs* w, int n, int ks, hipStream_t st) {
  if (n < 1) return RSIS_OK;
  WgradBf16Args* all = (WgradBf16Args*)malloc(sizeof(WgradBf16Args) * n * 2);
  WgbKey* key = (WgbKey*)malloc(sizeof(WgbKey) * n);
  if (!all || !key) { free(all); free(key); return RSIS_ERR_LAUNCH; }
  WgradBf16Args* bucket = all + n;
  for (int j = 0; j < n; ++j) {
    WgradBf16Args a = {};
    a.dy = w[j].dy; a.x = w[j].x; a.dw = w[j].dw; a.B = w[j].B; a.Cs = w[j].Cs; a.H = w[j].H; a.W = w[j].W; a.Cout = w[j].Cout;
    a.ldo = w[j].ldo; a.n_off = w[j].n_off; a.interleave_hid = (short)w[j].interleave_hid; a.blk = (short)w[j].blk;
    if (ks == 1) { a.W = w[j].H * w[j].W; a.H = 1; }      // 1x1: the flattened map
    all[j] = a;
    key[j] = wgb_key(a, ks);
  }
  int rc = RSIS_OK;
  for (int j = 0; j < n && rc == RSIS_OK; ++j) {
    if (key[j].ks < 0) continue;
    const WgbKey k = key[j];
    int m = 0;
    for (int i = j; i < n; ++i)
      if (key[i].ks >= 0 && wgb_same(key[i], k)) { bucket[m++] = all[i]; key[i].ks = -1; }
    rc = wgb_dispatch(bucket, m, k, st);
  }
  free(all); free(key);
  return rc;
}
