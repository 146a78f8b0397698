// Direct weight-gradient of 3x3 / stride 1 / pad 1 convolutions for gfx950 (exact-f32 MFMA 32x32x2), NCHW fp32:
//     dW[co][ci][r][s] += sum_{b,y,x} dy[b][co][y][x] * x[b][ci][y+r-1][x+s-1]
// (autograd of nn.Conv2d in reference src/modules/clstm.py:17,44 -- the ConvLSTM gates, time-batched over T*B images --
//  model.py:43-47,109 and the 3x3 convs of the torchvision bottlenecks).
//
// Same idea as conv3x3_direct.hip: per spatial tile the block stages the input patch WITH halo (zero filled) and the dy
// tile in LDS once, and the MFMA loop is address-free.  GEMM view D[co][n] with n = ci*9 + rs (exactly the reference's
// weight layout, so the result is accumulated straight into dW) and the PIXELS as the reduction axis: MFMA A = dy
// (lane = co, k = pixel pair), B = patch (lane = n, k = pixel pair) whose LDS address is  lane_base(n) + pixel offset
// with lane_base(n) = ci*CHS + r*PW + s fixed per lane and the pixel offset a compile-time immediate.  One A read feeds
// TN MFMAs.  Blocks loop over a range of spatial tiles (split-K over tiles and images) and finish with fp32 atomics.
#include "common.h"
#include <stdlib.h>

typedef const float __attribute__((address_space(1)))* gcf_t;
typedef float __attribute__((address_space(1)))* gf_t;

struct WgradDirectArgs {
  const float* dy;   // [B][Cout][H][W]
  const float* x;    // [B][Cs][H][W]
  float* dw;         // [Cout][ldo]
  int B, Cs, H, W, Cout;
  int ldo, n_off, interleave_hid;
  int n_co_tiles, n_ci_chunks, n_sp_tiles, tiles_per_block;
};

// WC x WN x WP = 4 waves: co tiles x n-tile groups x pixel-row split.  TN = 32-wide n tiles per wave.
template <int WC, int WN, int WP, int TN, int TW, int TH>
__global__ __launch_bounds__(256) void conv3x3_wgrad_direct_kernel(const WgradDirectArgs p) {
  constexpr int BM = 32 * WC;
  constexpr int CI_T = (WN * TN * 32 + 8) / 9;       // input channels whose 9 taps cover the block's n range
  constexpr int PW = TW + 2, PH = TH + 2;
  constexpr int CHS = (PH * PW) | 1;                 // odd channel stride: lanes of different channels spread over banks
  constexpr int TP = TW * TH, PXS = TP + 1;          // dy row stride (odd)
  constexpr int XS = CI_T * CHS, DS = BM * PXS;
  constexpr int ROWS = TH / WP;                      // pixel rows per wave
  static_assert(WC * WN * WP == 4 && TH % WP == 0 && TW % 2 == 0, "config");

  __shared__ float lds[XS + DS];
  float* const Xs = lds;
  float* const Ds = lds + XS;

  const gcf_t dyp = (gcf_t)p.dy, xp = (gcf_t)p.x;
  const int B = p.B, Cs = p.Cs, H = p.H, W = p.W, HW = H * W, Cout = p.Cout;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave / (WN * WP), wn = (wave / WP) % WN, wp = wave % WP;

  const int co_t = blockIdx.x % p.n_co_tiles, ci_c = blockIdx.x / p.n_co_tiles;
  const int co0 = co_t * BM, ci0 = ci_c * CI_T;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int t_begin = blockIdx.y * p.tiles_per_block;
  const int t_end = min(t_begin + p.tiles_per_block, p.n_sp_tiles);

  // per-lane LDS bases
  int xbase[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nl = (wn * TN + j) * 32 + l31;
    const int cl = nl / 9, rs = nl - cl * 9;
    const int r = rs / 3, s = rs - r * 3;
    xbase[j] = cl * CHS + (wp * ROWS + r) * PW + s + hi;
  }
  const int dbase = (wc * 32 + l31) * PXS + wp * ROWS * TW + hi;

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  for (int t = t_begin; t < t_end; ++t) {
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
    const int x0 = tx * TW, y0 = ty * TH;
    __syncthreads();   // previous tile fully consumed
    // ---- stage the input patch (halo, zero fill) and the dy tile ----
    // loads are issued in batches of U before any LDS store, so U global loads are in flight per thread
    constexpr int U = 12;
    constexpr int NXE = CI_T * PH * PW, NDE = BM * TP;
    const gcf_t xb = xp + ((size_t)b * Cs + ci0) * HW;
#pragma unroll 1
    for (int e0 = 0; e0 < NXE; e0 += 256 * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * 256 + tid;
        const int cl = e / (PH * PW), rem = e - cl * (PH * PW);
        const int py = rem / PW, pxx = rem - py * PW;
        const int gy = y0 + py - 1, gx = x0 + pxx - 1;
        v[u] = 0.f;
        if (e < NXE && ci0 + cl < Cs && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
          v[u] = xb[(size_t)cl * HW + gy * W + gx];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * 256 + tid;
        const int cl = e / (PH * PW), rem = e - cl * (PH * PW);
        if (e < NXE) Xs[cl * CHS + rem] = v[u];
      }
    }
    const gcf_t db = dyp + ((size_t)b * Cout + co0) * HW;
#pragma unroll 1
    for (int e0 = 0; e0 < NDE; e0 += 256 * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * 256 + tid;
        const int cl = e / TP, rem = e - cl * TP;
        const int py = rem / TW, pxx = rem - py * TW;
        const int gy = y0 + py, gx = x0 + pxx;
        v[u] = 0.f;
        if (e < NDE && co0 + cl < Cout && gy < H && gx < W) v[u] = db[(size_t)cl * HW + gy * W + gx];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * 256 + tid;
        const int cl = e / TP, rem = e - cl * TP;
        if (e < NDE) Ds[cl * PXS + rem] = v[u];
      }
    }
    __syncthreads();
    // ---- MFMA over this wave's pixel rows: k = pixel pairs (x, x+1) ----
#pragma unroll
    for (int y = 0; y < ROWS; ++y) {
#pragma unroll
      for (int xx = 0; xx < TW; xx += 2) {
        const float a = Ds[dbase + y * TW + xx];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float bv = Xs[xbase[j] + y * PW + xx];
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[j], 0, 0, 0);
        }
      }
    }
  }

  // ---- accumulate into dW (reference layout): column n = ci*9 + rs is linear in memory ----
  float* const dw = p.dw;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nl = (wn * TN + j) * 32 + l31;
    if (ci0 * 9 + nl >= Cs * 9) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (co >= Cout) continue;
      const int row = p.interleave_hid > 0 ? (co & 3) * p.interleave_hid + (co >> 2) : co;
      atomicAdd(dw + (size_t)row * p.ldo + p.n_off + ci0 * 9 + nl, acc[j][r]);
    }
  }
}

template <int WC, int WN, int WP, int TN, int TW, int TH>
static int launch_wgd(WgradDirectArgs& a, hipStream_t st) {
  constexpr int BM = 32 * WC;
  constexpr int CI_T = (WN * TN * 32 + 8) / 9;
  a.n_co_tiles = rsis_cdiv(a.Cout, BM);
  a.n_ci_chunks = rsis_cdiv(a.Cs, CI_T);
  a.n_sp_tiles = rsis_cdiv(a.W, TW) * rsis_cdiv(a.H, TH) * a.B;
  const int pairs = a.n_co_tiles * a.n_ci_chunks;
  static const int target = getenv("RSIS_WGD_BLOCKS") ? atoi(getenv("RSIS_WGD_BLOCKS")) : 768;
  int nsplit = rsis_cdiv(target, pairs);
  if (nsplit > a.n_sp_tiles) nsplit = a.n_sp_tiles;
  if (nsplit < 1) nsplit = 1;
  a.tiles_per_block = rsis_cdiv(a.n_sp_tiles, nsplit);
  nsplit = rsis_cdiv(a.n_sp_tiles, a.tiles_per_block);
  hipLaunchKernelGGL((conv3x3_wgrad_direct_kernel<WC, WN, WP, TN, TW, TH>), dim3(pairs, nsplit), dim3(256), 0, st, a);
  return rsis_check_launch();
}

template <int TW, int TH>
static int launch_wgd_geom(WgradDirectArgs& a, hipStream_t st) {
  if (a.Cout > 32) {
    if (a.Cs > 32) return launch_wgd<2, 2, 1, 9, TW, TH>(a, st);   // 64 co x 64 ci
    return launch_wgd<2, 1, 2, 9, TW, TH>(a, st);                   // 64 co x 32 ci, pixel rows split in 2
  }
  if (a.Cs > 10) return launch_wgd<1, 1, 4, 9, TW, TH>(a, st);      // 32 co x 32 ci, pixel rows split in 4
  return launch_wgd<1, 1, 4, 3, TW, TH>(a, st);                     // 32 co x <=10 ci
}

int rsis_launch_conv3x3_wgrad_direct(const WgradArgs& w, hipStream_t st) {
  WgradDirectArgs a = {};
  a.dy = w.dy; a.x = w.x; a.dw = w.dw; a.B = w.B; a.Cs = w.Cs; a.H = w.H; a.W = w.W; a.Cout = w.Cout;
  a.ldo = w.ldo; a.n_off = w.n_off; a.interleave_hid = w.interleave_hid;
  if (a.W <= 8 && a.H <= 8) return launch_wgd_geom<8, 8>(a, st);
  return launch_wgd_geom<16, 8>(a, st);
}
