// Weight-gradient of the NCHW convolutions of the RSIS hot path for gfx950: split-K implicit GEMM on the exact-f32
// MFMA (v_mfma_f32_32x32x2_f32), accumulating with fp32 global atomics into dW kept in the REFERENCE layout
// [Cout][Ctot][ks][ks] (what autograd produced for nn.Conv2d in reference src/modules/clstm.py:17, model.py:43-47,109
// and the torchvision trunk).
//
// GEMM view:  dW[co][n] += sum_px dy[co][px] * Xcol[px][n],   n = (ci, r, s) of ONE source tensor (a channel-concat
// conv issues one launch per source with that source's channel offset), px = (b, ho, wo) is the reduction axis.
// Both operands are contiguous along px in NCHW, so global loads run along px and the LDS tiles are [row][BKW+1]
// (padded) which makes both the stores and the MFMA operand ds_read_b32 conflict-free.
#include "common.h"
#include <type_traits>

#define BKW 32
#define LDK (BKW + 1)


// V4 (1x1 / stride 1 / H*W % 4 == 0): both operands are contiguous along the reduction (pixel) axis -> float4 global loads.
template <int BM, int BN, int WGM, int WGN, int KS, bool V4>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs p) {
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  constexpr int KK = KS * KS;
  constexpr int A_LOADS = BM / 8, B_LOADS = BN / 8;
  __shared__ float lds[2 * (BM + BN) * LDK];
  float* As0 = lds;                  // [2][BM][LDK]
  float* Bs0 = lds + 2 * BM * LDK;   // [2][BN][LDK]

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int co_t = blockIdx.x % p.n_co_tiles, n_t = blockIdx.x / p.n_co_tiles;
  const int HoWo = p.Ho * p.Wo, HW = p.H * p.W;
  const int Npx = p.B * HoWo;
  const int px_begin = blockIdx.y * p.chunk;
  const int px_end = min(px_begin + p.chunk, Npx);
  if (px_begin >= px_end) return;
  const int ntiles = (px_end - px_begin + BKW - 1) / BKW;
  const int Nn = p.Cs * KK;

  const int kl = tid & 31, row0 = tid >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float ra[A_LOADS], rb[B_LOADS];
  typedef typename std::conditional<(KK > 31), unsigned long long, unsigned>::type mask_t;
  // loop-invariant decode of this thread's B rows: n -> (ci, r, s) -> element offset from the (hi0, wi0) corner + tap bit
  int noff[B_LOADS], nbit[B_LOADS];
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    const int n = n_t * BN + row0 + i * 8;
    const int ci = n / KK;
    const int rs = n - ci * KK;
    const int r = rs / KS, s2 = rs - r * KS;
    noff[i] = ci * HW + r * p.W + s2;
    nbit[i] = n < Nn ? rs : (int)(8 * sizeof(mask_t) - 1);   // the top bit of vmask is never set
  }
  // V4 mapping: 8 float4 groups per 32-pixel tile row; thread -> (group kg, first row r4), rows step by 32
  const int kg = tid & 7, r4 = tid >> 3;

  auto load_tile = [&](int t) {
    if constexpr (V4) {
      const int px = px_begin + t * BKW + kg * 4;
      const bool pv = px < px_end;
      const int b = pv ? px / HoWo : 0;
      const int sp = pv ? px - b * HoWo : 0;
      const float* __restrict__ dyb = p.dy + (size_t)b * p.Cout * HoWo + sp;
      const float* __restrict__ xb = p.x + (size_t)b * p.Cs * HW + sp;
#pragma unroll
      for (int i = 0; i < A_LOADS / 4; ++i) {
        const int co = co_t * BM + r4 + i * 32;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pv && co < p.Cout) v = *reinterpret_cast<const f32x4*>(dyb + (size_t)co * HoWo);
        ra[4 * i] = v[0]; ra[4 * i + 1] = v[1]; ra[4 * i + 2] = v[2]; ra[4 * i + 3] = v[3];
      }
#pragma unroll
      for (int i = 0; i < B_LOADS / 4; ++i) {
        const int n = n_t * BN + r4 + i * 32;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pv && n < Nn) v = *reinterpret_cast<const f32x4*>(xb + (size_t)n * HW);
        rb[4 * i] = v[0]; rb[4 * i + 1] = v[1]; rb[4 * i + 2] = v[2]; rb[4 * i + 3] = v[3];
      }
      return;
    }
    const int px = px_begin + t * BKW + kl;
    const bool pv = px < px_end;
    const int b = pv ? px / HoWo : 0;
    const int sp = pv ? px - b * HoWo : 0;
    const int ho = sp / p.Wo, wo = sp - ho * p.Wo;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    const float* __restrict__ dyb = p.dy + (size_t)b * p.Cout * HoWo + sp;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const int co = co_t * BM + row0 + i * 8;
      ra[i] = (pv && co < p.Cout) ? dyb[(size_t)co * HoWo] : 0.f;
    }
    // per-pixel validity bit of every filter tap (once per tile); the tap / channel offsets of this thread's B rows are
    // loop invariant (noff / nbit, computed before the K loop)
    mask_t vmask = 0;
    if (pv) {
#pragma unroll
      for (int r = 0; r < KS; ++r)
#pragma unroll
        for (int s = 0; s < KS; ++s)
          if (((unsigned)(hi0 + r) < (unsigned)p.H) && ((unsigned)(wi0 + s) < (unsigned)p.W)) vmask |= (mask_t)1 << (r * KS + s);
    }
    const float* __restrict__ xb = p.x + (size_t)b * p.Cs * HW + hi0 * p.W + wi0;
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      const bool ok = (vmask >> nbit[i]) & 1;
      rb[i] = ok ? xb[noff[i]] : 0.f;
    }
  };
  auto store_tile = [&](int buf) {
    float* As = As0 + buf * BM * LDK;
    float* Bs = Bs0 + buf * BN * LDK;
    if constexpr (V4) {
#pragma unroll
      for (int i = 0; i < A_LOADS / 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) As[(r4 + i * 32) * LDK + kg * 4 + k] = ra[4 * i + k];
#pragma unroll
      for (int i = 0; i < B_LOADS / 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) Bs[(r4 + i * 32) * LDK + kg * 4 + k] = rb[4 * i + k];
      return;
    }
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) As[(row0 + i * 8) * LDK + kl] = ra[i];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) Bs[(row0 + i * 8) * LDK + kl] = rb[i];
  };
  auto compute = [&](int buf) {
    const float* As = As0 + buf * BM * LDK + (wm * TM * 32 + l31) * LDK;
    const float* Bs = Bs0 + buf * BN * LDK + (wn * TN * 32 + l31) * LDK;
#pragma unroll
    for (int kk = 0; kk < BKW / 2; ++kk) {
      const int k = kk * 2 + hi;
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[i * 32 * LDK + k];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[j * 32 * LDK + k];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < ntiles) load_tile(t + 1);
    compute(cur);
    if (t + 1 < ntiles) store_tile(cur ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n_t * BN + wn * TN * 32 + j * 32 + l31;
    if (n >= Nn) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_t * BM + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (co >= p.Cout) continue;
        const int row = p.interleave_hid > 0 ? (co & 3) * p.interleave_hid + (co >> 2) : co;
        atomicAdd(p.dw + (size_t)row * p.ldo + p.n_off + n, acc[i][j][r]);
      }
    }
  }
}

template <int BM, int BN, int WGM, int WGN, int KS>
static int launch_wgrad_cfg(WgradArgs& a, hipStream_t st) {
  const long Npx = (long)a.B * a.Ho * a.Wo;
  a.n_co_tiles = rsis_cdiv(a.Cout, BM);
  a.n_n_tiles = rsis_cdiv((long)a.Cs * KS * KS, BN);
  const int ntile = a.n_co_tiles * a.n_n_tiles;
  // split-K: fill the resident block slots of the chip in ONE round (blocks/CU from the kernel's LDS/VGPR budget: 2 for the
  // 128x128 tile, 3 otherwise) -- a few blocks more than the slot count would cost a whole extra round -- while keeping at
  // least 4 K-tiles (128 px) per split
  const int slots = 256 * (BM * BN >= 128 * 128 ? 2 : 3);
  int nsplit = ntile >= slots ? 1 : slots / ntile;
  const int max_split = (int)((Npx + 4 * BKW - 1) / (4 * BKW));
  if (nsplit > max_split) nsplit = max_split;
  if (nsplit < 1 || rsis_deterministic()) nsplit = 1;
  a.chunk = rsis_roundup(rsis_cdiv(Npx, nsplit), BKW);
  nsplit = rsis_cdiv(Npx, a.chunk);   // (rounding the chunk up can only lower the split count)
  if constexpr (KS == 1 && BM % 32 == 0 && BN % 32 == 0) {
    if (a.stride == 1 && a.pad == 0 && (a.H * a.W) % 4 == 0 && a.H == a.Ho && a.W == a.Wo) {
      hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WGM, WGN, KS, true>), dim3(ntile, nsplit), dim3(256), 0, st, a);
      return rsis_check_launch();
    }
  }
  hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WGM, WGN, KS, false>), dim3(ntile, nsplit), dim3(256), 0, st, a);
  return rsis_check_launch();
}

template <int KS>
static int launch_wgrad_ks(WgradArgs& a, hipStream_t st) {
  if (a.Cout <= 32) return launch_wgrad_cfg<32, 128, 1, 4, KS>(a, st);
  if (a.Cout <= 64) return launch_wgrad_cfg<64, 128, 2, 2, KS>(a, st);
  return launch_wgrad_cfg<128, 128, 2, 2, KS>(a, st);
}

int rsis_launch_conv_wgrad(WgradArgs& a, int ks, hipStream_t st) {
  if (ks == 1) return launch_wgrad_ks<1>(a, st);
  if (ks == 3) return launch_wgrad_ks<3>(a, st);
  if (ks == 7) return launch_wgrad_ks<7>(a, st);
  return RSIS_ERR_UNSUPPORTED;
}
