// BatchNorm (+ residual add + ReLU) and the small layout helpers on channel-blocked bf16 activations (conv_blk.hip: logical
// [B][C][H][W] stored as bf16 [B][C/8][H][W][8], one 16-byte cell = the 8 channels of a pixel).
// Reference: nn.BatchNorm2d / ReLU / the residual add of the torchvision bottlenecks behind src/modules/vision.py:12-19 (train-mode
// batch statistics, running-statistics update with the unbiased variance; eval mode normalises with the running statistics).
//
// A thread works on whole cells: eight per-channel fp32 accumulators (statistics) or eight scale / shift pairs (apply), 16-byte
// loads and stores, half the bytes of the fp32 NCHW kernels of pointwise.hip.  Statistics and the two backward sums are reduced
// in two levels -- S partial sums per channel written by the grid (double), summed again by every block of the apply pass (8
// threads x S adds) -- so the result does not depend on the order blocks run in: bit-reproducible without atomics.
// The arithmetic is fp32 on the exact bf16 inputs, with ONE rounding to bf16 at each store.
#include "common.h"
#include <type_traits>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void cell_unpack(const u32x4 c, float* v) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[2 * k] = __uint_as_float(c[k] << 16);
    v[2 * k + 1] = __uint_as_float(c[k] & 0xFFFF0000u);
  }
}
__device__ __forceinline__ u32x4 cell_pack(const float* v) {
  u32x4 c;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x2 f = {v[2 * k], v[2 * k + 1]};
    c[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
  }
  return c;
}
// cell index of element n = b * HW + sp of channel block cb
// (n < B * HW < 2^31, checked by the launchers: ONE 32-bit division per cell -- the 64-bit one this replaced is ~80 instructions, and
//  an apply thread does four of them before it can request its first cell)
__device__ __forceinline__ size_t cell_index(long n, int cb, int Cb, int HW) {
  const unsigned un = (unsigned)n, b = un / (unsigned)HW;
  return ((size_t)b * Cb + cb) * HW + (size_t)(un - b * (unsigned)HW);
}

// sum of 16 per-thread values over the 256 threads of the block -> out[16] (double), written by threads 0..15
__device__ __forceinline__ void block_sum16(float* a, double* out) {
  __shared__ float red[4][16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float v = a[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    a[k] = v;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) red[wave][k] = a[k];
  }
  __syncthreads();
  if (threadIdx.x < 16) out[threadIdx.x] = (double)red[0][threadIdx.x] + (double)red[1][threadIdx.x] + (double)red[2][threadIdx.x] + (double)red[3][threadIdx.x];
}

// the same over the NW waves of a larger block, result in LDS (tot[16], valid after the barrier for every thread)
template <int NW>
__device__ __forceinline__ void block_sum16_lds(float* a, double* tot) {
  __shared__ float redw[NW][16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float v = a[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    a[k] = v;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) redw[wave][k] = a[k];
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += (double)redw[w][threadIdx.x];
    tot[threadIdx.x] = s;
  }
  __syncthreads();
}

// ---- small maps (B * HW <= 4096 cells per channel block: layer 4 at batch 32; more cells per thread spill): ONE block of 512 threads per channel block does
//      the statistics pass and the apply pass with its <= 8 cells per thread held in registers (every load issued before the first
//      use: a loop with one load per iteration pays a memory round trip per iteration).  One launch instead of two -- the trunk has
//      ~80 such BatchNorms per step, each way -- one read of the inputs, no partial-sum round trip. ----
template <int NPT>
__global__ __launch_bounds__(512) void blk_bn_fwd_block_kernel(const u32x4* __restrict__ x, const u32x4* __restrict__ res, u32x4* __restrict__ y,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ run_mean, float* __restrict__ run_var,
                                                                float* __restrict__ save_mean, float* __restrict__ save_rstd, int Cb, int HW,
                                                                int N, float eps, float momentum, int relu) {
  __shared__ double tot[16];
  __shared__ float sc[8], sh[8];
  const int cb = blockIdx.x;
  // the thread's NPT cells stay in registers between the two passes: all loads are issued before the first use
  u32x4 c[NPT];
#pragma unroll
  for (int u = 0; u < NPT; ++u) {
    const int n = threadIdx.x + u * 512;
    c[u] = n < N ? x[cell_index(n, cb, Cb, HW)] : u32x4{0u, 0u, 0u, 0u};
  }
  float a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = 0.f;
#pragma unroll
  for (int u = 0; u < NPT; ++u) {
    float v[8];
    cell_unpack(c[u], v);
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] += v[k]; a[8 + k] += v[k] * v[k]; }
  }
  u32x4 r[NPT];
  if (res) {      // (the residual's loads overlap the reduction)
#pragma unroll
    for (int u = 0; u < NPT; ++u) r[u] = threadIdx.x + u * 512 < N ? res[cell_index(threadIdx.x + u * 512, cb, Cb, HW)] : u32x4{0u, 0u, 0u, 0u};
  }
  block_sum16_lds<8>(a, tot);
  if (threadIdx.x < 8) {
    const int k = threadIdx.x, ch = cb * 8 + k;
    const double m = tot[k] / (double)N;
    double var = tot[8 + k] / (double)N - m * m;
    if (var < 0.0) var = 0.0;
    const float mean = (float)m, rstd = rsqrtf((float)var + eps);
    save_mean[ch] = mean;
    save_rstd[ch] = rstd;
    if (run_mean) {
      const float unb = N > 1 ? (float)(var * (double)N / (double)(N - 1)) : (float)var;
      run_mean[ch] = (1.f - momentum) * run_mean[ch] + momentum * mean;
      run_var[ch] = (1.f - momentum) * run_var[ch] + momentum * unb;
    }
    const float g = gamma[ch] * rstd;
    sc[k] = g;
    sh[k] = beta[ch] - mean * g;
  }
  __syncthreads();
  float scv[8], shv[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { scv[k] = sc[k]; shv[k] = sh[k]; }
#pragma unroll
  for (int u = 0; u < NPT; ++u) {
    if (threadIdx.x + u * 512 >= N) continue;
    float v[8];
    cell_unpack(c[u], v);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = v[k] * scv[k] + shv[k];
    if (res) {
      float rv[8];
      cell_unpack(r[u], rv);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += rv[k];
    }
    if (relu) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    y[cell_index(threadIdx.x + u * 512, cb, Cb, HW)] = cell_pack(v);
  }
}

// sum of the S per-split partials of channel block cb (part[s][cb][16]) by the whole block: 16 lanes per value, each adds the splits
// i = j, j + 16, ... and the 16 lanes fold by shuffles -- a fixed order.  Result in tot[16] (LDS), valid for every thread on return.
// (The 8 threads that used to add S partials one after the other waited one L2 round trip per partial.)
__device__ __forceinline__ void partial_sums16(const double* __restrict__ part, int S, int cb, int Cb, double* tot) {
  const int v = threadIdx.x >> 4, j = threadIdx.x & 15;       // 256 threads: value v = 0..15, lane j of its 16
  double s = 0.0;
  for (int i = j; i < S; i += 16) s += part[((size_t)i * Cb + cb) * 16 + v];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (j == 0) tot[v] = s;
  __syncthreads();
}

// ---- statistics: part[s][cb][0..7] = sum x, [8..15] = sum x^2 over split s of the B * HW cells of channel block cb ----
__global__ __launch_bounds__(256) void blk_bn_stats_kernel(const u32x4* __restrict__ x, double* __restrict__ part, int Cb, int HW, long N, long per) {
  const int cb = blockIdx.y, s = blockIdx.x;
  const long n0 = s * per, n1 = min(N, n0 + per);
  float a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = 0.f;
  for (long n = n0 + threadIdx.x; n < n1; n += 1024) {
    u32x4 c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long nn = n + u * 256;
      const u32x4 cv = x[cell_index(nn < n1 ? nn : n0, cb, Cb, HW)];        // (clamped address + select: no branch around the load)
      c[u] = nn < n1 ? cv : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v[8];
      cell_unpack(c[u], v);
#pragma unroll
      for (int k = 0; k < 8; ++k) { a[k] += v[k]; a[8 + k] += v[k] * v[k]; }
    }
  }
  block_sum16(a, part + ((size_t)s * Cb + cb) * 16);
}

// ---- y = relu?( (x - mean) * rstd * gamma + beta (+ res) ) ----
template <int U>
__global__ __launch_bounds__(256) void blk_bn_apply_kernel(const u32x4* __restrict__ x, const u32x4* __restrict__ res, u32x4* __restrict__ y,
                                                           const double* __restrict__ part, int S, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ run_mean,
                                                           float* __restrict__ run_var, float* __restrict__ save_mean,
                                                           float* __restrict__ save_rstd, int Cb, int HW, long N, float eps, float momentum,
                                                           int relu, int train) {
  __shared__ float sc[8], sh[8];
  __shared__ double ptot[16];
  const int cb = blockIdx.y;
  // the block's cells are requested BEFORE the per-channel prologue (a chain of dependent loads and double arithmetic on 8 threads):
  // their latency hides behind it
  // (ONE uniform branch, then no load under a condition: `res ? res[idx] : 0` per cell compiles to a branch around each load and a
  //  vmcnt(0) behind every second one -- two dependent round trips where one will do)
  u32x4 xc[U], rc[U];
  if (res) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long n = ((long)blockIdx.x * U + u) * 256 + threadIdx.x;
      const size_t idx = cell_index(n < N ? n : 0, cb, Cb, HW);
      xc[u] = x[idx]; rc[u] = res[idx];
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long n = ((long)blockIdx.x * U + u) * 256 + threadIdx.x;
      xc[u] = x[cell_index(n < N ? n : 0, cb, Cb, HW)]; rc[u] = u32x4{0u, 0u, 0u, 0u};
    }
  }
  float gam_c = 0.f, bet_c = 0.f;            // (the affine parameters of this thread's channel: requested before the partial sums)
  if (threadIdx.x < 8) { gam_c = gamma[cb * 8 + threadIdx.x]; bet_c = beta[cb * 8 + threadIdx.x]; }
  if (train) partial_sums16(part, S, cb, Cb, ptot);
  if (threadIdx.x < 8) {
    const int k = threadIdx.x, c = cb * 8 + k;
    float mean, rstd;
    if (train) {
      const double s = ptot[k], q = ptot[8 + k];
      const double m = s / (double)N;
      double var = q / (double)N - m * m;
      if (var < 0.0) var = 0.0;
      mean = (float)m;
      rstd = rsqrtf((float)var + eps);
      if (blockIdx.x == 0) {
        save_mean[c] = mean;
        save_rstd[c] = rstd;
        if (run_mean) {
          const float unb = N > 1 ? (float)(var * (double)N / (double)(N - 1)) : (float)var;
          run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
          run_var[c] = (1.f - momentum) * run_var[c] + momentum * unb;
        }
      }
    } else {
      mean = run_mean[c];
      rstd = rsqrtf(run_var[c] + eps);
    }
    const float g = gam_c * rstd;
    sc[k] = g;
    sh[k] = bet_c - mean * g;
  }
  __syncthreads();
  float scv[8], shv[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { scv[k] = sc[k]; shv[k] = sh[k]; }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long n = ((long)blockIdx.x * U + u) * 256 + threadIdx.x;
    if (n >= N) continue;
    const size_t idx = cell_index(n, cb, Cb, HW);
    float v[8];
    cell_unpack(xc[u], v);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = v[k] * scv[k] + shv[k];
    if (res) {
      float r[8];
      cell_unpack(rc[u], r);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += r[k];
    }
    if (relu) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    y[idx] = cell_pack(v);
  }
}

// ---- backward.  g = dy * [y > 0] (relu; y = the forward output.  Without y the mask is recomputed from x: no residual then);
//      part[s][cb][0..7] = sum g, [8..15] = sum g * xhat ----
// (scm / shm: the forward's fused scale / shift per channel, sc = gamma * rstd, sh = beta - mean * sc, computed once per thread)
__device__ __forceinline__ void bn_bwd_g(const u32x4 dyc, const u32x4 xc, const bool has_y, const u32x4 yc, const float* mean, const float* rstd,
                                         const float* scm, const float* shm, const int relu, float* g, float* xh) {
  float dv[8], xv[8], yv[8];
  cell_unpack(dyc, dv);
  cell_unpack(xc, xv);
  if (has_y) cell_unpack(yc, yv);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    xh[k] = (xv[k] - mean[k]) * rstd[k];
    bool on = true;
    // (without y the mask is recomputed in the forward's own fused form x * sc + sh: the same roundings, so the mask is the one of
    //  the output the forward stored)
    if (relu) on = has_y ? (yv[k] > 0.f) : (xv[k] * scm[k] + shm[k] > 0.f);
    g[k] = on ? dv[k] : 0.f;
  }
}

__global__ __launch_bounds__(256) void blk_bn_bwd_reduce_kernel(const u32x4* __restrict__ dy, const u32x4* __restrict__ x, const u32x4* __restrict__ y,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                                                double* __restrict__ part, int Cb, int HW, long N, long per, int relu) {
  const int cb = blockIdx.y, s = blockIdx.x;
  const long n0 = s * per, n1 = min(N, n0 + per);
  float mean[8], rstd[8], gam[8], bet[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { mean[k] = save_mean[cb * 8 + k]; rstd[k] = save_rstd[cb * 8 + k]; gam[k] = gamma[cb * 8 + k]; bet[k] = beta[cb * 8 + k]; }
  float scm[8], shm[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { scm[k] = gam[k] * rstd[k]; shm[k] = bet[k] - mean[k] * scm[k]; }
  float a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = 0.f;
  const bool has_y = y != nullptr;
  constexpr int RB = 4;        // cells requested per thread before the first use (a split is 2048+ cells: 8+ per thread)
  // (measured both ways: this loop with its loads under their conditions -- 9.0 us per launch -- beats the branch-free forms,
  //  9.9-10.6 us: a split is a few cells per thread and 66 of the 97 layers have no y stream to skip)
  for (long n = n0 + threadIdx.x; n < n1; n += 256 * RB) {
    u32x4 dc[RB], xc[RB], yc[RB];
    bool ok[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const long nn = n + u * 256;
      ok[u] = nn < n1;
      const size_t idx = ok[u] ? cell_index(nn, cb, Cb, HW) : 0;
      dc[u] = ok[u] ? dy[idx] : u32x4{0u, 0u, 0u, 0u};
      xc[u] = x[idx];
      yc[u] = has_y ? y[idx] : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      float g[8], xh[8];
      bn_bwd_g(dc[u], xc[u], has_y, yc[u], mean, rstd, scm, shm, relu, g, xh);
#pragma unroll
      for (int k = 0; k < 8; ++k) { a[k] += g[k]; a[8 + k] += g[k] * xh[k]; }
    }
  }
  block_sum16(a, part + ((size_t)s * Cb + cb) * 16);
}

// dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat));  dres = g (the gradient of the residual branch);  dgamma / dbeta
template <int U>
__global__ __launch_bounds__(256) void blk_bn_bwd_apply_kernel(const u32x4* __restrict__ dy, const u32x4* __restrict__ x, const u32x4* __restrict__ y,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                                               const double* __restrict__ part, int S, u32x4* __restrict__ dx,
                                                               u32x4* __restrict__ dres, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               int accumulate, int Cb, int HW, long N, int relu) {
  __shared__ float sm[6][8];
  __shared__ double ptot[16];
  const int cb = blockIdx.y;
  const bool has_y = y != nullptr;
  u32x4 dcv[U], xcv[U], ycv[U];       // requested before the prologue, as in blk_bn_apply_kernel (one uniform branch, no load under a condition)
  if (has_y) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long n = ((long)blockIdx.x * U + u) * 256 + threadIdx.x;
      const size_t idx = cell_index(n < N ? n : 0, cb, Cb, HW);
      dcv[u] = dy[idx]; xcv[u] = x[idx]; ycv[u] = y[idx];
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long n = ((long)blockIdx.x * U + u) * 256 + threadIdx.x;
      const size_t idx = cell_index(n < N ? n : 0, cb, Cb, HW);
      dcv[u] = dy[idx]; xcv[u] = x[idx]; ycv[u] = u32x4{0u, 0u, 0u, 0u};
    }
  }
  float pm = 0.f, pr = 0.f, pg = 0.f, pb = 0.f;      // (this thread's channel parameters: requested before the partial sums)
  if (threadIdx.x < 8) { const int c = cb * 8 + threadIdx.x; pm = save_mean[c]; pr = save_rstd[c]; pg = gamma[c]; pb = beta[c]; }
  partial_sums16(part, S, cb, Cb, ptot);
  if (threadIdx.x < 8) {
    const int k = threadIdx.x, c = cb * 8 + k;
    const double s = ptot[k], q = ptot[8 + k];
    if (blockIdx.x == 0) {
      if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s;
      if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)q;
    }
    sm[0][k] = pm; sm[1][k] = pr; sm[2][k] = pg; sm[3][k] = pb;
    sm[4][k] = (float)(s / (double)N); sm[5][k] = (float)(q / (double)N);
  }
  __syncthreads();
  float mean[8], rstd[8], gam[8], bet[8], c1[8], c2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { mean[k] = sm[0][k]; rstd[k] = sm[1][k]; gam[k] = sm[2][k]; bet[k] = sm[3][k]; c1[k] = sm[4][k]; c2[k] = sm[5][k]; }
  float scm[8], shm[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { scm[k] = gam[k] * rstd[k]; shm[k] = bet[k] - mean[k] * scm[k]; }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long n = ((long)blockIdx.x * U + u) * 256 + threadIdx.x;
    if (n >= N) continue;
    const size_t idx = cell_index(n, cb, Cb, HW);
    float g[8], xh[8], o[8];
    bn_bwd_g(dcv[u], xcv[u], has_y, ycv[u], mean, rstd, scm, shm, relu, g, xh);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = gam[k] * rstd[k] * (g[k] - c1[k] - xh[k] * c2[k]);
    dx[idx] = cell_pack(o);
    if (dres) dres[idx] = cell_pack(g);
  }
}

// one block per channel block (small maps, see blk_bn_fwd_block_kernel)
template <int NPT, bool HAS_Y>
__global__ __launch_bounds__(512) void blk_bn_bwd_block_kernel(const u32x4* __restrict__ dy, const u32x4* __restrict__ x, const u32x4* __restrict__ y,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
                                                                u32x4* __restrict__ dx, u32x4* __restrict__ dres, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int accumulate, int Cb, int HW, int N, int relu) {
  __shared__ double tot[16];
  const int cb = blockIdx.x;
  constexpr bool has_y = HAS_Y;
  u32x4 dc[NPT], xc[NPT], yc[HAS_Y ? NPT : 1];
#pragma unroll
  for (int u = 0; u < NPT; ++u) {
    const int n = threadIdx.x + u * 512;
    const bool ok = n < N;
    const size_t id = cell_index(ok ? n : 0, cb, Cb, HW);
    dc[u] = ok ? dy[id] : u32x4{0u, 0u, 0u, 0u};
    xc[u] = x[id];
    if constexpr (HAS_Y) yc[u] = y[id];
  }
  float mean[8], rstd[8], gam[8], bet[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { mean[k] = save_mean[cb * 8 + k]; rstd[k] = save_rstd[cb * 8 + k]; gam[k] = gamma[cb * 8 + k]; bet[k] = beta[cb * 8 + k]; }
  float scm[8], shm[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { scm[k] = gam[k] * rstd[k]; shm[k] = bet[k] - mean[k] * scm[k]; }
  float a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = 0.f;
#pragma unroll
  for (int u = 0; u < NPT; ++u) {
    float g[8], xh[8];
    bn_bwd_g(dc[u], xc[u], has_y, yc[HAS_Y ? u : 0], mean, rstd, scm, shm, relu, g, xh);
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] += g[k]; a[8 + k] += g[k] * xh[k]; }
  }
  block_sum16_lds<8>(a, tot);
  if (threadIdx.x < 8) {
    const int ch = cb * 8 + threadIdx.x;
    if (dbeta) dbeta[ch] = (accumulate ? dbeta[ch] : 0.f) + (float)tot[threadIdx.x];
    if (dgamma) dgamma[ch] = (accumulate ? dgamma[ch] : 0.f) + (float)tot[8 + threadIdx.x];
  }
  float c1[8], c2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { c1[k] = (float)(tot[k] / (double)N); c2[k] = (float)(tot[8 + k] / (double)N); }
#pragma unroll
  for (int u = 0; u < NPT; ++u) {
    if (threadIdx.x + u * 512 >= N) continue;
    float g[8], xh[8], o[8];
    bn_bwd_g(dc[u], xc[u], has_y, yc[HAS_Y ? u : 0], mean, rstd, scm, shm, relu, g, xh);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = gam[k] * rstd[k] * (g[k] - c1[k] - xh[k] * c2[k]);
    const size_t id = cell_index(threadIdx.x + u * 512, cb, Cb, HW);
    dx[id] = cell_pack(o);
    if (dres) dres[id] = cell_pack(g);
  }
}

// ---- spatial helpers for the strided layers (a stride-s conv = the stride-1 conv followed by this sub-sampling) ----
__global__ __launch_bounds__(256) void blk_subsample_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, int H, int W, int Ho, int Wo,
                                                            int stride, long cells) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= cells) return;
  const int ow = (int)(e % Wo);
  const long t = e / Wo;
  const int oh = (int)(t % Ho);
  const long bc = t / Ho;
  y[e] = x[((size_t)bc * H + oh * stride) * W + ow * stride];
}
// the transpose: dx[h][w] = dy[h / s][w / s] where both are multiples of s, zero elsewhere
__global__ __launch_bounds__(256) void blk_upscatter_kernel(const u32x4* __restrict__ dy, u32x4* __restrict__ dx, int H, int W, int Ho, int Wo,
                                                            int stride, long cells) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= cells) return;
  const int w = (int)(e % W);
  const long t = e / W;
  const int h = (int)(t % H);
  const long bc = t / H;
  const int oh = h / stride, ow = w / stride;
  const bool on = oh * stride == h && ow * stride == w && oh < Ho && ow < Wo;
  dx[e] = on ? dy[((size_t)bc * Ho + oh) * Wo + ow] : u32x4{0u, 0u, 0u, 0u};
}

// ------------------------------------------------------------------------------------------------
#define BLK_BN_BLOCK_MAX 4096      // cells per channel block up to which one 512-thread block does both passes
#include <stdlib.h>
static bool bn_block_ok() {        // RSIS_BLK_BN_BLOCK=0: always the two-level grid kernels (A/B)
  static const bool ok = !(getenv("RSIS_BLK_BN_BLOCK") && getenv("RSIS_BLK_BN_BLOCK")[0] == '0');
  return ok;
}
static void bn_splits(int Cb, long N, int& S, long& per) {
  // ~2048 blocks over the grid, at least `mincells` cells per block, at most 64 splits
  // (at least 1024 cells per block: re-measured in round 5, after the apply kernels stopped adding the partial sums serially (NOTES (27)) --
  //  2048: 12.10 / 12.54 ms per bf16 224^2 step on two boxes, 1024: 11.99 / 12.47, 512: 12.35; configs[4] geometry unchanged)
  static const long mincells = getenv("RSIS_BLK_BN_MINCELLS") ? atol(getenv("RSIS_BLK_BN_MINCELLS")) : 1024;
  long s = 2048 / (Cb > 0 ? Cb : 1);
  if (s < 1) s = 1;
  const long smax = (N + mincells - 1) / mincells;
  if (s > smax) s = smax;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  per = (N + s - 1) / s;
  per = (per + 255) / 256 * 256;
  S = (int)((N + per - 1) / per);
}

// scratch: >= rsis_blk_bn_scratch_doubles(C) doubles
long rsis_l_blk_bn_scratch(int C) { return (long)64 * C * 2; }

int rsis_l_blk_bn_fwd(const void* x, const void* res, void* y, double* scratch, const float* gamma, const float* beta, float* run_mean,
                      float* run_var, float* save_mean, float* save_rstd, int B, int C, int HW, float eps, float momentum, int relu, int train,
                      hipStream_t st) {
  const int Cb = C >> 3;
  const long N = (long)B * HW;
  if (N >= (1L << 31)) return RSIS_ERR_UNSUPPORTED;      // (cell_index divides in 32 bits)
  int S = 1;
  long per = N;
  if (train && N <= BLK_BN_BLOCK_MAX && bn_block_ok()) {
#define BN_FB(NPT) hipLaunchKernelGGL((blk_bn_fwd_block_kernel<NPT>), dim3(Cb), dim3(512), 0, st, (const u32x4*)x, (const u32x4*)res, (u32x4*)y, \
                                     gamma, beta, run_mean, run_var, save_mean, save_rstd, Cb, HW, (int)N, eps, momentum, relu)
    if (N <= 2048) BN_FB(4); else BN_FB(8);
#undef BN_FB
    return rsis_check_launch();
  }
  if (train) {
    bn_splits(Cb, N, S, per);
    hipLaunchKernelGGL(blk_bn_stats_kernel, dim3(S, Cb), dim3(256), 0, st, (const u32x4*)x, scratch, Cb, HW, N, per);
  }
  constexpr int U = 4;
  hipLaunchKernelGGL((blk_bn_apply_kernel<U>), dim3((unsigned)((N + 256 * U - 1) / (256 * U)), Cb), dim3(256), 0, st, (const u32x4*)x,
                     (const u32x4*)res, (u32x4*)y, scratch, S, gamma, beta, run_mean, run_var, save_mean, save_rstd, Cb, HW, N, eps, momentum,
                     relu, train);
  return rsis_check_launch();
}

int rsis_l_blk_bn_bwd(const void* dy, const void* x, const void* y, double* scratch, const float* gamma, const float* beta,
                      const float* save_mean, const float* save_rstd, void* dx, void* dres, float* dgamma, float* dbeta, int accumulate,
                      int B, int C, int HW, int relu, hipStream_t st) {
  const int Cb = C >> 3;
  const long N = (long)B * HW;
  if (N >= (1L << 31)) return RSIS_ERR_UNSUPPORTED;
  if (N <= BLK_BN_BLOCK_MAX && bn_block_ok()) {
#define BN_BB(NPT, HY) hipLaunchKernelGGL((blk_bn_bwd_block_kernel<NPT, HY>), dim3(Cb), dim3(512), 0, st, (const u32x4*)dy, (const u32x4*)x, \
                                         (const u32x4*)y, gamma, beta, save_mean, save_rstd, (u32x4*)dx, (u32x4*)dres, dgamma, dbeta, accumulate, Cb, HW, (int)N, relu)
    if (y) { if (N <= 2048) BN_BB(4, true); else BN_BB(8, true); }
    else { if (N <= 2048) BN_BB(4, false); else BN_BB(8, false); }
#undef BN_BB
    return rsis_check_launch();
  }
  int S = 1;
  long per = N;
  bn_splits(Cb, N, S, per);
  hipLaunchKernelGGL(blk_bn_bwd_reduce_kernel, dim3(S, Cb), dim3(256), 0, st, (const u32x4*)dy, (const u32x4*)x, (const u32x4*)y, gamma, beta,
                     save_mean, save_rstd, scratch, Cb, HW, N, per, relu);
  constexpr int U = 4;
  hipLaunchKernelGGL((blk_bn_bwd_apply_kernel<U>), dim3((unsigned)((N + 256 * U - 1) / (256 * U)), Cb), dim3(256), 0, st, (const u32x4*)dy,
                     (const u32x4*)x, (const u32x4*)y, gamma, beta, save_mean, save_rstd, scratch, S, (u32x4*)dx, (u32x4*)dres, dgamma, dbeta,
                     accumulate, Cb, HW, N, relu);
  return rsis_check_launch();
}

int rsis_l_blk_subsample(const void* x, void* y, long BCb, int H, int W, int Ho, int Wo, int stride, hipStream_t st) {
  const long cells = BCb * Ho * Wo;
  hipLaunchKernelGGL(blk_subsample_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, (const u32x4*)x, (u32x4*)y, H, W, Ho, Wo,
                     stride, cells);
  return rsis_check_launch();
}
int rsis_l_blk_upscatter(const void* dy, void* dx, long BCb, int H, int W, int Ho, int Wo, int stride, hipStream_t st) {
  const long cells = BCb * H * W;
  hipLaunchKernelGGL(blk_upscatter_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, (const u32x4*)dy, (u32x4*)dx, H, W, Ho, Wo,
                     stride, cells);
  return rsis_check_launch();
}
