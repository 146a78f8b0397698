// Soft-IoU matching scores and loss gradient of the RSIS training step on the device (gfx950).
//
// reference: src/train.py:98-110 builds, per timestep, the all-pairs cost  softIoU(y_mask[b, g], sigmoid(out_mask_t[b]))  by
// repeating every prediction gt_T times; src/utils/hungarian.py:62-89 (softIoU) reduces each pair; after the assignment the
// same sums are recomputed for the matched pairs (train.py:162-163) and autograd differentiates them.
//
// Here ONE pass over the logits and the ground-truth masks produces every sum at once as a [T+1] x [G+1] GEMM per image on
// the exact-f32 MFMA:   S[b][i][j] = sum_n A[i][n] * Bm[j][n],   A = [sigmoid(P[b, 0..T-1]); ones],  Bm = [Y[b, 0..G-1]; ones]
//   S[t][g] = intersection,  S[t][G] = sum_n p_t,  S[T][g] = sum_n y_g.
// The matched loss needs nothing else (its forward is a gather of S), and its backward is one elementwise kernel:
//   d cost / d logit_n = (ca * y_n + cb * (1 - y_n)) * p_n (1 - p_n),  ca = -g / U,  cb = g * I / U^2,  U = sum p + sum y - I + e.
// Both kernels are HBM-bound (252 MB in / 84 MB out at B=32, T=10, G=20, 256x256): bytes ~ (T + G) * N * 4 per image.
#include "common.h"

typedef const f32x4 __attribute__((address_space(1)))* gcf4_t;
typedef f32x4 __attribute__((address_space(1)))* gf4_t;

// grid = (nsplit, B), 256 threads = 4 waves; wave w of split s walks the pixels [k0, k1) in steps of 8 (N % 8 == 0):
// lanes 0-31 take pixels +0..3, lanes 32-63 pixels +4..7 of each step (the MFMA K index is only a summation index).
__global__ __launch_bounds__(256) void softiou_sums_kernel(const float* __restrict__ logits, const float* __restrict__ y,
                                                           float* __restrict__ S, int T, int G, long N, long per_wave) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long k0 = wid * per_wave, k1 = min(N, k0 + per_wave);
  if (k0 >= k1) return;
  const bool arow = l31 < T, aone = l31 == T;      // A rows: predictions, then the all-ones row
  const bool brow = l31 < G, bone = l31 == G;
  const float* pa = logits + ((size_t)b * T + (arow ? l31 : 0)) * N + 4 * hi;
  const float* pb = y + ((size_t)b * G + (brow ? l31 : 0)) * N + 4 * hi;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // Loads are UNCONDITIONAL (rows beyond T / G read row 0 and are selected away) and four steps are requested before the first is used:
  // `if (arow) v = load` compiled to branch + load + vmcnt(0) twice per 8-pixel step -- two dependent round trips per step, 120 us for a
  // pass whose bytes need 50 (NOTES (32a)).
  auto step = [&](const f32x4 va, const f32x4 vb) {
    f32x4 a, bv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] = arow ? rsis_sigmoid(va[e]) : (aone ? 1.f : 0.f);
      bv[e] = brow ? vb[e] : (bone ? 1.f : 0.f);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bv[e], acc, 0, 0, 0);
  };
  long k = k0;
  for (; k + 32 <= k1; k += 32) {
    f32x4 va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { va[u] = *(gcf4_t)(pa + k + 8 * u); vb[u] = *(gcf4_t)(pb + k + 8 * u); }
#pragma unroll
    for (int u = 0; u < 4; ++u) step(va[u], vb[u]);
  }
  for (; k < k1; k += 8) step(*(gcf4_t)(pa + k), *(gcf4_t)(pb + k));
  float* Sb = S + (size_t)b * (T + 1) * (G + 1);
  if (l31 <= G) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (i <= T && !(i == T && l31 == G)) atomicAdd(Sb + i * (G + 1) + l31, acc[r]);
    }
  }
}

// dlogits[b][t][n] = (ca[b][t] * y + cb[b][t] * (1 - y)) * p (1 - p),  y = Y[b][perm[b][t]][n],  p = sigmoid(logits[b][t][n])
__global__ __launch_bounds__(256) void softiou_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ y,
                                                          const long long* __restrict__ perm, int perm_ld,
                                                          const float* __restrict__ ca, const float* __restrict__ cb,
                                                          float* __restrict__ dlogits, int T, int G, long N4, long total) {
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long bt = e / N4, n4 = e - bt * N4;
    const int b = (int)(bt / T), t = (int)(bt - (long)b * T);
    const int g = (int)perm[(size_t)b * perm_ld + t];
    const f32x4 v = *(gcf4_t)(logits + (size_t)e * 4);
    const f32x4 yy = *(gcf4_t)(y + (((size_t)b * G + g) * N4 + n4) * 4);
    const float a = ca[bt], c = cb[bt];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float p = rsis_sigmoid(v[k]);
      o[k] = (a * yy[k] + c * (1.f - yy[k])) * p * (1.f - p);
    }
    *(gf4_t)(dlogits + (size_t)e * 4) = o;
  }
}

int rsis_l_softiou_sums(const float* logits, const float* y, float* S, int B, int T, int G, long N, hipStream_t st) {
  if (B < 0 || T < 0 || G < 0 || N < 0) return RSIS_ERR_ARG;
  if (B == 0) return RSIS_OK;
  if (rsis_zero_async(S, sizeof(float) * (size_t)B * (T + 1) * (G + 1), st) != RSIS_OK) return RSIS_ERR_LAUNCH;
  // no prediction or no ground-truth slot: every sum is zero, and the kernel's unconditional loads of row 0 (lanes beyond T / G
  // read it and discard the value) would touch a row that does not exist
  if (T == 0 || G == 0 || N == 0) return RSIS_OK;
  // ~4 blocks per CU over the whole batch; each wave gets a multiple of 8 pixels
  int nsplit = (int)((1024 + B - 1) / B);
  long per_wave = (N + (long)nsplit * 4 - 1) / ((long)nsplit * 4);
  per_wave = (per_wave + 7) / 8 * 8;
  if (per_wave < 64) per_wave = 64;
  if (rsis_deterministic()) per_wave = (N + 7) / 8 * 8;      // one wave per image walks every pixel: a single contributor per sum
  nsplit = (int)((N + per_wave * 4 - 1) / (per_wave * 4));
  hipLaunchKernelGGL(softiou_sums_kernel, dim3(nsplit, B), dim3(256), 0, st, logits, y, S, T, G, N, per_wave);
  return rsis_check_launch();
}

int rsis_l_softiou_bwd(const float* logits, const float* y, const long long* perm, int perm_ld, const float* ca, const float* cb,
                       float* dlogits, int B, int T, int G, long N, hipStream_t st) {
  const long total = (long)B * T * (N / 4);
  long grid = (total + 255) / 256;
  if (grid > 256L * 32) grid = 256L * 32;
  hipLaunchKernelGGL(softiou_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, st, logits, y, perm, perm_ld, ca, cb, dlogits, T, G,
                     N / 4, total);
  return rsis_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Class / stop heads of one decoder timestep (reference src/modules/model.py:169-182): side = cat of the five global-max-pool
// vectors (248 features at hidden 128) -> class_probs = softmax(fc_class(side)), stop = fc_stop(side).  ~20 tiny launches
// per timestep in eager form (cat, 2 addmm, softmax; backward: softmax_backward, 4 mm, 5 slice copies, 4 parameter adds): one
// block per image here, the concatenation is by pointer, the backward reduces the parameter gradients over the batch in-kernel.
// ------------------------------------------------------------------------------------------------
#define HEADS_MAXK 2048
#define HEADS_MAXC 64
struct HeadSides { const float* p[5]; float* d[5]; int C[5]; int n; const unsigned long long* key[5]; float* so[5]; int* ao[5]; };

// side features of image b -> sv.  key[i] set (forward only): source i comes as the packed (value, pixel) keys the ConvLSTM epilogue
// left with its atomic max (common.h rsis_side_key); decode them and write the float feature and the arg-max pixel the backward
// passes read (what rsis_global_maxpool_fwd would have written)
__device__ __forceinline__ void heads_load_side(const HeadSides& s, int b, float* sv) {
  int base = 0;
  for (int i = 0; i < s.n; ++i) {
    if (s.key[i]) {
      for (int k = threadIdx.x; k < s.C[i]; k += blockDim.x) {
        const unsigned long long kk = s.key[i][(size_t)b * s.C[i] + k];
        const float v = rsis_side_value(kk);
        sv[base + k] = v;
        s.so[i][(size_t)b * s.C[i] + k] = v;
        s.ao[i][(size_t)b * s.C[i] + k] = rsis_side_index(kk);
      }
    } else {
      for (int k = threadIdx.x; k < s.C[i]; k += blockDim.x) sv[base + k] = s.p[i][(size_t)b * s.C[i] + k];
    }
    base += s.C[i];
  }
}

__global__ __launch_bounds__(256) void heads_fwd_kernel(HeadSides s, int K, const float* __restrict__ Wc, const float* __restrict__ bc,
                                                        int ncls, const float* __restrict__ Ws, const float* __restrict__ bs,
                                                        float* __restrict__ probs, float* __restrict__ stop) {
  __shared__ float sv[HEADS_MAXK];
  __shared__ float lg[HEADS_MAXC + 1];
  const int b = blockIdx.x;
  heads_load_side(s, b, sv);
  __syncthreads();
  // one wave-quarter (16 lanes) per output row: rows 0..ncls-1 = fc_class, row ncls = fc_stop
  const int row = threadIdx.x >> 4, sub = threadIdx.x & 15;
  for (int r = row; r <= ncls; r += 16) {
    const float* w = r < ncls ? Wc + (size_t)r * K : Ws;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;     // independent chains: the (L2-latency) weight loads overlap
    for (int k = sub; k < K; k += 64) {
      const float w0 = w[k], w1 = k + 16 < K ? w[k + 16] : 0.f, w2 = k + 32 < K ? w[k + 32] : 0.f, w3 = k + 48 < K ? w[k + 48] : 0.f;
      a0 = fmaf(w0, sv[k], a0);
      a1 = fmaf(w1, k + 16 < K ? sv[k + 16] : 0.f, a1);
      a2 = fmaf(w2, k + 32 < K ? sv[k + 32] : 0.f, a2);
      a3 = fmaf(w3, k + 48 < K ? sv[k + 48] : 0.f, a3);
    }
    float acc = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_down(acc, o, 16);
    if (sub == 0) lg[r] = acc + (r < ncls ? bc[r] : bs[0]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = lg[0];
    for (int c = 1; c < ncls; ++c) m = fmaxf(m, lg[c]);
    float z = 0.f;
    for (int c = 0; c < ncls; ++c) { const float e = expf(lg[c] - m); lg[c] = e; z += e; }
    for (int c = 0; c < ncls; ++c) probs[(size_t)b * ncls + c] = lg[c] / z;
    stop[b] = lg[ncls];
  }
}

// Backward.  Block b writes d side of image b.  The parameter gradients are sums over the batch: instead of B-deep atomics on
// every one of the (ncls + 1) * K weights every block recomputes the (tiny) d-logits of ALL images and owns a slab of
// K / gridDim.x feature columns, which it reduces over the batch and adds to the gradient buffers with plain read-modify-writes.
// Everything a thread loops over is first staged in LDS with independent parallel loads, and the remaining global loads are
// issued 8 at a time: the first version (one dependent global load per loop iteration) took 37 us for ~0.5 MFLOP.
// (B = the rows of one launch: the batch, or -- decoder_seq -- the T * B rows of all timesteps at once; dl / gp / svs are carved
//  from dynamic LDS, 2 * B * (ncls + 1) + B * ceil(K / B) floats: the launcher refuses what does not fit 64 KB)
__global__ __launch_bounds__(256) void heads_bwd_kernel(HeadSides s, int K, int B, const float* __restrict__ Wc, int ncls,
                                                        const float* __restrict__ Ws, const float* __restrict__ probs,
                                                        const float* __restrict__ dprobs, const float* __restrict__ dstop,
                                                        float* __restrict__ dWc, float* __restrict__ dbc, float* __restrict__ dWs,
                                                        float* __restrict__ dbs) {
  extern __shared__ __attribute__((aligned(16))) float heads_lds[];
  const int b = blockIdx.x, R = ncls + 1;
  float* const dl = heads_lds;                         // [B][R]: probs, then d logits of fc_class | d stop, of every row
  float* const gp = dl + (size_t)B * R;                // [B][R]: dprobs
  float* const svs = gp + (size_t)B * R;               // side features of every row, this block's columns
#define DL(i, c) dl[(i) * R + (c)]
#define GP(i, c) gp[(i) * R + (c)]
  const int slab = (K + gridDim.x - 1) / gridDim.x, k0 = min(K, b * slab), k1 = min(K, k0 + slab), nk = k1 - k0;
  for (int e = threadIdx.x; e < B * ncls; e += blockDim.x) {
    const int i = e / ncls, c = e - i * ncls;
    DL(i, c) = probs[e];
    GP(i, c) = dprobs ? dprobs[e] : 0.f;
  }
  for (int e = threadIdx.x; e < B * nk; e += blockDim.x) {
    const int i = e / nk, k = k0 + e - i * nk;
    int sb = 0, si = 0;                                  // which side vector holds feature k
    while (k >= sb + s.C[si]) { sb += s.C[si]; ++si; }
    svs[e] = s.p[si][(size_t)i * s.C[si] + (k - sb)];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < B; i += blockDim.x) {   // softmax backward: p * (g - sum g p)
    float dot = 0.f;
    for (int c = 0; c < ncls; ++c) dot += GP(i, c) * DL(i, c);
    for (int c = 0; c < ncls; ++c) DL(i, c) = DL(i, c) * (GP(i, c) - dot);
    DL(i, ncls) = dstop ? dstop[i] : 0.f;
  }
  __syncthreads();
  // d side[k] = sum_c dl[c] * Wc[c][k] + dstop * Ws[k]   (coalesced along k)
  int base = 0;
  for (int i = 0; i < s.n; ++i) {
    for (int k = threadIdx.x; k < s.C[i]; k += blockDim.x) {
      const int kk = base + k;
      float acc = DL(b, ncls) * Ws[kk];
      for (int c0 = 0; c0 < ncls; c0 += 8) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = c0 + j < ncls ? Wc[(size_t)(c0 + j) * K + kk] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = fmaf(c0 + j < ncls ? DL(b, c0 + j) : 0.f, wv[j], acc);
      }
      if (s.d[i]) s.d[i][(size_t)b * s.C[i] + k] = acc;
    }
    base += s.C[i];
  }
  // parameter gradients of this block's feature columns [k0, k1)
  for (int e = threadIdx.x; e < R * nk; e += blockDim.x) {
    const int r = e / nk, kl = e - r * nk;
    if (r < ncls ? !dWc : !dWs) continue;
    float g = 0.f;
    for (int i = 0; i < B; ++i) g = fmaf(DL(i, r), svs[i * nk + kl], g);
    if (r < ncls) dWc[(size_t)r * K + k0 + kl] += g;
    else dWs[k0 + kl] += g;
  }
  if (b == 0 && threadIdx.x < R) {
    float g = 0.f;
    for (int i = 0; i < B; ++i) g += DL(i, threadIdx.x);
    if (threadIdx.x < ncls) { if (dbc) dbc[threadIdx.x] += g; }
    else if (dbs) dbs[0] += g;
  }
}
#undef DL
#undef GP

static int heads_sides(HeadSides& s, const float* const* side, float* const* dside, const int* C, int n) {
  if (n < 1 || n > 5) return -1;
  int K = 0;
  s.n = n;
  for (int i = 0; i < 5; ++i) {
    s.p[i] = i < n ? side[i] : nullptr; s.d[i] = (i < n && dside) ? dside[i] : nullptr; s.C[i] = i < n ? C[i] : 0; K += s.C[i];
    s.key[i] = nullptr; s.so[i] = nullptr; s.ao[i] = nullptr;
  }
  return K;
}

int rsis_l_heads_fwd(const float* const* side, const int* C, int n, int B, const float* Wc, const float* bc, int ncls, const float* Ws,
                     const float* bs, float* probs, float* stop, hipStream_t st, const unsigned long long* const* keys,
                     float* const* side_out, int* const* arg_out) {
  HeadSides s;
  const int K = heads_sides(s, side, nullptr, C, n);
  if (K < 1 || K > HEADS_MAXK || ncls < 1 || ncls > HEADS_MAXC) return RSIS_ERR_UNSUPPORTED;
  if (keys)
    for (int i = 0; i < n; ++i) { s.key[i] = keys[i]; s.so[i] = side_out[i]; s.ao[i] = arg_out[i]; }
  hipLaunchKernelGGL(heads_fwd_kernel, dim3(B), dim3(256), 0, st, s, K, Wc, bc, ncls, Ws, bs, probs, stop);
  return rsis_check_launch();
}

int rsis_l_heads_bwd(const float* const* side, const int* C, int n, int B, const float* Wc, int ncls, const float* Ws,
                     const float* probs, const float* dprobs, const float* dstop, float* const* dside, float* dWc, float* dbc,
                     float* dWs, float* dbs, hipStream_t st) {
  HeadSides s;
  const int K = heads_sides(s, side, dside, C, n);
  if (K < 1 || K > HEADS_MAXK || ncls < 1 || ncls > HEADS_MAXC || B < 1) return RSIS_ERR_UNSUPPORTED;
  const size_t lds = ((size_t)2 * B * (ncls + 1) + (size_t)B * ((K + B - 1) / B)) * sizeof(float);
  if (lds > 64 * 1024) return RSIS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(heads_bwd_kernel, dim3(B), dim3(256), lds, st, s, K, B, Wc, ncls, Ws, probs, dprobs, dstop, dWc, dbc, dWs, dbs);
  return rsis_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Loss tail of a training iteration (reference src/train.py:159-176 with utils/hungarian.py:10-59): the masked means of the
// class NLL, the matched soft-IoU costs and the balanced stop BCE, and their weighted sum -- ~100 tiny eager launches (gather,
// log, neg, where, sum, div, mul, ... and their autograd) folded into one single-block kernel each way.  n = B*T samples.
//   out[0] = total, out[1] = mean soft-IoU, out[2] = mean stop BCE, out[3] = mean class NLL   (train.py:189 order)
// bw < 0: the BCE balance weight is taken from the targets (positives / total, hungarian.py:45-49); cls_w: optional per-class
// weights multiplying log p (hungarian.py:24-27).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum256(float v, float* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void loss_tail_kernel(const float* __restrict__ probs, const long long* __restrict__ y_class,
                                                        const float* __restrict__ stop, const float* __restrict__ siou,
                                                        const float* __restrict__ sw_mask, const float* __restrict__ sw_class,
                                                        const float* __restrict__ cls_w, int n, int C, float bw, float w_iou,
                                                        float w_cls, float w_stop, float* __restrict__ out,
                                                        float* __restrict__ dprobs, float* __restrict__ dstop,
                                                        float* __restrict__ dsiou, const float* __restrict__ gout_p) {
  __shared__ float sh[4];
  const float gout = gout_p ? gout_p[0] : 1.f;
  float s_m = 0.f, s_c = 0.f, s_pos = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { s_m += sw_mask[i]; s_c += sw_class[i]; s_pos += sw_mask[i]; }
  const float sum_m = block_sum256(s_m, sh), sum_c = block_sum256(s_c, sh), npos = block_sum256(s_pos, sh);
  if (bw < 0.f) bw = npos / (float)n;                          // target of the stop loss is sw_mask (train.py:167)
  float a_iou = 0.f, a_cls = 0.f, a_stop = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float wm = sw_mask[i], wc = sw_class[i];
    const int yc = (int)y_class[i];
    const float p = probs[(size_t)i * C + yc];
    const float cw = cls_w ? cls_w[yc] : 1.f;
    const float nll = -logf(p) * cw;
    const float x = stop[i], t = sw_mask[i];
    const float mx = fmaxf(-x, 0.f);
    const float lv = x - x * t + mx + logf(expf(-mx) + expf(-x - mx));       // hungarian.py:51-52
    const float bal = (1.f - bw) * t + bw * (1.f - t);
    if (wm > 0.f) { a_iou += siou[i]; a_cls += nll; }            // masked_select(costs, sw) semantics: unselected rows never enter
    if (wc > 0.f) a_stop += lv * bal;
    if (dprobs) {                                                // backward (gout = upstream gradient of the total)
      dsiou[i] = wm > 0.f ? gout * w_iou / sum_m : 0.f;
      dstop[i] = wc > 0.f ? gout * w_stop / sum_c * bal * (1.f / (1.f + expf(-x)) - t) : 0.f;
      for (int c = 0; c < C; ++c) dprobs[(size_t)i * C + c] = 0.f;
      if (wm > 0.f) dprobs[(size_t)i * C + yc] = -gout * w_cls * cw / (sum_m * p);
    }
  }
  const float l_iou = block_sum256(a_iou, sh) / sum_m, l_cls = block_sum256(a_cls, sh) / sum_m, l_stop = block_sum256(a_stop, sh) / sum_c;
  if (threadIdx.x == 0 && out) {
    out[0] = w_iou * l_iou + w_cls * l_cls + w_stop * l_stop;
    out[1] = l_iou; out[2] = l_stop; out[3] = l_cls;
  }
}

int rsis_l_loss_tail(const float* probs, const long long* y_class, const float* stop, const float* siou, const float* sw_mask,
                     const float* sw_class, const float* cls_w, int n, int C, float bw, float w_iou, float w_cls, float w_stop, float* out,
                     float* dprobs, float* dstop, float* dsiou, const float* gout, hipStream_t st) {
  hipLaunchKernelGGL(loss_tail_kernel, dim3(1), dim3(256), 0, st, probs, y_class, stop, siou, sw_mask, sw_class, cls_w, n, C, bw, w_iou,
                     w_cls, w_stop, out, dprobs, dstop, dsiou, gout);
  return rsis_check_launch();
}
