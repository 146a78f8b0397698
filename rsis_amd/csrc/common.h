// Internal helpers shared by the HIP translation units of librsis_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RSIS_OK 0
#define RSIS_ERR_ARG 1
#define RSIS_ERR_LAUNCH 2
#define RSIS_ERR_UNSUPPORTED 3

#define RSIS_KPAD 32        // packed-weight K axis is zero-padded to a multiple of this (BK of every kernel divides it)
#define RSIS_CK 8            // input channels per LDS chunk of the direct 3x3 kernel (packed per concat source)
#define RSIS_CKB3 16         // bf16 path: input channels per LDS chunk (and per packed chunk) of the 3x3 kernel
#define RSIS_CKB1 64         // ... of the 1x1 GEMM kernel
#define RSIS_LDW_ALIGN 128  // packed-weight row stride is a multiple of this many floats
#define RSIS_MAX_SRC 3      // channel-concatenated sources / split destinations per conv

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int rsis_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? RSIS_OK : RSIS_ERR_LAUNCH;
}

// Zero-fill on a stream with a KERNEL instead of hipMemsetAsync: inside a captured hipGraph a memset becomes a memset NODE, and
// replaying graphs that contain them back to back corrupted the iteration on this stack (ROCm 7.0 / gfx950).
int rsis_zero_async(void* p, size_t bytes, hipStream_t st);

// rsis_set_deterministic (api.hip): 1 = every reduction of the library runs in a fixed order (no grid split-K, one block per
// reduced address, no same-address atomics from more than one contributor) -- bit-reproducible run to run, slower.
int rsis_deterministic();

static inline int rsis_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline int rsis_roundup(int a, int b) { return ((a + b - 1) / b) * b; }

// Eval-mode BatchNorm (reference: nn.BatchNorm2d in eval(), model.py:50-54 and the torchvision trunk) as a per-channel affine map.
// ONE definition for the stand-alone launch (pointwise.hip: bn_apply_kernel) and for the conv epilogues that fold it in at inference
// (conv_igemm.hip / conv3x3_direct.hip, ConvArgs::ep_*): both produce the same bits by construction.
__device__ __forceinline__ void rsis_bn_affine(float rstd, float gamma, float beta, float mean, float& sc, float& sh) {
  sc = rstd * gamma;
  sh = __builtin_fmaf(-mean, sc, beta);
}
__device__ __forceinline__ float rsis_bn_eval_rstd(float var, float eps) { return 1.0f / sqrtf(var + eps); }
__device__ __forceinline__ float rsis_bn_apply(float v, float sc, float sh, float r, int relu) {
  const float t = __builtin_fmaf(v, sc, sh) + r;
  return relu ? fmaxf(t, 0.f) : t;
}
__device__ __forceinline__ float rsis_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// The same two functions on the hardware's transcendental units (v_exp_f32, v_rcp_f32: ~1 ulp each), for the bf16 (blk) decoder
// kernels: 4 / 5 instructions instead of ~25 (expf + an IEEE division) / ~35 (tanhf).  The ConvLSTM cell of the 112 x 112 level has
// 216 MACs per gate row and 5 of these per hidden channel and pixel: with the library functions the gate launch was bound by its
// epilogue's VALU (rocprofv3: 1600 VALU instructions per wave, 47 % of the wave's lifetime issuing), not by HBM.  Absolute error
// ~2e-7, far below the bf16 rounding of everything these values feed except the fp32 cell state (where it is the fp32 noise floor).
__device__ __forceinline__ float rsis_sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float rsis_tanh_fast(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.0f; }

// align_corners=True bilinear source coordinate (nn.UpsamplingBilinear2d: model.py:149,163): output index o reads inputs i0, i1
// with weights (1 - l1, l1).  One definition for every kernel that must agree bit for bit on which inputs an output touches.
__device__ __forceinline__ void ac_coord(int o, float scale, int in, int& i0, int& i1, float& l1) {
  const float src = scale * o;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - i0;
}
static inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

// Arguments of the implicit-GEMM convolution kernels (NCHW fp32).
// GEMM view:  D[co][px] = sum_k  Wp[k][co] * Xcol[k][px],   px = (b, ho, wo),  k = (segment, ci, r, s).
struct ConvArgs {
  const float* src[RSIS_MAX_SRC];  // gathered tensors, each [B][C[s]][H][W]
  int C[RSIS_MAX_SRC];
  int nsrc;
  int K;                           // sum(C[0..nsrc-1]) * ks * ks  (rows of the packed weights actually walked)
  int B, H, W;                     // gathered-tensor geometry
  int Ho, Wo;                      // output geometry
  int stride, pad, sshift;         // sshift = log2(stride) (dgrad mode)
  const float* wp;                 // packed weights [roundup(Ktot, RSIS_KPAD)][ldw]
  int ldw;
  int Cout;                        // real number of output rows
  const float* bias;               // [Cout] (packed row order) or null
  const float* addend;             // optional, same layout as dst[0] (single destination only)
  // inference: the eval-mode BatchNorm behind the conv in its epilogue (conv_blk.hip; fp32: the single-destination epilogues of
  // conv_igemm.hip / conv3x3_direct.hip, rsis_conv2d_fwd_bn_eval, where ep_round is unused: fp32 has no intermediate rounding) -- y = relu?(v * sc + sh (+ addend)) with
  // sc = gamma * rsqrtf(var + eps), sh = beta - mean * sc, v = the product rounded to bf16 first (ep_round: the bits of the separate
  // BatchNorm launch) or kept in fp32 (one rounding in all).  ep_gamma == null: none.
  const float* ep_gamma; const float* ep_beta; const float* ep_mean; const float* ep_var;
  float ep_eps;
  int ep_relu, ep_round;
  float* dst[RSIS_MAX_SRC];        // output tensors, rows split by Cd[]; each [B][Cd[i]][Ho][Wo]
  int Cd[RSIS_MAX_SRC];
  int ndst;
  int n_co_tiles, n_px_tiles;
  int ostride, oH, oW;             // EPI_PLAIN of the igemm kernel: output pixel (ho,wo) is stored at (ho*ostride, wo*ostride) of an oH x oW map
  int ksplit;                      // direct 3x3 kernel: split the channel chunks over gridDim.y blocks (atomics into a zeroed output)
  // ConvLSTM epilogue (EPI_LSTM): rows are gate-interleaved, row = 4*j + gate, gate in (i,f,o,g)
  int hid;
  const float* c_prev;             // [B][hid][Ho][Wo] or null (zero state)
  float* h_out;                    // [B][hid][Ho][Wo]
  float* c_out;                    // [B][hid][Ho][Wo]
  float* act_out;                  // [B][4*hid][Ho][Wo] post-nonlinearity gates (interleaved rows) or null
  unsigned long long* side_key;    // [B][hid] or null: global max-pool of h fused into the epilogue (rsis_side_key, zeroed by the caller)
  int precise;                     // direct 3x3 kernel: sum the reduction in segments whatever its length (inference calls; conv3x3_direct.hip)
};

// ---- global max-pool of the hidden state (reference model.py:143) fused into the ConvLSTM epilogue ----
// key = (order-preserving bits of the value) << 32 | (0x7FFFFFFF - pixel index): the LARGEST key is the largest value and, among
// equal values, the smallest pixel index (the first maximum in scan order); 0 = "no value yet" (every real key is > 0).  Blocks
// combine their candidates with a 64-bit atomic max on [B][hid] keys -- order-independent, hence bit-reproducible.
__device__ __forceinline__ unsigned long long rsis_side_key(float v, int idx) {
  unsigned u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (unsigned)(0x7FFFFFFF - idx);
}
__device__ __forceinline__ float rsis_side_value(unsigned long long k) {
  unsigned u = (unsigned)(k >> 32);
  u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  return __uint_as_float(u);
}
__device__ __forceinline__ int rsis_side_index(unsigned long long k) { return 0x7FFFFFFF - (int)(unsigned)(k & 0xFFFFFFFFull); }
// max over the 32 lanes of a half wave (xor offsets < 32 stay inside the half)
__device__ __forceinline__ unsigned long long rsis_key_max32(unsigned long long k) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_xor(k, o, 64);
    k = t > k ? t : k;
  }
  return k;
}

// Max-combine the four packed side keys of a half wave (hidden channels j0, j0 + 2, j0 + 4, j0 + 6 of image row `row` = side_key + b * hid)
// into their slots.  Same-address atomics serialise in L2 (~40 ns each), and the finest level of a 512 x 1024 input sends 2048 of them
// to every slot (8 images: 64 slots for 4096 blocks) -- ~100 us on a launch whose other work takes 145.  A slot only ever grows, so a key
// that does not beat the value a load returns cannot change it: with `check` the four slots are loaded first (one round trip, device-
// coherent) and the read-modify-write is issued only for keys that beat them -- after the first blocks almost none does.  A stale load
// only costs an unnecessary atomic; the result is the same maximum in any order (still bit-reproducible).  The round trip costs a short
// block ~2 us of tail, so `check` is for maps with many blocks per image only (RSIS_SIDE_CHECK_TILES); measured with the check always on:
// configs[4] geometry 34.97 -> 33.85 ms per step, but bf16 224^2 13.06 -> 14.06 and fp32 256^2 37.80 -> 38.62.
#define RSIS_SIDE_CHECK_TILES 192
__device__ __forceinline__ void rsis_side_key_max4(unsigned long long* row, int j0, int hid, const unsigned long long* k, bool lane_ok, bool check) {
  unsigned long long cur[4] = {0ull, 0ull, 0ull, 0ull};
  if (check && lane_ok) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) cur[r4] = __hip_atomic_load(row + (j0 + 2 * r4 < hid ? j0 + 2 * r4 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4)
    if (lane_ok && k[r4] != 0ull && j0 + 2 * r4 < hid && k[r4] > cur[r4]) atomicMax(row + j0 + 2 * r4, k[r4]);
}

// ... and the form the gate kernels use since round 5: the half waves of a BLOCK first park their keys in LDS, the block combines them,
// and ONE global atomic per hidden channel and block goes out -- 8 x fewer.  What it buys is set by the slot, not by the traffic: global
// same-address atomics complete one after the other (~40-50 ns each), and with one per half wave the 112 x 112 level of a 224 x 224 batch
// sent 448 to every slot -- 22 us on a launch of 24 (bf16), found when the bench's roofline leg started to launch with the keys as the
// product does.  park: LDS, nwave x 8 slots of 8 bytes that nobody else touches any more (no initialisation needed: every half wave
// writes its four slots, zeros included); EVERY thread of the block must call (it holds a barrier).  A wave covers the 8 hidden
// channels wm * 8 .. + 7 (slot 2 r4 + hi); the WGN waves wm * WGN .. + WGN - 1 share them.
__device__ __forceinline__ void rsis_side_key_block(unsigned long long* park, int wave, int WGM, int WGN, int hi, const unsigned long long* kk,
                                                    bool lane_ok, unsigned long long* row, int jbase, int hid, bool check) {
  if (lane_ok) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) park[wave * 8 + 2 * r4 + hi] = kk[r4];
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < WGM * 8) {
    const int wm = t >> 3, c = t & 7;
    unsigned long long k = 0ull;
    for (int wn = 0; wn < WGN; ++wn) {
      const unsigned long long v = park[(wm * WGN + wn) * 8 + c];
      k = v > k ? v : k;
    }
    const int j = jbase + t;
    if (k != 0ull && j < hid) {
      const unsigned long long cur = check ? __hip_atomic_load(row + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      if (k > cur) atomicMax(row + j, k);
    }
  }
}

// Arguments of the split-K weight-gradient kernel (conv_wgrad.hip).
struct WgradArgs {
  const float* dy;   // [B][CoutDy][Ho][Wo]   (for ConvLSTM: the gate pre-activation grads, interleaved rows)
  const float* x;    // [B][Cs][H][W]
  float* dw;         // [Cout][ldo]  reference layout, ldo = Ctot*ks*ks
  int B, Cs, H, W, Ho, Wo, Cout;
  int stride, pad;
  int ldo, n_off;    // row stride of dW, column offset of this source (= channel offset * ks*ks)
  int interleave_hid;  // >0: dy rows are 4*j+gate -> dW row gate*hid + j
  int chunk;         // px per split (multiple of BKW)
  int n_co_tiles, n_n_tiles;
  int blk;           // bf16 kernels only: dy and x are channel-blocked bf16 tensors (conv_blk.hip)
};

// ---- the recurrent decoder on channel-blocked bf16 tensors (conv_blk_dec.hip, blk_dec.hip; include/rsis_hip.h rsis_blk_*_batch) ----
// One 3x3 / stride 1 / pad 1 conv over the channel concat of <= 3 blk sources ([B][C/8][H][W][8] bf16, C % 8 == 0), bf16 pack of
// pack.hip (16-channel chunks per source).  Plain epilogue: (+ fp32 bias in packed row order) (+ blk addend) -> <= 2 blk destinations
// splitting the output channels (Cd % 8 == 0).  LSTM epilogue (hid > 0): rows are gate-interleaved (4 j + gate); addend = the
// time-invariant gate term (blk, 4 hid channels), c_prev / c_out fp32 NCHW, h_out blk (hid % 8 == 0), act_out blk (4 hid channels).
struct BlkConvJob {
  const void* src[RSIS_MAX_SRC];
  int C[RSIS_MAX_SRC];
  int nsrc;
  int B, H, W;
  const void* wp;
  int ldw;
  int Cout;
  const float* bias;
  const void* addend;
  void* dst[2];
  int Cd[2];
  int ndst;
  int hid;
  const float* c_prev;
  float* c_out;
  void* h_out;
  void* act_out;
  unsigned long long* side_key;
  int n_co_tiles, n_px_tiles;
};
// align-corners bilinear resize of a blk tensor (planes = B * C / 8 cell planes); backward: dx = resize^T(dy) (+ dpool[b][c] at pixel
// arg[b][c] of every channel plane: the side max-pool's gradient; both [B][C] arrays, null = none)
struct BlkResizeJob {
  const void* src;       // forward: x [planes][Hi][Wi][8]; backward: dy [planes][Ho][Wo][8]
  void* dst;             // forward: y [planes][Ho][Wo][8]; backward: dx [planes][Hi][Wi][8]
  const float* dpool;
  const int* arg;
  int planes, Hi, Wi, Ho, Wo;
};
// ConvLSTM pointwise backward on blk tensors: dh / dh2 [B][hid/8][HW][8] (dh2 may be null), act / da [B][4 hid / 8][HW][8] (rows
// 4 j + gate), c / c_prev / dc_next / dc_prev fp32 [B][hid][HW] (c_prev, dc_next, dc_prev may be null)
struct BlkLstmBwdJob {
  const void* dh;
  const void* dh2;
  const float* dc_next;
  const void* act;
  const float* c_prev;
  const float* c;
  void* da;
  float* dc_prev;
  int B, hid, HW;
};
