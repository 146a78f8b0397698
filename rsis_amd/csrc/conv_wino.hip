// Winograd F(2x2, 3x3) convolution on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950, NCHW fp32, 3x3 / stride 1 / pad 1,
// fully fused: input transform in the staging path, 16 batched [tiles x Cin] . [Cin x Cout] products on the matrix cores, output
// transform in the epilogue.  No transformed tensor ever touches HBM.  2.25x fewer MFMA flops than the direct kernel
// (conv3x3_direct.hip), which runs AT the f32 MFMA roofline in its loop (NOTES (13)): the one lever left on the fp32 trunk.
//
// Used for the 3x3 convs of the ResNet-101 bottlenecks (torchvision Bottleneck.conv2 through reference src/modules/vision.py:16-19;
// 22 of the trunk's 33 are 256 -> 256 on the 1/16-scale map) -- forward and data gradient (the same kernel on weights that were
// flipped and transposed before the weight transform: pack.hip modes 7 / 8).  Selected by RSIS_DTYPE_F32_WINO in the packed copy.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per 2 x 2 output tile, 4 x 4 input window d, summed over input channels
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1]
//
// Block = one 16 x 16 output REGION of one image (8 x 8 Winograd tiles; maps that are not multiples of 16 get partial regions,
// masked at the store) x 32 output channels, 8 waves (two per SIMD: one wave's LDS / DMA / barrier waits hide behind the other's
// MFMAs -- a 4-wave version spent 44 of its 65 us outside the MFMAs, a wave cannot issue an MFMA while it sits in s_waitcnt).
// Wave w owns the Winograd positions xi = 4 r + 2 h + c (r = w >> 1 the row, h = w & 1 the column half, c = 0..1): 2 positions x
// (32 output channels x 64 tiles) = 4 accumulator tiles.  Per chunk of 8 input channels
//   * the raw 18 x 18 patch (halo and out-of-map pixels zero-filled by the buffer descriptor) and the chunk's pre-transformed weights
//     U[xi][ci][co] (16 KB, contiguous in the packed copy) arrive by LDS-DMA through a 3-deep ring: the chunk issued in iteration t is
//     first read in iteration t + 2, the end-of-chunk wait is vmcnt(DMA instructions of one chunk), never vmcnt(0);
//   * every thread transforms ONE (tile, channel) window of chunk t + 1 (B^T d B: 32 adds) into V[xi][ci][tile] while the wave runs
//     the 16 MFMAs of chunk t;  one s_barrier per chunk (NOT __syncthreads: its fence would drain the DMA ring).
// Epilogue: the row half of A^T M A in registers, the column half across the 8 waves through LDS (two rounds of 64 KB in the dead V
// stages), + bias + addend, masked stores through a buffer descriptor.
// Accuracy (tools/exp/wino, 256 -> 256 on 16 x 16): max |y - float64| 1.1e-6 against 2.4e-6 for a sequential fp32 direct sum --
// each of the 16 products is a 256-deep chain instead of one 2304-deep chain.
#include "common.h"

typedef __attribute__((address_space(3))) void* lds_vp_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define WCK 8                 // input channels per chunk
#define WRG 16                // region side (outputs)
#define WPW 18                // patch side
#define WNW 8                 // waves per block
#define WNT (WNW * 64)
#define WRING 3

namespace {
constexpr int RAW_E = WCK * WPW * WPW;                    // elements of one raw stage
constexpr int RAW_N = (RAW_E + WNT - 1) / WNT;            // dword DMA loads per thread per chunk
constexpr int RAW_S = RAW_N * WNT;                        // padded stage (the DMA writes whole 64-lane rows)
constexpr int U_S = 16 * WCK * 32;                        // floats of one weight stage = one (co tile, chunk) block of the packed copy
constexpr int U_N = U_S / 4 / WNT;                        // float4 DMA loads per thread per chunk
constexpr int V_S = 16 * WCK * 64;                        // floats of one transformed-input stage
constexpr int LDS_FLOATS = WRING * RAW_S + WRING * U_S + 2 * V_S;
constexpr int DMA_PER_ITER = RAW_N + U_N;

struct WinoArgs {
  const float* x;        // [B][C][H][W]
  const float* U;        // [Cout/32][C/8][16][8][32]  (pack.hip modes 7 / 8)
  float* y;              // [B][C0][H][W]: output channels [0, C0)
  float* y1;             // [B][Cout - C0][H][W]: output channels [C0, Cout) (the second destination of a split data gradient) or null
  const float* bias;     // [Cout] or null
  const float* addend;   // [B][Cout][H][W] or null (single destination only)
  int B, C, Cout, C0, H, W;
  int ry, rx;            // regions per image
  int n_regions;         // B * ry * rx
};
// grouped launch: up to 4 independent convs in one grid (the gate data gradients of the levels of one reverse wavefront diagonal that
// take the Winograd kernel); jobs by value in the kernel arguments, block b belongs to the last job whose begin <= b
#define WINO_MAXJ 4
struct WinoGroup {
  int n;
  int begin[WINO_MAXJ + 1];
  WinoArgs job[WINO_MAXJ];
};
}  // namespace

// FLUSH (inference calls): every WINO_FLUSH chunks (32 input channels) the MFMA accumulators are added to a second register set and cleared --
// the segmented accumulation of conv3x3_direct.hip (its NOTES (13) / round-4 reason: an MFMA accumulates its k-steps as ONE chain).
#define WINO_FLUSH 4
template <bool FLUSH>
__global__ __launch_bounds__(WNT) void conv_wino_f32_kernel(const WinoGroup g) {
#if __HIP_DEVICE_COMPILE__
  int jb = 0;
#pragma unroll
  for (int k = 1; k < WINO_MAXJ; ++k) jb += (k < g.n && g.begin[k] <= (int)blockIdx.x) ? 1 : 0;
  const WinoArgs& p = g.job[jb];
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const RAW = lds;
  float* const US = lds + WRING * RAW_S;
  float* const VS = US + WRING * U_S;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_co = p.Cout >> 5, nq = p.C / WCK;
  const int H = p.H, W = p.W, HW = H * W;
  // blocks b, b + 8, ... share an XCD: an XCD owns a contiguous range of regions x every output-channel tile, so the packed
  // weights (Cout x C x 64 bytes: 4 MB for 256 -> 256) are fetched into its L2 once and a region's input stays there for its n_co blocks
  const int bid = (int)blockIdx.x - g.begin[jb], xcd = bid & 7, qq = bid >> 3;
  const int co_t = qq % n_co;
  const int reg = xcd * ((p.n_regions + 7) >> 3) + qq / n_co;
  if (reg >= p.n_regions) return;
  const int img = reg / (p.ry * p.rx), rr_ = reg - img * (p.ry * p.rx);
  const int y0 = (rr_ / p.rx) * WRG, x0 = (rr_ % p.rx) * WRG;

  unsigned xvo[RAW_N];
#pragma unroll
  for (int i = 0; i < RAW_N; ++i) {
    const int e = tid + i * WNT;
    const int cl = e / (WPW * WPW), rem = e - cl * (WPW * WPW);
    const int py = rem / WPW, px = rem - py * WPW;
    const int gy = y0 + py - 1, gx = x0 + px - 1;
    const bool ok = e < RAW_E && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    xvo[i] = ok ? (unsigned)(cl * HW + gy * W + gx) * 4u : 0x7FFFFFF0u;
  }
  const float* xb = p.x + (size_t)img * p.C * HW;
  const float* ub = p.U + (size_t)co_t * nq * U_S;

  // beyond the last chunk: a zero-range descriptor (the loads write zeros into a dead stage) -- no branch around the issue
#define WINO_ISSUE_RAW(Q, SLOT)                                                                                         \
  {                                                                                                                     \
    const int q_ = (Q) < nq ? (Q) : 0;                                                                                  \
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)(xb + (size_t)q_ * WCK * HW), 0, (Q) < nq ? WCK * HW * 4 : 0, 0x00020000); \
    float* dst = RAW + (SLOT) * RAW_S + wave * 64;                                                                      \
    _Pragma("unroll") for (int i = 0; i < RAW_N; ++i)                                                                   \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_, (lds_vp_t)(dst + i * WNT), 4, xvo[i], 0, 0, 0);                     \
  }
#define WINO_ISSUE_U(Q, SLOT)                                                                                           \
  {                                                                                                                     \
    const int q_ = (Q) < nq ? (Q) : 0;                                                                                  \
    const __amdgpu_buffer_rsrc_t ru_ = __builtin_amdgcn_make_buffer_rsrc((void*)(ub + (size_t)q_ * U_S), 0, (Q) < nq ? U_S * 4 : 0, 0x00020000); \
    float* dst = US + (SLOT) * U_S + wave * 256;                                                                        \
    _Pragma("unroll") for (int i = 0; i < U_N; ++i)                                                                     \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ru_, (lds_vp_t)(dst + i * WNT * 4), 16, (unsigned)(tid + i * WNT) * 16u, 0, 0, 0); \
  }
  // s_waitcnt: vmcnt(N) (+ lgkmcnt(0) in the second form), expcnt untouched
#define WINO_WAIT_VM(N) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N) & 15) | (((N) >> 4) << 14));
#define WINO_WAIT_VM_LGKM0(N) __builtin_amdgcn_s_waitcnt(0x0070 | ((N) & 15) | (((N) >> 4) << 14));

  const int ty = lane >> 3, tx = lane & 7;
  const int roff = wave * (WPW * WPW) + (2 * ty) * WPW + 2 * tx;     // this thread's transform item: channel `wave` of the chunk, tile `lane`
  const int wrow = wave >> 1, whalf = wave & 1;
  const int xi0 = 4 * wrow + 2 * whalf;
  const int aoff = xi0 * (WCK * 32) + hi * 32 + l31;       // U[xi0 + c][2 kp + hi][l31]
  const int boff = xi0 * (WCK * 64) + hi * 64 + l31;       // V[xi0 + c][2 kp + hi][tg * 32 + l31]

  f32x16 acc[2][2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.f;

#define WINO_T_LOAD(RAWP)                                                                        \
  {                                                                                              \
    const float* sp_ = (RAWP) + roff;                                                            \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                              \
      const f32x2 a_ = *(const f32x2*)(sp_ + r * WPW), b_ = *(const f32x2*)(sp_ + r * WPW + 2);  \
      d[r][0] = a_[0]; d[r][1] = a_[1]; d[r][2] = b_[0]; d[r][3] = b_[1];                        \
    }                                                                                            \
  }
#define WINO_T_STORE(VP)                                                                         \
  {                                                                                              \
    float e_[4][4];                                                                              \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                              \
      e_[0][c] = d[0][c] - d[2][c]; e_[1][c] = d[1][c] + d[2][c]; e_[2][c] = d[2][c] - d[1][c]; e_[3][c] = d[1][c] - d[3][c]; \
    }                                                                                            \
    float* o_ = (VP) + wave * 64 + lane;                                                         \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                              \
      o_[(4 * r + 0) * (WCK * 64)] = e_[r][0] - e_[r][2];                                        \
      o_[(4 * r + 1) * (WCK * 64)] = e_[r][1] + e_[r][2];                                        \
      o_[(4 * r + 2) * (WCK * 64)] = e_[r][2] - e_[r][1];                                        \
      o_[(4 * r + 3) * (WCK * 64)] = e_[r][1] - e_[r][3];                                        \
    }                                                                                            \
  }
  float d[4][4];
  WINO_ISSUE_RAW(0, 0)
  WINO_ISSUE_U(0, 0)
  WINO_ISSUE_RAW(1, 1)
  WINO_WAIT_VM(0)
  __builtin_amdgcn_s_barrier();
  WINO_T_LOAD(RAW)
  WINO_T_STORE(VS)
  WINO_ISSUE_RAW(2, 2)          // the group an "iteration -1" would have issued
  WINO_ISSUE_U(1, 1)
  WINO_WAIT_VM_LGKM0(DMA_PER_ITER)
  __builtin_amdgcn_s_barrier();

#define WINO_SB() __builtin_amdgcn_sched_barrier(0)
#define WINO_LOAD_OPS(KP, S)                                                               \
  _Pragma("unroll") for (int c = 0; c < 2; ++c) {                                          \
    av[S][c] = Us[c * (WCK * 32) + (2 * (KP)) * 32];                                       \
    bv[S][c][0] = Vs[c * (WCK * 64) + (2 * (KP)) * 64];                                    \
    bv[S][c][1] = Vs[c * (WCK * 64) + (2 * (KP)) * 64 + 32];                               \
  }
#define WINO_MF(S, c)                                                                      \
  acc[c][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[S][c], bv[S][c][0], acc[c][0], 0, 0, 0); \
  acc[c][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[S][c], bv[S][c][1], acc[c][1], 0, 0, 0);
  float av[2][2], bv[2][2][2];
  int s3 = 0;                                   // t % WRING
  f32x16 tot[FLUSH ? 2 : 1][FLUSH ? 2 : 1];
  if constexpr (FLUSH) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[c][j][r] = 0.f;
  }
  for (int t = 0; t < nq; ++t) {
    if constexpr (FLUSH) {
      if (t > 0 && (t & (WINO_FLUSH - 1)) == 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { tot[c][j][r] += acc[c][j][r]; acc[c][j][r] = 0.f; }
      }
    }
    const int cur = t & 1, nxt = cur ^ 1;
    const int s3n = s3 + 1 == WRING ? 0 : s3 + 1;                  // (t + 1) % 3
    const int s3p = s3n + 1 == WRING ? 0 : s3n + 1;                // (t + 2) % 3
    const float* Us = US + s3 * U_S + aoff;
    const float* Vs = VS + cur * V_S + boff;
    WINO_LOAD_OPS(0, 0)
    WINO_T_LOAD(RAW + s3n * RAW_S)               // raw(t + 1) (in the last iteration: a stale stage, transformed into a dead one)
    WINO_ISSUE_RAW(t + 3, s3)                    // raw(t) was consumed in iteration t - 1
    WINO_ISSUE_U(t + 2, s3p)                     // U(t - 1)'s stage
    WINO_SB();
    WINO_LOAD_OPS(1, 1)
    WINO_MF(0, 0)
    WINO_MF(0, 1)
    WINO_SB();
    WINO_LOAD_OPS(2, 0)
    WINO_MF(1, 0)
    WINO_MF(1, 1)
    WINO_SB();
    WINO_T_STORE(VS + nxt * V_S)
    WINO_LOAD_OPS(3, 1)
    WINO_MF(0, 0)
    WINO_MF(0, 1)
    WINO_SB();
    WINO_MF(1, 0)
    WINO_MF(1, 1)
    WINO_SB();
    // this wave's LDS writes (the transform) are complete at lgkmcnt(0); of its DMA only the group just issued may still be in flight
    WINO_WAIT_VM_LGKM0(DMA_PER_ITER)
    __builtin_amdgcn_s_barrier();
    s3 = s3n;
  }
  WINO_WAIT_VM(0)      // (the zero-range tail loads still write LDS: they must have landed before the stages are reused below)
  __builtin_amdgcn_s_barrier();
  if constexpr (FLUSH) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][j][r] += tot[c][j][r];
  }

  // ---- output transform.  Row part in registers: with m_c = M[r][2 h + c], half 0 contributes P0 = m0 + m1, P1 = m1 and half 1
  // P0 = m0, P1 = -m0 - m1 to T[r][j] = (M A)[r][j]; then Y[0][j] = T0j + T1j + T2j, Y[1][j] = T1j - T2j - T3j.  The 8 waves park
  // P0 / P1 of one 32-tile group in LDS (64 KB: the V stages), wave w' finishes accumulator rows 2 w', 2 w' + 1; two rounds. ----
  float* Ps = VS;                                              // [w][j][r][lane]: 8 * 2 * 16 * 64 floats = 64 KB
  // destination of this block's 32 output channels (C0 is a multiple of 32: the choice is uniform over the block)
  const bool second = co_t * 32 >= p.C0;
  const int Cout = second ? p.Cout - p.C0 : p.C0;          // channels of the destination tensor
  const int cbase = second ? co_t * 32 - p.C0 : co_t * 32; // this block's first channel inside it
  const unsigned span = (unsigned)((size_t)p.B * Cout * HW * 4);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(second ? p.y1 : p.y), 0, span, 0x00020000);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.addend ? p.addend : p.y), 0, p.addend ? span : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? (const void*)p.bias : (const void*)p.y), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);
  float bvv[2];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int r = 2 * wave + rr;
    bvv[rr] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (unsigned)(co_t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * 4u, 0, 0));
  }
#pragma unroll
  for (int tg = 0; tg < 2; ++tg) {
    if (tg) { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); }      // round 0's reads are done before round 1 overwrites
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][tg][r], m1 = acc[1][tg][r];
      Ps[((wave * 2 + 0) * 16 + r) * 64 + lane] = whalf ? m0 : m0 + m1;
      Ps[((wave * 2 + 1) * 16 + r) * 64 + lane] = whalf ? -m0 - m1 : m1;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    const int tile = tg * 32 + l31, oy = y0 + 2 * (tile >> 3), ox = x0 + 2 * (tile & 7);
    unsigned vo[2][2][2];
    float T[2][4][2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = 2 * wave + rr;
      const int co = cbase + (r & 3) + 8 * (r >> 2) + 4 * hi;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          T[rr][q][j] = Ps[(((2 * q) * 2 + j) * 16 + r) * 64 + lane] + Ps[(((2 * q + 1) * 2 + j) * 16 + r) * 64 + lane];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          vo[rr][i][j] = (oy + i < H && ox + j < W) ? (unsigned)(((img * Cout + co) * H + oy + i) * W + ox + j) * 4u : 0x7FFFFFF0u;
    }
    float av2[2][2][2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) av2[rr][i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, vo[rr][i][j], 0, 0));
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float ya = T[rr][0][j] + T[rr][1][j] + T[rr][2][j] + bvv[rr] + av2[rr][0][j];
        const float yb = T[rr][1][j] - T[rr][2][j] - T[rr][3][j] + bvv[rr] + av2[rr][1][j];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ya), ro, vo[rr][0][j], 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yb), ro, vo[rr][1][j], 0, 0);
      }
  }
#endif
}

static int wino_fill(WinoArgs& a, const float* x, const void* U, const float* bias, const float* addend, float* y, float* y1, int B, int C,
                     int Cout, int C0, int H, int W) {
  if (!x || !U || !y || B < 1 || H < 1 || W < 1 || C < WCK || C % WCK != 0 || Cout < 32 || Cout % 32 != 0) return RSIS_ERR_ARG;
  if (C0 < 32 || C0 > Cout || C0 % 32 != 0 || (C0 < Cout && (!y1 || addend))) return RSIS_ERR_ARG;
  if ((size_t)B * Cout * H * W * 4 >= (1ull << 31) || (size_t)WCK * H * W * 4 >= (1ull << 31)) return RSIS_ERR_UNSUPPORTED;
  a.x = x; a.U = (const float*)U; a.y = y; a.y1 = y1; a.bias = bias; a.addend = addend;
  a.B = B; a.C = C; a.Cout = Cout; a.C0 = C0; a.H = H; a.W = W;
  a.ry = rsis_cdiv(H, WRG); a.rx = rsis_cdiv(W, WRG);
  a.n_regions = B * a.ry * a.rx;
  return RSIS_OK;
}
static int wino_launch(WinoGroup& g, int blocks, hipStream_t st, bool flush = false) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_wino_f32_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * 4) != hipSuccess) return RSIS_ERR_LAUNCH;
    if (hipFuncSetAttribute((const void*)conv_wino_f32_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * 4) != hipSuccess) return RSIS_ERR_LAUNCH;
    attr_set = true;
  }
  if (flush) hipLaunchKernelGGL(conv_wino_f32_kernel<true>, dim3(blocks), dim3(WNT), LDS_FLOATS * 4, st, g);
  else hipLaunchKernelGGL(conv_wino_f32_kernel<false>, dim3(blocks), dim3(WNT), LDS_FLOATS * 4, st, g);
  return rsis_check_launch();
}

// x [B][C][H][W] -> y [B][Cout][H][W] (+ bias, + addend); U: the Winograd packed copy.  C % 8 == 0, Cout % 32 == 0.
int rsis_launch_conv_wino(const float* x, const void* U, const float* bias, const float* addend, float* y, int B, int C, int Cout,
                          int H, int W, hipStream_t st, int precise) {
  WinoGroup g = {};
  const int rc = wino_fill(g.job[0], x, U, bias, addend, y, nullptr, B, C, Cout, Cout, H, W);
  if (rc) return rc;
  g.n = 1;
  const int blocks = 8 * (Cout / 32) * rsis_cdiv(g.job[0].n_regions, 8);
  for (int k = 1; k <= WINO_MAXJ; ++k) g.begin[k] = blocks;
  return wino_launch(g, blocks, st, precise != 0);
}

// n <= WINO_MAXJ independent convs in one grid, each with up to two destinations splitting its output channels at C0[j] (a multiple
// of 32): the gate data gradients d(up) | dh_prev of one reverse wavefront diagonal (rsis_conv2d_dgrad_batch)
int rsis_launch_conv_wino_group(int n, const float* const* x, const void* const* U, float* const* y, float* const* y1, const int* B, const int* C,
                                const int* Cout, const int* C0, const int* H, const int* W, hipStream_t st) {
  if (n < 1 || n > WINO_MAXJ) return RSIS_ERR_ARG;
  WinoGroup g = {};
  g.n = n;
  int blocks = 0;
  for (int j = 0; j < n; ++j) {
    const int rc = wino_fill(g.job[j], x[j], U[j], nullptr, nullptr, y[j], y1[j], B[j], C[j], Cout[j], C0[j], H[j], W[j]);
    if (rc) return rc;
    g.begin[j] = blocks;
    blocks += 8 * (Cout[j] / 32) * rsis_cdiv(g.job[j].n_regions, 8);
  }
  for (int k = n; k <= WINO_MAXJ; ++k) g.begin[k] = blocks;
  return wino_launch(g, blocks, st);
}
