// EXPERIMENT (round 6): Winograd F(2x2, 3x3) for the 3x3 / stride 1 / pad 1 convs of ResNet-101 layer 3
// (256 -> 256 channels on 16 x 16 maps, B = 32: torchvision Bottleneck conv2 via reference src/modules/vision.py:16-19) on the exact-f32
// MFMA, fully fused: input transform in the staging path, 16 batched [tiles x Cin] . [Cin x Cout] products on v_mfma_f32_32x32x2_f32,
// output transform in the epilogue.  Standalone: builds its own inputs, checks against a float64 direct convolution on the host,
// times the launch.  Not part of librsis_hip.so.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o wino_f32 wino_f32.hip && ./wino_f32 [iters]
//
// Block = ONE image (16 x 16 outputs = 8 x 8 Winograd tiles of 2 x 2) x 32 output channels, 4 waves.  Wave w owns ROW w of the 4 x 4
// Winograd domain (xi = 4 w + c, c = 0..3): 4 positions x (32 output channels x 64 tiles) = 8 accumulator tiles of 32 x 32 (128 AGPR/VGPRs),
// so the row half of the output transform (M A) stays in registers and only the column half (A^T .) crosses waves through LDS.
// Per chunk of 8 input channels:  raw patch (8 x 18 x 18, halo zero-filled by the buffer descriptor) and the chunk's pre-transformed
// weights U[xi][ci][co] (16 KB) arrive by LDS-DMA two / one chunks ahead; the 4 waves transform raw(t+1) -> V[xi][ci][tile] (B^T d B,
// 32 adds per 16 values) WHILE they run the 32 MFMAs of chunk t.  One barrier per chunk.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_vp_t;

#define CK 8
#define HW 16                 // map side
#define PW 18                 // patch side (map + halo)
#ifndef RSTRIDE
#define RSTRIDE 18            // LDS row stride of the raw patch (floats)
#endif
constexpr int RAW_E = CK * PW * RSTRIDE;                 // elements of one raw stage
constexpr int RAW_N = (RAW_E + 255) / 256;               // dword DMA loads per thread per chunk
constexpr int RAW_S = RAW_N * 256;                       // padded stage (the DMA writes whole 64-lane rows)
constexpr int U_S = 16 * CK * 32;                        // floats of one weight stage
constexpr int V_S = 16 * CK * 64;                        // floats of one transformed-input stage
constexpr int LDS_FLOATS = 2 * RAW_S + 2 * U_S + 2 * V_S;

struct WinoArgs {
  const float* x;      // [B][C][16][16]
  const float* U;      // [Cout/32][C/8][16][8][32]
  float* y;            // [B][Cout][16][16]
  int B, C, Cout;
};

__global__ __launch_bounds__(256) void wino_f32_kernel(const WinoArgs p) {
#if __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const RAW = lds;
  float* const US = lds + 2 * RAW_S;
  float* const VS = US + 2 * U_S;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef NQ_FORCE
  const int n_co = p.Cout / 32, nq = NQ_FORCE;
#else
  const int n_co = p.Cout / 32, nq = p.C / CK;
#endif
  // blocks b, b + 8, ... share an XCD: an XCD takes B/8 images x all output-channel tiles (the 4 MB of U stay in its L2)
  const int bid = blockIdx.x, xcd = bid & 7, qq = bid >> 3;
  const int co_t = qq % n_co;
  const int img = xcd * (p.B / 8) + qq / n_co;
  if (img >= p.B) return;

  // loop-invariant DMA offsets
  unsigned xvo[RAW_N];
#pragma unroll
  for (int i = 0; i < RAW_N; ++i) {
    const int e = tid + i * 256;
    const int cl = e / (PW * RSTRIDE), rem = e - cl * (PW * RSTRIDE);
    const int py = rem / RSTRIDE, px = rem - py * RSTRIDE;
    const int gy = py - 1, gx = px - 1;
    const bool ok = e < RAW_E && (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
    xvo[i] = ok ? (unsigned)(cl * HW * HW + gy * HW + gx) * 4u : 0x7FFFFFF0u;
  }
  const float* xb = p.x + (size_t)img * p.C * HW * HW;
  const float* ub = p.U + (size_t)co_t * nq * U_S;

#define ISSUE_RAW(Q, SLOT)                                                                                              \
  {                                                                                                                     \
    const int q_ = (Q) < nq ? (Q) : 0;  /* beyond the last chunk: a zero-range descriptor (the loads write zeros into a dead stage) */ \
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)(xb + (size_t)q_ * CK * HW * HW), 0, (Q) < nq ? CK * HW * HW * 4 : 0, 0x00020000); \
    float* dst = RAW + (SLOT) * RAW_S + wave * 64;                                                                      \
    _Pragma("unroll") for (int i = 0; i < RAW_N; ++i)                                                                   \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_, (lds_vp_t)(dst + i * 256), 4, xvo[i], 0, 0, 0);                     \
  }
#define ISSUE_U(Q, SLOT)                                                                                                \
  {                                                                                                                     \
    const int q_ = (Q) < nq ? (Q) : 0;                                                                                  \
    const __amdgpu_buffer_rsrc_t ru_ = __builtin_amdgcn_make_buffer_rsrc((void*)(ub + (size_t)q_ * U_S), 0, (Q) < nq ? U_S * 4 : 0, 0x00020000); \
    float* dst = US + (SLOT) * U_S + wave * 256;                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                       \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ru_, (lds_vp_t)(dst + i * 1024), 16, (unsigned)(tid + i * 256) * 16u, 0, 0, 0); \
  }
#define LAND() __builtin_amdgcn_s_waitcnt(0x0F70);

  // input transform of one (tile, channel) item: V = B^T d B, written to V[xi][ci][tile]
  const int ty = lane >> 3, tx = lane & 7;
  const int roff = (2 * ty) * RSTRIDE + 2 * tx;          // window origin inside a channel's patch
  auto transform = [&](const float* raw, float* vs, int ci) {
    const float* s = raw + ci * (PW * RSTRIDE) + roff;
    float d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x2 a = *(const f32x2*)(s + r * RSTRIDE), b = *(const f32x2*)(s + r * RSTRIDE + 2);
      d[r][0] = a[0]; d[r][1] = a[1]; d[r][2] = b[0]; d[r][3] = b[1];
    }
    float e[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      e[0][c] = d[0][c] - d[2][c];
      e[1][c] = d[1][c] + d[2][c];
      e[2][c] = d[2][c] - d[1][c];
      e[3][c] = d[1][c] - d[3][c];
    }
    float* o = vs + ci * 64 + lane;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[(4 * r + 0) * (CK * 64)] = e[r][0] - e[r][2];
      o[(4 * r + 1) * (CK * 64)] = e[r][1] + e[r][2];
      o[(4 * r + 2) * (CK * 64)] = e[r][2] - e[r][1];
      o[(4 * r + 3) * (CK * 64)] = e[r][1] - e[r][3];
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.f;

  ISSUE_RAW(0, 0)
  ISSUE_U(0, 0)
  if (nq > 1) ISSUE_RAW(1, 1)
  LAND()
  __syncthreads();
  transform(RAW, VS, wave);
  transform(RAW, VS, wave + 4);
  __syncthreads();

  const int aoff = (4 * wave) * (CK * 32) + hi * 32 + l31;       // U[xi = 4 wave + c][2 kp + hi][l31]
  const int boff = (4 * wave) * (CK * 64) + hi * 64 + l31;       // V[xi][2 kp + hi][tg * 32 + l31]
  // The chunk loop, scheduled by hand (sched_barrier fences; inside a fence-delimited piece the order is the source order):
  // the 8 MFMAs of k-pair kp are issued in 4 pairs, and in the shadow of each pair (2 x 64 cycles of the matrix pipe) the wave
  // issues a slice of everything else -- the operand reads of k-pair kp + 1, and a quarter of the input transform of chunk t + 1
  // (item 0 = channel `wave` in pieces 0-1, item 1 = channel `wave + 4` in pieces 2-3: reads + row pass, then column pass + writes).
  // The transform runs unconditionally (in the last iteration on stale data into a dead stage): one basic block.
#define SB() __builtin_amdgcn_sched_barrier(0)
#define LOAD_OPS(KP, S)                                                                    \
  _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                          \
    av[S][c] = Us[c * (CK * 32) + (2 * (KP)) * 32];                                        \
    bv[S][c][0] = Vs[c * (CK * 64) + (2 * (KP)) * 64];                                     \
    bv[S][c][1] = Vs[c * (CK * 64) + (2 * (KP)) * 64 + 32];                                \
  }
#ifdef NO_MFMA
#define MF(S, c) acc[c][0][0] += av[S][c] * bv[S][c][0]; acc[c][1][0] += av[S][c] * bv[S][c][1];
#else
#define MF(S, c)                                                                           \
  acc[c][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[S][c], bv[S][c][0], acc[c][0], 0, 0, 0); \
  acc[c][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[S][c], bv[S][c][1], acc[c][1], 0, 0, 0);
#endif
  float av[2][4], bv[2][4][2];
  for (int t = 0; t < nq; ++t) {
    const int cur = t & 1, nxt = cur ^ 1;
    const float* Us = US + cur * U_S + aoff;
    const float* Vs = VS + cur * V_S + boff;
    LOAD_OPS(0, 0)
#ifndef NO_DMA
    ISSUE_RAW(t + 2, cur)
    ISSUE_U(t + 1, nxt)
#endif
    const float* raw = RAW + nxt * RAW_S + roff;
    float* vo = VS + nxt * V_S + lane;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int ci = wave + 4 * it;
      const float* sp = raw + ci * (PW * RSTRIDE);
      float* o = vo + ci * 64;
      float d[4][4], e[4][4];
      // ---- piece 2 it: k-pair 2 it; transform reads + row pass ----
      SB();
      LOAD_OPS(2 * it + 1, 1)
      MF(0, 0)
      SB();
#ifdef NO_TRANSFORM
#pragma unroll
      for (int r = 0; r < 4; ++r) { d[r][0] = d[r][1] = d[r][2] = d[r][3] = (float)t; }
#else
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x2 a = *(const f32x2*)(sp + r * RSTRIDE), b = *(const f32x2*)(sp + r * RSTRIDE + 2);
        d[r][0] = a[0]; d[r][1] = a[1]; d[r][2] = b[0]; d[r][3] = b[1];
      }
#endif
      MF(0, 1)
      SB();
#pragma unroll
      for (int c = 0; c < 2; ++c) { e[0][c] = d[0][c] - d[2][c]; e[1][c] = d[1][c] + d[2][c]; e[2][c] = d[2][c] - d[1][c]; e[3][c] = d[1][c] - d[3][c]; }
      MF(0, 2)
      SB();
#pragma unroll
      for (int c = 2; c < 4; ++c) { e[0][c] = d[0][c] - d[2][c]; e[1][c] = d[1][c] + d[2][c]; e[2][c] = d[2][c] - d[1][c]; e[3][c] = d[1][c] - d[3][c]; }
      MF(0, 3)
      SB();
      // ---- piece 2 it + 1: k-pair 2 it + 1; column pass + writes ----
      if (it == 0) { LOAD_OPS(2, 0) }
      MF(1, 0)
      SB();
#ifdef NO_TRANSFORM
      if (e[0][0] == 12345.f)
#endif
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        o[(4 * r + 0) * (CK * 64)] = e[r][0] - e[r][2];
        o[(4 * r + 1) * (CK * 64)] = e[r][1] + e[r][2];
        o[(4 * r + 2) * (CK * 64)] = e[r][2] - e[r][1];
        o[(4 * r + 3) * (CK * 64)] = e[r][1] - e[r][3];
      }
      MF(1, 1)
      SB();
#ifdef NO_TRANSFORM
      if (e[0][0] == 12345.f)
#endif
#pragma unroll
      for (int r = 2; r < 4; ++r) {
        o[(4 * r + 0) * (CK * 64)] = e[r][0] - e[r][2];
        o[(4 * r + 1) * (CK * 64)] = e[r][1] + e[r][2];
        o[(4 * r + 2) * (CK * 64)] = e[r][2] - e[r][1];
        o[(4 * r + 3) * (CK * 64)] = e[r][1] - e[r][3];
      }
      MF(1, 2)
      MF(1, 3)
      SB();
    }
    LAND()
    __syncthreads();
  }

  // ---- output transform.  Row half in registers: T[w][0] = m0 + m1 + m2, T[w][1] = m1 - m2 - m3 (m_c = M[w][c]); parked in LDS
  // (the V stages are dead), then wave w' finishes rows 4 w' .. 4 w' + 3 of the accumulator tiles: Y[0][j] = T0j + T1j + T2j,
  // Y[1][j] = T1j - T2j - T3j, stored as float2 rows of the 2 x 2 output tile ----
  float* Ts = VS;                                              // [w][j][tg][r][lane]: 4 * 2 * 2 * 16 * 64 floats = 64 KB
#pragma unroll
  for (int tg = 0; tg < 2; ++tg)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0][tg][r], m1 = acc[1][tg][r], m2 = acc[2][tg][r], m3 = acc[3][tg][r];
      Ts[(((wave * 2 + 0) * 2 + tg) * 16 + r) * 64 + lane] = m0 + m1 + m2;
      Ts[(((wave * 2 + 1) * 2 + tg) * 16 + r) * 64 + lane] = m1 - m2 - m3;
    }
  __syncthreads();
  float* yb = p.y + ((size_t)img * p.Cout + co_t * 32) * HW * HW;
#pragma unroll
  for (int tg = 0; tg < 2; ++tg) {
    const int tile = tg * 32 + l31, oy = 2 * (tile >> 3), ox = 2 * (tile & 7);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = 4 * wave + rr;
      const int co = rr + 8 * wave + 4 * hi;
      float t[4][2];
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int j = 0; j < 2; ++j) t[w][j] = Ts[(((w * 2 + j) * 2 + tg) * 16 + r) * 64 + lane];
      f32x2 y0, y1;
      y0[0] = t[0][0] + t[1][0] + t[2][0]; y0[1] = t[0][1] + t[1][1] + t[2][1];
      y1[0] = t[1][0] - t[2][0] - t[3][0]; y1[1] = t[1][1] - t[2][1] - t[3][1];
      *(f32x2*)(yb + (size_t)co * HW * HW + oy * HW + ox) = y0;
      *(f32x2*)(yb + (size_t)co * HW * HW + (oy + 1) * HW + ox) = y1;
    }
  }
#endif
}

// ---- host ----
static void pack_U(const std::vector<float>& w, std::vector<float>& U, int Cout, int C) {
  const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  const int nq = C / CK;
  U.assign((size_t)(Cout / 32) * nq * U_S, 0.f);
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < C; ++ci) {
      const float* g = &w[((size_t)co * C + ci) * 9];
      double t[4][3], u[4][4];
      for (int i = 0; i < 4; ++i)
        for (int s = 0; s < 3; ++s) t[i][s] = G[i][0] * g[0 * 3 + s] + G[i][1] * g[1 * 3 + s] + G[i][2] * g[2 * 3 + s];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) u[i][j] = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
      for (int xi = 0; xi < 16; ++xi)
        U[(((size_t)(co / 32) * nq + ci / CK) * 16 + xi) * (CK * 32) + (ci % CK) * 32 + (co % 32)] = (float)u[xi / 4][xi % 4];
    }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 50;
  const int B = 32, C = 256, Cout = 256;
  std::vector<float> x((size_t)B * C * 256), w((size_t)Cout * C * 9), U;
  srand(1);
  for (auto& v : x) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto& v : w) v = ((float)rand() / RAND_MAX * 2.f - 1.f) / 48.f;
  pack_U(w, U, Cout, C);
  float *dx, *dU, *dy;
  hipMalloc(&dx, x.size() * 4); hipMalloc(&dU, U.size() * 4); hipMalloc(&dy, (size_t)B * Cout * 256 * 4);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dU, U.data(), U.size() * 4, hipMemcpyHostToDevice);
  hipMemset(dy, 0xFF, (size_t)B * Cout * 256 * 4);
  WinoArgs a{dx, dU, dy, B, C, Cout};
  const int grid = (Cout / 32) * B, lds = LDS_FLOATS * 4;
  if (hipFuncSetAttribute((const void*)wino_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { printf("lds attr failed\n"); return 1; }
  hipLaunchKernelGGL(wino_f32_kernel, dim3(grid), dim3(256), lds, 0, a);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  std::vector<float> y((size_t)B * Cout * 256);
  hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost);
  // float64 direct conv of images 0, 13 and B-1
  double worst = 0, worst_direct32 = 0, ymax = 0;
  for (int b : {0, 13, B - 1})
    for (int co = 0; co < Cout; co += 3)
      for (int oy = 0; oy < 16; ++oy)
        for (int ox = 0; ox < 16; ++ox) {
          double s = 0; float s32 = 0.f;
          for (int ci = 0; ci < C; ++ci)
            for (int r = 0; r < 3; ++r)
              for (int q = 0; q < 3; ++q) {
                const int iy = oy + r - 1, ix = ox + q - 1;
                if ((unsigned)iy < 16u && (unsigned)ix < 16u) {
                  const float xv = x[((size_t)b * C + ci) * 256 + iy * 16 + ix], wv = w[((size_t)co * C + ci) * 9 + r * 3 + q];
                  s += (double)xv * wv; s32 += xv * wv;
                }
              }
          const double e = fabs(s - y[((size_t)b * Cout + co) * 256 + oy * 16 + ox]);
          worst = e > worst ? e : worst;
          worst_direct32 = fabs(s - s32) > worst_direct32 ? fabs(s - s32) : worst_direct32;
          ymax = fabs(s) > ymax ? fabs(s) : ymax;
        }
  printf("winograd F(2x2,3x3) f32: max |y - f64 direct| = %.3e (a sequential fp32 direct sum: %.3e), max|y| %.3f\n", worst, worst_direct32, ymax);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(wino_f32_kernel, dim3(grid), dim3(256), lds, 0, a);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(wino_f32_kernel, dim3(grid), dim3(256), lds, 0, a);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = 1e3 * ms / iters, gf_direct = 2.0 * B * 256 * C * 9 * Cout / 1e9;
  printf("wino_f32_kernel: %.1f us per launch (%d blocks, %d KB LDS); direct-conv-equivalent %.1f TFLOP/s, executed (x 16/36) %.1f TFLOP/s\n",
         us, grid, lds / 1024, gf_direct / us * 1e3, gf_direct * 16 / 36 / us * 1e3);
  return worst < 2e-4 ? 0 : 2;
}
