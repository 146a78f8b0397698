// Inference post-processing of the predicted instance masks on the device (gfx950): what the reference does per mask on the
// host with scipy + pycocotools (reference src/eval.py:96-127 resize_mask: scipy.ndimage.zoom(order=1) to the image size,
// `> mask_th`, ignore pixels cleared, area test, pycocotools mask.encode; RLE semantics from src/coco/common/maskApi.c).
//   mask_resize_threshold_kernel: align-corners bilinear resample (== zoom order 1) of n probability maps + threshold, written
//     directly in COLUMN-major order (the order the run-length encoding walks), plus the per-mask area.  HBM-bound: one byte
//     out per pixel (two with the raw copy), the small probability map stays in cache.
//   rle_encode_kernel: run lengths of one column-major mask per block: per chunk the threads count value changes, a block scan
//     turns the counts into output slots for the change positions, and a last pass differences the positions in place.
//   rsis_rle_to_string (host): the 6-bit varint text form of the counts.
#include "common.h"

__device__ __forceinline__ void mp_coord(int o, float scale, int in, int& i0, int& i1, float& l1) {
  const float src = scale * o;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - i0;
}

// grid = (ceil(h*w / 256), n); element e of mask k is column-major: e = x * h + y
__global__ __launch_bounds__(256) void mask_resize_threshold_kernel(const float* __restrict__ prob, int Hm, int Wm,
                                                                    const unsigned char* __restrict__ ignore, float th,
                                                                    unsigned char* __restrict__ seg, unsigned char* __restrict__ raw,
                                                                    unsigned int* __restrict__ area, int h, int w, float sh,
                                                                    float sw) {
  const int k = blockIdx.y;
  const long hw = (long)h * w;
  const long e = blockIdx.x * 256L + threadIdx.x;
  unsigned int on = 0;
  if (e < hw) {
    const int x = (int)(e / h), y = (int)(e - (long)x * h);
    int y0, y1, x0, x1; float ly, lx;
    mp_coord(y, sh, Hm, y0, y1, ly);
    mp_coord(x, sw, Wm, x0, x1, lx);
    const float* p = prob + (size_t)k * Hm * Wm;
    const float v = (1.f - ly) * ((1.f - lx) * p[y0 * Wm + x0] + lx * p[y0 * Wm + x1]) +
                    ly * ((1.f - lx) * p[y1 * Wm + x0] + lx * p[y1 * Wm + x1]);
    const unsigned char r = v > th ? 1 : 0;
    unsigned char s = r;
    if (ignore && ignore[(size_t)y * w + x] == 1) s = 0;
    seg[(size_t)k * hw + e] = s;
    if (raw) raw[(size_t)k * hw + e] = r;
    on = s;
  }
  // block reduction of the area
  __shared__ unsigned int red[4];
  unsigned int v = on;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = red[0] + red[1] + red[2] + red[3];
    if (t) atomicAdd(area + k, t);
  }
}

#define RLE_T 1024   // threads per block
#define RLE_E 16     // consecutive elements per thread per chunk
__global__ __launch_bounds__(RLE_T) void rle_encode_kernel(const unsigned char* __restrict__ masks, long len,
                                                           unsigned int* __restrict__ counts, int cap, int* __restrict__ nruns) {
  const int k = blockIdx.x;
  const unsigned char* m = masks + (size_t)k * len;
  unsigned int* out = counts + (size_t)k * cap;
  __shared__ unsigned int wsum[RLE_T / 64];
  __shared__ unsigned int running;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) running = 0;
  __syncthreads();
  // ---- pass 1: positions of the value changes (the mask is preceded by an implicit 0) ----
  for (long base = 0; base < len; base += (long)RLE_T * RLE_E) {
    const long j0 = base + (long)tid * RLE_E;
    unsigned char v[RLE_E + 1];
    v[0] = (j0 > 0 && j0 - 1 < len) ? m[j0 - 1] : 0;
#pragma unroll
    for (int i = 0; i < RLE_E; ++i) {
      const bool ok = j0 + i < len;
      v[i + 1] = ok ? m[j0 + i] : v[i];
    }
    unsigned int c = 0;
#pragma unroll
    for (int i = 0; i < RLE_E; ++i) c += (v[i + 1] != v[i]) ? 1u : 0u;
    // exclusive scan over the block: wave scan by shuffles, wave totals through LDS
    unsigned int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    unsigned int woff = 0, total = 0;
#pragma unroll
    for (int i = 0; i < RLE_T / 64; ++i) {
      const unsigned int s = wsum[i];
      if (i < wv) woff += s;
      total += s;
    }
    unsigned int slot = running + woff + inc - c;
#pragma unroll
    for (int i = 0; i < RLE_E; ++i)
      if (v[i + 1] != v[i]) {
        if (slot < (unsigned)cap) out[slot] = (unsigned int)(j0 + i);
        ++slot;
      }
    __syncthreads();
    if (tid == 0) running += total;
    __syncthreads();
  }
  const unsigned int nchg = running;          // runs = nchg + 1 (the first run counts zeros, possibly 0 of them)
  if (tid == 0) nruns[k] = nchg + 1 <= (unsigned)cap ? (int)(nchg + 1) : -(int)(nchg + 1);
  if (nchg + 1 > (unsigned)cap) return;
  // ---- pass 2: positions -> run lengths, in place, from the last chunk down (chunk c needs the last position of chunk c-1) ----
  if (tid == 0) out[nchg] = (unsigned int)len - (nchg ? out[nchg - 1] : 0u);
  __syncthreads();
  for (long hi = nchg; hi > 0; hi -= RLE_T) {
    const long i = hi - 1 - tid;              // this chunk covers indices (hi - RLE_T, hi - 1]
    unsigned int cur = 0, prev = 0;
    if (i >= 0) { cur = out[i]; prev = i > 0 ? out[i - 1] : 0u; }
    __syncthreads();
    if (i >= 0) out[i] = cur - prev;
    __syncthreads();
  }
}

int rsis_l_mask_resize_threshold(const float* prob, int n, int Hm, int Wm, const unsigned char* ignore, float th, unsigned char* seg,
                                 unsigned char* raw, unsigned int* area, int h, int w, hipStream_t st) {
  if (rsis_zero_async(area, sizeof(unsigned int) * (size_t)n, st) != RSIS_OK) return RSIS_ERR_LAUNCH;
  const float sh = h > 1 ? (float)(Hm - 1) / (float)(h - 1) : 0.f, sw = w > 1 ? (float)(Wm - 1) / (float)(w - 1) : 0.f;
  const long hw = (long)h * w;
  hipLaunchKernelGGL(mask_resize_threshold_kernel, dim3((unsigned)((hw + 255) / 256), n), dim3(256), 0, st, prob, Hm, Wm, ignore, th,
                     seg, raw, area, h, w, sh, sw);
  return rsis_check_launch();
}

int rsis_l_rle_encode(const unsigned char* masks, int n, long len, unsigned int* counts, int cap, int* nruns, hipStream_t st) {
  hipLaunchKernelGGL(rle_encode_kernel, dim3(n), dim3(RLE_T), 0, st, masks, len, counts, cap, nruns);
  return rsis_check_launch();
}

// host: counts -> text (each count, for i > 2 minus counts[i-2], as a base-32 varint, 5 payload bits + continuation bit per
// character, least significant group first, sign-extended, offset 48); returns the string length or -1 if `cap` is too small
int rsis_l_rle_to_string(const unsigned int* counts, int m, char* out, int cap) {
  int p = 0;
  for (int i = 0; i < m; ++i) {
    long x = (long)counts[i];
    if (i > 2) x -= (long)counts[i - 2];
    bool more = true;
    while (more) {
      char c = (char)(x & 0x1f);
      x >>= 5;
      more = (c & 0x10) ? x != -1 : x != 0;
      if (more) c |= 0x20;
      if (p + 1 >= cap) return -1;
      out[p++] = (char)(c + 48);
    }
  }
  out[p] = 0;
  return p;
}

// ------------------------------------------------------------------------------------------------
// Largest 8-connected component of a binary mask (reference src/eval_cityscapes.py:131-150: skimage.measure.label with the
// default full connectivity, then the most frequent non-background label).  Union-find on the device, no host loop:
//   init (every foreground pixel is its own root) -> merge (each pixel unites with its W / NW / N / NE foreground neighbours;
//   roots are linked larger -> smaller with atomicMin, so the structure stays a forest and the result does not depend on
//   the order of the unions) -> flatten -> per-root pixel counts -> arg-max (ties: the component that comes first in raster
//   order; the reference's tie order is that of a python-2 dict, i.e. unspecified) -> select.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int cc_find(const int* L, int x) {
  int p = L[x];
  while (p != x) { x = p; p = L[x]; }
  return x;
}
__device__ __forceinline__ void cc_union(int* L, int a, int b) {
  while (true) {
    a = cc_find(L, a);
    b = cc_find(L, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }      // link the larger root a under the smaller root b
    const int old = atomicMin(&L[a], b);
    if (old == a) return;                                // a was still a root: linked
    a = old;                                             // someone re-parented a meanwhile: retry from its new parent
  }
}

__global__ void cc_init_kernel(const unsigned char* __restrict__ m, int* __restrict__ L, int* __restrict__ cnt, long total, int hw) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    L[e] = m[e] ? (int)(e % hw) : -1;
    cnt[e] = 0;
  }
}
__global__ void cc_merge_kernel(const unsigned char* __restrict__ m, int* __restrict__ L, long total, int h, int w) {
  const int hw = h * w;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    if (!m[e]) continue;
    const long k = e / hw;
    const int i = (int)(e - k * hw), y = i / w, x = i - y * w;
    const unsigned char* mk = m + k * hw;
    int* Lk = L + k * hw;
    if (x > 0 && mk[i - 1]) cc_union(Lk, i, i - 1);
    if (y > 0) {
      if (mk[i - w]) cc_union(Lk, i, i - w);
      if (x > 0 && mk[i - w - 1]) cc_union(Lk, i, i - w - 1);
      if (x + 1 < w && mk[i - w + 1]) cc_union(Lk, i, i - w + 1);
    }
  }
}
__global__ void cc_count_kernel(const unsigned char* __restrict__ m, int* __restrict__ L, int* __restrict__ cnt, long total, int hw) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    if (!m[e]) continue;
    const long k = e / hw;
    const int r = cc_find(L + k * hw, (int)(e - k * hw));
    L[e] = r;                                            // (a pixel's own entry is only read by finds that pass through it: any root-ward value is valid)
    atomicAdd(&cnt[k * hw + r], 1);
  }
}
// one block per mask: best[k] = root with the largest count (ties: smallest root index)
__global__ __launch_bounds__(256) void cc_argmax_kernel(const int* __restrict__ cnt, int* __restrict__ best, int hw) {
  const int k = blockIdx.x;
  int bc = 0, bi = -1;
  for (int i = threadIdx.x; i < hw; i += 256) {
    const int c = cnt[(size_t)k * hw + i];
    if (c > bc) { bc = c; bi = i; }                      // ascending i per thread: the first maximum wins
  }
  __shared__ int sc[256], si[256];
  sc[threadIdx.x] = bc; si[threadIdx.x] = bi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const int c2 = sc[threadIdx.x + o], i2 = si[threadIdx.x + o];
      if (c2 > sc[threadIdx.x] || (c2 == sc[threadIdx.x] && c2 > 0 && i2 < si[threadIdx.x])) { sc[threadIdx.x] = c2; si[threadIdx.x] = i2; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) best[k] = si[0];
}
__global__ void cc_select_kernel(const unsigned char* __restrict__ m, const int* __restrict__ L, const int* __restrict__ best,
                                 unsigned char* __restrict__ out, long total, int hw) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long k = e / hw;
    out[e] = (m[e] && L[e] == best[k]) ? 1 : 0;
  }
}

int rsis_l_largest_component(const unsigned char* mask, unsigned char* out, int* labels, int* counts, int* best, int n, int h, int w,
                             hipStream_t st) {
  const int hw = h * w;
  const long total = (long)n * hw;
  long g = (total + 255) / 256;
  if (g > 256L * 32) g = 256L * 32;
  const dim3 grid((unsigned)g), blk(256);
  hipLaunchKernelGGL(cc_init_kernel, grid, blk, 0, st, mask, labels, counts, total, hw);
  hipLaunchKernelGGL(cc_merge_kernel, grid, blk, 0, st, mask, labels, total, h, w);
  hipLaunchKernelGGL(cc_count_kernel, grid, blk, 0, st, mask, labels, counts, total, hw);
  hipLaunchKernelGGL(cc_argmax_kernel, dim3(n), blk, 0, st, counts, best, hw);
  hipLaunchKernelGGL(cc_select_kernel, grid, blk, 0, st, mask, labels, best, out, total, hw);
  return rsis_check_launch();
}
