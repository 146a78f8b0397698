/* librsis_hip.so -- C ABI of the MI355X (gfx950) RSIS hot path.
 *
 * The reference (imatge-upc/rsis) has no FFI: its hot path sits behind PyTorch nn.Modules
 * (src/modules/clstm.py, model.py, vision.py) that dispatch to cuDNN/THCUNN.  This header is the drop-in
 * boundary one level below that module surface: every entry point replaces the torch op(s) cited next to
 * it, on caller-owned device buffers, and is what the Python binding in rsis_amd/_lib.py (ctypes) loads.
 *
 * Conventions
 *   - every pointer is a device pointer to a caller-owned, contiguous fp32 NCHW tensor unless stated (also under
 *     RSIS_DTYPE_BF16: only the private packed weight copies are bf16);
 *   - no hidden allocation, no host synchronisation, graph-capture safe; work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream);
 *   - return value: 0 = OK, nonzero = error code (rsis_error_string); nothing throws across the ABI;
 *   - thread-safe / re-entrant per stream.
 *   - "packed" weights are a private MFMA-friendly copy ([K rows][Cout padded to 128 columns]; K order and padding
 *     depend on the kernel that consumes them: implicit-GEMM (k = ci,r,s padded to 32 rows), direct 3x3 (8-channel
 *     chunks per concat source, channel pairs interleaved per tap) or bf16 cells (8 consecutive input channels per 16-byte
 *     cell, 16- / 64-channel chunks per concat source); ConvLSTM rows gate-interleaved 4*j+gate);
 *     they are rebuilt from the reference-layout weight ([Cout][Cin][k][k], gate order i,f,o,g) and never
 *     serialised.
 */
#ifndef RSIS_HIP_H
#define RSIS_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RSIS_ABI_VERSION 2

/* Arithmetic of the MFMA kernels behind the conv / ConvLSTM entry points (the `dtype` arguments below).
 *   RSIS_DTYPE_F32 : exact-f32 MFMA (v_mfma_f32_32x32x2_f32), bit-for-bit an fp32 fmaf chain.
 *   RSIS_DTYPE_BF16: operands rounded to bf16 (round-to-nearest-even) when they are staged, fp32 accumulation
 *                    (v_mfma_f32_32x32x16_bf16).  Tensors stay fp32 NCHW in memory; only the packed weight copy is bf16.
 *                    Layers without a bf16 kernel (the 7x7 stem, 3x3 / stride 2, the 8 -> 1 channel conv_out, odd-shaped weight
 *                    gradients) run their f32 kernel under either dtype; rsis_conv_uses_bf16 tells which. */
#define RSIS_DTYPE_F32 0
#define RSIS_DTYPE_BF16 1
/* rsis_conv2d_wgrad / rsis_wgrad_job.dtype only: dy and x are channel-blocked bf16 tensors (the "blk" layout of the entry points at
 * the end of this header), dW is fp32 in the reference layout as always; ks in {1, 3}, stride 1, "same" padding, Cout % 8 == 0 and
 * Cs % 8 == 0 (RSIS_ERR_UNSUPPORTED otherwise) */
#define RSIS_DTYPE_BF16_BLK 2
/* Exact-f32 arithmetic as RSIS_DTYPE_F32, but a 3x3 / stride 1 / pad 1 conv whose input AND output channel counts are multiples of 32
 * (one source, no ConvLSTM rows: rsis_conv_uses_wino) runs as Winograd F(2x2, 3x3) on the f32 MFMA (conv_wino.hip: 2.25x fewer
 * matrix flops, transforms fused into the staging path and the epilogue).  The packed copies then hold the TRANSFORMED weights
 * (G g G^T, 16 values per (output, input) channel pair); rsis_conv2d_fwd / _dgrad pick the kernel from this dtype.  Every other
 * geometry, and the weight gradient, behaves exactly as under RSIS_DTYPE_F32.  Results differ from the direct kernel by fp32
 * rounding only (measured closer to float64 than a sequential fp32 sum: the 2304-deep chain becomes 16 chains of 256). */
#define RSIS_DTYPE_F32_WINO 3

int rsis_version(void);
const char* rsis_error_string(int code);
/* 1 when a conv of this geometry runs on the Winograd kernel under RSIS_DTYPE_F32_WINO */
int rsis_conv_uses_wino(int ks, int stride, int pad, int Cin, int Cout, int nseg, int lstm_hid);

/* Bit-reproducible mode (process-wide; initial value 1 when the environment has RSIS_DETERMINISTIC=1, else 0).  The reference's
 * CPU path is deterministic; this library's default is not: split-K sums, weight gradients and BatchNorm / bias / soft-IoU
 * reductions that span several blocks end in fp32 (fp64 for BatchNorm) atomics whose order varies run to run.  With the mode on
 * every such reduction has ONE contributor per address (no grid split-K, one block per reduced channel / dW tile, one wave per
 * soft-IoU image): same kernels, same results up to summation order, identical bits on every run -- at the price of launches
 * that no longer fill the chip.  Meant for debugging and for asserting graph replay == eager execution exactly.
 * rsis_set_deterministic returns the previous value. */
int rsis_set_deterministic(int on);
int rsis_get_deterministic(void);

/* ---- weight repacking (private cache of nn.Conv2d.weight; clstm.py:17, model.py:43-47,109, torchvision trunk) ---- */
/* 1 when a conv of this geometry runs on the bf16 kernels under RSIS_DTYPE_BF16 (3x3 / stride 1 / pad 1, and 1x1 / pad 0 -- the
 * strided 1x1 convs through their stride-1 form on a sub-sampled input -- with more than one output channel), 0 when it keeps
 * its f32 kernel and f32 packed layout */
int rsis_conv_uses_bf16(int ks, int stride, int pad, int Cout);
/* number of BYTES of the packed forward copy for a conv whose input is the channel concat of nseg tensors.
 * (dtype, ks, stride, pad) select the layout: bf16 cells, the direct-kernel layout (3x3/s1/p1, 3x3/s2/p1) or the implicit-GEMM one */
long rsis_conv_packed_bytes_fwd(int dtype, int Cout, int ks, int stride, int pad, int nseg, const int* Cseg);
/* number of BYTES of the packed dgrad copy producing c_count input channels */
long rsis_conv_packed_bytes_dgrad(int dtype, int Cout, int ks, int stride, int pad, int c_count);
/* The packed copy covers nseg (<= 3) input-channel segments of W: Cseg[s] channels starting at Coff[s] (Coff == NULL:
 * consecutive from 0) -- the channel concat the conv will gather from, or any subset of the input channels.
 * lstm_hid > 0: rows of W are [i|f|o|g] x hid (clstm.py:47) and are interleaved to 4*j+gate */
int rsis_conv_pack_fwd(const float* W, void* Wp, int Cout, int Ctot, int ks, int stride, int pad, int nseg, const int* Cseg,
                       const int* Coff, int lstm_hid, int dtype, void* stream);
/* dgrad copy: produces the gradient of the sum(Cseg) input channels of the segments (in segment order) */
int rsis_conv_pack_dgrad(const float* W, void* Wd, int Cout, int Ctot, int ks, int stride, int pad, int nseg, const int* Cseg,
                         const int* Coff, int lstm_hid, int dtype, void* stream);

/* ---- nn.Conv2d forward (model.py:59-63 skip convs, :167 conv_out, vision.py:12-19 trunk convs) ----
 * out[B][Cout][Ho][Wo] = conv(cat(src[0..nsrc-1], dim=1), W, stride, pad) + bias (+ addend, same shape as out).
 * torch.cat (model.py:153) is folded in: up to 3 sources.  tile = 0 lets the library choose the MFMA tiling; tile + 100
 * additionally allows a split-K schedule (fp32 atomics: faster on deep-K / few-pixel layers such as sk5, but the summation
 * order is not reproducible -- the Python binding allows it only while training). */
int rsis_conv2d_fwd(const float* const* src, const int* Csrc, int nsrc, int B, int H, int W, const void* Wp, int Cout,
                    int ks, int stride, int pad, const float* bias, const float* addend, float* out, int Ho, int Wo,
                    int tile, int dtype, void* stream);
/* Inference: the conv above followed by the eval-mode nn.BatchNorm2d of its output (model.py:50-54,59-63; the bn1 / bn2 / bn3 /
 * downsample[1] of every torchvision Bottleneck behind vision.py:12-19), the residual add and the ReLU, in the conv's epilogue:
 *   out = relu?( (conv + bias - running_mean) / sqrt(running_var + eps) * gamma + beta  (+ addend) )
 * -- bit for bit what rsis_conv2d_fwd followed by rsis_bn_fwd(train = 0) writes (one shared definition of the arithmetic), without the
 * BatchNorm launch and its read + write of the activation.  fp32 kernels (dtype RSIS_DTYPE_F32, or a RSIS_DTYPE_BF16 pack of a conv the
 * library runs on the fp32 kernels anyway: the 7x7 stem), one destination, Cout % 4 == 0, no split-K (conv_out is not
 * covered): otherwise RSIS_ERR_UNSUPPORTED is returned BEFORE anything is launched and the caller runs the two launches. */
int rsis_conv2d_fwd_bn_eval(const float* const* src, const int* Csrc, int nsrc, int B, int H, int W, const void* Wp, int Cout,
                            int ks, int stride, int pad, const float* bias, const float* addend, const float* gamma, const float* beta,
                            const float* running_mean, const float* running_var, float eps, int relu, float* out, int Ho, int Wo,
                            int tile, int dtype, void* stream);

/* ---- nn.Conv2d backward-data (autograd of the above): dx for input channels [c_lo,c_hi) of the conv input, written
 * to ndst tensors dx[i] = [B][Cdx[i]][Hx][Wx] (the inverse of the channel concat). dy = [B][Cout][Hy][Wy].
 * Cin_packed = sum(Cseg) given to rsis_conv_pack_dgrad; sum(Cdx) <= Cin_packed (leading channels are produced).
 * addend (optional, ndst == 1): dx[0] = dgrad + addend -- the gradient the same tensor receives through another consumer (the
 * identity branch of a residual block), summed in the epilogue instead of by a separate pass.  stride 1: any pointer; 1x1 with
 * stride > 1: addend must BE dx[0] -- the strided gradient is accumulated into the existing tensor in place (no zero fill). ---- */
int rsis_conv2d_dgrad(const float* dy, int B, int Cout, int Hy, int Wy, const void* Wd, int Cin_packed, int ks, int stride,
                      int pad, float* const* dx, const int* Cdx, int ndst, int Hx, int Wx, const float* addend, int tile,
                      int dtype, void* stream);

/* ---- nn.Conv2d backward-weight: dW[Cout][Ctot][ks][ks] (reference layout) += corr(x, dy) for the source tensor
 * x = [B][Cs][H][W] that occupies input channels [c_off, c_off+Cs).  ACCUMULATES (fp32 atomics): zero dW first.
 * lstm_hid > 0: dy rows are gate-interleaved (4*j+gate) and are mapped back to reference rows. ---- */
int rsis_conv2d_wgrad(const float* dy, const float* x, float* dW, int B, int Cs, int H, int W, int Cout, int Ho, int Wo,
                      int ks, int stride, int pad, int Ctot, int c_off, int lstm_hid, int dtype, void* stream);

/* ---- many weight gradients in one call (same semantics as njobs calls of rsis_conv2d_wgrad, every dW ACCUMULATED; the dW regions
 * of different jobs may coincide -- e.g. the sources of one conv).  A weight gradient is off the critical path of the backward pass
 * (autograd of train.py:184 only needs it before the optimizer step), so the caller may park them and flush them here: the jobs that
 * run on the exact-f32 LDS-DMA tiled kernel are launched as ONE grid per tile configuration, sized for the whole set -- a layer on
 * its own must split its pixel axis 8-32 ways to fill the chip and pays a dW-sized pass of fp32 atomics per split.  The other jobs
 * are launched one by one.  The job array is host memory and is not referenced after the call returns. ---- */
typedef struct rsis_wgrad_job {
  const float* dy;     /* [B][Cout][Ho][Wo] */
  const float* x;      /* [B][Cs][H][W] */
  float* dW;           /* [Cout][Ctot][ks][ks], this source's channels start at c_off */
  int B, Cs, H, W, Cout, Ho, Wo, ks, stride, pad, Ctot, c_off, lstm_hid, dtype;
} rsis_wgrad_job;
int rsis_conv2d_wgrad_batch(const rsis_wgrad_job* jobs, int njobs, void* stream);

/* ---- conv bias gradient: db[Cout] += sum_{b,h,w} dy  (ACCUMULATES; lstm_hid as above) ---- */
int rsis_bias_grad(const float* dy, float* db, int B, int C, int HW, int lstm_hid, void* stream);

/* ---- ConvLSTMCell.forward (clstm.py:19-62): cat(x.., h_prev) -> Gates conv -> chunk -> sigmoid/tanh -> c,h in ONE
 * kernel.  src = the x tensor(s) followed by h_prev (omit h_prev and pass c_prev = NULL for the zero state of
 * clstm.py:26-37).  bias_packed / act_out / addend use interleaved rows.  act_out (post-nonlinearity i,f,o,g, needed
 * by the backward) may be NULL for inference. addend: optional precomputed time-invariant gate contribution (the
 * skip-feature part of the gate conv + bias, computed once per iteration by rsis_conv2d_fwd on interleaved rows); with an
 * addend nsrc may be 0 (level 0 at t = 0).  RSIS_DTYPE_BF16 covers ks == 3 only (RSIS_ERR_UNSUPPORTED otherwise, from
 * this call and from rsis_conv_pack_fwd with lstm_hid > 0): cells with another kernel size run under RSIS_DTYPE_F32. ---- */
int rsis_convlstm_fwd(const float* const* src, const int* Csrc, int nsrc, int B, int H, int W, const void* Wp,
                      const float* bias_packed, const float* addend, const float* c_prev, float* h_out, float* c_out,
                      float* act_out, int hid, int ks, int pad, int tile, int dtype, void* stream);

/* Several INDEPENDENT rsis_convlstm_fwd calls as ONE launch: the cells (level i, timestep d - i) of one diagonal of the decoder's
 * (level, timestep) wavefront -- level i at step t needs level i-1 at step t and itself at step t-1 (model.py:129-165 inside the
 * loop of train.py:85-94 / test.py:37-38), so the cells of a diagonal do not depend on each other.  Exact-f32 3x3 jobs share one
 * grid (conv3x3_direct_group_kernel: blocks of different jobs have different lengths and phases, so one job's prologue / epilogue
 * overlaps another's MFMA loop); other jobs, and every job in the deterministic mode, are launched one after the other -- the
 * results are those of njobs single calls either way (same kernels, same tiles, same summation order).  Fields as the arguments
 * of rsis_convlstm_fwd (src / Csrc hold nsrc <= 3 entries). */
typedef struct rsis_lstm_job {
  const float* src[3];
  int Csrc[3];
  int nsrc, B, H, W;
  const void* Wp;
  const float* bias_packed;
  const float* addend;
  const float* c_prev;
  float* h_out;
  float* c_out;
  float* act_out;
  int hid, ks, pad, tile, dtype;
  /* optional [B][hid] 64-bit keys, ZEROED by the caller: the global max-pool of h (model.py:143, the side feature) is then taken in
   * the kernel's epilogue -- every half wave adds its best (value, pixel) with an atomic max; key = order-preserving bits of the
   * value << 32 | (0x7FFFFFFF - flat pixel index); order-independent, so bit-reproducible.  rsis_heads_fwd_keys decodes them.
   * 3x3 gates only (RSIS_ERR_UNSUPPORTED otherwise). */
  unsigned long long* side_key;
} rsis_lstm_job;
int rsis_convlstm_fwd_batch(const rsis_lstm_job* jobs, int njobs, void* stream);

/* ---- ConvLSTMCell pointwise backward: (dh + dh2, dc_next, saved act, c_prev, c) -> da (gate pre-activation grads,
 * interleaved rows) and dc_prev.  dh / dh2 / dc_next / c_prev / dc_prev / da_sum may be NULL. da_sum += da if given.
 * dh2: the gradient reaching h through a second consumer (the next timestep's recurrence), summed in the kernel. ---- */
int rsis_convlstm_bwd_gates(const float* dh, const float* dh2, const float* dc_next, const float* act, const float* c_prev,
                            const float* c, float* da, float* dc_prev, float* da_sum, int B, int hid, int HW, void* stream);

/* Several INDEPENDENT rsis_convlstm_bwd_gates calls (da_sum == NULL) in one launch, and several independent rsis_conv2d_dgrad calls in
 * one launch where they run on the exact-f32 direct 3x3 kernel (stride 1, pad 1, no addend; other jobs, and every job in the
 * deterministic mode, are issued one by one): the cells (level i, timestep d - i) of a diagonal of the decoder's (level, timestep)
 * wavefront in the BACKWARD pass -- cell (i, t) needs d(up) from cell (i + 1, t) and (dh, dc) from cell (i, t + 1), both one diagonal
 * later, so the cells of a diagonal are independent (reference: autograd of model.py:129-165 inside train.py:85-94).  Fields as the
 * arguments of the single calls. */
typedef struct rsis_lstm_bwd_job {
  const float* dh;
  const float* dh2;
  const float* dc_next;
  const float* act;
  const float* c_prev;
  const float* c;
  float* da;
  float* dc_prev;
  int B, hid, HW;
} rsis_lstm_bwd_job;
int rsis_convlstm_bwd_gates_batch(const rsis_lstm_bwd_job* jobs, int njobs, void* stream);
typedef struct rsis_dgrad_job {
  const float* dy;
  int B, Cout, Hy, Wy;
  const void* Wd;
  int Cin_packed, ks, stride, pad;
  float* dx[3];
  int Cdx[3];
  int ndst, Hx, Wx;
  const float* addend;
  int tile, dtype;
} rsis_dgrad_job;
int rsis_conv2d_dgrad_batch(const rsis_dgrad_job* jobs, int njobs, void* stream);

/* ---- nn.UpsamplingBilinear2d(size) = bilinear, align_corners=True (model.py:149,163; train.py:96; test.py:39) ---- */
int rsis_upsample_bilinear_ac_fwd(const float* x, float* y, long BC, int Hi, int Wi, int Ho, int Wo, void* stream);
int rsis_upsample_bilinear_ac_bwd(const float* dy, float* dx, long BC, int Hi, int Wi, int Ho, int Wo, void* stream);
/* the two consumers of a level's hidden state besides the recurrence (model.py:143,149-150) back-propagated in one launch:
 * dx = upsample_backward(dy) + dpool[bc] at pixel argmax[bc] of every plane (argmax from rsis_global_maxpool_fwd) */
int rsis_upsample_maxpool_bwd(const float* dy, const float* dpool, const int* argmax, float* dx, long BC, int Hi, int Wi, int Ho,
                              int Wo, void* stream);

/* ---- nn.MaxPool2d(full map) side features (model.py:143): y[BC], argmax[BC] (int32 flat index) ---- */
int rsis_global_maxpool_fwd(const float* x, float* y, int* argmax, long BC, int HW, void* stream);
int rsis_global_maxpool_bwd(const float* dy, const int* argmax, float* dx, long BC, int HW, void* stream);
/* dx[bc][argmax[bc]] += dy[bc] (dx already holds the gradient of another consumer of the same tensor) */
int rsis_global_maxpool_bwd_add(const float* dy, const int* argmax, float* dx, long BC, int HW, void* stream);

/* ---- nn.BatchNorm2d (+ residual add + ReLU of the bottleneck) (model.py:50-54,59-63; torchvision trunk) ----
 * train bit0: batch statistics, running-stat update (momentum, unbiased var), saves mean / rstd for the backward;
 * train bit1: `stats` (scratch of 2*C doubles) was zeroed by the caller (one memset for a whole arena instead of one per
 * layer). res may be NULL. */
int rsis_bn_fwd(const float* x, const float* res, float* y, double* stats, const float* gamma, const float* beta,
                float* running_mean, float* running_var, float* save_mean, float* save_rstd, int B, int C, int HW,
                float eps, float momentum, int relu, int train, void* stream);
/* train-mode backward. relu bit0: the forward applied ReLU (y, the forward output, is read for the mask); bit1: `stats`
 * was zeroed by the caller; bit2: ACCUMULATE into dgamma / dbeta instead of overwriting. dres (grad of the residual
 * input = masked dy) may be NULL. */
int rsis_bn_bwd(const float* dy, const float* x, const float* y, const float* save_mean, const float* save_rstd,
                const float* gamma, double* stats, float* dx, float* dres, float* dgamma, float* dbeta, int B, int C,
                int HW, int relu, void* stream);

/* eval-mode backward (the module in .eval(): y = relu?(gamma (x - running_mean) / sqrt(running_var + eps) + beta (+ res)), statistics are
 * constants): g = dy * [y > 0] if relu bit0, dx = gamma / sqrt(running_var + eps) * g, dres = g, dgamma = sum g xhat, dbeta = sum g.
 * Flags and scratch as rsis_bn_bwd (bit1: stats zeroed by the caller; bit2: accumulate into dgamma / dbeta). */
int rsis_bn_bwd_eval(const float* dy, const float* x, const float* y, const float* running_mean, const float* running_var,
                     const float* gamma, double* stats, float* dx, float* dres, float* dgamma, float* dbeta, int B, int C, int HW,
                     float eps, int relu, void* stream);

/* ---- y[bc][ho][wo] = x[bc][ho * stride][wo * stride], y is [BC][(H-1)/stride+1][(W-1)/stride+1]: the dense input of a
 * 1x1 / stride-s conv (torchvision Bottleneck downsample, layers 2-4; reference vision.py:16-19) -- x[:, :, ::s, ::s].contiguous().
 * The conv then runs as its stride-1 form (rsis_conv2d_fwd on y) and y is what its weight gradient reads. ---- */
int rsis_subsample2d(const float* x, float* y, long BC, int H, int W, int stride, void* stream);

/* ---- nn.MaxPool2d(3, stride 2, padding 1) of the ResNet stem (vision.py:15) ---- */
int rsis_maxpool3x3s2_fwd(const float* x, float* y, unsigned char* argmax, long BC, int H, int W, int Ho, int Wo,
                          void* stream);
/* accumulate != 0: dx += (dx already holds the gradient the pooled tensor's input receives from another consumer) */
int rsis_maxpool3x3s2_bwd(const float* dy, const unsigned char* argmax, float* dx, long BC, int H, int W, int Ho, int Wo,
                          int accumulate, void* stream);

/* ---- torch.optim.Adam step on a flat parameter range (utils/utils.py:83-84; train.py:185-187); g is scaled by gscale
 * (1/world_size after the RCCL sum all-reduce) before the L2 weight-decay term is added.  step: the 1-based update count of
 * the range (bias correction); step_dev != NULL: the count is read from that device int32 instead (the caller increments it on
 * the stream), so that a captured hipGraph replays with the live count. ---- */
int rsis_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, float gscale, const int* step_dev, void* stream);

/* ---- Hungarian matching of predictions to ground-truth slots (hungarian.py:91-125, munkres.Munkres().compute per
 * sample): scores[B][G][T] fp32 (rows = GT slots, columns = predictions, T <= G <= 64) -> perm[B][G] int64 with
 * perm[b][t] = GT slot of prediction t for t < T and 0 elsewhere.  Minimum total cost; runs on the device so the training
 * iteration needs no host synchronisation. ---- */
int rsis_assign_min_cost(const float* scores, long long* perm, int B, int G, int T, void* stream);

/* ---- batched repack: every packed weight copy of a model in ONE launch (after an optimizer step).  `jobs` is a DEVICE array
 * of njobs descriptors, sorted by block_begin; job i is processed by the blocks [block_begin_i, block_begin_{i+1}) of a grid
 * of total_blocks blocks, rsis_conv_pack_job_blocks() blocks each.  rsis_conv_pack_job_fill() fills the derived fields
 * (imode, ldw, krows) of a host-side descriptor exactly as rsis_conv_pack_fwd / _dgrad would choose them and returns the
 * job's block count (< 0: invalid). ---- */
typedef struct rsis_pack_job {
  const float* W;      /* reference-layout weight [Cout][Ctot][ks][ks] (device) */
  void* out;           /* packed copy (device), rsis_conv_packed_bytes_fwd / _dgrad bytes */
  int dgrad;           /* 0: forward copy, 1: data-gradient copy */
  int dtype;           /* RSIS_DTYPE_* of the kernels that will read the copy */
  int Cout, Ctot, ks, stride, pad, nseg, Cseg[3], Coff[3], lstm_hid;
  int imode, ldw, krows;   /* derived (rsis_conv_pack_job_fill) */
  int block_begin;     /* first block of this job in the batch grid */
} rsis_pack_job;
int rsis_conv_pack_job_fill(rsis_pack_job* job);
int rsis_conv_pack_batch(const rsis_pack_job* jobs_dev, int njobs, int total_blocks, void* stream);

/* ---- conv_out (nn.Conv2d(Cin, 1, 3, padding=1), model.py:109,167): weight AND bias gradient in one launch (the bias gradient is
 * the sum of dy, which the weight-gradient kernel reads anyway).  dy[B][1][H][W], x[B][Cin][H][W]; dW[Cin*9] (reference layout
 * [1][Cin][3][3]) and db[1] (may be NULL) are ACCUMULATED.  Cin in {4, 8, 16}, W % 4 == 0, else RSIS_ERR_UNSUPPORTED. ---- */
int rsis_conv_out_wgrad(const float* dy, const float* x, float* dW, float* db, int B, int Cin, int H, int W, void* stream);

/* ---- conv_out over ALL T timesteps of a decoded sequence in one launch each way (model.py:167 inside the loop of train.py:85-94:
 * out_mask_t = conv_out(upsampled hidden state of the last level at step t); the T applications are independent).  The Cin-channel
 * tensors are [T][B][Cin][H][W] (the decoder's stacked per-step buffers), the one-channel tensors are [B][T][H*W] -- the (B, T, N)
 * layout train.py:118 stacks the mask logits into, so no torch.stack copy is needed.  Same kernels and per-image arithmetic as
 * rsis_conv2d_fwd / _dgrad / rsis_conv_out_wgrad on one timestep.  Cin in {4, 8, 16}, W % 4 == 0, else RSIS_ERR_UNSUPPORTED.
 * Wp / Wd: the RSIS_DTYPE_F32 packs of conv_out's weight (rsis_conv_pack_fwd / _dgrad; conv_out keeps its f32 kernels under
 * RSIS_DTYPE_BF16); dW / db are ACCUMULATED. ---- */
int rsis_conv_out_seq_fwd(const float* x, const void* Wp, const float* bias, float* y, int T, int B, int Cin, int H, int W, void* stream);
int rsis_conv_out_seq_dgrad(const float* dy, const void* Wd, float* dx, int T, int B, int Cin, int H, int W, void* stream);
int rsis_conv_out_seq_wgrad(const float* dy, const float* x, float* dW, float* db, int T, int B, int Cin, int H, int W, void* stream);

/* ---- y[n] = sum_t x[t][n], t ascending (fixed order): sum over the timesteps of the stacked gate gradients, i.e. the gradient of a
 * ConvLSTM level's time-invariant gate term (autograd's accumulation over the T uses of the skip features, train.py:85-94) ---- */
int rsis_sum_leading(const float* x, float* y, int T, long n, void* stream);

/* ---- data-layer augmentation: nearest-neighbour affine warp (dataloader/transforms/utils.py:67-147 th_affine2d(mode='nearest',
 * center=True); applied by transforms.py:23-142 RandomAffine to the image, the instance map and the class map of a sample).
 * x, y: [N][C][H][W] float32 (y != x); mat: [N][mat_rows][3] float32 in DEVICE memory, mat_rows = 3 (the reference's 3x3, last
 * row ignored) or 2.  y[n][c][i][j] = x[n][c][round(clamp(A (i-ci, j-cj) + b + (ci, cj)))], ci = H/2 - 0.5, cj = W/2 - 0.5,
 * float32 arithmetic in the reference's operation order, round half to even: bit-exact with the reference. ---- */
int rsis_affine_nearest(const float* x, float* y, const float* mat, int mat_rows, int N, int C, int H, int W, void* stream);

/* ---- soft-IoU matching scores and matched-loss gradient (train.py:98-110,127-131,162-163; hungarian.py:62-89 softIoU) ----
 * rsis_softiou_sums: logits[B][T][N] (mask logits of the T predictions), y[B][G][N] (ground-truth masks, 0/1 floats) ->
 *   S[B][T+1][G+1]:  S[t][g] = sum_n sigmoid(logits[t][n]) * y[g][n],  S[t][G] = sum_n sigmoid(logits[t][n]),
 *   S[T][g] = sum_n y[g][n]  (S[T][G] is left 0).  One pass over both tensors; requires T < 32, G < 32, N % 8 == 0.
 *   The reference's cost is then 1 - S[t][g] / (S[t][G] + S[T][g] - S[t][g] + 1e-6) for every pair.
 * rsis_softiou_bwd: gradient of the matched costs w.r.t. the logits,
 *   dlogits[b][t][n] = (ca[b][t]*y + cb[b][t]*(1-y)) * p*(1-p),  y = y[b][perm[b*perm_ld + t]][n], p = sigmoid(logits[b][t][n]);
 *   the caller supplies ca = -g/U, cb = g*I/U^2 (g = upstream gradient of the cost, U = the cost's denominator). N % 4 == 0. */
int rsis_softiou_sums(const float* logits, const float* y, float* S, int B, int T, int G, long N, void* stream);
int rsis_softiou_bwd(const float* logits, const float* y, const long long* perm, int perm_ld, const float* ca, const float* cb,
                     float* dlogits, int B, int T, int G, long N, void* stream);

/* ---- class / stop heads of one decoder timestep (model.py:169-182): side = channel concat (by pointer) of nside <= 5 vectors
 * side[i][B][Cside[i]] (the global max-pools of the hidden states); class_probs[B][ncls] = softmax(Wc side + bc) with
 * Wc[ncls][K], K = sum Cside; stop[B] = Ws . side + bs (a logit).  rsis_heads_bwd: dside[i][B][Cside[i]] (any pointer may be
 * null) from dprobs[B][ncls] / dstop[B] (either may be null = zero), and dWc, dbc, dWs, dbs are ACCUMULATED into (null = skip).
 * K <= 2048, ncls <= 64; rsis_heads_bwd: 2 B (ncls + 1) + B ceil(K / B) floats of LDS <= 64 KB -- B = 320 rows at 21 classes (the parameter gradients are reduced over the rows in-kernel, no atomics; decoder_seq passes the T * B rows of all timesteps in one call). ---- */
int rsis_heads_fwd(const float* const* side, const int* Cside, int nside, int B, const float* Wc, const float* bc, int ncls,
                   const float* Ws, const float* bs, float* class_probs, float* stop, void* stream);
/* rsis_heads_fwd with the pooled side features given as the keys of rsis_lstm_job.side_key (one [B][Cside[i]] key array per level):
 * decodes them, ALSO writes the float features to side_out[i] and the arg-max pixels to arg_out[i] (what rsis_global_maxpool_fwd
 * would have produced: the backward passes read those), then computes the heads as rsis_heads_fwd does. */
int rsis_heads_fwd_keys(const unsigned long long* const* keys, float* const* side_out, int* const* arg_out, const int* Cside, int nside,
                        int B, const float* Wc, const float* bc, int ncls, const float* Ws, const float* bs, float* class_probs, float* stop,
                        void* stream);
int rsis_heads_bwd(const float* const* side, const int* Cside, int nside, int B, const float* Wc, int ncls, const float* Ws,
                   const float* class_probs, const float* dprobs, const float* dstop, float* const* dside, float* dWc, float* dbc,
                   float* dWs, float* dbs, void* stream);

/* ---- loss tail of a training iteration (train.py:159-176; hungarian.py:10-59): masked means of the class NLL (probs[n][C],
 * targets y_class[n] int64, optional per-class weights cls_w[C]), of the matched soft-IoU costs siou[n] and of the balanced stop
 * BCE (logits stop[n], target = sw_mask, balance weight bw or < 0 = positives / n), n = B*T samples; sw_mask / sw_class are the
 * 0/1 sample weights.  out[4] = {w_iou*iou + w_cls*cls + w_stop*stop, iou, stop, cls} (pass w_cls / w_stop = 0 for a disabled
 * loss).  With dprobs / dstop / dsiou non-null the same launch also writes the gradient of out[0] * gout[0] (gout: device
 * pointer to the upstream gradient, null = 1) w.r.t. the three inputs (out may then be null). ---- */
int rsis_loss_tail(const float* probs, const long long* y_class, const float* stop, const float* siou, const float* sw_mask,
                   const float* sw_class, const float* cls_w, int n, int C, float bw, float w_iou, float w_cls, float w_stop,
                   float* out, float* dprobs, float* dstop, float* dsiou, const float* gout, void* stream);

/* ---- inference post-processing of predicted masks (eval.py:96-127 resize_mask + pycocotools mask.encode; RLE semantics of
 * src/coco/common/maskApi.c:32-41,196-209) ----
 * rsis_mask_resize_threshold: prob[n][Hm][Wm] fp32 -> seg[n][w][h] uint8 (COLUMN-major h x w masks: align-corners bilinear
 *   resample == scipy.ndimage.zoom(order=1), `> th`, pixels with ignore[h][w] == 1 cleared (ignore may be null)), the same
 *   without the ignore mask in raw (may be null), area[n] = number of set pixels of seg.
 * rsis_rle_encode: masks[n][len] uint8 column-major -> counts[n][cap] uint32 = lengths of the alternating runs starting with
 *   the (possibly empty) run of zeros; nruns[n] = number of runs, or -(number of runs) when cap is too small (counts then
 *   undefined for that mask).
 * rsis_rle_to_string: HOST function (host pointers): the COCO text form of `m` counts into out[cap]; returns its length
 *   (NUL-terminated) or -1 when cap is too small. */
int rsis_mask_resize_threshold(const float* prob, int n, int Hm, int Wm, const unsigned char* ignore, float th, unsigned char* seg,
                               unsigned char* raw, unsigned int* area, int h, int w, void* stream);
int rsis_rle_encode(const unsigned char* masks, int n, long len, unsigned int* counts, int cap, int* nruns, void* stream);
int rsis_rle_to_string(const unsigned int* counts, int m, char* out, int cap);
/* largest 8-connected component of n binary masks [n][h][w] (row-major, 0/1) -> out [n][h][w] (eval_cityscapes.py:131-150:
 * skimage.measure.label + most frequent label; ties: the component first in raster order).  Caller-owned workspaces:
 * labels[n*h*w], counts[n*h*w] int32, best[n] int32.  h*w < 2^31. */
int rsis_largest_component(const unsigned char* mask, unsigned char* out, int* labels, int* counts, int* best, int n, int h, int w,
                           void* stream);

/* ---- channel-blocked bf16 activations: the storage half of the bf16 path (BASELINE.json configs[2..4]) --------------------------
 * A logical [B][C][H][W] tensor stored as bf16 [B][C/8][H][W][8] ("blk": the 8 channels of a pixel are one 16-byte cell, C % 8 == 0).
 * The reference has no counterpart (fp32 NCHW throughout); these entry points serve the ResNet-101 trunk of
 * src/modules/vision.py:12-19 under -dtype bf16. */

/* fp32 NCHW -> blk (round-to-nearest-even) and back (exact) */
int rsis_blk_from_nchw(const float* x, void* y_blk, int B, int C, int H, int W, void* stream);
int rsis_blk_to_nchw(const void* x_blk, float* y, int B, int C, int H, int W, void* stream);

/* out_blk[B][Cout] = conv(x_blk[B][C], W), ks in {1, 3}, stride 1, "same" padding, no bias (torchvision's trunk convs have none);
 * Wp: the bf16 pack rsis_conv_pack_fwd (forward) or rsis_conv_pack_dgrad (data gradient: x = dy, out = dx) produce for dtype
 * RSIS_DTYPE_BF16.  addend_blk (optional, the output's shape; may alias out_blk) is added in fp32 before the ONE rounding to bf16 at
 * the store.  variant: 0 = pick a tile, > 0 force (tests).  C and Cout need only be multiples of 8: the ring stages hold 16 (3x3) or 32
 * (1x1) input channels, and a ragged last stage reads zero cells (descriptor range) against the pack's zero-padded rows. */
int rsis_blk_conv2d(const void* x_blk, int B, int C, int H, int W, const void* Wp, int Cout, int ks, const void* addend_blk,
                    void* out_blk, int variant, void* stream);
/* ... followed by the eval-mode BatchNorm of its output IN THE EPILOGUE (inference: test(), reference src/test.py:35-38; every trunk
 * conv is followed by one): out = relu?(v * sc + sh (+ addend)), sc = gamma * rsqrt(running_var + eps), sh = beta - running_mean * sc, where
 * v is the product rounded to bf16 (single_rounding = 0: bit for bit what rsis_blk_conv2d + rsis_blk_bn_fwd (eval) compute, without the
 * BatchNorm launch) or kept in fp32 (single_rounding = 1: closer to the fp32 result).  gamma .. running_var: fp32 [Cout]. */
int rsis_blk_conv2d_bn_eval(const void* x_blk, int B, int C, int H, int W, const void* Wp, int Cout, int ks, const void* addend_blk,
                            const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, int relu,
                            int single_rounding, void* out_blk, int variant, void* stream);

/* BatchNorm2d (+ residual add) (+ ReLU) on blk tensors: y = relu?((x - mean) * rstd * gamma + beta (+ res)).  train != 0: batch
 * statistics (saved to save_mean / save_rstd for the backward; run_mean / run_var, if given, updated with `momentum` and the unbiased
 * variance, as nn.BatchNorm2d does); train == 0: normalises with run_mean / run_var.  scratch: rsis_blk_bn_scratch_doubles(C) doubles
 * (per-split partial sums: the reduction order is fixed, results are bit-reproducible). */
long rsis_blk_bn_scratch_doubles(int C);
int rsis_blk_bn_fwd(const void* x_blk, const void* res_blk, void* y_blk, double* scratch, const float* gamma, const float* beta,
                    float* run_mean, float* run_var, float* save_mean, float* save_rstd, int B, int C, int H, int W, float eps,
                    float momentum, int relu, int train, void* stream);
/* backward of the train-mode forward: g = dy * [y > 0] if relu (y_blk = the forward output; NULL: the mask is recomputed from x -- only
 * valid without a residual), dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)), dres_blk (optional) = g, the gradient of the
 * residual branch; dgamma / dbeta (optional) = sum g * xhat / sum g, added to the buffers if accumulate != 0, stored otherwise. */
int rsis_blk_bn_bwd(const void* dy_blk, const void* x_blk, const void* y_blk, double* scratch, const float* gamma, const float* beta,
                    const float* save_mean, const float* save_rstd, void* dx_blk, void* dres_blk, float* dgamma, float* dbeta,
                    int accumulate, int B, int C, int H, int W, int relu, void* stream);
/* y[oh][ow] = x[oh * stride][ow * stride] (a strided conv = the stride-1 conv, then this) and its transpose (dx of H x W: dy at the
 * multiples of stride, zero elsewhere) */
int rsis_blk_subsample2d(const void* x_blk, void* y_blk, int B, int C, int H, int W, int stride, void* stream);
int rsis_blk_upscatter2d(const void* dy_blk, void* dx_blk, int B, int C, int H, int W, int stride, void* stream);

/* ---- the recurrent decoder on blk tensors (ConvLSTM cells of clstm.py:19-62 inside the loop of model.py:129-165, run over the
 * (level, timestep) wavefront: the cells of a diagonal are independent, so every entry point below takes SEVERAL independent jobs
 * and runs them in one grid where it can).  Reference arithmetic with bf16 operands / storage: fp32 accumulation and cell update,
 * ONE rounding at each blk store; the cell state c stays fp32 NCHW. ---- */

/* One 3x3 / stride 1 / pad 1 conv over the channel concat of nsrc <= 3 blk sources.  Wp: the RSIS_DTYPE_BF16 pack of
 * rsis_conv_pack_fwd (or _dgrad: the job is then the data gradient, sources = dy) built for the same segments; Cpack = the pack's
 * row count (Cout of rsis_conv_pack_fwd, Cin_packed of rsis_conv_pack_dgrad; 0 = Cout).
 *   hid == 0, plain epilogue: out = conv (+ bias[Cout], fp32, packed row order) (+ addend, blk [B][Cout]) split over ndst <= 2 blk
 *     destinations of Cdst[i] channels (Cdst % 8 == 0, sum = Cout <= Cpack: leading rows of the pack);
 *   hid > 0, fused ConvLSTM cell (rows gate-interleaved 4 j + gate, Cout = 4 hid, hid % 8 == 0): gates = conv (+ bias) + addend (blk
 *     [B][4 hid], the time-invariant skip term) -> i, f, o = sigmoid, g = tanh, c = f c_prev + i g, h = o tanh(c).  c_prev (NULL = zero
 *     state) / c_out: fp32 [B][hid][H][W]; h_out: blk [B][hid]; act_out (may be NULL): blk [B][4 hid], the post-nonlinearity gates;
 *     side_key (may be NULL): [B][hid] keys of the global max-pool of h (fp32, before the rounding of h_out), as rsis_lstm_job.side_key.  nsrc may be 0. */
typedef struct rsis_blk_conv_job {
  const void* src[3];
  int Csrc[3];
  int nsrc, B, H, W;
  const void* Wp;
  int Cout, Cpack;
  const float* bias;
  const void* addend;
  void* dst[2];
  int Cdst[2];
  int ndst;
  int hid;
  const float* c_prev;
  float* c_out;
  void* h_out;
  void* act_out;
  unsigned long long* side_key;
  int tile;              /* 0 = the library picks the MFMA tile; 1 / 4 / 5 force one (tests) */
} rsis_blk_conv_job;
int rsis_blk_conv3x3_batch(const rsis_blk_conv_job* jobs, int njobs, void* stream);

/* align-corners bilinear resize (nn.UpsamplingBilinear2d, model.py:149-150,163-164) of blk tensors, x [B][C][Hi][Wi] -> y [B][C][Ho][Wo],
 * and its transpose dx = resize^T(dy); with dpool / arg (both [B][C], or both NULL) the transpose also adds dpool[b][c] at flat pixel
 * arg[b][c] of every channel plane: the gradient of the global max-pool side feature (model.py:143) of the same tensor. */
typedef struct rsis_blk_resize_job {
  const void* src;       /* forward: x; backward: dy */
  void* dst;             /* forward: y; backward: dx */
  const float* dpool;
  const int* arg;
  int B, C, Hi, Wi, Ho, Wo;
} rsis_blk_resize_job;
int rsis_blk_upsample_fwd_batch(const rsis_blk_resize_job* jobs, int njobs, void* stream);
int rsis_blk_upsample_bwd_batch(const rsis_blk_resize_job* jobs, int njobs, void* stream);

/* rsis_convlstm_bwd_gates on blk tensors: dh, dh2 (may be NULL) blk [B][hid]; act, da blk [B][4 hid] (rows 4 j + gate); c, c_prev,
 * dc_next, dc_prev fp32 [B][hid][HW] (c_prev / dc_next / dc_prev may be NULL). */
typedef struct rsis_blk_lstm_bwd_job {
  const void* dh;
  const void* dh2;
  const float* dc_next;
  const void* act;
  const float* c_prev;
  const float* c;
  void* da;
  float* dc_prev;
  int B, hid, HW;
} rsis_blk_lstm_bwd_job;
int rsis_blk_lstm_bwd_batch(const rsis_blk_lstm_bwd_job* jobs, int njobs, void* stream);

/* y_blk[cell] = sum_t x_blk[t][cell] over T stacked blk tensors of ncells 16-byte cells each (fp32 accumulation, t ascending, one
 * rounding): the sum over the timesteps of d(gates), i.e. the gradient of a level's time-invariant gate term */
int rsis_blk_sum_leading(const void* x_blk, void* y_blk, int T, long ncells, void* stream);
/* rsis_bias_grad for a blk dy [B][C][HW]: db[C] += sum over (b, pixel); lstm_hid > 0: blk channel 4 j + gate -> db[gate * hid + j] */
int rsis_blk_bias_grad(const void* dy_blk, float* db, int B, int C, int HW, int lstm_hid, void* stream);

/* conv_out (model.py:109,167; Cin == 8: hidden_size / 16 at hidden_size 128) over all T timesteps from the blk hidden state:
 * x / dx blk [T][B][8][H][W]; y / dy fp32 [B][T][H*W]; Wref = the reference-layout weight [1][8][3][3] (fp32, no pack); bias [1] or
 * NULL; dW[72] / db[1] (db may be NULL) are ACCUMULATED.  W % 4 == 0. */
int rsis_blk_conv_out_seq_fwd(const void* x, const float* Wref, const float* bias, float* y, int T, int B, int H, int W, void* stream);
int rsis_blk_conv_out_seq_dgrad(const float* dy, const float* Wref, void* dx, int T, int B, int H, int W, void* stream);
int rsis_blk_conv_out_seq_wgrad(const float* dy, const void* x, float* dW, float* db, int T, int B, int H, int W, void* stream);

/* The decoder's tail as one op per direction (upconv_out.hip): nn.UpsamplingBilinear2d (align_corners, model.py:163-164) of the last
 * level's hidden state followed by conv_out (model.py:109,167: Conv2d(8 -> 1, 3x3, pad 1)) over the images of all T timesteps, without
 * forming the upsampled tensor.  h / dh: fp32 [T][B][8][Hs][Ws] (h_blk = 0) or bf16 blk [T][B][1][Hs][Ws][8] (h_blk = 1); out / dout:
 * fp32 [B][T][Ho*Wo]; w = the reference-layout weight [1][8][3][3]; bias [1] or NULL.  Backward: dh is WRITTEN (plus, when dside /
 * arg are given, the gradient dside[T*B][8] of the side max-pool feature at its arg-max pixel arg[T*B][8], model.py:143); dW[72] and
 * db[1] (either may be NULL) are ACCUMULATED, from per-block sums in `partial` (rsis_upconv_out_bwd_blocks(..) x 80 floats of
 * scratch) added in a fixed order: reproducible run to run.  rsis_upconv_out_supported: C == 8 and an upsampling factor of ~2
 * (scale (in-1)/(out-1) in [0.46, 0.52]); otherwise RSIS_ERR_UNSUPPORTED (run rsis_upsample_* + rsis_conv_out_seq_* instead). */
int rsis_upconv_out_supported(int C, int Hs, int Ws, int Ho, int Wo);
int rsis_upconv_out_bwd_blocks(int T, int B, int Hs, int Ws);
int rsis_upconv_out_fwd(const void* h, int h_blk, const float* w, const float* bias, float* out, int T, int B, int C, int Hs, int Ws, int Ho, int Wo,
                        void* stream);
int rsis_upconv_out_bwd(const float* dout, const void* h, int h_blk, const float* w, void* dh, float* dW, float* db, const float* dside, const int* arg,
                        float* partial, int T, int B, int C, int Hs, int Ws, int Ho, int Wo, void* stream);

/* ---- gradient exchange: RCCL bound directly (replaces nn.DataParallel, reference src/train.py:269-274: one process per GPU, the flat
 * gradient buffers SUM-all-reduced over xGMI once per iteration).  A collective issued here is an ordinary operation of `stream`: it
 * can be captured into the hipGraph of the training iteration (torch.distributed's ProcessGroupNCCL cannot: its watchdog thread's
 * event queries abort a concurrent stream capture).  RCCL is resolved at run time (dlopen); RSIS_ERR_UNSUPPORTED if it is not there.
 *   rsis_comm_available : RSIS_OK iff librccl resolves in this process (dlopen + symbols only: no bootstrap listener, no device work);
 *                         what every rank calls to AGREE on the direct exchange before anything that can block inside RCCL;
 *   rsis_comm_unique_id : rank 0 fills id_out[128] (ncclUniqueId); the caller distributes it to the other ranks (any channel);
 *   rsis_comm_init      : collective over the `world` ranks, each on its own current HIP device; *comm receives the communicator;
 *   rsis_comm_size      : number of ranks of the communicator (-1 on error);
 *   rsis_comm_allreduce_sum_f32 : buf[n] <- sum over ranks, in place, enqueued on `stream`;
 *   rsis_comm_destroy, rsis_comm_last_error (text of the last RCCL failure of this process). ---- */
int rsis_comm_available(void);
int rsis_comm_unique_id(void* id_out);
int rsis_comm_init(void** comm, int world, int rank, const void* id);
int rsis_comm_size(void* comm);
int rsis_comm_allreduce_sum_f32(void* comm, float* buf, long n, void* stream);
int rsis_comm_destroy(void* comm);
const char* rsis_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RSIS_HIP_H */
