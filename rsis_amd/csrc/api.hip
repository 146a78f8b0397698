// extern "C" entry points of librsis_hip.so (declared in include/rsis_hip.h).  Argument validation + dispatch only.
#include "common.h"
#include "../../include/rsis_hip.h"
#include <stdlib.h>

// launchers implemented in the other translation units
int rsis_launch_conv_igemm(ConvArgs& a, int ks, bool dgrad, int epi, int force_tile, hipStream_t st);
int rsis_launch_conv_wgrad(WgradArgs& a, int ks, hipStream_t st);
int rsis_launch_conv3x3_direct(ConvArgs& a, int epi, int force_variant, hipStream_t st);
int rsis_launch_conv_wino(const float* x, const void* U, const float* bias, const float* addend, float* y, int B, int C, int Cout, int H, int W,
                          hipStream_t st, int precise);
int rsis_launch_conv_wino_group(int n, const float* const* x, const void* const* U, float* const* y, float* const* y1, const int* B, const int* C,
                                const int* Cout, const int* C0, const int* H, const int* W, hipStream_t st);
int rsis_launch_convlstm_direct_group(ConvArgs* jobs, int n, const int* force_variant, hipStream_t st);
int rsis_launch_conv3x3_direct_group_plain(ConvArgs* jobs, int n, const int* force_variant, hipStream_t st);
int rsis_l_lstm_bwd_group(const void* const*, const int*, int, hipStream_t);
int rsis_launch_conv_bf16(ConvArgs& a, int ks, int epi, int force_variant, hipStream_t st);
int rsis_launch_conv_blk(ConvArgs& a, int ks, int variant, hipStream_t st);
int rsis_l_blk_from_nchw(const float*, void*, int, int, int, hipStream_t);
int rsis_l_blk_to_nchw(const void*, float*, int, int, int, hipStream_t);
long rsis_l_blk_bn_scratch(int);
int rsis_l_blk_bn_fwd(const void*, const void*, void*, double*, const float*, const float*, float*, float*, float*, float*, int, int, int, float,
                      float, int, int, hipStream_t);
int rsis_l_blk_bn_bwd(const void*, const void*, const void*, double*, const float*, const float*, const float*, const float*, void*, void*,
                      float*, float*, int, int, int, int, int, hipStream_t);
int rsis_l_blk_subsample(const void*, void*, long, int, int, int, int, int, hipStream_t);
int rsis_l_blk_upscatter(const void*, void*, long, int, int, int, int, int, hipStream_t);
bool rsis_wgrad_bf16_supported(const WgradArgs& w, int ks);
int rsis_launch_conv_wgrad_bf16(const WgradArgs& w, int ks, hipStream_t st);
int rsis_launch_conv_wgrad_bf16_group(const WgradArgs* w, int n, int ks, hipStream_t st);
bool rsis_wgrad_tiled_supported(const WgradArgs& w, int ks);
int rsis_launch_conv_wgrad_tiled(const WgradArgs& w, int ks, hipStream_t st);
int rsis_launch_conv_wgrad_tiled_group(const WgradArgs* w, int n, int ks, hipStream_t st);
bool rsis_c1_supported(int Cin);
int rsis_l_c1_fwd(const float*, const float*, int, const float*, float*, int, int, int, int, int, hipStream_t);
int rsis_l_c1_dgrad(const float*, const float*, int, float*, int, int, int, int, int, hipStream_t);
int rsis_l_c1_wgrad(const float*, const float*, float*, float*, int, int, int, int, int, hipStream_t);
int rsis_l_sum_leading(const float*, float*, int, long, hipStream_t);
int rsis_launch_conv_blk_dec(BlkConvJob*, int, int, const int*, hipStream_t);
int rsis_l_blk_upsample_fwd(const BlkResizeJob*, int, hipStream_t);
int rsis_l_blk_upsample_bwd(const BlkResizeJob*, int, hipStream_t);
int rsis_l_blk_lstm_bwd(const BlkLstmBwdJob*, int, hipStream_t);
int rsis_l_blk_sum_leading(const void*, void*, int, long, hipStream_t);
int rsis_l_blk_channel_sum(const void*, float*, int, int, int, int, hipStream_t);
int rsis_l_blk_c1_fwd(const void*, const float*, const float*, float*, int, int, int, int, hipStream_t);
int rsis_l_blk_c1_dgrad(const float*, const float*, void*, int, int, int, int, hipStream_t);
int rsis_l_blk_c1_wgrad(const float*, const void*, float*, float*, int, int, int, int, hipStream_t);
int rsis_l_pack(int, const float*, void*, int, int, int, int, const int*, const int*, int, int, int, hipStream_t);

int rsis_l_lstm_bwd(const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, float*, int, int, int,
                    hipStream_t);
int rsis_l_upsample_fwd(const float*, float*, long, int, int, int, int, hipStream_t);
int rsis_l_upsample_bwd(const float*, float*, long, int, int, int, int, const float*, const int*, hipStream_t);
int rsis_l_gmax_fwd(const float*, float*, int*, long, int, hipStream_t);
int rsis_l_gmax_bwd(const float*, const int*, float*, long, int, hipStream_t);
int rsis_l_bn_fwd(const float*, const float*, float*, double*, const float*, const float*, float*, float*, float*, float*, int,
                  int, int, float, float, int, int, hipStream_t);
int rsis_l_bn_bwd(const float*, const float*, const float*, const float*, const float*, const float*, double*, float*, float*,
                  float*, float*, int, int, int, int, float, hipStream_t);
int rsis_l_maxpool_fwd(const float*, float*, unsigned char*, long, int, int, int, int, hipStream_t);
int rsis_l_subsample(const float*, float*, long, int, int, int, int, int, hipStream_t);
int rsis_l_maxpool_bwd(const float*, const unsigned char*, float*, long, int, int, int, int, int, hipStream_t);
int rsis_l_channel_sum(const float*, float*, int, int, int, int, hipStream_t);
int rsis_l_adam(float*, const float*, float*, float*, long, float, float, float, float, float, int, float, const int*, hipStream_t);
int rsis_l_assign(const float*, long long*, int, int, int, hipStream_t);
int rsis_l_gmax_bwd_add(const float*, const int*, float*, long, int, hipStream_t);
int rsis_l_pack_batch(const rsis_pack_job*, int, int, hipStream_t);
int rsis_l_pack_blocks(int mode, int krows, int ldw, int ks);
int rsis_l_affine_nearest(const float*, float*, const float*, int, int, int, int, int, hipStream_t);
int rsis_l_mask_resize_threshold(const float*, int, int, int, const unsigned char*, float, unsigned char*, unsigned char*, unsigned int*,
                                 int, int, hipStream_t);
int rsis_l_rle_encode(const unsigned char*, int, long, unsigned int*, int, int*, hipStream_t);
int rsis_l_rle_to_string(const unsigned int*, int, char*, int);
int rsis_l_largest_component(const unsigned char*, unsigned char*, int*, int*, int*, int, int, int, hipStream_t);
int rsis_l_heads_fwd(const float* const*, const int*, int, int, const float*, const float*, int, const float*, const float*, float*, float*,
                     hipStream_t, const unsigned long long* const* keys = nullptr, float* const* side_out = nullptr, int* const* arg_out = nullptr);
int rsis_l_heads_bwd(const float* const*, const int*, int, int, const float*, int, const float*, const float*, const float*, const float*,
                     float* const*, float*, float*, float*, float*, hipStream_t);
int rsis_l_loss_tail(const float*, const long long*, const float*, const float*, const float*, const float*, const float*, int, int, float,
                     float, float, float, float*, float*, float*, float*, const float*, hipStream_t);
int rsis_l_softiou_sums(const float*, const float*, float*, int, int, int, long, hipStream_t);
int rsis_l_softiou_bwd(const float*, const float*, const long long*, int, const float*, const float*, float*, int, int, int, long,
                       hipStream_t);

static inline int krows_of(int C, int ks) { return rsis_roundup(C * ks * ks, RSIS_KPAD); }
static inline int log2i(int s) { int l = 0; while ((1 << l) < s) ++l; return l; }
// 3x3 / stride 1 / pad 1 convs (and their data gradients) run on the direct LDS-patch kernel with its own packed layout
static inline bool use_direct(int ks, int stride, int pad) { return ks == 3 && stride == 1 && pad == 1; }
// 3x3 / stride 2 / pad 1 data gradients run on the same kernel (EPI_S2: parity classes of the input pixel)
static inline bool use_direct_s2(int ks, int stride, int pad) { return ks == 3 && stride == 2 && pad == 1; }
// ... and so does their forward (EPI_F2): the forward copy of the weights uses the direct layout for both strides
// (RSIS_CONV_F2=0: A/B switch back to the implicit-GEMM forward for the strided convs)
static inline bool use_direct_f2(int ks, int stride, int pad) {
  static const bool on = !(getenv("RSIS_CONV_F2") && getenv("RSIS_CONV_F2")[0] == '0');
  return on && use_direct_s2(ks, stride, pad);
}
static inline bool use_direct_fwd(int ks, int stride, int pad) { return use_direct(ks, stride, pad) || use_direct_f2(ks, stride, pad); }
static inline int direct_rows(int nseg, const int* Cseg) {
  int q = 0;
  for (int s = 0; s < nseg; ++s) q += (Cseg[s] + RSIS_CK - 1) / RSIS_CK;
  return q * RSIS_CK * 9;
}
// bf16 kernels (conv_bf16.hip): 3x3 / s1 / p1 and 1x1 / p0 (a strided 1x1 runs as its stride-1 form on a sub-sampled input)
// -- convs with ONE output channel (conv_out) keep their HBM-bound vector kernels (conv_c1.hip)
static inline bool bf16_geom(int ks, int stride, int pad, int Cout) {
  return Cout > 1 && ((ks == 3 && stride == 1 && pad == 1) || (ks == 1 && pad == 0));
}
static inline bool use_bf16(int dtype, int ks, int stride, int pad, int Cout) {
  return dtype == RSIS_DTYPE_BF16 && bf16_geom(ks, stride, pad, Cout);
}
static inline int bf16_ckb(int ks) { return ks == 1 ? RSIS_CKB1 : RSIS_CKB3; }
// -- RSIS_DTYPE_F32_WINO: Winograd F(2x2, 3x3) on the f32 MFMA (conv_wino.hip) for 3x3 / stride 1 / pad 1 convs with ONE source covering
//    all input channels, both channel counts multiples of 32 (forward tiles 32 output channels over 8-channel chunks, the data gradient
//    swaps the roles); anything else under that dtype is plain RSIS_DTYPE_F32
static inline bool wino_geom(int ks, int stride, int pad, int Cin, int Cout, int nseg, int lstm_hid) {
  return ks == 3 && stride == 1 && pad == 1 && nseg == 1 && lstm_hid == 0 && Cin >= 32 && Cout >= 32 && Cin % 32 == 0 && Cout % 32 == 0;
}
// the DATA GRADIENT also covers convs whose input is a concat of segments and whose rows are ConvLSTM gate rows (the gate convs of the
// decoder: d(up) | dh_prev): the kernel's output channels are the concat (a multiple of 32), its reduction the rows (a multiple of 8)
static inline bool wino_geom_dgrad(int ks, int stride, int pad, int Cin, int Cout) {
  return ks == 3 && stride == 1 && pad == 1 && Cin >= 32 && Cout >= 32 && Cin % 32 == 0 && Cout % 32 == 0;
}
static inline bool use_wino(int dtype, int ks, int stride, int pad, int Cin, int Cout, int nseg, int lstm_hid) {
  return dtype == RSIS_DTYPE_F32_WINO && wino_geom(ks, stride, pad, Cin, Cout, nseg, lstm_hid);
}
// cell rows of a bf16 packed copy whose reduction axis is the concat of nseg channel segments
static inline int bf16_rows(int ks, int nseg, const int* Cseg) {
  const int ckb = bf16_ckb(ks);
  int q = 0;
  for (int s = 0; s < nseg; ++s) q += (Cseg[s] + ckb - 1) / ckb;
  return q * ks * ks * (ckb / 8);
}
// ---- bit-reproducible mode (rsis_set_deterministic; initial value from RSIS_DETERMINISTIC=1) ----
static int g_deterministic = -1;
int rsis_deterministic() {
  if (g_deterministic < 0) g_deterministic = (getenv("RSIS_DETERMINISTIC") && getenv("RSIS_DETERMINISTIC")[0] == '1') ? 1 : 0;
  return g_deterministic;
}
// grid split-K of the forward / data-gradient convs (atomics into a zeroed output): RSIS_CONV_SPLITK=0 or the deterministic mode turn it off
static inline bool conv_splitk_ok() {
  static const bool env_ok = !(getenv("RSIS_CONV_SPLITK") && getenv("RSIS_CONV_SPLITK")[0] == '0');
  return env_ok && !rsis_deterministic();
}
static inline int direct_variant(int tile) { const int v = tile % 10; return (tile > 0 && v >= 1 && v <= 9) ? v : 0; }

extern "C" {

int rsis_version(void) { return RSIS_ABI_VERSION; }

int rsis_set_deterministic(int on) {
  const int prev = rsis_deterministic();
  g_deterministic = on ? 1 : 0;
  return prev;
}
int rsis_get_deterministic(void) { return rsis_deterministic(); }

const char* rsis_error_string(int code) {
  switch (code) {
    case RSIS_OK: return "ok";
    case RSIS_ERR_ARG: return "invalid argument";
    case RSIS_ERR_LAUNCH: return "HIP launch failed";
    case RSIS_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown error";
  }
}

int rsis_conv_uses_bf16(int ks, int stride, int pad, int Cout) { return bf16_geom(ks, stride, pad, Cout) ? 1 : 0; }
int rsis_conv_uses_wino(int ks, int stride, int pad, int Cin, int Cout, int nseg, int lstm_hid) {
  return wino_geom(ks, stride, pad, Cin, Cout, nseg, lstm_hid) ? 1 : 0;
}

long rsis_conv_packed_bytes_fwd(int dtype, int Cout, int ks, int stride, int pad, int nseg, const int* Cseg) {
  if (nseg < 1 || nseg > RSIS_MAX_SRC || !Cseg) return -1;
  const long ldw = rsis_roundup(Cout, RSIS_LDW_ALIGN);
  if (use_wino(dtype, ks, stride, pad, Cseg[0], Cout, nseg, 0)) return (long)Cout * Cseg[0] * 16 * 4;
  if (use_bf16(dtype, ks, stride, pad, Cout)) return (long)bf16_rows(ks, nseg, Cseg) * ldw * 16;
  if (use_direct_fwd(ks, stride, pad)) return (long)direct_rows(nseg, Cseg) * ldw * 4;
  int c = 0;
  for (int s = 0; s < nseg; ++s) c += Cseg[s];
  return (long)krows_of(c, ks) * ldw * 4;
}

long rsis_conv_packed_bytes_dgrad(int dtype, int Cout, int ks, int stride, int pad, int c_count) {
  const long ldw = rsis_roundup(c_count, RSIS_LDW_ALIGN);
  if (dtype == RSIS_DTYPE_F32_WINO && wino_geom_dgrad(ks, stride, pad, c_count, Cout)) return (long)Cout * c_count * 16 * 4;
  if (use_bf16(dtype, ks, stride, pad, Cout)) return (long)bf16_rows(ks, 1, &Cout) * ldw * 16;
  if (use_direct(ks, stride, pad)) return (long)direct_rows(1, &Cout) * ldw * 4;
  if (use_direct_s2(ks, stride, pad)) {      // direct layout for one destination, implicit-GEMM layout for several: room for either
    const long a = (long)direct_rows(1, &Cout) * ldw, b = (long)krows_of(Cout, ks) * ldw;
    return (a > b ? a : b) * 4;
  }
  return (long)krows_of(Cout, ks) * ldw * 4;
}

static int check_segments(int Ctot, int nseg, const int* Cseg, const int* Coff) {
  if (nseg < 1 || nseg > RSIS_MAX_SRC || !Cseg) return RSIS_ERR_ARG;
  int base = 0;
  for (int s = 0; s < nseg; ++s) {
    const int off = Coff ? Coff[s] : base;
    if (Cseg[s] < 1 || off < 0 || off + Cseg[s] > Ctot) return RSIS_ERR_ARG;
    base += Cseg[s];
  }
  return RSIS_OK;
}

// the pack kernels index the reference weight with 32-bit arithmetic
static inline bool weight_too_big(int Cout, int Ctot, int ks) { return Cout < 1 || ks < 1 || (long)Cout * Ctot * ks * ks >= (1L << 31); }

int rsis_conv_pack_fwd(const float* W, void* Wp, int Cout, int Ctot, int ks, int stride, int pad, int nseg, const int* Cseg,
                       const int* Coff, int lstm_hid, int dtype, void* stream) {
  if (!W || !Wp || check_segments(Ctot, nseg, Cseg, Coff) || weight_too_big(Cout, Ctot, ks)) return RSIS_ERR_ARG;
  if (lstm_hid > 0 && Cout != 4 * lstm_hid) return RSIS_ERR_ARG;
  if (lstm_hid > 0 && dtype == RSIS_DTYPE_BF16 && ks != 3) return RSIS_ERR_UNSUPPORTED;     // (see rsis_convlstm_fwd)
  const int ldw = rsis_roundup(Cout, RSIS_LDW_ALIGN);
  int csum = 0;
  for (int s = 0; s < nseg; ++s) csum += Cseg[s];
  if (use_wino(dtype, ks, stride, pad, csum, Cout, nseg, lstm_hid) && csum == Ctot && (!Coff || Coff[0] == 0))
    return rsis_l_pack(7, W, Wp, Cout, Ctot, ks, nseg, Cseg, Coff, ldw, (Cout / 32) * (Ctot / 8), 0, (hipStream_t)stream);
  if (dtype == RSIS_DTYPE_F32_WINO && wino_geom(ks, stride, pad, csum, Cout, nseg, lstm_hid)) return RSIS_ERR_UNSUPPORTED;   // (a channel subset)
  if (use_bf16(dtype, ks, stride, pad, Cout))
    return rsis_l_pack(5, W, Wp, Cout, Ctot, ks, nseg, Cseg, Coff, ldw, bf16_rows(ks, nseg, Cseg), lstm_hid, (hipStream_t)stream);
  if (use_direct_fwd(ks, stride, pad))
    return rsis_l_pack(2, W, Wp, Cout, Ctot, ks, nseg, Cseg, Coff, ldw, direct_rows(nseg, Cseg), lstm_hid, (hipStream_t)stream);
  return rsis_l_pack(0, W, Wp, Cout, Ctot, ks, nseg, Cseg, Coff, ldw, krows_of(csum, ks), lstm_hid, (hipStream_t)stream);
}

int rsis_conv_pack_dgrad(const float* W, void* Wd, int Cout, int Ctot, int ks, int stride, int pad, int nseg, const int* Cseg,
                         const int* Coff, int lstm_hid, int dtype, void* stream) {
  if (!W || !Wd || check_segments(Ctot, nseg, Cseg, Coff) || weight_too_big(Cout, Ctot, ks)) return RSIS_ERR_ARG;
  if (lstm_hid > 0 && Cout != 4 * lstm_hid) return RSIS_ERR_ARG;
  int csum = 0;
  for (int s = 0; s < nseg; ++s) csum += Cseg[s];
  const int ldw = rsis_roundup(csum, RSIS_LDW_ALIGN);
  if (dtype == RSIS_DTYPE_F32_WINO && wino_geom_dgrad(ks, stride, pad, csum, Cout))
    return rsis_l_pack(8, W, Wd, Cout, Ctot, ks, nseg, Cseg, Coff, ldw, (csum / 32) * (Cout / 8), lstm_hid, (hipStream_t)stream);
  if (use_bf16(dtype, ks, stride, pad, Cout))
    return rsis_l_pack(6, W, Wd, Cout, Ctot, ks, nseg, Cseg, Coff, ldw, bf16_rows(ks, 1, &Cout), lstm_hid, (hipStream_t)stream);
  if (use_direct(ks, stride, pad))
    return rsis_l_pack(3, W, Wd, Cout, Ctot, ks, nseg, Cseg, Coff, ldw, direct_rows(1, &Cout), lstm_hid, (hipStream_t)stream);
  if (use_direct_s2(ks, stride, pad) && nseg == 1)      // (the parity-class kernel writes one destination)
    return rsis_l_pack(4, W, Wd, Cout, Ctot, ks, nseg, Cseg, Coff, ldw, direct_rows(1, &Cout), lstm_hid, (hipStream_t)stream);
  return rsis_l_pack(1, W, Wd, Cout, Ctot, ks, nseg, Cseg, Coff, ldw, krows_of(Cout, ks), lstm_hid, (hipStream_t)stream);
}

int rsis_conv_out_wgrad(const float* dy, const float* x, float* dW, float* db, int B, int Cin, int H, int W, void* stream) {
  if (!dy || !x || !dW || B < 1) return RSIS_ERR_ARG;
  if (!rsis_c1_supported(Cin) || W % 4 != 0) return RSIS_ERR_UNSUPPORTED;
  return rsis_l_c1_wgrad(dy, x, dW, db, B, Cin, H, W, 1, (hipStream_t)stream);
}

// conv_out over all T timesteps of a decoded sequence, one launch each way (rsis_hip.h)
static inline bool c1_seq_ok(int T, int B, int Cin, int H, int W) {
  return T >= 1 && B >= 1 && rsis_c1_supported(Cin) && W % 4 == 0 && (long)T * B * Cin * H * W < (1L << 40);
}
int rsis_conv_out_seq_fwd(const float* x, const void* Wp, const float* bias, float* y, int T, int B, int Cin, int H, int W, void* stream) {
  if (!x || !Wp || !y) return RSIS_ERR_ARG;
  if (!c1_seq_ok(T, B, Cin, H, W)) return RSIS_ERR_UNSUPPORTED;
  return rsis_l_c1_fwd(x, (const float*)Wp, rsis_roundup(1, RSIS_LDW_ALIGN), bias, y, T * B, Cin, H, W, T, (hipStream_t)stream);
}
int rsis_conv_out_seq_dgrad(const float* dy, const void* Wd, float* dx, int T, int B, int Cin, int H, int W, void* stream) {
  if (!dy || !Wd || !dx) return RSIS_ERR_ARG;
  if (!c1_seq_ok(T, B, Cin, H, W)) return RSIS_ERR_UNSUPPORTED;
  return rsis_l_c1_dgrad(dy, (const float*)Wd, rsis_roundup(Cin, RSIS_LDW_ALIGN), dx, T * B, Cin, H, W, T, (hipStream_t)stream);
}
int rsis_conv_out_seq_wgrad(const float* dy, const float* x, float* dW, float* db, int T, int B, int Cin, int H, int W, void* stream) {
  if (!dy || !x || !dW) return RSIS_ERR_ARG;
  if (!c1_seq_ok(T, B, Cin, H, W)) return RSIS_ERR_UNSUPPORTED;
  return rsis_l_c1_wgrad(dy, x, dW, db, T * B, Cin, H, W, T, (hipStream_t)stream);
}
int rsis_sum_leading(const float* x, float* y, int T, long n, void* stream) {
  if (!x || !y || T < 1 || n < 1) return RSIS_ERR_ARG;
  return rsis_l_sum_leading(x, y, T, n, (hipStream_t)stream);
}

int rsis_affine_nearest(const float* x, float* y, const float* mat, int mat_rows, int N, int C, int H, int W, void* stream) {
  if (!x || !y || !mat || x == y || N < 1 || C < 1 || H < 1 || W < 1 || (mat_rows != 2 && mat_rows != 3)) return RSIS_ERR_ARG;
  if ((long)H * W >= (1L << 31) || N > 65535) return RSIS_ERR_UNSUPPORTED;
  return rsis_l_affine_nearest(x, y, mat, N, C, H, W, mat_rows * 3, (hipStream_t)stream);
}

int rsis_conv_pack_job_fill(rsis_pack_job* j) {
  if (!j || check_segments(j->Ctot, j->nseg, j->Cseg, j->Coff) || weight_too_big(j->Cout, j->Ctot, j->ks)) return -1;
  if (j->lstm_hid > 0 && j->Cout != 4 * j->lstm_hid) return -1;
  if (j->lstm_hid > 0 && j->dtype == RSIS_DTYPE_BF16 && j->ks != 3) return -1;                  // (see rsis_convlstm_fwd)
  int csum = 0;
  for (int s = 0; s < j->nseg; ++s) csum += j->Cseg[s];
  const bool wino = j->dgrad ? (j->dtype == RSIS_DTYPE_F32_WINO && wino_geom_dgrad(j->ks, j->stride, j->pad, csum, j->Cout))
                             : use_wino(j->dtype, j->ks, j->stride, j->pad, csum, j->Cout, j->nseg, j->lstm_hid);
  if (wino && !j->dgrad && (csum != j->Ctot || j->Coff[0] != 0)) return -1;
  if (wino) {
    j->ldw = rsis_roundup(j->dgrad ? csum : j->Cout, RSIS_LDW_ALIGN);
    j->imode = j->dgrad ? 8 : 7;
    j->krows = j->dgrad ? (csum / 32) * (j->Cout / 8) : (j->Cout / 32) * (csum / 8);
    return rsis_l_pack_blocks(j->imode, j->krows, j->ldw, j->ks);
  }
  if (!j->dgrad) {
    j->ldw = rsis_roundup(j->Cout, RSIS_LDW_ALIGN);
    if (use_bf16(j->dtype, j->ks, j->stride, j->pad, j->Cout)) { j->imode = 5; j->krows = bf16_rows(j->ks, j->nseg, j->Cseg); }
    else if (use_direct_fwd(j->ks, j->stride, j->pad)) { j->imode = 2; j->krows = direct_rows(j->nseg, j->Cseg); }
    else { j->imode = 0; j->krows = krows_of(csum, j->ks); }
  } else {
    j->ldw = rsis_roundup(csum, RSIS_LDW_ALIGN);
    if (use_bf16(j->dtype, j->ks, j->stride, j->pad, j->Cout)) { j->imode = 6; j->krows = bf16_rows(j->ks, 1, &j->Cout); }
    else if (use_direct(j->ks, j->stride, j->pad)) { j->imode = 3; j->krows = direct_rows(1, &j->Cout); }
    else if (use_direct_s2(j->ks, j->stride, j->pad) && j->nseg == 1) { j->imode = 4; j->krows = direct_rows(1, &j->Cout); }
    else { j->imode = 1; j->krows = krows_of(j->Cout, j->ks); }
  }
  return rsis_l_pack_blocks(j->imode, j->krows, j->ldw, j->ks);
}

int rsis_conv_pack_batch(const rsis_pack_job* jobs_dev, int njobs, int total_blocks, void* stream) {
  if (!jobs_dev || njobs < 1 || total_blocks < 1) return RSIS_ERR_ARG;
  return rsis_l_pack_batch(jobs_dev, njobs, total_blocks, (hipStream_t)stream);
}

static int fill_sources(ConvArgs& a, const float* const* src, const int* Csrc, int nsrc, int ks, bool allow_empty = false) {
  if (nsrc < (allow_empty ? 0 : 1) || nsrc > RSIS_MAX_SRC || (nsrc > 0 && (!src || !Csrc))) return RSIS_ERR_ARG;
  a.nsrc = nsrc;
  a.K = 0;
  for (int s = 0; s < RSIS_MAX_SRC; ++s) { a.src[s] = nsrc > 0 ? src[0] : nullptr; a.C[s] = 0; }
  for (int s = 0; s < nsrc; ++s) {
    if (!src[s] || Csrc[s] < 1) return RSIS_ERR_ARG;
    a.src[s] = src[s]; a.C[s] = Csrc[s]; a.K += Csrc[s] * ks * ks;
  }
  return RSIS_OK;
}

// the eval-mode BatchNorm folded into the epilogue (rsis_conv2d_fwd_bn_eval), or null
struct BnEval { const float* gamma; const float* beta; const float* mean; const float* var; float eps; int relu; };

static int conv2d_fwd_impl(const float* const* src, const int* Csrc, int nsrc, int B, int H, int W, const void* Wp, int Cout,
                           int ks, int stride, int pad, const float* bias, const float* addend, float* out, int Ho, int Wo,
                           int tile, int dtype, void* stream, const BnEval* bn) {
  const bool allow_splitk = tile >= 100;     // tile + 100: the caller accepts a split-K (atomic, order-nondeterministic) sum
  tile %= 100;
  ConvArgs a = {};
  int rc = fill_sources(a, src, Csrc, nsrc, ks);
  if (rc) return rc;
  if (!Wp || !out || B < 1 || stride < 1) return RSIS_ERR_ARG;
  if (Ho != (H + 2 * pad - ks) / stride + 1 || Wo != (W + 2 * pad - ks) / stride + 1) return RSIS_ERR_ARG;
  a.B = B; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.stride = stride; a.pad = pad; a.sshift = 0;
  a.wp = (const float*)Wp; a.ldw = rsis_roundup(Cout, RSIS_LDW_ALIGN); a.Cout = Cout; a.bias = bias; a.addend = addend;
  a.dst[0] = out; a.Cd[0] = Cout; a.ndst = 1;
  a.ostride = 1; a.oH = Ho; a.oW = Wo; a.ksplit = 1;
  a.precise = allow_splitk ? 0 : 1;       // (tile + 100 marks a training call; everything else is the inference / parity path)
  if (bn) {
    // the fp32 single-destination epilogues of conv_igemm.hip / conv3x3_direct.hip only (inference calls): everything else is
    // refused BEFORE anything is launched, and the caller runs the conv and the BatchNorm as two launches
    if (!bn->gamma || !bn->beta || !bn->mean || !bn->var) return RSIS_ERR_ARG;
    // (a conv of a bf16 model that the library runs on the fp32 kernels anyway -- the 7x7 stem -- is covered: same packed layout)
    if (allow_splitk || Cout % 4 != 0 || (size_t)B * Cout * Ho * Wo * 4 >= (1ull << 31)) return RSIS_ERR_UNSUPPORTED;
    if (dtype != RSIS_DTYPE_F32 && (dtype != RSIS_DTYPE_BF16 || use_bf16(dtype, ks, stride, pad, Cout))) return RSIS_ERR_UNSUPPORTED;
    if (use_direct(ks, stride, pad) && Cout == 1) return RSIS_ERR_UNSUPPORTED;
    a.ep_gamma = bn->gamma; a.ep_beta = bn->beta; a.ep_mean = bn->mean; a.ep_var = bn->var; a.ep_eps = bn->eps; a.ep_relu = bn->relu ? 1 : 0;
  }
  if (use_wino(dtype, ks, stride, pad, Csrc[0], Cout, nsrc, 0))
    return rsis_launch_conv_wino(src[0], Wp, bias, addend, out, B, Csrc[0], Cout, H, W, (hipStream_t)stream, a.precise);
  if (use_bf16(dtype, ks, stride, pad, Cout)) {
    if (stride != 1) return RSIS_ERR_UNSUPPORTED;      // strided 1x1: run the stride-1 form on a sub-sampled input
    if (ks == 1 && nsrc != 1) return RSIS_ERR_UNSUPPORTED;
    if (ks == 3) {     // deep-K convs on tiny maps (sk5): split over the channel chunks into a zeroed output
      const bool splitk_ok = conv_splitk_ok();
      int nq = 0;
      for (int s = 0; s < nsrc; ++s) nq += (Csrc[s] + RSIS_CKB3 - 1) / RSIS_CKB3;
      const long px_tiles = (long)B * rsis_cdiv(H, 8) * rsis_cdiv(W, W <= 8 ? 8 : 16);
      if (allow_splitk && splitk_ok && !addend && nq >= 16 && px_tiles * rsis_cdiv(Cout, 64) < 160) {
        if (rsis_zero_async(out, sizeof(float) * (size_t)B * Cout * Ho * Wo, (hipStream_t)stream) != RSIS_OK) return RSIS_ERR_LAUNCH;
        a.ksplit = 0;
      }
    }
    return rsis_launch_conv_bf16(a, ks, 0, direct_variant(tile), (hipStream_t)stream);
  }
  if (use_direct(ks, stride, pad) && Cout == 1 && nsrc == 1 && !addend && rsis_c1_supported(Csrc[0]) && W % 4 == 0)   // conv_out: HBM-bound VALU kernel
    return rsis_l_c1_fwd(src[0], (const float*)Wp, a.ldw, bias, out, B, Csrc[0], H, W, 1, (hipStream_t)stream);
  if (use_direct(ks, stride, pad)) {
    // deep-K convs on tiny maps (sk5: 2048x9 deep, 64 blocks) are split over the channel chunks: zero the output here and let
    // the launcher decide (a.ksplit = 0 means "split allowed, output is zeroed")
    int nq = 0;
    for (int s = 0; s < nsrc; ++s) nq += (Csrc[s] + RSIS_CK - 1) / RSIS_CK;
    const long px_tiles = (long)B * rsis_cdiv(H, 8) * rsis_cdiv(W, W <= 8 ? 8 : 16);
    const bool splitk_ok = conv_splitk_ok();   // A/B switch / deterministic mode
    if (allow_splitk && splitk_ok && nq >= 32 && px_tiles * rsis_cdiv(Cout, 64) < 160) {
      if (rsis_zero_async(out, sizeof(float) * (size_t)B * Cout * Ho * Wo, (hipStream_t)stream) != RSIS_OK) return RSIS_ERR_LAUNCH;
      a.ksplit = 0;
    }
    return rsis_launch_conv3x3_direct(a, 0, direct_variant(tile), (hipStream_t)stream);
  }
  if (use_direct_f2(ks, stride, pad))        // 3x3 / stride 2 forward: the direct kernel on a (2T+1)^2 patch (EPI_F2 = 3)
    return rsis_launch_conv3x3_direct(a, 3, direct_variant(tile), (hipStream_t)stream);
  return rsis_launch_conv_igemm(a, ks, false, 0, tile, (hipStream_t)stream);
}

int rsis_conv2d_fwd(const float* const* src, const int* Csrc, int nsrc, int B, int H, int W, const void* Wp, int Cout,
                    int ks, int stride, int pad, const float* bias, const float* addend, float* out, int Ho, int Wo,
                    int tile, int dtype, void* stream) {
  return conv2d_fwd_impl(src, Csrc, nsrc, B, H, W, Wp, Cout, ks, stride, pad, bias, addend, out, Ho, Wo, tile, dtype, stream, nullptr);
}

int rsis_conv2d_fwd_bn_eval(const float* const* src, const int* Csrc, int nsrc, int B, int H, int W, const void* Wp, int Cout,
                            int ks, int stride, int pad, const float* bias, const float* addend, const float* gamma, const float* beta,
                            const float* running_mean, const float* running_var, float eps, int relu, float* out, int Ho, int Wo,
                            int tile, int dtype, void* stream) {
  const BnEval bn = {gamma, beta, running_mean, running_var, eps, relu};
  return conv2d_fwd_impl(src, Csrc, nsrc, B, H, W, Wp, Cout, ks, stride, pad, bias, addend, out, Ho, Wo, tile % 100, dtype, stream, &bn);
}

int rsis_conv2d_dgrad(const float* dy, int B, int Cout, int Hy, int Wy, const void* Wd, int Cin_packed, int ks, int stride,
                      int pad, float* const* dx, const int* Cdx, int ndst, int Hx, int Wx, const float* addend, int tile,
                      int dtype, void* stream) {
  if (!dy || !Wd || !dx || !Cdx || ndst < 1 || ndst > RSIS_MAX_SRC) return RSIS_ERR_ARG;
  // addend: stride 1 (any kernel the epilogue supports), or the strided 1x1 scatter accumulating in place (addend == dx[0])
  const bool inplace = addend && ndst == 1 && stride > 1 && ks == 1 && pad == 0 && addend == dx[0];
  if (addend && !inplace && (ndst != 1 || stride != 1)) return RSIS_ERR_UNSUPPORTED;
  if (stride != 1 && stride != 2 && stride != 4) return RSIS_ERR_UNSUPPORTED;
  ConvArgs a = {};
  const float* srcs[1] = {dy};
  const int cs[1] = {Cout};
  int rc = fill_sources(a, srcs, cs, 1, ks);
  if (rc) return rc;
  int ctot = 0;
  for (int i = 0; i < ndst; ++i) { if (!dx[i] || Cdx[i] < 1) return RSIS_ERR_ARG; a.dst[i] = dx[i]; a.Cd[i] = Cdx[i]; ctot += Cdx[i]; }
  a.ndst = ndst;
  a.B = B; a.H = Hy; a.W = Wy; a.Ho = Hx; a.Wo = Wx; a.stride = stride; a.pad = pad; a.sshift = log2i(stride);
  if (ctot > Cin_packed) return RSIS_ERR_ARG;
  a.ostride = 1; a.oH = Hx; a.oW = Wx; a.ksplit = 1;
  a.wp = (const float*)Wd; a.ldw = rsis_roundup(Cin_packed, RSIS_LDW_ALIGN); a.Cout = ctot; a.bias = nullptr; a.addend = addend;
  if (dtype == RSIS_DTYPE_F32_WINO && wino_geom_dgrad(ks, stride, pad, Cin_packed, Cout)) {
    // the data gradient of a Winograd conv is the same kernel on the transposed, rotated weights (pack mode 8): dy plays the input.
    // One destination (any leading multiple of 32 channels of the concat) or two splitting it at a multiple of 32.
    if (ndst > 2 || ctot % 32 != 0 || Cdx[0] % 32 != 0 || Hx != Hy || Wx != Wy || (ndst == 2 && addend)) return RSIS_ERR_UNSUPPORTED;
    if (ndst == 1) return rsis_launch_conv_wino(dy, Wd, nullptr, addend, dx[0], B, Cout, ctot, Hy, Wy, (hipStream_t)stream, 0);
    const float* xs[1] = {dy}; const void* us[1] = {Wd}; float* y0[1] = {dx[0]}; float* y1[1] = {dx[1]};
    const int b1[1] = {B}, c1[1] = {Cout}, co1[1] = {ctot}, c01[1] = {Cdx[0]}, h1[1] = {Hy}, w1[1] = {Wy};
    return rsis_launch_conv_wino_group(1, xs, us, y0, y1, b1, c1, co1, c01, h1, w1, (hipStream_t)stream);
  }
  if (use_bf16(dtype, ks, stride, pad, Cout)) {
    if (ks == 3) {
      if (Hx != Hy || Wx != Wy) return RSIS_ERR_ARG;
      const bool splitk_ok = conv_splitk_ok();
      const int nq = (Cout + RSIS_CKB3 - 1) / RSIS_CKB3;
      const long px_tiles = (long)B * rsis_cdiv(Hx, 8) * rsis_cdiv(Wx, Wx <= 8 ? 8 : 16);
      if (splitk_ok && !addend && nq >= 16 && px_tiles * rsis_cdiv(ctot, 64) < 160) {
        for (int i = 0; i < ndst; ++i)
          if (rsis_zero_async(dx[i], sizeof(float) * (size_t)B * Cdx[i] * Hx * Wx, (hipStream_t)stream) != RSIS_OK) return RSIS_ERR_LAUNCH;
        a.ksplit = 0;
      }
      return rsis_launch_conv_bf16(a, 3, 0, direct_variant(tile), (hipStream_t)stream);
    }
    // 1x1: a GEMM over the dy grid; stride > 1 scatters its rows to every stride-th pixel of a zeroed (or, in place, the parked) dx
    if (stride > 1) {
      if (Hy != (Hx - 1) / stride + 1 || Wy != (Wx - 1) / stride + 1) return RSIS_ERR_ARG;
      for (int i = 0; i < ndst && !inplace; ++i)
        if (rsis_zero_async(dx[i], sizeof(float) * (size_t)B * Cdx[i] * Hx * Wx, (hipStream_t)stream) != RSIS_OK) return RSIS_ERR_LAUNCH;
      a.ostride = stride;
    } else if (Hx != Hy || Wx != Wy) return RSIS_ERR_ARG;
    a.Ho = Hy; a.Wo = Wy; a.stride = 1; a.sshift = 0;
    return rsis_launch_conv_bf16(a, 1, 0, direct_variant(tile), (hipStream_t)stream);
  }
  if (use_direct(ks, stride, pad)) {
    if (Hx != Hy || Wx != Wy) return RSIS_ERR_ARG;
    if (Cout == 1 && ndst == 1 && Cin_packed == Cdx[0] && rsis_c1_supported(Cdx[0]) && !addend)
      return rsis_l_c1_dgrad(dy, (const float*)Wd, a.ldw, dx[0], B, Cdx[0], Hx, Wx, 1, (hipStream_t)stream);
    {  // deep-K data gradients on tiny maps (ConvLSTM level 0: 512 gate rows x 9 taps on 8x8): split over the channel chunks
      const bool splitk_ok = conv_splitk_ok();
      const int nq = (Cout + RSIS_CK - 1) / RSIS_CK;
      const long px_tiles = (long)B * rsis_cdiv(Hx, 8) * rsis_cdiv(Wx, Wx <= 8 ? 8 : 16);
      if (splitk_ok && !addend && nq >= 32 && px_tiles * rsis_cdiv(ctot, 64) < 160) {
        for (int i = 0; i < ndst; ++i)
          if (rsis_zero_async(dx[i], sizeof(float) * (size_t)B * Cdx[i] * Hx * Wx, (hipStream_t)stream) != RSIS_OK) return RSIS_ERR_LAUNCH;
        a.ksplit = 0;
      }
    }
    return rsis_launch_conv3x3_direct(a, 0, direct_variant(tile), (hipStream_t)stream);
  }
  if (use_direct_s2(ks, stride, pad) && ndst == 1 && Cdx[0] == Cin_packed && Hy == (Hx + 1) / 2 && Wy == (Wx + 1) / 2) {
    // 3x3 / stride 2: the direct kernel walks the dy grid and scatters the four input-pixel parity classes (every dx pixel is
    // written exactly once: no memset)
    a.Ho = Hy; a.Wo = Wy; a.oH = Hx; a.oW = Wx;
    return rsis_launch_conv3x3_direct(a, 2, direct_variant(tile), (hipStream_t)stream);
  }
  if (ks == 1 && pad == 0 && stride > 1) {
    // 1x1 / stride-s data gradient: only the (s*ho, s*wo) input pixels receive a gradient -> zero dx, then run the plain
    // 1x1 GEMM over the dy grid and scatter its rows to those pixels (instead of gathering with 1/s^2 useful taps)
    for (int i = 0; i < ndst && !inplace; ++i)
      if (rsis_zero_async(dx[i], sizeof(float) * (size_t)B * Cdx[i] * Hx * Wx, (hipStream_t)stream) != RSIS_OK) return RSIS_ERR_LAUNCH;
    a.Ho = Hy; a.Wo = Wy; a.stride = 1; a.sshift = 0; a.ostride = stride;
    return rsis_launch_conv_igemm(a, ks, false, 0, tile, (hipStream_t)stream);
  }
  return rsis_launch_conv_igemm(a, ks, true, 0, tile, (hipStream_t)stream);
}

// route of one weight gradient: 0 conv_out vector kernel, 1 bf16 tiled, 2 exact-f32 LDS-DMA tiled, 3 generic split-K implicit GEMM
static int wgrad_route(WgradArgs& a, int ks, int lstm_hid, int dtype) {
  if (dtype == RSIS_DTYPE_BF16_BLK) {     // dy / x are channel-blocked bf16 tensors: the bf16 kernels with the blk staging, or nothing
    a.blk = 1;
    return rsis_wgrad_bf16_supported(a, ks) ? 1 : -1;
  }
  if (use_direct(ks, a.stride, a.pad) && a.Cout == 1 && lstm_hid == 0 && rsis_c1_supported(a.Cs) && a.H == a.Ho && a.W == a.Wo && a.W % 4 == 0)
    return 0;
  // stride-1 "same" convs on tile-aligned maps: the LDS-DMA tiled kernel (conv_wgrad_tiled.hip); RSIS_WGRAD_TILED=0 forces the
  // generic split-K implicit GEMM (conv_wgrad.hip), which also covers every other shape
  if (dtype == RSIS_DTYPE_BF16 && rsis_wgrad_bf16_supported(a, ks)) return 1;
  static const bool tiled_ok = !(getenv("RSIS_WGRAD_TILED") && getenv("RSIS_WGRAD_TILED")[0] == '0');
  if (tiled_ok && rsis_wgrad_tiled_supported(a, ks)) return 2;
  return 3;
}
static int wgrad_fill(WgradArgs& a, const float* dy, const float* x, float* dW, int B, int Cs, int H, int W, int Cout, int Ho, int Wo,
                      int ks, int stride, int pad, int Ctot, int c_off, int lstm_hid) {
  if (!dy || !x || !dW || c_off < 0 || c_off + Cs > Ctot) return RSIS_ERR_ARG;
  a = WgradArgs{};
  a.dy = dy; a.x = x; a.dw = dW; a.B = B; a.Cs = Cs; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
  a.stride = stride; a.pad = pad; a.ldo = Ctot * ks * ks; a.n_off = c_off * ks * ks; a.interleave_hid = lstm_hid;
  return RSIS_OK;
}
static int wgrad_launch_one(WgradArgs& a, int route, int ks, hipStream_t st) {
  switch (route) {
    case 0: return rsis_l_c1_wgrad(a.dy, a.x, a.dw + a.n_off, nullptr, a.B, a.Cs, a.H, a.W, 1, st);
    case 1: return rsis_launch_conv_wgrad_bf16(a, ks, st);
    case 2: return rsis_launch_conv_wgrad_tiled(a, ks, st);
    default: return rsis_launch_conv_wgrad(a, ks, st);
  }
}

int rsis_conv2d_wgrad(const float* dy, const float* x, float* dW, int B, int Cs, int H, int W, int Cout, int Ho, int Wo,
                      int ks, int stride, int pad, int Ctot, int c_off, int lstm_hid, int dtype, void* stream) {
  WgradArgs a;
  const int rc = wgrad_fill(a, dy, x, dW, B, Cs, H, W, Cout, Ho, Wo, ks, stride, pad, Ctot, c_off, lstm_hid);
  if (rc) return rc;
  const int route = wgrad_route(a, ks, lstm_hid, dtype);
  if (route < 0) return RSIS_ERR_UNSUPPORTED;
  return wgrad_launch_one(a, route, ks, (hipStream_t)stream);
}

int rsis_conv2d_wgrad_batch(const rsis_wgrad_job* jobs, int njobs, void* stream) {
  if (!jobs || njobs < 1) return RSIS_ERR_ARG;
  // [0], [1]: exact-f32 tiled, 1x1 / 3x3; [2], [3]: bf16, 1x1 / 3x3
  WgradArgs* tiled[4] = {(WgradArgs*)malloc(sizeof(WgradArgs) * njobs), (WgradArgs*)malloc(sizeof(WgradArgs) * njobs),
                         (WgradArgs*)malloc(sizeof(WgradArgs) * njobs), (WgradArgs*)malloc(sizeof(WgradArgs) * njobs)};
  int nt[4] = {0, 0, 0, 0};
  int rc = (tiled[0] && tiled[1] && tiled[2] && tiled[3]) ? RSIS_OK : RSIS_ERR_LAUNCH;
  for (int j = 0; j < njobs && rc == RSIS_OK; ++j) {
    const rsis_wgrad_job& q = jobs[j];
    WgradArgs a;
    rc = wgrad_fill(a, q.dy, q.x, q.dW, q.B, q.Cs, q.H, q.W, q.Cout, q.Ho, q.Wo, q.ks, q.stride, q.pad, q.Ctot, q.c_off, q.lstm_hid);
    if (rc) break;
    const int route = wgrad_route(a, q.ks, q.lstm_hid, q.dtype);
    if (route < 0) { rc = RSIS_ERR_UNSUPPORTED; break; }
    // (deterministic mode: one launch per job, in order -- several jobs may accumulate into the SAME dW (a conv applied at every
    //  decoder timestep), and inside one grid their atomics would land in a run-dependent order)
    if ((route == 1 || route == 2) && (q.ks == 1 || q.ks == 3) && !rsis_deterministic()) {      // grouped below
      const int li = (route == 1 ? 2 : 0) + (q.ks == 3);
      tiled[li][nt[li]++] = a;
    } else rc = wgrad_launch_one(a, route, q.ks, (hipStream_t)stream);
  }
  if (rc == RSIS_OK) rc = rsis_launch_conv_wgrad_tiled_group(tiled[0], nt[0], 1, (hipStream_t)stream);
  if (rc == RSIS_OK) rc = rsis_launch_conv_wgrad_tiled_group(tiled[1], nt[1], 3, (hipStream_t)stream);
  if (rc == RSIS_OK) rc = rsis_launch_conv_wgrad_bf16_group(tiled[2], nt[2], 1, (hipStream_t)stream);
  if (rc == RSIS_OK) rc = rsis_launch_conv_wgrad_bf16_group(tiled[3], nt[3], 3, (hipStream_t)stream);
  for (int i = 0; i < 4; ++i) free(tiled[i]);
  return rc;
}

int rsis_bias_grad(const float* dy, float* db, int B, int C, int HW, int lstm_hid, void* stream) {
  if (!dy || !db) return RSIS_ERR_ARG;
  return rsis_l_channel_sum(dy, db, B, C, HW, lstm_hid, (hipStream_t)stream);
}

// argument checks + ConvArgs of one fused ConvLSTM forward; route: 0 bf16 kernel, 1 direct 3x3 (exact f32), 2 implicit GEMM
static int lstm_fill(ConvArgs& a, int& route, const float* const* src, const int* Csrc, int nsrc, int B, int H, int W, const void* Wp,
                     const float* bias_packed, const float* addend, const float* c_prev, float* h_out, float* c_out, float* act_out, int hid,
                     int ks, int pad, int dtype, unsigned long long* side_key = nullptr) {
  a = ConvArgs{};
  int rc = fill_sources(a, src, Csrc, nsrc, ks, /*allow_empty=*/addend != nullptr);   // nsrc == 0: gates = addend only
  if (rc) return rc;
  if (!Wp || !h_out || !c_out || hid < 1 || B < 1) return RSIS_ERR_ARG;
  if (2 * pad != ks - 1) return RSIS_ERR_UNSUPPORTED;   // "same" conv only (the state keeps its size)
  // bf16 gates: 3x3 only (the fused-cell epilogue of conv_bf16_kernel); rsis_conv_pack_fwd refuses the same combination, so a
  // bf16-packed buffer can never reach the f32 kernels below
  if (dtype == RSIS_DTYPE_BF16 && ks != 3) return RSIS_ERR_UNSUPPORTED;
  a.B = B; a.H = H; a.W = W; a.Ho = H; a.Wo = W; a.stride = 1; a.pad = pad; a.sshift = 0;
  a.wp = (const float*)Wp; a.ldw = rsis_roundup(4 * hid, RSIS_LDW_ALIGN); a.Cout = 4 * hid; a.bias = bias_packed; a.addend = addend;
  a.ndst = 1; a.dst[0] = nullptr; a.Cd[0] = 4 * hid;
  a.hid = hid; a.c_prev = c_prev; a.h_out = h_out; a.c_out = c_out; a.act_out = act_out;
  a.ostride = 1; a.oH = H; a.oW = W; a.ksplit = 1;
  route = use_bf16(dtype, ks, 1, pad, 4 * hid) ? 0 : (use_direct(ks, 1, pad) ? 1 : 2);
  a.side_key = side_key;
  a.precise = act_out ? 0 : 1;            // no saved gates: an inference call
  if (side_key && (route == 2 || (long)H * W >= 0x7FFFFFFFL)) return RSIS_ERR_UNSUPPORTED;   // the fused side max-pool lives in the 3x3 epilogues
  return RSIS_OK;
}
static int lstm_launch_one(ConvArgs& a, int route, int ks, int tile, hipStream_t st) {
  if (route == 0) return rsis_launch_conv_bf16(a, 3, 1, direct_variant(tile), st);
  if (route == 1) return rsis_launch_conv3x3_direct(a, 1, direct_variant(tile), st);
  return rsis_launch_conv_igemm(a, ks, false, 1, tile, st);
}

int rsis_convlstm_fwd(const float* const* src, const int* Csrc, int nsrc, int B, int H, int W, const void* Wp,
                      const float* bias_packed, const float* addend, const float* c_prev, float* h_out, float* c_out,
                      float* act_out, int hid, int ks, int pad, int tile, int dtype, void* stream) {
  ConvArgs a;
  int route = 0;
  const int rc = lstm_fill(a, route, src, Csrc, nsrc, B, H, W, Wp, bias_packed, addend, c_prev, h_out, c_out, act_out, hid, ks, pad, dtype);
  if (rc) return rc;
  return lstm_launch_one(a, route, ks, tile, (hipStream_t)stream);
}

int rsis_convlstm_fwd_batch(const rsis_lstm_job* jobs, int njobs, void* stream) {
  if (!jobs || njobs < 1) return RSIS_ERR_ARG;
  ConvArgs* grp = (ConvArgs*)malloc(sizeof(ConvArgs) * njobs);
  int* force = (int*)malloc(sizeof(int) * njobs);
  if (!grp || !force) { free(grp); free(force); return RSIS_ERR_LAUNCH; }
  int ng = 0, rc = RSIS_OK;
  for (int j = 0; j < njobs && rc == RSIS_OK; ++j) {
    const rsis_lstm_job& q = jobs[j];
    if (q.nsrc < 0 || q.nsrc > RSIS_MAX_SRC) { rc = RSIS_ERR_ARG; break; }
    ConvArgs a;
    int route = 0;
    rc = lstm_fill(a, route, q.src, q.Csrc, q.nsrc, q.B, q.H, q.W, q.Wp, q.bias_packed, q.addend, q.c_prev, q.h_out, q.c_out, q.act_out,
                   q.hid, q.ks, q.pad, q.dtype, q.side_key);
    if (rc) break;
    // grouped: the exact-f32 direct kernel with 32-bit epilogue addressing (its buffer-descriptor cell update); everything else --
    // and every job in the deterministic mode, where launches are kept as the single-call path issues them -- one by one
    const bool groupable = route == 1 && (size_t)q.B * 4 * q.hid * q.H * q.W * 4 < (1ull << 31) && !rsis_deterministic();
    if (groupable) { force[ng] = direct_variant(q.tile % 100); grp[ng++] = a; }
    else rc = lstm_launch_one(a, route, q.ks, q.tile, (hipStream_t)stream);
  }
  if (rc == RSIS_OK && ng == 1) rc = rsis_launch_conv3x3_direct(grp[0], 1, force[0], (hipStream_t)stream);
  else if (rc == RSIS_OK && ng > 1) rc = rsis_launch_convlstm_direct_group(grp, ng, force, (hipStream_t)stream);
  free(grp); free(force);
  return rc;
}

int rsis_convlstm_bwd_gates(const float* dh, const float* dh2, const float* dc_next, const float* act, const float* c_prev,
                            const float* c, float* da, float* dc_prev, float* da_sum, int B, int hid, int HW, void* stream) {
  if (!act || !c || !da) return RSIS_ERR_ARG;
  return rsis_l_lstm_bwd(dh, dh2, dc_next, act, c_prev, c, da, dc_prev, da_sum, B, hid, HW, (hipStream_t)stream);
}

int rsis_upsample_bilinear_ac_fwd(const float* x, float* y, long BC, int Hi, int Wi, int Ho, int Wo, void* stream) {
  if (!x || !y) return RSIS_ERR_ARG;
  return rsis_l_upsample_fwd(x, y, BC, Hi, Wi, Ho, Wo, (hipStream_t)stream);
}
int rsis_upsample_bilinear_ac_bwd(const float* dy, float* dx, long BC, int Hi, int Wi, int Ho, int Wo, void* stream) {
  if (!dy || !dx) return RSIS_ERR_ARG;
  return rsis_l_upsample_bwd(dy, dx, BC, Hi, Wi, Ho, Wo, nullptr, nullptr, (hipStream_t)stream);
}
int rsis_upsample_maxpool_bwd(const float* dy, const float* dpool, const int* argmax, float* dx, long BC, int Hi, int Wi, int Ho,
                              int Wo, void* stream) {
  if (!dy || !dx || !dpool || !argmax) return RSIS_ERR_ARG;
  return rsis_l_upsample_bwd(dy, dx, BC, Hi, Wi, Ho, Wo, dpool, argmax, (hipStream_t)stream);
}
int rsis_global_maxpool_fwd(const float* x, float* y, int* argmax, long BC, int HW, void* stream) {
  if (!x || !y || !argmax) return RSIS_ERR_ARG;
  return rsis_l_gmax_fwd(x, y, argmax, BC, HW, (hipStream_t)stream);
}
int rsis_global_maxpool_bwd(const float* dy, const int* argmax, float* dx, long BC, int HW, void* stream) {
  if (!dy || !dx || !argmax) return RSIS_ERR_ARG;
  return rsis_l_gmax_bwd(dy, argmax, dx, BC, HW, (hipStream_t)stream);
}
int rsis_global_maxpool_bwd_add(const float* dy, const int* argmax, float* dx, long BC, int HW, void* stream) {
  if (!dy || !dx || !argmax) return RSIS_ERR_ARG;
  return rsis_l_gmax_bwd_add(dy, argmax, dx, BC, HW, (hipStream_t)stream);
}
int rsis_bn_fwd(const float* x, const float* res, float* y, double* stats, const float* gamma, const float* beta,
                float* running_mean, float* running_var, float* save_mean, float* save_rstd, int B, int C, int HW,
                float eps, float momentum, int relu, int train, void* stream) {
  if (!x || !y || !gamma || !beta || !running_mean || !running_var) return RSIS_ERR_ARG;
  if ((train & 1) && (!stats || !save_mean || !save_rstd)) return RSIS_ERR_ARG;
  return rsis_l_bn_fwd(x, res, y, stats, gamma, beta, running_mean, running_var, save_mean, save_rstd, B, C, HW, eps,
                       momentum, relu, train, (hipStream_t)stream);
}
int rsis_bn_bwd(const float* dy, const float* x, const float* y, const float* save_mean, const float* save_rstd,
                const float* gamma, double* stats, float* dx, float* dres, float* dgamma, float* dbeta, int B, int C,
                int HW, int relu, void* stream) {
  if (!dy || !x || !save_mean || !save_rstd || !gamma || !stats || !dx || !dgamma || !dbeta) return RSIS_ERR_ARG;
  if ((relu & 1) && !y) return RSIS_ERR_ARG;
  return rsis_l_bn_bwd(dy, x, y, save_mean, save_rstd, gamma, stats, dx, dres, dgamma, dbeta, B, C, HW, relu, -1.f,
                       (hipStream_t)stream);
}
int rsis_bn_bwd_eval(const float* dy, const float* x, const float* y, const float* running_mean, const float* running_var,
                     const float* gamma, double* stats, float* dx, float* dres, float* dgamma, float* dbeta, int B, int C, int HW,
                     float eps, int relu, void* stream) {
  if (!dy || !x || !running_mean || !running_var || !gamma || !stats || !dx || !dgamma || !dbeta || !(eps >= 0.f)) return RSIS_ERR_ARG;
  if ((relu & 1) && !y) return RSIS_ERR_ARG;
  return rsis_l_bn_bwd(dy, x, y, running_mean, running_var, gamma, stats, dx, dres, dgamma, dbeta, B, C, HW, relu, eps, (hipStream_t)stream);
}
// ---- channel-blocked bf16 activations (conv_blk.hip) ----
int rsis_blk_conv2d(const void* x, int B, int C, int H, int W, const void* Wp, int Cout, int ks, const void* addend, void* out, int variant,
                    void* stream) {
  return rsis_blk_conv2d_bn_eval(x, B, C, H, W, Wp, Cout, ks, addend, nullptr, nullptr, nullptr, nullptr, 0.f, 0, 0, out, variant, stream);
}
int rsis_blk_conv2d_bn_eval(const void* x, int B, int C, int H, int W, const void* Wp, int Cout, int ks, const void* addend, const float* gamma,
                            const float* beta, const float* running_mean, const float* running_var, float eps, int relu, int single_rounding,
                            void* out, int variant, void* stream) {
  if (!x || !Wp || !out || x == out || B < 1 || C < 8 || H < 1 || W < 1 || Cout < 8) return RSIS_ERR_ARG;
  if (gamma && !(beta && running_mean && running_var)) return RSIS_ERR_ARG;
  if (ks != 1 && ks != 3) return RSIS_ERR_UNSUPPORTED;
  ConvArgs a = {};
  a.nsrc = 1; a.src[0] = a.src[1] = a.src[2] = (const float*)x; a.C[0] = C; a.K = C * ks * ks;
  a.B = B; a.H = H; a.W = W; a.Ho = H; a.Wo = W; a.stride = 1; a.pad = ks / 2;
  a.wp = (const float*)Wp; a.ldw = rsis_roundup(Cout, RSIS_LDW_ALIGN); a.Cout = Cout; a.addend = (const float*)addend;
  a.ep_gamma = gamma; a.ep_beta = beta; a.ep_mean = running_mean; a.ep_var = running_var; a.ep_eps = eps;
  a.ep_relu = gamma ? relu : 0; a.ep_round = single_rounding ? 0 : 1;
  a.dst[0] = (float*)out; a.Cd[0] = Cout; a.ndst = 1; a.ostride = 1; a.oH = H; a.oW = W; a.ksplit = 1;
  return rsis_launch_conv_blk(a, ks, variant, (hipStream_t)stream);
}
int rsis_blk_from_nchw(const float* x, void* y, int B, int C, int H, int W, void* stream) {
  if (!x || !y || B < 1 || C < 8 || (C & 7) || H < 1 || W < 1) return RSIS_ERR_ARG;
  return rsis_l_blk_from_nchw(x, y, B, C, H * W, (hipStream_t)stream);
}
int rsis_blk_to_nchw(const void* x, float* y, int B, int C, int H, int W, void* stream) {
  if (!x || !y || B < 1 || C < 8 || (C & 7) || H < 1 || W < 1) return RSIS_ERR_ARG;
  return rsis_l_blk_to_nchw(x, y, B, C, H * W, (hipStream_t)stream);
}

long rsis_blk_bn_scratch_doubles(int C) { return C >= 8 && !(C & 7) ? rsis_l_blk_bn_scratch(C) : 0; }
int rsis_blk_bn_fwd(const void* x, const void* res, void* y, double* scratch, const float* gamma, const float* beta, float* run_mean,
                    float* run_var, float* save_mean, float* save_rstd, int B, int C, int H, int W, float eps, float momentum, int relu,
                    int train, void* stream) {
  if (!x || !y || !gamma || !beta || B < 1 || C < 8 || (C & 7) || H < 1 || W < 1) return RSIS_ERR_ARG;
  if (train ? (!scratch || !save_mean || !save_rstd || (!run_mean != !run_var)) : (!run_mean || !run_var)) return RSIS_ERR_ARG;
  return rsis_l_blk_bn_fwd(x, res, y, scratch, gamma, beta, run_mean, run_var, save_mean, save_rstd, B, C, H * W, eps, momentum, relu ? 1 : 0,
                           train ? 1 : 0, (hipStream_t)stream);
}
int rsis_blk_bn_bwd(const void* dy, const void* x, const void* y, double* scratch, const float* gamma, const float* beta,
                    const float* save_mean, const float* save_rstd, void* dx, void* dres, float* dgamma, float* dbeta, int accumulate, int B,
                    int C, int H, int W, int relu, void* stream) {
  if (!dy || !x || !scratch || !gamma || !beta || !save_mean || !save_rstd || !dx || B < 1 || C < 8 || (C & 7) || H < 1 || W < 1) return RSIS_ERR_ARG;
  return rsis_l_blk_bn_bwd(dy, x, y, scratch, gamma, beta, save_mean, save_rstd, dx, dres, dgamma, dbeta, accumulate ? 1 : 0, B, C, H * W,
                           relu ? 1 : 0, (hipStream_t)stream);
}
int rsis_blk_subsample2d(const void* x, void* y, int B, int C, int H, int W, int stride, void* stream) {
  if (!x || !y || x == y || B < 1 || C < 8 || (C & 7) || H < 1 || W < 1 || stride < 1) return RSIS_ERR_ARG;
  return rsis_l_blk_subsample(x, y, (long)B * (C >> 3), H, W, (H - 1) / stride + 1, (W - 1) / stride + 1, stride, (hipStream_t)stream);
}
int rsis_blk_upscatter2d(const void* dy, void* dx, int B, int C, int H, int W, int stride, void* stream) {
  if (!dy || !dx || dx == dy || B < 1 || C < 8 || (C & 7) || H < 1 || W < 1 || stride < 1) return RSIS_ERR_ARG;
  return rsis_l_blk_upscatter(dy, dx, (long)B * (C >> 3), H, W, (H - 1) / stride + 1, (W - 1) / stride + 1, stride, (hipStream_t)stream);
}

int rsis_subsample2d(const float* x, float* y, long BC, int H, int W, int stride, void* stream) {
  if (!x || !y || x == y || BC < 1 || H < 1 || W < 1 || stride < 1) return RSIS_ERR_ARG;
  return rsis_l_subsample(x, y, BC, H, W, (H - 1) / stride + 1, (W - 1) / stride + 1, stride, (hipStream_t)stream);
}
int rsis_maxpool3x3s2_fwd(const float* x, float* y, unsigned char* argmax, long BC, int H, int W, int Ho, int Wo,
                          void* stream) {
  if (!x || !y || !argmax) return RSIS_ERR_ARG;
  return rsis_l_maxpool_fwd(x, y, argmax, BC, H, W, Ho, Wo, (hipStream_t)stream);
}
int rsis_maxpool3x3s2_bwd(const float* dy, const unsigned char* argmax, float* dx, long BC, int H, int W, int Ho, int Wo,
                          int accumulate, void* stream) {
  if (!dy || !dx || !argmax) return RSIS_ERR_ARG;
  return rsis_l_maxpool_bwd(dy, argmax, dx, BC, H, W, Ho, Wo, accumulate ? 1 : 0, (hipStream_t)stream);
}
int rsis_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, float gscale, const int* step_dev, void* stream) {
  if (!p || !g || !m || !v || n < 0 || (!step_dev && step < 1)) return RSIS_ERR_ARG;
  if (n == 0) return RSIS_OK;
  return rsis_l_adam(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step_dev ? 1 : step, gscale, step_dev, (hipStream_t)stream);
}

int rsis_assign_min_cost(const float* scores, long long* perm, int B, int G, int T, void* stream) {
  if (!scores || !perm || B < 1 || T < 1 || G < T || G > 64) return RSIS_ERR_ARG;
  return rsis_l_assign(scores, perm, B, G, T, (hipStream_t)stream);
}

}  // extern "C"

int rsis_softiou_sums(const float* logits, const float* y, float* S, int B, int T, int G, long N, void* stream) {
  if (!logits || !y || !S || B < 1 || T < 1 || G < 1 || T >= 32 || G >= 32 || N < 8 || N % 8 != 0) return RSIS_ERR_ARG;
  return rsis_l_softiou_sums(logits, y, S, B, T, G, N, (hipStream_t)stream);
}

int rsis_softiou_bwd(const float* logits, const float* y, const long long* perm, int perm_ld, const float* ca, const float* cb,
                     float* dlogits, int B, int T, int G, long N, void* stream) {
  if (!logits || !y || !perm || !ca || !cb || !dlogits || B < 1 || T < 1 || G < 1 || perm_ld < T || N < 4 || N % 4 != 0)
    return RSIS_ERR_ARG;
  return rsis_l_softiou_bwd(logits, y, perm, perm_ld, ca, cb, dlogits, B, T, G, N, (hipStream_t)stream);
}

int rsis_mask_resize_threshold(const float* prob, int n, int Hm, int Wm, const unsigned char* ignore, float th, unsigned char* seg,
                               unsigned char* raw, unsigned int* area, int h, int w, void* stream) {
  if (!prob || !seg || !area || n < 1 || Hm < 1 || Wm < 1 || h < 1 || w < 1) return RSIS_ERR_ARG;
  return rsis_l_mask_resize_threshold(prob, n, Hm, Wm, ignore, th, seg, raw, area, h, w, (hipStream_t)stream);
}

int rsis_rle_encode(const unsigned char* masks, int n, long len, unsigned int* counts, int cap, int* nruns, void* stream) {
  if (!masks || !counts || !nruns || n < 1 || len < 1 || cap < 1 || len >= (1L << 32)) return RSIS_ERR_ARG;
  return rsis_l_rle_encode(masks, n, len, counts, cap, nruns, (hipStream_t)stream);
}

int rsis_rle_to_string(const unsigned int* counts, int m, char* out, int cap) {
  if (!counts || !out || m < 0 || cap < 1) return -1;
  return rsis_l_rle_to_string(counts, m, out, cap);
}

int rsis_heads_fwd(const float* const* side, const int* Cside, int nside, int B, const float* Wc, const float* bc, int ncls,
                   const float* Ws, const float* bs, float* class_probs, float* stop, void* stream) {
  if (!side || !Cside || !Wc || !bc || !Ws || !bs || !class_probs || !stop || B < 1) return RSIS_ERR_ARG;
  return rsis_l_heads_fwd(side, Cside, nside, B, Wc, bc, ncls, Ws, bs, class_probs, stop, (hipStream_t)stream);
}

int rsis_heads_fwd_keys(const unsigned long long* const* keys, float* const* side_out, int* const* arg_out, const int* Cside, int nside,
                        int B, const float* Wc, const float* bc, int ncls, const float* Ws, const float* bs, float* class_probs, float* stop,
                        void* stream) {
  if (!keys || !side_out || !arg_out || !Cside || !Wc || !bc || !Ws || !bs || !class_probs || !stop || B < 1) return RSIS_ERR_ARG;
  for (int i = 0; i < nside; ++i) if (!keys[i] || !side_out[i] || !arg_out[i]) return RSIS_ERR_ARG;
  return rsis_l_heads_fwd((const float* const*)side_out, Cside, nside, B, Wc, bc, ncls, Ws, bs, class_probs, stop, (hipStream_t)stream, keys,
                          side_out, arg_out);
}

int rsis_heads_bwd(const float* const* side, const int* Cside, int nside, int B, const float* Wc, int ncls, const float* Ws,
                   const float* class_probs, const float* dprobs, const float* dstop, float* const* dside, float* dWc, float* dbc,
                   float* dWs, float* dbs, void* stream) {
  if (!side || !Cside || !Wc || !Ws || !class_probs || B < 1) return RSIS_ERR_ARG;
  return rsis_l_heads_bwd(side, Cside, nside, B, Wc, ncls, Ws, class_probs, dprobs, dstop, dside, dWc, dbc, dWs, dbs, (hipStream_t)stream);
}

int rsis_largest_component(const unsigned char* mask, unsigned char* out, int* labels, int* counts, int* best, int n, int h, int w,
                           void* stream) {
  if (!mask || !out || !labels || !counts || !best || n < 1 || h < 1 || w < 1 || (long)h * w >= (1L << 31)) return RSIS_ERR_ARG;
  return rsis_l_largest_component(mask, out, labels, counts, best, n, h, w, (hipStream_t)stream);
}

int rsis_loss_tail(const float* probs, const long long* y_class, const float* stop, const float* siou, const float* sw_mask,
                   const float* sw_class, const float* cls_w, int n, int C, float bw, float w_iou, float w_cls, float w_stop,
                   float* out, float* dprobs, float* dstop, float* dsiou, const float* gout, void* stream) {
  if (!probs || !y_class || !stop || !siou || !sw_mask || !sw_class || n < 1 || C < 1) return RSIS_ERR_ARG;
  if ((dprobs || dstop || dsiou) && !(dprobs && dstop && dsiou)) return RSIS_ERR_ARG;
  if (!out && !dprobs) return RSIS_ERR_ARG;
  return rsis_l_loss_tail(probs, y_class, stop, siou, sw_mask, sw_class, cls_w, n, C, bw, w_iou, w_cls, w_stop, out, dprobs, dstop, dsiou,
                          gout, (hipStream_t)stream);
}


// ---- the recurrent decoder on blk tensors (conv_blk_dec.hip, blk_dec.hip) ----
int rsis_blk_conv3x3_batch(const rsis_blk_conv_job* jobs, int njobs, void* stream) {
  if (!jobs || njobs < 1 || njobs > 64) return RSIS_ERR_ARG;
  BlkConvJob plain[64], lstm[64];
  int fp[64], fl[64], np = 0, nl = 0;
  for (int j = 0; j < njobs; ++j) {
    const rsis_blk_conv_job& q = jobs[j];
    BlkConvJob a = {};
    if (q.nsrc < 0 || q.nsrc > RSIS_MAX_SRC || !q.Wp || q.B < 1 || q.H < 1 || q.W < 1 || q.Cout < 8 || (q.Cout & 7)) return RSIS_ERR_ARG;
    for (int s = 0; s < RSIS_MAX_SRC; ++s) { a.src[s] = q.nsrc > 0 ? q.src[0] : nullptr; a.C[s] = 0; }
    for (int s = 0; s < q.nsrc; ++s) {
      if (!q.src[s] || q.Csrc[s] < 8 || (q.Csrc[s] & 7)) return RSIS_ERR_ARG;
      if ((size_t)(q.Csrc[s] >> 3) * q.H * q.W * 16 >= (1ull << 31)) return RSIS_ERR_UNSUPPORTED;
      a.src[s] = q.src[s]; a.C[s] = q.Csrc[s];
    }
    if ((size_t)(q.Cout >> 3) * q.H * q.W * 16 >= (1ull << 31)) return RSIS_ERR_UNSUPPORTED;
    a.nsrc = q.nsrc; a.B = q.B; a.H = q.H; a.W = q.W; a.wp = q.Wp; a.Cout = q.Cout;
    const int cpack = q.Cpack > 0 ? q.Cpack : q.Cout;
    if (cpack < q.Cout) return RSIS_ERR_ARG;
    a.ldw = rsis_roundup(cpack, RSIS_LDW_ALIGN);
    a.bias = q.bias; a.addend = q.addend;
    if (q.hid > 0) {
      if ((q.hid & 7) || q.Cout != 4 * q.hid || !q.c_out || !q.h_out || (q.nsrc == 0 && !q.addend)) return RSIS_ERR_ARG;
      if ((long)q.H * q.W >= 0x7FFFFFFFL / (4L * q.hid)) return RSIS_ERR_UNSUPPORTED;
      a.hid = q.hid; a.c_prev = q.c_prev; a.c_out = q.c_out; a.h_out = q.h_out; a.act_out = q.act_out; a.side_key = q.side_key;
      a.dst[0] = a.dst[1] = q.h_out;
      fl[nl] = q.tile; lstm[nl++] = a;
    } else {
      if (q.nsrc < 1 || q.ndst < 1 || q.ndst > 2) return RSIS_ERR_ARG;
      int tot = 0;
      for (int d = 0; d < q.ndst; ++d) {
        if (!q.dst[d] || q.Cdst[d] < 8 || (q.Cdst[d] & 7)) return RSIS_ERR_ARG;
        a.dst[d] = q.dst[d]; a.Cd[d] = q.Cdst[d]; tot += q.Cdst[d];
      }
      if (tot != q.Cout) return RSIS_ERR_ARG;
      a.ndst = q.ndst;
      if (q.ndst == 1) { a.dst[1] = a.dst[0]; a.Cd[1] = 0; }
      fp[np] = q.tile; plain[np++] = a;
    }
  }
  int rc = RSIS_OK;
  if (nl) rc = rsis_launch_conv_blk_dec(lstm, nl, 1, fl, (hipStream_t)stream);
  if (rc == RSIS_OK && np) rc = rsis_launch_conv_blk_dec(plain, np, 0, fp, (hipStream_t)stream);
  return rc;
}
static int blk_resize_jobs(const rsis_blk_resize_job* jobs, int njobs, BlkResizeJob* out, bool bwd) {
  if (!jobs || njobs < 1 || njobs > 64) return RSIS_ERR_ARG;
  for (int j = 0; j < njobs; ++j) {
    const rsis_blk_resize_job& q = jobs[j];
    if (!q.src || !q.dst || q.B < 1 || q.C < 8 || (q.C & 7) || q.Hi < 1 || q.Wi < 1 || q.Ho < 1 || q.Wo < 1) return RSIS_ERR_ARG;
    if ((!q.dpool) != (!q.arg) || (!bwd && q.dpool)) return RSIS_ERR_ARG;
    if ((long)q.Hi * q.Wi >= (1L << 27) || (long)q.Ho * q.Wo >= (1L << 27)) return RSIS_ERR_UNSUPPORTED;
    out[j].src = q.src; out[j].dst = q.dst; out[j].dpool = q.dpool; out[j].arg = q.arg;
    out[j].planes = q.B * (q.C >> 3); out[j].Hi = q.Hi; out[j].Wi = q.Wi; out[j].Ho = q.Ho; out[j].Wo = q.Wo;
  }
  return RSIS_OK;
}
int rsis_blk_upsample_fwd_batch(const rsis_blk_resize_job* jobs, int njobs, void* stream) {
  BlkResizeJob a[64];
  const int rc = blk_resize_jobs(jobs, njobs, a, false);
  return rc ? rc : rsis_l_blk_upsample_fwd(a, njobs, (hipStream_t)stream);
}
int rsis_blk_upsample_bwd_batch(const rsis_blk_resize_job* jobs, int njobs, void* stream) {
  BlkResizeJob a[64];
  const int rc = blk_resize_jobs(jobs, njobs, a, true);
  return rc ? rc : rsis_l_blk_upsample_bwd(a, njobs, (hipStream_t)stream);
}
int rsis_blk_lstm_bwd_batch(const rsis_blk_lstm_bwd_job* jobs, int njobs, void* stream) {
  if (!jobs || njobs < 1 || njobs > 64) return RSIS_ERR_ARG;
  BlkLstmBwdJob a[64];
  for (int j = 0; j < njobs; ++j) {
    const rsis_blk_lstm_bwd_job& q = jobs[j];
    if (!q.dh || !q.act || !q.c || !q.da || q.B < 1 || q.hid < 8 || (q.hid & 7) || q.HW < 1) return RSIS_ERR_ARG;
    a[j].dh = q.dh; a[j].dh2 = q.dh2; a[j].dc_next = q.dc_next; a[j].act = q.act; a[j].c_prev = q.c_prev; a[j].c = q.c; a[j].da = q.da;
    a[j].dc_prev = q.dc_prev; a[j].B = q.B; a[j].hid = q.hid; a[j].HW = q.HW;
  }
  return rsis_l_blk_lstm_bwd(a, njobs, (hipStream_t)stream);
}
static inline bool blk_c1_ok(int T, int B, int H, int W) { return T >= 1 && B >= 1 && H >= 1 && W >= 4 && W % 4 == 0 && (long)H * W < (1L << 27); }
int rsis_blk_conv_out_seq_fwd(const void* x, const float* Wref, const float* bias, float* y, int T, int B, int H, int W, void* stream) {
  if (!x || !Wref || !y) return RSIS_ERR_ARG;
  if (!blk_c1_ok(T, B, H, W)) return RSIS_ERR_UNSUPPORTED;
  return rsis_l_blk_c1_fwd(x, Wref, bias, y, T, B, H, W, (hipStream_t)stream);
}
int rsis_blk_conv_out_seq_dgrad(const float* dy, const float* Wref, void* dx, int T, int B, int H, int W, void* stream) {
  if (!dy || !Wref || !dx) return RSIS_ERR_ARG;
  if (!blk_c1_ok(T, B, H, W)) return RSIS_ERR_UNSUPPORTED;
  return rsis_l_blk_c1_dgrad(dy, Wref, dx, T, B, H, W, (hipStream_t)stream);
}
int rsis_blk_conv_out_seq_wgrad(const float* dy, const void* x, float* dW, float* db, int T, int B, int H, int W, void* stream) {
  if (!dy || !x || !dW) return RSIS_ERR_ARG;
  if (!blk_c1_ok(T, B, H, W)) return RSIS_ERR_UNSUPPORTED;
  return rsis_l_blk_c1_wgrad(dy, x, dW, db, T, B, H, W, (hipStream_t)stream);
}
int rsis_blk_sum_leading(const void* x, void* y, int T, long ncells, void* stream) {
  if (!x || !y || T < 1 || ncells < 1) return RSIS_ERR_ARG;
  return rsis_l_blk_sum_leading(x, y, T, ncells, (hipStream_t)stream);
}
int rsis_blk_bias_grad(const void* dy, float* db, int B, int C, int HW, int lstm_hid, void* stream) {
  if (!dy || !db || B < 1 || C < 8 || (C & 7) || HW < 1 || (lstm_hid > 0 && C != 4 * lstm_hid)) return RSIS_ERR_ARG;
  return rsis_l_blk_channel_sum(dy, db, B, C, HW, lstm_hid, (hipStream_t)stream);
}

// ---- grouped launches of the fp32 decoder's backward (the cells of a reverse diagonal of the (level, timestep) wavefront) ----
int rsis_convlstm_bwd_gates_batch(const rsis_lstm_bwd_job* jobs, int njobs, void* stream) {
  if (!jobs || njobs < 1 || njobs > 64) return RSIS_ERR_ARG;
  const void* ptrs[64 * 8];
  int dims[64 * 3];
  for (int j = 0; j < njobs; ++j) {
    const rsis_lstm_bwd_job& q = jobs[j];
    if (!q.act || !q.c || !q.da || q.B < 1 || q.hid < 1 || q.HW < 1) return RSIS_ERR_ARG;
    const void* p8[8] = {q.dh, q.dh2, q.dc_next, q.act, q.c_prev, q.c, q.da, q.dc_prev};
    for (int k = 0; k < 8; ++k) ptrs[j * 8 + k] = p8[k];
    dims[j * 3] = q.B; dims[j * 3 + 1] = q.hid; dims[j * 3 + 2] = q.HW;
  }
  return rsis_l_lstm_bwd_group(ptrs, dims, njobs, (hipStream_t)stream);
}
int rsis_conv2d_dgrad_batch(const rsis_dgrad_job* jobs, int njobs, void* stream) {
  if (!jobs || njobs < 1 || njobs > 64) return RSIS_ERR_ARG;
  ConvArgs grp[64];
  int force[64], ng = 0, rc = RSIS_OK;
  // jobs whose packed copy is a Winograd one (RSIS_DTYPE_F32_WINO: the decoder levels with >= 32-channel sources on >= 16-pixel maps) go
  // into grouped Winograd launches of up to 4 jobs; the rest as before
  const float* wx[4]; const void* wu[4]; float* wy0[4]; float* wy1[4];
  int wb[4], wc[4], wco[4], wc0[4], wh[4], ww[4], nw = 0;
  for (int j = 0; j < njobs && rc == RSIS_OK; ++j) {
    const rsis_dgrad_job& q = jobs[j];
    if (!q.dy || !q.Wd || q.ndst < 1 || q.ndst > RSIS_MAX_SRC) return RSIS_ERR_ARG;
    int ctot = 0;
    for (int i = 0; i < q.ndst; ++i) { if (!q.dx[i] || q.Cdx[i] < 1) return RSIS_ERR_ARG; ctot += q.Cdx[i]; }
    if (q.dtype == RSIS_DTYPE_F32_WINO && wino_geom_dgrad(q.ks, q.stride, q.pad, q.Cin_packed, q.Cout) && !q.addend && q.ndst <= 2 &&
        ctot % 32 == 0 && q.Cdx[0] % 32 == 0 && q.Hx == q.Hy && q.Wx == q.Wy) {
      wx[nw] = q.dy; wu[nw] = q.Wd; wy0[nw] = q.dx[0]; wy1[nw] = q.ndst == 2 ? q.dx[1] : nullptr;
      wb[nw] = q.B; wc[nw] = q.Cout; wco[nw] = ctot; wc0[nw] = q.Cdx[0]; wh[nw] = q.Hy; ww[nw] = q.Wy;
      if (++nw == 4) { rc = rsis_launch_conv_wino_group(nw, wx, wu, wy0, wy1, wb, wc, wco, wc0, wh, ww, (hipStream_t)stream); nw = 0; }
      continue;
    }
    // grouped: the exact-f32 direct kernel (3x3 / stride 1 / pad 1, no addend) with 32-bit epilogue addressing; everything else -- and
    // every job in the deterministic mode, where launches stay as the single-call path issues them -- through rsis_conv2d_dgrad
    const bool groupable = q.dtype == RSIS_DTYPE_F32 && use_direct(q.ks, q.stride, q.pad) && !q.addend && q.Hx == q.Hy && q.Wx == q.Wy &&
                           !(q.Cout == 1) && ctot <= q.Cin_packed && (size_t)q.B * (ctot > q.Cout ? ctot : q.Cout) * q.Hy * q.Wy * 4 < (1ull << 31) &&
                           !rsis_deterministic();
    if (!groupable) {
      rc = rsis_conv2d_dgrad(q.dy, q.B, q.Cout, q.Hy, q.Wy, q.Wd, q.Cin_packed, q.ks, q.stride, q.pad, q.dx, q.Cdx, q.ndst, q.Hx, q.Wx, q.addend, q.tile,
                             q.dtype, stream);
      continue;
    }
    ConvArgs a = {};
    const float* srcs[1] = {q.dy};
    const int cs[1] = {q.Cout};
    rc = fill_sources(a, srcs, cs, 1, 3);
    if (rc) break;
    for (int i = 0; i < q.ndst; ++i) { a.dst[i] = q.dx[i]; a.Cd[i] = q.Cdx[i]; }
    a.ndst = q.ndst;
    a.B = q.B; a.H = q.Hy; a.W = q.Wy; a.Ho = q.Hx; a.Wo = q.Wx; a.stride = 1; a.pad = 1; a.sshift = 0;
    a.ostride = 1; a.oH = q.Hx; a.oW = q.Wx; a.ksplit = 1;
    a.wp = (const float*)q.Wd; a.ldw = rsis_roundup(q.Cin_packed, RSIS_LDW_ALIGN); a.Cout = ctot; a.bias = nullptr; a.addend = nullptr;
    force[ng] = direct_variant(q.tile); grp[ng++] = a;
  }
  if (rc == RSIS_OK && nw > 0) rc = rsis_launch_conv_wino_group(nw, wx, wu, wy0, wy1, wb, wc, wco, wc0, wh, ww, (hipStream_t)stream);
  if (rc == RSIS_OK && ng == 1) rc = rsis_launch_conv3x3_direct(grp[0], 0, force[0], (hipStream_t)stream);
  else if (rc == RSIS_OK && ng > 1) rc = rsis_launch_conv3x3_direct_group_plain(grp, ng, force, (hipStream_t)stream);
  return rc;
}

// ---- upsample x2 (align corners) + conv_out as one op per direction (upconv_out.hip) ----
bool rsis_upconv_supported(int C, int Hs, int Ws, int Ho, int Wo);
int rsis_upconv_bwd_blocks(int T, int B, int Hs, int Ws);
int rsis_l_upconv_fwd(const void* h, int h_blk, const float* w, const float* bias, float* out, int T, int B, int Hs, int Ws, int Ho, int Wo, hipStream_t st);
int rsis_l_upconv_bwd(const float* dout, const void* h, int h_blk, const float* w, void* dh, float* dW, float* db, const float* dside, const int* arg,
                      float* partial, int T, int B, int Hs, int Ws, int Ho, int Wo, hipStream_t st);
extern "C" {
int rsis_upconv_out_supported(int C, int Hs, int Ws, int Ho, int Wo) { return rsis_upconv_supported(C, Hs, Ws, Ho, Wo) ? 1 : 0; }
int rsis_upconv_out_bwd_blocks(int T, int B, int Hs, int Ws) { return rsis_upconv_bwd_blocks(T, B, Hs, Ws); }
int rsis_upconv_out_fwd(const void* h, int h_blk, const float* w, const float* bias, float* out, int T, int B, int C, int Hs, int Ws, int Ho, int Wo,
                        void* stream) {
  if (!h || !w || !out || T < 1 || B < 1) return RSIS_ERR_ARG;
  if (!rsis_upconv_supported(C, Hs, Ws, Ho, Wo)) return RSIS_ERR_UNSUPPORTED;
  return rsis_l_upconv_fwd(h, h_blk, w, bias, out, T, B, Hs, Ws, Ho, Wo, (hipStream_t)stream);
}
int rsis_upconv_out_bwd(const float* dout, const void* h, int h_blk, const float* w, void* dh, float* dW, float* db, const float* dside, const int* arg,
                        float* partial, int T, int B, int C, int Hs, int Ws, int Ho, int Wo, void* stream) {
  if (!dout || !h || !w || !dh || !partial || T < 1 || B < 1 || (dside && !arg)) return RSIS_ERR_ARG;
  if (!rsis_upconv_supported(C, Hs, Ws, Ho, Wo)) return RSIS_ERR_UNSUPPORTED;
  return rsis_l_upconv_bwd(dout, h, h_blk, w, dh, dW, db, dside, arg, partial, T, B, Hs, Ws, Ho, Wo, (hipStream_t)stream);
}
}
