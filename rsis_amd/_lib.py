"""ctypes binding of librsis_hip.so (C ABI in include/rsis_hip.h).

The library is the product: there is NO fallback.  If it has not been built (or cannot be loaded) every op
raises -- build it with `python -c "import __graft_entry__ as g; g.build()"` (hipcc, gfx950).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RSIS_HIP_LIB selects another build of the same C ABI (a replacement .so must export every symbol of include/rsis_hip.h)
LIB_PATH = os.environ.get("RSIS_HIP_LIB") or os.path.join(_HERE, "lib", "librsis_hip.so")

_vp, _i, _l, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
_ip = ctypes.POINTER(ctypes.c_int)
_vpp = ctypes.POINTER(ctypes.c_void_p)

class PackJob(ctypes.Structure):
    """struct rsis_pack_job of include/rsis_hip.h"""
    _fields_ = [("W", ctypes.c_void_p), ("out", ctypes.c_void_p), ("dgrad", ctypes.c_int), ("dtype", ctypes.c_int), ("Cout", ctypes.c_int),
                ("Ctot", ctypes.c_int), ("ks", ctypes.c_int), ("stride", ctypes.c_int), ("pad", ctypes.c_int),
                ("nseg", ctypes.c_int), ("Cseg", ctypes.c_int * 3), ("Coff", ctypes.c_int * 3), ("lstm_hid", ctypes.c_int),
                ("imode", ctypes.c_int), ("ldw", ctypes.c_int), ("krows", ctypes.c_int), ("block_begin", ctypes.c_int)]


class WgradJob(ctypes.Structure):
    """struct rsis_wgrad_job of include/rsis_hip.h"""
    _fields_ = [("dy", ctypes.c_void_p), ("x", ctypes.c_void_p), ("dW", ctypes.c_void_p)] + \
               [(k, ctypes.c_int) for k in ("B", "Cs", "H", "W", "Cout", "Ho", "Wo", "ks", "stride", "pad", "Ctot", "c_off", "lstm_hid", "dtype")]


class LstmJob(ctypes.Structure):
    """struct rsis_lstm_job of include/rsis_hip.h"""
    _fields_ = [("src", ctypes.c_void_p * 3), ("Csrc", ctypes.c_int * 3), ("nsrc", ctypes.c_int), ("B", ctypes.c_int), ("H", ctypes.c_int),
                ("W", ctypes.c_int), ("Wp", ctypes.c_void_p), ("bias_packed", ctypes.c_void_p), ("addend", ctypes.c_void_p),
                ("c_prev", ctypes.c_void_p), ("h_out", ctypes.c_void_p), ("c_out", ctypes.c_void_p), ("act_out", ctypes.c_void_p),
                ("hid", ctypes.c_int), ("ks", ctypes.c_int), ("pad", ctypes.c_int), ("tile", ctypes.c_int), ("dtype", ctypes.c_int),
                ("side_key", ctypes.c_void_p)]


class LstmBwdJob(ctypes.Structure):
    """struct rsis_lstm_bwd_job of include/rsis_hip.h"""
    _fields_ = [(k, ctypes.c_void_p) for k in ("dh", "dh2", "dc_next", "act", "c_prev", "c", "da", "dc_prev")] + \
               [(k, ctypes.c_int) for k in ("B", "hid", "HW")]


class DgradJob(ctypes.Structure):
    """struct rsis_dgrad_job of include/rsis_hip.h"""
    _fields_ = [("dy", ctypes.c_void_p), ("B", ctypes.c_int), ("Cout", ctypes.c_int), ("Hy", ctypes.c_int), ("Wy", ctypes.c_int),
                ("Wd", ctypes.c_void_p), ("Cin_packed", ctypes.c_int), ("ks", ctypes.c_int), ("stride", ctypes.c_int), ("pad", ctypes.c_int),
                ("dx", ctypes.c_void_p * 3), ("Cdx", ctypes.c_int * 3), ("ndst", ctypes.c_int), ("Hx", ctypes.c_int), ("Wx", ctypes.c_int),
                ("addend", ctypes.c_void_p), ("tile", ctypes.c_int), ("dtype", ctypes.c_int)]


class BlkConvJob(ctypes.Structure):
    """struct rsis_blk_conv_job of include/rsis_hip.h"""
    _fields_ = [("src", ctypes.c_void_p * 3), ("Csrc", ctypes.c_int * 3), ("nsrc", ctypes.c_int), ("B", ctypes.c_int), ("H", ctypes.c_int),
                ("W", ctypes.c_int), ("Wp", ctypes.c_void_p), ("Cout", ctypes.c_int), ("Cpack", ctypes.c_int), ("bias", ctypes.c_void_p),
                ("addend", ctypes.c_void_p), ("dst", ctypes.c_void_p * 2), ("Cdst", ctypes.c_int * 2), ("ndst", ctypes.c_int),
                ("hid", ctypes.c_int), ("c_prev", ctypes.c_void_p), ("c_out", ctypes.c_void_p), ("h_out", ctypes.c_void_p),
                ("act_out", ctypes.c_void_p), ("side_key", ctypes.c_void_p), ("tile", ctypes.c_int)]


class BlkResizeJob(ctypes.Structure):
    """struct rsis_blk_resize_job of include/rsis_hip.h"""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("dpool", ctypes.c_void_p), ("arg", ctypes.c_void_p)] + \
               [(k, ctypes.c_int) for k in ("B", "C", "Hi", "Wi", "Ho", "Wo")]


class BlkLstmBwdJob(ctypes.Structure):
    """struct rsis_blk_lstm_bwd_job of include/rsis_hip.h"""
    _fields_ = [(k, ctypes.c_void_p) for k in ("dh", "dh2", "dc_next", "act", "c_prev", "c", "da", "dc_prev")] + \
               [(k, ctypes.c_int) for k in ("B", "hid", "HW")]


# name -> (restype, argtypes); must list every symbol declared in include/rsis_hip.h
SIGNATURES = {
    "rsis_version": (_i, []),
    "rsis_error_string": (ctypes.c_char_p, [_i]),
    "rsis_set_deterministic": (_i, [_i]),
    "rsis_get_deterministic": (_i, []),
    "rsis_conv_uses_bf16": (_i, [_i, _i, _i, _i]),
    "rsis_conv_uses_wino": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "rsis_conv_packed_bytes_fwd": (_l, [_i, _i, _i, _i, _i, _i, _ip]),
    "rsis_conv_packed_bytes_dgrad": (_l, [_i, _i, _i, _i, _i, _i]),
    "rsis_conv_pack_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _ip, _ip, _i, _i, _vp]),
    "rsis_conv_pack_dgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _ip, _ip, _i, _i, _vp]),
    "rsis_conv_pack_job_fill": (_i, [ctypes.POINTER(PackJob)]),
    "rsis_conv_pack_batch": (_i, [_vp, _i, _i, _vp]),
    "rsis_conv2d_fwd": (_i, [_vpp, _ip, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rsis_conv2d_fwd_bn_eval": (_i, [_vpp, _ip, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _i, _i, _i, _i, _vp]),
    "rsis_conv2d_dgrad": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vpp, _ip, _i, _i, _i, _vp, _i, _i, _vp]),
    "rsis_conv2d_wgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "rsis_conv2d_wgrad_batch": (_i, [ctypes.POINTER(WgradJob), _i, _vp]),
    "rsis_affine_nearest": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rsis_conv_out_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rsis_conv_out_seq_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rsis_conv_out_seq_dgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rsis_conv_out_seq_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rsis_sum_leading": (_i, [_vp, _vp, _i, _l, _vp]),
    "rsis_bias_grad": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "rsis_convlstm_fwd": (_i, [_vpp, _ip, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rsis_convlstm_fwd_batch": (_i, [ctypes.POINTER(LstmJob), _i, _vp]),
    "rsis_convlstm_bwd_gates": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "rsis_upsample_bilinear_ac_fwd": (_i, [_vp, _vp, _l, _i, _i, _i, _i, _vp]),
    "rsis_upsample_bilinear_ac_bwd": (_i, [_vp, _vp, _l, _i, _i, _i, _i, _vp]),
    "rsis_upsample_maxpool_bwd": (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _vp]),
    "rsis_global_maxpool_fwd": (_i, [_vp, _vp, _vp, _l, _i, _vp]),
    "rsis_global_maxpool_bwd": (_i, [_vp, _vp, _vp, _l, _i, _vp]),
    "rsis_global_maxpool_bwd_add": (_i, [_vp, _vp, _vp, _l, _i, _vp]),
    "rsis_bn_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _i, _vp]),
    "rsis_bn_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rsis_bn_bwd_eval": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "rsis_subsample2d": (_i, [_vp, _vp, _l, _i, _i, _i, _vp]),
    "rsis_maxpool3x3s2_fwd": (_i, [_vp, _vp, _vp, _l, _i, _i, _i, _i, _vp]),
    "rsis_maxpool3x3s2_bwd": (_i, [_vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _vp]),
    "rsis_assign_min_cost": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "rsis_heads_fwd": (_i, [_vpp, _ip, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "rsis_heads_fwd_keys": (_i, [_vpp, _vpp, _vpp, _ip, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "rsis_blk_from_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "rsis_blk_to_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "rsis_blk_bn_scratch_doubles": (ctypes.c_long, [_i]),
    "rsis_blk_bn_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _i, _i, _vp]),
    "rsis_blk_bn_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "rsis_blk_subsample2d": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rsis_blk_upscatter2d": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rsis_blk_conv2d": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i, _vp]),
    "rsis_blk_conv2d_bn_eval": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _i, _i, _vp, _i, _vp]),
    "rsis_convlstm_bwd_gates_batch": (_i, [ctypes.POINTER(LstmBwdJob), _i, _vp]),
    "rsis_conv2d_dgrad_batch": (_i, [ctypes.POINTER(DgradJob), _i, _vp]),
    "rsis_blk_conv3x3_batch": (_i, [ctypes.POINTER(BlkConvJob), _i, _vp]),
    "rsis_blk_upsample_fwd_batch": (_i, [ctypes.POINTER(BlkResizeJob), _i, _vp]),
    "rsis_blk_upsample_bwd_batch": (_i, [ctypes.POINTER(BlkResizeJob), _i, _vp]),
    "rsis_blk_lstm_bwd_batch": (_i, [ctypes.POINTER(BlkLstmBwdJob), _i, _vp]),
    "rsis_blk_sum_leading": (_i, [_vp, _vp, _i, _l, _vp]),
    "rsis_blk_bias_grad": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "rsis_blk_conv_out_seq_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rsis_blk_conv_out_seq_dgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rsis_blk_conv_out_seq_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rsis_upconv_out_supported": (_i, [_i, _i, _i, _i, _i]),
    "rsis_upconv_out_bwd_blocks": (_i, [_i, _i, _i, _i]),
    "rsis_upconv_out_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "rsis_upconv_out_bwd": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "rsis_comm_available": (_i, []),
    "rsis_comm_unique_id": (_i, [_vp]),
    "rsis_comm_init": (_i, [_vpp, _i, _i, _vp]),
    "rsis_comm_size": (_i, [_vp]),
    "rsis_comm_allreduce_sum_f32": (_i, [_vp, _vp, _l, _vp]),
    "rsis_comm_destroy": (_i, [_vp]),
    "rsis_comm_last_error": (ctypes.c_char_p, []),
    "rsis_heads_bwd": (_i, [_vpp, _ip, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vpp, _vp, _vp, _vp, _vp, _vp]),
    "rsis_loss_tail": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rsis_softiou_sums": (_i, [_vp, _vp, _vp, _i, _i, _i, _l, _vp]),
    "rsis_softiou_bwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _l, _vp]),
    "rsis_mask_resize_threshold": (_i, [_vp, _i, _i, _i, _vp, _f, _vp, _vp, _vp, _i, _i, _vp]),
    "rsis_rle_encode": (_i, [_vp, _i, _l, _vp, _i, _vp, _vp]),
    "rsis_largest_component": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "rsis_rle_to_string": (_i, [_vp, _i, ctypes.c_char_p, _i]),
    "rsis_adam_step": (_i, [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _i, _f, _vp, _vp]),
}

_LIB = None


class RsisHipError(RuntimeError):
    pass


def lib():
    """Load librsis_hip.so once; raise loudly if it is missing (no CPU / eager fallback exists)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RsisHipError("%s is missing: the HIP library is the product path and has no fallback; build it with "
                               "`python -c \"import __graft_entry__ as g; g.build()\"`" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc, what):
    if rc != 0:
        raise RsisHipError("%s failed: %s (code %d)" % (what, lib().rsis_error_string(rc).decode(), rc))


def ptr(t):
    """Device pointer of a contiguous fp32/int CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def int_array(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def require_cuda_f32(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RsisHipError("rsis_amd ops run on the GPU only (got a %s tensor); there is no CPU path" % t.device)
        if t.dtype != torch.float32:
            raise RsisHipError("rsis_amd ops are fp32 (got %s)" % t.dtype)
