"""TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/_ref/libmaskapi_ref.so = the reference's own src/coco/common/maskApi.c
compiled unmodified by oracle/Makefile ("reference"-kind oracle).  Gives the tests the reference's rleEncode / rleToString /
rleArea (what pycocotools' mask.encode / mask.area return in reference src/eval.py:97-127).  Never imported by the product."""
import ctypes
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libmaskapi_ref.so")


class _RLE(ctypes.Structure):
    _fields_ = [("h", ctypes.c_ulong), ("w", ctypes.c_ulong), ("m", ctypes.c_ulong), ("cnts", ctypes.POINTER(ctypes.c_uint))]


def available():
    return os.path.exists(_PATH)


def _lib():
    L = ctypes.CDLL(_PATH)
    L.rleEncode.argtypes = [ctypes.POINTER(_RLE), ctypes.c_void_p, ctypes.c_ulong, ctypes.c_ulong, ctypes.c_ulong]
    L.rleEncode.restype = None
    L.rleToString.argtypes = [ctypes.POINTER(_RLE)]
    L.rleToString.restype = ctypes.c_void_p
    L.rleFree.argtypes = [ctypes.POINTER(_RLE)]
    L.rleFree.restype = None
    return L


def encode(mask_hw):
    """mask_hw: (h, w) uint8 -> (counts uint32 array, compressed bytes) exactly as pycocotools.mask.encode(asfortranarray(m))"""
    L = _lib()
    m = np.asfortranarray(mask_hw.astype(np.uint8))
    h, w = m.shape
    R = _RLE()
    L.rleEncode(ctypes.byref(R), m.ctypes.data_as(ctypes.c_void_p), h, w, 1)
    counts = np.ctypeslib.as_array(R.cnts, shape=(R.m,)).copy()
    sp = L.rleToString(ctypes.byref(R))
    s = ctypes.string_at(sp)
    ctypes.CDLL(None).free(ctypes.c_void_p(sp))
    L.rleFree(ctypes.byref(R))
    return counts.astype(np.uint32), s
