"""Affine augmentation of a training sample on the device.

Mirrors the reference's `RandomAffine(rotation_range, translation_range, shear_range, zoom_range, interp='nearest')`
(src/dataloader/transforms/transforms.py:23-103; instantiated in dataloader/pascal.py:47-51, cityscapes.py:40-49, leaves.py:37-46)
and `th_affine2d` (transforms/utils.py:67-128): the 3x3 float32 matrix is composed on the host exactly as the reference does
(rotation @ translation @ shear @ zoom, the draws taken from python's `random` in that order), the warp itself -- a gather over
every pixel of the image, the instance map and the class map -- runs in librsis_hip.so (rsis_affine_nearest).  Only the
'nearest' interpolation the reference's datasets use is provided."""
import math
import random

import torch

from .._lib import check, lib, ptr, stream


def rotation_matrix(degrees):                      # transforms.py:319-322
    t = math.pi / 180 * degrees
    return torch.tensor([[math.cos(t), -math.sin(t), 0], [math.sin(t), math.cos(t), 0], [0, 0, 1]], dtype=torch.float32)


def translation_matrix(frac_h, frac_w, H, W):      # transforms.py:484-489 (fractions of the height / width)
    return torch.tensor([[1, 0, frac_h * H], [0, 1, frac_w * W], [0, 0, 1]], dtype=torch.float32)


def shear_matrix(degrees):                         # transforms.py:608-611
    t = (math.pi * degrees) / 180
    return torch.tensor([[1, -math.sin(t), 0], [0, math.cos(t), 0], [0, 0, 1]], dtype=torch.float32)


def zoom_matrix(zx, zy):                           # transforms.py:760-762
    return torch.tensor([[zx, 0, 0], [0, zy, 0], [0, 0, 1]], dtype=torch.float32)


def affine_nearest(x, matrix):
    """x: (C, H, W) or (N, C, H, W) float32 CUDA tensor; matrix: (3, 3) / (2, 3), or (N, 3, 3) / (N, 2, 3) (one per sample).
    Returns the warped tensor (th_affine2d(x, matrix, mode='nearest', center=True) per sample)."""
    if not (x.is_cuda and x.dtype == torch.float32):
        raise RuntimeError("affine_nearest runs in librsis_hip.so: a float32 CUDA tensor is required (there is no CPU path)")
    squeeze = x.dim() == 3
    xb = (x.unsqueeze(0) if squeeze else x).contiguous()
    N, C, H, W = xb.shape
    m = matrix.to(dtype=torch.float32)
    if m.dim() == 2:
        m = m.unsqueeze(0).expand(N, -1, -1)
    if m.shape[0] != N or m.shape[1] not in (2, 3) or m.shape[2] != 3:
        raise ValueError("matrix must be (3,3), (2,3) or one such matrix per sample")
    m = m.contiguous().to(xb.device)
    y = torch.empty_like(xb)
    check(lib().rsis_affine_nearest(ptr(xb), ptr(y), ptr(m), int(m.shape[1]), N, C, H, W, stream()), "rsis_affine_nearest")
    return y.squeeze(0) if squeeze else y


class RandomAffine(object):
    """Same constructor arguments and draw order as the reference's RandomAffine.  `matrix(H, W)` draws one transform;
    calling the object warps every input (tensors of one sample, (C, H, W) each) with the same freshly drawn matrix."""

    def __init__(self, rotation_range=None, translation_range=None, shear_range=None, zoom_range=None, interp="nearest"):
        if isinstance(translation_range, float):
            translation_range = (translation_range, translation_range)
        if interp != "nearest":
            raise NotImplementedError("the reference's datasets use interp='nearest'; bilinear is not provided")
        if rotation_range is None and translation_range is None and shear_range is None and zoom_range is None:
            raise Exception("Must give at least one transform parameter")          # transforms.py:88-89
        self.rotation_range, self.translation_range = rotation_range, translation_range
        self.shear_range, self.zoom_range = shear_range, zoom_range
        self.tform_matrix = None

    def matrix(self, H, W):
        ms = []
        if self.rotation_range is not None:
            ms.append(rotation_matrix(random.uniform(-self.rotation_range, self.rotation_range)))
        if self.translation_range is not None:
            fh = random.uniform(-self.translation_range[0], self.translation_range[0])
            fw = random.uniform(-self.translation_range[1], self.translation_range[1])
            ms.append(translation_matrix(fh, fw, H, W))
        if self.shear_range is not None:
            ms.append(shear_matrix(random.uniform(-self.shear_range, self.shear_range)))
        if self.zoom_range is not None:
            zx = random.uniform(self.zoom_range[0], self.zoom_range[1])
            zy = random.uniform(self.zoom_range[0], self.zoom_range[1])
            ms.append(zoom_matrix(zx, zy))
        m = ms[0]
        for t in ms[1:]:
            m = m.mm(t)                                                           # transforms.py:93-95
        self.tform_matrix = m
        return m

    def __call__(self, *inputs):
        m = self.matrix(inputs[0].size(1), inputs[0].size(2))
        outs = [affine_nearest(x, m) for x in inputs]
        return outs if len(outs) > 2 else outs[0]       # transforms.py:142 (`idx > 1`: a list only for three or more inputs)
