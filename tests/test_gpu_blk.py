"""GPU parity tests of the channel-blocked bf16 kernels (rsis_amd/csrc/conv_blk.hip, blk_norm.hip): the storage half of the bf16
path (BASELINE.json configs[2..4]).  A blk tensor is the logical [B][C][H][W] tensor stored as bf16 [B][C/8][H][W][8].

Semantics under test (stated before measuring): the kernels compute in fp32 on EXACT bf16 inputs and round ONCE to bf16 at the
store.  The reference is therefore the same op in float64 on the same bf16-valued inputs (and bf16-rounded weights for the convs);
a result may differ from it by half a bf16 ulp of the exact value (2^-9 relative) plus the fp32 accumulation noise
(1e-5 of the output scale), and nothing else."""
import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu

HALF_ULP = 2.0 ** -8        # |x - bf16(x)| <= 2^-9 |x|; the bar leaves a factor 2 for values next to a binade boundary


def to_blk(x):
    """fp32 NCHW -> the blk layout, by torch (round-to-nearest-even)"""
    B, C, H, W = x.shape
    return x.view(B, C // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous().to(torch.bfloat16)


def from_blk(y):
    B, Cb, H, W, _ = y.shape
    return y.float().permute(0, 1, 4, 2, 3).reshape(B, Cb * 8, H, W)


def _bf16(x):
    return x.to(torch.bfloat16).float()


def test_layout_converters_round_trip():
    from rsis_amd import ops
    torch.manual_seed(0)
    x = torch.randn(3, 24, 5, 7, device="cuda") * 3
    y = ops.blk_from_nchw(x)
    assert y.dtype == torch.bfloat16 and tuple(y.shape) == (3, 3, 5, 7, 8)
    assert torch.equal(y, to_blk(x))
    assert torch.equal(ops.blk_to_nchw(y), _bf16(x))


SHAPES = [  # (B, Cin, Cout, H, W, ks)
    (2, 64, 64, 56, 56, 3), (2, 256, 256, 14, 14, 3), (3, 512, 512, 7, 7, 3), (2, 128, 128, 28, 28, 3), (2, 32, 48, 10, 13, 3),
    (2, 64, 256, 56, 56, 1), (2, 1024, 256, 14, 14, 1), (3, 2048, 512, 7, 7, 1), (2, 256, 64, 56, 56, 1), (2, 40, 24, 9, 11, 1),
    (2, 16, 8, 5, 5, 3), (1, 8, 8, 3, 3, 1),
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(str(v) for v in s))
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6])
def test_blk_conv_forward_and_data_gradient(shape, variant):
    """rsis_blk_conv2d as the forward conv and, on the data-gradient pack, as conv_transpose (reference: the trunk convs of
    vision.py:12-19 / torchvision bottlenecks, bias-free) -- every tile variant on trunk shapes, ragged maps and channel tails."""
    from rsis_amd import ops
    B, Cin, Cout, H, W, ks = shape
    if ks == 1 and variant == 6:
        pytest.skip("five 1x1 variants")
    torch.manual_seed(sum(shape))
    x = _bf16(torch.randn(B, Cin, H, W, device="cuda"))
    w = torch.randn(Cout, Cin, ks, ks, device="cuda") / (Cin * ks * ks) ** 0.5
    pk = ops.PackedConv(ks, [Cin], stride=1, pad=ks // 2, dtype=ops.DTYPE_BF16)
    y = from_blk(ops.blk_conv2d(to_blk(x), pk.fwd(w), Cout, ks, variant))
    ref = torch.nn.functional.conv2d(x.double(), _bf16(w).double(), padding=ks // 2)
    scale = float(ref.abs().max())
    assert_close("fwd", y, ref, 1e-5 * scale, HALF_ULP)
    dy = _bf16(torch.randn(B, Cout, H, W, device="cuda"))
    dx = from_blk(ops.blk_conv2d(to_blk(dy), pk.dgrad(w), Cin, ks, variant))
    dref = torch.nn.functional.conv_transpose2d(dy.double(), _bf16(w).double(), padding=ks // 2)
    assert_close("dgrad", dx, dref, 1e-5 * float(dref.abs().max()), HALF_ULP)
