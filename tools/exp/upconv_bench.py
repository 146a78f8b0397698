"""Timing of the fused decoder tail (rsis_upconv_out_fwd / _bwd) at the bench shapes next to its HBM bytes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rsis_amd._lib import check, lib, ptr, stream          # noqa: E402


def t_us(fn, n=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


L = lib()
for (T, B, Hs, blk) in ((10, 32, 112, 1), (10, 32, 128, 0), (20, 8, 256, 1)):
    Ws = Hs if Hs != 256 else 512
    Ho, Wo = 2 * Hs, 2 * Ws
    h = torch.randn(T, B, 1, Hs, Ws, 8, device="cuda").to(torch.bfloat16) if blk else torch.randn(T, B, 8, Hs, Ws, device="cuda")
    w, b = torch.randn(1, 8, 3, 3, device="cuda") / 8, torch.randn(1, device="cuda")
    out, dout = torch.empty(B, T, Ho * Wo, device="cuda"), torch.randn(B, T, Ho * Wo, device="cuda")
    dh, dW, db = torch.empty_like(h), torch.zeros(72, device="cuda"), torch.zeros(1, device="cuda")
    arg = torch.randint(0, Hs * Ws, (T, B, 8), device="cuda", dtype=torch.int32)
    dside = torch.randn(T, B, 8, device="cuda")
    partial = torch.empty(L.rsis_upconv_out_bwd_blocks(T, B, Hs, Ws) * 80, device="cuda")
    f = t_us(lambda: check(L.rsis_upconv_out_fwd(ptr(h), blk, ptr(w), ptr(b), ptr(out), T, B, 8, Hs, Ws, Ho, Wo, stream()), "fwd"))
    g = t_us(lambda: check(L.rsis_upconv_out_bwd(ptr(dout), ptr(h), blk, ptr(w), ptr(dh), ptr(dW), ptr(db), ptr(dside), ptr(arg), ptr(partial),
                                                 T, B, 8, Hs, Ws, Ho, Wo, stream()), "bwd"))
    hb, ob = h.numel() * h.element_size(), out.numel() * 4
    print("T %d B %d %dx%d %s: fwd %.1f us (%.0f MB -> %.2f TB/s)  bwd+finalize %.1f us (%.0f MB -> %.2f TB/s)" % (
        T, B, Hs, Ws, "blk" if blk else "fp32", f, (hb + ob) / 1e6, (hb + ob) / f / 1e6, g, (2 * hb + ob) / 1e6, (2 * hb + ob) / g / 1e6))
