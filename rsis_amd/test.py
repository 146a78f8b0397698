"""Inference forward -- drop-in for reference src/test.py:16-50 (`test(args, encoder, decoder, x)`).

Eval-mode encoder once, decoder T = args.maxseqlen steps, masks resized to the input size, returns
(sigmoid(masks) [B,T,H,W], class probabilities [B,T,C], sigmoid(stop) [B,T,1]).  Runs under torch.no_grad()
(the reference pre-dates it and relied on volatile Variables); `return_logits=True` returns the raw mask / stop logits.
"""
import torch

from . import ops


@torch.no_grad()
def test(args, encoder, decoder, x, return_logits=False):
    T = args.maxseqlen
    hidden = None
    out_masks, out_classes, out_stops = [], [], []
    encoder.eval()
    decoder.eval()
    feats = encoder(x)                                                  # test.py:35
    for _t in range(0, T):
        out_mask, out_class, out_stop, hidden = decoder(feats, hidden)  # test.py:38
        out_mask = ops.upsample_bilinear_ac(out_mask, (x.size()[-2], x.size()[-1]))   # test.py:39-40
        out_masks.append(out_mask)
        out_classes.append(out_class)
        out_stops.append(out_stop)
    out_masks = torch.cat(out_masks, 1)                                 # test.py:46
    out_classes = torch.cat(out_classes, 1).view(out_class.size(0), len(out_classes), -1)   # test.py:47
    out_stops = torch.cat(out_stops, 1).view(out_stop.size(0), len(out_stops), -1)          # test.py:48
    if return_logits:
        return out_masks, out_classes, out_stops
    return torch.sigmoid(out_masks), out_classes, torch.sigmoid(out_stops)                   # test.py:50
