"""Shared test helpers (oracle = CPU checker; the product runs on cuda:0 through librsis_hip.so)."""
import argparse
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def mk_args(hidden_size=128, num_classes=21, maxseqlen=10, **kw):
    a = argparse.Namespace(use_gpu=True, base_model="resnet101", hidden_size=hidden_size, kernel_size=3,
                           num_classes=num_classes, dropout=0.0, dropout_stop=0.0, dropout_cls=0.0, skip_mode="concat",
                           maxseqlen=maxseqlen, gt_maxseqlen=20, iou_weight=1.0, class_weight=0.1, stop_weight=0.5,
                           stop_balance_weight=0.5, use_class_loss=True, use_stop_loss=True, curriculum_learning=False,
                           update_encoder=True)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def assert_close(what, got, want, atol, rtol=0.0):
    got = torch.as_tensor(got).detach().double().cpu()
    want = torch.as_tensor(np.asarray(want) if not torch.is_tensor(want) else want).detach().double().cpu()
    assert got.shape == want.shape, "%s: shape %s vs %s" % (what, tuple(got.shape), tuple(want.shape))
    if got.numel() == 0:
        return
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    if os.environ.get("RSIS_TEST_MARGINS"):        # margin survey: worst err / tol of every comparison, one line per call
        with open(os.environ["RSIS_TEST_MARGINS"], "a") as f:
            ratio = float((err / tol.clamp_min(1e-300)).max()) if tol.numel() else 0.0
            f.write("%.4f\t%s\t%s\n" % (ratio, os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], what))
    bad = err > tol
    if bad.any() or torch.isnan(got).any():
        i = int(torch.argmax(err - tol))
        idx = np.unravel_index(i, got.shape) if got.dim() else ()
        raise AssertionError("%s: max abs err %.3e (tol %.1e + %.1e*|ref|) at %s: got %.6g want %.6g; %d/%d bad; |ref|max %.3g"
                             % (what, float(err.max()), atol, rtol, idx, float(got.reshape(-1)[i]), float(want.reshape(-1)[i]),
                                int(bad.sum()), got.numel(), float(want.abs().max())))


def same_matching(what, assignment, class_perm, ref_scores, ref_class_perm, tie=1e-5):
    """The matching of reference train.py:137 is an arg-min over assignments: it is defined only up to ties of the cost matrix.  With
    random weights the predicted masks of an image barely change over the timesteps, so the columns of its cost matrix are almost equal
    and the optimum beats the next assignment by ~1e-6 (BASELINE configs[1] on the synthetic batch: every image) -- less than two fp32
    evaluations of the soft IoU differ by.  So: the product's permuted class targets must EQUAL the reference's (returns True), or
    its assignment must be as good as the optimum UNDER THE REFERENCE'S OWN COSTS to within `tie` for every image (returns False: the
    caller then compares what depends on the assignment against the reference evaluated under that assignment)."""
    from scipy.optimize import linear_sum_assignment
    got = torch.as_tensor(class_perm).cpu().numpy()
    want = np.asarray(ref_class_perm)
    if got.shape == want.shape and (got == want).all():
        return True
    S = np.asarray(ref_scores, dtype=np.float64)
    A = torch.as_tensor(assignment).cpu().numpy()
    T = S.shape[2]
    for b in range(S.shape[0]):
        cols = A[b, :T]
        assert len(set(cols.tolist())) == T, "%s: image %d: not an assignment: %s" % (what, b, cols)
        ri, ci = linear_sum_assignment(S[b])
        best, mine = S[b][ri, ci].sum(), S[b][cols, np.arange(T)].sum()
        assert mine <= best + tie, "%s: image %d: assignment %s costs %.7f under the reference's scores, the optimum %.7f" % (what, b, cols, mine, best)
    return False


def sub_idx(n, cap=4096):
    """the deterministic sub-sample oracle/make_golden.py stores of a flat gradient / parameter vector"""
    return slice(0, n, max(1, n // cap))
