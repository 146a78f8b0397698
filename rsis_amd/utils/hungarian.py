"""Loss / matching math -- same names, argument meaning and results as reference src/utils/hungarian.py, on device
tensors.  The assignment itself (munkres.Munkres().compute in the reference: pure-Python O(n^3) per sample) is
scipy.optimize.linear_sum_assignment on ONE device->host copy of the whole (B, gt_T, T) score tensor."""
import numpy as np
import torch


def MaskedNLL(target, probs, balance_weights=None):
    """hungarian.py:10-32: -log(probs)[target]; no epsilon (a zero probability gives inf, as in the reference)."""
    log_probs = torch.log(probs)
    if balance_weights is not None:
        log_probs = torch.mul(log_probs, balance_weights.to(log_probs.device))
    losses = -torch.gather(log_probs, dim=1, index=target)
    return losses.squeeze()


def StableBalancedMaskedBCE(target, out, balance_weight=None):
    """hungarian.py:34-59: numerically stable BCE-with-logits, positives weighted (1-bw), negatives bw."""
    if balance_weight is None:
        num_positive = target.sum()
        num_negative = (1 - target).sum()
        total = num_positive + num_negative
        balance_weight = num_positive / total
    max_val = (-out).clamp(min=0)
    loss_values = out - out * target + max_val + ((-max_val).exp() + (-out - max_val).exp()).log()
    loss_positive = loss_values * target
    loss_negative = loss_values * (1 - target)
    losses = (1 - balance_weight) * loss_positive + balance_weight * loss_negative
    return losses.squeeze()


def softIoU(target, out, e=1e-6):
    """hungarian.py:62-89: cost = 1 - sum(p*y) / (sum(p + y - p*y) + e) per row, p = sigmoid(out)."""
    out = torch.sigmoid(out)
    num = (out * target).sum(1, True)
    den = (out + target - out * target).sum(1, True) + e
    iou = num / den
    cost = (1 - iou)
    return cost.squeeze()


def softIoU_matrix(y_mask, out_masks, e=1e-6):
    """All-pairs soft-IoU cost in one batched contraction: cost[b, g, t] = softIoU(y_mask[b, g], out_masks[b, t]).
    Equals the reference's per-timestep `repeat` + softIoU of train.py:102-109 (same sums, one GEMM instead of a
    gt_T-fold copy of every prediction)."""
    p = torch.sigmoid(out_masks)                                    # (B, T, N)
    inter = torch.bmm(y_mask, p.transpose(1, 2))                    # (B, G, T)
    den = y_mask.sum(2, keepdim=True) + p.sum(2).unsqueeze(1) - inter + e
    return 1 - inter / den


def assignment(cost):
    """Munkres().compute(cost): list of (row, col) of a minimum-cost assignment (rectangular allowed)."""
    from scipy.optimize import linear_sum_assignment
    r, c = linear_sum_assignment(np.asarray(cost, dtype=np.float64))
    return list(zip(r.tolist(), c.tolist()))


def match_indices(overlaps):
    """Permutation indices of hungarian.py:91-125 for a (B, gt_T, T) cost tensor: perm[b, col] = row; columns that
    are never assigned keep index 0."""
    ov = overlaps.detach().cpu().numpy()
    B, G, _T = ov.shape
    perm = np.zeros((B, G), dtype=np.int64)
    for b in range(B):
        for row, col in assignment(ov[b]):
            perm[b, col] = row
    return perm


def match(masks, classes, overlaps):
    """hungarian.py:91-125.  Returns (t_mask_perm [B,gt_T,N], t_class_perm [B,gt_T], permute_indices) -- as device
    tensors gathered on the GPU (the reference round-trips every GT mask through the host)."""
    t_mask, _p_mask = masks
    t_class, _p_class = classes
    perm = match_indices(overlaps)
    idx = torch.from_numpy(perm).to(t_mask.device)
    t_mask_perm = torch.gather(t_mask, 1, idx.unsqueeze(-1).expand(-1, -1, t_mask.size(2)))
    t_class_perm = torch.gather(t_class, 1, idx)
    return t_mask_perm, t_class_perm, perm
