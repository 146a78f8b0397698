"""sequence_from_masks (reference src/dataloader/dataset.py:86-146) -- and a device-side equivalent that skips the reference's
float64 [gt_T, H*W + 3] array (10.5 MB per 256x256 sample, split again by utils.batch_to_var) and emits the tensors runIter
reads (y_mask fp32, y_class int64, sw_mask, sw_class) directly on the GPU."""
import numpy as np
import torch


def _order_by_area(sizes):
    """reference: np.argsort(size_masks)[::-1] with numpy's default (unstable) sort: the order of instances of EQUAL area is
    implementation-defined there (it differs between numpy builds).  Here: stable ascending sort, reversed."""
    return np.argsort(np.asarray(sizes), kind="stable")[::-1]


def sequence_from_masks(ins, seg, max_seq_len):
    """ins: (H, W) instance-id map (0 = background), seg: (H, W) class-id map -> float64 (max_seq_len, H*W + 3) target:
    rows = instances sorted by area (largest first), columns = [binary mask | class id | sw_mask | sw_class]; the row after
    the last instance carries sw_class = 1 (the end-of-sequence sample of the stop loss)."""
    ins, seg = np.asarray(ins), np.asarray(seg)
    h, w = ins.shape
    ids = np.unique(ins)[1:]                                  # dataset.py:92 (drops the smallest id = background)
    n = len(ids)
    rows = max(max_seq_len, n)
    gt_classes = np.zeros((rows, 1))
    gt_seg = np.zeros((rows, h * w))
    sizes = np.zeros((rows,))
    sw_mask = np.zeros((rows, 1))
    sw_class = np.zeros((rows, 1))
    for i, k in enumerate(ids):
        m = ins == k
        gt_classes[i] = np.unique(seg[m])[0]                  # :110-113 smallest class id under the instance
        gt_seg[i] = m.reshape(-1)
        sizes[i] = gt_seg[i].sum()
        sw_mask[i] = 1
        sw_class[i] = 1
    order = _order_by_area(sizes)
    gt_classes, gt_seg = gt_classes[order][:max_seq_len], gt_seg[order][:max_seq_len]
    sw_mask, sw_class = sw_mask[order][:max_seq_len], sw_class[order][:max_seq_len]
    if max_seq_len > n:                                       # :133-137 end-of-sequence token
        gt_classes[n:] = 0
        gt_seg[n:, :] = 0
        sw_class[n] = 1
    return np.concatenate((gt_seg, gt_classes, sw_mask, sw_class), axis=1)


def targets_from_maps(ins, seg, max_seq_len, device="cuda"):
    """Same content as utils.batch_to_var(sequence_from_masks(...)) for a batch of maps, built on the device:
    ins, seg: (B, H, W) integer arrays / tensors -> (y_mask (B, T, H*W) fp32, y_class (B, T) int64, sw_mask (B, T) fp32,
    sw_class (B, T) fp32) with T = max_seq_len."""
    ins = torch.as_tensor(np.asarray(ins) if not torch.is_tensor(ins) else ins).to(device).long()
    seg = torch.as_tensor(np.asarray(seg) if not torch.is_tensor(seg) else seg).to(device).long()
    B, H, W = ins.shape
    T = int(max_seq_len)
    y_mask = torch.zeros((B, T, H * W), dtype=torch.float32, device=device)
    y_class = torch.zeros((B, T), dtype=torch.int64, device=device)
    sw_mask = torch.zeros((B, T), dtype=torch.float32, device=device)
    sw_class = torch.zeros((B, T), dtype=torch.float32, device=device)
    big = int(seg.max()) + 1 if seg.numel() else 1
    for b in range(B):
        ids, counts = torch.unique(ins[b], return_counts=True)          # sorted ascending, like np.unique
        ids, counts = ids[1:], counts[1:]                                 # dataset.py:92
        n = int(ids.numel())
        if n:
            # stable ascending sort reversed == the reference's np.argsort(...)[::-1] (ties in reverse original order)
            order = torch.flip(torch.sort(counts, stable=True).indices, dims=[0])[:T]
            sel = ids[order]
            m = ins[b].reshape(1, -1) == sel.reshape(-1, 1)               # (k, HW)
            k = int(sel.numel())
            y_mask[b, :k] = m.float()
            cls = torch.where(m, seg[b].reshape(1, -1), torch.full_like(seg[b].reshape(1, -1), big)).min(dim=1).values
            y_class[b, :k] = cls
            sw_mask[b, :k] = 1
            sw_class[b, :k] = 1
        if T > n:
            sw_class[b, n] = 1
    return y_mask, y_class, sw_mask, sw_class
