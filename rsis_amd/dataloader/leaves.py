"""CVPPP A1 leaves data layer -- the reference's src/dataloader/leaves.py:9-113 + dataset.py:47-84 + dataset_utils.py:28-58, wired to
the device-side target construction and affine augmentation (BASELINE configs[0]).

Split of the work:
  host   : file list and train / val split (leaves.py:64-94), PNG decode and the bilinear resize of the image (PIL, as the
           reference: dataset.py:49-54), nearest resize of the instance map to the image size (scipy zoom order 0:
           dataset_utils.py:133-140), horizontal flip and random crop (index operations, dataset_utils.py:45-58), the draw of the
           affine matrix (python `random`, same draw order as the reference) -- into PINNED staging buffers, decoded by a small
           thread pool one batch ahead;
  device : uint8 -> float, ImageNet normalisation (train.py:34-37), the affine warp of image + instance map (rsis_affine_nearest,
           bit-equal to the reference's th_affine2d(mode='nearest')), and the target tensors runIter reads
           (dataloader.targets_from_maps == sequence_from_masks + batch_to_var).
Deviation: the file list is sorted (the reference takes glob's order, which is file-system dependent)."""
import glob
import os
import random
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .augment import RandomAffine, affine_nearest
from .targets import targets_from_maps

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


class LeavesDataset(object):
    """reference LeavesDataset(MyDataset): same constructor arguments, `classes`, `get_raw_sample`, `__len__`, `get_sample_list`."""

    def __init__(self, args, transform=None, target_transform=None, augment=False, split="train", resize=False, imsize=256):
        self.split = split
        self.classes = ["<eos>", "leaf"]                                   # leaves.py:20
        self.num_classes = len(self.classes)
        self.max_seq_len = args.gt_maxseqlen
        self.batch_size = args.batch_size
        self.crop = self.batch_size != 1                                   # :31-34
        self.flip = augment
        self.augment, self.resize, self.imsize, self.zoom = augment, resize, imsize, args.zoom
        if augment:                                                         # :36-47
            self.augmentation_transform = RandomAffine(rotation_range=args.rotation, translation_range=args.translation,
                                                       shear_range=args.shear, zoom_range=(args.zoom, 1) if resize else None,
                                                       interp="nearest")
        else:
            self.augmentation_transform = None
        total = sorted(glob.glob(os.path.join(args.leaves_dir, "*_rgb.png")))
        gts = [w.replace("_rgb", "_label") for w in total]
        if split == "train":                                                # :72-94: the first 96 images train, the rest validate
            self.image_files, self.gt_files = total[:96], gts[:96]
        elif split == "val":
            self.image_files, self.gt_files = total[96:], gts[96:]
        else:
            self.image_files = sorted(glob.glob(os.path.join(args.leaves_test_dir, "*_rgb.png")))
            self.gt_files = []
        self._cache, self._cache_bytes = {}, 0
        self._cache_limit = int(float(os.environ.get("RSIS_LOADER_CACHE_MB", "1024")) * (1 << 20))

    def get_classes(self):
        return self.classes

    def get_sample_list(self):
        return self.image_files

    def __len__(self):
        return len(self.image_files)

    def get_raw_sample(self, index):
        """(PIL RGB image, instance-id map, class map) in raw size -- leaves.py:96-113"""
        from PIL import Image
        img = Image.open(self.image_files[index]).convert("RGB")
        if self.split != "test":
            gt = np.array(Image.open(self.gt_files[index]))
            ins = gt.copy()
            seg = gt.copy()
            seg[seg > 0] = 1
            return img, ins, seg
        fake = np.array(img)[:, :, 0]
        return img, fake, fake

    def _decoded(self, index):
        """the deterministic part of host_item -- PNG decode, the bilinear resize of the image, the nearest zoom of the instance map -- kept
        after its first evaluation: the reference re-decodes every file in every epoch (dataset.py:47-54), which at CVPPP's 128 images
        is 7-15 ms of host time per image for the same bytes, and made `train.py` on this loader host-bound (NOTES (54): 62 images/s at
        batch 2).  Bounded by RSIS_LOADER_CACHE_MB (default 1024; 0 = off): A1 at 256 x 256 is 58 MB.  The random part (flip, crop) works
        on views and copies, never on the cached arrays."""
        hit = self._cache.get(index)
        if hit is not None:
            return hit
        item = self._decode(index)
        nbytes = item[0].nbytes + item[1].nbytes
        if self._cache_bytes + nbytes <= self._cache_limit:
            self._cache[index] = item
            self._cache_bytes += nbytes
        return item

    def _decode(self, index):
        from PIL import Image
        from scipy.ndimage import zoom
        img, ins, _seg = self.get_raw_sample(index)
        S = self.imsize
        if self.resize:
            img = img.resize((S, S), Image.BILINEAR)                        # transforms.Scale((S, S))
        else:                                                               # transforms.Scale(S): shorter side -> S
            w, h = img.size
            if w <= h:
                img = img.resize((S, max(S, int(S * h / w))), Image.BILINEAR)
            else:
                img = img.resize((max(S, int(S * w / h)), S), Image.BILINEAR)
        im = np.asarray(img, dtype=np.uint8).transpose(2, 0, 1)             # (3, h, w)
        h, w = im.shape[1:]
        ins = zoom(ins, [float(h) / ins.shape[0], float(w) / ins.shape[1]], mode="nearest", order=0)   # dataset_utils.py:133-140
        return np.ascontiguousarray(im), np.ascontiguousarray(ins)

    def host_item(self, index, rng):
        """everything of dataset.py:47-62 that is decoding / index arithmetic: -> (uint8 image (3, S, S), int32 instance map (S, S))"""
        im, ins = self._decoded(index)
        S = self.imsize
        h, w = im.shape[1:]
        if self.flip and rng.random() < 0.5:                                # dataset_utils.py:51-55
            im, ins = im[:, :, ::-1], ins[:, ::-1]
        if self.crop:                                                       # transforms.py:15-21 random_crop (centred range)
            rw, rh = (w - S) // 2, (h - S) // 2
            ow = 0 if rw <= 0 else rng.randrange(rw)
            oh = 0 if rh <= 0 else rng.randrange(rh)
            im, ins = im[:, oh:oh + S, ow:ow + S], ins[oh:oh + S, ow:ow + S]
        return np.ascontiguousarray(im), np.ascontiguousarray(ins.astype(np.int32))


def shard_batches(order, batch_size, rank=0, world=1, drop_last=True):
    """Per-rank index lists of one epoch: the (already shuffled, identical on every rank) sample order is cut into GLOBAL batches of
    batch_size * world (the tail that does not fill one is dropped: drop_last=True, train.py:46-49) and rank r takes elements
    r, r + world, ... of each -- the ranks' shards of a step are disjoint and together are the reference's global batch."""
    gb = int(batch_size) * int(world)
    out = [order[i * gb:(i + 1) * gb][rank::world] for i in range(len(order) // gb)]
    if not drop_last and len(order) % gb and int(world) == 1:       # evaluation (eval*.py: drop_last=False): the short last batch
        out.append(order[len(order) // gb * gb:])
    return out


class DeviceLoader(object):
    """DataLoader(dataset, batch_size, shuffle=True, drop_last=True) of train.py:46-49 yielding DEVICE batches
    (x, y_mask, y_class, sw_mask, sw_class) -- what utils.batch_to_var returns.  The next batch is decoded into pinned memory by
    `num_workers` threads and copied on a side stream while the current one trains.

    One process per GPU (rank / world): `batch_size` is the PER-RANK batch; every rank shuffles the whole dataset with the SAME
    seed and takes samples r, r + world, ... of each global batch of batch_size * world, so that an epoch is
    len(dataset) // (batch_size * world) steps of the reference's global batch (train.py:46-49 under nn.DataParallel) and no
    sample is seen twice in a step; only the per-sample augmentation draws are rank-specific."""

    def __init__(self, dataset, batch_size, shuffle=True, num_workers=4, seed=0, device="cuda", rank=0, world=1, drop_last=True):
        self.drop_last = bool(drop_last)
        if dataset.crop is False and batch_size != 1:
            raise ValueError("un-cropped samples have different sizes: batch_size must be 1")
        self.ds, self.bs, self.shuffle, self.device = dataset, int(batch_size), shuffle, device
        self.rank, self.world = int(rank), int(world)
        self.order_rng = random.Random(seed)                                # the same on every rank
        self.rng = random.Random(seed * 1000003 + 17 * self.rank + 1)       # per-sample draws (crop / flip / warp)
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(num_workers)))
        self.copy_stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        self.mean = torch.tensor(MEAN, device=device).view(1, 3, 1, 1)
        self.std = torch.tensor(STD, device=device).view(1, 3, 1, 1)
        self._lock = threading.Lock()
        self._pins = {}

    def __len__(self):
        n, gb = len(self.ds), self.bs * self.world                          # drop_last=True on the GLOBAL batch (training)
        return n // gb + (1 if (not self.drop_last and self.world == 1 and n % gb) else 0)

    def steps_to_run(self, args, sw_mask):
        """early-stop rule of train.py:80-92 for this batch (one host sync; the batch is fresh, so nothing can be cached)"""
        from ..train import steps_to_run
        if self.copy_stream is None:
            return steps_to_run(args, sw_mask)
        with torch.cuda.stream(self.copy_stream):           # (sw_mask was produced there: the sync waits for the batch, not for the step in flight)
            return steps_to_run(args, sw_mask)

    def _pinned(self, slot, n, S):
        """two sets of pinned staging buffers, reused: `torch.empty(...).pin_memory()` per batch is a hipHostMalloc + first-touch page faults
        -- 1.9 of the 2.15 ms a batch of two CACHED samples took to stage (tools/exp/stage_bench.py), more than the GPU needs for a tenth
        of its step.  A set is refilled only after the host-to-device copy that last read it has completed (event)."""
        key = (slot & 1, n, tuple(S))
        ent = self._pins.get(key)
        if ent is None:
            ent = [torch.empty((n, 3) + tuple(S), dtype=torch.uint8).pin_memory(), torch.empty((n,) + tuple(S), dtype=torch.int32).pin_memory(), None]
            ent += [ent[0].numpy(), ent[1].numpy()]            # (filled through numpy: a torch CPU copy_ of 200 KB costs ~1 ms of thread-pool wake-up)
            self._pins[key] = ent
        if ent[2] is not None:
            ent[2].synchronize()
            ent[2] = None
        return ent

    def _stage(self, idxs, slot=0):
        """decode one batch into pinned buffers (host side only)"""
        seeds = [self.rng.getrandbits(32) for _ in idxs]
        work = list(zip(idxs, seeds))
        if all(i in self.ds._cache for i in idxs):             # cache hits are index arithmetic: not worth a hand-over to the pool
            items = [self.ds.host_item(i, random.Random(sd)) for i, sd in work]
        else:
            items = list(self.pool.map(lambda a: self.ds.host_item(a[0], random.Random(a[1])), work))
        S = items[0][0].shape[1:]
        ent = self._pinned(slot, len(items), S)
        img, ins = ent[0], ent[1]
        for i, (a, b) in enumerate(items):
            np.copyto(ent[3][i], a)
            np.copyto(ent[4][i], b)
        mats = None
        if self.ds.augmentation_transform is not None:                      # one matrix per sample, drawn as the reference draws them
            with self._lock:
                mats = torch.stack([self.ds.augmentation_transform.matrix(S[0], S[1]) for _ in items])
        return img, ins, mats, ent

    def _to_device(self, staged):
        """H2D copy, normalisation, warp and target construction of one staged batch -- ALL on the copy stream: none of it depends on the
        training step in flight on the caller's stream, and the host syncs inside (torch.unique, the early-stop rule) then wait for this
        stream only, not for the step (at configs[0]'s batch of 2 the step is 8.6 ms and the serialized batch preparation was ~2 ms of GPU
        idle time per iteration).  The caller's stream waits for the copy stream once, at the end."""
        img, ins, mats, ent = staged
        st = self.copy_stream
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(st):
            x = img.to(self.device, non_blocking=True)
            m = ins.to(self.device, non_blocking=True)
            if ent is not None:                 # the staging set may be refilled once these two copies have run
                ent[2] = torch.cuda.Event()
                ent[2].record(st)
            x = (x.float() / 255.0 - self.mean) / self.std                      # ToTensor + Normalize (train.py:34-37)
            mf = m.float().unsqueeze(1)
            if mats is not None:                                                # dataset.py:66-67: image and maps share one warp
                x = affine_nearest(x, mats)
                mf = affine_nearest(mf, mats)
            ins_d = mf.squeeze(1).round().long()
            seg_d = (ins_d > 0).long()                                          # leaves.py:105-106
            y_mask, y_class, sw_mask, sw_class = targets_from_maps(ins_d, seg_d, self.ds.max_seq_len, device=self.device)
            out = (x.contiguous(), y_mask, y_class, sw_mask, sw_class)
        cur.wait_stream(st)
        for t in out:                       # allocated on the copy stream, consumed on the caller's: keep the caching allocator from handing
            t.record_stream(cur)            # the blocks to the next batch's side-stream work while the step still reads them
        return out

    def __iter__(self):
        order = list(range(len(self.ds)))
        if self.shuffle:
            self.order_rng.shuffle(order)
        batches = shard_batches(order, self.bs, self.rank, self.world, self.drop_last)
        if not batches:
            return
        staged = self._stage(batches[0], 0)
        for k in range(len(batches)):
            fut, box = None, []
            if k + 1 < len(batches):                                        # decode the next batch while this one trains
                fut = threading.Thread(target=lambda kk=k + 1, out=box: out.append(self._stage(batches[kk], kk)))
                fut.start()
            yield self._to_device(staged)
            if fut is not None:
                fut.join()
                staged = box[0]


def synthesize_leaves_dir(path, n=100, size=(192, 208), seed=0):
    """Write n synthetic CVPPP-A1-style pairs plantNNN_rgb.png / plantNNN_label.png (instance ids 1..k, 0 = background): elliptical
    'leaves' on a textured background.  For tests and smoke runs of the data path only."""
    from PIL import Image
    os.makedirs(path, exist_ok=True)
    r = np.random.default_rng(seed)
    H, W = size
    yy, xx = np.mgrid[0:H, 0:W]
    for i in range(n):
        rgb = r.integers(0, 80, (H, W, 3)).astype(np.uint8)
        lab = np.zeros((H, W), np.uint8)
        for k in range(1, int(r.integers(3, 9)) + 1):
            cy, cx = r.uniform(0.2, 0.8) * H, r.uniform(0.2, 0.8) * W
            a, b, th = r.uniform(8, 0.25 * H), r.uniform(6, 0.15 * W), r.uniform(0, np.pi)
            u = (yy - cy) * np.cos(th) + (xx - cx) * np.sin(th)
            v = -(yy - cy) * np.sin(th) + (xx - cx) * np.cos(th)
            m = (u / a) ** 2 + (v / b) ** 2 <= 1.0
            lab[m] = k
            rgb[m] = (r.integers(20, 90), r.integers(120, 255), r.integers(20, 90))
        Image.fromarray(rgb).save(os.path.join(path, "plant%03d_rgb.png" % i))
        Image.fromarray(lab).save(os.path.join(path, "plant%03d_label.png" % i))
    return path
