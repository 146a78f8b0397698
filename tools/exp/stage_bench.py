import os, sys, time, argparse
sys.path.insert(0, os.getcwd())
import torch
from rsis_amd.dataloader.leaves import LeavesDataset, DeviceLoader, synthesize_leaves_dir
d = synthesize_leaves_dir("/tmp/sb/A1", n=104, size=(272, 288), seed=3)
a = argparse.Namespace(gt_maxseqlen=16, batch_size=2, leaves_dir=d, leaves_test_dir=d, rotation=10, translation=0.1, shear=0.1, zoom=0.7)
ds = LeavesDataset(a, split="train", resize=True, imsize=256)
ld = DeviceLoader(ds, 2, num_workers=4)
idx = [[2*i, 2*i+1] for i in range(48)]
for rep in range(3):
    t = time.time()
    for b in idx: s = ld._stage(b)
    print("stage pass %d: %.2f ms per batch" % (rep, 1e3*(time.time()-t)/48))
t = time.time()
for _ in range(48):
    x = torch.empty((2,3,256,256), dtype=torch.uint8).pin_memory(); y = torch.empty((2,256,256), dtype=torch.int32).pin_memory()
print("pin_memory pair: %.2f ms" % (1e3*(time.time()-t)/48))
torch.cuda.synchronize()
t = time.time()
for b in idx[:24]:
    out = ld._to_device(ld._stage(b)); torch.cuda.synchronize()
print("_stage + _to_device + sync: %.2f ms per batch" % (1e3*(time.time()-t)/24))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for b in idx[:24]:
    out = ld._to_device(ld._stage(b)); torch.cuda.synchronize()
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(12)
