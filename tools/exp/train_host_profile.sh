#!/bin/bash
# cProfile of `python -m rsis_amd.train --graph` on a synthesised CVPPP A1 directory (configs[0] flags): where the HOST spends an epoch.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
D=/tmp/thp; rm -rf $D; mkdir -p $D
python - <<PY
from rsis_amd.dataloader.leaves import synthesize_leaves_dir
synthesize_leaves_dir("$D/A1", n=104, size=(272, 288), seed=3)
PY
python -m cProfile -o $D/prof.out -m rsis_amd.train -dataset leaves -leaves_dir $D/A1 -leaves_test_dir $D/A1 -imsize 256 --resize -batch_size ${1:-2} -maxseqlen 16 -gt_maxseqlen 16 -num_classes 2 --log_term -max_epoch 6 -print_every 1000 -model_name thp -models_root $D/models -num_workers 4 -class_loss_after -1 --graph > $D/run.log 2>&1
python - <<PY
import pstats
p = pstats.Stats("$D/prof.out"); p.sort_stats("tottime").print_stats(40)
PY
