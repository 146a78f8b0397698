#!/bin/bash
# Second exploratory pass (see cli_matrix.sh): corner sizes and the announced fallbacks through the CLI.   bash tools/exp/cli_matrix2.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
D=/tmp/cli_matrix2
rm -rf $D; mkdir -p $D
python - <<PY
from rsis_amd.dataloader.leaves import synthesize_leaves_dir
synthesize_leaves_dir("$D/A1", n=104, size=(150, 140), seed=10)
PY
BASE="-dataset leaves -leaves_dir $D/A1 -leaves_test_dir $D/A1 -num_classes 2 --resize -base_model resnet101 --log_term -max_epoch 1 -print_every 50 -models_root $D/models -num_workers 2 -class_loss_after 0 -stop_loss_after 0"
run() {
  local name=$1; shift
  timeout 900 python -m rsis_amd.train -model_name $name $BASE "$@" > $D/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc nan_lines=$(grep -ci nan $D/$name.log) :: $(grep -E "Epoch 0:.*val" $D/$name.log | tail -1 | cut -c1-80) :: $(grep -E "Error|error|Traceback|\[train\]|\[rsis" $D/$name.log | tail -1 | cut -c1-200)"
}
run t1 -imsize 128 -maxseqlen 1 -gt_maxseqlen 10 -batch_size 4 -hidden_size 32
run b2 -imsize 128 -maxseqlen 4 -gt_maxseqlen 10 -batch_size 2 -hidden_size 32
run h64 -imsize 128 -maxseqlen 4 -gt_maxseqlen 10 -batch_size 4 -hidden_size 64
run h128g -imsize 96 -maxseqlen 4 -gt_maxseqlen 10 -batch_size 4 -hidden_size 128 --graph
run gt40 -imsize 128 -maxseqlen 4 -gt_maxseqlen 40 -batch_size 4 -hidden_size 32
run gt70 -imsize 128 -maxseqlen 4 -gt_maxseqlen 70 -batch_size 4 -hidden_size 32
run gt70g -imsize 128 -maxseqlen 4 -gt_maxseqlen 70 -batch_size 4 -hidden_size 32 --graph
run im160 -imsize 160 -maxseqlen 4 -gt_maxseqlen 10 -batch_size 4 -hidden_size 32
run im200bf -imsize 200 -maxseqlen 4 -gt_maxseqlen 10 -batch_size 4 -hidden_size 32 -dtype bf16
run frozen_bf16g -imsize 128 -maxseqlen 4 -gt_maxseqlen 10 -batch_size 4 -hidden_size 32 -dtype bf16 --graph -finetune_after 5
run r50 -imsize 128 -maxseqlen 4 -gt_maxseqlen 10 -batch_size 4 -hidden_size 32 -base_model resnet50
run drop -imsize 128 -maxseqlen 4 -gt_maxseqlen 10 -batch_size 4 -hidden_size 32 -dropout 0.2 -dropout_cls 0.2 -dropout_stop 0.2
timeout 600 python -m rsis_amd.eval --synthetic -model_name h64 -models_root $D/models -batch_size 4 -dtype bf16 > $D/eval_bf16.log 2>&1; echo "eval_bf16 rc=$? :: $(tail -1 $D/eval_bf16.log | cut -c1-200)"
timeout 600 python -m rsis_amd.eval_cityscapes --synthetic -model_name h64 -models_root $D/models -batch_size 4 > $D/eval_cs.log 2>&1; echo "eval_cityscapes rc=$? :: $(tail -1 $D/eval_cs.log | cut -c1-200)"
