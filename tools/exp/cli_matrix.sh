#!/bin/bash
# Exploratory: `python -m rsis_amd.train` / eval under less common flag combinations of the reference (args.py), one short epoch each on a
# synthesised CVPPP A1 directory.  Prints one line per combination (rc + last error line).   bash tools/exp/cli_matrix.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
D=/tmp/cli_matrix
rm -rf $D; mkdir -p $D
python - <<PY
from rsis_amd.dataloader.leaves import synthesize_leaves_dir
synthesize_leaves_dir("$D/A1", n=104, size=(150, 140), seed=9)      # the first 96 images train, the rest validate (leaves.py:72-94)
PY
BASE="-dataset leaves -leaves_dir $D/A1 -leaves_test_dir $D/A1 -num_classes 2 --resize -imsize 128 -maxseqlen 5 -gt_maxseqlen 10 -batch_size 4 -base_model resnet101 -hidden_size 32 --log_term -max_epoch 1 -print_every 50 -models_root $D/models -num_workers 2"
run() {  # name, extra flags
  local name=$1; shift
  timeout 600 python -m rsis_amd.train -model_name $name $BASE "$@" > $D/$name.log 2>&1
  local rc=$?
  local nanc=$(grep -ci "nan" $D/$name.log)
  echo "$name rc=$rc nan_lines=$nanc :: $(grep -E "Epoch 0:.*train" $D/$name.log | tail -1 | cut -c1-80) :: $(grep -E "Error|error|Traceback" $D/$name.log | tail -1 | cut -c1-160)"
}
run plain
run sum -skip_mode sum
run mul -skip_mode mul
run none -skip_mode none
run k1 -kernel_size 1
run cl --curriculum_learning -steps_cl 1 -min_steps 2
run aug --augment
run cls -class_loss_after 0 -stop_loss_after 0
run bf16g -dtype bf16 --graph
run graph --graph -class_loss_after 0 -stop_loss_after 0
run quirk --enc_lr_quirk
run ft -finetune_after 0 -class_loss_after 0
run resume_src -class_loss_after 0 -stop_loss_after 0
timeout 600 python -m rsis_amd.train -model_name resume_src $BASE --resume -max_epoch 2 -epoch_resume 1 > $D/resume.log 2>&1; echo "resume rc=$? :: $(grep -E "Epoch" $D/resume.log | tail -1 | cut -c1-80) :: $(grep -E "Error|Traceback" $D/resume.log | tail -1 | cut -c1-160)"
timeout 600 python -m rsis_amd.eval --synthetic -model_name resume_src -models_root $D/models -batch_size 4 > $D/eval.log 2>&1; echo "eval rc=$? :: $(tail -2 $D/eval.log | tr '\n' ' ' | cut -c1-200)"
timeout 600 python -m rsis_amd.eval_leaves -model_name resume_src -leaves_dir $D/A1 -leaves_test_dir $D/A1 -models_root $D/models -batch_size 4 > $D/eval_leaves.log 2>&1; echo "eval_leaves rc=$? :: $(tail -2 $D/eval_leaves.log | tr '\n' ' ' | cut -c1-200)"
