"""Command-line flags -- the same names, single-/double-dash spelling, defaults and `dest`s as the reference's
src/args.py:get_parser (args.py:3-157), so that its scripts/*.sh invocations parse unchanged.  Additions (all
non-breaking) are at the end: --synthetic, -synthetic_batches, -local_rank plumbing via torchrun env.
"""
import argparse

# (flag, kwargs) table; booleans are store_true/store_false with set_defaults below
_FLAGS = [
    # training
    ("--resume", dict(dest="resume", action="store_true")),
    ("-epoch_resume", dict(dest="epoch_resume", default=0, type=int)),
    ("-seed", dict(dest="seed", default=123, type=int)),
    ("-batch_size", dict(dest="batch_size", default=28, type=int)),
    ("-lr", dict(dest="lr", default=1e-3, type=float)),
    ("-lr_cnn", dict(dest="lr_cnn", default=1e-6, type=float)),
    ("-optim_cnn", dict(dest="optim_cnn", default="adam", choices=["adam", "sgd", "rmsprop"])),
    ("-momentum", dict(dest="momentum", default=0.9, type=float)),
    ("-weight_decay", dict(dest="weight_decay", default=1e-6, type=float)),
    ("-weight_decay_cnn", dict(dest="weight_decay_cnn", default=1e-6, type=float)),
    ("-optim", dict(dest="optim", default="adam", choices=["adam", "sgd", "rmsprop"])),
    ("-maxseqlen", dict(dest="maxseqlen", default=10, type=int)),
    ("-gt_maxseqlen", dict(dest="gt_maxseqlen", default=20, type=int)),
    ("-best_val_loss", dict(dest="best_val_loss", default=1000, type=float)),
    ("--crop", dict(dest="crop", action="store_true")),
    ("--smooth_curves", dict(dest="smooth_curves", action="store_true")),
    # base model fine tuning
    ("-finetune_after", dict(dest="finetune_after", default=0, type=int)),
    ("--update_encoder", dict(dest="update_encoder", action="store_true")),
    ("--transfer", dict(dest="transfer", action="store_true")),
    ("-transfer_from", dict(dest="transfer_from", default="model")),
    ("--curriculum_learning", dict(dest="curriculum_learning", action="store_true")),
    ("-steps_cl", dict(dest="steps_cl", default=1, type=int)),
    ("-min_steps", dict(dest="min_steps", default=1, type=int)),
    ("-min_delta", dict(dest="min_delta", default=0.0, type=float)),
    # losses
    ("-class_loss_after", dict(dest="class_loss_after", default=20, type=int)),
    ("--use_class_loss", dict(dest="use_class_loss", action="store_true")),
    ("-stop_loss_after", dict(dest="stop_loss_after", default=3000, type=int)),
    ("--use_stop_loss", dict(dest="use_stop_loss", action="store_true")),
    # stopping criterion
    ("-patience", dict(dest="patience", default=15, type=int)),
    ("-patience_stop", dict(dest="patience_stop", default=60, type=int)),
    ("-max_epoch", dict(dest="max_epoch", default=4000, type=int)),
    # logging
    ("-print_every", dict(dest="print_every", default=10, type=int)),
    ("--log_term", dict(dest="log_term", action="store_true")),
    ("--visdom", dict(dest="visdom", action="store_true")),
    ("-port", dict(dest="port", default=8097, type=int)),
    ("-server", dict(dest="server", default="http://localhost")),
    # loss weights
    ("-class_weight", dict(dest="class_weight", default=0.1, type=float)),
    ("-iou_weight", dict(dest="iou_weight", default=1.0, type=float)),
    ("-stop_weight", dict(dest="stop_weight", default=0.5, type=float)),
    ("-stop_balance_weight", dict(dest="stop_balance_weight", default=0.5, type=float)),
    # augmentation
    ("--augment", dict(dest="augment", action="store_true")),
    ("-rotation", dict(dest="rotation", default=10, type=int)),
    ("-translation", dict(dest="translation", default=0.1, type=float)),
    ("-shear", dict(dest="shear", default=0.1, type=float)),
    ("-zoom", dict(dest="zoom", default=0.7, type=float)),
    # device
    ("--cpu", dict(dest="use_gpu", action="store_false")),
    ("-ngpus", dict(dest="ngpus", default=1, type=int)),
    # model
    ("-base_model", dict(dest="base_model", default="resnet101", choices=["resnet101", "resnet50", "resnet34", "vgg16"])),
    ("-skip_mode", dict(dest="skip_mode", default="concat", choices=["sum", "concat", "mul", "none"])),
    ("-model_name", dict(dest="model_name", default="model")),
    ("-log_file", dict(dest="log_file", default="train.log")),
    ("-hidden_size", dict(dest="hidden_size", default=128, type=int)),
    ("-kernel_size", dict(dest="kernel_size", default=3, type=int)),
    ("-dropout", dict(dest="dropout", default=0.0, type=float)),
    ("-dropout_stop", dict(dest="dropout_stop", default=0.0, type=float)),
    ("-dropout_cls", dict(dest="dropout_cls", default=0.0, type=float)),
    # dataset
    ("-imsize", dict(dest="imsize", default=256, type=int)),
    ("--resize", dict(dest="resize", action="store_true")),
    ("-num_classes", dict(dest="num_classes", default=21, type=int)),
    ("-dataset", dict(dest="dataset", default="pascal", choices=["pascal", "cityscapes", "leaves"])),
    ("-pascal_dir", dict(dest="pascal_dir", default="/work/asalvador/dev/data/rsis/VOCAug/")),
    ("-cityscapes_dir", dict(dest="cityscapes_dir", default="/gpfs/scratch/bsc31/bsc31429/CityScapes/")),
    ("-leaves_dir", dict(dest="leaves_dir", default="/gpfs/scratch/bsc31/bsc31429/LeavesDataset/A1/")),
    ("-leaves_test_dir", dict(dest="leaves_test_dir", default="/gpfs/scratch/bsc31/bsc31429/CVPPP2014_LSC_testing_data/A1/")),
    ("-num_workers", dict(dest="num_workers", default=4, type=int)),
    # testing
    ("-eval_split", dict(dest="eval_split", default="test")),
    ("-mask_th", dict(dest="mask_th", default=0.5, type=float)),
    ("-stop_th", dict(dest="stop_th", default=0.5, type=float)),
    ("-class_th", dict(dest="class_th", default=0.5, type=float)),
    ("-max_dets", dict(dest="max_dets", default=100, type=int)),
    ("-min_size", dict(dest="min_size", default=0.001, type=float)),
    ("-cat_id", dict(dest="cat_id", default=-1, type=int)),
    ("--ignore_cats", dict(dest="use_cats", action="store_false")),
    ("--display", dict(dest="display", action="store_true")),
    ("--no_display_text", dict(dest="no_display_text", action="store_true")),
    ("--all_classes", dict(dest="all_classes", action="store_true")),
    ("--no_run_coco_eval", dict(dest="no_run_coco_eval", action="store_true")),
    ("--display_route", dict(dest="display_route", action="store_true")),
    # ---- additions of this build (non-breaking) ----
    ("--synthetic", dict(dest="synthetic", action="store_true")),
    ("-synthetic_batches", dict(dest="synthetic_batches", default=20, type=int)),
    ("-synthetic_instances", dict(dest="synthetic_instances", default=12, type=int)),
    ("-models_root", dict(dest="models_root", default="../models")),
    # arithmetic of the conv / ConvLSTM-gate MFMA kernels: fp32 (exact-f32 MFMA) or bf16 operands with fp32 accumulation
    ("-dtype", dict(dest="dtype", default="fp32", choices=["fp32", "bf16"])),
    # capture one training iteration per (shapes, T, loss switches) as a hipGraph and replay it (no host work per kernel)
    ("--graph", dict(dest="graph", action="store_true")),
    # reproduce the reference's accidental 1x/3x/4x trunk learning rate (utils/utils.py:34-52: duplicated tensors handed to Adam)
    ("--enc_lr_quirk", dict(dest="enc_lr_quirk", action="store_true")),
]

_DEFAULTS = dict(resume=False, crop=False, smooth_curves=False, update_encoder=False, transfer=False,
                 curriculum_learning=False, use_class_loss=False, use_stop_loss=False, log_term=False, visdom=False,
                 augment=False, use_gpu=True, resize=False, display=False, display_route=False, use_cats=True,
                 all_classes=False, no_display_text=False, use_gt_cats=False, use_gt_masks=False, use_gt_stop=False,
                 synthetic=False, graph=False, enc_lr_quirk=False)


def get_parser():
    parser = argparse.ArgumentParser(description="RIASS")
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **kw)
    parser.set_defaults(**_DEFAULTS)
    return parser
