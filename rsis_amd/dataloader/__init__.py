"""Target-tensor construction of the reference's data layer (reference src/dataloader/dataset.py:86-146), SURVEY.md section
8(f) row N3: the step that turns an instance-id map + a class map into what `runIter` consumes (targets), the affine augmentation
(augment), and the CVPPP A1 leaves reader + device batch loader of BASELINE configs[0] (leaves).  The Pascal VOC (HDF5 / SBD) and
Cityscapes (JSON polygons) readers are host I/O outside the hot path and are not part of this build."""
from .targets import sequence_from_masks, targets_from_maps  # noqa: F401
