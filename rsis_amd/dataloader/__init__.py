"""Target-tensor construction of the reference's data layer (reference src/dataloader/dataset.py:86-146), SURVEY.md section
8(f) row N3.  The dataset readers themselves (PNG / HDF5 / Cityscapes JSON decoding, PIL augmentation) are host I/O outside
the hot path and are not part of this build; what IS here is the step that turns an instance-id map + a class map into what
`runIter` consumes."""
from .targets import sequence_from_masks, targets_from_maps  # noqa: F401
