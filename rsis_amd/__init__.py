"""rsis_amd -- MI355X (gfx950) native implementation of the RSIS hot path:
ResNet-101 feature-pyramid encoder -> 5-scale ConvLSTM recurrent decoder, forward + backward.

Module surface mirrors the reference's src/modules (FeatureExtractor, RSIS, ConvLSTMCell) and its train.py /
test.py / eval.py entry points; all arithmetic runs in librsis_hip.so (hand-written HIP, C ABI in
include/rsis_hip.h).  There is no CPU fallback.
"""
__version__ = "0.1.0"
