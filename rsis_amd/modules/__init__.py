from .clstm import ConvLSTMCell  # noqa: F401
from .model import FeatureExtractor, RSIS  # noqa: F401
from .vision import ResNet101  # noqa: F401
