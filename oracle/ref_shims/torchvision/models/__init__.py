from . import resnet, vgg  # noqa: F401
from .resnet import resnet101, resnet50, resnet34  # noqa: F401
from .vgg import vgg16  # noqa: F401
