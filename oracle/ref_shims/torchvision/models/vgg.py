"""torchvision.models.vgg stand-in: names only (VGG16 is out of scope; vision.py:2 imports them)."""
import torch.nn as nn


class VGG(nn.Module):
    def __init__(self, features=None, num_classes=1000):
        super().__init__()
        self.features = features


def make_layers(cfg, batch_norm=False):
    return nn.Sequential()


def vgg16(pretrained=False, **kw):
    return VGG(make_layers([]))
