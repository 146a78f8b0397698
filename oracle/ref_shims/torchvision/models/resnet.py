"""torchvision.models.resnet stand-in -> the oracle's ResNet restatement."""
from oracle.rsis_oracle import ResNet, Bottleneck, BasicBlock  # noqa: F401


def resnet101(pretrained=False, **kw):
    # `pretrained=True` would download weights; here an un-trained net is returned and the
    # golden generator overwrites every tensor with the deterministic filler.
    return ResNet(Bottleneck, [3, 4, 23, 3])


def resnet50(pretrained=False, **kw):
    return ResNet(Bottleneck, [3, 4, 6, 3])


def resnet34(pretrained=False, **kw):
    return ResNet(BasicBlock, [3, 4, 6, 3])
