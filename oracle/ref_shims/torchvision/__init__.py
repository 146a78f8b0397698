"""Minimal `torchvision` stand-in so that the UNMODIFIED reference modules
(src/modules/{model,vision}.py, src/test.py) import in the build container, where the real
torchvision is absent.  Only oracle/make_golden.py puts this directory on sys.path.  The ResNet
arithmetic is the oracle's restatement of torchvision's published definition (the reference
never vendored or pinned torchvision: README.md:17)."""
from . import models, transforms  # noqa: F401
