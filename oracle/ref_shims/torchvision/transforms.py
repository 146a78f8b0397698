"""Placeholder: the reference imports `torchvision.transforms` at module scope
(model.py:7, test.py:7) but the forward path never touches it."""
