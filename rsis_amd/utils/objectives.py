"""nn.Module loss wrappers -- same classes / call signatures as reference src/utils/objectives.py."""
import torch
import torch.nn as nn

from .hungarian import MaskedNLL, StableBalancedMaskedBCE, softIoU


class MaskedNLLLoss(nn.Module):
    def __init__(self, balance_weight=None):
        super().__init__()
        self.balance_weight = balance_weight

    def forward(self, y_true, y_pred, sw):
        costs = MaskedNLL(y_true, y_pred, self.balance_weight).view(-1, 1)   # objectives.py:11
        return torch.masked_select(costs, sw.bool())                          # :13 (un-reduced)


class MaskedBCELoss(nn.Module):
    def __init__(self, balance_weight=None):
        super().__init__()
        self.balance_weight = balance_weight

    def forward(self, y_true, y_pred, sw):
        costs = StableBalancedMaskedBCE(y_true, y_pred, self.balance_weight).view(-1, 1)   # objectives.py:22
        return torch.masked_select(costs, sw.bool())                                        # :23


class softIoULoss(nn.Module):
    def forward(self, y_true, y_pred, sw):
        costs = softIoU(y_true, y_pred).view(-1, 1)                       # objectives.py:31
        return torch.mean(torch.masked_select(costs, sw.bool()))          # :32
