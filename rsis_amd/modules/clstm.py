"""ConvLSTMCell -- drop-in for reference src/modules/clstm.py:7-62 on MI355X.

Same constructor, same `Gates` parameter (reference layout [4*hid, in+hid, k, k], gate order i,f,o,g: clstm.py:17,47),
same forward contract (prev_state None or (h, c); returns the python list [hidden, cell]: clstm.py:60-62).
The cat + conv + chunk + sigmoid/tanh + cell update of clstm.py:43-58 run as ONE fused gfx950 kernel
(rsis_convlstm_fwd); the backward is rsis_convlstm_bwd_gates + rsis_conv2d_dgrad + rsis_conv2d_wgrad.
"""
import math

import torch
from torch import nn

from .. import ops


class _GatesParams(nn.Module):
    """Holds `weight` / `bias` under the reference's `Gates.*` state_dict keys with nn.Conv2d's default init."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(cin * k * k)
        nn.init.uniform_(self.bias, -bound, bound)


class ConvLSTMCell(nn.Module):
    """Generate a convolutional LSTM cell (reference clstm.py:7-17)."""

    def __init__(self, args, input_size, hidden_size, kernel_size, padding):
        super().__init__()
        self.use_gpu = getattr(args, "use_gpu", True)
        self.input_size = int(input_size)
        self.hidden_size = int(hidden_size)
        self.kernel_size = int(kernel_size)
        self.padding = int(padding)
        self.Gates = _GatesParams(self.input_size + self.hidden_size, 4 * self.hidden_size, self.kernel_size)
        self._packs = {}
        self.dtype = ops.DTYPES[getattr(args, "dtype", "fp32")]

    def _set_rsis_dtype(self, d):
        if self.dtype != d:
            self.dtype = d
            self._packs = {}

    def _pack(self, x_channels):
        key = tuple(x_channels)
        if key not in self._packs:
            if sum(key) != self.input_size:
                raise Exception("ConvLSTMCell: input has %d channels, expected %d" % (sum(key), self.input_size))
            self._packs[key] = ops.PackedConv(self.kernel_size, list(key) + [self.hidden_size], lstm_hid=self.hidden_size, stride=1,
                                              pad=self.padding, dtype=self.dtype)
        return self._packs[key]

    def forward_multi(self, inputs, prev_state):
        """Same as forward() with the cell input given as a list of tensors whose channel concat is `input_`
        (the torch.cat of model.py:153 is folded into the kernel's operand loader)."""
        pack = self._pack([t.shape[1] for t in inputs])
        h, c = ops.convlstm(list(inputs), prev_state, self.Gates.weight, self.Gates.bias, self.padding, pack)
        return [h, c]

    def forward(self, input_, prev_state):
        return self.forward_multi([input_], prev_state)
