"""Variance-scaling initialised Linear, as used by AdaGN (reference models/dense.py:53-68)."""
import math

import torch
import torch.nn as nn
from torch.nn.init import _calculate_fan_in_and_fan_out


def variance_scaling_init_(tensor, scale):
    """U(-b, b), b = sqrt(3 * gain / fan_avg-as-implemented): the reference passes mode='fan_avg'
    to a helper that only distinguishes 'fan_in' (dense.py:27-28), so fan_out is what is used."""
    gain = 1e-10 if scale == 0 else scale
    _, fan_out = _calculate_fan_in_and_fan_out(tensor)
    bound = math.sqrt(3.0 * gain / max(1.0, fan_out))
    with torch.no_grad():
        return tensor.uniform_(-bound, bound)


def dense(in_channels, out_channels, init_scale=1.0):
    lin = nn.Linear(in_channels, out_channels)
    variance_scaling_init_(lin.weight, init_scale)
    nn.init.zeros_(lin.bias)
    return lin
