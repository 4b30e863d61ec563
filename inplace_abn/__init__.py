"""``inplace_abn`` for PyTorch-ROCm: the two classes ``networks/ccnet.py:16-17`` imports.

The reference depends on mapillary/inplace_abn (a CUDA extension: not installable on ROCm, not vendored in the
reference tree).  What CCNet uses of it is small -- ``InPlaceABNSync(C)`` (batch norm + leaky-ReLU(0.01)) in the
RCCA / DSN heads (ccnet.py:103-112,150-155) and ``InPlaceABNSync(C, activation='identity')`` as the backbone's
BatchNorm2d (ccnet.py:17) -- so it is restated here on torch ops: same constructor arguments, same parameter and
buffer names (``weight, bias, running_mean, running_var`` and nothing else, so checkpoints written with the real
package load strictly), cross-rank statistics through ``torch.distributed`` (RCCL on MI355X) when a process group
is up.  The memory trick that gives the package its name (recomputing the BN input from its output in backward)
is not reproduced: 288 GB of HBM per MI355X make it unnecessary at CCNet's sizes.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

__all__ = ["ABN", "InPlaceABN", "InPlaceABNSync", "ACT_LEAKY_RELU", "ACT_RELU", "ACT_ELU", "ACT_NONE"]

ACT_LEAKY_RELU, ACT_RELU, ACT_ELU, ACT_NONE = "leaky_relu", "relu", "elu", "identity"


class ABN(nn.Module):
    """Batch normalisation followed by an activation (default leaky-ReLU with slope 0.01)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation=ACT_LEAKY_RELU,
                 activation_param=0.01, slope=None):
        super().__init__()
        if activation not in (ACT_LEAKY_RELU, ACT_RELU, ACT_ELU, ACT_NONE, "none"):
            raise ValueError(f"unknown activation '{activation}'")
        self.num_features, self.eps, self.momentum, self.affine = num_features, eps, momentum, affine
        self.activation = activation
        self.activation_param = activation_param if slope is None else slope
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def _normalise(self, x):
        return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias,
                            self.training, self.momentum, self.eps)

    def _activate(self, y):
        if self.activation == ACT_LEAKY_RELU:
            return F.leaky_relu(y, negative_slope=self.activation_param)
        if self.activation == ACT_RELU:
            return F.relu(y)
        if self.activation == ACT_ELU:
            return F.elu(y, alpha=self.activation_param)
        return y

    def forward(self, x):
        return self._activate(self._normalise(x))

    def extra_repr(self):
        return (f"{self.num_features}, eps={self.eps}, momentum={self.momentum}, affine={self.affine}, "
                f"activation={self.activation}[{self.activation_param}]")


class InPlaceABN(ABN):
    """Same arithmetic as :class:`ABN` (the in-place memory saving is not reproduced, see the module docstring)."""


class InPlaceABNSync(ABN):
    """:class:`ABN` whose training statistics are reduced over all ranks of the default process group
    (``engine.py:52-57`` runs one process per GPU) -- the all-reduces go over RCCL / xGMI."""

    def _normalise(self, x):
        if (self.training and x.is_cuda and dist.is_available() and dist.is_initialized()
                and dist.get_world_size() > 1):
            from torch.nn.modules._functions import SyncBatchNorm as sync_batch_norm
            return sync_batch_norm.apply(x, self.weight, self.bias, self.running_mean, self.running_var,
                                         self.eps, self.momentum, dist.group.WORLD, dist.get_world_size())
        return super()._normalise(x)
