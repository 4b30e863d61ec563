"""Minimal stand-in for the (not installable here) ``inplace_abn`` package, for the drop-in import test only:
BatchNorm2d followed by the activation InPlaceABN fuses (leaky-ReLU 0.01 by default, identity on request)."""
import torch.nn as nn


class InPlaceABN(nn.BatchNorm2d):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation="leaky_relu", slope=0.01):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine)
        self.activation, self.slope = activation, slope

    def forward(self, x):
        y = super().forward(x)
        if self.activation == "leaky_relu":
            return nn.functional.leaky_relu(y, self.slope)
        if self.activation == "relu":
            return nn.functional.relu(y)
        return y


InPlaceABNSync = InPlaceABN
