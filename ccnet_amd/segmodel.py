"""The callers either side of the hot path: CCNet's RCCA head and the dilated
ResNet-101 it sits on, restated on PyTorch-ROCm around the MI355X criss-cross attention module.

It mirrors ``/root/reference/networks/ccnet.py``:

    RCCAModule(in_channels, out_channels, num_classes)      ccnet.py:99-123   3x3 conv+ABN -> CCA x R -> 3x3
                                                                               conv+ABN -> cat(x, .) -> classifier
    Seg_Model(num_classes, criterion, recurrence)           ccnet.py:125-201  deep-stem ResNet-101, output stride
                                                                               8 (layer3 dilation 2, layer4
                                                                               dilation 4), DSN head on layer3
    CriterionDSN                                            loss/criterion.py:11-35

with the SAME module attribute names, so a ``state_dict`` written by the reference (with the real ``inplace_abn``)
loads strictly (``tests/test_segmodel.py`` compares the key/shape tables where the reference tree is mounted).
Only the criss-cross attention inside is hand-written HIP; convolutions, normalisation and the loss are torch ops
(MIOpen / hipBLASLt), ``InPlaceABNSync`` is this repository's ``inplace_abn`` restatement.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from inplace_abn import InPlaceABNSync

from .functions import CrissCrossAttention

__all__ = ["RCCAModule", "ResNetCCNet", "Seg_Model", "CriterionDSN", "load_model"]

# (planes, blocks, stride, dilation) of the four residual stages -- ccnet.py:142-145, 195
_RESNET101_STAGES = ((64, 3, 1, 1), (128, 4, 2, 1), (256, 23, 1, 2), (512, 3, 1, 4))
_EXPANSION = 4


def _bn(channels):
    """The backbone's normalisation: ABN with the activation switched off (ccnet.py:17)."""
    return InPlaceABNSync(channels, activation="identity")


def _conv_abn(cin, cout, bias=False):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=bias), InPlaceABNSync(cout))


class Bottleneck(nn.Module):
    """1x1 -> 3x3 (dilated) -> 1x1 residual unit (ccnet.py:32-69)."""

    def __init__(self, cin, planes, stride=1, dilation=1, project=False):
        super().__init__()
        cout = planes * _EXPANSION
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = _bn(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = _bn(planes)
        self.conv3 = nn.Conv2d(planes, cout, 1, bias=False)
        self.bn3 = _bn(cout)
        self.downsample = (nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), _bn(cout))
                           if project else None)

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class RCCAModule(nn.Module):
    """Recurrent criss-cross attention head.  ``forward(x, recurrence)`` applies the SAME attention module
    ``recurrence`` times (shared weights; R = 2 in the published recipe)."""

    def __init__(self, in_channels, out_channels, num_classes):
        super().__init__()
        inter = in_channels // 4
        self.conva = _conv_abn(in_channels, inter)
        self.cca = CrissCrossAttention(inter)
        self.convb = _conv_abn(inter, inter)
        self.bottleneck = nn.Sequential(
            nn.Conv2d(in_channels + inter, out_channels, 3, padding=1, bias=False),
            InPlaceABNSync(out_channels),
            nn.Dropout2d(0.1),
            nn.Conv2d(out_channels, num_classes, 1, bias=True))

    def forward(self, x, recurrence=1):
        y = self.conva(x)
        for _ in range(recurrence):
            y = self.cca(y)
        y = self.convb(y)
        return self.bottleneck(torch.cat([x, y], 1))


class ResNetCCNet(nn.Module):
    """Deep-stem dilated ResNet + RCCA head + DSN head; returns ``[main_logits, dsn_logits]`` at 1/8 resolution,
    or the loss when a criterion and labels are given (ccnet.py:171-187)."""

    def __init__(self, num_classes, criterion=None, recurrence=2, stages=_RESNET101_STAGES):
        super().__init__()
        # stem: three 3x3 convolutions (the first with stride 2) and a ceil-mode max-pool -> 1/4 resolution
        self.conv1 = nn.Conv2d(3, 64, 3, stride=2, padding=1, bias=False)
        self.bn1 = _bn(64)
        self.conv2 = nn.Conv2d(64, 64, 3, padding=1, bias=False)
        self.bn2 = _bn(64)
        self.conv3 = nn.Conv2d(64, 128, 3, padding=1, bias=False)
        self.bn3 = _bn(128)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1, ceil_mode=True)
        cin = 128
        for idx, (planes, blocks, stride, dilation) in enumerate(stages, start=1):
            units = []
            for u in range(blocks):
                first = u == 0
                units.append(Bottleneck(cin, planes, stride if first else 1, dilation,
                                        project=first and (stride != 1 or cin != planes * _EXPANSION)))
                cin = planes * _EXPANSION
            setattr(self, f"layer{idx}", nn.Sequential(*units))
        c3 = stages[2][0] * _EXPANSION
        self.head = RCCAModule(cin, 512, num_classes)
        self.dsn = nn.Sequential(nn.Conv2d(c3, 512, 3, padding=1), InPlaceABNSync(512), nn.Dropout2d(0.1),
                                 nn.Conv2d(512, num_classes, 1, bias=True))
        self.criterion = criterion
        self.recurrence = recurrence

    def forward(self, x, labels=None):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x = F.relu(self.bn3(self.conv3(x)))
        x = self.layer2(self.layer1(self.maxpool(x)))
        x = self.layer3(x)
        aux = self.dsn(x)
        outs = [self.head(self.layer4(x), self.recurrence), aux]
        if self.criterion is not None and labels is not None:
            return self.criterion(outs, labels)
        return outs


def load_model(model, model_file):
    """Checkpoint loading with the semantics of the reference's ``utils/pyt_utils.py:47-85`` (what ccnet.py:198-199
    calls): unwrap a ``{'model': ...}`` (or ``{'state_dict': ...}``) wrapper, strip a DataParallel ``module.`` prefix,
    load non-strictly, WARN about missing / unexpected keys and refuse a checkpoint none of whose keys match
    (the reference would silently train from random init)."""
    import logging
    log = logging.getLogger("ccnet_amd.segmodel")
    state = torch.load(model_file, map_location="cpu") if isinstance(model_file, (str, bytes)) or hasattr(model_file, "read") else model_file
    if isinstance(state, dict):
        for wrapper in ("model", "state_dict"):
            if wrapper in state and isinstance(state[wrapper], dict):
                state = state[wrapper]
                break
    own = set(model.state_dict().keys())
    if state and all(k.startswith("module.") for k in state) and not any(k.startswith("module.") for k in own):
        state = {k[len("module."):]: v for k, v in state.items()}
    ckpt = set(state.keys())
    if own and ckpt and not (own & ckpt):
        raise RuntimeError(f"checkpoint shares no key with the model ({len(ckpt)} keys, e.g. {sorted(ckpt)[:3]}); "
                           "refusing to continue from random initialisation")
    model.load_state_dict(state, strict=False)
    missing, unexpected = sorted(own - ckpt), sorted(ckpt - own)
    if missing:
        log.warning("Missing key(s) in state_dict: %s", ", ".join(missing))
    if unexpected:
        log.warning("Unexpected key(s) in state_dict: %s", ", ".join(unexpected))
    return model


def Seg_Model(num_classes, criterion=None, pretrained_model=None, recurrence=0, **kwargs):
    """Same call as ccnet.py:195-201 (``pretrained_model``: checkpoint path, loaded by :func:`load_model`)."""
    model = ResNetCCNet(num_classes, criterion, recurrence)
    if pretrained_model is not None:
        load_model(model, pretrained_model)
    return model


class CriterionDSN(nn.Module):
    """Cross-entropy on the up-sampled main logits + 0.4 x cross-entropy on the up-sampled DSN logits
    (loss/criterion.py:22-35); label 255 is ignored."""

    def __init__(self, ignore_index=255, aux_weight=0.4):
        super().__init__()
        self.ignore_index, self.aux_weight = ignore_index, aux_weight

    def forward(self, preds, target):
        size = target.shape[1:]
        losses = [F.cross_entropy(F.interpolate(p, size=size, mode="bilinear", align_corners=True), target,
                                  ignore_index=self.ignore_index) for p in preds[:2]]
        return losses[0] if len(losses) == 1 else losses[0] + self.aux_weight * losses[1]
