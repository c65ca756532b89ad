"""Functional stand-ins for torchvision.models.mobilenetv3.{InvertedResidualConfig, InvertedResidual} (build-container only).

Written from the published MobileNetV3 block algorithm as torchvision implements it -- NOT a copy of its source:
  config: channel counts rounded with _make_divisible(c * width_mult, 8); use_hs = (activation == "HS");
  block = [1x1 expand conv + norm + act  (only if expanded != input channels)]
          -> depthwise kxk conv (groups = expanded, stride 1 when dilation > 1) + norm + act
          -> [SqueezeExcitation(expanded, _make_divisible(expanded // 4, 8), scale = Hardsigmoid)  if use_se]
          -> 1x1 project conv + norm (no activation);
  forward: block(x), plus x when stride == 1 and input channels == output channels (residual connection).
The reference builds its with-gods trunk from this block (santorini/SantoriniNNet.py:172-178: 64 -> 192 -> 64, kernel 3, no SE,
ReLU).  For a module unpickled from `full_model` the sub-modules and flags (`block`, `use_res_connect`) come from the pickle."""
from functools import partial

import torch.nn as nn

from ..ops.misc import Conv2dNormActivation, SqueezeExcitation
from ._utils import _make_divisible


class InvertedResidualConfig:
    def __init__(self, input_channels, kernel, expanded_channels, out_channels, use_se, activation, stride, dilation, width_mult):
        self.input_channels = self.adjust_channels(input_channels, width_mult)
        self.kernel = kernel
        self.expanded_channels = self.adjust_channels(expanded_channels, width_mult)
        self.out_channels = self.adjust_channels(out_channels, width_mult)
        self.use_se = use_se
        self.use_hs = activation == 'HS'
        self.stride = stride
        self.dilation = dilation

    @staticmethod
    def adjust_channels(channels, width_mult):
        return _make_divisible(channels * width_mult, 8)


class InvertedResidual(nn.Module):
    def __init__(self, cnf, norm_layer, se_layer=partial(SqueezeExcitation, scale_activation=nn.Hardsigmoid)):
        super().__init__()
        if not 1 <= cnf.stride <= 2:
            raise ValueError('illegal stride value')
        self.use_res_connect = cnf.stride == 1 and cnf.input_channels == cnf.out_channels
        act = nn.Hardswish if cnf.use_hs else nn.ReLU
        layers = []
        if cnf.expanded_channels != cnf.input_channels:
            layers.append(Conv2dNormActivation(cnf.input_channels, cnf.expanded_channels, kernel_size=1, norm_layer=norm_layer,
                                               activation_layer=act))
        stride = 1 if cnf.dilation > 1 else cnf.stride
        layers.append(Conv2dNormActivation(cnf.expanded_channels, cnf.expanded_channels, kernel_size=cnf.kernel, stride=stride,
                                           dilation=cnf.dilation, groups=cnf.expanded_channels, norm_layer=norm_layer,
                                           activation_layer=act))
        if cnf.use_se:
            layers.append(se_layer(cnf.expanded_channels, _make_divisible(cnf.expanded_channels // 4, 8)))
        layers.append(Conv2dNormActivation(cnf.expanded_channels, cnf.out_channels, kernel_size=1, norm_layer=norm_layer,
                                           activation_layer=None))
        self.block = nn.Sequential(*layers)
        self.out_channels = cnf.out_channels
        self._is_cn = cnf.stride > 1

    def forward(self, x):
        y = self.block(x)
        if self.use_res_connect:
            y = y + x
        return y
