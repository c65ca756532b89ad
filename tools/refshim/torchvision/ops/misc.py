"""Functional stand-ins for torchvision.ops.misc (build-container only; torchvision itself is not installed here).

Written from the published algorithm of torchvision 0.1x `ops/misc.py` -- NOT a copy of its source:
  * ConvNormActivation / Conv2dNormActivation: an nn.Sequential of Conv2d (padding = (kernel-1)//2 * dilation, bias only
    when there is no norm layer), the norm layer, the activation (inplace) -- forward is nn.Sequential's;
  * SqueezeExcitation: scale = scale_activation(fc2(activation(fc1(avgpool(x))))) with 1x1 convolutions, output scale * x.
Objects unpickled from a reference checkpoint (`full_model`) get their sub-modules and attributes from the pickle (that is
what the real torchvision stored); only the class bodies -- constructor and forward -- come from here."""
import torch
import torch.nn as nn


class ConvNormActivation(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=None, groups=1, norm_layer=nn.BatchNorm2d,
                 activation_layer=nn.ReLU, dilation=1, inplace=True, bias=None, conv_layer=nn.Conv2d):
        if padding is None:
            padding = (kernel_size - 1) // 2 * dilation
        if bias is None:
            bias = norm_layer is None
        layers = [conv_layer(in_channels, out_channels, kernel_size, stride, padding, dilation=dilation, groups=groups, bias=bias)]
        if norm_layer is not None:
            layers.append(norm_layer(out_channels))
        if activation_layer is not None:
            layers.append(activation_layer(**({} if inplace is None else {'inplace': inplace})))
        super().__init__(*layers)
        self.out_channels = out_channels


class Conv2dNormActivation(ConvNormActivation):
    pass


class SqueezeExcitation(nn.Module):
    def __init__(self, input_channels, squeeze_channels, activation=nn.ReLU, scale_activation=nn.Sigmoid):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(input_channels, squeeze_channels, 1)
        self.fc2 = nn.Conv2d(squeeze_channels, input_channels, 1)
        self.activation = activation()
        self.scale_activation = scale_activation()

    def _scale(self, x):
        return self.scale_activation(self.fc2(self.activation(self.fc1(self.avgpool(x)))))

    def forward(self, x):
        return self._scale(x) * x
