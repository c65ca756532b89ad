import torch.nn as nn


class ConvNormActivation(nn.Sequential):
    pass


class Conv2dNormActivation(ConvNormActivation):
    pass


class SqueezeExcitation(nn.Module):
    pass
