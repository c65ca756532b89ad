import torch.nn as nn


class InvertedResidual(nn.Module):
    pass


class InvertedResidualConfig:
    pass
