import torch.nn.functional as F
from torch import nn

class ReluSquared(nn.Module):
    def forward(self, x): return F.relu(x) ** 2

class SugarBSiLU(nn.Module):
    def forward(self, x): raise NotImplementedError('off the imagination path')
