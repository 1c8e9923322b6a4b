"""Inert stubs for packages dreamer4.py imports but the imagination path never calls."""
import sys, types
from torch import nn

class _Off(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError(f'{type(self).__name__} is off the imagination path')

def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m

def install():
    def off(n): return type(n, (_Off,), {})
    tv = _mod('torchvision'); tvm = _mod('torchvision.models', VGG16_Weights = type('VGG16_Weights', (), dict(DEFAULT = None))); tv.models = tvm
    _mod('adam_atan2_pytorch', MuonAdamAtan2 = off('MuonAdamAtan2'))
    _mod('x_transformers', Decoder = off('Decoder'))
    vp = _mod('vit_pytorch')
    vp.vit_with_decorr = _mod('vit_pytorch.vit_with_decorr', DecorrelationLoss = off('DecorrelationLoss'))
    vp.vivit_with_moss = _mod('vit_pytorch.vivit_with_moss', MOSS = off('MOSS'))
    _mod('h_net_dynamic_chunking', HNet = off('HNet'))
    _mod('PoPE_pytorch', PoPE = off('PoPE'), AxialPoPE = off('AxialPoPE'), flash_attn_with_pope = None)
    _mod('memmap_replay_buffer', ReplayBuffer = off('ReplayBuffer'))
