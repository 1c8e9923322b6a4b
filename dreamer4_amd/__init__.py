"""MI355X-native (gfx950) implementation of dreamer4's imagination hot path.

Python surface mirrors the reference (lucidrains/dreamer4, dreamer4/dreamer4.py):
`DynamicsWorldModel.generate` / `.learn_from_experience`, `Experience`, `Actions`.
All compute runs in hand-written HIP kernels behind the C-ABI of include/d4hip.h.
"""
from dreamer4_amd.experience import Actions, Experience, combine_experiences
from dreamer4_amd.world_model import DynamicsWorldModel, TimeCache
from dreamer4_amd.trainer import DreamTrainer
from dreamer4_amd.tokenizer import VideoTokenizer
from dreamer4_amd import ops  # noqa: F401  (registers torch.ops.d4hip.*)

__all__ = ['Actions', 'Experience', 'combine_experiences', 'DynamicsWorldModel', 'TimeCache', 'DreamTrainer', 'VideoTokenizer']
