"""`Experience` / `Actions`: the hand-off object between generate() and learn_from_experience().

Field-for-field the reference dataclass (dreamer4.py:132-154, 240-246) and its
`combine_experiences` (dreamer4.py:248-309); the replay-buffer adapters
(dreamer4.py:172-236) depend on the third-party memmap_replay_buffer and are out of scope.
"""
from __future__ import annotations

from collections import namedtuple
from dataclasses import dataclass, fields

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.utils._pytree import tree_flatten, tree_map, tree_unflatten

Actions = namedtuple('Actions', ['discrete', 'continuous'])

MaybeTensor = Tensor | None


@dataclass
class Experience:
    latents: Tensor
    video: MaybeTensor = None
    proprio: MaybeTensor = None
    critic_state: MaybeTensor = None
    agent_embed: MaybeTensor = None
    rewards: Tensor | None = None
    terminals: Tensor | None = None
    actions: Actions | None = None
    log_probs: Actions | None = None
    old_action_unembeds: Actions | None = None
    values: MaybeTensor = None
    step_size: int | None = None
    lens: MaybeTensor = None
    is_truncated: MaybeTensor = None
    agent_index: int = 0
    is_from_world_model: bool | Tensor = True
    episode_return: Tensor | None = None

    @property
    def payload(self):
        for v in (self.latents, self.video, self.critic_state):
            if v is not None:
                return v

    def cpu(self):
        return self.to(torch.device('cpu'))

    def to(self, device):
        d = {f.name: getattr(self, f.name) for f in fields(self)}
        d = tree_map(lambda t: t.to(device) if torch.is_tensor(t) else t, d)
        return Experience(**d)


def _pad_right_to(t, length, dim):
    cur = t.shape[dim]
    if cur >= length:
        return t
    pad = [0, 0] * (t.ndim - dim - 1) + [0, length - cur]
    return F.pad(t, pad)


def combine_experiences(exps: list[Experience]) -> Experience:
    """Padding-concat of variable-length experience batches (dreamer4.py:248-309)."""
    assert len(exps) > 0
    for exp in exps:
        payload = exp.latents if exp.latents is not None else exp.video
        batch, time = payload.shape[:2]
        device = payload.device
        if exp.lens is None:
            exp.lens = torch.full((batch,), time, device=device)
        if exp.is_truncated is None:
            exp.is_truncated = torch.full((batch,), True, device=device)
        if isinstance(exp.is_from_world_model, bool):
            exp.is_from_world_model = torch.full((batch,), exp.is_from_world_model, device=device, dtype=torch.bool)

    dicts = [{f.name: getattr(e, f.name) for f in fields(e)} for e in exps]
    values, specs = zip(*[tree_flatten(d) for d in dicts])
    spec = specs[0]
    columns = list(zip(*values))
    assert all(all(torch.is_tensor(v) for v in col) or len(set(col)) == 1 for col in columns)
    out = []
    for col in columns:
        first = col[0]
        if torch.is_tensor(first):
            col = list(col)
            for dim in (1, 2):
                if dim >= first.ndim:
                    continue
                m = max(t.shape[dim] for t in col)
                col = [_pad_right_to(t, m, dim) for t in col]
            out.append(torch.cat(col) if first.ndim > 0 else torch.stack(col))
        else:
            out.append(first)
    return Experience(**tree_unflatten(out, spec))
