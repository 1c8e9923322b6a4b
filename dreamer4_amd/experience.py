"""`Experience` / `Actions`: the hand-off object between generate() and learn_from_experience().

Field-for-field the reference dataclass (dreamer4.py:132-154, 240-246), its `combine_experiences`
(dreamer4.py:248-309) and the flat-dictionary form the replay buffer stores (`to_buffer_dict` /
`from_buffer_dict`, dreamer4.py:172-186, 218-236).  The buffer object itself is the third-party
`memmap_replay_buffer.ReplayBuffer` (not in this image): `create_memmap_replay_buffer` /
`add_to_memmap_buffer` import it lazily, exactly as the reference does, and raise ImportError without it.
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

    _meta_fields = frozenset({'step_size', 'lens', 'is_truncated', 'terminals', 'agent_index', 'is_from_world_model', 'episode_return'})

    @property
    def payload(self):
        for v in (self.latents, self.video, self.critic_state):
            if v is not None:
                return v

    # ---- memmap replay buffer format (dreamer4.py:172-236): per-step data fields and per-episode meta fields, `Actions` flattened
    # to `<field>_discrete` / `<field>_continuous`
    def to_buffer_dict(self):
        data_dict, meta_dict = {}, {}
        for f in fields(self):
            k, v = f.name, getattr(self, f.name)
            target = meta_dict if k in self._meta_fields else data_dict
            if isinstance(v, Actions):
                if v.discrete is not None:
                    target[f'{k}_discrete'] = v.discrete
                if v.continuous is not None:
                    target[f'{k}_continuous'] = v.continuous
            elif v is not None:
                target[k] = v
        return data_dict, meta_dict

    @classmethod
    def from_buffer_dict(cls, data_dict):
        kwargs = {}
        for f in fields(cls):
            k = f.name
            dk, ck = f'{k}_discrete', f'{k}_continuous'
            if dk in data_dict or ck in data_dict:
                kwargs[k] = Actions(data_dict.get(dk), data_dict.get(ck))
            elif k in data_dict:
                kwargs[k] = data_dict[k]
        return cls(**kwargs)

    # ---- memmap replay buffer (third-party `memmap_replay_buffer.ReplayBuffer`, dreamer4.py:186-216).  `buffer_cls` lets a caller (or a
    # test) supply the buffer class; by default the third-party package is imported, as the reference does.
    @staticmethod
    def _field_spec(value, leading_dims):
        """Field declaration the buffer expects: a python type name for plain values, (dtype name, per-item shape) for tensors; dtype
        names follow the reference's rule (dreamer4.py: tensor_dtype_to_string): 'bool', 'float' for EVERY floating dtype, else 'int'."""
        if not torch.is_tensor(value):
            return type(value).__name__
        name = 'bool' if value.dtype == torch.bool else ('float' if value.is_floating_point() else 'int')
        return name, tuple(value.shape[leading_dims:])

    @classmethod
    def create_memmap_replay_buffer(cls, template_experience, *args, buffer_cls=None, **kwargs):
        if buffer_cls is None:
            from memmap_replay_buffer import ReplayBuffer as buffer_cls
        per_step, per_episode = template_experience.to_buffer_dict()
        return buffer_cls(*args, fields={k: cls._field_spec(v, 2) for k, v in per_step.items()},          # (batch, time, ...) -> item shape
                          meta_fields={k: cls._field_spec(v, 1) for k, v in per_episode.items()}, **kwargs)   # (batch, ...) -> item shape

    def add_to_memmap_buffer(self, buffer):
        """One batched episode per call: per-episode values as keyword arguments of `batched_episode` (plain values repeated per
        trajectory), then one `store_batch` per time step with the step's slice of every per-step field."""
        per_step, per_episode = self.to_buffer_dict()
        batch = self.payload.shape[0]
        episode_kw = {k: (v if torch.is_tensor(v) else [v] * batch) for k, v in per_episode.items()}
        steps = self.payload.shape[1]                      # the payload's time length drives the loop (dreamer4.py:205-213): a shorter field raises
        with buffer.batched_episode(batch_size=batch, **episode_kw):
            for step in range(steps):
                buffer.store_batch(**{k: v[:, step] for k, v in per_step.items()})

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
