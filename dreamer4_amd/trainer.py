"""DreamTrainer: generate -> learn_from_experience -> clip + AdamW on each head, the loop of the
reference's DreamTrainer.forward (dreamer4/trainers.py:1416-1468) without HF Accelerate.

The optimiser step runs natively (d4_adamw_clip) on the flat parameter group of each head:
`clip_grad_norm_(head, 0.5)` then AdamW(lr 3e-4, weight_decay 0) as trainers.py:1375-1376, 1436-1452.
Multi-GPU: one process per GPU, trajectories sharded by rank, one RCCL sum all-reduce per head over
its flat gradient bucket; with global statistics the local gradients are already scaled by the
global masked-mean denominators, so the sum IS the single-process gradient at B_global.
"""
from __future__ import annotations

import ctypes as C

import torch

from dreamer4_amd import _lib, parallel
from dreamer4_amd.learner import run_learner


class DreamTrainer:
    def __init__(self, model, optim_klass=None, batch_size=16, generate_timesteps=16, learning_rate=3e-4, max_grad_norm=0.5,
                 num_train_steps=10_000, weight_decay=0., objective='ppo', accelerate_kwargs: dict | None = None,
                 optim_kwargs: dict | None = None, cpu=False, use_tensorboard=False, use_wandb=False, log_dir=None,
                 project_name='dreamer4', *, betas=(0.9, 0.999), adam_eps=1e-8, process_group=None, stats='global', seed=None,
                 generate_kwargs: dict | None = None):
        """Positional / keyword order of the reference's constructor (trainers.py:1331-1349).  `optim_klass` None or
        torch.optim.AdamW: the fused native clip + AdamW step; any other optimiser class is instantiated on the two heads'
        parameters and stepped by PyTorch (with `clip_grad_norm_`) on the gradients the engine computed.  As in the
        reference the optimisers get `lr` / `weight_decay` only (its `optim_kwargs` argument is overwritten, trainers.py:1371).
        There is no CPU path and no Accelerate / tracker integration: those arguments raise instead of being ignored."""
        if cpu:
            raise NotImplementedError('DreamTrainer(cpu=True): this implementation has no CPU path')
        if use_tensorboard or use_wandb or accelerate_kwargs:
            raise NotImplementedError('experiment trackers / Accelerate options are not part of the imagination path')
        self.torch_optims = None
        if optim_klass is not None and optim_klass is not torch.optim.AdamW:
            kw = dict(lr=learning_rate, weight_decay=weight_decay)
            self.torch_optims = (optim_klass(model.policy_head_parameters(), **kw), optim_klass(model.value_head_parameters(), **kw))
        self.model, self.objective = model, objective
        self.batch_size, self.generate_timesteps = batch_size, generate_timesteps
        self.lr, self.max_grad_norm, self.weight_decay = learning_rate, max_grad_norm, weight_decay
        self.betas, self.adam_eps = betas, adam_eps
        self.num_train_steps = num_train_steps
        self.process_group, self.stats = process_group, stats
        self.generate_kwargs = dict(return_rewards_per_frame=True, return_agent_actions=True, return_log_probs_and_values=True)
        if generate_kwargs:
            self.generate_kwargs.update(generate_kwargs)
        self.step = 0
        self._state = {}
        self.generator = None
        if seed is not None:
            self.generator = torch.Generator(device=model.device).manual_seed(parallel.rank_seed(seed, process_group))

    def _optim_step(self, name):
        m = self.model
        g = m._groups[name]
        n = g['flat'].numel()
        st = self._state.get(name)
        if st is None or st['m'].numel() != n or st['m'].device != g['flat'].device:
            st = dict(m=torch.zeros(n, device=m.device), v=torch.zeros(n, device=m.device),
                      scratch=torch.zeros(1025, device=m.device), t=0)
            self._state[name] = st
        grad_scale = 1.
        if parallel.world_size(self.process_group) > 1 or parallel.force_collectives():
            parallel.all_reduce_sum_(g['grad'], self.process_group)           # ONE collective per head
            if self.stats != 'global':
                grad_scale = 1. / parallel.world_size(self.process_group)     # average of per-rank means (what DDP would do)
        st['t'] += 1
        lib = _lib.load()
        P = _lib.ptr
        _lib.check(lib.d4_adamw_clip(P(g['flat']), P(g['grad']), P(st['m']), P(st['v']), n, st['t'], self.lr, self.betas[0],
                                     self.betas[1], self.adam_eps, self.weight_decay,
                                     self.max_grad_norm if self.max_grad_norm is not None else 0., grad_scale,
                                     P(st['scratch']), m._stream()))
        return st['scratch'][0]

    def generate(self):
        return self.model.generate(self.generate_timesteps + 1, batch_size=self.batch_size, generator=self.generator,
                                   **self.generate_kwargs)

    def learn(self, dreams):
        """learn_from_experience + backward + clip + AdamW for the policy head, then the value head.
        Returns the device tensor [total_policy_loss, value_loss] (no host sync)."""
        if self.torch_optims is not None:
            return self._learn_with_torch_optims(dreams)
        losses, _ = run_learner(self.model, dreams, self.objective, process_group=self.process_group, stats=self.stats)
        self._optim_step('policy')
        self._optim_step('value')
        self.step += 1
        return losses

    def _learn_with_torch_optims(self, dreams):
        """trainers.py:1430-1452 with user-chosen optimiser classes: engine gradients, PyTorch clip + step."""
        m = self.model
        pl, vl = m.learn_from_experience(dreams, objective=self.objective, process_group=self.process_group, stats=self.stats)
        for loss, params, optim in ((pl, m.policy_head_parameters(), self.torch_optims[0]),
                                    (vl, m.value_head_parameters(), self.torch_optims[1])):
            loss.backward()
            # ONE sum all-reduce per head over a flat bucket of its gradients (as the native optimiser path), also in a one-rank world under D4_FORCE_PG=1
            parallel.all_reduce_grads_([p.grad for p in params], self.process_group, average=self.stats != 'global')
            if self.max_grad_norm is not None:
                torch.nn.utils.clip_grad_norm_(params, self.max_grad_norm)
            optim.step()
            optim.zero_grad()
        self.step += 1
        return torch.stack((pl.detach(), vl.detach()))

    def train_step(self):
        return self.learn(self.generate())

    def forward(self):
        for _ in range(self.num_train_steps):
            losses = self.train_step()
            pl, vl = losses.tolist()
            if parallel.rank(self.process_group) == 0:
                print(f'policy head loss: {pl:.3f} | value head loss: {vl:.3f}')

    __call__ = forward
