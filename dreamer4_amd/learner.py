"""Host side of learn_from_experience (reference dreamer4.py:5893-6305): marshal the Experience into
the d4_learn C call, bridge the natively computed gradients into autograd, optional optimiser step."""
from __future__ import annotations

import ctypes as C

import torch

from dreamer4_amd import _lib
from dreamer4_amd.experience import Experience
from dreamer4_amd import parallel

_OBJECTIVES = dict(ppo=0, spo=1, pmpo=2)


class _NativeLoss(torch.autograd.Function):
    """Scalar loss whose gradient w.r.t. a head's parameters was already computed by the HIP learner."""

    @staticmethod
    def forward(ctx, loss, grad_flat, *params):
        ctx.grad_flat = grad_flat
        ctx.shapes = [p.shape for p in params]
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        out, off = [], 0
        for shp in ctx.shapes:
            n = 1
            for s in shp:
                n *= s
            out.append(ctx.grad_flat[off:off + n].view(shp) * g)
            off += n
        return (None, None, *out)


class _NativeLossFull(torch.autograd.Function):
    """The same, for learn_from_experience(only_learn_policy_value_heads=False): the loss also depends on the agent embeddings (which carry
    the world model's graph), with d loss / d agent_embed computed by the HIP learner as well."""

    @staticmethod
    def forward(ctx, loss, grad_flat, d_embed, agent_embed, *params):
        ctx.grad_flat, ctx.d_embed = grad_flat, d_embed
        ctx.shapes = [p.shape for p in params]
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        out, off = [], 0
        for shp in ctx.shapes:
            n = 1
            for s in shp:
                n *= s
            out.append(ctx.grad_flat[off:off + n].view(shp) * g)
            off += n
        return (None, None, None, ctx.d_embed * g, *out)


def agent_embed_with_grad(model, exp):
    """dreamer4.py:6045-6070 with gradient: ONE parallel forward over the experience's latents at the clean signal level, conditioned on
    the stored actions, through the differentiable HIP trunk blocks (dreamer4_amd/trunk_ops.py).  Returns agent embeddings (B, T, dim)
    connected to every world-model parameter."""
    from math import log2
    from dreamer4_amd import trunk_ops
    dev = model.device
    if exp.latents is None:
        raise ValueError('fine-tuning the world model from an Experience needs its latents')
    lat = exp.latents.to(dev).float()
    if lat.ndim == 5:
        lat = lat[:, :, 0]
    if lat.shape[1] != exp.values.shape[1]:
        raise NotImplementedError('recomputing agent embeddings for an experience with prompt frames is not implemented')
    B, T = lat.shape[:2]
    W = dict(model.named_parameters())
    W.update({k: v for k, v in model.named_buffers() if k.endswith('inv_freq')})
    is_time = [(i + 1) % model.time_block_every == 0 for i in range(model.depth)]
    sig = torch.full((B, T), model.max_steps - 1, dtype=torch.long, device=dev)
    step_log2 = torch.full((B,), int(log2(int(exp.step_size))), dtype=torch.long, device=dev)
    da = exp.actions.discrete if exp.actions is not None else None
    ca = exp.actions.continuous if exp.actions is not None else None
    if da is not None:
        da = (da[..., None] if da.ndim == 2 else da).to(dev).long()
    if ca is not None:
        ca = (ca[..., None] if ca.ndim == 2 else ca).to(dev).float()
    tasks = getattr(exp, 'tasks', None)
    _, agent = trunk_ops.world_model_prediction(
        W, lat, sig, step_log2, is_time=is_time, num_spatial_tokens=model.num_spatial_tokens, num_register_tokens=model.num_register_tokens,
        num_discrete_actions=tuple(model.num_discrete_actions), discrete_actions=da, continuous_actions=ca,
        tasks=tasks.to(dev).long() if tasks is not None else None, softclamp_value=model.attn_softclamp_value)
    return agent


def run_learner(model, experience: Experience, objective='ppo', use_delight_gating=None, delight_temperature=None,
                normalize_advantages=None, eps=1e-6, process_group=None, stats='global', want_returns=False, agent_embed=None,
                want_embed_grads=False):
    """Runs d4_learn: returns (losses[2] device tensor, returns or None) — plus (d policy_loss / d agent_embed, d value_loss / d agent_embed)
    with `want_embed_grads`.  Head gradients land in the per-group flat gradient buffers (model._groups[...]['grad']).  `agent_embed`
    overrides the experience's stored embeddings."""
    assert isinstance(experience, Experience)
    if objective not in _OBJECTIVES:
        raise ValueError(f'unknown objective {objective}')
    dev = model.device
    exp = experience.to(dev)
    assert all(v is not None for v in (exp.log_probs, exp.actions, exp.values, exp.rewards, exp.step_size)), \
        'the generations need to contain the log probs, values, and rewards for policy optimization'
    if agent_embed is not None:
        agent = agent_embed
    elif exp.agent_embed is None:
        # generate(store_agent_embed=False): recompute the agent embeddings with one parallel forward over the stored latents at
        # the clean signal level, conditioned on the stored actions (dreamer4.py:6045-6070)
        if exp.latents is None:
            raise ValueError('an Experience without agent embeddings needs its latents to recompute them')
        if exp.latents.shape[1] != exp.values.shape[1]:
            raise NotImplementedError('recomputing agent embeddings for an experience with prompt frames is not implemented')
        tasks = getattr(exp, 'tasks', None)
        with torch.no_grad():
            _, (agent, _) = model.forward(latents=exp.latents, signal_levels=model.max_steps - 1, step_sizes=exp.step_size,
                                          discrete_actions=exp.actions.discrete, continuous_actions=exp.actions.continuous, tasks=tasks)
    else:
        agent = exp.agent_embed
    agent = agent.float().contiguous()
    B, T = agent.shape[:2]
    na = len(model.num_discrete_actions)

    def bt(t, trailing=()):          # align histories that include a prompt part to the generated part
        t = t[:, -T:] if t.shape[1] > T else t
        assert t.shape[:2] == (B, T), f'experience field of shape {tuple(t.shape)} does not match agent_embed {(B, T)}'
        return t.contiguous()

    nc = model.num_continuous_actions
    actions = old_lp = actions_c = old_lp_c = None
    if na > 0:
        actions = exp.actions.discrete
        actions = actions[..., None] if actions.ndim == 2 else actions
        actions = bt(actions).long()
        old_lp = bt(exp.log_probs.discrete.float().reshape(B, exp.log_probs.discrete.shape[1], na))      # (explicit time length: -1 is ambiguous for an empty shard)
    if nc > 0:
        actions_c = exp.actions.continuous
        actions_c = actions_c[..., None] if actions_c.ndim == 2 else actions_c
        actions_c = bt(actions_c.float())
        old_lp_c = bt(exp.log_probs.continuous.float().reshape(B, exp.log_probs.continuous.shape[1], nc))
    old_values = bt(exp.values.float())
    rewards = bt(exp.rewards.float())
    old_logits = old_cparams = None
    if exp.old_action_unembeds is not None and exp.old_action_unembeds.discrete is not None and na > 0:
        old_logits = bt(exp.old_action_unembeds.discrete.float())
    if exp.old_action_unembeds is not None and exp.old_action_unembeds.continuous is not None and nc > 0:
        old_cparams = bt(exp.old_action_unembeds.continuous.float())
    prompt_off = exp.latents.shape[1] - T if exp.latents is not None else 0
    lens = exp.lens.long() - prompt_off if exp.lens is not None else torch.full((B,), T, device=dev)
    lens = lens.contiguous()
    trunc = (exp.is_truncated if exp.is_truncated is not None else torch.ones(B, dtype=torch.bool, device=dev))
    trunc = trunc.to(torch.uint8).contiguous()
    terms = exp.terminals.to(torch.uint8).contiguous() if exp.terminals is not None else None

    model._ensure_engine(learn_rows=max(B * T, 1))       # (an EMPTY trajectory shard still needs the learner's scalar workspace for the collectives it joins)
    lib = _lib.load()
    losses = torch.zeros(2, device=dev)
    returns = torch.empty(B, T, device=dev) if want_returns else None

    io = _lib.LearnIO()
    io.batch, io.time, io.objective = B, T, _OBJECTIVES[objective]
    na_flag = normalize_advantages if normalize_advantages is not None else model.normalize_advantages
    io.normalize_advantages = -1 if na_flag is None else int(bool(na_flag))
    io.eps = eps
    io.use_delight_gating = -1 if use_delight_gating is None else int(bool(use_delight_gating))
    io.delight_temperature = -1. if delight_temperature is None else float(delight_temperature)
    P = _lib.ptr
    io.agent_embed, io.actions, io.old_log_probs, io.old_values = P(agent), P(actions), P(old_lp), P(old_values)
    io.actions_cont, io.old_log_probs_cont, io.old_cont_params = P(actions_c), P(old_lp_c), P(old_cparams)
    io.rewards, io.old_action_logits, io.lens, io.is_truncated, io.terminals = P(rewards), P(old_logits), P(lens), P(trunc), P(terms)
    io.losses, io.returns = P(losses), P(returns)
    d_pol = d_val = None
    if want_embed_grads:
        d_pol, d_val = torch.zeros_like(agent), torch.zeros_like(agent)
        io.d_agent_embed_policy, io.d_agent_embed_value = P(d_pol), P(d_val)

    cb, failure = None, []
    if stats == 'global' and (parallel.world_size(process_group) > 1 or parallel.force_collectives()):
        ws, base = model._ws, model._ws.data_ptr()

        def _allreduce(ptr, n, _user):
            # ctypes swallows exceptions raised in a callback (the return value would silently become 0 and the ranks would
            # carry on with rank-local statistics): catch, remember, and make d4_learn fail with a non-zero code instead
            try:
                off = ptr - base
                view = ws[off:off + 4 * n].view(torch.float32)
                parallel.all_reduce_sum_(view, process_group)
                return 0
            except BaseException as exc:       # noqa: BLE001 - re-raised below
                failure.append(exc)
                return 1

        cb = _lib.ALLREDUCE_FN(_allreduce)
        io.allreduce_sum = cb
    rc = lib.d4_learn(model._engine, C.byref(io), model._stream())
    del cb
    if failure:
        raise failure[0]
    _lib.check(rc)
    if want_embed_grads:
        return losses, returns, (d_pol, d_val)
    return losses, returns


def learn(model, experience, policy_optim, value_optim, only_learn_policy_value_heads, objective,
          use_delight_gating, delight_temperature, normalize_advantages, eps, process_group, stats):
    if not only_learn_policy_value_heads:
        # fine-tuning the whole world model (dreamer4.py:6045-6075): agent embeddings from a forward WITH gradient through the HIP trunk
        # blocks; the learner also returns d loss / d agent_embed, which autograd carries on into the trunk.  Data parallel: the gradients of
        # EVERY parameter (trunk and heads) are summed over the ranks in one flat bucket per loss before the optimiser steps.
        agent = agent_embed_with_grad(model, experience.to(model.device))
        losses, _, (d_pol, d_val) = run_learner(model, experience, objective, use_delight_gating, delight_temperature, normalize_advantages,
                                                eps, process_group, stats, agent_embed=agent.detach(), want_embed_grads=True)
        gp, gv = model._groups['policy'], model._groups['value']           # (bound by the engine set-up inside run_learner)
        policy_loss = _NativeLossFull.apply(losses[0], gp['grad'].clone(), d_pol, agent, *gp['params'])
        value_loss = _NativeLossFull.apply(losses[1], gv['grad'].clone(), d_val, agent, *gv['params'])
        # both losses hang off ONE world-model graph built before any step (as in the reference, whose value branch reuses the agent
        # embeddings computed up front): take both gradients first — an optimiser step writes parameters the graph has saved — then step
        params = [p for p in model.parameters() if p.requires_grad]

        def grads_of(loss, retain):
            for p in params:
                p.grad = None
            loss.backward(retain_graph=retain)
            out = [p.grad for p in params]
            for p in params:
                p.grad = None
            parallel.all_reduce_grads_(out, process_group, average=stats != 'global')
            return out

        gpol = grads_of(policy_loss, value_optim is not None) if policy_optim is not None else None
        gval = grads_of(value_loss, False) if value_optim is not None else None
        for optim, grads in ((policy_optim, gpol), (value_optim, gval)):
            if optim is None:
                continue
            for p, gr in zip(params, grads):
                p.grad = gr
            optim.step()
            optim.zero_grad()
        return policy_loss, value_loss
    losses, _ = run_learner(model, experience, objective, use_delight_gating, delight_temperature,
                            normalize_advantages, eps, process_group, stats)
    gp, gv = model._groups['policy'], model._groups['value']
    policy_loss = _NativeLoss.apply(losses[0], gp['grad'].clone(), *gp['params'])
    value_loss = _NativeLoss.apply(losses[1], gv['grad'].clone(), *gv['params'])
    if policy_optim is not None:
        policy_loss.backward()
        policy_optim.step()
        policy_optim.zero_grad()
    if value_optim is not None:
        value_loss.backward()
        value_optim.step()
        value_optim.zero_grad()
    return policy_loss, value_loss
