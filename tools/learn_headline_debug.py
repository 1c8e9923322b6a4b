import sys, os
ROOT='/root/repo' if os.path.isdir('/root/repo/tests') else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import torch
from dreamer4_amd import DynamicsWorldModel, DreamTrainer
from dreamer4_amd.learner import run_learner
from util import randomize_weights, oracle_config, oracle_weights, make_noise
from oracle import restate
HEADS = ('policy_head', 'value_head', 'action_embedder.discrete_action_unembed')
CFG2_ARCH = dict(dim=512, dim_latent=32, num_latent_tokens=32, depth=6, attn_heads=8, attn_dim_head=64, num_spatial_tokens=4,
                 num_register_tokens=8, max_steps=64, multi_token_pred_len=8, num_discrete_actions=4)
torch.set_num_threads(16)
for B in [int(a) for a in sys.argv[1:]] or [16, 256]:
    torch.manual_seed(0)
    m = randomize_weights(DynamicsWorldModel(**CFG2_ARCH), seed=0, terminal_bias=-10.)
    cfg, W = oracle_config(m), oracle_weights(m)
    T = 16
    m = m.cuda()
    nz = make_noise(cfg, T, B, 1234)
    e = m.generate(T, batch_size=B, return_for_policy_optimization=True, num_steps=4, noise=nz)
    cpu = lambda x: x.detach().cpu()
    ref = dict(latents=cpu(e.latents), agent_embed=cpu(e.agent_embed), rewards=cpu(e.rewards), values=cpu(e.values), log_probs=cpu(e.log_probs.discrete),
               actions=cpu(e.actions.discrete), lens=cpu(e.lens), terminals=cpu(e.terminals), is_truncated=cpu(e.is_truncated),
               old_action_unembeds=cpu(e.old_action_unembeds.discrete), step_size=e.step_size)
    Wg = {k: (v.clone().requires_grad_() if k.startswith(HEADS) else v) for k, v in W.items()}
    pl_o, vl_o = restate.learn_losses(cfg, Wg, ref, 'ppo')
    pl_o.backward(); vl_o.backward()
    Wd = {k: (v.double().clone().requires_grad_() if k.startswith(HEADS) else (v.double() if v.is_floating_point() else v)) for k, v in W.items()}
    refd = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in ref.items()}
    try:
        pl_d, vl_d = restate.learn_losses(cfg, Wd, refd, 'ppo')
        pl_d.backward(); vl_d.backward()
    except Exception as ex:
        print('float64 oracle failed:', ex); Wd = None
    losses, _ = run_learner(m, e, 'ppo')
    print(f'B={B}: losses gpu {losses.tolist()} oracle {pl_o.item()} {vl_o.item()}', 'f64', (pl_d.item(), vl_d.item()) if Wd else None)
    # recomputed log-probs vs stored
    pe = restate.policy_head(cfg, W, ref['agent_embed'])
    logits = restate.policy_logits(cfg, W, pe)
    lp = restate.discrete_log_probs(cfg, logits, ref['actions'])
    print('  oracle lp - stored lp: max', (lp - ref['log_probs']).abs().max().item(), ' logits max', logits.abs().max().item(), 'lp min', lp.min().item())
    print('  oracle logits - stored unembeds max', (logits - ref['old_action_unembeds']).abs().max().item())
    names = {id(p): k for k, p in m.named_parameters()}
    for head in ('policy', 'value'):
        grp = m._groups[head]
        g = grp['grad'].detach().cpu()
        off = 0
        for p in grp['params']:
            k = names[id(p)]
            gg = g[off:off + p.numel()].view(p.shape); off += p.numel()
            go = Wg[k].grad
            s = f'  {k:55s} |g| {go.norm().item():.4e} gpu-vs-f32oracle rel {((gg-go).norm()/go.norm()).item():.2e}'
            if Wd:
                gd = Wd[k].grad
                s += f'  gpu-vs-f64 {((gg.double()-gd).norm()/gd.norm()).item():.2e}  f32oracle-vs-f64 {((go.double()-gd).norm()/gd.norm()).item():.2e}'
            print(s)
