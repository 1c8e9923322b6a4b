"""Synthetic (random-init) weights for benchmarks and tests: there is no network for checkpoints."""
import torch


@torch.no_grad()
def randomize_weights(m, seed=0, unembed_scale=30., terminal_bias=-2.5):
    """Default init leaves logits/values ~0 (unembed is randn*1e-2, dreamer4.py:1226): give every
    norm / gamma / learned token / head a non-trivial value so each code path is numerically visible."""
    g = torch.Generator().manual_seed(seed)
    post_ln = getattr(m, 'head_mlp_recipe', 'pre_rms') == 'post_layer'
    for name, p in m.named_parameters():
        if p.numel() == 0 or name.startswith('video_tokenizer.'):          # a nested tokenizer keeps the weights it came with
            continue
        if name.endswith('gamma'):
            p.copy_(torch.randn(p.shape, generator=g) * 0.2)
        elif p.ndim == 1 and ('norm' in name or name.endswith('.0.weight') or (post_ln and name.endswith('.1.weight'))):
            p.copy_(1. + torch.randn(p.shape, generator=g) * 0.1)
        elif p.ndim == 1 and post_ln and name.endswith('.1.bias') and name.split('.')[0] in ('policy_head', 'value_head', 'to_state_terminal_pred'):
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)          # LayerNorm biases of the head MLPs
        elif name in ('register_tokens', 'agent_learned_embed', 'action_learned_embed') or name.endswith('queries'):
            p.copy_(torch.randn(p.shape, generator=g) * 0.5)
        elif name == 'to_reward_pred.params.0':
            p.copy_(1. + torch.randn(p.shape, generator=g) * 0.1)
        elif name == 'to_reward_pred.params.1':
            p.mul_(5.)
        elif name == 'action_embedder.discrete_action_unembed':
            p.mul_(unembed_scale)
    if m.predict_terminals:
        m.head_mlp_output_linear('to_state_terminal_pred.0')[1].fill_(terminal_bias)
    return m
