"""GPU: VideoTokenizer.decode / tokenize on the HIP engine in decoder / encoder mode, against the fixtures
frozen from the reference tokenizer and against the oracle on fresh weights at a realistic token count (64 patches + 32 latents per
frame)."""
import numpy as np
import pytest
import torch

from dreamer4_amd import DynamicsWorldModel, VideoTokenizer
from oracle import restate
from util import golden_config_kwargs, golden_model, golden_noise, load_golden, make_noise, oracle_config, randomize_weights, t

pytestmark = pytest.mark.gpu


def close(a, b, atol=1e-4, rtol=1e-4):
    a = a.detach().float().cpu() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).float()
    b = b.detach().float().cpu() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).float()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a, b, atol=atol, rtol=rtol), f'max abs diff {(a - b).abs().max().item():.3e} (scale {b.abs().max().item():.3e})'


def _tokenizer_from_fixture(weights):
    kw = golden_config_kwargs(weights)
    tok = VideoTokenizer(**kw)
    W = {k: t(v) for k, v in load_golden(weights).items() if not k.startswith(('cfg_', 'meta_'))}
    missing, unexpected = tok.load_state_dict(W, strict=False)
    enc = VideoTokenizer._ENCODER_KEYS
    if 'encode' in weights:               # the fixture holds the encoder half: everything missing belongs to the decoder
        ok = lambda k: not k.startswith(enc)
    else:
        ok = lambda k: k.startswith(enc) or 'final_special' in k or k == 'mask_token'
    assert not unexpected and all(ok(k) for k in missing), (missing, unexpected)
    return tok, restate.TokenizerConfig(**kw), W


def test_decode_vs_reference_fixture():
    g = load_golden('decode.npz')
    tok, tc, W = _tokenizer_from_fixture('weights_decode.npz')
    tok = tok.cuda()
    video, preds = tok.decode(t(g['latents']), noise=t(g['noise']), return_recons_across_steps=True)
    close(preds[0], g['pred_step0']); close(video, g['video'])
    # decoding in chunks of trajectories gives the same video (trajectories are independent)
    close(tok.decode(t(g['latents']), noise=t(g['noise']), max_batch=1), g['video'])


def _fresh(kw, seed=0):
    torch.manual_seed(seed)
    tok = VideoTokenizer(**kw)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, p in tok.named_parameters():
            if p.ndim == 1 and ('norm' in k or k.endswith('.0.weight')):
                p.copy_(1. + torch.randn(p.shape, generator=g) * 0.1)
            elif k.endswith('gamma'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    return tok


@pytest.mark.parametrize('kw', [dict(dim=64, dim_latent=16, patch_size=4, image_size=32, num_latent_tokens=32, decoder_depth=4, time_block_every=2, attn_heads=2),
                                dict(dim=32, dim_latent=8, patch_size=8, image_height=16, image_width=8, num_latent_tokens=3, decoder_depth=2, time_block_every=4,
                                     attn_heads=1, decoder_flow_steps=3, decoder_pos_mlp_depth=1, head_mlp_recipe='post_layer', channels=1)])
def test_decode_vs_oracle_on_fresh_weights(kw):
    tok = _fresh(kw)
    W = {k: v.detach().clone() for k, v in tok.state_dict().items()}
    args = {k: v for k, v in kw.items() if k != 'image_size'}
    if 'image_size' in kw:
        args['image_height'] = args['image_width'] = kw['image_size']
    tc = restate.TokenizerConfig(**args)
    B, T = 2, 4
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(B, T, kw['num_latent_tokens'], kw['dim_latent'], generator=g).clamp(-1, 1)
    noise = torch.randn(B, tc.channels, T, tc.image_height, tc.image_width, generator=g)
    ref = restate.tokenizer_decode(tc, W, lat, noise)
    out = tok.cuda().decode(lat, noise=noise)
    close(out, ref)


def test_generate_returns_the_decoded_video():
    """generate(return_decoded_video=True) (dreamer4.py:6694-6711): the imagined latents go through the tokenizer's decoder."""
    tok = _fresh(dict(dim=32, dim_latent=8, patch_size=4, image_size=16, num_latent_tokens=6, decoder_depth=2, time_block_every=2, attn_heads=2), seed=3)
    torch.manual_seed(0)
    m = randomize_weights(DynamicsWorldModel(dim=64, dim_latent=8, depth=4, time_block_every=2, attn_heads=2, num_discrete_actions=4, video_tokenizer=tok))
    assert m.num_latent_tokens == 6 and 'video_tokenizer.latents_to_decoder.weight' in m.state_dict()      # a submodule, as in the reference (dreamer4.py:4794)
    m, tok = m.cuda(), tok.cuda()
    cfg = oracle_config(m)
    nz = make_noise(cfg, 3, 2, 11)
    nz['video'] = torch.randn(2, 3, 3, 16, 16, generator=torch.Generator().manual_seed(2))
    e = m.generate(3, batch_size=2, return_for_policy_optimization=True, return_terminals=False, noise=nz)     # decoded by default with a tokenizer
    assert e.video.shape == (2, 3, 3, 16, 16)
    tc = restate.TokenizerConfig(dim=32, dim_latent=8, patch_size=4, image_height=16, image_width=16, num_latent_tokens=6, decoder_depth=2, time_block_every=2, attn_heads=2)
    ref = restate.tokenizer_decode(tc, {k: v.detach().cpu() for k, v in tok.state_dict().items()}, e.latents.cpu(), nz['video'])
    close(e.video, ref)
    vid = m.generate(2, batch_size=2, noise={**make_noise(cfg, 2, 2, 12), 'video': nz['video'][:, :, :2]})     # plain call: the video IS the result
    assert vid.shape == (2, 3, 2, 16, 16)
    assert m.generate(2, batch_size=2, return_decoded_video=False).shape == (2, 2, 6, 8)


# ------------------------------------------------------------------------------------------------ tokenize (encoder)
def test_tokenize_vs_reference_fixture():
    g = load_golden('encode.npz')
    tok, tc, W = _tokenizer_from_fixture('weights_encode.npz')
    tok = tok.cuda()
    close(tok.tokenize(t(g['video'])), g['latents'])
    close(tok.tokenize(t(g['image'])), g['image_latents'])                       # images: one-frame videos (dreamer4.py:4257)
    close(tok.tokenize(t(g['video']), max_batch=1), g['latents'])
    # fewer frames than the engine was sized for: same latents for the leading frames (causal time attention)
    close(tok.tokenize(t(g['video'])[:, :, :2]), g['latents'][:, :2])
    tokw, _, _ = _tokenizer_from_fixture('weights_encode_wide.npz')
    close(tokw.cuda().tokenize(t(g['wide_video'])), g['wide_latents'])


@pytest.mark.parametrize('kw', [dict(dim=64, dim_latent=16, patch_size=4, image_size=32, num_latent_tokens=32, encoder_depth=4, time_block_every=2, attn_heads=2),
                                dict(dim=32, dim_latent=8, patch_size=8, image_height=16, image_width=8, num_latent_tokens=3, encoder_depth=2, time_block_every=4,
                                     attn_heads=1, channels=1),
                                dict(dim=64, dim_latent=8, patch_size=4, image_height=48, image_width=48, num_latent_tokens=16, encoder_depth=2, time_block_every=2,
                                     attn_heads=2)])
def test_tokenize_vs_oracle_on_fresh_weights(kw):
    tok = _fresh(dict(kw, decoder_depth=1))
    with torch.no_grad():
        tok.latent_tokens.mul_(30.)
    W = {k: v.detach().clone() for k, v in tok.state_dict().items()}
    args = {k: v for k, v in kw.items() if k != 'image_size'}
    if 'image_size' in kw:
        args['image_height'] = args['image_width'] = kw['image_size']
    tc = restate.TokenizerConfig(**args, decoder_depth=1)
    B, T = 2, 4
    video = torch.rand(B, tc.channels, T, tc.image_height, tc.image_width, generator=torch.Generator().manual_seed(5))
    ref = restate.tokenizer_tokenize(tc, W, video)
    out = tok.cuda().tokenize(video)
    assert out.abs().max().item() < 1. and ref.std().item() > 0.05
    close(out, ref)


def test_generate_with_a_video_prompt_vs_reference_fixture():
    """generate(prompt=video) (dreamer4.py:6376-6387): the prompt frames go through the tokenizer's encoder, then are teacher-forced."""
    from test_gpu_generate import check_exp
    g = load_golden('encode.npz')
    tok, _, _ = _tokenizer_from_fixture('weights_encode.npz')
    m = golden_model('weights_encode_dyn.npz', video_tokenizer=tok)
    m, tok = m.cuda(), tok.cuda()
    e = m.generate(5, batch_size=2, prompt=t(g['video'])[:, :, :2], prompt_discrete_actions=t(g['prompt_actions_in']),
                   prompt_rewards=t(g['prompt_rewards_in']), return_for_policy_optimization=True, return_decoded_video=False,
                   noise=golden_noise(g, 'prompt_'))
    check_exp(e, g, 'prompt_')
    # a single-channel prompt is repeated over the tokenizer's channels (dreamer4.py:6381-6382)
    grey = t(g['video'])[:, :1, :2]
    a = m.generate(3, batch_size=2, prompt=grey, return_decoded_video=False, noise=golden_noise(g, 'prompt_'))
    b = m.generate(3, batch_size=2, prompt=grey.expand(-1, 3, -1, -1, -1), return_decoded_video=False, noise=golden_noise(g, 'prompt_'))
    assert torch.equal(a, b)
