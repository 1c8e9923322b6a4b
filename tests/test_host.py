"""CPU: host-side logic of the product package (no kernels run here)."""
import os
import subprocess
import sys
import textwrap

import pytest
import torch

import dreamer4_amd
from dreamer4_amd import Actions, DynamicsWorldModel, Experience, combine_experiences, parallel
from dreamer4_amd._lib import D4Error
from util import golden_model, golden_oracle, small_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_keys_match_the_reference_fixture():
    m = golden_model()
    _, W = golden_oracle()
    mine = {k for k, p in m.state_dict().items() if p.numel() > 0}
    ref = set(W)
    # keys of modules that are off the imagination path in the reference's state_dict are allowed to be absent here
    assert not (mine - ref - {'reward_learned_embed'}), sorted(mine - ref)
    on_path = {k for k in ref if k.startswith(('transformer.', 'policy_head', 'value_head', 'to_', 'latents_to', 'action_embedder',
                                                'register_tokens', 'signal', 'step_size', 'agent_learned', 'action_learned', 'task_embed'))}
    assert not (on_path - mine), sorted(on_path - mine)


def test_policy_and_value_parameter_groups():
    m = small_model()
    names = {id(p): k for k, p in m.named_parameters()}
    pol = {names[id(p)] for p in m.policy_head_parameters()}
    assert 'action_embedder.discrete_action_unembed' in pol and all(k.startswith(('policy_head', 'action_embedder')) for k in pol)
    assert all(names[id(p)].startswith('value_head') for p in m.value_head_parameters())


@pytest.mark.parametrize('kw', [dict(num_agents=2), dict(dim_proprio=4), dict(use_time_rnn=True), dict(actor_depth=1),
                                dict(num_continuous_actions=2, continuous_dist_type='gaussian'), dict(continuous_norm_stats=((0., 1.),)), dict(add_state_pred_head=True), dict(mot_temporal=True)])
def test_out_of_scope_options_raise_instead_of_being_ignored(kw):
    with pytest.raises(NotImplementedError):
        DynamicsWorldModel(dim=64, dim_latent=8, num_latent_tokens=6, num_discrete_actions=4, **kw)
    with pytest.raises(TypeError):
        DynamicsWorldModel(dim=64, dim_latent=8, num_latent_tokens=6, not_an_option=1)


def test_no_cpu_fallback():
    m = small_model()
    with pytest.raises(D4Error, match='no CPU fallback'):
        m.generate(2, batch_size=1, return_for_policy_optimization=True)


def test_experience_to_and_combine():
    a = Experience(latents=torch.zeros(2, 3, 4, 5), rewards=torch.ones(2, 3), actions=Actions(torch.zeros(2, 3, 1, dtype=torch.long), None),
                   lens=torch.tensor([3, 2]), is_truncated=torch.tensor([True, False]), step_size=16)
    b = Experience(latents=torch.zeros(1, 5, 4, 5), rewards=torch.ones(1, 5), actions=Actions(torch.zeros(1, 5, 1, dtype=torch.long), None),
                   step_size=16)
    c = combine_experiences([a, b])
    assert c.latents.shape == (3, 5, 4, 5) and c.rewards.shape == (3, 5) and c.actions.discrete.shape == (3, 5, 1)
    assert c.lens.tolist() == [3, 2, 5] and c.is_truncated.tolist() == [True, False, True] and c.step_size == 16
    assert c.to('cpu').latents.device.type == 'cpu'


def test_shard_range_covers_the_global_batch():
    assert parallel.world_size() == 1 and parallel.rank() == 0
    assert parallel.shard_range(10) == (0, 10)


DP_SCRIPT = textwrap.dedent('''
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests'))
    from dreamer4_amd import parallel
    from oracle import restate
    from util import golden_oracle, load_golden, t
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    parallel.init_from_env('gloo')
    r, ws = parallel.rank(), parallel.world_size()
    assert ws == 2
    # shard ranges tile the global batch, seeds differ per rank
    lo, hi = parallel.shard_range(5)
    spans = [None, None]; dist.all_gather_object(spans, (lo, hi))
    assert spans == [(0, 3), (3, 5)], spans
    assert parallel.rank_seed(1234) == 1234 + r
    # global statistics: per-rank partial sums + all-reduce == single process over the whole batch (SURVEY.md 8e)
    cfg, W = golden_oracle(); G = load_golden('generate.npz')
    B = 3
    lo, hi = parallel.shard_range(B)
    full = dict(rewards=t(G['cached_rewards']), values=t(G['cached_values']), lens=t(G['cached_lens']),
                terminals=t(G['cached_terminals']), is_truncated=~t(G['cached_terminals']))
    _, _, adv_full, mask_full = restate.returns_and_advantage(cfg, full, normalize=True)
    local = {{k: v[lo:hi] for k, v in full.items()}}
    _, _, adv_raw, mask = restate.returns_and_advantage(cfg, local, normalize=False)
    s = torch.stack([(adv_raw * mask).sum(), mask.float().sum()]); parallel.all_reduce_sum_(s)
    mean = s[0] / s[1]
    sq = (((adv_raw - mean) ** 2) * mask).sum().reshape(1); parallel.all_reduce_sum_(sq)
    adv = (adv_raw - mean) / (sq[0] / s[1]).clamp(min=1e-6).sqrt()
    assert torch.allclose(adv, adv_full[lo:hi], atol=1e-5), (adv - adv_full[lo:hi]).abs().max()
    # one flat bucket per head: sum all-reduce of per-rank gradients scaled by the GLOBAL count == full-batch gradient
    g = torch.full((7,), float(r + 1)); parallel.all_reduce_sum_(g); assert g.tolist() == [3.0] * 7
    m = torch.tensor([float(r)]); parallel.all_reduce_max_(m); assert m.item() == 1.0
    # the whole-model form (learn_from_experience(only_learn_policy_value_heads=False) / torch optimisers): every parameter gradient,
    # trunk included, through ONE flat bucket; None gradients are skipped, shapes survive, `average` divides by the world size
    grads = [torch.full((2, 3), float(r + 1)), None, torch.arange(4.) * (r + 1)]
    parallel.all_reduce_grads_(grads); assert grads[0].tolist() == [[3.0] * 3] * 2 and grads[1] is None and grads[2].tolist() == [0., 3., 6., 9.]
    grads = [torch.full((5,), float(r + 1))]; parallel.all_reduce_grads_(grads, average=True); assert grads[0].tolist() == [1.5] * 5
    assert parallel.collective_active()
    parallel.barrier()
    os.write(1, ('DP_OK_' + str(r) + '|').encode())
''')


def test_two_process_gloo_data_parallel_semantics(tmp_path):
    script = tmp_path / 'dp.py'
    script.write_text(DP_SCRIPT.format(root=ROOT))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', '29653', str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count('DP_OK_') == 2, out.stdout


def test_custom_ops_are_registered_with_the_dispatcher_and_traceable():
    """the single-kernel entry points are torch.library ops (fake implementations let torch.compile trace them);
    on CPU tensors they raise instead of falling back."""
    import dreamer4_amd  # noqa: F401
    for name in ('rmsnorm', 'linear', 'hl_gauss_to_scalar', 'gae', 'rmsnorm_backward', 'linear_backward', 'swiglu_ff', 'swiglu_ff_backward',
                 'attn_block_space', 'attn_block_space_backward', 'attn_block_time', 'attn_block_time_backward', 'attn_block_cross',
                 'attn_block_cross_backward', 'flow_euler_step', 'ppo_policy_loss', 'hl_gauss_ce', 'categorical_sample_logp'):
        assert hasattr(torch.ops.d4hip, name), name
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(3, 5, 64); w = torch.empty(128, 64)
        y = torch.ops.d4hip.linear(x, w, None, None, 4, 1e-6)           # SiLU-GLU pairs halve the width
        assert y.shape == (3, 5, 64)
        # the trunk blocks (FeedForward, attention within a frame / over a context): output + the forward workspace
        n, wi, bi, wo, bo = torch.empty(64), torch.empty(340, 64), torch.empty(340), torch.empty(64, 170), torch.empty(64)
        y, ws = torch.ops.d4hip.swiglu_ff(x, n, wi, bi, wo, bo)
        assert y.shape == x.shape and ws.dtype == torch.uint8
        g = torch.ops.d4hip.swiglu_ff_backward(x, y, n, wi, bi, wo, ws)
        assert [t.shape for t in g] == [x.shape, n.shape, wi.shape, bi.shape, wo.shape, bo.shape]
        hq, gam = torch.empty(128, 64), torch.empty(2, 64)
        y, ws = torch.ops.d4hip.attn_block_space(x, None, n, hq, hq, hq, torch.empty(64, 128), torch.empty(2, 64), None, None, gam, 50., 1, True)
        assert y.shape == x.shape
        g = torch.ops.d4hip.attn_block_space_backward(x, None, y, n, hq, hq, hq, torch.empty(64, 128), torch.empty(2, 64), None, None, gam, 50., 1, True, ws)
        assert g[0].shape == x.shape and g[1].numel() == 0 and g[10].shape == gam.shape
        assert torch.ops.d4hip.flow_euler_step(x, x, 0.5, 0.25).shape == x.shape
        assert torch.ops.d4hip.rmsnorm(x, torch.empty(64), 1e-6).shape == x.shape
        assert torch.ops.d4hip.hl_gauss_to_scalar(torch.empty(7, 255), torch.empty(255)).shape == (7,)
    with pytest.raises(D4Error, match='no CPU fallback'):
        torch.ops.d4hip.rmsnorm(torch.zeros(2, 8), torch.ones(8), 1e-6)


def test_experience_replay_buffer_dictionaries_round_trip():
    """dreamer4.py:172-186, 218-236: `Actions` are flattened to <field>_discrete / <field>_continuous, per-episode fields go to the meta dict."""
    e = Experience(latents=torch.zeros(2, 3, 4, 5), rewards=torch.ones(2, 3), actions=Actions(torch.zeros(2, 3, 1, dtype=torch.long), torch.rand(2, 3, 2)),
                   log_probs=Actions(torch.zeros(2, 3, 1), None), lens=torch.tensor([3, 2]), is_truncated=torch.tensor([True, False]), step_size=16)
    data, meta = e.to_buffer_dict()
    assert set(data) == {'latents', 'rewards', 'actions_discrete', 'actions_continuous', 'log_probs_discrete'}
    assert set(meta) == {'step_size', 'lens', 'is_truncated', 'agent_index', 'is_from_world_model'}
    back = Experience.from_buffer_dict({**data, **meta})
    assert torch.equal(back.actions.continuous, e.actions.continuous) and back.log_probs.continuous is None and back.step_size == 16
    with pytest.raises(ImportError):                       # default: the third-party buffer, as the reference (absent from the image)
        Experience.create_memmap_replay_buffer(e, './x', max_episodes=1, max_timesteps=4)


class _StubReplayBuffer:
    """The call pattern dreamer4.py:186-216 makes against memmap_replay_buffer.ReplayBuffer, recorded: constructor (fields=,
    meta_fields=), `with buffer.batched_episode(batch_size=, **meta)`, one `store_batch(**fields)` per time step."""

    def __init__(self, folder, max_episodes, max_timesteps, fields, meta_fields):
        self.fields, self.meta_fields, self.max_timesteps = fields, meta_fields, max_timesteps
        self.episodes, self._open = [], None

    def batched_episode(self, batch_size, **meta):
        import contextlib

        @contextlib.contextmanager
        def cm():
            self._open = dict(batch_size=batch_size, meta=meta, steps=[])
            yield
            self.episodes.append(self._open); self._open = None
        return cm()

    def store_batch(self, **step):
        assert self._open is not None and set(step) == set(self.fields)
        for k, v in step.items():
            kind, shape = self.fields[k]
            assert tuple(v.shape) == (self._open['batch_size'], *shape), k
            assert kind == ('bool' if v.dtype == torch.bool else 'float' if v.is_floating_point() else 'int')
        self._open['steps'].append(step)


def test_experience_memmap_buffer_call_pattern_with_a_stub_buffer():
    """create_memmap_replay_buffer / add_to_memmap_buffer against a stub with the reference's call pattern (dreamer4.py:186-216): field
    declarations use the reference's dtype rule ('bool' / 'float' for every floating dtype / 'int' for everything else — uint8 terminals
    and int32 lens included), per-step fields drop (batch, time), per-episode fields drop (batch), and the stored steps restore the tensors."""
    B, T = 2, 3
    e = Experience(latents=torch.randn(B, T, 4, 5).half(), rewards=torch.ones(B, T), terminals=torch.zeros(B, dtype=torch.uint8),
                   actions=Actions(torch.arange(B * T).reshape(B, T, 1), torch.rand(B, T, 2)), lens=torch.tensor([3, 2], dtype=torch.int32),
                   is_truncated=torch.tensor([True, False]), step_size=16)
    buf = Experience.create_memmap_replay_buffer(e, './x', max_episodes=4, max_timesteps=8, buffer_cls=_StubReplayBuffer)
    assert buf.fields == {'latents': ('float', (4, 5)), 'rewards': ('float', ()), 'actions_discrete': ('int', (1,)), 'actions_continuous': ('float', (2,))}
    assert buf.meta_fields['terminals'] == ('int', ())            # a per-episode (meta) field of the reference, uint8 here -> 'int'
    assert buf.meta_fields['lens'] == ('int', ()) and buf.meta_fields['is_truncated'] == ('bool', ()) and buf.meta_fields['step_size'] == 'int'
    e.add_to_memmap_buffer(buf)
    (ep,) = buf.episodes
    assert ep['batch_size'] == B and len(ep['steps']) == T and ep['meta']['step_size'] == [16, 16] and torch.equal(ep['meta']['lens'], e.lens)
    stacked = {k: torch.stack([s[k] for s in ep['steps']], 1) for k in buf.fields}
    back = Experience.from_buffer_dict({**stacked, **{k: v for k, v in ep['meta'].items() if torch.is_tensor(v)}, 'step_size': 16})
    assert torch.equal(back.latents, e.latents) and torch.equal(back.actions.discrete, e.actions.discrete) and torch.equal(back.terminals, e.terminals)


def test_save_load_and_init_and_load(tmp_path):
    """@save_load surface of the reference (dreamer4.py:4660; trainers.py:1083-1088): save / load / init_and_load rebuild the model from the file."""
    from dreamer4_amd import VideoTokenizer
    m = small_model(num_continuous_actions=2, head_mlp_recipe='post_layer')
    p = tmp_path / 'dynamics.pt'
    m.save(p)
    m2 = DynamicsWorldModel.init_and_load(p)
    assert m2.num_continuous_actions == 2 and m2.head_mlp_recipe == 'post_layer' and m2.depth == m.depth
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    m3 = small_model(num_continuous_actions=2, head_mlp_recipe='post_layer')
    with torch.no_grad():
        m3.register_tokens.zero_()
    m3.load(p)
    assert torch.equal(m3.register_tokens, m.register_tokens)
    tok = VideoTokenizer(dim=32, dim_latent=8, patch_size=4, image_size=16, num_latent_tokens=6, decoder_depth=2)
    tok.save(tmp_path / 'tok.pt')
    tok2 = VideoTokenizer.init_and_load(tmp_path / 'tok.pt')
    assert tok2.image_height == 16 and torch.equal(tok2.state_dict()['latents_to_decoder.weight'], tok.state_dict()['latents_to_decoder.weight'])


def test_checkpoint_files_laid_out_as_the_reference_trainer_writes_them(tmp_path):
    """trainers.py:1074-1088 is the only in-tree evidence of the file layout: dict(model=state_dict, config=pickle.dumps(dehydrated
    `_config`) or None, step=int), `_config` a live object on the module.  Files written that way — by this package's `save`, by hand
    with a plain (args, kwargs) config, with config=None, and as a bare state_dict (trainers.py:1048-1072 accepts `pkg.get('model', pkg)`)
    — all load; a nested video tokenizer survives the round trip as a rebuilt module; an unknown config structure fails with a message."""
    import pickle
    from dreamer4_amd import VideoTokenizer
    tok = VideoTokenizer(dim=32, dim_latent=8, patch_size=4, image_size=16, num_latent_tokens=4, decoder_depth=2)
    m = small_model(dim_latent=8, num_latent_tokens=4, video_tokenizer=tok, latent_flow_loss_weight=0.5, multi_token_pred_len=2, reward_loss_weight=[1., 0.25])
    assert isinstance(m._config, tuple) and m._config[1]['video_tokenizer'] is tok and '_config' not in m.state_dict()
    m.save(tmp_path / 'a.pt', step=7)
    pkg = torch.load(tmp_path / 'a.pt', weights_only=False)
    assert set(pkg) >= {'model', 'config', 'step'} and pkg['step'] == 7 and isinstance(pkg['config'], bytes)
    args, kwargs = pickle.loads(pkg['config'])                       # unpickles without this package's classes: plain containers only
    assert args == () and kwargs['video_tokenizer']['__d4_module__'] == 'VideoTokenizer'
    # the tokenizer is a frozen deep copy registered as a submodule (dreamer4.py:4787-4794): its weights travel in the state_dict
    assert m.video_tokenizer is not tok and not any(p.requires_grad for p in m.video_tokenizer.parameters()) and not m.video_tokenizer.training
    tok_keys = [k for k in pkg['model'] if k.startswith('video_tokenizer.')]
    assert len(tok_keys) == len(tok.state_dict()) > 0
    with torch.no_grad():
        for p in m.video_tokenizer.parameters():
            p.add_(0.37)                                                 # not the constructor's init any more
    m.save(tmp_path / 'a.pt', step=7)
    m2 = DynamicsWorldModel.init_and_load(tmp_path / 'a.pt')
    assert isinstance(m2.video_tokenizer, VideoTokenizer) and m2.video_tokenizer.image_height == 16
    for k, v in m.video_tokenizer.state_dict().items():
        assert torch.equal(m2.video_tokenizer.state_dict()[k], v), k
    assert not torch.equal(m2.video_tokenizer.state_dict()['latents_to_decoder.weight'], tok.state_dict()['latents_to_decoder.weight'])
    assert m2.latent_flow_loss_weight == 0.5 and torch.equal(m2.reward_loss_weight, torch.tensor([1., 0.25]))
    # by hand, the way save_checkpoint does it, with an already-plain config and a step
    plain = dict(dim=32, dim_latent=8, num_latent_tokens=4, depth=2, num_discrete_actions=4, multi_token_pred_len=2)
    src = DynamicsWorldModel(**plain)
    torch.save(dict(model=src.state_dict(), config=pickle.dumps(((), plain)), step=3), tmp_path / 'b.pt')
    assert DynamicsWorldModel.init_and_load(tmp_path / 'b.pt', strict=False).dim == 32
    torch.save(dict(model=src.state_dict(), config=pickle.dumps(dict(args=(), kwargs=plain)), step=3), tmp_path / 'b2.pt')
    assert DynamicsWorldModel.init_and_load(tmp_path / 'b2.pt').depth == 2
    # config = None (a model without `_config`): needs the constructor arguments, or .load on a built model
    torch.save(dict(model=src.state_dict(), config=None, step=3), tmp_path / 'c.pt')
    with pytest.raises(TypeError, match='without a config'):
        DynamicsWorldModel.init_and_load(tmp_path / 'c.pt')
    assert DynamicsWorldModel.init_and_load(tmp_path / 'c.pt', **plain).depth == 2
    DynamicsWorldModel(**plain).load(tmp_path / 'c.pt')
    torch.save(src.state_dict(), tmp_path / 'd.pt')                     # bare state_dict
    DynamicsWorldModel(**plain).load(tmp_path / 'd.pt')
    torch.save(dict(model=src.state_dict(), config=pickle.dumps(42)), tmp_path / 'e.pt')
    with pytest.raises(TypeError, match='unknown structure'):
        DynamicsWorldModel.init_and_load(tmp_path / 'e.pt')
    # a reference checkpoint trained with loss normalisation carries BOTH action normalizers (dreamer4.py:5254-5255) and loss-weight buffers
    ref_like = DynamicsWorldModel(**plain, use_loss_normalization=True).state_dict()
    assert {'discrete_actions_loss_normalizer.exp_avg_sq', 'continuous_actions_loss_normalizer.exp_avg_sq', 'reward_loss_weight',
            'terminal_loss_weight', 'discrete_action_loss_weight', 'continuous_action_loss_weight'} <= set(ref_like)


def test_loss_normalizer_state_and_beta_zero_property():
    """The training forward's LossNormalizer (reference tests/test_dreamer.py:558-569): with beta = 0 the second call of the same loss
    returns exactly 1; buffers carry the reference's state_dict keys only when `use_loss_normalization=True`."""
    import torch
    from dreamer4_amd import DynamicsWorldModel
    kw = dict(dim=32, dim_latent=8, num_latent_tokens=4, depth=2, num_discrete_actions=4, multi_token_pred_len=2)
    plain = DynamicsWorldModel(**kw)
    assert not any('loss_normalizer' in k for k in plain.state_dict())
    m = DynamicsWorldModel(**kw, use_loss_normalization=True)
    keys = {k for k in m.state_dict() if 'loss_normalizer' in k}
    assert keys == {'flow_loss_normalizer.exp_avg_sq', 'shortcut_flow_loss_normalizer.exp_avg_sq', 'reward_loss_normalizer.exp_avg_sq',
                    'state_terminal_loss_normalizer.exp_avg_sq', 'discrete_actions_loss_normalizer.exp_avg_sq',
                    'continuous_actions_loss_normalizer.exp_avg_sq'}          # both action normalizers, always (dreamer4.py:5254-5255: exists(0))
    assert m.state_dict()['reward_loss_normalizer.exp_avg_sq'].shape == (2,)
    loss = torch.tensor([3., 0.5])
    first = m._normalize_loss('reward_loss_normalizer', loss, True, beta=0.)
    assert torch.equal(first, loss)                                  # initial running mean square is 1
    second = m._normalize_loss('reward_loss_normalizer', loss, True, beta=0.)
    assert torch.allclose(second, torch.ones(2))
    frozen = m._normalize_loss('flow_loss_normalizer', torch.tensor(2.), False)
    assert frozen.item() == 2. and m.flow_loss_normalizer.exp_avg_sq.item() == 1.


def test_every_environment_switch_is_in_the_knob_table():
    """dreamer4_amd/knobs.py is the one table of D4_* switches: every getenv / os.environ name in the sources is in it (with a default and a
    kind), and nothing in the table is stale.  tests/conftest.py and bench.py refuse to run with an 'experiment' switch set, so the defaults
    the GPU tests run under are the defaults the bench runs under."""
    import glob, os, re
    from dreamer4_amd.knobs import KNOBS, experiment_overrides
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    found = set()
    for path in glob.glob(os.path.join(root, 'dreamer4_amd', 'csrc', '*')) + glob.glob(os.path.join(root, 'dreamer4_amd', '*.py')) + [
            os.path.join(root, 'bench.py'), os.path.join(root, 'tests', 'dp_gpu_worker.py')]:
        if os.path.isfile(path) and not path.endswith('knobs.py'):
            src = open(path, errors='ignore').read()
            found |= set(re.findall(r'getenv\("(D4_[A-Z0-9_]+)"\)', src)) | set(re.findall(r"environ(?:\.get|\.setdefault)?[\(\[]\s*'(D4_[A-Z0-9_]+)'", src))
    assert found - set(KNOBS) == set(), f'switches missing from dreamer4_amd/knobs.py: {sorted(found - set(KNOBS))}'
    assert set(KNOBS) - found == set(), f'stale entries in dreamer4_amd/knobs.py: {sorted(set(KNOBS) - found)}'
    assert all(kind in ('experiment', 'mode', 'io') and doc for _, kind, doc in KNOBS.values())
    assert len(KNOBS) <= 12 and not [k for k, (_, kind, _) in KNOBS.items() if kind == 'experiment'], 'round 5 retired every experiment switch'
    assert experiment_overrides({'D4_FORCE_PG': '1'}) == {}
    # a retired name is silently ignored by the library, so the guards treat it like a live experiment switch
    assert list(experiment_overrides({'D4_FRAME_FUSED': '0', 'D4_GEMM_X3': '0', 'D4_FORCE_PG': '1'})) == ['D4_FRAME_FUSED', 'D4_GEMM_X3']


def test_shortcut_coin_replays_when_the_generator_is_reseeded():
    """The training step's shortcut coin is a host draw from a CPU companion of the caller's generator (dreamer4.py:6965 draws it on the host):
    re-seeding the same generator object with the same seed replays the same coin sequence, another seed gives another one."""
    m = small_model()
    g = torch.Generator().manual_seed(5)

    def run(n=8):
        out = []
        for _ in range(n):
            out.append(m._shortcut_coin(g, 0.5))
            torch.rand(3, generator=g)                       # a step's other draws advance the caller's generator
        return out
    a = run()
    g.manual_seed(5)
    assert run() == a
    g.manual_seed(6)
    assert run(32) != (a * 4)


def test_bench_launches_itself_for_several_gpus():
    """`python bench.py --gpus 2` with no launcher around it must become the launcher (one process per GPU under torch.distributed.run) instead of
    failing on WORLD_SIZE.  Without a GPU the ranks then stop at bench.py's "needs an MI355X" check - which proves they were started as ranks."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip('covered on the GPU by tests/test_gpu_dp.py (bare launch)')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode != 0
    assert 'needs an MI355X' in out.stderr and 'WORLD_SIZE=1' not in out.stderr, out.stderr[-2000:]


def test_state_dict_without_tokenizer_keys_needs_an_explicit_opt_in():
    """The tokenizer is a registered submodule (reference: dreamer4.py:4787-4794): a strict load of a checkpoint without `video_tokenizer.*` keys raises, as
    the reference's does.  A checkpoint from before the tokenizer was registered loads with allow_missing_tokenizer=True: the constructor's tokenizer keeps
    its weights, a warning says so, `tokenizer_restored` records it, every other key is still checked; a partial set of tokenizer keys is never patched
    (ADVICE r4 / r5)."""
    from dreamer4_amd import VideoTokenizer
    torch.manual_seed(0)
    tok = VideoTokenizer(dim=32, dim_latent=8, patch_size=4, image_height=8, image_width=8, num_latent_tokens=4, encoder_depth=1, decoder_depth=1, attn_heads=2)
    m = DynamicsWorldModel(dim=32, dim_latent=8, depth=1, num_discrete_actions=3, attn_heads=2, attn_dim_head=16, video_tokenizer=tok)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    tok_keys = [k for k in sd if k.startswith('video_tokenizer.')]
    assert tok_keys
    m.load_state_dict(sd)
    assert m.tokenizer_restored is True
    old = {k: v for k, v in sd.items() if not k.startswith('video_tokenizer.')}
    with pytest.raises(RuntimeError, match='video_tokenizer'):
        m.load_state_dict(old)                                       # strict: missing tokenizer keys are an error by default
    before = {k: v.clone() for k, v in m.video_tokenizer.state_dict().items()}
    with pytest.warns(UserWarning, match='no video_tokenizer'):
        m.load_state_dict(old, allow_missing_tokenizer=True)
    assert m.tokenizer_restored is False
    assert all(torch.equal(v, before[k]) for k, v in m.video_tokenizer.state_dict().items())
    truncated = dict(old); truncated[tok_keys[0]] = sd[tok_keys[0]]
    with pytest.raises(RuntimeError, match='video_tokenizer'):
        m.load_state_dict(truncated, allow_missing_tokenizer=True)   # some tokenizer keys present: nothing is filled in
    del old['register_tokens']
    with pytest.raises(RuntimeError):
        m.load_state_dict(old, allow_missing_tokenizer=True)
