"""`.save(path)` / `.load(path)` / `Klass.init_and_load(path)` of the reference's `@save_load` classes (dreamer4.py:3489, 3684, 4660).

What is pinned by the reference's own code (the only in-tree evidence; the decorator lives in the third-party `torch_einops_utils`,
absent from the image):

* `trainers.py:1074-1088` (`save_checkpoint`): one `torch.save` file, a dict with
      model  = model.state_dict()
      config = pickle.dumps(dehydrate_config(model._config, '_config'))  or  None when the model has no `_config`
      step   = int
  and `model._config` is a live OBJECT on the module (it is pickled at save time, not stored pickled);
* `trainers.py:1048-1072` (`load`): `torch.load(..., weights_only=False)`, `pkg.get('model', pkg)` — a bare state_dict is accepted;
* `cli.py:254, 329`: `Klass.init_and_load(path, strict=False)`.

That is the layout written and read here.  What is NOT pinned is the structure `dehydrate_config` gives the config (it replaces
modules nested in the constructor arguments — a `video_tokenizer=` — by their own config): `init_and_load` therefore accepts the
forms a config can reasonably take — `(args, kwargs)`, `{'args': .., 'kwargs': ..}`, a plain kwargs dict — rebuilds nested
`{'__d4_module__': class name, 'config': ..}` markers (this package's own dehydrated form), and otherwise fails with an error that
says what it found and how to proceed (`Klass(**kwargs).load(path)`).  INTERCHANGE CLAIM: state_dicts (`load`) — yes, key for key;
pickled configs of the real `torch_einops_utils` — unverified.

`torch.load(weights_only=False)` and `pickle.loads` execute code from the file: load checkpoints from trusted sources only (the
reference has the same property)."""
from __future__ import annotations

import os
import pickle

import torch

VERSION = 'dreamer4_amd-0.3'
_MODULE_MARK = '__d4_module__'


def _registry():
    from . import tokenizer, world_model
    return {'VideoTokenizer': tokenizer.VideoTokenizer, 'DynamicsWorldModel': world_model.DynamicsWorldModel}


def dehydrate_config(config, attr='_config'):
    """Replace every module that carries its own `attr` inside (args, kwargs) by a marker holding its class name and (dehydrated)
    config, so that the result pickles without tensors (the role of torch_einops_utils.save_load.dehydrate_config)."""
    def conv(v):
        if isinstance(v, torch.nn.Module):
            inner = getattr(v, attr, None)
            if inner is None:
                raise TypeError(f'{type(v).__name__} in the constructor arguments carries no {attr}: cannot be saved by config')
            return {_MODULE_MARK: type(v).__name__, 'config': dehydrate_config(inner, attr)}
        if isinstance(v, tuple):
            return tuple(conv(x) for x in v)
        if isinstance(v, list):
            return [conv(x) for x in v]
        if isinstance(v, dict):
            return {k: conv(x) for k, x in v.items()}
        return v
    args, kwargs = config
    return conv(tuple(args)), conv(dict(kwargs))


def rehydrate_config(config):
    def conv(v):
        if isinstance(v, dict) and _MODULE_MARK in v:
            klass = _registry().get(v[_MODULE_MARK])
            if klass is None:
                raise TypeError(f'checkpoint config names a nested {v[_MODULE_MARK]!r}: not a class of this package')
            a, k = rehydrate_config(v['config'])
            return klass(*a, **k)
        if isinstance(v, tuple):
            return tuple(conv(x) for x in v)
        if isinstance(v, list):
            return [conv(x) for x in v]
        if isinstance(v, dict):
            return {k: conv(x) for k, x in v.items()}
        return v
    args, kwargs = _as_args_kwargs(config)
    return conv(tuple(args)), conv(dict(kwargs))


def _as_args_kwargs(config):
    if isinstance(config, (tuple, list)) and len(config) == 2 and isinstance(config[0], (tuple, list)) and isinstance(config[1], dict):
        return tuple(config[0]), dict(config[1])
    if isinstance(config, dict) and set(config) >= {'args', 'kwargs'}:
        return tuple(config['args']), dict(config['kwargs'])
    if isinstance(config, dict) and all(isinstance(k, str) for k in config):
        return (), dict(config)
    raise TypeError(f'checkpoint config has an unknown structure ({type(config).__name__}: {str(config)[:120]}...); build the model with '
                    'its constructor arguments and call .load(path) instead')


class SaveLoad:
    """Mixin: the subclass calls `self._record_config(locals())` first thing in __init__.  `self._config` = (args, kwargs), an object
    (as on the reference's modules); a nested `video_tokenizer` stays in it and is dehydrated at save time."""

    def _record_config(self, local_vars):
        kw = {k: v for k, v in local_vars.items() if k not in ('self', '__class__', 'kwargs')}
        kw.update(local_vars.get('kwargs', {}))
        object.__setattr__(self, '_config', ((), kw))           # not a submodule / buffer: out of the module tree and the state_dict

    def save(self, path, overwrite=True, step=None, **extra):
        if not overwrite and os.path.exists(path):
            raise FileExistsError(path)
        config = getattr(self, '_config', None)
        pkg = dict(model=self.state_dict(), config=pickle.dumps(dehydrate_config(config, '_config')) if config else None, version=VERSION, **extra)
        if step is not None:
            pkg['step'] = int(step)
        torch.save(pkg, str(path))

    def load(self, path, strict=True):
        pkg = torch.load(str(path), map_location='cpu', weights_only=False)
        state = pkg.get('model', pkg) if isinstance(pkg, dict) else pkg
        self.load_state_dict(state, strict=strict)
        return pkg

    @classmethod
    def init_and_load(cls, path, strict=True, **override):
        pkg = torch.load(str(path), map_location='cpu', weights_only=False)
        if not (isinstance(pkg, dict) and 'model' in pkg):
            raise TypeError(f'{path} holds a bare state_dict (no config): build {cls.__name__}(...) and call .load(path)')
        raw = pkg.get('config')
        if raw is None:
            if not override:
                raise TypeError(f'{path} was saved without a config (config = None, trainers.py:1085): pass the constructor arguments to '
                                f'{cls.__name__}.init_and_load(path, **kwargs) or build the model and call .load(path)')
            args, kwargs = (), {}
        else:
            args, kwargs = rehydrate_config(pickle.loads(raw) if isinstance(raw, (bytes, bytearray)) else raw)
        kwargs.update(override)
        model = cls(*args, **kwargs)
        model.load_state_dict(pkg['model'], strict=strict)
        return model
