"""`.save(path)` / `.load(path)` / `Klass.init_and_load(path)` of the reference's `@save_load` classes (dreamer4.py:3489, 3684, 4660;
trainers.py:1083-1088): one `torch.save` file holding the state_dict and the pickled constructor arguments, so a model can be
rebuilt from the file alone.  The decorator itself comes from the third-party `torch_einops_utils` (absent from the image): the
layout below — {'model': state_dict, 'config': pickle((args, kwargs)), 'version': str} — follows its published behaviour and is
UNVERIFIED against the real package; `load` also accepts a bare state_dict."""
from __future__ import annotations

import pickle

import torch

VERSION = 'dreamer4_amd-0.2'


class SaveLoad:
    """Mixin: the subclass calls `self._record_config(locals())` first thing in __init__."""

    def _record_config(self, local_vars):
        kw = {k: v for k, v in local_vars.items() if k not in ('self', '__class__', 'kwargs')}
        kw.update(local_vars.get('kwargs', {}))
        kw.pop('video_tokenizer', None)                      # a module, not a constructor constant: re-attach after loading
        self._config = pickle.dumps(((), kw))

    def save(self, path, overwrite=True, **extra):
        import os
        if not overwrite and os.path.exists(path):
            raise FileExistsError(path)
        torch.save(dict(model=self.state_dict(), config=self._config, version=VERSION, **extra), str(path))

    def load(self, path, strict=True):
        pkg = torch.load(str(path), map_location='cpu', weights_only=False)
        state = pkg['model'] if isinstance(pkg, dict) and 'model' in pkg else pkg
        self.load_state_dict(state, strict=strict)
        return pkg

    @classmethod
    def init_and_load(cls, path, strict=True, **override):
        pkg = torch.load(str(path), map_location='cpu', weights_only=False)
        args, kwargs = pickle.loads(pkg['config'])
        kwargs.update(override)
        model = cls(*args, **kwargs)
        model.load_state_dict(pkg['model'], strict=strict)
        return model
