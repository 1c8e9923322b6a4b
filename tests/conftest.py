import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs an MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (build container only)')


def pytest_sessionstart(session):
    """The GPU tests must run under the product's defaults (the defaults bench.py runs under): refuse a session with an experiment switch
    set in the environment.  Tests that exercise the other arm of a switch set it themselves for one engine (monkeypatch)."""
    from dreamer4_amd.knobs import experiment_overrides
    over = experiment_overrides()
    if over:
        raise pytest.UsageError(f'experiment switches set in the environment: {over} - unset them (dreamer4_amd/knobs.py)')
