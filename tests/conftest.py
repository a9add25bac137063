import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture
def sg_env(monkeypatch):
    """Sets SG_* diagnostic switches for one test.  The library reads them once, so a new snapshot is taken after every
    change (sg_config_reload) and again when the test's environment is restored."""
    from saragan_amd import _lib

    def set_(**kv):
        for k, v in kv.items():
            monkeypatch.setenv(k, str(v))
        assert _lib.load().sg_config_reload() == 0
    yield set_
    monkeypatch.undo()
    _lib.load().sg_config_reload()
