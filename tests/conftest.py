import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def engine_factory():
    """Creates pindel_amd Engines; fails loudly (no fallback) when the HIP library cannot run."""
    from pindel_amd import binding
    engines = []

    def make(**kw):
        e = binding.Engine(**kw)
        engines.append(e)
        return e
    yield make
    for e in engines:
        e.close()


@pytest.fixture
def pg_env():
    """set(name, value) / unset(name) for the library's environment switches (PG_HOST_CHUNK, PG_GENERIC_KERNELS ...): the
    library reads them once per process, so every change -- and the restore at the end of the test -- is followed by
    binding.reload_env()."""
    from types import SimpleNamespace
    from pindel_amd import binding
    saved = {}

    def set_(k, v):
        saved.setdefault(k, os.environ.get(k))
        os.environ[k] = v
        binding.reload_env()

    def unset(k):
        saved.setdefault(k, os.environ.get(k))
        os.environ.pop(k, None)
        binding.reload_env()
    yield SimpleNamespace(set=set_, unset=unset)
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    binding.reload_env()
