"""Shared pytest configuration: the `gpu` marker and fixture loaders."""

import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO / 'tests') not in sys.path:
    sys.path.insert(0, str(REPO / 'tests'))
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = REPO / 'tests' / 'golden'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    """Dict name -> NpzFile of tests/golden (data only: numpy.load without pickle)."""
    return {p.stem: np.load(p, allow_pickle=False) for p in sorted(GOLDEN.glob('*.npz'))}


def pytest_sessionstart(session):
    # The oracle is test infrastructure: build it once up front so failures are obvious.
    import oracle

    oracle.build()


@pytest.fixture(params=['host-logic', pytest.param('hip', marks=pytest.mark.gpu)])
def api_runtime(request):
    """Run an API-mirror test twice: with the oracle-backed stand-in library on CPU tensors (checks
    the host-side logic in a container without a GPU) and with the real HIP library on the GPU."""
    from simulate_2048_amd import runtime

    if request.param == 'hip':
        yield runtime.current()
    else:
        from oracle_backend import OracleLibrary

        with runtime.use(OracleLibrary(), 'cpu') as rt:
            yield rt
