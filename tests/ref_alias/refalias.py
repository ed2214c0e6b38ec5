"""pytest plugin (``-p refalias``) used by tests/test_reference_suite_on_mirrors.py: makes ``import twentyfortyeight...``
resolve to the drop-in mirrors ``simulate_2048_amd.twentyfortyeight...`` (and ``reinforce.game...`` to ``simulate_2048_amd.reinforce.game...``) so that the REFERENCE'S OWN test files can be
collected from where they lie (/root/reference/tests, build container only) and run against this package.  Without a GPU
the mirrors run on the oracle-backed stand-in; with one, on the HIP library."""

import importlib
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / 'tests'))
sys.dont_write_bytecode = True

for name in ('', '.core', '.core.gameboard', '.core.gamemove', '.envs', '.envs.twentyfortyeight', '.utils', '.utils.binary',
             '.utils.normalize'):
    sys.modules['twentyfortyeight' + name] = importlib.import_module('simulate_2048_amd.twentyfortyeight' + name)
for name in ('', '.game', '.game.core', '.game.env'):
    sys.modules['reinforce' + name] = importlib.import_module('simulate_2048_amd.reinforce' + name)

import torch  # noqa: E402

if not torch.cuda.is_available():
    from oracle_backend import OracleLibrary  # noqa: E402
    from simulate_2048_amd import runtime  # noqa: E402

    runtime._current = runtime.Runtime(OracleLibrary(), 'cpu')


def pytest_terminal_summary(terminalreporter):
    import twentyfortyeight.core.gameboard as g

    import reinforce.game.env as e

    terminalreporter.write_line(f'refalias: twentyfortyeight.core.gameboard -> {g.__file__}')
    terminalreporter.write_line(f'refalias: reinforce.game.env -> {e.__file__}')
