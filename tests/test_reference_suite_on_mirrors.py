"""
The reference's OWN test files (tests/test_board.py, test_environment.py, test_move.py: 35 tests) run against the drop-in
mirrors: ``twentyfortyeight`` is aliased to ``simulate_2048_amd.twentyfortyeight`` by the plugin tests/ref_alias/refalias.py
and the files are collected from where they lie under /root/reference -- nothing of the reference is copied, and where the
reference is absent (the GPU box) the test skips.  This is the strongest form of "a user of the reference can switch":
its own suite cannot tell the difference.

tests/test_jax_game.py (37 tests of ``reinforce.game``) imports ``jax``, which is not installed here and cannot be.  It is run
too, against the ``reinforce.game`` mirrors, with a TEST DOUBLE of the few jax / jax.numpy / jax.random names the test file
itself uses (tests/ref_alias/jax_shim: NumPy underneath, jit = identity, vmap = a loop) -- the engine under test is this
repo's, never the reference's.  All 37 pass; the two that assert ``obs.dtype == jnp.float32`` / ``mask.dtype == jnp.bool_``
do so because the DOUBLE's ``jnp.float32`` / ``jnp.bool_`` compare equal to the torch dtype of the same element type
(jax_shim/jax/numpy.py) -- the product's ``.dtype`` is a plain ``torch.dtype`` (round 2 bent the product's return type for
these two assertions).
(The same known-answer vectors are also transcribed in tests/test_api_reinforce_game.py, which runs on the GPU box.)
"""

import os
import re
import subprocess
import sys
from pathlib import Path

import pytest

from conftest import REPO

REFERENCE_TESTS = Path('/root/reference/tests')
FILES = ['test_board.py', 'test_environment.py', 'test_move.py']


@pytest.mark.skipif(not all((REFERENCE_TESTS / f).exists() for f in FILES), reason='the reference is not present on this machine')
def test_reference_numpy_env_suite_passes_on_the_mirrors(tmp_path):
    env = dict(os.environ, PYTHONPATH=str(REPO / 'tests' / 'ref_alias'), PYTHONDONTWRITEBYTECODE='1')
    cmd = [sys.executable, '-m', 'pytest', '-p', 'no:cacheprovider', '-p', 'refalias', '-q', '-rA'] + [str(REFERENCE_TESTS / f) for f in FILES]
    proc = subprocess.run(cmd, env=env, cwd=tmp_path, capture_output=True, text=True, timeout=600)
    out = proc.stdout + proc.stderr
    assert 'refalias: twentyfortyeight.core.gameboard -> ' + str(REPO / 'simulate_2048_amd') in out, out[-2000:]  # really the mirror
    assert proc.returncode == 0, out[-3000:]
    passed = re.search(r'(\d+) passed', out)
    assert passed and int(passed.group(1)) >= 35 and 'failed' not in out.splitlines()[-1], out[-1500:]


@pytest.mark.skipif(not (REFERENCE_TESTS / 'test_jax_game.py').exists(), reason='the reference is not present on this machine')
def test_reference_jax_game_suite_on_the_mirrors_with_a_jax_test_double(tmp_path):
    shim = REPO / 'tests' / 'ref_alias'
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(shim), str(shim / 'jax_shim')]), PYTHONDONTWRITEBYTECODE='1')
    cmd = [sys.executable, '-m', 'pytest', '-p', 'no:cacheprovider', '-p', 'refalias', '-q', '-rA', str(REFERENCE_TESTS / 'test_jax_game.py')]
    proc = subprocess.run(cmd, env=env, cwd=tmp_path, capture_output=True, text=True, timeout=600)
    out = proc.stdout + proc.stderr
    assert 'refalias: reinforce.game.env -> ' + str(REPO / 'simulate_2048_amd') in out, out[-2000:]
    assert proc.returncode == 0, out[-3000:]
    passed = re.search(r'(\d+) passed', out)
    assert passed and int(passed.group(1)) >= 37 and 'failed' not in out.splitlines()[-1], out[-1500:]
