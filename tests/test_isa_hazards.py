"""
CPU-side guard (no GPU): compile the product kernels to gfx950 assembly and check mechanically that every store of more
than 64 bits -- in particular the inline-asm write-through stores of the step kernel -- carries its wait state
(tools/isa_hazards.py).  The bug this prevents corrupted boards only under memory back-pressure at 2^20 boards
(round 1, gpurun_out/abort.log); nothing smaller showed it.
"""

import sys

import pytest

from conftest import REPO

sys.path.insert(0, str(REPO / 'tools'))
import isa_hazards  # noqa: E402


@pytest.fixture(scope='module')
def assembly(tmp_path_factory):
    out = tmp_path_factory.mktemp('isa') / 'env2048_kernels.s'
    return isa_hazards.build_assembly(out).read_text()


def test_every_wide_store_has_its_wait_state(assembly):
    problems, stats = isa_hazards.check(assembly)
    assert not problems, '\n'.join(problems)
    assert stats['wide_stores_in_asm'] >= 8, stats  # the write-through board / reward / obs stores exist and were seen
    assert stats['wide_stores'] > stats['wide_stores_in_asm']  # compiler-emitted ones were scanned as well


def test_the_checker_catches_both_hazard_forms():
    good = '\t;;#ASMSTART\n\tglobal_store_dwordx4 v[2:3], v[4:7], off sc1\n\ts_nop 1\n\t;;#ASMEND\n\tv_mov_b32_e32 v4, 0\n'
    assert isa_hazards.check(good)[0] == []
    missing_nop = good.replace('\ts_nop 1\n', '')
    assert any('not followed by s_nop' in p for p in isa_hazards.check(missing_nop)[0])
    compiler_store = '\tglobal_store_dwordx4 v[2:3], v[4:7], off\n\tv_add_u32_e32 v6, v1, v2\n'
    assert any('overwrites its data register' in p for p in isa_hazards.check(compiler_store)[0])
    harmless = '\tglobal_store_dwordx4 v[2:3], v[4:7], off\n\tv_add_u32_e32 v8, v1, v2\n\tglobal_store_dwordx2 v[2:3], v[4:5], off\n\tv_mov_b32_e32 v4, 0\n'
    assert isa_hazards.check(harmless)[0] == []  # other destination; 64-bit stores are not affected
    buffer_form = '\tbuffer_store_dwordx4 v[10:13], v1, s[0:3], 0 offen\n\tv_mov_b32_e32 v12, 0\n'
    assert any('v[12]' in p for p in isa_hazards.check(buffer_form)[0])
