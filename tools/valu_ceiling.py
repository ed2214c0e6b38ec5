#!/usr/bin/env python3
"""MEASURED VALU issue ceiling of the step kernel's own instruction mix (the 614 G wave-instr/s "peak"
of round 2 assumed 4 cycles per wave64 instruction; the hardware guide says 2).

Inputs:  the per-opcode issue-rate table measured by tools/issue_probe on an MI355X (profiles/*_issue_probe.json: cycles per
         wave64 instruction per SIMD at 1 / 2 / 4 / 8 waves per SIMD) and the static opcode histogram of
         step_kernel<false, false, 4> (compiled here with -save-temps; hipcc cross-compiles without a GPU).
Output:  profiles/valu_ceiling.json -- the weighted cycles per instruction of the kernel's mix at each occupancy, and the
         ceiling  1024 SIMDs x 2.4 GHz / cycles  in G wave-instructions/s.  bench.py uses the 4-waves-per-SIMD figure (what a
         2^20-board launch of the four-boards-per-lane tiling gives every SIMD) as roofline.peak, with this file as its source.

Every opcode of the kernel is matched to the probe row of the same opcode (and form: an inline-constant source makes
v_bitop3_b32 a "two register reads" instruction, which issues like a VOP2 add); the few opcodes without a row of their own
are priced like the SLOWER class (three-source VOP3), which can only lower the ceiling, never flatter the fraction.

    python tools/valu_ceiling.py [probe.json] [out.json]
"""
import collections
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
KERNEL = '_ZN12_GLOBAL__N_111step_kernelILb0ELb0ELi4EEEvPhPKhmmjjjjjPiS1_S1_PvS1_Py'
SIMDS, PEAK_GHZ = 256 * 4, 2.4

# opcode (without encoding suffix) -> probe row
ROW = {
    'v_add_u32': 'v_add_u32 (e32) indep', 'v_sub_u32': 'v_sub_u32 (e32) indep', 'v_subrev_u32': 'v_sub_u32 (e32) indep',
    'v_and_b32': 'v_and_b32 (e32) indep', 'v_or_b32': 'v_or_b32 (e32) indep', 'v_xor_b32': 'v_xor_b32 (e32) varied registers',
    'v_mov_b32': 'v_mov_b32 indep', 'v_not_b32': 'v_not_b32 (e32) indep', 'v_lshrrev_b32': 'v_lshrrev_b32 (e32) indep',
    'v_lshlrev_b32': 'v_lshlrev_b32 (e32) indep', 'v_ashrrev_i32': 'v_ashrrev_i32 (e32) indep', 'v_max_u32': 'v_max_u32 (e32) indep',
    'v_min_u32': 'v_min_u32 (e32) indep', 'v_perm_b32': 'v_perm_b32 chain', 'v_bitop3_b32': 'v_bitop3_b32 chain',
    'v_xad_u32': 'v_xad_u32 chain', 'v_mad_u64_u32': 'v_mad_u64_u32 chain', 'v_bcnt_u32_b32': 'v_bcnt_u32_b32 chain',
    'v_add_lshl_u32': 'v_add_lshl_u32 indep', 'v_and_or_b32': 'v_and_or_b32 indep', 'v_add3_u32': 'v_add3_u32 indep',
    'v_lshl_or_b32': 'v_lshl_or_b32 indep', 'v_or3_b32': 'v_or3_b32 indep', 'v_lshl_add_u64': 'v_lshl_add_u64 indep',
    'v_mul_hi_u32': 'v_mul_hi_u32 indep', 'v_mul_lo_u32': 'v_mul_lo_u32 indep', 'v_alignbit_b32': 'v_alignbit_b32 indep',
    'v_bfe_u32': 'v_bfe_u32 indep', 'v_addc_co_u32': 'v_add_co_u32 (e32 -> vcc) indep', 'v_add_co_u32': 'v_add_co_u32 (e32 -> vcc) indep',
}
SLOW_ROW = 'v_bitop3_b32 indep'  # three register sources: the class every unlisted opcode is priced as


def histogram() -> tuple[collections.Counter, int]:
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-fvisibility=hidden', '-DENV2048_BUILDING',
               f'-I{REPO / "include"}', f'-I{REPO / "simulate_2048_amd" / "csrc"}', '-mllvm', '-amdgpu-kernarg-preload-count=16',
               '-save-temps', str(REPO / 'simulate_2048_amd' / 'csrc' / 'step_kernels.hip'), '-o', str(Path(tmp) / 'lib.so')]
        subprocess.run(cmd, cwd=tmp, check=True, capture_output=True)
        text = next(Path(tmp).glob('*gfx950.s')).read_text()
    m = re.search(r'^' + re.escape(KERNEL) + r':[^\n]*\n(.*?)\n\s*s_endpgm', text, flags=re.S | re.M)
    if not m:
        raise SystemExit('step kernel not found in the assembly')
    rows = collections.Counter()
    nops = 0
    for line in m.group(1).splitlines():
        if not line.startswith('\t'):
            continue
        parts = line.strip().split(None, 1)
        op = parts[0]
        if op == 's_nop':
            nops += 1
        if not op.startswith('v_'):
            continue
        base = re.sub(r'_(e32|e64|sdwa|dpp)$', '', op)
        if op.endswith('_sdwa'):
            row = 'v_lshlrev_b32_sdwa chain'
        elif base.startswith('v_cmp_'):
            row = 'v_cmp_gt_u32 (e64 -> SGPR pair) indep' if op.endswith('_e64') else 'v_cmp_gt_u32 (e32 -> vcc) indep'
        elif base == 'v_cndmask_b32':
            row = 'v_cndmask_b32 (e64, SGPR-pair mask) indep' if op.endswith('_e64') else 'v_cmp_gt_u32 vcc + v_cndmask_b32 vcc pairs'
        elif base == 'v_bitop3_b32':
            operands = [o.strip() for o in parts[1].split('bitop3')[0].split(',')][1:]
            inline = sum(1 for o in operands if re.fullmatch(r'-?\d+|0x[0-9a-fA-F]+', o))
            row = 'v_bitop3_b32 inline constant source' if inline else ROW[base]
        else:
            row = ROW.get(base, SLOW_ROW)
        rows[(row, base if base not in ROW and not base.startswith(('v_cmp_', 'v_cndmask', 'v_bitop3', 'v_lshlrev_b32')) else '')] += 1
    return rows, nops


def main():
    probe_path = Path(sys.argv[1]) if len(sys.argv) > 1 else sorted((REPO / 'profiles').glob('*issue_probe.json'))[-1]
    out_path = Path(sys.argv[2]) if len(sys.argv) > 2 else REPO / 'profiles' / 'valu_ceiling.json'
    probe = json.loads(probe_path.read_text())
    ops = probe['ops']
    rows, nops = histogram()
    total = sum(rows.values())
    per_row = collections.Counter()
    unlisted = collections.Counter()
    for (row, unlisted_op), count in rows.items():
        if row not in ops:
            row = SLOW_ROW
        per_row[row] += count
        if unlisted_op:
            unlisted[unlisted_op] += count
    out = {
        'kernel': 'step_kernel<false, false, 4>',
        'probe': str(probe_path.relative_to(REPO)) if probe_path.is_relative_to(REPO) else str(probe_path),
        'static_valu_instructions': total,
        'static_s_nop': nops,
        'mix': {row: {'count': c, 'share': round(c / total, 4), 'cycles_at_waves_per_simd': dict(zip(('1', '2', '4', '8'), ops[row]['cycles']))}
                for row, c in per_row.most_common()},
        'priced_as_three_source_vop3': dict(unlisted),
        'ceiling': {},
        'how': 'weighted mean of the probe cycles over the static opcode histogram of the kernel (both the bounds-checked and the '
               'unchecked copy of the block code: the same mix twice); ceiling = 1024 SIMDs x 2.4 GHz / cycles',
    }
    for idx, waves in enumerate(('1', '2', '4', '8')):
        cycles = sum(c * ops[row]['cycles'][idx] for row, c in per_row.items()) / total
        out['ceiling'][waves] = {'cycles_per_wave_instruction': round(cycles, 3), 'g_wave_instr_per_s': round(SIMDS * PEAK_GHZ / cycles, 1)}
    # what the REAL instruction stream sustains when no memory access is in the way: the fused rollout kernel (same game logic,
    # same Philox, boards in registers), measured by tools/bench_rollout_valu.py + a rocprofv3 --pmc pass
    stream = sorted((REPO / 'profiles').glob('*rollout_valu_rate.json'))
    if stream:
        rows = json.loads(stream[-1].read_text())['rows']
        best = max(rows.values(), key=lambda r: r['g_wave_instr_per_s'])
        at4 = rows.get('2^20 bpl4')
        out['real_instruction_stream'] = {
            'source': str(stream[-1].relative_to(REPO)),
            'saturated_g_wave_instr_per_s': best['g_wave_instr_per_s'],
            'at_4_waves_per_simd_g_wave_instr_per_s': None if at4 is None else at4['g_wave_instr_per_s'],
            'note': 'env2048_rollout sustains 553-609 G wave-instr/s at 4 waves per SIMD and saturates at 630-643 from 8 waves per SIMD up '
                    '(3.5 cycles per instruction per SIMD): the single-opcode streams of the probe issue faster than the real, dependent, '
                    'register-diverse stream -- the probe-weighted figure is an upper bound, this is what the code can reach',
        }
    out_path.write_text(json.dumps(out, indent=1) + '\n')
    print(json.dumps(out['ceiling'], indent=1))
    print('unlisted opcodes priced as three-source VOP3:', dict(unlisted))


if __name__ == '__main__':
    main()
