#!/usr/bin/env python3
"""
Static check of the gfx950 assembly of the product kernels for the hazard that corrupted boards in round 1
(gpurun_out/abort.log): on gfx9 a VMEM store of MORE than 64 bits of data reads its data VGPRs one cycle after issue,
so the instruction right after it must not be a VALU write to one of those registers.  hipcc inserts the wait state
for stores it emits itself, but it cannot see inside an `asm volatile` statement -- and the write-through stores of
the step kernel (global_store_dwordx4 ... sc1) are exactly that.  Rules checked on every kernel of the .s file:

  R1  inside an inline-asm block (;;#ASMSTART .. ;;#ASMEND) every store of > 64 bits is followed, INSIDE the block, by
      an s_nop (the block carries its own wait state, whatever the compiler schedules behind it);
  R2  anywhere: the instruction after a > 64-bit store is not a VALU instruction whose destination overlaps the
      store's data registers (covers compiler-emitted stores as well: a regression test of the toolchain).

usage: tools/isa_hazards.py file.s      (exit code 1 and a list of offending lines on violation)
       build_assembly(out) compiles the product source to device assembly with the flags of simulate_2048_amd/build.py
"""

from __future__ import annotations

import re
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
WIDE_STORE = re.compile(r'^\s*(global_store_dwordx[34]|buffer_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\s+(.*)$')
VREG_RANGE = re.compile(r'v\[(\d+):(\d+)\]')
VREG = re.compile(r'\bv(\d+)\b')


def build_assembly(out: Path) -> Path:
    sys.path.insert(0, str(REPO))
    from simulate_2048_amd import build as hip_build

    from concurrent.futures import ThreadPoolExecutor

    def one(src: Path) -> str:  # one translation unit -> its device assembly
        part = out.with_name(out.stem + '.' + src.stem + '.s')
        cmd = [hip_build.hipcc(), f'--offload-arch={hip_build.ARCH}', '-O3', '-std=c++17', '-DENV2048_BUILDING', f'-I{REPO / "include"}',
               f'-I{hip_build.PKG / "csrc"}', '-mllvm', '-amdgpu-kernarg-preload-count=16', '-S', '--cuda-device-only', '-o', str(part),
               str(src)]
        subprocess.run(cmd, check=True, capture_output=True)
        return part.read_text()

    with ThreadPoolExecutor(max_workers=len(hip_build.SOURCES)) as pool:
        out.write_text('\n'.join(pool.map(one, hip_build.SOURCES)))
    return out


def data_registers(opcode: str, operands: str) -> set[int]:
    """VGPRs holding the store DATA.  global/flat/scratch: `addr, data, ...` (data is the 2nd operand);
    buffer: `data, vaddr|off, srsrc, ...` (data is the 1st)."""
    parts = [p.strip() for p in operands.split(',')]
    data = parts[0] if opcode.startswith('buffer_') else (parts[1] if len(parts) > 1 else '')
    m = VREG_RANGE.search(data)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = VREG.search(data)
    return {int(m.group(1))} if m else set()


def destination_registers(line: str) -> set[int]:
    """VGPRs written by a VALU instruction = its first operand (v_cmp* etc. write SGPRs/VCC: nothing)."""
    parts = line.split(None, 1)
    if len(parts) < 2:
        return set()
    first = parts[1].split(',')[0].strip()
    m = VREG_RANGE.fullmatch(first)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = VREG.fullmatch(first)
    return {int(m.group(1))} if m else set()


def instructions(text: str):
    """(line number, text, inside_asm_block) of every instruction / asm marker line."""
    inside = False
    for no, raw in enumerate(text.splitlines(), 1):
        line = raw.strip()
        if line.startswith(';;#ASMSTART'):
            inside = True
            yield no, 'ASMSTART', inside
            continue
        if line.startswith(';;#ASMEND'):
            inside = False
            yield no, 'ASMEND', inside
            continue
        if not raw.startswith('\t') or not line or line.startswith(('.', ';', '//')) or line.endswith(':'):
            continue
        yield no, line.split(';')[0].strip(), inside


def check(text: str) -> tuple[list[str], dict]:
    seq = list(instructions(text))
    problems = []
    stats = {'wide_stores': 0, 'wide_stores_in_asm': 0, 'asm_blocks': sum(1 for _, t, _ in seq if t == 'ASMSTART')}
    for idx, (no, line, inside) in enumerate(seq):
        m = WIDE_STORE.match(line)
        if not m:
            continue
        stats['wide_stores'] += 1
        nxt = seq[idx + 1] if idx + 1 < len(seq) else (no, 'END', False)
        if inside:
            stats['wide_stores_in_asm'] += 1
            if not nxt[1].startswith('s_nop'):  # R1: the wait state must be part of the asm statement itself
                problems.append(f'line {no}: inline-asm `{line}` is not followed by s_nop inside its asm block (next: `{nxt[1]}`)')
                continue
        # R2: skip markers; the next real instruction must not be a VALU write to the data registers
        j = idx + 1
        while j < len(seq) and seq[j][1] in ('ASMSTART', 'ASMEND'):
            j += 1
        if j < len(seq) and seq[j][1].startswith('v_'):
            clash = data_registers(m.group(1), m.group(2)) & destination_registers(seq[j][1])
            if clash:
                problems.append(f'line {no}: `{line}` is followed directly by `{seq[j][1]}` (line {seq[j][0]}), which overwrites its '
                                f'data register(s) v{sorted(clash)}')
    return problems, stats


def main(path: str) -> int:
    problems, stats = check(Path(path).read_text())
    print(stats)
    for p in problems:
        print('HAZARD', p)
    return 1 if problems else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1]))
