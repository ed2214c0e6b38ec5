"""Compile the HIP kernels + C-ABI into simulate_2048_amd/lib/libenv2048.so (gfx950 only)."""

from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
REPO = PKG.parent
SOURCES = [PKG / 'csrc' / 'env2048_kernels.hip']
HEADERS = [PKG / 'csrc' / 'board_ops.hpp', REPO / 'include' / 'env2048.h']
LIB = PKG / 'lib' / 'libenv2048.so'
ARCH = 'gfx950'


def hipcc() -> str:
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not Path(exe).exists():
        raise RuntimeError('hipcc not found: the HIP toolchain (ROCm) is required to build libenv2048.so')
    return exe


def needs_build() -> bool:
    if not LIB.exists():
        return True
    newest = max(p.stat().st_mtime for p in SOURCES + HEADERS)
    return LIB.stat().st_mtime < newest


def build_library(force: bool = False, verbose: bool = False, extra_flags: list[str] | None = None) -> Path:
    """hipcc --offload-arch=gfx950 -shared -fPIC ... -> lib/libenv2048.so. Cross-compiles without a GPU."""
    if not force and not needs_build():
        return LIB
    LIB.parent.mkdir(exist_ok=True)
    tmp = LIB.with_suffix('.so.tmp%d' % os.getpid())
    cmd = [hipcc(), f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-shared', '-fPIC', '-fvisibility=hidden',
           '-pthread', '-DENV2048_BUILDING', f'-I{REPO / "include"}', f'-I{PKG / "csrc"}', '-Wall', '-Wno-unused-function',
           # leading kernel arguments arrive in SGPRs with the wave instead of through an s_load (gfx950 kernarg
           # preload; the code object keeps the s_load prologue as the fallback for firmware without it)
           '-mllvm', '-amdgpu-kernarg-preload-count=16']
    cmd += extra_flags or []
    cmd += [str(s) for s in SOURCES] + ['-o', str(tmp)]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or proc.returncode != 0:
        print(' '.join(cmd))
        print(proc.stdout + proc.stderr)
    if proc.returncode != 0:
        tmp.unlink(missing_ok=True)
        raise RuntimeError('hipcc failed:\n' + proc.stderr)
    os.replace(tmp, LIB)
    return LIB


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))
