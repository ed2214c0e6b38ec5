"""Compile the HIP kernels + C-ABI into simulate_2048_amd/lib/libenv2048.so (gfx950 only).

One translation unit per kernel family (csrc/*.hip), compiled side by side and linked into one shared library; there are no
device calls across translation units, so no relocatable device code is needed."""

from __future__ import annotations

import os
import shutil
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
REPO = PKG.parent
CSRC = PKG / 'csrc'
SOURCES = [CSRC / name for name in ('step_kernels.hip', 'rollout_kernels.hip', 'state_i32_kernels.hip', 'search_kernels.hip',
                                    'transform_kernels.hip', 'nxn_kernels.hip')]
HEADERS = sorted(CSRC.glob('*.hpp')) + [REPO / 'include' / 'env2048.h', CSRC / 'direct_image.S']
LIB = PKG / 'lib' / 'libenv2048.so'
ARCH = 'gfx950'
COMMON_FLAGS = [f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-pthread', '-DENV2048_BUILDING',
                f'-I{REPO / "include"}', f'-I{CSRC}', '-Wall', '-Wno-unused-function',
                # leading kernel arguments arrive in SGPRs with the wave instead of through an s_load (gfx950 kernarg
                # preload; the code object keeps the s_load prologue as the fallback for firmware without it)
                '-mllvm', '-amdgpu-kernarg-preload-count=16']


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


def _run(cmd: list[str], verbose: bool) -> None:
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or proc.returncode != 0:
        print(' '.join(cmd))
        print(proc.stdout + proc.stderr)
    if proc.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + proc.stderr)


def build_library(force: bool = False, verbose: bool = False, extra_flags: list[str] | None = None, out: Path | None = None) -> Path:
    """hipcc --offload-arch=gfx950 -c csrc/*.hip (in parallel) + one -shared link -> lib/libenv2048.so (or `out`, for the
    experiment builds of tools/build_exp.sh).  Cross-compiles without a GPU."""
    target = Path(out) if out else LIB
    if not force and out is None and not needs_build():
        return LIB
    target.parent.mkdir(exist_ok=True)
    flags = COMMON_FLAGS + (extra_flags or [])
    with tempfile.TemporaryDirectory(prefix='env2048_build_') as tmpdir:
        objects = [Path(tmpdir) / (s.stem + '.o') for s in SOURCES]
        # the step kernels once more as a BARE code object (device code only, same flags): direct_dispatch.hpp loads it through
        # ROCr for the queues the library drives itself; direct_image.S links it into the shared object as read-only data
        image = Path(tmpdir) / 'step_kernels.hsaco'
        image_object = Path(tmpdir) / 'direct_image.o'
        with ThreadPoolExecutor(max_workers=min(len(SOURCES) + 1, os.cpu_count() or 1)) as pool:
            jobs = [pool.submit(_run, [hipcc(), *flags, '-c', str(s), '-o', str(o)], verbose) for s, o in zip(SOURCES, objects)]
            jobs.append(pool.submit(_run, [hipcc(), *flags, '--offload-device-only', '--no-gpu-bundle-output', str(SOURCES[0]), '-o', str(image)],
                                    verbose))
            for job in jobs:
                job.result()
        _run(['gcc', '-c', f'-DENV2048_DIRECT_IMAGE="{image}"', str(CSRC / 'direct_image.S'), '-o', str(image_object)], verbose)
        tmp = target.with_suffix('.so.tmp%d' % os.getpid())
        try:
            _run([hipcc(), f'--offload-arch={ARCH}', '-shared', '-fPIC', '-pthread', *[str(o) for o in objects], str(image_object), '-L/opt/rocm/lib',
                  '-lhsa-runtime64', '-o', str(tmp)], verbose)
            os.replace(tmp, target)
        finally:
            tmp.unlink(missing_ok=True)
    return target


if __name__ == '__main__':
    import sys

    args = sys.argv[1:]
    out_path = None
    if '--out' in args:
        i = args.index('--out')
        out_path = Path(args[i + 1])
        del args[i:i + 2]
    print(build_library(force=True, verbose=True, extra_flags=args, out=out_path))
