"""
simulate_2048_amd -- MI355X-native batched 2048 environment (the env hot path of
schalappe/simulate_2048, rebuilt as hand-written HIP kernels for gfx950 behind a C-ABI).

Layers
------
csrc/      HIP kernels + the C-ABI of include/env2048.h  ->  lib/libenv2048.so
capi       ctypes binding of that ABI (raw pointers, no torch types)
batch      BatchedEnv2048: device buffers + kernel launches for N boards on one GPU
trajectory TrajectoryBuffer: rows written by the step kernel, episode records cut on device
sharding   contiguous slicing of a global batch over ranks/GPUs (no collective on the hot path)
interop    DeviceArray (np.array / __dlpack__) and array-like input conversion for the mirrors
twentyfortyeight/, reinforce/game/   drop-in mirrors of the reference's Python APIs

There is no CPU implementation in this package: without the HIP library and a GPU every compute
call raises.  The CPU oracle used by the tests lives in the top-level ``oracle/`` directory.
"""

from . import capi  # noqa: F401
from .batch import BatchedEnv2048  # noqa: F401
from .batch_nxn import BatchedEnvNxN  # noqa: F401
from .trajectory import TrajectoryBuffer  # noqa: F401

__all__ = ['capi', 'BatchedEnv2048', 'BatchedEnvNxN', 'TrajectoryBuffer']
