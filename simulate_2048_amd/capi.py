"""
ctypes binding of ``libenv2048.so`` (include/env2048.h) -- the thin layer between Python host code
and the HIP kernels.  No torch types cross the ABI: callers hand over raw device pointers
(``tensor.data_ptr()``), a device ordinal and a ``hipStream_t`` handle.

There is deliberately no CPU fallback here: if the library is missing, cannot be loaded, or no HIP
device is usable, the calls raise.
"""

from __future__ import annotations

import ctypes
from pathlib import Path

ABI_VERSION = 5

OK = 0
ERR_INVALID_ARG = -1
ERR_ALIGNMENT = -2
ERR_DEVICE = -3
ERR_RANGE = -4
ERR_HIP_BASE = -1000

FLAG_AUTO_RESET = 1
FLAG_RANDOM_ACTIONS = 2
FLAG_OBS_F16 = 4
FLAG_OBS_BF16 = 8
FLAG_STORES_WRITE_THROUGH = 16
FLAG_STORES_WRITE_BACK = 32
FLAG_REWARD_LOG2_F32 = 64
FLAG_TILE_SMALL = 128
FLAG_TILE_LARGE = 256
FLAG_SINGLE_CHAIN = 512
FLAG_NO_DIRECT = 1024

LIB_PATH = Path(__file__).resolve().parent / 'lib' / 'libenv2048.so'

_c_int = ctypes.c_int
_u32 = ctypes.c_uint32
_u64 = ctypes.c_uint64
_ptr = ctypes.c_void_p



class StepArgs(ctypes.Structure):
    """``env2048_step_args`` of include/env2048.h: one env2048_step call as a record (env2048_step_many)."""

    _fields_ = [('n', _u64), ('env_base', _u64), ('boards', _ptr), ('actions', _ptr), ('reward', _ptr), ('done', _ptr),
                ('legal', _ptr), ('obs', _ptr), ('seed', _u64), ('step_index', _u64), ('flags', _u32), ('reserved', _u32),
                ('max_exp', _ptr), ('step_counter', _ptr)]


class Episode(ctypes.Structure):
    """``env2048_episode`` of include/env2048.h (24 bytes): one record of env2048_segment_episodes."""

    _fields_ = [('env', _u32), ('start', _u32), ('length', _u32), ('terminated', ctypes.c_uint8), ('max_exp', ctypes.c_uint8),
                ('reserved', ctypes.c_uint16), ('total_reward', ctypes.c_double)]


class HelperInfo(ctypes.Structure):
    """``env2048_helper_info`` of include/env2048.h: what the library's helper thread of one device has cost so far."""

    _fields_ = [('created', _u32), ('spin_us', _u32), ('jobs', _u64), ('cpu_ns', _u64)]


class DirectPlanInfo(ctypes.Structure):
    """``env2048_direct_plan_info`` of include/env2048.h."""

    _fields_ = [('eligible', _u32), ('chains', _u32), ('part_first', _u64 * 3), ('part_blocks', _u32 * 2), ('part_boards_per_lane', _u32 * 2),
                ('segments', _u32), ('segment_steps', _u32)]


class DirectInfo(ctypes.Structure):
    """``env2048_direct_info`` of include/env2048.h: the direct-dispatch route of env2048_step_sequence on one device."""

    _fields_ = [('available', _u32), ('last_chains', _u32), ('calls', _u64), ('steps', _u64), ('packets', _u64), ('declined', _u64),
                ('reruns', _u64), ('last_gpu_ns', _u64), ('reason', ctypes.c_char * 160)]


EPISODES_FROZEN = 1
EPISODES_REWARD_F32 = 2
ROLLOUT_STATS_WORDS = 40
SIGNAL_MAX_ENVS = 512
EXPAND_OFFSETS_BLOCK = 512
INLINE_MAX_ENVS = 64
NXN_MAX_SIZE = 8

# name -> (restype, argtypes): exactly the prototypes of include/env2048.h
PROTOTYPES = {
    'env2048_abi_version': (_c_int, []),
    'env2048_last_error': (ctypes.c_char_p, []),
    'env2048_device_count': (_c_int, []),
    'env2048_reset': (_c_int, [_c_int, _ptr, _u64, _u64, _ptr, _ptr, _u64, _u64]),
    'env2048_step': (_c_int, [_c_int, _ptr, _u64, _u64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _u64, _u64, _u32]),
    'env2048_step_many': (_c_int, [_c_int, _ptr, _u64, ctypes.POINTER(StepArgs)]),
    'env2048_step_plan': (_c_int, [_u64, _u32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    'env2048_step_sequence_plan': (_c_int, [_u64, _u32, _u32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                            ctypes.POINTER(ctypes.c_int32)]),
    'env2048_step_many_plan': (_c_int, [_u64, ctypes.POINTER(StepArgs), ctypes.POINTER(_u32), ctypes.POINTER(ctypes.c_int32),
                                        ctypes.POINTER(ctypes.c_int32)]),
    'env2048_advance_counter': (_c_int, [_c_int, _ptr, _ptr, _u64]),
    'env2048_step_sequence': (_c_int, [_c_int, _ptr, _u64, _u64, _ptr, ctypes.POINTER(_ptr), _ptr, _ptr, _ptr, _ptr, _u64, _u64, _u64,
                                        _u32, _u32]),
    'env2048_helper_stats': (_c_int, [_c_int, ctypes.POINTER(HelperInfo)]),
    'env2048_direct_stats': (_c_int, [_c_int, ctypes.POINTER(DirectInfo)]),
    'env2048_direct_eligible': (_c_int, [_u64, _u32, _u32, ctypes.c_int32]),
    'env2048_direct_plan': (_c_int, [_u64, _u32, _u32, ctypes.c_int32, ctypes.POINTER(DirectPlanInfo)]),
    'env2048_step_state_i32': (_c_int, [_c_int, _ptr, _u64, _u64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _u64, _u64]),
    'env2048_step_state_i32_to': (_c_int, [_c_int, _ptr, _u64, _u64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _u64,
                                            _u64]),
    'env2048_step_state_i32_keyed': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                               _ptr]),
    'env2048_reset_keyed_i32': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr]),
    'env2048_step_state_i32_keyed_status': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                                      _ptr, _ptr, _ptr, _ptr]),
    'env2048_reset_keyed_i32_status': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr]),
    'env2048_reset_state_i32_keyed': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, ctypes.c_int64]),
    'env2048_step_state_i32_keyed_mirrored': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                                        _ptr, _ptr, _ptr, _ptr, ctypes.c_int64]),
    'env2048_reset_state_i32_keyed_signalled': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, ctypes.c_int64, _ptr,
                                                          _u64]),
    'env2048_step_state_i32_keyed_signalled': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                                         _ptr, _ptr, _ptr, _ptr, ctypes.c_int64, _ptr, _u64]),
    'env2048_wait_sequence': (_c_int, [_ptr, _u64, _u32]),
    'env2048_stream_signal': (_c_int, [_c_int, _ptr, _ptr, _u64]),
    'env2048_step_state_i32_keyed_inline': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                                      _ptr, _ptr, _ptr, _ptr, ctypes.c_int64, _ptr, _u64]),
    'env2048_status_i32': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr]),
    'env2048_afterstate_i32': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _u32, _ptr, _ptr, _ptr]),
    'env2048_rollout': (_c_int, [_c_int, _ptr, _u64, _u64, _ptr, _u64, _u64, _u32, _ptr, _ptr, _ptr, _u32]),
    'env2048_replay': (_c_int, [_c_int, _ptr, _u64, _u64, _ptr, _ptr, _ptr, _ptr, _u64, _u64, _u64, _u32, _u32, _ptr, _ptr, _ptr]),
    'env2048_rollout_episodes': (_c_int, [_c_int, _ptr, _u64, _u64, _ptr, _u64, _u64, _u32, _u32, _ptr, _ptr, _ptr]),
    'env2048_count_episodes': (_c_int, [_c_int, _ptr, _u64, _u64, _u32, _u32, _u32, _ptr, _u32, _ptr]),
    'env2048_segment_episodes': (_c_int, [_c_int, _ptr, _u64, _u64, _u32, _u32, _u32, _ptr, _ptr, _ptr, _u32, _ptr, _ptr, _ptr]),
    'env2048_afterstate': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'env2048_afterstates4': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr]),
    'env2048_spawn': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr]),
    'env2048_spawn_random': (_c_int, [_c_int, _ptr, _u64, _u64, _ptr, _u64, _u64, _u32]),
    'env2048_random_actions': (_c_int, [_c_int, _ptr, _u64, _u64, _u64, _u64, _ptr]),
    'env2048_chance_outcomes': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr]),
    'env2048_expectimax_ply': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'env2048_expectimax_depth2': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'env2048_expand_count': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr]),
    'env2048_expand_scratch_words': (_u64, [_u64]),
    'env2048_expand_offsets': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr, _u64]),
    'env2048_expand': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'env2048_expectimax_depth2_f64': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr]),
    'env2048_expectimax_backup': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'env2048_nxn_afterstate': (_c_int, [_c_int, _ptr, _u64, _u32, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'env2048_nxn_step': (_c_int, [_c_int, _ptr, _u64, _u32, _u64, _ptr, _ptr, _ptr, _ptr, _ptr, _u64, _u64, _u32]),
    'env2048_nxn_status': (_c_int, [_c_int, _ptr, _u64, _u32, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'env2048_nxn_reset': (_c_int, [_c_int, _ptr, _u64, _u32, _u64, _ptr, _ptr, _u64, _u64]),
    'env2048_nxn_spawn_random': (_c_int, [_c_int, _ptr, _u64, _u32, _u64, _ptr, _u64, _u64, _u32]),
    'env2048_nxn_chance_outcomes': (_c_int, [_c_int, _ptr, _u64, _u32, _ptr, _ptr, _ptr]),
    'env2048_status': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    'env2048_encode_onehot': (_c_int, [_c_int, _ptr, _u64, _ptr, _u32, _ptr]),
    'env2048_pack_i32': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr]),
    'env2048_unpack_i32': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr]),
    'env2048_pack_i64': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr]),
    'env2048_unpack_i64': (_c_int, [_c_int, _ptr, _u64, _ptr, _ptr]),
}


class Env2048Error(RuntimeError):
    """A C-ABI call returned a negative code; ``code`` holds it, the message is env2048_last_error()."""

    def __init__(self, code: int, message: str):
        super().__init__(f'libenv2048 error {code}: {message}')
        self.code = code


class Library:
    """Loaded ``libenv2048.so`` with typed entry points; every compute call raises on failure."""

    def __init__(self, path: str | Path | None = None):
        self.path = Path(path) if path else LIB_PATH
        if not self.path.exists():
            raise FileNotFoundError(
                f'{self.path} not found: build it with `python -m simulate_2048_amd.build` '
                '(needs hipcc; there is no CPU fallback)'
            )
        self._dll = ctypes.CDLL(str(self.path))
        for name, (restype, argtypes) in PROTOTYPES.items():
            fn = getattr(self._dll, name)  # AttributeError if the .so does not export it
            fn.restype = restype
            fn.argtypes = argtypes
        version = self._dll.env2048_abi_version()
        if version != ABI_VERSION:
            raise RuntimeError(f'{self.path}: ABI version {version}, this binding expects {ABI_VERSION}')

    # -- plumbing -----------------------------------------------------------------------------
    def last_error(self) -> str:
        return (self._dll.env2048_last_error() or b'').decode('utf-8', 'replace')

    def _check(self, rc: int) -> None:
        if rc != OK:
            raise Env2048Error(rc, self.last_error())

    def abi_version(self) -> int:
        return self._dll.env2048_abi_version()

    def device_count(self) -> int:
        """Number of visible HIP devices; raises Env2048Error when the HIP runtime is unusable."""
        rc = self._dll.env2048_device_count()
        if rc < 0:
            raise Env2048Error(rc, self.last_error())
        return rc

    # -- compute entry points (raw pointers in, nothing out; asynchronous on `stream`) --------
    def reset(self, device, stream, n, env_base, boards, done, seed, step_index) -> None:
        self._check(self._dll.env2048_reset(device, stream, n, env_base, boards, done, seed, step_index))

    def step(self, device, stream, n, env_base, boards, actions, reward, done, legal, obs, seed, step_index,
             flags) -> None:
        self._check(self._dll.env2048_step(device, stream, n, env_base, boards, actions, reward, done, legal, obs,
                                           seed, step_index, flags))

    def step_many(self, device, stream, records) -> None:
        """``records``: a ctypes array of StepArgs (build it once with ``(StepArgs * k)(...)`` and reuse it)."""
        self._check(self._dll.env2048_step_many(device, stream, len(records), records))

    def step_plan(self, n: int, flags: int = 0) -> dict:
        """How env2048_step would launch n boards: {'boards_per_lane': 1 | 4, 'write_through': bool}. No device needed."""
        bpl, through = ctypes.c_int32(), ctypes.c_int32()
        self._check(self._dll.env2048_step_plan(n, flags, ctypes.byref(bpl), ctypes.byref(through)))
        return {'boards_per_lane': bpl.value, 'write_through': bool(through.value)}

    def step_sequence_plan(self, n: int, count: int, flags: int = 0) -> dict:
        """How env2048_step_sequence would run: {'chains': 1 | 2, 'boards_per_lane', 'write_through'}. No device needed."""
        chains, bpl, through = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        self._check(self._dll.env2048_step_sequence_plan(n, count, flags, ctypes.byref(chains), ctypes.byref(bpl), ctypes.byref(through)))
        return {'chains': chains.value, 'boards_per_lane': bpl.value, 'write_through': bool(through.value)}

    def step_many_plan(self, records) -> dict:
        """How env2048_step_many would run ``records``: {'launches': L, 'launch_of': [...], 'boards_per_lane': [...],
        'write_through': [...]} -- records with the same ``launch_of`` value share one kernel launch. No device needed."""
        count = len(records)
        launch_of = (_u32 * count)()
        bpl = (ctypes.c_int32 * count)()
        through = (ctypes.c_int32 * count)()
        rc = self._dll.env2048_step_many_plan(count, records, launch_of, bpl, through)
        if rc < 0:
            raise Env2048Error(rc, self.last_error())
        return {'launches': rc, 'launch_of': list(launch_of), 'boards_per_lane': list(bpl), 'write_through': [bool(x) for x in through]}

    def advance_counter(self, device, stream, counter, delta) -> None:
        self._check(self._dll.env2048_advance_counter(device, stream, counter, delta))

    def step_sequence(self, device, stream, n, env_base, boards, actions, reward, done, legal, obs, row_stride, seed, step_index,
                      count, flags) -> None:
        """``actions``: None or a ctypes array ``(c_void_p * count)`` of device pointers, one per step."""
        self._check(self._dll.env2048_step_sequence(device, stream, n, env_base, boards, actions, reward, done, legal, obs,
                                                    row_stride, seed, step_index, count, flags))

    def direct_stats(self, device: int) -> dict:
        """The direct-dispatch route of env2048_step_sequence on ``device`` (started up by this call if it is its first use):
        {'available', 'reason', 'calls', 'steps', 'packets', 'declined', 'reruns', 'last_chains', 'last_gpu_ns'}."""
        info = DirectInfo()
        self._check(self._dll.env2048_direct_stats(device, ctypes.byref(info)))
        return {'available': bool(info.available), 'reason': info.reason.decode(errors='replace'), 'calls': int(info.calls), 'steps': int(info.steps),
                'packets': int(info.packets), 'declined': int(info.declined), 'reruns': int(info.reruns), 'last_chains': int(info.last_chains),
                'last_gpu_ns': int(info.last_gpu_ns)}

    def direct_eligible(self, n: int, count: int, flags: int, extras: bool = False) -> bool:
        """Would env2048_step_sequence offer a call of this shape to the direct route?  (sizes and flags only; no device call)"""
        rc = self._dll.env2048_direct_eligible(n, count, flags, 1 if extras else 0)
        if rc < 0:
            raise Env2048Error(rc, self.last_error())
        return rc == 1

    def direct_plan(self, n: int, count: int, flags: int, extras: bool = False) -> dict:
        """How the direct route would run a sequence: {'eligible', 'chains', 'parts': [(first board, boards, workgroups, boards per lane)],
        'segments', 'segment_steps'}.  No device call."""
        info = DirectPlanInfo()
        self._check(self._dll.env2048_direct_plan(n, count, flags, 1 if extras else 0, ctypes.byref(info)))
        parts = [(int(info.part_first[c]), int(info.part_first[c + 1] - info.part_first[c]), int(info.part_blocks[c]), int(info.part_boards_per_lane[c]))
                 for c in range(info.chains)]
        return {'eligible': bool(info.eligible), 'chains': int(info.chains), 'parts': parts, 'segments': int(info.segments),
                'segment_steps': int(info.segment_steps)}

    def helper_stats(self, device: int) -> dict:
        """The helper thread of ``device`` (env2048_step_sequence's second chain): {'created', 'spin_us', 'jobs', 'cpu_ns'}.
        No device call."""
        info = HelperInfo()
        self._check(self._dll.env2048_helper_stats(device, ctypes.byref(info)))
        return {'created': bool(info.created), 'spin_us': info.spin_us, 'jobs': info.jobs, 'cpu_ns': info.cpu_ns}

    def step_state_i32(self, device, stream, n, env_base, boards, actions, step_count, done_io, total_reward, reward, max_tile,
                       seed, step_index) -> None:
        self._check(self._dll.env2048_step_state_i32(device, stream, n, env_base, boards, actions, step_count, done_io,
                                                     total_reward, reward, max_tile, seed, step_index))

    def step_state_i32_to(self, device, stream, n, env_base, boards_in, boards_out, actions, step_count_in, step_count_out,
                          done_in, done_out, total_in, total_out, reward, max_tile, seed, step_index) -> None:
        self._check(self._dll.env2048_step_state_i32_to(device, stream, n, env_base, boards_in, boards_out, actions, step_count_in,
                                                        step_count_out, done_in, done_out, total_in, total_out, reward, max_tile,
                                                        seed, step_index))

    def step_state_i32_keyed(self, device, stream, n, keys, boards_in, boards_out, actions, step_count_in, step_count_out,
                             done_in, done_out, total_in, total_out, reward, max_tile) -> None:
        self._check(self._dll.env2048_step_state_i32_keyed(device, stream, n, keys, boards_in, boards_out, actions, step_count_in,
                                                           step_count_out, done_in, done_out, total_in, total_out, reward,
                                                           max_tile))

    def reset_keyed_i32(self, device, stream, n, keys, boards) -> None:
        self._check(self._dll.env2048_reset_keyed_i32(device, stream, n, keys, boards))

    def step_state_i32_keyed_status(self, device, stream, n, keys, boards_in, boards_out, actions, step_count_in, step_count_out,
                                    done_in, done_out, total_in, total_out, reward, max_tile, obs, legal4) -> None:
        self._check(self._dll.env2048_step_state_i32_keyed_status(device, stream, n, keys, boards_in, boards_out, actions,
                                                                  step_count_in, step_count_out, done_in, done_out, total_in,
                                                                  total_out, reward, max_tile, obs, legal4))

    def reset_keyed_i32_status(self, device, stream, n, keys, boards, obs, legal4) -> None:
        self._check(self._dll.env2048_reset_keyed_i32_status(device, stream, n, keys, boards, obs, legal4))

    def reset_state_i32_keyed(self, device, stream, n, keys, boards, step_count, done, total, obs, legal4, host_mirror_delta=0) -> None:
        self._check(self._dll.env2048_reset_state_i32_keyed(device, stream, n, keys, boards, step_count, done, total, obs, legal4,
                                                            host_mirror_delta))

    def step_state_i32_keyed_mirrored(self, device, stream, n, keys, boards_in, boards_out, actions, step_count_in, step_count_out,
                                      done_in, done_out, total_in, total_out, reward, max_tile, obs, legal4, host_mirror_delta=0) -> None:
        self._check(self._dll.env2048_step_state_i32_keyed_mirrored(device, stream, n, keys, boards_in, boards_out, actions,
                                                                    step_count_in, step_count_out, done_in, done_out, total_in,
                                                                    total_out, reward, max_tile, obs, legal4, host_mirror_delta))

    def reset_state_i32_keyed_signalled(self, device, stream, n, keys, boards, step_count, done, total, obs, legal4, host_mirror_delta,
                                        host_sequence, sequence_value) -> None:
        self._check(self._dll.env2048_reset_state_i32_keyed_signalled(device, stream, n, keys, boards, step_count, done, total, obs, legal4,
                                                                      host_mirror_delta, host_sequence, sequence_value))

    def step_state_i32_keyed_signalled(self, device, stream, n, keys, boards_in, boards_out, actions, step_count_in, step_count_out,
                                       done_in, done_out, total_in, total_out, reward, max_tile, obs, legal4, host_mirror_delta,
                                       host_sequence, sequence_value) -> None:
        self._check(self._dll.env2048_step_state_i32_keyed_signalled(device, stream, n, keys, boards_in, boards_out, actions,
                                                                     step_count_in, step_count_out, done_in, done_out, total_in,
                                                                     total_out, reward, max_tile, obs, legal4, host_mirror_delta,
                                                                     host_sequence, sequence_value))

    def step_state_i32_keyed_inline(self, device, stream, n, host_keys, boards_in, boards_out, host_actions, step_count_in, step_count_out,
                                    done_in, done_out, total_in, total_out, reward, max_tile, obs, legal4, host_mirror_delta=0,
                                    host_sequence=None, sequence_value=0) -> None:
        """``host_keys`` / ``host_actions``: HOST addresses (read during the call); n <= INLINE_MAX_ENVS."""
        self._check(self._dll.env2048_step_state_i32_keyed_inline(device, stream, n, host_keys, boards_in, boards_out, host_actions,
                                                                  step_count_in, step_count_out, done_in, done_out, total_in, total_out,
                                                                  reward, max_tile, obs, legal4, host_mirror_delta, host_sequence,
                                                                  sequence_value))

    def stream_signal(self, device, stream, host_word, value) -> None:
        self._check(self._dll.env2048_stream_signal(device, stream, host_word, value))

    def wait_sequence(self, host_sequence, value, timeout_us) -> bool:
        """Poll the completion word of a *_signalled launch: True once it holds ``value``, False after ``timeout_us``."""
        rc = self._dll.env2048_wait_sequence(host_sequence, value, timeout_us)
        if rc < 0:
            raise Env2048Error(rc, self.last_error())
        return rc == 1

    def random_actions(self, device, stream, n, env_base, seed, step_index, actions) -> None:
        self._check(self._dll.env2048_random_actions(device, stream, n, env_base, seed, step_index, actions))

    def expectimax_depth2(self, device, stream, n, boards, terms, q, best_action, value) -> None:
        self._check(self._dll.env2048_expectimax_depth2(device, stream, n, boards, terms, q, best_action, value))

    def expand_count(self, device, stream, n, boards, counts) -> None:
        self._check(self._dll.env2048_expand_count(device, stream, n, boards, counts))

    def expand_scratch_words(self, n) -> int:
        """uint64 words of the `scratch` argument of ``expand_offsets`` for n boards (the library's block size decides)."""
        return int(self._dll.env2048_expand_scratch_words(n))

    def expand_offsets(self, device, stream, n, boards, offsets, scratch, host_total=None, sequence_value=0) -> None:
        self._check(self._dll.env2048_expand_offsets(device, stream, n, boards, offsets, scratch, host_total, sequence_value))

    def expand(self, device, stream, n, boards, offsets, children, reward, n_empty) -> None:
        self._check(self._dll.env2048_expand(device, stream, n, boards, offsets, children, reward, n_empty))

    def expectimax_depth2_f64(self, device, stream, n, boards, value) -> None:
        self._check(self._dll.env2048_expectimax_depth2_f64(device, stream, n, boards, value))

    def expectimax_backup(self, device, stream, n, reward, n_empty, offsets, child_value, value, q, best_action) -> None:
        self._check(self._dll.env2048_expectimax_backup(device, stream, n, reward, n_empty, offsets, child_value, value, q, best_action))

    def status_i32(self, device, stream, n, boards, obs, legal4, done) -> None:
        self._check(self._dll.env2048_status_i32(device, stream, n, boards, obs, legal4, done))

    def afterstate_i32(self, device, stream, n, boards_in, actions, action, boards_out, reward, changed) -> None:
        self._check(self._dll.env2048_afterstate_i32(device, stream, n, boards_in, actions, action, boards_out, reward, changed))

    def rollout(self, device, stream, n, env_base, boards, seed, step0, T, return_sum, episodes, valid_moves,
                flags) -> None:
        self._check(self._dll.env2048_rollout(device, stream, n, env_base, boards, seed, step0, T, return_sum,
                                              episodes, valid_moves, flags))

    def replay(self, device, stream, n, env_base, boards, actions, reward, done, row_stride, seed, step0, T, flags, return_sum,
               episodes, valid_moves) -> None:
        self._check(self._dll.env2048_replay(device, stream, n, env_base, boards, actions, reward, done, row_stride, seed, step0, T,
                                             flags, return_sum, episodes, valid_moves))

    def rollout_episodes(self, device, stream, n, env_base, boards, seed, step0, T, flags, ep_length, ep_return, stats) -> None:
        self._check(self._dll.env2048_rollout_episodes(device, stream, n, env_base, boards, seed, step0, T, flags, ep_length,
                                                       ep_return, stats))

    def count_episodes(self, device, stream, n, row_stride, rows, first_row, ring_rows, done, flags, counts) -> None:
        self._check(self._dll.env2048_count_episodes(device, stream, n, row_stride, rows, first_row, ring_rows, done, flags, counts))

    def segment_episodes(self, device, stream, n, row_stride, rows, first_row, ring_rows, done, reward, max_exp, flags, offsets,
                         episodes, episode_of) -> None:
        self._check(self._dll.env2048_segment_episodes(device, stream, n, row_stride, rows, first_row, ring_rows, done, reward,
                                                       max_exp, flags, offsets, episodes, episode_of))

    def afterstate(self, device, stream, n, boards_in, actions, boards_out, reward, changed) -> None:
        self._check(self._dll.env2048_afterstate(device, stream, n, boards_in, actions, boards_out, reward, changed))

    def afterstates4(self, device, stream, n, boards, out_boards, out_reward, out_legal) -> None:
        self._check(self._dll.env2048_afterstates4(device, stream, n, boards, out_boards, out_reward, out_legal))

    def spawn(self, device, stream, n, boards, spawn_k, spawn_e, placed) -> None:
        self._check(self._dll.env2048_spawn(device, stream, n, boards, spawn_k, spawn_e, placed))

    def spawn_random(self, device, stream, n, env_base, boards, seed, step_index, count) -> None:
        self._check(self._dll.env2048_spawn_random(device, stream, n, env_base, boards, seed, step_index, count))

    def chance_outcomes(self, device, stream, n, boards, out_boards, out_prob, out_count) -> None:
        self._check(self._dll.env2048_chance_outcomes(device, stream, n, boards, out_boards, out_prob, out_count))

    def expectimax_ply(self, device, stream, n, boards, succ_boards, succ_prob, succ_status, succ_count, after_reward, legal) -> None:
        self._check(self._dll.env2048_expectimax_ply(device, stream, n, boards, succ_boards, succ_prob, succ_status, succ_count,
                                                     after_reward, legal))

    # -- boards of any size S x S (one byte per cell): the general form behind the mirrors for size != 4 ------------------------
    def nxn_afterstate(self, device, stream, n, size, boards_in, actions, boards_out, reward, changed) -> None:
        self._check(self._dll.env2048_nxn_afterstate(device, stream, n, size, boards_in, actions, boards_out, reward, changed))

    def nxn_step(self, device, stream, n, size, env_base, boards, actions, reward, done, legal, seed, step_index, flags) -> None:
        self._check(self._dll.env2048_nxn_step(device, stream, n, size, env_base, boards, actions, reward, done, legal, seed, step_index, flags))

    def nxn_status(self, device, stream, n, size, boards, done, legal, n_empty, max_exp) -> None:
        self._check(self._dll.env2048_nxn_status(device, stream, n, size, boards, done, legal, n_empty, max_exp))

    def nxn_reset(self, device, stream, n, size, env_base, boards, done, seed, step_index) -> None:
        self._check(self._dll.env2048_nxn_reset(device, stream, n, size, env_base, boards, done, seed, step_index))

    def nxn_spawn_random(self, device, stream, n, size, env_base, boards, seed, step_index, count) -> None:
        self._check(self._dll.env2048_nxn_spawn_random(device, stream, n, size, env_base, boards, seed, step_index, count))

    def nxn_chance_outcomes(self, device, stream, n, size, boards, out_boards, out_count) -> None:
        self._check(self._dll.env2048_nxn_chance_outcomes(device, stream, n, size, boards, out_boards, out_count))

    def encode_onehot(self, device, stream, n, boards, width, out) -> None:
        self._check(self._dll.env2048_encode_onehot(device, stream, n, boards, width, out))

    def status(self, device, stream, n, boards, done, legal, obs, max_exp, n_empty) -> None:
        self._check(self._dll.env2048_status(device, stream, n, boards, done, legal, obs, max_exp, n_empty))

    def pack_i32(self, device, stream, n, values, boards) -> None:
        self._check(self._dll.env2048_pack_i32(device, stream, n, values, boards))

    def unpack_i32(self, device, stream, n, boards, values) -> None:
        self._check(self._dll.env2048_unpack_i32(device, stream, n, boards, values))

    def pack_i64(self, device, stream, n, values, boards) -> None:
        self._check(self._dll.env2048_pack_i64(device, stream, n, values, boards))

    def unpack_i64(self, device, stream, n, boards, values) -> None:
        self._check(self._dll.env2048_unpack_i64(device, stream, n, boards, values))


_default: Library | None = None


def library() -> Library:
    """Process-wide default Library (loads simulate_2048_amd/lib/libenv2048.so on first use)."""
    global _default
    if _default is None:
        _default = Library()
    return _default
