"""
Single-board calls of the API mirrors (``twentyfortyeight.core.gameboard`` / ``gamemove`` / the
``TwentyFortyEight`` env): one kernel launch and one stream synchronisation per call.

The kernels behind the C-ABI take *device-accessible* pointers.  Pinned host memory is
device-accessible on HIP (unified addressing), so for a batch of one the operands live in a small set
of pinned host buffers that the kernel reads and writes across the bus directly: no staging copies,
no device allocations and no torch operators on the call path -- the Python side fills NumPy views of
those buffers, launches, waits for the stream and reads the views.  Tile values <-> log2 exponents are
converted on the host (16 integers).

Thread-safe: every host thread gets its own scratch set (created on its first call, ~1 KiB of pinned memory), so two
threads stepping their own ``TwentyFortyEight`` objects never share an operand buffer.  (The reference's unseeded
draws share one module-level generator; here they share one atomic counter of a Philox stream, ``runtime.entropy_stream``.)
With an injected host library (tests) the buffers are ordinary host memory and the same code runs.
"""

from __future__ import annotations

import threading

import numpy as np
import torch

from . import capi, runtime


_tables: dict = {}


def _value_table(dtype) -> np.ndarray:
    """exponent byte -> tile value of `dtype` for all 256 bytes (2^e, 0 for e = 0: the mapping of env2048_unpack_i64); one gather
    converts a board back instead of four array operations."""
    key = np.dtype(dtype)
    table = _tables.get(key)
    if table is None:
        e = np.arange(256, dtype=np.int64)
        with np.errstate(all='ignore'):
            table = np.where(e > 0, np.left_shift(np.int64(1), e), 0).astype(key)
        _tables[key] = table
    return table


def _exponents_into(board: np.ndarray, values) -> int:
    """Tile values -> log2 exponents (e = floor(log2(v)) for v > 0, else 0: the mapping of env2048_pack_i64) into the first
    cells of `board`; returns the number of cells.  Up to 16 cells a Python loop is the faster way; above that one frexp over the
    array (the binary exponent, exact for integers below 2^53, minus one)."""
    flat = np.asarray(values).reshape(-1)
    cells = flat.shape[0]
    if cells > 16 and flat.dtype.kind in 'iu':
        e = np.frexp(np.maximum(flat, 0))[1]
        np.subtract(e, 1, out=e)
        np.maximum(e, 0, out=e)
        board[:cells] = e
    else:
        for i, v in enumerate(flat.tolist()):
            board[i] = (int(v).bit_length() - 1) & 0xFF if v > 0 else 0
    return cells


class ScalarOps:
    """Scratch buffers + the single-board operations, bound to one ``runtime.Runtime``."""

    def __init__(self, rt: runtime.Runtime):
        self.rt = rt
        pin = rt.device.type == 'cuda'

        def buf(shape, dtype):
            t = torch.zeros(shape, dtype=dtype, pin_memory=pin)
            return t, t.numpy(), t.data_ptr()

        self._board_t, self.board, self._board = buf((16,), torch.uint8)
        self._out_t, self.out, self._out = buf((16,), torch.uint8)
        self._outcomes_t, self.outcomes, self._outcomes = buf((32, 16), torch.uint8)
        self._prob_t, self.prob, self._prob = buf((32,), torch.float32)
        self._action_t, self.action, self._action = buf((1,), torch.uint8)
        self._reward_t, self.reward, self._reward = buf((1,), torch.int32)
        self._flags_t, self.flags, self._flags_ptr = buf((4,), torch.uint8)  # done, legal, changed, count
        self._ptr_done = self._flags_ptr
        self._ptr_legal = self._flags_ptr + 1
        self._ptr_changed = self._flags_ptr + 2
        self._ptr_count = self._flags_ptr + 3
        # boards of other sizes than 4 x 4 (one byte per cell, up to 8 x 8): created on first use
        self._nxn = None
        # completion word of this thread's calls (see _sync)
        self._word_t, self._word, self._word_ptr = buf((2,), torch.int64)
        self._serial = 0
        self._signal_ok = pin

    # -- host-side value <-> exponent conversion (same mapping as env2048_pack_i64 / unpack_i64) -----
    def load(self, values) -> None:
        """(4,4) tile values -> the packed input buffer: e = floor(log2(v)) for v > 0, else 0."""
        _exponents_into(self.board, np.asarray(values).reshape(16))

    @staticmethod
    def values(packed: np.ndarray, dtype) -> np.ndarray:
        """packed exponents (..., 16) -> tile values (..., 4, 4) of `dtype`."""
        return _value_table(dtype)[packed].reshape(packed.shape[:-1] + (4, 4))

    def _sync(self) -> None:
        """Wait for the call's one launch.  A stream synchronisation is an interrupt-driven wake-up (11-12 us around a 4 us
        kernel); instead the device is asked to store a serial into a pinned host word once the stream has drained
        (env2048_stream_signal: one stream memory operation) and the host polls that word (launch + signal + wait 12.7 us
        against 16.3 us, profiles/r04_i_move_latency_breakdown.json).  Falls back to synchronising when the word does not
        show up within 200 us, and for good when the stream operation is not available."""
        if self.rt.device.type != 'cuda':
            return
        if self._signal_ok:
            try:
                self._serial += 1
                dev, stream = self.rt.handles()
                self.rt.lib.stream_signal(dev, stream, self._word_ptr, self._serial)
                if self.rt.lib.wait_sequence(self._word_ptr, self._serial, 200):
                    return
            except (capi.Env2048Error, AttributeError):
                self._signal_ok = False
        torch.cuda.current_stream(self.rt.device).synchronize()

    # -- operations ----------------------------------------------------------------------------------
    def afterstate(self, values, action: int, dtype):
        """-> (afterstate values, reward int, changed bool); no tile is spawned."""
        self.load(values)
        self.action[0] = action & 3
        dev, stream = self.rt.handles()
        self.rt.lib.afterstate(dev, stream, 1, self._board, self._action, self._out, self._reward, self._ptr_changed)
        self._sync()
        return self.values(self.out, dtype), int(self.reward[0]), bool(self.flags[2])

    def step(self, values, action: int, seed: int, step_index: int, dtype, onehot_width: int = 0):
        """Full transition -> (new values, reward int, done bool). The input array is not touched.  With `onehot_width` a
        second launch behind the first encodes the NEW board (the exponents the step has just written) as
        eye(width)[log2] rows and both are waited for once: -> (..., one-hot (16, width) bytes)."""
        self.load(values)
        self.action[0] = action & 3
        dev, stream = self.rt.handles()
        self.rt.lib.step(dev, stream, 1, 0, self._board, self._action, self._reward, self._ptr_done, None, None, seed,
                         step_index, 0)
        if onehot_width:
            out = self._nxn_buffers()['outcomes']
            self.rt.lib.encode_onehot(dev, stream, 1, self._board, int(onehot_width), out[2])
        self._sync()
        result = self.values(self.board, dtype), int(self.reward[0]), bool(self.flags[0])
        if onehot_width:
            return result + (out[1][:16 * onehot_width].reshape(16, onehot_width).copy(),)
        return result

    def status(self, values):
        """-> (done bool, legal bitmask int)."""
        self.load(values)
        dev, stream = self.rt.handles()
        self.rt.lib.status(dev, stream, 1, self._board, self._ptr_done, self._ptr_legal, None, None, None)
        self._sync()
        return bool(self.flags[0]), int(self.flags[1])

    def spawn_random(self, values, seed: int, step_index: int, count: int, dtype) -> np.ndarray:
        self.load(values)
        dev, stream = self.rt.handles()
        self.rt.lib.spawn_random(dev, stream, 1, 0, self._board, seed, step_index, count)
        self._sync()
        return self.values(self.board, dtype)

    def chance_outcomes(self, values, dtype):
        """-> (outcome values (count,4,4), count) in the reference's enumeration order."""
        self.load(values)
        dev, stream = self.rt.handles()
        self.rt.lib.chance_outcomes(dev, stream, 1, self._board, self._outcomes, self._prob, self._ptr_count)
        self._sync()
        count = int(self.flags[3])
        return self.values(self.outcomes[:count], dtype), count


    # -- boards of any size S x S, 2 <= S <= 8 (env2048_nxn_*): the same five operations, one byte per cell ------------------------
    def _nxn_buffers(self):
        if self._nxn is None:
            pin = self.rt.device.type == 'cuda'

            def buf(shape, dtype):
                t = torch.zeros(shape, dtype=dtype, pin_memory=pin)
                return t, t.numpy(), t.data_ptr()

            self._nxn = {'board': buf((64,), torch.uint8), 'out': buf((64,), torch.uint8), 'outcomes': buf((128 * 64,), torch.uint8)}
        return self._nxn

    @staticmethod
    def _size_of(values) -> int:
        arr = np.asarray(values)
        if arr.ndim != 2 or arr.shape[0] != arr.shape[1] or not 2 <= arr.shape[0] <= capi.NXN_MAX_SIZE:
            raise ValueError(f'board: the MI355X engine handles square boards of size 2 ... {capi.NXN_MAX_SIZE}, got shape {arr.shape}')
        return int(arr.shape[0])

    def _load_nxn(self, values) -> int:
        size = self._size_of(values)
        _exponents_into(self._nxn_buffers()['board'][1], values)
        return size

    @staticmethod
    def values_nxn(packed: np.ndarray, size: int, dtype) -> np.ndarray:
        return _value_table(dtype)[packed[..., :size * size]].reshape(packed.shape[:-1] + (size, size))

    def afterstate_nxn(self, values, action: int, dtype):
        size = self._load_nxn(values)
        nb = self._nxn_buffers()
        self.action[0] = action & 3
        dev, stream = self.rt.handles()
        self.rt.lib.nxn_afterstate(dev, stream, 1, size, nb['board'][2], self._action, nb['out'][2], self._reward, self._ptr_changed)
        self._sync()
        return self.values_nxn(nb['out'][1], size, dtype), int(self.reward[0]), bool(self.flags[2])

    def step_nxn(self, values, action: int, seed: int, step_index: int, dtype, onehot_width: int = 0):
        size = self._load_nxn(values)
        nb = self._nxn_buffers()
        self.action[0] = action & 3
        dev, stream = self.rt.handles()
        self.rt.lib.nxn_step(dev, stream, 1, size, 0, nb['board'][2], self._action, self._reward, self._ptr_done, None, seed, step_index, 0)
        cells = size * size
        if onehot_width:  # (as in step(): the one-hot rows of the new board from a second launch, one wait for both)
            records = (cells + 15) // 16
            nb['board'][1][cells:records * 16] = 0  # (whole 16-cell records for the encoder)
            self.rt.lib.encode_onehot(dev, stream, records, nb['board'][2], int(onehot_width), nb['outcomes'][2])
        self._sync()
        result = self.values_nxn(nb['board'][1], size, dtype), int(self.reward[0]), bool(self.flags[0])
        if onehot_width:
            return result + (nb['outcomes'][1][:records * 16 * onehot_width].reshape(records * 16, onehot_width)[:cells].copy(),)
        return result

    def status_nxn(self, values):
        size = self._load_nxn(values)
        dev, stream = self.rt.handles()
        self.rt.lib.nxn_status(dev, stream, 1, size, self._nxn_buffers()['board'][2], self._ptr_done, self._ptr_legal, None, None)
        self._sync()
        return bool(self.flags[0]), int(self.flags[1])

    def spawn_random_nxn(self, values, seed: int, step_index: int, count: int, dtype) -> np.ndarray:
        size = self._load_nxn(values)
        nb = self._nxn_buffers()
        dev, stream = self.rt.handles()
        self.rt.lib.nxn_spawn_random(dev, stream, 1, size, 0, nb['board'][2], seed, step_index, count)
        self._sync()
        return self.values_nxn(nb['board'][1], size, dtype)

    def chance_outcomes_nxn(self, values, dtype):
        size = self._load_nxn(values)
        nb = self._nxn_buffers()
        dev, stream = self.rt.handles()
        self.rt.lib.nxn_chance_outcomes(dev, stream, 1, size, nb['board'][2], nb['outcomes'][2], self._ptr_count)
        self._sync()
        count = int(self.flags[3])
        cells = size * size
        return self.values_nxn(nb['outcomes'][1][:2 * cells * cells].reshape(2 * cells, cells)[:count], size, dtype), count


    # -- one-hot observation of up to 64 cells (encode / encode_flatten of twentyfortyeight/utils/binary.py) ------------------------
    ONEHOT_MAX_CELLS = 64
    ONEHOT_MAX_WIDTH = 64

    def onehot(self, values, width: int) -> np.ndarray:
        """(cells,) tile values -> (cells, width) 0/1 bytes: eye(width)[log2(value)] with 0 staying index 0, computed by
        env2048_encode_onehot on pinned host operands (one launch + one polled completion word, like the other single-board
        calls).  `cells` is padded to whole 16-cell records for the kernel and trimmed again."""
        nb = self._nxn_buffers()
        board = nb['board'][1]
        board[:] = 0
        cells = _exponents_into(board, values)
        records = (cells + 15) // 16
        dev, stream = self.rt.handles()
        self.rt.lib.encode_onehot(dev, stream, records, nb['board'][2], int(width), nb['outcomes'][2])
        self._sync()
        return nb['outcomes'][1][:records * 16 * width].reshape(records * 16, width)[:cells].copy()


_per_thread = threading.local()


def current() -> ScalarOps:
    """The calling thread's ScalarOps for the current runtime (created on first use, replaced when the runtime changes)."""
    rt = runtime.current()
    ops = getattr(_per_thread, 'ops', None)
    if ops is None or ops.rt is not rt:
        ops = ScalarOps(rt)
        _per_thread.ops = ops
    return ops


__all__ = ['ScalarOps', 'current', 'capi']
