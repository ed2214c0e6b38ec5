"""
Allocation-free results and inputs for the ``reinforce.game`` mirrors at the reference's batch sizes (N = 2 ... 8,
reinforce/training/config.py:79,151,177), where a move's kernel takes 4 microseconds and everything else is host time.

``ResultSet``  -- ALL outputs of one ``batched_step`` / ``batched_reset`` (board, step_count, done, total_reward, reward,
max_tile AND the observation / legal bytes of the new state) carved ONCE out of one device allocation and wrapped ONCE as
``DeviceArray`` objects.  A set is handed out again only when nobody else can still see it: every tensor object's Python
reference count is back at its idle value (the caller dropped the ``GameState``, the reward array ...) and the allocation's
storage use count is back at its idle value (no view such as ``states.board[0]`` is alive either).  That is the rule torch's
caching allocator applies to a freed block -- reuse in stream order once the last reference is gone -- applied one level up,
so that a move creates no tensor object at all.  A caller that keeps every state it ever received simply finds no idle set
and gets fresh ones (the pool is capped; beyond the cap sets are created and dropped as before).

``registry``   -- ``id(tensor) -> (set, pointer)`` for the tensors of live sets: the next ``batched_step`` recognises "its own"
state tensors without asking torch for dtype / device / contiguity, and ``batched_get_observation`` /
``batched_get_legal_actions`` return the stash written by the launch that produced the state (valid while torch's version
counters say neither the board nor the stash was modified in place).

``HostStager`` -- small host arrays (PRNG keys, actions) reach the kernel through a ring of PINNED host memory, which is
device-accessible on HIP: one NumPy copy into the ring instead of a pageable H2D copy (8-12 microseconds each at N = 8).  The
ring has two halves; the device is synchronised once whenever the writer moves into the other half (every 512 KiB of staged
inputs), so a kernel never reads a slot that is being refilled -- with no per-call synchronisation.
"""

from __future__ import annotations

import sys
import threading

import numpy as np
import torch

from .interop import DeviceArray

_use_count = getattr(torch._C, '_storage_Use_Count', None)  # references to a StorageImpl (tensors, views, wrappers)

MAX_SETS_PER_KEY = 6
MAX_KEYS = 16  # distinct (n, device, stream) keys kept; the least recently used shape's sets are dropped beyond that
STAGE_MAX_BYTES = 16 << 10  # larger host arrays take an ordinary H2D copy (the kernel would read them across the bus)
MIRROR_MAX_BYTES = 64 << 10  # result sets up to this size (N <= ~430) are read back through one pinned host mirror per launch


def _align(x: int, a: int = 16) -> int:
    return (x + a - 1) // a * a


class ResultSet:
    __slots__ = ('n', 'arena', 'storage', 'cdata', 'board', 'obs', 'step_count', 'max_tile', 'total', 'reward', 'legal', 'done',
                 'tensors', 'ptr', 'idle_refs', 'idle_use', 'version', 'pooled', 'leased', 'layout', 'serial',
                 'mirror', 'mirror_fields', 'mirror_key', 'mirror_copies', 'mirror_delta', 'direct_serial', 'stream')

    def __init__(self, n: int, device: torch.device):
        self.n = n
        sizes = (('board', 64 * n), ('obs', 64 * n), ('step_count', 4 * n), ('total', 4 * n), ('done', n), ('max_tile', 4 * n),
                 ('reward', 4 * n), ('legal', 4 * n))
        offsets, at = {}, 0
        for name, size in sizes:
            offsets[name] = at
            at = _align(at + size)
        arena = torch.empty((max(at, 16),), dtype=torch.uint8, device=device)
        self.arena = arena
        self.storage = arena.untyped_storage()
        self.cdata = self.storage._cdata
        base = arena.data_ptr()

        def carve(name, size, dtype, shape):
            o = offsets[name]
            return arena[o:o + size].view(dtype).view(shape).as_subclass(DeviceArray)

        self.board = carve('board', 64 * n, torch.int32, (n, 4, 4))
        self.obs = carve('obs', 64 * n, torch.float32, (n, 16))
        self.step_count = carve('step_count', 4 * n, torch.int32, (n,))
        self.total = carve('total', 4 * n, torch.float32, (n,))
        self.done = carve('done', n, torch.bool, (n,))
        self.max_tile = carve('max_tile', 4 * n, torch.int32, (n,))
        self.reward = carve('reward', 4 * n, torch.float32, (n,))
        self.legal = carve('legal', 4 * n, torch.bool, (n, 4))
        self.tensors = (self.board, self.obs, self.step_count, self.total, self.done, self.max_tile, self.reward, self.legal)
        self.ptr = {name: base + offsets[name] for name, _ in sizes}
        self.version = arena._version  # ONE counter: the carved tensors are views of the arena and share its version counter
        ts = self.tensors  # (measured exactly the way idle() measures: indexing adds no reference, a loop variable would)
        self.idle_refs = (sys.getrefcount(ts[0]), sys.getrefcount(ts[1]), sys.getrefcount(ts[2]), sys.getrefcount(ts[3]),
                          sys.getrefcount(ts[4]), sys.getrefcount(ts[5]), sys.getrefcount(ts[6]), sys.getrefcount(ts[7]))
        self.idle_use = _use_count(self.cdata) if _use_count else -1
        self.pooled = False
        self.leased = False  # set by ResultPool.acquire, cleared by the mirror once the caller-facing references exist
        # host mirror (see host_array): field -> (offset, bytes, NumPy dtype, shape); created on the first NumPy read
        self.layout = {'board': (offsets['board'], 64 * n, np.int32, (n, 4, 4)), 'obs': (offsets['obs'], 64 * n, np.float32, (n, 16)),
                       'step_count': (offsets['step_count'], 4 * n, np.int32, (n,)), 'total': (offsets['total'], 4 * n, np.float32, (n,)),
                       'done': (offsets['done'], n, np.bool_, (n,)), 'max_tile': (offsets['max_tile'], 4 * n, np.int32, (n,)),
                       'reward': (offsets['reward'], 4 * n, np.float32, (n,)), 'legal': (offsets['legal'], 4 * n, np.bool_, (n, 4))}
        self.serial = 0  # launches that have written this set (kernel writes do not move torch's version counter)
        self.mirror = None
        self.mirror_fields = None
        self.mirror_key = None
        self.mirror_delta = 0  # (mirror address) - (arena address): what the mirrored entry points of the C-ABI take
        self.direct_serial = -1  # the launch (serial) whose kernel wrote the mirror itself
        # the pool keeps sets per launch stream, so this is the stream every launch into this set is enqueued on
        self.stream = torch.cuda.current_stream(arena.device) if arena.is_cuda else None
        if arena.is_cuda and arena.numel() <= MIRROR_MAX_BYTES:
            self.mirror = torch.empty((arena.numel(),), dtype=torch.uint8, pin_memory=True)
            host = self.mirror.numpy()
            self.mirror_fields = {k: host[o:o + b].view(dt).reshape(shape) for k, (o, b, dt, shape) in self.layout.items()}
            delta = self.mirror.data_ptr() - base
            self.mirror_delta = delta if delta % 16 == 0 else 0
        self.mirror_copies = 0  # device-to-host copies made for NumPy reads (tests / tools read it)

    def idle(self) -> bool:
        """Nobody but this set references its tensors or their memory, and no other thread is between acquire() and its return."""
        if self.leased or _use_count is None or _use_count(self.cdata) != self.idle_use:
            return False
        refs = self.idle_refs
        ts = self.tensors
        return (sys.getrefcount(ts[0]) == refs[0] and sys.getrefcount(ts[1]) == refs[1] and sys.getrefcount(ts[2]) == refs[2]
                and sys.getrefcount(ts[3]) == refs[3] and sys.getrefcount(ts[4]) == refs[4] and sys.getrefcount(ts[5]) == refs[5]
                and sys.getrefcount(ts[6]) == refs[6] and sys.getrefcount(ts[7]) == refs[7])

    def mark_written(self, mirrored: bool = False) -> None:
        """A launch has just been given this set to fill: the stash is valid from the current version on.  ``mirrored``: the
        launch writes the host mirror too (it was given ``mirror_delta``)."""
        self.version = self.arena._version
        self.serial += 1
        if mirrored:
            self.direct_serial = self.serial

    def host_array(self, name: str):
        """Field `name` of this set as a NumPy array backed by the set's PINNED HOST MIRROR, or None when the set is not
        mirrored (host tensors; arenas above MIRROR_MAX_BYTES).  The reference's loop reads four arrays of every state back
        (``np.array(observations)``, ``np.array(legal_masks)``, ``np.array(rewards)``, ``np.array(dones)``,
        self_play.py:346-356): as four device-to-host copies that is four synchronisations (~18 us each).  Here the kernel
        that fills the set writes every result into the mirror as well (``mirror_delta``, env2048_*_mirrored), so the FIRST read
        after a launch is one stream synchronisation and every further read of that state is a host-side copy.  The mirror is
        current for one (launch serial, torch version) pair: after an in-place torch edit of any field it is refreshed with
        one asynchronous copy of the set.  The caller gets a COPY (what np.array means); the mirror itself is never handed out."""
        if self.mirror is None:
            return None
        arena = self.arena
        key = (self.serial, arena._version)
        if self.mirror_key != key:
            if self.direct_serial == self.serial and arena._version == self.version:
                self.stream.synchronize()  # the launch has written the mirror itself: wait for it (on ITS stream)
            else:
                self.mirror.copy_(arena, non_blocking=True)
                torch.cuda.current_stream(arena.device).synchronize()
                self.mirror_copies += 1
            self.mirror_key = key
        return self.mirror_fields[name]

    def stash_valid(self) -> bool:
        """The set is as the kernel wrote it: no in-place torch operation touched any of its tensors since (they share the
        arena's version counter, so an edit of ANY field -- say ``state.step_count += 1`` -- drops the stash: conservative).
        Writes torch cannot see (through a DLPack alias held by another framework) are the caller's responsibility."""
        return self.arena._version == self.version


class ResultPool:
    """Per (n, device, stream) a few ResultSets; ``acquire`` returns an idle one or a new one, LEASED to the calling thread until
    it clears ``leased`` (another thread's acquire() must not pick the same set before the first caller holds its tensors).
    The stream is part of the key for the caching allocator's reason: reuse is ordered behind the previous use only on the
    stream that previous use was enqueued on."""

    def __init__(self):
        self._sets: dict = {}
        self._lock = threading.Lock()
        self.registry: dict = {}  # id(tensor) -> (ResultSet, field name); only tensors of POOLED sets (kept alive by the pool)
        self.created = 0  # sets ever created (tests read it)

    def acquire(self, n: int, device: torch.device, stream: int = 0) -> ResultSet:
        key = (n, device, stream)
        with self._lock:
            sets = self._sets.pop(key, None)  # (re-inserted below: dict order = recency of use)
            if sets is None:
                sets = []
                while len(self._sets) >= MAX_KEYS:  # a caller cycling through many batch sizes must not pin memory for all of them
                    _, dropped = next(iter(self._sets.items()))
                    for old in dropped:
                        for t in old.tensors:
                            self.registry.pop(id(t), None)
                    del self._sets[next(iter(self._sets))]
            self._sets[key] = sets
            for rs in sets:
                if rs.idle():
                    rs.leased = True
                    return rs
            rs = ResultSet(n, device)
            rs.leased = True
            self.created += 1
            if len(sets) < MAX_SETS_PER_KEY and _use_count is not None:
                sets.append(rs)
                rs.pooled = True
                reg = self.registry
                for name, t in zip(('board', 'obs', 'step_count', 'total', 'done', 'max_tile', 'reward', 'legal'), rs.tensors):
                    reg[id(t)] = (rs, name)
            return rs

    def lookup(self, tensor):
        """(set, field) if `tensor` IS one of a pooled set's tensors (the very object), else None."""
        hit = self.registry.get(id(tensor))
        if hit is not None and getattr(hit[0], hit[1]) is tensor:
            return hit
        return None

    def clear(self) -> None:
        with self._lock:
            self._sets.clear()
            self.registry.clear()


class HostStager:
    """Ring of pinned (device-accessible) host memory for small kernel inputs that live on the host."""

    def __init__(self, device: torch.device, capacity: int = 1 << 20):
        self.device = device
        self.gpu = device.type == 'cuda'
        self.buf = torch.empty((capacity,), dtype=torch.uint8, pin_memory=self.gpu)
        self.host = self.buf.numpy()
        self.base = self.buf.data_ptr()
        self.capacity = capacity
        self.half = capacity // 2
        self.offset = 0

    def _enter_half(self, h: int) -> None:
        """Before half `h` is written again every kernel that read it must be done.  Those kernels were enqueued -- on whatever
        streams the caller used -- before the writer left the half, i.e. at least half a ring (512 KiB of ~100-byte inputs: thousands
        of moves) ago; one device synchronisation per half a ring makes that certain at no measurable cost."""
        if self.gpu:
            torch.cuda.synchronize(self.device)

    def stage(self, array: np.ndarray) -> int:
        """Copy a small contiguous host array into the ring; returns the device-visible address (16-byte aligned)."""
        nbytes = array.nbytes
        size = _align(max(nbytes, 1))
        off = self.offset
        h0 = 0 if off < self.half else 1
        if off + size > (self.half if h0 == 0 else self.capacity):  # does not fit the rest of this half: move to the other one
            h0 = 1 - h0
            off = 0 if h0 == 0 else self.half
            self._enter_half(h0)
        self.host[off:off + nbytes] = array.reshape(-1).view(np.uint8)
        self.offset = off + size
        return self.base + off


_local = threading.local()


def stager(device: torch.device) -> HostStager:
    """This thread's stager for `device` (a kernel launch follows every stage() before the thread stages again)."""
    table = getattr(_local, 'stagers', None)
    if table is None:
        table = _local.stagers = {}
    s = table.get(device)
    if s is None:
        s = table[device] = HostStager(device)
    return s


pool = ResultPool()
