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

from . import capi
from .interop import DeviceArray, _registry

_use_count = getattr(torch._C, '_storage_Use_Count', None)  # references to a StorageImpl (tensors, views, wrappers)

MAX_SETS_PER_KEY = 6
MAX_KEYS = 16  # distinct (n, device, stream) keys kept; the least recently used shape's sets are dropped beyond that
STAGE_MAX_BYTES = 16 << 10  # larger host arrays take an ordinary H2D copy (the kernel would read them across the bus)
MIRROR_MAX_BYTES = 64 << 10  # result sets up to this size (N <= ~430) are read back through one pinned host mirror per launch
SIGNAL_TIMEOUT_US = 50  # how long a read polls a launch's completion word before it falls back to synchronising the stream


ALL_FIELDS = frozenset(('board', 'obs', 'step_count', 'total', 'done', 'max_tile', 'reward', 'legal'))
RESET_FIELDS = frozenset(('board', 'obs', 'step_count', 'total', 'done', 'legal'))  # a reset writes no reward / max_tile
NEXT_STATE_FIELDS = frozenset(('board', 'obs', 'done', 'max_tile', 'reward', 'legal'))  # core.next_state: no bookkeeping fields


def _align(x: int, a: int = 16) -> int:
    return (x + a - 1) // a * a


class ResultSet:
    __slots__ = ('n', 'arena', 'storage', 'cdata', 'board', 'obs', 'step_count', 'max_tile', 'total', 'reward', 'legal', 'done',
                 'tensors', 'ptr', 'idle_refs', 'idle_use', 'version', 'pooled', 'leased', 'layout', 'serial',
                 'mirror', 'mirror_fields', 'mirror_key', 'mirror_copies', 'mirror_delta', 'direct_serial', 'stream',
                 'seq_ptr', 'signal_serial', 'signal_hits', 'stream_syncs', 'wait_sequence', 'single', 'fields')

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
        # A set of ONE environment also carries the un-batched aliases the single-environment API hands out (board (4, 4),
        # obs (16,), legal (4,), scalars of shape ()): reset / step / get_observation of reinforce/game/env.py:55-169 then
        # return objects of the set itself -- no view is created per call, the next step recognises its state by identity,
        # and np.array / bool / float of them read the host mirror like the batched arrays do.
        self.single = None
        if n == 1:
            self.single = {'board': self.board.view(4, 4), 'obs': self.obs.view(16), 'step_count': self.step_count.view(()),
                           'total': self.total.view(()), 'done': self.done.view(()), 'max_tile': self.max_tile.view(()),
                           'reward': self.reward.view(()), 'legal': self.legal.view(4)}
            self.tensors = self.tensors + tuple(self.single.values())
        self.ptr = {name: base + offsets[name] for name, _ in sizes}
        self.version = arena._version  # ONE counter: the carved tensors are views of the arena and share its version counter
        self.recount_idle()
        self.pooled = False
        self.leased = False  # set by ResultPool.acquire, cleared by the mirror once the caller-facing references exist
        # host mirror (see host_array): field -> (offset, bytes, NumPy dtype, shape); created on the first NumPy read
        self.layout = {'board': (offsets['board'], 64 * n, np.int32, (n, 4, 4)), 'obs': (offsets['obs'], 64 * n, np.float32, (n, 16)),
                       'step_count': (offsets['step_count'], 4 * n, np.int32, (n,)), 'total': (offsets['total'], 4 * n, np.float32, (n,)),
                       'done': (offsets['done'], n, np.bool_, (n,)), 'max_tile': (offsets['max_tile'], 4 * n, np.int32, (n,)),
                       'reward': (offsets['reward'], 4 * n, np.float32, (n,)), 'legal': (offsets['legal'], 4 * n, np.bool_, (n, 4))}
        self.serial = 0  # launches that have written this set (kernel writes do not move torch's version counter)
        self.fields = ALL_FIELDS  # which fields the last launch into this set wrote (a reset writes no reward / max_tile ...)
        self.mirror = None
        self.mirror_fields = None
        self.mirror_key = None
        self.mirror_delta = 0  # (mirror address) - (arena address): what the mirrored entry points of the C-ABI take
        self.direct_serial = -1  # the launch (serial) whose kernel wrote the mirror itself
        self.seq_ptr = None  # address of the mirror's completion word (single-workgroup launches: n <= capi.SIGNAL_MAX_ENVS)
        self.signal_serial = -1  # the launch (serial) that was asked to store its serial into the completion word
        self.wait_sequence = None
        # the pool keeps sets per launch stream, so this is the stream every launch into this set is enqueued on
        self.stream = torch.cuda.current_stream(arena.device) if arena.is_cuda else None
        self.mirror_copies = 0  # device-to-host copies made for NumPy reads (tests / tools read these three)
        self.signal_hits = 0  # reads that found the launch's completion word (no stream synchronisation)
        self.stream_syncs = 0  # reads that synchronised the stream

    def recount_idle(self) -> None:
        """Take the current reference counts as "nobody but the set (and the pool's registry) holds these tensors"."""
        ts = self.tensors  # (measured exactly the way idle() measures: indexing adds no reference, a loop variable would)
        self.idle_refs = (sys.getrefcount(ts[0]), sys.getrefcount(ts[1]), sys.getrefcount(ts[2]), sys.getrefcount(ts[3]),
                          sys.getrefcount(ts[4]), sys.getrefcount(ts[5]), sys.getrefcount(ts[6]), sys.getrefcount(ts[7]))
        if self.single is not None:  # the eight aliases of a one-environment set
            self.idle_refs = self.idle_refs + tuple(sys.getrefcount(ts[k]) for k in range(8, 16))
        self.idle_use = _use_count(self.cdata) if _use_count else -1

    def attach_mirror(self) -> None:
        """Give this set its pinned host mirror (+ 16 bytes for the completion word).  Only the POOL calls this, and only for
        sets it keeps: a launch into a mirrored set writes pinned memory asynchronously, so the mirror must stay allocated
        until that launch is done -- the pool holds its sets (and parks the ones it drops in a graveyard until an event
        behind their last launch has completed); a set nobody keeps takes no mirror and the plain copy path."""
        arena = self.arena
        if self.mirror is not None or not arena.is_cuda or arena.numel() > MIRROR_MAX_BYTES:
            return
        size = arena.numel()
        self.mirror = torch.zeros((size + 16,), dtype=torch.uint8, pin_memory=True)  # (zeros: completion word 0 = no launch yet)
        host = self.mirror.numpy()
        self.mirror_fields = {k: host[o:o + b].view(dt).reshape(shape) for k, (o, b, dt, shape) in self.layout.items()}
        delta = self.mirror.data_ptr() - arena.data_ptr()
        if delta % 16 == 0:
            self.mirror_delta = delta
            if self.n <= capi.SIGNAL_MAX_ENVS:
                self.seq_ptr = self.mirror.data_ptr() + size
                self.wait_sequence = capi.library().wait_sequence

    def idle(self) -> bool:
        """Nobody but this set references its tensors or their memory, and no other thread is between acquire() and its return."""
        if self.leased or _use_count is None or _use_count(self.cdata) != self.idle_use:
            return False
        refs = self.idle_refs
        ts = self.tensors
        if not (sys.getrefcount(ts[0]) == refs[0] and sys.getrefcount(ts[1]) == refs[1] and sys.getrefcount(ts[2]) == refs[2]
                and sys.getrefcount(ts[3]) == refs[3] and sys.getrefcount(ts[4]) == refs[4] and sys.getrefcount(ts[5]) == refs[5]
                and sys.getrefcount(ts[6]) == refs[6] and sys.getrefcount(ts[7]) == refs[7]):
            return False
        if self.single is not None:
            for k in range(8, 16):
                if sys.getrefcount(ts[k]) != refs[k]:
                    return False
        return True

    def next_serial(self) -> int:
        """The serial the NEXT launch into this set will have (what a *_signalled launch stores into the completion word)."""
        return self.serial + 1

    def mark_written(self, mirrored: bool = False, signalled: bool = False, fields=None) -> None:
        """A launch has just been given this set to fill: the stash is valid from the current version on.  ``mirrored``: the
        launch writes the host mirror too (it was given ``mirror_delta``); ``signalled``: it also stores ``next_serial()`` into
        the mirror's completion word when it is done; ``fields``: the fields that launch writes (default: all)."""
        self.version = self.arena._version
        self.serial += 1
        self.fields = ALL_FIELDS if fields is None else fields
        if mirrored:
            self.direct_serial = self.serial
            if signalled:
                self.signal_serial = self.serial

    def host_array(self, name: str):
        """Field `name` of this set as a NumPy array backed by the set's PINNED HOST MIRROR, or None when the set is not
        mirrored (host tensors; arenas above MIRROR_MAX_BYTES).  The reference's loop reads four arrays of every state back
        (``np.array(observations)``, ``np.array(legal_masks)``, ``np.array(rewards)``, ``np.array(dones)``,
        self_play.py:346-356): as four device-to-host copies that is four synchronisations (~18 us each).  Here the kernel
        that fills the set writes every result into the mirror as well (``mirror_delta``, env2048_*_signalled) and then stores
        its serial into the mirror's completion word, so the FIRST read after a launch polls that word for a few microseconds
        (no stream synchronisation: that is an interrupt-driven wake-up of ~20 us around a 4 us kernel; after
        SIGNAL_TIMEOUT_US without the word the read synchronises the stream after all, so a lost launch cannot hang it) and
        every further read of that state is a host-side copy.  The mirror is current for one (launch serial, torch version)
        pair: after an in-place torch edit of any field it is refreshed with one asynchronous copy of the set.  The caller gets
        a COPY (what np.array means); the mirror itself is never handed out."""
        if self.mirror is None:
            return None
        arena = self.arena
        key = (self.serial, arena._version)
        if self.mirror_key != key:
            if self.direct_serial == self.serial and arena._version == self.version:
                # the launch has written the mirror itself: wait for its completion word, else for ITS stream
                if self.signal_serial == self.serial and self.wait_sequence(self.seq_ptr, self.serial, SIGNAL_TIMEOUT_US):
                    self.signal_hits += 1
                else:
                    self.stream.synchronize()
                    self.stream_syncs += 1
            else:
                self.mirror[:arena.numel()].copy_(arena, non_blocking=True)  # (the mirror's tail is the completion word)
                torch.cuda.current_stream(arena.device).synchronize()
                self.mirror_copies += 1
            self.mirror_key = key
        return self.mirror_fields[name]

    def has(self, field: str) -> bool:
        """`field` holds what the last launch into this set wrote, untouched since."""
        return field in self.fields and self.arena._version == self.version

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
        self.registry: dict = _registry  # id(tensor) -> (ResultSet, field name, tensor); only tensors of POOLED sets (kept alive by the pool)
        self.created = 0  # sets ever created (tests read it)
        # sets the pool has let go of (shape eviction, clear()) whose last launch may still be writing their pinned mirror:
        # (event recorded on the set's stream at the time of the drop, set); released once the event has completed
        self._graveyard: list = []

    def _bury(self, sets) -> None:
        for old in sets:
            for t in old.tensors:
                self.registry.pop(id(t), None)
            if old.mirror is not None and old.stream is not None:
                event = torch.cuda.Event()
                event.record(old.stream)  # behind every launch that was ever given this set (they all went to this stream)
                self._graveyard.append((event, old))

    def _reap(self) -> None:
        if self._graveyard:
            self._graveyard = [(event, old) for event, old in self._graveyard if not event.query()]

    def acquire(self, n: int, device: torch.device, stream: int = 0) -> ResultSet:
        key = (n, device, stream)
        with self._lock:
            sets = self._sets.pop(key, None)  # (re-inserted below: dict order = recency of use)
            if self._graveyard:  # (an empty-list test on the steady-state path; parked sets go as soon as their last launch is done)
                self._reap()
            if sets is None:
                sets = []
                while len(self._sets) >= MAX_KEYS:  # a caller cycling through many batch sizes must not pin memory for all of them
                    oldest = next(iter(self._sets))
                    self._bury(self._sets.pop(oldest))
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
                rs.attach_mirror()  # (only sets the pool keeps: see attach_mirror)
                reg = self.registry
                for name, t in zip(('board', 'obs', 'step_count', 'total', 'done', 'max_tile', 'reward', 'legal'), rs.tensors):
                    reg[id(t)] = (rs, name, t, False)
                if rs.single is not None:
                    for name, t in rs.single.items():
                        reg[id(t)] = (rs, name, t, True)  # True: the un-batched alias (row 0 of the field)
                del t  # (the loop variable is a reference too)
                rs.recount_idle()  # the registry's references belong to the idle state
            return rs

    def lookup(self, tensor):
        """(set, field) if `tensor` IS one of a pooled set's tensors (the very object), else None."""
        hit = self.registry.get(id(tensor))
        if hit is not None and hit[2] is tensor:
            return hit
        return None

    def clear(self) -> None:
        with self._lock:
            for sets in self._sets.values():
                self._bury(sets)
            self._sets.clear()
            self.registry.clear()
            self._reap()


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
