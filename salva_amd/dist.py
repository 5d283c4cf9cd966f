"""Multi-GPU runs: one process (or, in tests, one host thread) and one `LiquidWorld` per GPU, the domain cut into slabs of
grid-cell planes along x (include/salva_hip.h, "multi-GPU").  The reference is single-process; the only
contract kept here is that N slabs step to the same particle states as one world holding everything.

Host-side pieces: the communicator handles and the partition helpers (pure numpy, covered by CPU tests).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib as L

F32 = np.float32


def single_node_rccl_env():
    """RCCL's bootstrap probes the network interfaces (and InfiniBand) of the box before it talks to anybody; in a
    container without a routable interface that probing can stall for minutes (measured: 5.7 s one run, 279 s the next, for
    the same single-rank communicator).  All ranks of this project live on one node: pin the bootstrap to the loopback
    interface unless the caller decided otherwise."""
    import os

    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")


class Comm:
    """A rank's handle on the slab exchange transport (RCCL over xGMI, or the in-process loopback used by tests)."""

    def __init__(self, handle, rank: int, size: int):
        self._h, self.rank, self.size = handle, rank, size

    @staticmethod
    def unique_id() -> bytes:
        single_node_rccl_env()
        buf = (C.c_ubyte * 128)()
        L.check(L.lib().salva_hip_comm_rccl_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def rccl(rank: int, size: int, unique_id: bytes, device: int) -> "Comm":
        assert len(unique_id) == 128
        single_node_rccl_env()
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        h = C.c_void_p()
        L.check(L.lib().salva_hip_comm_rccl_create(rank, size, buf, device, C.byref(h)))
        return Comm(h, rank, size)

    PEER_HANDLE_BYTES = 64

    @staticmethod
    def peer(rank: int, size: int, device: int, gather, slot_bytes: int = 64 << 20) -> "Comm":
        """xGMI peer-direct transport for the ranks of one node (salva_hip_comm_peer_begin / _connect).  `gather(handle: bytes)
        -> list of every rank's handle in rank order` is the caller's all-gather (e.g. torch.distributed.all_gather_object);
        it doubles as the barrier that makes every window exist before anybody writes to it."""
        hb = (C.c_ubyte * Comm.PEER_HANDLE_BYTES)()
        setup = C.c_void_p()
        L.check(L.lib().salva_hip_comm_peer_begin(rank, size, device, slot_bytes, hb, C.byref(setup)))
        try:
            handles = [bytes(h) for h in gather(bytes(hb))]
            if len(handles) != size or any(len(h) != Comm.PEER_HANDLE_BYTES for h in handles):
                raise ValueError(f"peer transport: expected {size} handles of {Comm.PEER_HANDLE_BYTES} bytes")
        except BaseException:
            L.lib().salva_hip_comm_peer_abort(setup)
            raise
        allh = (C.c_ubyte * (size * Comm.PEER_HANDLE_BYTES)).from_buffer_copy(b"".join(handles))
        h = C.c_void_p()
        L.check(L.lib().salva_hip_comm_peer_connect(setup, allh, C.byref(h)))  # consumes `setup`, also on failure
        return Comm(h, rank, size)

    @staticmethod
    def loopback(size: int) -> List["Comm"]:
        hs = (C.c_void_p * size)()
        L.check(L.lib().salva_hip_comm_loopback_create(size, hs))
        return [Comm(C.c_void_p(hs[r]), r, size) for r in range(size)]

    def selftest(self, max_bytes: int = 1 << 20, rounds: int = 6):
        """Collective: patterned exchanges, count exchange and both all-reduces through this transport (raises on a mismatch)."""
        L.check(L.lib().salva_hip_comm_selftest(self._h, max_bytes, rounds))

    def time(self, nbytes: int = 64 << 10, iters: int = 200) -> Tuple[float, float]:
        """Collective: (us per exchange of `nbytes` each way with both neighbours, us per all-reduce of four floats)."""
        a, b = C.c_float(), C.c_float()
        L.check(L.lib().salva_hip_comm_time(self._h, nbytes, iters, C.byref(a), C.byref(b)))
        return a.value, b.value

    def destroy(self):
        if self._h:
            L.lib().salva_hip_comm_destroy(self._h)
            self._h = None


def cell_x(positions: np.ndarray, h: float) -> np.ndarray:
    """floor(x / h) in f32 arithmetic — the same cell coordinate the device computes (hgrid.rs:63-71)."""
    x = np.asarray(positions, F32).reshape(-1, 3)[:, 0]
    return np.floor(x / F32(h)).astype(np.int64)


GHOST_PLANES = 2  # cell planes mirrored on the neighbour per face (salva_amd/csrc/dist.h)


def split_slabs(cx: np.ndarray, nranks: int, min_planes: int = 2 * GHOST_PLANES) -> List[Tuple[int, int]]:
    """Cut the occupied cell planes [cx.min(), cx.max()] into `nranks` contiguous slabs [lo, hi] holding about the same
    number of particles each, every slab at least `min_planes` planes thick (a particle must be mirrored to at most
    one neighbour).  Raises when there are not enough planes."""
    cx = np.asarray(cx, np.int64)
    if cx.size == 0:
        raise ValueError("no particles to partition")
    lo, hi = int(cx.min()), int(cx.max())
    planes = hi - lo + 1
    if planes < nranks * min_planes:
        raise ValueError(f"{planes} cell planes cannot be cut into {nranks} slabs of >= {min_planes} planes")
    hist = np.bincount(cx - lo, minlength=planes)
    csum = np.concatenate([[0], np.cumsum(hist)])
    cuts = [0]
    for r in range(1, nranks):
        target = csum[-1] * r / nranks
        c = int(np.searchsorted(csum, target, side="left"))
        c = max(c, cuts[-1] + min_planes)                  # thick enough on the left ...
        c = min(c, planes - (nranks - r) * min_planes)     # ... and room for the slabs still to come
        cuts.append(c)
    cuts.append(planes)
    return [(lo + cuts[r], lo + cuts[r + 1] - 1) for r in range(nranks)]


def owner_of(cx: np.ndarray, slabs: Sequence[Tuple[int, int]]) -> np.ndarray:
    """Rank owning each cell-x (the first / last slab are open-ended)."""
    his = np.array([s[1] for s in slabs[:-1]], np.int64)
    return np.searchsorted(his, np.asarray(cx, np.int64), side="left").astype(np.int32)


def boundary_subset(bpos: np.ndarray, h: float, slab: Tuple[int, int], rank: int, nranks: int,
                    margin: int = GHOST_PLANES + 1) -> np.ndarray:
    """Indices of the boundary particles a rank must hold: everything within `margin` cell planes of its slab — the
    contacts of its own particles and of its inner ghost plane (which computes its own fields here) reach 2 planes out,
    and those boundary particles need all their own neighbours (one more plane) to get the volume they have in a
    single-domain run; open-ended at the two outer ranks."""
    cx = cell_x(bpos, h)
    lo = -(1 << 62) if rank == 0 else slab[0] - margin
    hi = (1 << 62) if rank == nranks - 1 else slab[1] + margin
    return np.nonzero((cx >= lo) & (cx <= hi))[0]


# ---- host mirrors of the device-side selection rules (salva_amd/csrc/dist.hip, k_dist_flags), used by the CPU tests
def select_migration(cx: np.ndarray, slab: Tuple[int, int], has_lo: bool, has_hi: bool):
    """Phase 1: (keep, to_lo, to_hi) masks of a rank's owned particles.  A particle beyond an open end stays."""
    cx = np.asarray(cx, np.int64)
    to_lo = (cx < slab[0]) if has_lo else np.zeros(len(cx), bool)
    to_hi = (cx > slab[1]) if has_hi else np.zeros(len(cx), bool)
    return ~(to_lo | to_hi), to_lo, to_hi


def select_ghost_planes(cx: np.ndarray, slab: Tuple[int, int], has_lo: bool, has_hi: bool):
    """Phase 2: (to_lo, to_hi) masks of the owned particles mirrored on each neighbour: the GHOST_PLANES edge planes
    facing it (and, at an open-ended slab, nothing more — such a slab has no neighbour on that side)."""
    cx = np.asarray(cx, np.int64)
    to_lo = (cx <= slab[0] + GHOST_PLANES - 1) if has_lo else np.zeros(len(cx), bool)
    to_hi = (cx >= slab[1] - GHOST_PLANES + 1) if has_hi else np.zeros(len(cx), bool)
    return to_lo, to_hi


def allocate_new_ids(adds_per_rank: Sequence[int], rank: int, gid_next: int) -> Tuple[int, int]:
    """Global ids for particles created in a running decomposed world (world_dist.hip, World::dist_add_particles): every rank
    contributes its count to one all-reduced vector; ids continue after the largest id in the run, rank by rank.
    Returns (first id of THIS rank's new particles, the next free id afterwards) — the same pair on the host side of every
    rank once `adds_per_rank` has been all-reduced."""
    adds = [int(a) for a in adds_per_rank]
    return int(gid_next) + sum(adds[:rank]), int(gid_next) + sum(adds)


def apply_owned_deletions(owned_ids: np.ndarray, doomed_ids: np.ndarray) -> np.ndarray:
    """`salva_hip_delete_owned`: every rank gets the same list and drops what it owns; ids owned elsewhere are ignored.
    Returns the keep mask over `owned_ids`."""
    return ~np.isin(np.asarray(owned_ids), np.asarray(doomed_ids))


# ---- DynamicContactSampling in a decomposed world (csrc/world_dist.hip dist_gather_emitted): every rank ends up with every rank's
# emitted points, in rank order.  The transport has sums, not gathers: each rank writes its section of a zeroed table and the
# sections are added as 64-bit integers (x + 0 is exact on bit patterns; the 32-bit fluid words of two ranks that share a 64-bit word
# cannot carry into each other).  These three functions are the host mirror of that packing, used by the gloo protocol test.
def emitted_table_words(total: int) -> int:
    """64-bit words of the table: `total` rows of (x, y, z, source id) as two words each, then one 32-bit fluid index per row."""
    return 2 * total + (total + 1) // 2


def pack_emitted(points: np.ndarray, source_ids: np.ndarray, fluids: np.ndarray, before: int, total: int) -> np.ndarray:
    """This rank's section of the table: `points` (k, 3) f32, `source_ids` / `fluids` (k,) u32, written behind the `before` rows of
    the lower ranks; zero everywhere else."""
    k = len(points)
    rows = np.zeros((total, 4), np.uint32)
    rows[before:before + k, :3] = np.ascontiguousarray(points, np.float32).view(np.uint32).reshape(k, 3)
    rows[before:before + k, 3] = np.asarray(source_ids, np.uint32)
    models = np.zeros(2 * ((total + 1) // 2), np.uint32)
    models[before:before + k] = np.asarray(fluids, np.uint32)
    return np.concatenate([rows.reshape(-1).view(np.uint64), models.view(np.uint64)])


def unpack_emitted(words: np.ndarray, total: int):
    """The summed table -> (points (total, 3) f32, source ids (total,) u32, fluids (total,) u32)."""
    words = np.ascontiguousarray(words, np.uint64)
    rows = words[:2 * total].view(np.uint32).reshape(total, 4)
    models = words[2 * total:].view(np.uint32)[:total]
    return rows[:, :3].copy().view(np.float32), rows[:, 3].copy(), models.copy()

