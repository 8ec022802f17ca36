"""One-process-per-GPU sharding of the two hot paths over ``torch.distributed`` (RCCL on MI355X, gloo on CPU).

The reference is single-process (rayon) — SURVEY.md §8e defines the MI355X-native exchange:

* MCCFR: trees sharded by rank; each rank reduces its Decisions to one composed map per table cell; ONE
  all-gather per step; every rank folds the maps in rank order (replicas stay bit-identical).
* k-means: points (and their Elkan bounds) sharded by rank; ONE all-reduce(sum) of the integer centroid
  sums per iteration; k-means++ draws with an exact integer prefix over ranks (same picks as one GPU).

``engine`` is anything with the C-ABI's sharded surface (``summary_bytes/step_local/step_apply`` resp.
``partial_bytes/step_local/step_finish/kpp_*``): the HIP ``Solver``/``Layer`` on a GPU box, the CPU oracle in
the gloo tests.  Buffers are torch tensors on ``device``; only raw pointers cross into the engine.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

MASK64 = (1 << 64) - 1


def rp_mix64(z: int) -> int:
    z &= MASK64
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & MASK64
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & MASK64
    z ^= z >> 31
    return z


def rp_stream(seed: int, counter: int) -> int:
    """include/rp_math.h rp_stream (k-means++ round hash)."""
    return rp_mix64(rp_mix64((seed + 0x9E3779B97F4A7C15) & MASK64) ^ ((counter * 0xD1342543DE82EF95 + 1) & MASK64))


def rp_mulhi64(a: int, b: int) -> int:
    return ((a & MASK64) * (b & MASK64)) >> 64


def _flat_all_gather_supported(group=None) -> bool:
    """whether the backend has all_gather_into_tensor: decided ONCE, by capability, never by catching a failed collective
    (a rank that retried a different collective after a real RCCL error would desynchronise the job)"""
    if not hasattr(dist, "all_gather_into_tensor"):
        return False
    return dist.get_backend(group) in ("nccl", "gloo")


def _all_gather_bytes(out: torch.Tensor, mine: torch.Tensor, group=None, flat=None):
    """every rank's `mine` back to back in `out`; an error of the collective propagates"""
    if flat is None:
        flat = _flat_all_gather_supported(group)
    if flat:
        dist.all_gather_into_tensor(out, mine, group=group)
    else:
        world = dist.get_world_size(group)
        dist.all_gather(list(out.view(world, -1).unbind(0)), mine, group=group)


class Comm:
    """rp_comm: the library's own RCCL communicator (csrc/comm.cpp) — what a host without torch.distributed uses.  Here the
    128-byte id travels over torch.distributed's broadcast; a Rust / C host ships it by MPI, a socket or a file."""

    def __init__(self, rank: int, world: int, device: int, unique_id: bytes):
        import ctypes as C

        from . import _lib

        self._lib = _lib.load()
        self._h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _lib.check(self._lib.rp_comm_create(buf, rank, world, device, C.byref(self._h)))
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C

        from . import _lib

        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.load().rp_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_process_group(cls, device: int, group=None):
        """rank 0 makes the id, torch.distributed broadcasts it, every rank joins (collective)"""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        on_gpu = dist.get_backend(group) == "nccl"
        t = torch.zeros(128, dtype=torch.uint8, device="cuda" if on_gpu else "cpu")
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(cls.unique_id()), dtype=torch.uint8))
        dist.broadcast(t, src=0, group=group)
        return cls(rank, world, device, bytes(t.cpu().numpy().tobytes()))

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rp_comm_destroy(self._h)
            self._h = None

    __del__ = close


class _StreamScope:
    """On a GPU the engine's kernels and the collectives must share ONE stream so they are ordered without host
    synchronisation: a dedicated (non-null) torch stream is handed to the engine and made current around every
    step.  On CPU (gloo tests) this is a no-op."""

    def __init__(self, engine, device):
        self.stream = None
        if str(device) != "cpu":
            self.stream = torch.cuda.Stream()
            engine.set_stream(self.stream.cuda_stream)

    def __enter__(self):
        if self.stream is not None:
            self._ctx = torch.cuda.stream(self.stream)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.stream is not None:
            self._ctx.__exit__(*exc)
        return False


class ShardedSolver:
    """Tree-sharded MCCFR: rank r samples tree ids [r*B, (r+1)*B) of a world*B-tree epoch.

    ``window`` = local steps per exchange (the PERIODIC all-gather of north_star): the composed maps of ``window``
    consecutive steps — all traversed against the start-of-window table — are folded locally and gathered once."""

    def __init__(self, engine, device="cpu", group=None, window=1):
        self.engine = engine
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.window = max(1, int(window))
        self.pending = 0
        self.flat = _flat_all_gather_supported(group)
        engine.set_shard(self.rank, self.world)
        self.scope = _StreamScope(engine, device)
        n = engine.summary_bytes()
        self.mine = torch.empty(n, dtype=torch.uint8, device=device)
        self.all = torch.empty(n * self.world, dtype=torch.uint8, device=device)

    def step(self):
        with self.scope:
            self.engine.window_local(self.mine.data_ptr(), self.pending == 0)
            self.pending += 1
            if self.pending == self.window:
                self._exchange()

    def _exchange(self):
        _all_gather_bytes(self.all, self.mine, self.group, self.flat)
        self.engine.window_apply(self.all.data_ptr(), self.world)
        self.pending = 0

    def flush(self):
        """close a partly filled window (end of a run)"""
        if self.pending:
            with self.scope:
                self._exchange()

    def solve(self, trees_per_rank: int, batch: int):
        for _ in range(trees_per_rank // batch):
            self.step()
        self.flush()
        return self


class ShardedLayer:
    """Point-sharded Elkan k-means: this rank owns the contiguous point range [lo, hi) of the global set."""

    def __init__(self, engine, K: int, bins: int, seed: int, device="cpu", group=None):
        self.engine = engine
        self.K, self.bins, self.seed = K, bins, seed
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device
        self.flat = _flat_all_gather_supported(group)
        self.scope = _StreamScope(engine, device)
        self.nbytes = engine.partial_bytes()
        self.buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        # partial layout: [K*bins u32][K u32][pad to 8][K u64]
        self.words32 = K * bins + K
        self.off64 = ((self.words32 * 4) + 7) & ~7

    def init_centroids(self):
        """Layer::init_centroids (k-means++, layer.rs:140-181) with a global exact-integer draw."""
        e = self.engine
        e.kpp_begin()
        chosen = []
        for k in range(self.K):
            local = e.kpp_total()
            totals = torch.zeros(self.world, dtype=torch.int64)
            mine = torch.tensor([local], dtype=torch.int64)
            if self.device != "cpu":
                totals, mine = totals.to(self.device), mine.to(self.device)
            _all_gather_bytes(totals, mine, self.group, self.flat)
            totals = [int(t) for t in totals.cpu().tolist()]
            total = sum(totals)
            h = rp_stream(self.seed, k)
            counts_all = self._point_counts()
            if total == 0:
                g = rp_mulhi64(h, sum(counts_all))  # uniform over the global point index
                owner, before = 0, 0
                while g >= before + counts_all[owner]:
                    before += counts_all[owner]
                    owner += 1
                idx = g - before
            else:
                r = rp_mulhi64(h, total)
                owner, before = 0, 0
                while r >= before + totals[owner]:  # first rank whose inclusive prefix exceeds r
                    before += totals[owner]
                    owner += 1
                idx = e.kpp_pick(r - before) if owner == self.rank else -1
            hist = torch.zeros(self.bins, dtype=torch.int64)
            if owner == self.rank:
                hist = torch.from_numpy(e.get_point(idx).astype(np.int64))
            if self.device != "cpu":
                hist = hist.to(self.device)
            dist.broadcast(hist, src=owner, group=self.group)
            e.set_centroid(k, hist.cpu().numpy().astype(np.uint32))
            e.kpp_update(k)
            chosen.append((owner, idx if owner == self.rank else None))
        return chosen

    def _point_counts(self):
        n = torch.tensor([self.engine.N], dtype=torch.int64)
        out = torch.zeros(self.world, dtype=torch.int64)
        if self.device != "cpu":
            n, out = n.to(self.device), out.to(self.device)
        _all_gather_bytes(out, n, self.group, self.flat)
        return [int(v) for v in out.cpu().tolist()]

    def init_bounds(self):
        self.engine.init_bounds()

    def step(self):
        """Kmeans::next with the centroid sums all-reduced across ranks (exact integers, order free)."""
        with self.scope:
            self.engine.step_local(self.buf.data_ptr())
            dist.all_reduce(self.buf[: self.words32 * 4].view(torch.int32), op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(self.buf[self.off64:].view(torch.int64), op=dist.ReduceOp.SUM, group=self.group)
            return self.engine.step_finish(self.buf.data_ptr())


class ShardedProfile:
    """Batch-sharded sparse profile (rp_profile_*): every rank holds a replica of the table and summarises ITS
    Decisions into per-row composed entries; the entry lists are all-gathered (padded to the longest list) and every
    rank folds the same rank-major list, so replicas stay bit-identical.  SURVEY.md §8e: the sparse variant of the
    periodic regret/strategy exchange — traffic is proportional to the rows touched, not to the table size.

    ``engine`` has ``entry_bytes / summarize / fold`` (robopoker_amd.sparse.SparseProfile on a GPU; the oracle in the
    gloo tests); ``make_buffer(nbytes)`` returns a uint8 torch tensor on the engine's device."""

    def __init__(self, engine, max_batch: int, device="cpu", group=None):
        self.engine = engine
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device
        self.eb = engine.entry_bytes()
        self.flat = _flat_all_gather_supported(group)
        self.scope = _StreamScope(engine, device)
        self.mine = torch.zeros(max_batch * self.eb, dtype=torch.uint8, device=device)
        self.all = torch.zeros(self.world * max_batch * self.eb, dtype=torch.uint8, device=device)
        self.packed = torch.zeros(self.world * max_batch * self.eb, dtype=torch.uint8, device=device)

    def step(self, batch):
        with self.scope:
            n = self.engine.summarize(batch, self.mine.data_ptr())
            counts = torch.zeros(self.world, dtype=torch.int64, device=self.device)
            mine_n = torch.tensor([n], dtype=torch.int64, device=self.device)
            _all_gather_bytes(counts, mine_n, self.group, self.flat)
            counts = [int(c) for c in counts.cpu().tolist()]
            width = max(counts) * self.eb  # every rank sends the same number of bytes
            if width == 0:
                self.engine.fold(self.packed.data_ptr(), 0)
                return 0
            _all_gather_bytes(self.all[: self.world * width], self.mine[:width], self.group, self.flat)
            # compact the padded lists into one rank-major list
            off = 0
            for r, c in enumerate(counts):
                self.packed[off: off + c * self.eb] = self.all[r * width: r * width + c * self.eb]
                off += c * self.eb
            total = sum(counts)
            self.engine.fold(self.packed.data_ptr(), total)
            return total


class ShardedNlhe:
    """Tree-sharded NLHE blueprint MCCFR (BASELINE configs[3]): rank r traverses tree ids [r*B, (r+1)*B) of a world*B-tree
    epoch against its replica of the table; the per-infoset composed entries are exchanged BY KEY (every rank's table assigns
    rows in its own insertion order) — all-gathered padded to the longest list, packed rank-major on the device, mapped to the
    local rows and folded in rank order.  Replicas stay identical as key -> Encounter maps.

    ``engine``: ``set_shard / entry_bytes / step_local / step_apply`` (robopoker_amd.nlhe.NlheSolver on a GPU; the oracle in the
    gloo tests)."""

    def __init__(self, engine, device="cpu", group=None):
        self.engine = engine
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device
        self.flat = _flat_all_gather_supported(group)
        engine.set_shard(self.rank, self.world)
        self.scope = _StreamScope(engine, device)
        self.eb, cap = engine.entry_bytes()
        self.cap = cap
        self._mk = lambda n, dt: torch.zeros(n, dtype=dt, device=device)  # noqa: E731
        mk = self._mk
        # step_local writes one entry per infoset the rank touched into `mine` (capacity: the batch's Decisions, the C contract);
        # the gathered and packed lists are sized by what the ranks actually send (grow-only): a few 10^5 entries, not the 4 x 10^7
        # of the worst case
        self.mine = {"ent": mk(cap * self.eb, torch.uint8), "past": mk(cap, torch.int64), "present": mk(cap, torch.int32),
                     "choices": mk(cap, torch.int64)}
        self.all = {k: mk(0, v.dtype) for k, v in self.mine.items()}
        self.packed = {k: mk(0, v.dtype) for k, v in self.mine.items()}

    def _room(self, bufs, k, n):
        if bufs[k].numel() < n:
            bufs[k] = self._mk(n + n // 4, bufs[k].dtype)
        return bufs[k]

    def step(self) -> int:
        with self.scope:
            return self._step()

    def _step(self) -> int:
        m = self.mine
        # a rank whose part fails (full table, node budget) says so in the count exchange: every rank leaves the step together
        # instead of the others waiting in the gather (as rp_nlhe_step_comm does)
        failure = None
        try:
            n = self.engine.step_local(m["ent"].data_ptr(), m["past"].data_ptr(), m["present"].data_ptr(), m["choices"].data_ptr())
        except Exception as exc:  # noqa: BLE001
            failure, n = exc, -1
        counts = torch.zeros(self.world, dtype=torch.int64, device=self.device)
        _all_gather_bytes(counts, torch.tensor([n], dtype=torch.int64, device=self.device), self.group, self.flat)
        counts = [int(c) for c in counts.cpu().tolist()]
        if failure is not None:
            raise failure
        if min(counts) < 0:
            raise RuntimeError(f"ShardedNlhe.step: rank {counts.index(min(counts))} failed its part of the step")
        width, total = max(counts), sum(counts)
        for k, unit in (("ent", self.eb), ("past", 1), ("present", 1), ("choices", 1)):
            w = width * unit
            allk, pk = self._room(self.all, k, max(self.world * w, 1)), self._room(self.packed, k, max(total * unit, 1))
            if not width:
                continue
            _all_gather_bytes(allk[: self.world * w], m[k][:w], self.group, self.flat)
            off = 0
            for r, c in enumerate(counts):  # rank-major packing, on the device
                pk[off: off + c * unit] = allk[r * w: r * w + c * unit]
                off += c * unit
        p = self.packed
        self.engine.step_apply(p["ent"].data_ptr(), p["past"].data_ptr(), p["present"].data_ptr(), p["choices"].data_ptr(), total)
        return total
