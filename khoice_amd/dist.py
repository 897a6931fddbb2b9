"""Multi-GPU experiment type 1: one process per MI355X, torch.distributed over RCCL/xGMI.

Sharding
  * steps 1-6 (per-genome sets, within-group unions, step_5 histograms) are independent per
    group: every rank owns whole groups and needs NO communication.
  * steps 7-8 (in how many groups does a k-mer occur) are the one real exchange:
      1. the mixed key space is cut into `world` equal-width, order-preserving slots; a set is
         sorted by mixed key, so what rank j owns of a group set is one contiguous slice;
      2. all-to-all of the slices (one variable-size all_to_all_single of 8W-byte keys; the
         counters travel too only when some set carries any) — a full-mesh pattern that uses
         every xGMI link at once, never a ring over a bitmap;
      3. each rank union-sums the slices it received in ONE pass (counter = number of groups
         holding the key, saturating at cs) with the histogram fused in;
      4. all-reduce (sum) of the small histogram.
    Mixed keys are uniform, so the slots are balanced without sampling.

The exchange is written against a small `ops` interface so that the very same code runs under
`gloo` on CPU in the tests (with an oracle-backed stand-in for the engine).
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch
import torch.distributed as dist


class _Trace:
    """KHOICE_TRACE=1: wall time between the phases of one exchange, on stderr (rank 0)."""

    def __init__(self, name):
        import os
        self.on = bool(os.environ.get("KHOICE_TRACE")) and (not dist.is_initialized() or dist.get_rank() == 0)
        self.name, self.marks = name, []
        if self.on:
            import time
            self.clock = time.perf_counter
            self.t = self.clock()

    def __call__(self, label):
        if self.on:
            now = self.clock()
            self.marks.append(f"{label} {1e3 * (now - self.t):.3f}")
            self.t = now

    def report(self):
        if self.on:
            import sys
            print(f"[khoice trace] {self.name}: " + " | ".join(self.marks) + " ms", file=sys.stderr)


class EngineOps:
    """Engine side of the exchange for real runs: sets live in HBM, the exchange buffers are
    torch tensors on the same device, received slices are wrapped without copying."""

    def __init__(self, eng, device: torch.device, stage_on_host: bool = False):
        self.eng = eng
        self.device = device
        # rehearsal mode (gloo, several ranks sharing one GPU): collectives run on host copies
        self.stage_on_host = stage_on_host

    def to_comm(self, t):
        return t.cpu() if self.stage_on_host else t

    def from_comm(self, t):
        return t.to(self.device) if self.stage_on_host else t

    @property
    def comm_device(self):
        return torch.device("cpu") if self.stage_on_host else self.device

    def words(self, k):
        return 1 if k <= 32 else 2

    def has_counts(self, s):
        i = s.info()
        return bool(i["has_counts"]) or i["uniform"] != 1

    def partition_bounds(self, s, nparts):
        return s.partition_bounds(nparts)

    def partition_bounds_all(self, sets, nparts):
        return self.eng.partition_bounds(sets, nparts)

    def export_range(self, s, lo, hi, keys_t, counts_t):
        s.export_range(lo, hi, keys_t.data_ptr(), counts_t.data_ptr() if counts_t is not None else None)

    def view_range(self, s, lo, hi):
        """Zero-copy set over records [lo, hi) of `s` (the slice this rank keeps for itself)."""
        keys_ptr, counts_ptr = s.device_ptrs()
        w = self.words(s.k)
        return self.eng.wrap_device(s.k, hi - lo, keys_ptr + lo * 8 * w,
                                    counts_ptr + lo * 4 if counts_ptr else None, uniform=s.info()["uniform"])

    def export_all(self, s):
        """(keys int64[n*W], counts int32[n]) device tensors holding the set's raw storage (mixed keys
        ascending, counters materialised): the send buffer of the exchange.  One copy each; the slices
        of a sorted set are already destination-major, so nothing is packed."""
        n, w = len(s), self.words(s.k)
        keys = torch.empty(n * w, dtype=torch.int64, device=self.device)
        cnt = torch.empty(n, dtype=torch.int32, device=self.device)
        if n:
            s.export_device(keys.data_ptr(), cnt.data_ptr())    # synchronises the engine's stream
        return keys, cnt

    def flush(self):
        self.eng.sync()                       # export copies ran on the engine's stream

    def wrap(self, k, n, keys_t, counts_t):
        return self.eng.wrap_device(k, n, keys_t.data_ptr(), counts_t.data_ptr() if counts_t is not None else None)

    def before_wrap(self):
        torch.cuda.synchronize(self.device)   # the collective ran on torch's stream

    def union_hist(self, sets, cs, hist_len):
        return None, self.eng.union_histogram(sets, cs, hist_len)

    # -- direct-addressed occurrence table (k <= 16)
    def table_add(self, s, table_t):
        self.eng.table_add_set(s, table_t.data_ptr(), table_t.element_size())

    def table_hist(self, table_t, lo, hi, cs, hist_len):
        return self.eng.table_histogram(table_t.data_ptr(), table_t.element_size(), lo, hi, cs, hist_len)

    def set_len(self, s):
        return len(s)


def across_groups_distributed(ops, group_sets: Sequence, k: int, cs: int, hist_len: int,
                              group=None) -> np.ndarray:
    """Global step_8 histogram: hist[c] = number of distinct k-mers that occur in exactly c of
    ALL ranks' groups (c saturating at cs).  Collective: every rank must call it."""
    world = dist.get_world_size(group)
    dev = ops.device
    trace = _Trace("slots")
    cdev = getattr(ops, "comm_device", dev)          # where the collectives run
    to_comm = getattr(ops, "to_comm", lambda t: t)
    from_comm = getattr(ops, "from_comm", lambda t: t)
    w = ops.words(k)
    g_local = len(group_sets)
    # slot boundaries of every local group set: bounds[g][j] .. bounds[g][j+1] goes to rank j
    if hasattr(ops, "partition_bounds_all"):     # one launch + one synchronisation for all sets
        ball = np.asarray(ops.partition_bounds_all(list(group_sets), world), dtype=np.int64)
        bounds = [ball[g] for g in range(g_local)]
    else:
        bounds = [np.asarray(ops.partition_bounds(s, world), dtype=np.int64) for s in group_sets]
    meta = torch.tensor([g_local, int(any(ops.has_counts(s) for s in group_sets))], dtype=torch.int64, device=cdev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    trace("bounds+meta")
    g_all = [int(m[0]) for m in metas]
    with_counts = any(int(m[1]) for m in metas)
    g_max = max(max(g_all), 1)
    # slice lengths, padded to g_max per destination; the slice a rank owns itself never
    # travels: it is used in place
    rank = dist.get_rank(group)
    keep_local = hasattr(ops, "view_range")
    send_len = np.zeros((world, g_max), dtype=np.int64)
    for g in range(g_local):
        send_len[:, g] = bounds[g][1:] - bounds[g][:-1]
    if keep_local:
        send_len[rank, :] = 0
    sl = torch.from_numpy(send_len.reshape(-1)).to(cdev)
    rl = torch.empty(world * g_max, dtype=torch.int64, device=cdev)
    dist.all_to_all_single(rl, sl, group=group)
    recv_len = rl.cpu().numpy().reshape(world, g_max)
    trace("sizes")
    send_n = send_len.sum(axis=1)
    recv_n = recv_len.sum(axis=1)
    # pack: destination-major, group-minor
    skeys = torch.empty(int(send_n.sum()) * w, dtype=torch.int64, device=dev)
    scnt = torch.empty(int(send_n.sum()), dtype=torch.int32, device=dev) if with_counts else None
    off = 0
    for j in range(world):
        for g in range(g_local):
            n = int(send_len[j, g])
            if n:
                ops.export_range(group_sets[g], int(bounds[g][j]), int(bounds[g][j + 1]),
                                 skeys[off * w:(off + n) * w], scnt[off:off + n] if with_counts else None)
            off += n
    ops.flush()
    trace("pack")
    rkeys = torch.empty(int(recv_n.sum()) * w, dtype=torch.int64, device=cdev)
    dist.all_to_all_single(rkeys, to_comm(skeys), output_split_sizes=[int(n) * w for n in recv_n],
                           input_split_sizes=[int(n) * w for n in send_n], group=group)
    rkeys = from_comm(rkeys)
    rcnt = None
    if with_counts:
        rcnt = torch.empty(int(recv_n.sum()), dtype=torch.int32, device=cdev)
        dist.all_to_all_single(rcnt, to_comm(scnt), output_split_sizes=[int(n) for n in recv_n],
                               input_split_sizes=[int(n) for n in send_n], group=group)
        rcnt = from_comm(rcnt)
    ops.before_wrap()
    trace("all_to_all")
    # every received slice is sorted, distinct and inside this rank's slot: union them in one pass
    slices = []
    if keep_local:
        for g in range(g_local):
            lo, hi = int(bounds[g][rank]), int(bounds[g][rank + 1])
            if hi > lo:
                slices.append(ops.view_range(group_sets[g], lo, hi))
    off = 0
    for i in range(world):
        for g in range(g_max):
            n = int(recv_len[i, g])
            if n:
                slices.append(ops.wrap(k, n, rkeys[off * w:(off + n) * w],
                                       rcnt[off:off + n] if with_counts else None))
            off += n
    if slices:
        _, hist = ops.union_hist(slices, cs, hist_len)
        hist = np.asarray(hist)
    else:
        hist = np.zeros(hist_len, dtype=np.uint64)
    del slices
    trace("union")
    ht = torch.from_numpy(hist.astype(np.int64)).to(cdev)
    dist.all_reduce(ht, op=dist.ReduceOp.SUM, group=group)
    out = ht.cpu().numpy().astype(np.uint64)
    trace("all_reduce")
    trace.report()
    return out


def across_set_exchange(ops, aset, k: int, cs: int, hist_len: int, group=None) -> np.ndarray:
    """Global step_8 histogram from every rank's LOCAL across-group set (the union of its own group
    sets, counter = number of its groups holding the k-mer — what the fused kh_exp1_run emits).
    The set is sorted by mixed key, so its slices by owner rank are contiguous and already in
    destination order: the set's own storage is the send buffer.  Collectives: one small
    all_to_all of the slice sizes, one of the keys, one of the counters, one all_reduce of the
    histogram.  The receiver sums the counters of equal keys (saturating at cs) in one pass.
    Collective: every rank must call it."""
    world = dist.get_world_size(group)
    dev = ops.device
    trace = _Trace("across_set")
    cdev = getattr(ops, "comm_device", dev)
    to_comm = getattr(ops, "to_comm", lambda t: t)
    from_comm = getattr(ops, "from_comm", lambda t: t)
    w = ops.words(k)
    bounds = np.asarray(ops.partition_bounds(aset, world), dtype=np.int64)
    send_n = bounds[1:] - bounds[:-1]
    sl = torch.from_numpy(send_n.copy()).to(cdev)
    rl = torch.empty(world, dtype=torch.int64, device=cdev)
    dist.all_to_all_single(rl, sl, group=group)
    recv_n = rl.cpu().numpy()
    trace("bounds+sizes")
    skeys, scnt = ops.export_all(aset)
    trace("export")
    rkeys = torch.empty(int(recv_n.sum()) * w, dtype=torch.int64, device=cdev)
    rcnt = torch.empty(int(recv_n.sum()), dtype=torch.int32, device=cdev)
    dist.all_to_all_single(rkeys, to_comm(skeys), output_split_sizes=[int(n) * w for n in recv_n],
                           input_split_sizes=[int(n) * w for n in send_n], group=group)
    dist.all_to_all_single(rcnt, to_comm(scnt), output_split_sizes=[int(n) for n in recv_n],
                           input_split_sizes=[int(n) for n in send_n], group=group)
    rkeys, rcnt = from_comm(rkeys), from_comm(rcnt)
    ops.before_wrap()
    trace("all_to_all")
    slices, off = [], 0
    for i in range(world):
        n = int(recv_n[i])
        if n:
            slices.append(ops.wrap(k, n, rkeys[off * w:(off + n) * w], rcnt[off:off + n]))
        off += n
    if slices:
        _, hist = ops.union_hist(slices, cs, hist_len)
        hist = np.asarray(hist)
    else:
        hist = np.zeros(hist_len, dtype=np.uint64)
    del slices
    trace("union")
    ht = torch.from_numpy(hist.astype(np.int64)).to(cdev)
    dist.all_reduce(ht, op=dist.ReduceOp.SUM, group=group)
    out = ht.cpu().numpy().astype(np.uint64)
    trace("all_reduce")
    trace.report()
    return out


TABLE_MAX_K = 16      # 4^16 one-byte cells = 4 GiB of the 288 GB


def across_groups_table(ops, group_sets: Sequence, k: int, cs: int, hist_len: int, group=None) -> np.ndarray:
    """Small-k form of the same histogram (BASELINE.json north_star "RCCL
    all-reduce ... of the per-group unique-k-mer bitmaps"): every group set is a presence bitmap
    over the 4^k canonical key values; the sum of the bitmaps, widened to one byte (four when
    more than 255 groups exist), is accumulated locally in a direct-addressed table, all-reduced
    (sum) across the ranks, and each rank histograms its 1/world of the cells; the small
    histogram is all-reduced last.  Ring traffic is 2 * 4^k bytes per rank whatever the sets
    hold, so this wins only while 4^k is small next to the keys the slot exchange would move:
    `across_groups_auto` chooses."""
    if k > TABLE_MAX_K:
        raise ValueError(f"across_groups_table: k = {k} > {TABLE_MAX_K} has no direct-addressed table")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = ops.device
    cdev = getattr(ops, "comm_device", dev)
    to_comm = getattr(ops, "to_comm", lambda t: t)
    from_comm = getattr(ops, "from_comm", lambda t: t)
    ng = torch.tensor([len(group_sets)], dtype=torch.int64, device=cdev)
    dist.all_reduce(ng, op=dist.ReduceOp.SUM, group=group)
    dtype = torch.uint8 if int(ng.item()) <= 255 else torch.int32
    ncell = 4 ** k
    table = torch.zeros(ncell, dtype=dtype, device=dev)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)           # the zero fill ran on torch's stream
    for s in group_sets:
        ops.table_add(s, table)
    ops.flush()
    tc = to_comm(table)
    dist.all_reduce(tc, op=dist.ReduceOp.SUM, group=group)
    table = from_comm(tc)
    ops.before_wrap()
    lo = (rank * ncell // world) // 16 * 16
    hi = ncell if rank == world - 1 else ((rank + 1) * ncell // world) // 16 * 16
    hist = np.asarray(ops.table_hist(table, lo, hi, cs, hist_len))
    ht = torch.from_numpy(hist.astype(np.int64)).to(cdev)
    dist.all_reduce(ht, op=dist.ReduceOp.SUM, group=group)
    return ht.cpu().numpy().astype(np.uint64)


def across_groups_auto(ops, group_sets: Sequence, k: int, cs: int, hist_len: int, group=None) -> np.ndarray:
    """Table form when its ring traffic (2 * 4^k cells) is below what the slot exchange moves
    (8 bytes per local key); the slot exchange otherwise.  The choice is made on the global
    key count so that every rank takes the same branch."""
    if k <= TABLE_MAX_K and hasattr(ops, "table_add"):
        cdev = getattr(ops, "comm_device", ops.device)
        nk = torch.tensor([sum(ops.set_len(s) for s in group_sets), len(group_sets)], dtype=torch.int64, device=cdev)
        dist.all_reduce(nk, op=dist.ReduceOp.SUM, group=group)
        world = dist.get_world_size(group)
        cell = 1 if int(nk[1]) <= 255 else 4
        if 2 * cell * 4 ** k < 8 * int(nk[0]) // world:
            return across_groups_table(ops, group_sets, k, cs, hist_len, group)
    return across_groups_distributed(ops, group_sets, k, cs, hist_len, group)


def across_records_exchange(ops, eng, seqs, group_of: Sequence[int], k: int, cs: int, hist_len: int, group=None,
                            overlap=None):
    """Global step_8 histogram by exchange of MINIMIZER RECORDS (SURVEY.md §8e.2 in the super-k-mer form): slots are a
    global function of the minimizer, so rank j owns a range of slots.  Every rank packs its genomes' records (tag =
    local group, identical records merged) by owner, the packed arrays travel in four all-to-alls (records, masks, and
    per slot how many / where), the owner runs one phased union over the pieces — a piece is a phase, tags of
    different phases are different groups — and the small histogram is all-reduced.  0.5 GB of records per 250 Mbp and
    rank instead of 0.93 GB of keys, and no rank builds a key set.  Collective: every rank must call it.

    overlap: a callable run while the four large all-to-alls are in flight (they are posted asynchronously; the local
    steps 1-6 of the rank are independent of them).  Returns the histogram, or (histogram, overlap()) when given."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = ops.device
    cdev = getattr(ops, "comm_device", dev)
    to_comm = getattr(ops, "to_comm", lambda t: t)
    from_comm = getattr(ops, "from_comm", lambda t: t)
    trace = _Trace("records")
    lens = [int(s[1]) if isinstance(s, tuple) else len(s) for s in seqs]
    positions = sum(max(0, n - k + 1) for n in lens)
    sizes = np.bincount(np.asarray(group_of, dtype=np.int64))
    agree = torch.tensor([positions, int(sizes.max()) if sizes.size else 1], dtype=torch.int64, device=cdev)
    dist.all_reduce(agree, op=dist.ReduceOp.MAX, group=group)
    nslots, spp, cap = eng.skm_exchange_plan(k, int(agree[0]), int(agree[1]), world)
    rec = torch.empty((world, cap, 2), dtype=torch.int64, device=dev)
    msk = torch.empty((world, cap), dtype=torch.int32, device=dev)
    cnt = torch.empty((world, spp), dtype=torch.int32, device=dev)
    off = torch.empty((world, spp), dtype=torch.int32, device=dev)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    part_n = eng.skm_pack(seqs, group_of, k, nslots, world, cap, rec.data_ptr(), msk.data_ptr(), cnt.data_ptr(), off.data_ptr())
    trace("pack")
    send_n = part_n.astype(np.int64)
    send_n[rank] = 0                                  # the own part is used in place
    sl = torch.from_numpy(send_n.copy()).to(cdev)
    rl = torch.empty(world, dtype=torch.int64, device=cdev)
    dist.all_to_all_single(rl, sl, group=group)
    recv_n = rl.cpu().numpy()
    # contiguous send buffers: the parts' used prefixes, destination-major
    srec = torch.cat([rec[j, :int(send_n[j])].reshape(-1) for j in range(world)])
    smsk = torch.cat([msk[j, :int(send_n[j])] for j in range(world)])
    rrec = torch.empty(int(recv_n.sum()) * 2, dtype=torch.int64, device=cdev)
    rmsk = torch.empty(int(recv_n.sum()), dtype=torch.int32, device=cdev)
    rcnt = torch.empty((world, spp), dtype=torch.int32, device=cdev)
    roff = torch.empty((world, spp), dtype=torch.int32, device=cdev)
    works = [dist.all_to_all_single(rrec, to_comm(srec), output_split_sizes=[int(n) * 2 for n in recv_n],
                                    input_split_sizes=[int(n) * 2 for n in send_n], group=group, async_op=True),
             dist.all_to_all_single(rmsk, to_comm(smsk), output_split_sizes=[int(n) for n in recv_n],
                                    input_split_sizes=[int(n) for n in send_n], group=group, async_op=True),
             dist.all_to_all_single(rcnt.view(-1), to_comm(cnt.view(-1)), group=group, async_op=True),
             dist.all_to_all_single(roff.view(-1), to_comm(off.view(-1)), group=group, async_op=True)]
    local = overlap() if overlap is not None else None      # the rank's own steps 1-6 run under the exchange
    for w in works:
        w.wait()
    rrec, rmsk, rcnt, roff = from_comm(rrec), from_comm(rmsk), from_comm(rcnt), from_comm(roff)
    ops.before_wrap()
    trace("all_to_all")
    pieces, at = [], 0
    for i in range(world):
        if i == rank:
            pieces.append((rec[rank].data_ptr(), msk[rank].data_ptr(), cnt[rank].data_ptr(), off[rank].data_ptr()))
        else:
            pieces.append((rrec.data_ptr() + 16 * at, rmsk.data_ptr() + 4 * at, rcnt[i].data_ptr(), roff[i].data_ptr()))
        at += int(recv_n[i])
    hist = eng.skm_phased_histogram(k, pieces, spp, cs, hist_len)
    trace("union")
    ht = torch.from_numpy(hist.astype(np.int64)).to(cdev)
    dist.all_reduce(ht, op=dist.ReduceOp.SUM, group=group)
    out = ht.cpu().numpy().astype(np.uint64)
    trace("all_reduce")
    trace.report()
    return out if overlap is None else (out, local)


def exp1_step(eng, seqs, group_of: Sequence[int], k: int, cs: int = 5000, hist_len: int = 5001,
              group=None):
    """One benchmark step on N GPUs: steps 1-6 locally, steps 7-8 through the exchange.
    Returns the same dict as Engine.exp1_run (across_hist is the GLOBAL histogram)."""
    device = torch.device("cuda", eng.device)
    ops = EngineOps(eng, device, stage_on_host=dist.get_backend(group) == "gloo")
    if k <= TABLE_MAX_K:      # small k: the bitmap / table form may win, it needs the group sets
        res = eng.exp1_run(seqs, group_of, k, cs=cs, hist_len=hist_len, want_sets=True, across=False)
        gsets = [s.set_counts(1) for s in res["group_sets"]]
        res["across_hist"] = across_groups_auto(ops, gsets, k, cs, hist_len, group)
        del res["group_sets"]
        return res
    import os
    lo, hi = eng.SKM_EXCHANGE_K
    if lo <= k <= hi and max(group_of) < 32 and not os.environ.get("KHOICE_DIST_SET_EXCHANGE"):
        # the super-k-mer form on every rank: steps 1-6 locally (no set is built), steps 7-8 by exchange of records
        # (the local run is handed to the exchange: it runs while the packed records travel)
        across, res = across_records_exchange(ops, eng, seqs, group_of, k, cs, hist_len, group,
                                              overlap=lambda: eng.exp1_run(seqs, group_of, k, cs=cs, hist_len=hist_len, across=False))
        res["across_hist"] = across
        res["exchange"] = "records"
        return res
    # the fused local step, emitting only this rank's across-group set (counter = local groups)
    res = eng.exp1_run(seqs, group_of, k, cs=cs, hist_len=hist_len, across=False, want_across_set=True)
    aset = res.pop("across_set")
    res["across_hist"] = across_set_exchange(ops, aset, k, cs, hist_len, group)
    aset.free()
    return res
