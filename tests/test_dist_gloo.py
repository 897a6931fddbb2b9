"""The N>1 path on CPU: khoice_amd.dist.across_groups_distributed under world_size-2 and -3 gloo,
with an oracle-backed stand-in for the engine (tests only).  Checks the exchange plumbing:
slot bounds, variable-size all-to-all, received slices inside the owner's slot, histogram
all-reduce, against the single-process oracle answer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from khoice_amd import synth
from oracle import kmer_oracle as O

K = 21


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleOps:
    """numpy stand-in: a set is (mixed keys uint64[n] ascending, counts uint32[n] or None)."""
    device = torch.device("cpu")

    def __init__(self, k, rank=0, world=1):
        self.k, self.rank, self.world = k, rank, world

    def words(self, k):
        return 1

    def has_counts(self, s):
        return s[1] is not None

    def slot(self, keys, nparts):
        n = 2 * self.k
        top32 = keys >> np.uint64(n - 32) if n >= 32 else keys << np.uint64(32 - n)
        return (top32 * np.uint64(nparts)) >> np.uint64(32)

    def partition_bounds(self, s, nparts):
        return np.searchsorted(self.slot(s[0], nparts), np.arange(nparts + 1), side="left")

    def export_range(self, s, lo, hi, keys_t, counts_t):
        keys_t.copy_(torch.from_numpy(s[0][lo:hi].view(np.int64).copy()))
        if counts_t is not None:
            c = s[1][lo:hi] if s[1] is not None else np.ones(hi - lo, dtype=np.uint32)
            counts_t.copy_(torch.from_numpy(c.view(np.int32).copy()))

    def view_range(self, s, lo, hi):
        assert (self.slot(s[0][lo:hi], self.world) == self.rank).all()
        return s[0][lo:hi], (s[1][lo:hi] if s[1] is not None else None)

    def export_all(self, s):
        c = s[1] if s[1] is not None else np.ones(s[0].size, dtype=np.uint32)
        return torch.from_numpy(s[0].view(np.int64).copy()), torch.from_numpy(c.view(np.int32).copy())

    def flush(self):
        pass

    def before_wrap(self):
        pass

    def wrap(self, k, n, keys_t, counts_t):
        keys = keys_t.numpy().view(np.uint64).copy()
        assert (self.slot(keys, self.world) == self.rank).all()      # inside this rank's slot
        assert keys.size < 2 or (keys[1:] > keys[:-1]).all()         # sorted and distinct
        return keys, (counts_t.numpy().view(np.uint32).copy() if counts_t is not None else None)

    def union_hist(self, sets, cs, hist_len):
        keys = np.concatenate([s[0] for s in sets])
        cnts = np.concatenate([s[1] if s[1] is not None else np.ones(s[0].size, dtype=np.uint32) for s in sets])
        u, inv = np.unique(keys, return_inverse=True)
        c = np.zeros(u.size, dtype=np.int64)
        np.add.at(c, inv, cnts.astype(np.int64))
        c = np.minimum(c, cs)
        return (u, c.astype(np.uint32)), np.bincount(np.minimum(c, hist_len - 1), minlength=hist_len).astype(np.uint64)

    # direct-addressed occurrence table (small k)
    def set_len(self, s):
        return int(s[0].size)

    def table_add(self, s, table_t):
        from khoice_amd import engine as E
        v = np.array([int(E.unmix_host(self.k, np.array([m], dtype=np.uint64))[0]) for m in s[0]], dtype=np.int64)
        assert np.unique(v).size == v.size and (v < 4 ** self.k).all()
        t = table_t.numpy()
        t[v] += 1

    def table_hist(self, table_t, lo, hi, cs, hist_len):
        assert lo % 16 == 0
        c = table_t.numpy()[lo:hi].astype(np.int64)
        c = np.minimum(c[c > 0], min(cs, hist_len - 1))
        return np.bincount(c, minlength=hist_len).astype(np.uint64)


CopyAllOps = type("CopyAllOps", (OracleOps,), {"view_range": property()})   # hasattr() -> False


def group_db(species, n_genomes=2, length=4000, k=K):
    anc = synth.ancestor(species, length)
    dbs = [O.set_counts(O.count_records([s.decode() for _, s in synth.genome_records(species, g, length, anc)], k), 1)
           for g in range(n_genomes)]
    return O.set_counts(O.union_sum(dbs, 5000), 1)


def to_mixed(db, k=K):
    from khoice_amd import engine as E
    keys = np.array(sorted(db), dtype=np.uint64)
    mixed = np.array([int(E.mix_host(k, np.array([v], dtype=np.uint64))[0]) for v in keys], dtype=np.uint64)
    return np.sort(mixed), None


def worker(rank, world, port, groups_per_rank, with_counts, q):
    from khoice_amd import dist as kdist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        ops = (CopyAllOps if with_counts else OracleOps)(K, rank, world)
        first = 1 + sum(groups_per_rank[:rank])
        mine = [to_mixed(group_db(first + g)) for g in range(groups_per_rank[rank])]
        if with_counts and mine:      # make one rank's sets carry counters: they must travel
            mine[0] = (mine[0][0], np.ones(mine[0][0].size, dtype=np.uint32))
        hist = kdist.across_groups_distributed(ops, mine, K, 5000, 64)
        q.put((rank, hist.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout,with_counts", [([3, 3], False), ([2, 0, 3], True)])
def test_across_groups_exchange_matches_single_process(layout, with_counts):
    from khoice_amd import build as kbuild
    kbuild.build_library()
    world = len(layout)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, layout, with_counts, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = sum(layout)
    want = O.histogram(O.union_sum([group_db(1 + g) for g in range(total)], 5000), 63)
    for _, h in got:
        assert h == want
    assert sum(want[2:]) > 0          # the shared block makes some k-mers multi-group


def aset_worker(rank, world, port, groups_per_rank, q):
    from khoice_amd import dist as kdist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        ops = OracleOps(K, rank, world)
        first = 1 + sum(groups_per_rank[:rank])
        # this rank's across-group set: union of its group sets, counter = local groups holding the key
        local = O.union_sum([group_db(first + g) for g in range(groups_per_rank[rank])], 5000) \
            if groups_per_rank[rank] else {}
        from khoice_amd import engine as E
        keys = np.array(sorted(local), dtype=np.uint64)
        mixed = np.array([int(E.mix_host(K, np.array([v], dtype=np.uint64))[0]) for v in keys], dtype=np.uint64)
        order = np.argsort(mixed)
        aset = (mixed[order], np.array([local[int(v)] for v in keys], dtype=np.uint32)[order])
        hist = kdist.across_set_exchange(ops, aset, K, 5000, 64)
        q.put((rank, hist.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout", [[3, 3], [2, 0, 3]])
def test_across_set_exchange_matches_single_process(layout):
    """The exchange the benchmark takes at N > 1: every rank ships slices of its local across-group
    set (keys + counters); the owners sum the counters."""
    from khoice_amd import build as kbuild
    kbuild.build_library()
    world = len(layout)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=aset_worker, args=(r, world, port, layout, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = O.histogram(O.union_sum([group_db(1 + g) for g in range(sum(layout))], 5000), 63)
    for _, h in got:
        assert h == want
    assert sum(want[2:]) > 0


def table_worker(rank, world, port, groups_per_rank, k, q):
    from khoice_amd import dist as kdist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        ops = OracleOps(k, rank, world)
        first = 1 + sum(groups_per_rank[:rank])
        mine = [to_mixed(group_db(first + g, k=k), k) for g in range(groups_per_rank[rank])]
        h_table = kdist.across_groups_table(ops, mine, k, 5000, 64)
        h_slots = kdist.across_groups_distributed(ops, mine, k, 5000, 64)
        h_auto = kdist.across_groups_auto(ops, mine, k, 5000, 64)
        q.put((rank, h_table.tolist(), h_slots.tolist(), h_auto.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout,k", [([2, 2], 7), ([1, 0, 3], 9)])
def test_small_k_table_all_reduce_matches_slot_exchange_and_oracle(layout, k):
    """presence bitmaps summed by all-reduce == the slot exchange == the oracle."""
    from khoice_amd import build as kbuild
    kbuild.build_library()
    world = len(layout)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=table_worker, args=(r, world, port, layout, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = O.histogram(O.union_sum([group_db(1 + g, k=k) for g in range(sum(layout))], 5000), 63)
    for _, ht, hs, ha in got:
        assert ht == want and hs == want and ha == want
    assert sum(want[2:]) > 0


class RecordStandIn:
    """Oracle-backed stand-in for the three engine calls of the RECORDS exchange (khoice_amd.dist.across_records_exchange):
    a "record" is one canonical k-mer (its code in the first word), its mask the local groups that hold it; slots are a
    hash of the k-mer.  What is under test is the Python side: the agreed geometry, the packed parts, the four all-to-alls
    with their split sizes, the per-slot counts / offsets, the pieces handed to the owner."""
    SKM_EXCHANGE_K = (20, 32)

    def __init__(self, rank, world):
        self.rank, self.world, self.nslots = rank, world, None

    @staticmethod
    def _view(ptr, n, dtype):
        import ctypes
        if n == 0:
            return np.zeros(0, dtype=dtype)
        return np.frombuffer((ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr), dtype=dtype)

    def skm_exchange_plan(self, k, positions_max, fan_max, nparts):
        nslots = max(nparts, positions_max // 50 + 7)
        spp = (nslots + nparts - 1) // nparts
        self.plan = (nslots, spp, positions_max + 64)
        return self.plan

    def slot_of(self, codes, nslots):
        x = (codes * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(20)
        return (x % np.uint64(nslots)).astype(np.int64)

    def skm_pack(self, seqs, tag_of, k, nslots, nparts, part_cap, rec_ptr, mask_ptr, count_ptr, off_ptr):
        assert (nslots, (nslots + nparts - 1) // nparts, part_cap) == self.plan
        spp = self.plan[1]
        masks = {}
        for s, t in zip(seqs, tag_of):
            for code in O.count_records(s.decode().split("\n"), k):
                masks[code] = masks.get(code, 0) | (1 << t)
        codes = np.array(sorted(masks), dtype=np.uint64)
        slot = self.slot_of(codes, nslots)
        order = np.lexsort((codes, slot))
        codes, slot = codes[order], slot[order]
        rec = self._view(rec_ptr, nparts * part_cap * 2, np.int64).reshape(nparts, part_cap, 2)
        msk = self._view(mask_ptr, nparts * part_cap, np.int32).reshape(nparts, part_cap)
        cnt = self._view(count_ptr, nparts * spp, np.int32).reshape(nparts, spp)
        off = self._view(off_ptr, nparts * spp, np.int32).reshape(nparts, spp)
        cnt[:] = 0
        off[:] = 0
        part_n = np.zeros(nparts, dtype=np.uint64)
        for p in range(nparts):
            sel = (slot // spp) == p
            n = int(sel.sum())
            part_n[p] = n
            rec[p, :n, 0] = codes[sel].view(np.int64)
            rec[p, :n, 1] = 0x5A5A
            msk[p, :n] = np.array([masks[int(c)] for c in codes[sel]], dtype=np.int64).astype(np.int32)
            local = slot[sel] - p * spp
            cnt[p] = np.bincount(local, minlength=spp)[:spp]
            off[p] = np.concatenate([[0], np.cumsum(cnt[p])[:-1]])
        return part_n

    def skm_phased_histogram(self, k, pieces, nslots, cs, hist_len):
        nslots_all, spp, _ = self.plan
        assert nslots == spp and len(pieces) == self.world
        tags = {}
        for rp, mp_, cp, op in pieces:
            cnt = self._view(cp, spp, np.int32)
            off = self._view(op, spp, np.int32)
            total = int(cnt.sum())
            assert total == 0 or int((off + cnt).max()) == total
            rec = self._view(rp, total * 2, np.int64).reshape(total, 2)
            msk = self._view(mp_, total, np.int32)
            assert (rec[:, 1] == 0x5A5A).all()
            codes = rec[:, 0].view(np.uint64)
            assert (self.slot_of(codes, nslots_all) // spp == self.rank).all()        # only this owner's slots arrive
            for s in range(spp):                                                    # and the per-slot tables are right
                seg = codes[off[s]:off[s] + cnt[s]]
                assert (self.slot_of(seg, nslots_all) == self.rank * spp + s).all()
            for c, m in zip(codes.tolist(), msk.tolist()):
                tags[c] = tags.get(c, 0) + bin(m & 0xffffffff).count("1")
        h = np.zeros(hist_len, dtype=np.uint64)
        for n in tags.values():
            h[min(n, cs, hist_len - 1)] += 1
        return h


def records_worker(rank, world, port, groups_per_rank, q):
    from khoice_amd import dist as kdist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        first = 1 + sum(groups_per_rank[:rank])
        seqs, group_of = [], []
        for g in range(groups_per_rank[rank]):
            anc = synth.ancestor(first + g, 3000)
            for j in range(2):
                seqs.append(synth.clean_text(synth.genome_records(first + g, j, 3000, anc)))
                group_of.append(g)
        if not seqs:                       # a rank without groups still takes part in every collective
            seqs, group_of = [b"ACGT"], [0]
        ops = OracleOps(K, rank, world)
        hist = kdist.across_records_exchange(ops, RecordStandIn(rank, world), seqs, group_of, K, 5000, 64)
        q.put((rank, hist.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout", [[3, 3], [2, 0, 3]])
def test_records_exchange_matches_single_process(layout):
    """The exchange the benchmark takes at N > 1 for 20 <= k <= 32 (minimizer records by slot owner), on CPU."""
    world = len(layout)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=records_worker, args=(r, world, port, layout, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = O.histogram(O.union_sum([group_db(1 + g, length=3000) for g in range(sum(layout))], 5000), 63)
    for _, h in got:
        assert h == want
    assert sum(want[2:]) > 0
