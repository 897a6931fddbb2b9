"""GPU parity: the HIP engine (through the C ABI) against the brute-force oracle on the
same seeded inputs.  Bit-exact: k-mer sets, counters, histograms, text dumps."""
import os
import random

import numpy as np
import pytest

from oracle import kmer_oracle as O
from tests.util import check_multiset_case, db_to_arrays, random_dna, set_to_db

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from khoice_amd import build as kbuild
    from khoice_amd import engine as E
    kbuild.build_library()
    e = E.Engine(0)
    yield e
    e.close()


def messy_fasta(rng, n_records, rec_len, with_n=True, lower=True):
    """Multi-FASTA text with N runs, IUPAC symbols, lower case and odd line lengths."""
    out = []
    for r in range(n_records):
        seq = list(random_dna(rng, rng.randrange(rec_len // 2, rec_len)))
        if with_n:
            for _ in range(3):
                at = rng.randrange(0, max(1, len(seq) - 10))
                for j in range(at, min(len(seq), at + rng.randrange(1, 9))):
                    seq[j] = rng.choice("NRYKM")
        if lower:
            at = rng.randrange(0, max(1, len(seq) - 50))
            for j in range(at, min(len(seq), at + 50)):
                seq[j] = seq[j].lower()
        seq = "".join(seq)
        width = rng.choice([60, 70, 80])
        out.append(f">rec{r} some description\n")
        out.extend(seq[i:i + width] + "\n" for i in range(0, len(seq), width))
    return "".join(out).encode()


def clean(fasta: bytes) -> bytes:
    return "\n".join(O.fasta_records(fasta)).encode()


def test_build_matches_reference_multisets(eng, golden):
    """K1 through the C ABI against reference-run output: tests/golden/kmer_multiset.json holds, for ACGT strings,
    collections.Counter(get_canonical_kmer(w) for w in process_read_into_kmers(s, k)) computed by the reference's
    own src/merge_lists.py:53-73 (repeats, both strands, even-k palindromes, k = 7 .. 63)."""
    cases = golden("kmer_multiset.json")["cases"]
    for k in sorted({c["k"] for c in cases}):
        mine = [c for c in cases if c["k"] == k]
        for cs in (0x7fffffff, 255):   # counters unsaturated, then KMC's default ceiling
            sets = eng.build_batch([c["seq"].encode() for c in mine], k, cs=cs)
            for case, st in zip(mine, sets):
                keys, counts = st.download_sorted()
                check_multiset_case(case, keys, counts, cs)
    # the fused experiment-type-1 path on the same strings: one genome per group -> the group's histogram is the
    # histogram of a set, the distinct count is the multiset's size
    for k in (21, 31, 41, 63):
        mine = [c for c in cases if c["k"] == k]
        res = eng.exp1_run([c["seq"].encode() for c in mine], list(range(len(mine))), k, cs=5000, hist_len=16)
        assert [int(x) for x in res["distinct_per_seq"]] == [c["distinct"] for c in mine]
        assert [int(x) for x in res["within_hist"][:, 1]] == [c["distinct"] for c in mine]


@pytest.mark.parametrize("k", [1, 3, 7, 15, 21, 31, 32, 33, 41, 63, 64])
def test_build_matches_oracle(eng, k):
    rng = random.Random(1000 + k)
    fasta = messy_fasta(rng, 4, 6000)
    want = O.build(fasta, k)
    got = eng.build(clean(fasta), k)
    assert set_to_db(got) == want
    # sets (-ci1 then set_counts 1) built without counters
    got1 = eng.build(clean(fasta), k, with_counts=False)
    assert set_to_db(got1) == O.set_counts(want, 1)


def test_build_from_fasta_files(eng, tmp_path):
    import gzip
    rng = random.Random(7)
    fasta = messy_fasta(rng, 3, 5000)
    p = tmp_path / "g.fna.gz"
    with gzip.open(p, "wb") as fh:
        fh.write(fasta)
    q = tmp_path / "g.fa"
    q.write_bytes(fasta)
    want = O.build(fasta, 21)
    assert set_to_db(eng.build_fasta(str(p), 21)) == want
    assert set_to_db(eng.build_fasta(str(q), 21)) == want
    assert eng.read_fasta(str(p)) == clean(fasta)


@pytest.mark.parametrize("k", [5, 31, 41])
def test_build_edge_cases(eng, k):
    cases = {
        "empty": b"",
        "short": b"ACGT"[: max(0, k - 1)] if k <= 5 else b"A" * (k - 1),
        "exact": (b"ACGTTGCA" * 9)[:k],
        "all_n": b"N" * 500,
        "breaks": (b"ACGTACGTAC" * 8)[:k] + b"N" + (b"TTGACCA" * 12)[:k - 1] + b"\n" + (b"GATTACA" * 12)[:k + 3],
        "palin": b"ACGT" * 40,
        "poly_a": b"A" * 70000,                       # one key, heavy duplicates, oversize bucket
        "two_keys": (b"A" * 30000) + b"N" + (b"C" * 30000),
    }
    for name, seq in cases.items():
        want = O.count_records(seq.decode().split("\n"), k)
        got = set_to_db(eng.build(seq, k))
        assert got == want, name
    # saturation above and below the default 255
    seq = b"A" * 70000
    assert set_to_db(eng.build(seq, k, cs=5000)) == O.count_records([seq.decode()], k, cs=5000)
    assert set_to_db(eng.build(seq, k, cs=100000)) == O.count_records([seq.decode()], k, cs=100000)


def test_low_complexity_small_k_duplicates(eng):
    # k=7 on 300 kbp: 8192 possible canonical keys, ~37 copies each -> duplicate-heavy buckets
    rng = random.Random(42)
    seq = random_dna(rng, 300_000, "AACGTT")
    for k in (5, 7, 9):
        want = O.count_records([seq], k, cs=1 << 30)
        assert set_to_db(eng.build(seq.encode(), k, cs=1 << 30)) == want


def test_ci_cx_filters(eng):
    rng = random.Random(3)
    seq = random_dna(rng, 30000, "ACGT")
    for ci, cx in ((2, 0xFFFFFFFF), (1, 3), (2, 2), (5, 9)):
        want = O.count_records([seq], 5, ci=ci, cx=cx if cx != 0xFFFFFFFF else 10**9, cs=1 << 30)
        got = set_to_db(eng.build(seq.encode(), 5, ci=ci, cx=cx, cs=1 << 30))
        assert got == want, (ci, cx)


def test_batch_equals_individual_and_is_deterministic(eng):
    rng = random.Random(11)
    seqs = [random_dna(rng, n, "ACGTN" if i % 2 else "ACGT").encode()
            for i, n in enumerate([0, 10, 40, 3000, 70000, 150000, 66000])]
    k = 31
    batch = eng.build_batch(seqs, k)
    for s, b in zip(seqs, batch):
        assert set_to_db(b) == O.count_records(s.decode().split("\n"), k)
    again = eng.build_batch(seqs, k)
    for a, b in zip(batch, again):
        ka, ca = a.download()
        kb, cb = b.download()
        assert (ka == kb).all() and (ca == cb).all()


@pytest.mark.parametrize("k", [9, 31, 41])
def test_union_sum_and_fused_histogram(eng, k):
    rng = random.Random(500 + k)
    base = random_dna(rng, 20000)
    dbs, sets = [], []
    for g in range(5):
        s = list(base)
        for _ in range(200):
            s[rng.randrange(len(s))] = rng.choice("ACGT")
        s = "".join(s)
        dbs.append(O.set_counts(O.count_records([s], k), 1))
        sets.append(eng.build(s.encode(), k).set_counts(1))
    want = O.union_sum(dbs, 5000)
    got, hist = eng.union_sum(sets, 5000, hist_len=5001)
    assert set_to_db(got) == want
    assert [int(x) for x in hist] == O.histogram(want, 5000)
    assert [int(x) for x in got.histogram(5001)] == O.histogram(want, 5000)
    # single-input complex "(set1 )" (exp_type_1.smk:54-58 with one genome)
    one, h1 = eng.union_sum(sets[:1], 5000, hist_len=256)
    assert set_to_db(one) == dbs[0]
    assert int(h1[1]) == len(dbs[0])
    # saturation: counted inputs, tiny cs
    counted = [eng.build((base + "N" + base).encode(), k), eng.build(base.encode(), k)]
    cdb = [O.count_records([base, base], k), O.count_records([base], k)]
    for cs in (2, 3, 255, 5000):
        assert set_to_db(eng.union_sum(counted, cs)) == O.union_sum(cdb, cs)


def test_union_fan_in_above_one_launch(eng):
    rng = random.Random(77)
    k = 15
    base = random_dna(rng, 3000)
    dbs, sets = [], []
    for g in range(150):
        s = list(base)
        for _ in range(30):
            s[rng.randrange(len(s))] = rng.choice("ACGT")
        s = "".join(s)
        dbs.append(O.set_counts(O.count_records([s], k), 1))
        sets.append(eng.build(s.encode(), k, with_counts=False))
    for cs in (50, 5000):
        got, hist = eng.union_sum(sets, cs, hist_len=5001)
        want = O.union_sum(dbs, cs)
        assert set_to_db(got) == want
        assert [int(x) for x in hist] == O.histogram(want, 5000)


@pytest.mark.parametrize("k", [7, 31, 63])
def test_simple_operations(eng, k):
    from khoice_amd import engine as E
    rng = random.Random(900 + k)
    a_seq = random_dna(rng, 15000)
    b_seq = a_seq[:7000] + random_dna(rng, 9000)
    da, db = O.count_records([a_seq, a_seq[:3000]], k), O.count_records([b_seq], k)
    a = eng.build((a_seq + "\n" + a_seq[:3000]).encode(), k)
    b = eng.build(b_seq.encode(), k)
    for mode in ("min", "max", "sum", "diff", "left", "right"):
        assert set_to_db(eng.intersect(a, b, mode)) == O.intersect(da, db, mode), mode
        assert set_to_db(eng.simple(a, b, E.UNION, mode)) == O.union2(da, db, mode), mode
    assert set_to_db(eng.intersect(a, b, "sum", cs=3)) == O.intersect(da, db, "sum", cs=3)
    assert set_to_db(eng.kmers_subtract(a, b)) == O.kmers_subtract(da, db)
    assert set_to_db(eng.kmers_subtract(b, a)) == O.kmers_subtract(db, da)
    assert set_to_db(eng.simple(a, b, E.COUNTERS_SUBTRACT, "min")) == O.counters_subtract(da, db)
    # reference invariants (exp_type_2.smk:183-184): pivot set vs group union with -ocsum
    pa, pb = a.set_counts(1), b.set_counts(1)
    inter = eng.intersect(pa, pb, "sum")
    sub = eng.kmers_subtract(pa, pb)
    hi, hs = inter.histogram(256), sub.histogram(256)
    assert int(hi[1]) == 0 and int(hs[2:].sum()) == 0
    assert int(hi.sum()) + int(hs.sum()) == len(da)
    # empty operands
    empty = eng.build(b"", k)
    assert set_to_db(eng.intersect(a, empty, "sum")) == {}
    assert set_to_db(eng.kmers_subtract(a, empty)) == da
    assert set_to_db(eng.union_sum([empty, empty], 5000)) == {}


def test_k_mismatch_is_an_error(eng):
    from khoice_amd import engine as E
    a = eng.build(b"ACGTACGTACGTACGT", 5)
    b = eng.build(b"ACGTACGTACGTACGT", 7)
    with pytest.raises(E.KhoiceError):
        eng.intersect(a, b)
    with pytest.raises(E.KhoiceError):
        eng.build(b"ACGT", 65)
    with pytest.raises(E.KhoiceError):
        eng.build(b"ACGT", 0)


@pytest.mark.parametrize("k", [21, 41])
def test_files_roundtrip_and_text_outputs(eng, k, tmp_path):
    rng = random.Random(k)
    fasta = messy_fasta(rng, 3, 4000)
    want = O.build(fasta, k)
    s = eng.build(clean(fasta), k)
    prefix = str(tmp_path / "db")
    s.save(prefix)
    assert os.path.exists(prefix + ".kmc_pre") and os.path.exists(prefix + ".kmc_suf")
    back = eng.load(prefix)
    assert set_to_db(back) == want
    s.set_counts(1).save(prefix + "_set")
    assert set_to_db(eng.load(prefix + "_set")) == O.set_counts(want, 1)
    dump = tmp_path / "dump.txt"
    back.dump_sorted(str(dump))
    assert dump.read_text() == O.dump_sorted_text(want, k)
    hist = tmp_path / "hist.txt"
    back.histogram_file(255, str(hist))
    assert hist.read_text() == O.histogram_text(want, 255)
    # upload / download round trip
    keys, counts = db_to_arrays(want, k)
    up = eng.upload(k, keys, counts)
    assert set_to_db(up) == want
    ks, cs = up.download_sorted()
    assert (ks == keys).all() and (cs == counts).all()
    from khoice_amd import engine as E
    with pytest.raises(E.KhoiceError):
        eng.load(str(tmp_path / "nope"))


def test_exp1_fused_matches_oracle_pipeline(eng):
    from khoice_amd import synth
    k, L = 21, 30000
    groups = synth.species_set(3, 3, L)
    seqs = [t for _, _, t in groups]
    group_of = [s - 1 for s, _, _ in groups]
    res = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=5001, want_sets=True)
    per_genome = [O.set_counts(O.count_records(t.decode().split("\n"), k), 1) for t in seqs]
    assert [int(x) for x in res["distinct_per_seq"]] == [len(d) for d in per_genome]
    unions = []
    for g in range(3):
        u = O.union_sum([per_genome[i] for i in range(len(seqs)) if group_of[i] == g], 5000)
        unions.append(u)
        assert [int(x) for x in res["within_hist"][g]] == O.histogram(u, 5000)
        assert set_to_db(res["group_sets"][g]) == u
    across = O.union_sum([O.set_counts(u, 1) for u in unions], 5000)
    assert [int(x) for x in res["across_hist"]] == O.histogram(across, 5000)
    assert set_to_db(res["across_set"]) == across
    assert int(res["across_hist"][2:].sum()) > 0      # the shared block is visible


def test_full_size_properties(eng):
    """Size-independent checks at a scale the Python oracle cannot reach."""
    from khoice_amd import synth
    k, L = 31, 1_000_000
    items = synth.species_set(2, 2, L)
    seqs = [t for _, _, t in items]
    sets = eng.build_batch(seqs, k, cs=1 << 30)
    for t, s in zip(seqs, sets):
        a = np.frombuffer(t, dtype=np.uint8)
        ok = np.isin(a, np.frombuffer(b"ACGT", dtype=np.uint8)).astype(np.int64)
        c = np.concatenate([[0], np.cumsum(ok)])
        valid = int(((c[k:] - c[:-k]) == k).sum())
        keys, counts = s.download()
        assert int(counts.sum()) == valid                      # every window counted once
        assert len(np.unique(keys[:, 0])) == len(keys)         # keys distinct
    a, b = sets[0].set_counts(1), sets[1].set_counts(1)
    u, hist = eng.union_sum([a, b], 5000, hist_len=16)
    i = eng.intersect(a, b, "sum")
    d = eng.kmers_subtract(a, b)
    assert len(u) == len(a) + len(b) - len(i)
    assert len(d) == len(a) - len(i)
    assert int(hist[1]) + int(hist[2]) == len(u) and int(hist[2]) == len(i)
    assert len(eng.intersect(a, a, "min")) == len(a)
    assert len(eng.kmers_subtract(a, a)) == 0
    uu, h2 = eng.union_sum([a, a, a], 5000, hist_len=16)
    assert int(h2[3]) == len(a) and int(h2.sum()) == len(a)
    # species are independent apart from the shared block: cross-species overlap is small
    x = eng.intersect(sets[0].set_counts(1), sets[2].set_counts(1), "sum")
    assert 0 < len(x) < 0.1 * len(a)


def test_distributed_step_world1_matches_local(eng):
    """khoice_amd.dist over RCCL with a single rank: the exchange degenerates to a self
    all-to-all but exercises export / import / union of received slices on the device."""
    import torch
    import torch.distributed as dist
    from khoice_amd import dist as kdist
    from khoice_amd import synth
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for k in (21, 41):
            items = synth.species_set(3, 2, 60_000)
            seqs = [t for _, _, t in items]
            group_of = [s - 1 for s, _, _ in items]
            want = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=5001)
            got = kdist.exp1_step(eng, seqs, group_of, k, cs=5000, hist_len=5001)
            assert (got["across_hist"] == want["across_hist"]).all()
            assert (got["within_hist"] == want["within_hist"]).all()
            assert (got["distinct_per_seq"] == want["distinct_per_seq"]).all()
        # small k: the table form (all-reduce of summed presence bitmaps) and the automatic choice
        ops = kdist.EngineOps(eng, torch.device("cuda", 0))
        for k in (9, 13):
            items = synth.species_set(3, 2, 60_000)
            seqs = [t for _, _, t in items]
            group_of = [s - 1 for s, _, _ in items]
            want = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=5001, want_sets=True)
            gsets = [g.set_counts(1) for g in want["group_sets"]]
            assert (kdist.across_groups_table(ops, gsets, k, 5000, 5001) == want["across_hist"]).all()
            assert (kdist.across_groups_distributed(ops, gsets, k, 5000, 5001) == want["across_hist"]).all()
            assert (kdist.exp1_step(eng, seqs, group_of, k)["across_hist"] == want["across_hist"]).all()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,cell", [(5, 1), (11, 1), (11, 4), (15, 1), (16, 1)])
def test_occurrence_table_matches_union_sum_and_oracle(eng, k, cell):
    """kh_table_add_set / kh_table_histogram: cell v = number of sets holding
    canonical k-mer v; histogram == the fused union-sum histogram == the oracle's."""
    import torch
    rng = random.Random(1000 + k)
    base = random_dna(rng, 30_000)
    texts = []
    for g in range(4):
        t = list(base)
        for _ in range(300 * g):
            t[rng.randrange(len(t))] = rng.choice("ACGT")
        texts.append("".join(t) + "N" + random_dna(rng, 2_000 * g))
    sets = [eng.build(t.encode(), k).set_counts(1) for t in texts]
    table = torch.zeros(4 ** k, dtype=torch.uint8 if cell == 1 else torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    for s_ in sets:
        eng.table_add_set(s_, table.data_ptr(), cell)
    eng.sync()
    dbs = [O.set_counts(O.count_records([t], k), 1) for t in texts]
    udb = O.union_sum(dbs, 5000)
    # the cells themselves
    assert int(torch.count_nonzero(table)) == len(udb)
    some = list(udb.items())[:2000]
    idx = torch.tensor([key for key, _ in some], dtype=torch.int64, device="cuda:0")
    assert table[idx].cpu().tolist() == [c for _, c in some]
    _, hu = eng.union_sum(sets, 5000, hist_len=16)
    want = np.array(O.histogram(udb, 15), dtype=np.uint64)
    h = eng.table_histogram(table.data_ptr(), cell, 0, 4 ** k, 5000, 16)
    assert (h == want).all() and int(h[0]) == 0
    assert (h == hu).all()
    # split ranges (what each rank histograms) add up; cs caps the counter like -cs does
    if 4 ** k >= 64:
        mid = (4 ** k // 3) // 16 * 16
        h1 = eng.table_histogram(table.data_ptr(), cell, 0, mid, 5000, 16)
        h2 = eng.table_histogram(table.data_ptr(), cell, mid, 4 ** k, 5000, 16)
        assert (h1 + h2 == h).all()
    hc = eng.table_histogram(table.data_ptr(), cell, 0, 4 ** k, 2, 16)
    assert int(hc[1]) == int(h[1]) and int(hc[2]) == int(h[2:].sum()) and int(hc[3:].sum()) == 0
    # a k above the table limit is refused, loudly
    from khoice_amd.engine import KhoiceError
    big = eng.build(texts[0].encode(), 21)
    with pytest.raises(KhoiceError):
        eng.table_add_set(big, table.data_ptr(), cell)


def test_direct_and_staged_scatter_agree(eng, monkeypatch):
    """pass B has two forms (LDS write combining / direct stores); both must give the same sets."""
    rng = random.Random(321)
    seqs = [random_dna(rng, n, "ACGTN").encode() for n in (200_000, 90_000)]
    seqs.append(random_dna(rng, 700_000).encode())           # several tiles, several staging rounds each
    for k in (31, 41, 63, 64, 33, 1, 16):                    # one- and two-word keys (staged for both)
        monkeypatch.delenv("KHOICE_DIRECT_SCATTER", raising=False)
        a = eng.build_batch(seqs, k)
        monkeypatch.setenv("KHOICE_DIRECT_SCATTER", "1")
        b = eng.build_batch(seqs, k)
        for x, y in zip(a, b):
            kx, cx = x.download()
            ky, cy = y.download()
            assert (kx == ky).all() and (cx == cy).all()


def test_exp1_group_waves_equal_single_wave(eng, monkeypatch):
    """kh_exp1_run processes groups in memory-bounded waves, and a group larger than a wave in
    sub-waves of genomes summed into a running union (the HBM-spill path): any budget gives the
    same result."""
    from khoice_amd import synth
    for k, n_genomes in ((31, 2), (41, 5)):
        items = synth.species_set(4, n_genomes, 40_000)
        seqs = [t for _, _, t in items]
        group_of = [s - 1 for s, _, _ in items]
        want = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=64)
        # 1: one genome per sub-wave; 90000: two genomes per sub-wave (5-genome groups) or one
        # group per wave (2-genome groups); 170000: two groups per wave / sub-waves of four
        for budget in ("1", "90000", "170000"):
            monkeypatch.setenv("KHOICE_WAVE_BASES", budget)
            got = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=64, want_sets=True)
            assert (got["within_hist"] == want["within_hist"]).all()
            assert (got["across_hist"] == want["across_hist"]).all()
            assert (got["distinct_per_seq"] == want["distinct_per_seq"]).all()
            monkeypatch.delenv("KHOICE_WAVE_BASES")
            ref = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=64, want_sets=True)
            for a, b in zip(got["group_sets"], ref["group_sets"]):
                ka, ca = a.download()
                kb, cb = b.download()
                assert (ka == kb).all() and (ca == cb).all()
    # saturation is applied once, at the end: 6 identical genomes, cs = 4
    monkeypatch.setenv("KHOICE_WAVE_BASES", "1")
    seqs = [items[0][2]] * 6
    got = eng.exp1_run(seqs, [0] * 6, 31, cs=4, hist_len=16)
    assert int(got["within_hist"][0][4]) == int(got["distinct_per_seq"][0]) and int(got["within_hist"][0].sum()) == int(got["distinct_per_seq"][0])


def test_long_sequences_are_chunked(eng, monkeypatch):
    """Sequences beyond one segment's bucket table are cut into overlapping chunks whose
    databases add up exactly (forced here by a tiny segment limit)."""
    rng = random.Random(99)
    seqs = [random_dna(rng, n, "ACGTN" if i else "ACGT").encode() for i, n in enumerate((50_000, 9_000, 31_000))]
    for k in (21, 41):
        want = [O.count_records(s.decode().split("\n"), k, cs=1 << 30) for s in seqs]
        monkeypatch.setenv("KHOICE_MAX_SEG_POS", "7000")
        got = eng.build_batch(seqs, k, cs=1 << 30)
        plain = eng.build_batch(seqs, k, with_counts=False)
        monkeypatch.delenv("KHOICE_MAX_SEG_POS")
        for w, g, p in zip(want, got, plain):
            assert set_to_db(g) == w
            assert set_to_db(p) == O.set_counts(w, 1)
        # default saturation applies after the chunks are added up
        monkeypatch.setenv("KHOICE_MAX_SEG_POS", "7000")
        sat = eng.build((b"A" * 30_000), k)
        monkeypatch.delenv("KHOICE_MAX_SEG_POS")
        assert set_to_db(sat) == {0: 255}


def test_sets_may_outlive_their_engine():
    """A binding's garbage collector may free set handles after kh_ctx_destroy: the context
    stays behind as a closed shell until its last buffer is gone."""
    from khoice_amd import engine as E
    e = E.Engine(0)
    s = e.build(("ACGTTGCA" * 500).encode(), 5)
    t = s.set_counts(1)
    u = e.union_sum([t, t], 5000)
    e.close()
    s.free()
    del t
    u.free()
    e2 = E.Engine(0)
    assert len(e2.build(b"ACGTACGTTTGA", 3)) > 0
    e2.close()


def test_ticket_order_gives_the_same_sets(monkeypatch):
    """Parts of the ordered single-pass kernels are taken in workgroup-index order by default and
    by an atomic ticket after a look-back timeout (or with KHOICE_TICKETS=1): same results."""
    from khoice_amd import engine as E
    from khoice_amd import synth
    items = synth.species_set(3, 2, 50_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    with E.Engine(0) as e1:
        want = e1.exp1_run(seqs, group_of, 31, cs=5000, hist_len=64, want_sets=True)
        wk = [g.download() for g in want["group_sets"]]
        assert e1.stats()["order_fallbacks"] == 0
    monkeypatch.setenv("KHOICE_TICKETS", "1")
    with E.Engine(0) as e2:
        got = e2.exp1_run(seqs, group_of, 31, cs=5000, hist_len=64, want_sets=True)
        for (ka, ca), g in zip(wk, got["group_sets"]):
            kb, cb = g.download()
            assert (ka == kb).all() and (ca == cb).all()
        assert (got["within_hist"] == want["within_hist"]).all() and (got["across_hist"] == want["across_hist"]).all()


@pytest.mark.parametrize("k", [21, 41])
def test_union_histogram_chains_equal_union_sum(eng, k):
    """kh_union_histogram runs the slots as independent chains (no compact output set): the
    histogram and the number of output records equal kh_union_sum's; also for fan-in > 128."""
    rng = random.Random(77 + k)
    base = random_dna(rng, 600_000)
    texts = []
    for g in range(6):
        a = rng.randrange(0, 100_000)
        texts.append(base[a:a + 450_000] + "N" + random_dna(rng, 30_000))
    sets = [eng.build(t.encode(), k).set_counts(1) for t in texts]
    u = eng.union_sum(sets, 5000, hist_len=16)
    uset, want = u if isinstance(u, tuple) else (u, None)
    before = eng.stats()["setop_out"]
    got = eng.union_histogram(sets, 5000, 16)
    assert (got == want).all()
    assert eng.stats()["setop_out"] - before == len(uset)      # every record was still produced
    assert int(got.sum()) == len(uset)
    many = [sets[i % 6] for i in range(140)]                  # beyond one launch's fan-in
    _, want2 = eng.union_sum(many, 5000, hist_len=200)
    assert (eng.union_histogram(many, 5000, 200) == want2).all()
    empty = eng.build(b"ACG", k).set_counts(1)
    assert int(eng.union_histogram([empty, empty], 5000, 8).sum()) == 0


def test_corrupt_slot_bounds_fail_closed(eng, monkeypatch):
    """A slot bound outside its operand (planted through the KHOICE_DEBUG_CORRUPT_BOUNDS hook) must
    end in an error from the library — the gather is never run through it — for both forms of the
    operand description (up to 64 operands per wave lane, more through LDS descriptors)."""
    from khoice_amd.engine import KhoiceError
    rng = random.Random(99)
    sets = [eng.build(random_dna(rng, 30_000).encode(), 31).set_counts(1) for _ in range(3)]
    want, _ = eng.union_sum(sets, 255, hist_len=8)
    monkeypatch.setenv("KHOICE_DEBUG_CORRUPT_BOUNDS", "1")
    for operands in (sets, sets * 22):                      # 3 operands / 66 operands
        with pytest.raises(KhoiceError) as ei:
            eng.union_sum(operands, 5000)
        assert "not sorted" in str(ei.value)
    monkeypatch.delenv("KHOICE_DEBUG_CORRUPT_BOUNDS")
    again, _ = eng.union_sum(sets, 255, hist_len=8)          # the context is still usable
    assert len(again) == len(want)
