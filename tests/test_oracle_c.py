"""Pin the C restatement to the Python oracle (which is pinned to the reference's
canonical-form vectors) on seeded inputs; all k widths, all set operations."""
import random

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import kmer_oracle as O
from tests.util import check_multiset_case, db_to_arrays, random_dna


def same(db_c, db_py, k):
    keys, counts = db_c.arrays()
    wk, wc = db_to_arrays(db_py, k)
    assert keys.shape == wk.shape, (keys.shape, wk.shape)
    assert (keys == wk).all() and (counts == wc).all()


@pytest.mark.parametrize("k", [1, 4, 7, 21, 31, 32, 33, 41, 64])
def test_count_matches_python(k):
    rng = random.Random(k)
    seq = random_dna(rng, 5000, "ACGTacgtN") + "\n" + random_dna(rng, 800, "ACGT")
    same(CO.count(seq.encode(), k), O.count_records(seq.split("\n"), k), k)
    same(CO.count(seq.encode(), k, ci=2, cx=5, cs=3), O.count_records(seq.split("\n"), k, ci=2, cx=5, cs=3), k)
    assert CO.count(b"", k).arrays()[0].shape[0] == 0


def test_canonical_golden_through_c(golden):
    for kmer, want in golden("canonical_kmers.json")["canonical"]:
        k = len(kmer)
        keys, counts = CO.count(kmer.encode(), k).arrays()
        code = int(keys[0, 0]) | (int(keys[0, 1]) << 64 if keys.shape[1] == 2 else 0)
        assert O.decode(code, k) == want and counts[0] == 1


@pytest.mark.parametrize("k", [9, 31, 41])
def test_set_ops_match_python(k):
    rng = random.Random(100 + k)
    base = random_dna(rng, 4000)
    seqs = []
    for g in range(4):
        s = list(base)
        for _ in range(60):
            s[rng.randrange(len(s))] = rng.choice("ACGT")
        seqs.append("".join(s) + "\n" + base[:500])
    cdbs = [CO.count(s.encode(), k) for s in seqs]
    pdbs = [O.count_records(s.split("\n"), k) for s in seqs]
    for cs in (2, 255, 5000):
        u = CO.union_sum(cdbs, cs)
        pu = O.union_sum(pdbs, cs)
        same(u, pu, k)
        assert [int(x) for x in u.histogram(cs + 1)] == O.histogram(pu, cs)
    modes = ["min", "max", "sum", "diff", "left", "right"]
    for mi, m in enumerate(modes):
        same(CO.simple(cdbs[0], cdbs[1], 1, mi), O.intersect(pdbs[0], pdbs[1], m), k)
        same(CO.simple(cdbs[0], cdbs[1], 0, mi), O.union2(pdbs[0], pdbs[1], m), k)
    same(CO.simple(cdbs[0], cdbs[1], 2, 0), O.kmers_subtract(pdbs[0], pdbs[1]), k)
    same(CO.simple(cdbs[0], cdbs[1], 3, 0), O.counters_subtract(pdbs[0], pdbs[1]), k)


def test_exp1_matches_python():
    from khoice_amd import synth
    k, L = 21, 8000
    items = synth.species_set(3, 2, L)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    res = CO.exp1(seqs, group_of, k, nthreads=2)
    per = [O.set_counts(O.count_records(t.decode().split("\n"), k), 1) for t in seqs]
    assert [int(x) for x in res["distinct_per_seq"]] == [len(p) for p in per]
    unions = []
    for g in range(3):
        u = O.union_sum([per[i] for i in range(len(seqs)) if group_of[i] == g], 5000)
        unions.append(u)
        assert [int(x) for x in res["within_hist"][g]] == O.histogram(u, 5000)
    across = O.union_sum([O.set_counts(u, 1) for u in unions], 5000)
    assert [int(x) for x in res["across_hist"]] == O.histogram(across, 5000)


def test_kmer_multisets_match_reference_through_c(golden):
    """The C restatement against the reference-run multisets (tests/golden/kmer_multiset.json)."""
    for case in golden("kmer_multiset.json")["cases"]:
        keys, counts = CO.count(case["seq"].encode(), case["k"], cs=0x7fffffff).arrays()
        check_multiset_case(case, keys, counts, 0x7fffffff)
        keys, counts = CO.count(case["seq"].encode(), case["k"]).arrays()   # KMC's default ceiling of 255
        check_multiset_case(case, keys, counts, 255)
