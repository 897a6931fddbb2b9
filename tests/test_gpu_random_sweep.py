"""Seeded random sweep: the fused experiment-type-1 path (kh_exp1_run), the plain set operations
and the histogram-only union against the C restatement on inputs whose shape is drawn at random
— k from 3 to 64, 1-4 groups of 1-6 genomes, 1 kb-300 kb each, shared blocks, N runs, repeats.
Bit-exact histograms, distinct counts and sets."""
import random

import numpy as np
import pytest

from oracle import c_oracle as CO
from tests.util import random_dna

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from khoice_amd import build as kbuild
    from khoice_amd import engine as E
    kbuild.build_library()
    e = E.Engine(0)
    yield e
    e.close()


def draw_case(seed):
    rng = random.Random(0xC0FFEE + seed)
    k = rng.choice([3, 5, 8, 11, 15, 16, 17, 21, 27, 31, 32, 33, 41, 47, 55, 63, 64])
    ngroups = rng.randrange(1, 5)
    shared = random_dna(rng, rng.randrange(200, 5000))
    seqs, group_of = [], []
    for g in range(ngroups):
        length = int(10 ** rng.uniform(3.0, 5.5))
        anc = random_dna(rng, length)
        for _ in range(rng.randrange(1, 7)):
            t = list(anc)
            for _ in range(int(len(t) * rng.choice([0.0, 0.001, 0.01, 0.05]))):
                t[rng.randrange(len(t))] = rng.choice("ACGT")
            for _ in range(rng.randrange(0, 4)):                       # N runs / lower case
                at = rng.randrange(len(t))
                t[at:at + rng.randrange(1, 40)] = "N" * min(rng.randrange(1, 40), len(t) - at)
            s = "".join(t)
            if rng.random() < 0.5:
                s += "N" + shared
            if rng.random() < 0.3:
                s += "\n" + "A" * rng.randrange(k, k + 300)             # a low-complexity record
            if rng.random() < 0.3:
                s = s[:len(s) // 2].lower() + s[len(s) // 2:]
            seqs.append(s.encode())
            group_of.append(g)
    return k, seqs, group_of


@pytest.mark.parametrize("seed", range(40))
def test_exp1_and_set_operations_on_random_shapes(eng, seed):
    k, seqs, group_of = draw_case(seed)
    cs = random.Random(seed).choice([5000, 3, 255])
    want = CO.exp1(seqs, group_of, k, cs=cs, hist_len=300)
    got = eng.exp1_run(seqs, group_of, k, cs=cs, hist_len=300, want_sets=True)
    assert (got["distinct_per_seq"] == want["distinct_per_seq"]).all()
    assert (got["within_hist"] == want["within_hist"]).all()
    assert (got["across_hist"] == want["across_hist"]).all()
    # the same across-group histogram through the histogram-only union and the compact one
    plain = [g.set_counts(1) for g in got["group_sets"]]
    assert (eng.union_histogram(plain, cs, 300) == want["across_hist"]).all()
    u, h = eng.union_sum(plain, cs, hist_len=300)
    assert (h == want["across_hist"]).all() and len(u) == int(want["across_hist"].sum())
    # per-genome sets and a binary operation, key for key
    dbs = [CO.count(s, k) for s in seqs[:3]]
    sets = eng.build_batch(seqs[:3], k, ci=1, with_counts=True)
    for d, s in zip(dbs, sets):
        kk, cc = s.download_sorted()
        wk, wc = d.arrays()
        assert (kk == wk).all() and (cc == wc).all()
    if len(sets) >= 2:
        x = eng.intersect(sets[0], sets[1], "sum")
        wx = CO.simple(dbs[0], dbs[1], 1, 2, 255)      # intersect, -ocsum
        kk, cc = x.download_sorted()
        wk, wc = wx.arrays()
        assert (kk == wk).all() and (cc == wc).all()


@pytest.mark.parametrize("seed", range(40))
def test_fused_path_on_random_shapes(eng, seed):
    """The same shapes through the fused form (no group sets requested: grid-mode build + tagged
    union, in batches when a shape has more than 64 genomes) — histograms, distinct counts and the
    emitted across-group set against the C restatement."""
    k, seqs, group_of = draw_case(seed)
    cs = random.Random(seed).choice([5000, 3, 255])
    want = CO.exp1(seqs, group_of, k, cs=cs, hist_len=300)
    got = eng.exp1_run(seqs, group_of, k, cs=cs, hist_len=300)
    assert (got["distinct_per_seq"] == want["distinct_per_seq"]).all()
    assert (got["within_hist"] == want["within_hist"]).all()
    assert (got["across_hist"] == want["across_hist"]).all()
    got2 = eng.exp1_run(seqs, group_of, k, cs=cs, hist_len=300, want_across_set=True)
    assert (got2["across_hist"] == want["across_hist"]).all()
    keys, counts = got2["across_set"].download_sorted()
    assert keys.shape[0] == int(want["across_hist"].sum())
    assert (np.bincount(np.minimum(counts, 299), minlength=300) == want["across_hist"]).all()
