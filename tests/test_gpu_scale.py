"""Full-size parity (BASELINE.json sizes) against the C restatement: a 5 Mbp genome group at
k = 31 and k = 41, every key and counter compared, plus the fused experiment histograms."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from khoice_amd import build as kbuild
    from khoice_amd import engine as E
    kbuild.build_library()
    e = E.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def group():
    from khoice_amd import synth
    items = synth.species_set(2, 3, 5_000_000)
    return [t for _, _, t in items], [s - 1 for s, _, _ in items]


@pytest.mark.parametrize("k", [31, 41])
def test_genome_sets_bit_exact_at_5mbp(eng, group, k):
    from oracle import c_oracle as CO
    seqs, _ = group
    sets = eng.build_batch(seqs[:2], k, cs=255)
    for t, s in zip(seqs[:2], sets):
        keys, counts = s.download_sorted()
        okeys, ocounts = CO.count(t, k, cs=255).arrays()
        assert keys.shape == okeys.shape
        assert (keys == okeys).all() and (counts == ocounts).all()


@pytest.mark.parametrize("k", [15, 31, 41])
def test_exp1_histograms_at_5mbp(eng, group, k):
    from oracle import c_oracle as CO
    seqs, group_of = group
    got = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=5001, want_sets=True)
    want = CO.exp1(seqs, group_of, k, cs=5000, hist_len=5001, nthreads=6)
    assert (got["distinct_per_seq"] == want["distinct_per_seq"]).all()
    assert (got["within_hist"] == want["within_hist"]).all()
    assert (got["across_hist"] == want["across_hist"]).all()
    # union keys too, for one group
    dbs = [CO.count(t, k).set_counts(1) for t, g in zip(seqs, group_of) if g == 0]
    u = CO.union_sum(dbs, 5000)
    keys, counts = got["group_sets"][0].download_sorted()
    okeys, ocounts = u.arrays()
    assert (keys == okeys).all() and (counts == ocounts).all()
