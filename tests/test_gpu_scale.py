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


def test_cfg3_shape_k_sweep_10x10(eng):
    """BASELINE configs[2] shape (10 species x 10 genomes, k in {15,21,27,31,41}) at 20 kbp per
    genome: step_5 / step_9 histograms of the fused device path against the C restatement."""
    from khoice_amd import synth
    from oracle import c_oracle as CO
    items = synth.species_set(10, 10, 20_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    for k in (15, 21, 27, 31, 41):
        got = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=5001)
        want = CO.exp1(seqs, group_of, k, cs=5000, hist_len=5001, nthreads=8)
        assert (got["distinct_per_seq"] == want["distinct_per_seq"]).all(), k
        assert (got["within_hist"] == want["within_hist"]).all(), k
        assert (got["across_hist"] == want["across_hist"]).all(), k


def test_cfg5_saturation_5001_clone_group(eng):
    """BASELINE configs[4] corner: a group of 5001 near-identical 10 kbp genomes — the union
    counters must saturate at the -cs5000 of exp_type_1.smk:61 (fan-in far above one launch)."""
    from khoice_amd import synth
    k, L = 31, 10_000
    anc = synth.ancestor(77, L)
    base = synth.clean_text(synth.genome_records(77, 0, L, anc))
    other = synth.clean_text(synth.genome_records(77, 1, L, anc))
    seqs = [base] * 5000 + [other]
    sets = eng.build_batch(seqs, k, with_counts=False)
    union, hist = eng.union_sum(sets, 5000, hist_len=5002)
    from oracle import kmer_oracle as O
    da = O.set_counts(O.count_records(base.decode().split("\n"), k), 1)
    db = O.set_counts(O.count_records(other.decode().split("\n"), k), 1)
    both = len(set(da) & set(db))
    assert int(hist[5001]) == 0                      # nothing above the saturation value
    assert int(hist[5000]) == len(da)                # 5000 or 5001 occurrences -> 5000
    assert int(hist[1]) == len(db) - both            # only in the odd genome
    assert len(union) == len(set(da) | set(db))
    # without saturation the shared k-mers would read 5001
    union2, hist2 = eng.union_sum(sets, 100000, hist_len=5002)
    assert int(hist2[5001]) == both and int(hist2[5000]) == len(da) - both
