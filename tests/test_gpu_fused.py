"""The fused form of kh_exp1_run (one batched build in grid mode + one tagged union; taken when no
per-group set is requested) against the C restatement and against the library's own general path
(per-genome sets -> group unions -> across-group union) on the same inputs.  Bit-exact."""
import random

import numpy as np
import pytest

from khoice_amd import synth
from oracle import c_oracle as CO
from tests.util import random_dna

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def key_array_form(monkeypatch):
    """This module tests the key-array form of the fused path (k_grid_bucket + k_union_hash /
    k_union_tagged): still what runs for k < 20, k > 32, more than 64 genomes and emitted sets.  The
    super-k-mer form that otherwise takes 20 <= k <= 32 has its own module (test_gpu_skm.py)."""
    monkeypatch.setenv("KHOICE_NO_SKM", "1")


@pytest.fixture(scope="module")
def eng():
    from khoice_amd import build as kbuild
    from khoice_amd import engine as E
    kbuild.build_library()
    e = E.Engine(0)
    yield e
    e.close()


def check(eng, seqs, group_of, k, cs=5000, hist_len=5001, expect_fused=True):
    want = CO.exp1(seqs, group_of, k, cs=cs, hist_len=hist_len)
    def fused_launches():   # either form of the fused path: key arrays (union_tagged) or super-k-mers (skm_union)
        kern = eng.stats()["kernels"]
        return kern["union_tagged"]["launches"] + kern["skm_union"]["launches"]
    before = fused_launches()
    eng.profile(True)
    got = eng.exp1_run(seqs, group_of, k, cs=cs, hist_len=hist_len)
    after = fused_launches()
    eng.profile(False)
    if expect_fused:
        assert after > before, "the fused path did not run"      # one launch per batch of groups and key-range wave
    assert (got["distinct_per_seq"] == want["distinct_per_seq"]).all()
    assert (got["within_hist"] == want["within_hist"]).all()
    assert (got["across_hist"] == want["across_hist"]).all()
    return got, want


@pytest.mark.parametrize("k", [7, 15, 21, 31, 32, 33, 41, 63, 64])
def test_fused_matches_oracle_over_k(eng, k):
    items = synth.species_set(3, 3, 60_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    check(eng, seqs, group_of, k)


def test_fused_edge_inputs(eng):
    rng = random.Random(7)
    anc = random_dna(rng, 30_000)
    seqs = [
        anc.encode(),
        (anc[:15_000].lower() + "N" * 40 + anc[15_000:]).encode(),      # lower case, an N run
        b"ACGT",                                                        # shorter than k
        b"",                                                            # empty
        ("A" * 20_000 + "\n" + anc[:5_000]).encode(),                   # low complexity: an oversize bucket
        (">x\n" + anc[::-1]).encode(),                                  # header symbols break runs
        random_dna(rng, 200).encode(),                                  # far shorter than the rest
    ]
    group_of = [0, 0, 0, 1, 1, 2, 2]
    for k in (5, 21, 31, 47):
        check(eng, seqs, group_of, k, cs=5000, hist_len=64)
    check(eng, seqs, group_of, 31, cs=2, hist_len=64)                   # saturation of both counters
    check(eng, seqs, group_of, 31, cs=5000, hist_len=3)                 # counters beyond the last bin
    check(eng, [seqs[0]], [0], 31)                                      # one group of one genome
    check(eng, [seqs[2], seqs[3]], [0, 1], 31)                          # nothing to count at all


def test_fused_many_genomes_and_groups(eng):
    """64 operands (the mask width), groups of very different sizes, shared blocks across groups."""
    rng = random.Random(11)
    shared = random_dna(rng, 3_000)
    sizes = [1, 2, 30, 7, 24]
    seqs, group_of = [], []
    for g, sz in enumerate(sizes):
        anc = random_dna(rng, 20_000)
        for j in range(sz):
            t = list(anc)
            for _ in range(len(t) // 200):
                t[rng.randrange(len(t))] = rng.choice("ACGT")
            seqs.append(("".join(t) + ("N" + shared if j % 2 == 0 else "")).encode())
            group_of.append(g)
    assert len(seqs) == 64
    order = list(range(64))
    rng.shuffle(order)                                                  # callers need not pass groups in order
    seqs = [seqs[i] for i in order]
    group_of = [group_of[i] for i in order]
    check(eng, seqs, group_of, 31)
    check(eng, seqs, group_of, 41, cs=9, hist_len=40)
    # one genome more: beyond the mask, the general path answers (same numbers)
    check(eng, seqs + [seqs[0]], group_of + [group_of[0]], 31, expect_fused=False)


def test_fused_equals_general_path_and_across_set(eng):
    items = synth.species_set(4, 3, 150_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    for k in (31, 41):
        fused = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=64, want_across_set=True)
        gen = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=64, want_sets=True)
        assert (fused["within_hist"] == gen["within_hist"]).all()
        assert (fused["across_hist"] == gen["across_hist"]).all()
        assert (fused["distinct_per_seq"] == gen["distinct_per_seq"]).all()
        fk, fc = fused["across_set"].download_sorted()
        gk, gc = gen["across_set"].download_sorted()
        assert (fk == gk).all() and (fc == gc).all()
        # the emitted set is a regular database: sorted by mixed key, usable by every operation
        plain = [g.set_counts(1) for g in gen["group_sets"]]
        x = eng.intersect(fused["across_set"], plain[0], "left", cs=5000)
        assert len(x) == len(plain[0])


def test_fused_full_size_genomes(eng):
    """2 species x 3 genomes x 5 Mbp, k = 31: the bucket grid of real-size genomes (1351 buckets,
    the staged scatter, sub-range index) against the C restatement."""
    items = synth.species_set(2, 3, 5_000_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    check(eng, seqs, group_of, 31)


def test_fused_all_ones_key(eng):
    """k = 32: the mixed key with all 64 bits set cannot live in the hash sets (it is their empty
    marker) and is carried beside them.  Find the k-mer that mixes to it and plant it."""
    from khoice_amd import engine as E
    k = 32
    raw = int(E.unmix_host(k, np.array([0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0])
    kmer = "".join("ACGT"[(raw >> (2 * (k - 1 - i))) & 3] for i in range(k))
    rng = random.Random(5)
    a = random_dna(rng, 5000) + "N" + kmer + "N" + random_dna(rng, 5000)
    b = random_dna(rng, 4000) + "N" + kmer
    c = random_dna(rng, 3000)
    check(eng, [a.encode(), b.encode(), c.encode(), (kmer + "N" + kmer).encode()], [0, 0, 1, 1], k, hist_len=16)


@pytest.mark.parametrize("k", [31, 41])
def test_fused_key_range_waves(eng, monkeypatch, k):
    """Under a memory budget smaller than the batch the fused path runs as key-range waves (every
    wave keeps one slice of the key space: BASELINE configs[4] "HBM-spill partitioning"): same
    histograms, distinct counts and across-group set for any number of waves."""
    items = synth.species_set(3, 3, 200_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    ref = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=64, want_across_set=True)
    rk, rc = ref["across_set"].download_sorted()
    for budget in ("1000000", "300000", "70000"):            # one group per batch; then 2 and 9 key-range waves per group
        monkeypatch.setenv("KHOICE_WAVE_BASES", budget)
        got, _ = check(eng, seqs, group_of, k, hist_len=64)
        with_set = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=64, want_across_set=True)
        gk, gc = with_set["across_set"].download_sorted()
        assert (gk == rk).all() and (gc == rc).all()
        assert (with_set["within_hist"] == ref["within_hist"]).all()
    monkeypatch.delenv("KHOICE_WAVE_BASES")


def test_fused_retries_with_finer_slots(eng):
    """Forty one-genome groups of the SAME sequence: the slot fill is planned for keys that come in
    single copies, every key comes in forty, so slots overflow; the fused path reads the fullest
    slot the kernel recorded and tries once more with finer slots (the general path after that)."""
    rng = random.Random(3)
    base = random_dna(rng, 150_000).encode()
    seqs, group_of = [base] * 40, list(range(40))
    r0 = eng.stats()["retries"]
    got, _ = check(eng, seqs, group_of, 31, hist_len=64)
    assert eng.stats()["retries"] > r0
    assert int(got["across_hist"][40]) == int(got["distinct_per_seq"][0])      # every k-mer is in all forty groups


def test_fused_fine_bin_form_for_one_word_keys(eng, monkeypatch):
    """One-word keys normally meet in the LDS hash set (k_union_hash); the fine-bin placement +
    leader search that two-word keys use must give the same answers for them."""
    items = synth.species_set(3, 3, 120_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    monkeypatch.setenv("KHOICE_NO_UNION_HASH", "1")
    for k in (13, 31, 32):
        check(eng, seqs, group_of, k)
