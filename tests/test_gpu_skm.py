"""The super-k-mer form of the fused path (khoice_amd/csrc/kh_skm.hip, kh_skm2.hip: minimizer records ->
two counting-sort levels -> one LDS hash set per slot; taken by kh_exp1_run for 20 <= k <= 63 when no
set is requested; two-word keys from k = 33) against the C restatement of exp_type_1.smk:156-259 and against the library's
key-array form on the same inputs.  Bit-exact: histograms and per-genome distinct counts."""
import os
import random

import numpy as np
import pytest

from khoice_amd import synth
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


def skm_launches(eng):
    return eng.stats()["kernels"]["skm_union"]["launches"]


def run_skm(eng, seqs, group_of, k, cs=5000, hist_len=5001, expect_skm=True):
    eng.profile(True)
    before = skm_launches(eng)
    got = eng.exp1_run(seqs, group_of, k, cs=cs, hist_len=hist_len)
    ran = skm_launches(eng) - before
    eng.profile(False)
    if expect_skm:
        assert ran == 1, "the super-k-mer form did not run"
    return got, ran


def same(a, b):
    assert (a["distinct_per_seq"] == b["distinct_per_seq"]).all()
    assert (a["within_hist"] == b["within_hist"]).all()
    assert (a["across_hist"] == b["across_hist"]).all()


def check(eng, seqs, group_of, k, cs=5000, hist_len=5001, expect_skm=True):
    want = CO.exp1(seqs, group_of, k, cs=cs, hist_len=hist_len)
    got, ran = run_skm(eng, seqs, group_of, k, cs, hist_len, expect_skm)
    same(got, want)
    return got, ran


@pytest.mark.parametrize("k", [15, 16, 17])
def test_skm_kernels_take_short_kmers(eng, k, monkeypatch):
    """kh_exp1_run leaves k < 17 to the key arrays (the minimizers would be too short to spread); the kernels themselves
    take k >= 15 — keys that fit the low word up to k = 16: the rolled strands are masked in both halves."""
    monkeypatch.setenv("KHOICE_SKM_MIN_K", "15")
    items = synth.species_set(3, 3, 60_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    check(eng, seqs, group_of, k)


@pytest.mark.parametrize("k", [17, 18, 19, 20, 21, 23, 24, 27, 30, 31, 32])
def test_skm_matches_oracle_over_k(eng, k):
    """Every minimizer geometry: minimizers of 12 bases (k = 17 .. 24: windows of 6 .. 13), 13 (k = 27), 15 and 16; w a
    power of two (k = 27, 30, 31) and not (two overlapping windows)."""
    items = synth.species_set(3, 3, 60_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    check(eng, seqs, group_of, k)


@pytest.mark.parametrize("k", [33, 34, 36, 40, 41, 47, 48, 49, 55, 62, 63])
def test_skm_two_word_keys_match_oracle_over_k(eng, k):
    """32-byte records, windows of 18 .. 51 m-mers (hashes of the next TWO threads), the two-step claim of the
    128-bit hash set."""
    items = synth.species_set(3, 3, 60_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    check(eng, seqs, group_of, k)


def test_skm_two_word_keys_edges_and_shapes(eng):
    rng = random.Random(17)
    anc = random_dna(rng, 30_000)
    body = random_dna(rng, 40_000)
    seqs = [
        anc.encode(),
        (anc[:15_000].lower() + "N" * 40 + anc[15_000:]).encode(),
        b"ACGT", b"",
        ("A" * 20_000 + "\n" + anc[:5_000]).encode(),                   # one minimizer for 20 000 positions: long runs cut at nmax
        ("AC" * 10_000).encode(),
        ("T" * 5_000 + "A" * 5_000).encode(),                           # ends in / starts with 32 T: the low key word near all-ones
    ]
    for cut in (8191, 8192, 8192 + 30, 16384 - 63, 16384 + 1):
        seqs.append((body[:cut] + "N" + body[cut:cut + 5_000] + "\n" + body[cut + 5_000:cut + 5_777]).encode())
    seqs.append(body[:8192 + 62].encode())
    seqs.append(body[:63].encode())
    group_of = [i % 4 for i in range(len(seqs))]
    for k in (33, 41, 63):
        check(eng, seqs, group_of, k, cs=5000, hist_len=64, expect_skm=False)
    # 64 operands, repeats inside genomes
    sizes = [1, 2, 30, 7, 24]
    seqs, group_of = [], []
    for g, sz in enumerate(sizes):
        a = random_dna(rng, 20_000)
        for j in range(sz):
            t = list(a)
            for _ in range(len(t) // 200):
                t[rng.randrange(len(t))] = rng.choice("ACGT")
            s2 = "".join(t)
            if j % 4 == 1:
                s2 += "\n" + s2[1000:3000]
            seqs.append(s2.encode())
            group_of.append(g)
    check(eng, seqs, group_of, 41)
    check(eng, seqs, group_of, 63, cs=3, hist_len=8)


def test_skm_two_word_keys_overfull_slots_and_form_equivalence(eng):
    items = synth.species_set(3, 4, 100_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    got, _ = run_skm(eng, seqs, group_of, 41)
    os.environ["KHOICE_NO_SKM"] = "1"
    try:
        ref, ran = run_skm(eng, seqs, group_of, 41, expect_skm=False)
    finally:
        del os.environ["KHOICE_NO_SKM"]
    assert ran == 0
    same(got, ref)
    os.environ["KHOICE_SKM_MEAN"] = "3000"      # > 2048 table entries: two key subsets per slot
    os.environ["KHOICE_SKM_SLACK"] = "1.1"
    try:
        check(eng, seqs, group_of, 41)
        check(eng, seqs, group_of, 63)
    finally:
        del os.environ["KHOICE_SKM_MEAN"]
        del os.environ["KHOICE_SKM_SLACK"]


def test_skm_matches_key_array_form(eng):
    items = synth.species_set(4, 4, 150_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    got, _ = run_skm(eng, seqs, group_of, 31)
    os.environ["KHOICE_NO_SKM"] = "1"
    try:
        ref, ran = run_skm(eng, seqs, group_of, 31, expect_skm=False)
    finally:
        del os.environ["KHOICE_NO_SKM"]
    assert ran == 0
    same(got, ref)


def test_skm_edge_inputs(eng):
    rng = random.Random(7)
    anc = random_dna(rng, 30_000)
    seqs = [
        anc.encode(),
        (anc[:15_000].lower() + "N" * 40 + anc[15_000:]).encode(),      # lower case, an N run
        b"ACGT",                                                        # shorter than k
        b"",                                                            # empty
        ("A" * 20_000 + "\n" + anc[:5_000]).encode(),                   # low complexity: one minimizer, long runs
        (">x\n" + anc[::-1]).encode(),                                  # header symbols break runs
        random_dna(rng, 200).encode(),                                  # far shorter than the rest
        ("ACGTN" * 4_000).encode(),                                     # no valid k-mer at all
        ("AC" * 10_000).encode(),                                       # two distinct k-mers, 20 000 instances
    ]
    group_of = [0, 0, 0, 1, 1, 2, 2, 2, 1]
    for k in (20, 25, 31, 32):
        check(eng, seqs, group_of, k, cs=5000, hist_len=64, expect_skm=False)
    check(eng, seqs, group_of, 31, cs=2, hist_len=64, expect_skm=False)   # saturation of both counters
    check(eng, seqs, group_of, 31, cs=5000, hist_len=3, expect_skm=False)  # counters beyond the last bin
    check(eng, [seqs[0]], [0], 31)                                      # one group of one genome
    check(eng, [seqs[2], seqs[3]], [0, 1], 31, expect_skm=False)        # nothing to count at all


def test_skm_record_boundaries(eng):
    """Sequence ends, N runs and record separators at every offset relative to the 32-position thread
    ranges and the 8192-position sub-tiles of the scatter kernel."""
    rng = random.Random(3)
    body = random_dna(rng, 40_000)
    seqs = []
    for cut in (8191, 8192, 8193, 8192 + 30, 8192 + 31, 16384 - 31, 16384 + 1, 24_000):
        t = body[:cut] + "N" + body[cut:cut + 5_000] + "\n" + body[cut + 5_000:cut + 5_000 + 777]
        seqs.append(t.encode())
    seqs.append(body[:8192 + 30].encode())      # ends one base short of a full last k-mer window of the sub-tile
    seqs.append(body[:8192 + 31].encode())
    seqs.append(body[:31].encode())             # exactly one k-mer
    group_of = [i % 3 for i in range(len(seqs))]
    for k in (31, 32, 24):
        check(eng, seqs, group_of, k)


def test_skm_many_genomes_and_groups(eng):
    """64 operands (the mask width), groups of very different sizes, shared blocks across groups,
    repeats inside a genome (the distinct count is instances minus repeated (k-mer, genome) pairs)."""
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
            s = "".join(t)
            if j % 3 == 0:
                s += "\n" + shared
            if j % 4 == 1:
                s += "\n" + s[1000:3000]          # a repeat inside the genome
            seqs.append(s.encode())
            group_of.append(g)
    order = list(range(len(seqs)))
    rng.shuffle(order)
    seqs = [seqs[i] for i in order]
    group_of = [group_of[i] for i in order]
    check(eng, seqs, group_of, 31)
    check(eng, seqs, group_of, 21, cs=3, hist_len=8)


def test_skm_more_than_64_genomes_without_across_step(eng):
    """Steps 1-4 only (BASELINE configs[2]: 10 x 10 genomes, within-group occurrence): batches of whole groups of
    at most 64 genomes are independent, each takes the super-k-mer form."""
    items = synth.species_set(7, 10, 30_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    for k in (31, 41):
        want = CO.exp1(seqs, group_of, k, cs=5000, hist_len=64)
        eng.profile(True)
        before = skm_launches(eng)
        got = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=64, across=False)
        ran = skm_launches(eng) - before
        eng.profile(False)
        assert ran == 2                                   # 6 groups (60 genomes) + 1 group
        assert (got["distinct_per_seq"] == want["distinct_per_seq"]).all()
        assert (got["within_hist"] == want["within_hist"]).all()
        # with the across-group step: the two batches, then one more pass whose records carry the GROUP number
        eng.profile(True)
        before = skm_launches(eng)
        full = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=64)
        ran = skm_launches(eng) - before
        eng.profile(False)
        assert ran == 3
        same(full, want)
        os.environ["KHOICE_NO_SKM_TWO_PASS"] = "1"      # and the key-array batches with emitted sets
        try:
            same(eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=64), want)
        finally:
            del os.environ["KHOICE_NO_SKM_TWO_PASS"]
    # saturation and a short histogram through the two-pass form
    want = CO.exp1(seqs, group_of, 31, cs=2, hist_len=4)
    same(eng.exp1_run(seqs, group_of, 31, cs=2, hist_len=4), want)


def test_skm_overfull_slot_rounds(eng):
    """Slots far above the hash set's capacity (KHOICE_SKM_MEAN forces few, large slots): key subsets
    are handled in rounds, several index passes per round."""
    items = synth.species_set(2, 3, 80_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    os.environ["KHOICE_SKM_MEAN"] = "6000"      # > 4096 table entries: two key subsets, two index passes each
    os.environ["KHOICE_SKM_SLACK"] = "1.1"      # (keeps the slot regions under the 2048-record limit)
    try:
        check(eng, seqs, group_of, 31)
        check(eng, seqs, group_of, 22)
    finally:
        del os.environ["KHOICE_SKM_MEAN"]
        del os.environ["KHOICE_SKM_SLACK"]


def test_skm_region_overflow_falls_back(eng):
    """A record region that is too small (KHOICE_SKM_SLACK below what the input needs) raises the
    capacity bit; the key-array form takes over and the answer is unchanged."""
    rng = random.Random(5)
    seqs = [("ACGTTGCA" * 3 + "N").encode() * 4_000, random_dna(rng, 50_000).encode()]   # one record per k-mer run of 1
    group_of = [0, 1]
    want = CO.exp1(seqs, group_of, 24, cs=5000, hist_len=16)
    retries = eng.stats()["retries"]
    os.environ["KHOICE_SKM_SLACK"] = "0.01"
    try:
        got, ran = run_skm(eng, seqs, group_of, 24, hist_len=16, expect_skm=False)
    finally:
        del os.environ["KHOICE_SKM_SLACK"]
    assert ran == 1 and eng.stats()["retries"] == retries + 1      # tried, overflowed, fell back
    same(got, want)


def test_skm_random_shapes(eng):
    rng = random.Random(2024)
    for it in range(12):
        ngroups = rng.randint(1, 5)
        seqs, group_of = [], []
        for g in range(ngroups):
            anc = random_dna(rng, rng.randint(500, 40_000))
            for j in range(rng.randint(1, 6)):
                t = list(anc)
                for _ in range(rng.randint(0, len(t) // 50)):
                    t[rng.randrange(len(t))] = rng.choice("ACGTN")
                seqs.append("".join(t).encode())
                group_of.append(g)
        k = rng.randint(17, 32)
        check(eng, seqs, group_of, k, cs=rng.choice([1, 2, 5000]), hist_len=rng.choice([2, 8, 5001]), expect_skm=False)
