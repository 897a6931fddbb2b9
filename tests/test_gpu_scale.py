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


def check_reference_invariants(eng, seqs, group_of, got, k):
    """The reference's own inline checks, as properties of a full-size run:
    exp_type_6.smk:222-236  the largest non-empty bin of a union histogram is <= the number of inputs;
    exp_type_3.smk:298      the histogram of a plain set (set_counts 1) lies entirely in bin 1;
    exp_type_3.smk:314 / exp_type_2.smk:183-184  an `intersect -ocsum` of two plain sets has bin 1 empty;
    exp_type_2.smk:202-203  a `kmers_subtract` result of a plain set lies entirely in bin 1;
    exp_type_1.smk:141-142  the four percentage metrics of the summariser add up to 1 +- 0.05."""
    from khoice_amd import summarize as S
    ngroups = max(group_of) + 1
    for g in range(ngroups):
        members = sum(1 for x in group_of if x == g)
        nz = np.nonzero(got["within_hist"][g])[0]
        assert nz.size and nz.max() <= members and nz.min() >= 1
        row = S.summarize_histogram_type1([int(x) for x in got["within_hist"][g][1:256]], members, False, k)
        assert abs(sum(row[:4]) - 1.0) <= 0.05
    nz = np.nonzero(got["across_hist"])[0]
    assert nz.size and nz.max() <= ngroups and nz.min() >= 1
    # pivot = the first genome, group = the rest of its species (the experiment-type-2/3 shape)
    first = [i for i, g in enumerate(group_of) if g == 0]
    sets = eng.build_batch([seqs[i] for i in first], k, with_counts=False)
    pivot, rest = sets[0], eng.union_sum(sets[1:], 5000).set_counts(1)
    assert int(pivot.histogram(256)[1]) == len(pivot) and int(pivot.histogram(256)[2:].sum()) == 0
    inter = eng.intersect(pivot, rest, "sum")
    hi = inter.histogram(256)
    assert int(hi[1]) == 0 and int(hi[2]) == len(inter)
    sub = eng.kmers_subtract(pivot, rest)
    hs = sub.histogram(256)
    assert int(hs[1]) == len(sub) and int(hs[2:].sum()) == 0
    assert len(inter) + len(sub) == len(pivot)
    return pivot


def test_cfg2_full_size_5x5x5mbp(eng):
    """BASELINE configs[1] exactly: 5 species x 5 genomes x 5 Mbp, k = 31 — per-genome distinct
    counts, all five step_4 histograms and the step_8 histogram against the C restatement."""
    from khoice_amd import synth
    from oracle import c_oracle as CO
    items = synth.species_set(5, 5, 5_000_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    eng.profile(True)
    before = eng.stats()["kernels"]["skm_union"]["launches"]
    got = eng.exp1_run(seqs, group_of, 31, cs=5000, hist_len=5001)
    assert eng.stats()["kernels"]["skm_union"]["launches"] - before == 1      # the super-k-mer form ran (no fall-back)
    eng.profile(False)
    want = CO.exp1(seqs, group_of, 31, cs=5000, hist_len=5001)
    assert (got["distinct_per_seq"] == want["distinct_per_seq"]).all()
    assert (got["within_hist"] == want["within_hist"]).all()
    assert (got["across_hist"] == want["across_hist"]).all()
    pivot = check_reference_invariants(eng, seqs, group_of, got, 31)
    # src/merge_lists.py:166-167 looks every canonical k-mer of a pivot read up in the pivot's dump
    # (a KeyError if one is missing): every window of a read cut from the pivot genome must be there
    from oracle import kmer_oracle as O
    keys, _ = pivot.download_sorted()
    have = set(int(x) for x in keys[:, 0])
    text = seqs[0].decode()
    rng = np.random.default_rng(7)
    checked = 0
    for _ in range(200):
        at = int(rng.integers(0, len(text) - 150))
        read = text[at:at + 150]
        if any(ch not in "ACGT" for ch in read):
            continue
        for w in O.windows(read, 31):
            assert O.encode(O.canonical_str(w)) in have
            checked += 1
    assert checked > 10_000


@pytest.mark.parametrize("k", [21, 41, 63])
def test_cfg2_full_size_other_k_super_kmer_form(eng, k):
    """The same 5 x 5 x 5 Mbp set at a narrow window (k = 21: seven m-mers per k-mer) and with two-word keys
    (k = 41, 63: 32-byte records, the two-step claim of the 128-bit hash set) against the C restatement."""
    from khoice_amd import synth
    from oracle import c_oracle as CO
    items = synth.species_set(5, 5, 5_000_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    eng.profile(True)
    before = eng.stats()["kernels"]["skm_union"]["launches"]
    got = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=5001)
    assert eng.stats()["kernels"]["skm_union"]["launches"] - before == 1
    eng.profile(False)
    want = CO.exp1(seqs, group_of, k, cs=5000, hist_len=5001)
    assert (got["distinct_per_seq"] == want["distinct_per_seq"]).all()
    assert (got["within_hist"] == want["within_hist"]).all()
    assert (got["across_hist"] == want["across_hist"]).all()
    eng.trim()


@pytest.mark.parametrize("k", [31, 41])
def test_cfg3_full_size_10x10x5mbp(eng, k):
    """BASELINE configs[2] at its stated size: 10 species x 10 genomes x 5 Mbp (100 genomes: the
    fused path runs it as batches of whole groups plus one pass whose records carry the group number) — step_4 and step_8 histograms and the
    per-genome distinct counts against the C restatement, plus the reference's invariants."""
    from khoice_amd import synth
    from oracle import c_oracle as CO
    items = synth.species_set(10, 10, 5_000_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    eng.profile(True)
    before = eng.stats()["kernels"]["skm_union"]["launches"]
    got = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=5001)
    assert eng.stats()["kernels"]["skm_union"]["launches"] - before == 3         # two batches of groups + the pass by group
    eng.profile(False)
    want = CO.exp1(seqs, group_of, k, cs=5000, hist_len=5001)
    assert (got["distinct_per_seq"] == want["distinct_per_seq"]).all()
    assert (got["within_hist"] == want["within_hist"]).all()
    assert (got["across_hist"] == want["across_hist"]).all()
    check_reference_invariants(eng, seqs, group_of, got, k)
    eng.trim()


def big_group_set(sizes, length):
    """One species per group, `sizes[g]` genomes in group g (the synthetic generator has no limit on genomes)."""
    from khoice_amd import synth
    seqs, group_of = [], []
    for g, n in enumerate(sizes):
        anc = synth.ancestor(g + 1, length)
        for j in range(n):
            seqs.append(synth.clean_text(synth.genome_records(g + 1, j, length, anc)))
            group_of.append(g)
    return seqs, group_of


@pytest.mark.parametrize("k", [31, 41])
def test_groups_of_more_than_64_genomes_stay_fused(eng, k):
    """BASELINE configs[4] shape ("all available genomes": exp_type_1.smk:36-61 lists whatever data/dataset_N holds):
    groups of 10, 70, 200 and 300 genomes.  Every histogram and distinct count equals the C restatement's, and the
    work was done by the fused forms (k <= 32: packed sub-batches of 32 genomes as phases of one phased union per
    group; above: tagged unions over sub-batches of 64 genomes; + the super-k-mer passes), not by per-genome
    databases."""
    from oracle import c_oracle as CO
    seqs, group_of = big_group_set([10, 70, 200, 300], 50_000)
    order = np.random.default_rng(5).permutation(len(seqs))          # groups interleaved, as a caller may pass them
    seqs = [seqs[i] for i in order]
    group_of = [group_of[i] for i in order]
    eng.profile(True)
    eng.stats_reset()
    got = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=5001)
    st = eng.stats()
    eng.profile(False)
    want = CO.exp1(seqs, group_of, k, cs=5000, hist_len=5001, nthreads=8)
    assert (got["distinct_per_seq"] == want["distinct_per_seq"]).all()
    assert (got["within_hist"] == want["within_hist"]).all()
    assert (got["across_hist"] == want["across_hist"]).all()
    assert int(got["within_hist"][3, 150:].sum()) > 0 and int(got["within_hist"][2, 100:].sum()) > 0   # k-mers most of a big group shares
    kern = st["kernels"]
    if k <= 32:   # one-word keys: sub-batches of 32 genomes are the phases of ONE phased union per big group
        assert kern["skm_pack"]["launches"] == 3 + 7 + 10 and kern["skm_phased"]["launches"] == 3, kern
        assert kern["union_tagged"]["launches"] == 0 and st["retries"] == 0, st
    else:         # two-word keys: tagged unions over sub-batches of 64 genomes + one counter-summing union per group
        assert kern["union_tagged"]["launches"] >= 2 + 4 + 5, kern
    assert kern["skm_union"]["launches"] >= 2                   # the 10-genome group, and the pass by group
    assert st["builds"] == len(seqs)                            # nothing was built twice (no fall-back)
    # a small saturation value goes through the same sums
    got3 = eng.exp1_run(seqs, group_of, k, cs=3, hist_len=8)
    want3 = CO.exp1(seqs, group_of, k, cs=3, hist_len=8, nthreads=8)
    assert (got3["within_hist"] == want3["within_hist"]).all() and (got3["across_hist"] == want3["across_hist"]).all()


@pytest.mark.parametrize("k", [19, 31])
def test_big_group_edge_genomes(eng, k):
    """A group of 100 genomes (four phases of the phased union) with the genomes an ingest can hand over: empty ones,
    one shorter than k, exact copies of another genome (whole records repeat under different tags), a genome that
    holds its own sequence twice (every k-mer a repeat under ONE tag: counted at the merge), homopolymer and
    dinucleotide runs (repeats found at the insertion), runs of N; next to it a group of 66, two small ones and a group
    of 70 genomes full of copies of one insertion sequence (overfull slots: the side list joins the phases).
    Histograms and distinct counts equal the C restatement's and the phased form did the work."""
    from oracle import c_oracle as CO
    from khoice_amd import synth
    rng = np.random.default_rng(11)
    anc = synth.ancestor(3, 30_000)
    base = [synth.clean_text(synth.genome_records(3, j, 30_000, anc)) for j in range(100)]
    sep = b"\n"          # cleaned text: records joined by a newline
    g0 = list(base)
    g0[3] = b""
    g0[40] = b""
    g0[7] = base[7][: k - 1]
    g0[10] = base[9]                                  # an exact copy of its neighbour
    g0[50] = base[50] + sep + base[50]                # itself twice
    g0[60] = base[60] + sep + b"A" * 3000 + sep + b"AC" * 2000
    g0[61] = base[61][:5000] + b"N" * 100 + base[61][5000:]
    g0[99] = b"T" * 500
    anc2 = synth.ancestor(4, 20_000)
    g1 = [synth.clean_text(synth.genome_records(4, j, 20_000, anc2)) for j in range(66)]
    g2 = [synth.clean_text(synth.genome_records(5, j, 10_000, synth.ancestor(5, 10_000))) for j in range(3)]
    g3 = [base[0], g1[0]]                             # a small group sharing with both big ones
    seqs = g0 + g1 + g2 + g3
    group_of = [0] * len(g0) + [1] * len(g1) + [2] * len(g2) + [3] * len(g3)
    if k == 31:
        # a group whose genomes hold 20 copies of an insertion sequence and a tandem repeat: a slot gets 24 x 20 copies
        # of the same records, above its region -> the side list joins the sub-batch's phase as extra pieces (with
        # windows of 7 at k = 19 the same records overflow the COARSE regions of so small an input: the call would
        # fall back as a whole)
        is_seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=600).tobytes())
        anc4 = synth.ancestor(6, 40_000)
        g4 = []
        for j in range(70):
            t = synth.clean_text(synth.genome_records(6, j, 40_000, anc4))
            g4.append(t + sep + sep.join([is_seq] * 20) + sep + b"ACGTTGCA" * 25)
        seqs += g4
        group_of += [4] * len(g4)
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    group_of = [group_of[i] for i in order]
    eng.profile(True)
    eng.stats_reset()
    got = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=300)
    st = eng.stats()
    eng.profile(False)
    want = CO.exp1(seqs, group_of, k, cs=5000, hist_len=300, nthreads=8)
    assert (got["distinct_per_seq"] == want["distinct_per_seq"]).all()
    assert (got["within_hist"] == want["within_hist"]).all()
    assert (got["across_hist"] == want["across_hist"]).all()
    kern = st["kernels"]
    big = 3 if k == 31 else 2
    assert kern["skm_phased"]["launches"] == big and kern["skm_pack"]["launches"] == 4 + 3 * (big - 1), kern
    assert kern["union_tagged"]["launches"] == 0 and st["retries"] == 0, st
    if k == 31:
        assert st["big_slots"] > 0   # overfull slots: through the side list (in the phases, and in the pass by group)
    # the same with a saturation value below the group sizes
    got2 = eng.exp1_run(seqs, group_of, k, cs=7, hist_len=9)
    want2 = CO.exp1(seqs, group_of, k, cs=7, hist_len=9, nthreads=8)
    assert (got2["within_hist"] == want2["within_hist"]).all() and (got2["across_hist"] == want2["across_hist"]).all()


def test_configs3_twenty_groups_on_one_gpu(eng):
    """BASELINE configs[3]'s workload (20 species x 5 genomes x 5 Mbp, k = 31) through kh_exp1_run on ONE GPU: 100 genomes
    = batches of whole groups + the pass by group.  What stays unmeasured for that config is the 8-way sharding."""
    from khoice_amd import synth
    from oracle import c_oracle as CO
    items = synth.species_set(20, 5, 5_000_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    eng.profile(True)
    eng.stats_reset()
    got = eng.exp1_run(seqs, group_of, 31, cs=5000, hist_len=5001)
    st = eng.stats()
    eng.profile(False)
    want = CO.exp1(seqs, group_of, 31, cs=5000, hist_len=5001, nthreads=0)
    assert (got["distinct_per_seq"] == want["distinct_per_seq"]).all()
    assert (got["within_hist"] == want["within_hist"]).all()
    assert (got["across_hist"] == want["across_hist"]).all()
    assert st["kernels"]["skm_union"]["launches"] == 3 and st["retries"] == 0     # two batches of ten groups + the pass by group


@pytest.mark.parametrize("k", [21, 31, 41])
def test_hard_genomes_at_full_size(eng, k):
    """Genomes with what real bacterial chromosomes have (khoice_amd.synth.hard_species_set: GC 70 %, 50 exact copies of an
    insertion sequence, seven rRNA-like operons, tandem repeats, homopolymer runs) at 5 x 5 x 5 Mbp against the C
    restatement.  The super-k-mer form keeps the run (one- and two-word keys): the few slots a hot minimizer overfills
    are taken by a kernel of their own (stats: big_slots), nothing falls back to the key arrays."""
    from khoice_amd import synth
    from oracle import c_oracle as CO
    items = synth.hard_species_set(5, 5, 5_000_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    eng.profile(True)
    eng.stats_reset()
    got = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=5001)
    st = eng.stats()
    eng.profile(False)
    want = CO.exp1(seqs, group_of, k, cs=5000, hist_len=5001, nthreads=0)
    assert (got["distinct_per_seq"] == want["distinct_per_seq"]).all()
    assert (got["within_hist"] == want["within_hist"]).all()
    assert (got["across_hist"] == want["across_hist"]).all()
    assert st["retries"] == 0 and st["big_slots"] > 0, st
    assert st["kernels"]["skm_union"]["launches"] == 1 and st["kernels"]["skm_big"]["launches"] == 1, st   # the union, then the overfull slots
    assert st["kernels"]["union_tagged"]["launches"] == 0
