"""Pin the Python oracle: canonical form / windowing against vectors captured from the
reference's src/merge_lists.py:53-73; counting semantics against hand-worked cases
(KMC itself is unavailable: those are 'parity unpinned', see oracle/kmer_oracle.py)."""
from oracle import kmer_oracle as O
from tests.util import check_multiset_case, db_to_arrays


def test_canonical_matches_reference(golden):
    g = golden("canonical_kmers.json")
    assert len(g["canonical"]) > 100
    for kmer, want in g["canonical"]:
        assert O.canonical_str(kmer) == want
        # integer form: canonical = min(code, code of revcomp)
        k = len(kmer)
        assert O.decode(min(O.encode(kmer), O.encode(O.reverse_complement(kmer))), k) == want


def test_windows_match_reference(golden):
    for read, k, want in golden("canonical_kmers.json")["windows"]:
        assert O.windows(read, k) == want


def test_build_hand_cases():
    fa = b">r1\nACGTN\nACGT\n>r2\nacgt\n"
    # record r1 = ACGTNACGT (lines concatenated), N splits; r2 = acgt (lower == upper)
    db = O.build(fa, 4)
    assert db == {O.encode("ACGT"): 3}
    db = O.build(fa, 3)
    # ACG/CGT are reverse complements of each other -> canonical ACG, 2 per ACGT run
    assert db == {O.encode("ACG"): 6}
    assert O.build(fa, 5) == {}
    assert O.build(b">x\nAAAA\n>y\nTTTT\n", 2) == {O.encode("AA"): 6}
    # saturation at cs, ci filter
    many = b">x\n" + b"A" * 400 + b"\n"
    assert O.build(many, 3) == {0: 255}
    assert O.build(many, 3, cs=5000) == {0: 398}
    assert O.build(b">x\nAAAC\n", 3, ci=2) == {}
    # k-mers never span records
    assert O.build(b">a\nAC\n>b\nGT\n", 3) == {}


def test_set_ops_hand_cases():
    a = {1: 2, 5: 250, 9: 1}
    b = {5: 10, 7: 3}
    assert O.set_counts(a, 1) == {1: 1, 5: 1, 9: 1}
    assert O.union_sum([a, b], 5000) == {1: 2, 5: 260, 9: 1, 7: 3}
    assert O.union_sum([a, b], 255) == {1: 2, 5: 255, 9: 1, 7: 3}
    assert O.union_sum([a], 5000) == a
    assert O.intersect(a, b, "sum") == {5: 255}
    assert O.intersect(a, b, "min") == {5: 10}
    assert O.kmers_subtract(a, b) == {1: 2, 9: 1}
    assert O.histogram({1: 1, 2: 1, 3: 4}, 5) == [0, 2, 0, 0, 1, 0]
    assert O.histogram_text({1: 1, 2: 3}, 3) == "1\t1\n2\t0\n3\t1\n"
    assert O.dump_sorted_text({O.encode("TTA"): 2, O.encode("ACG"): 1}, 3) == "ACG\t1\nTTA\t2\n"


def test_parse_complex_ops_reference_grammar():
    # exactly what exp_type_1.smk:52-61 writes, incl. the trailing space before ')'
    txt = ("INPUT:\nset1 = step_2/k_21/dataset_1/a.transformed\n"
           "set2 = step_2/k_21/dataset_1/b.transformed\nOUTPUT:\n"
           "step_3/k_21/dataset_1/dataset_1.transformed.combined = (set1 + set2 )\n"
           "OUTPUT_PARAMS:\n-cs5000\n")
    inputs, out, names, cs = O.parse_complex_ops(txt)
    assert names == ["set1", "set2"] and cs == 5000
    assert out == "step_3/k_21/dataset_1/dataset_1.transformed.combined"
    assert inputs["set2"].endswith("b.transformed")
    one = "INPUT:\nset1 = p\nOUTPUT:\nq = (set1 )\nOUTPUT_PARAMS:\n-cs5000\n"
    assert O.parse_complex_ops(one)[2] == ["set1"]


def test_kmer_multisets_match_reference(golden):
    """K1's counting on ACGT input, pinned by reference-run output: for every case of kmer_multiset.json the
    oracle's database of the string equals collections.Counter(get_canonical_kmer(w) for w in
    process_read_into_kmers(s, k)) (src/merge_lists.py:53-73), counters unsaturated."""
    cases = golden("kmer_multiset.json")["cases"]
    assert len(cases) >= 80
    for case in cases:
        db = O.count_records([case["seq"]], case["k"], cs=1 << 30)
        keys, counts = db_to_arrays(db, case["k"])
        check_multiset_case(case, keys, counts, 1 << 30)
