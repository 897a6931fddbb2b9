"""Host logic of the experiment-type-1 runner on CPU: the DAG order, paths, complex-ops files
and CSV stage, driven through oracle-backed stand-in executables (tests/fakebin).  The GPU
suite runs the same DAG through the real bin/kmc and bin/kmc_tools."""
import os

from khoice_amd import synth
from khoice_amd.workflow import exp_type_1 as W
from oracle import kmer_oracle as O

FAKE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fakebin")


def test_ops_files_match_reference_text(golden, tmp_path, monkeypatch):
    g = golden("complex_ops.json")
    for num, names in g["listing"].items():
        d = tmp_path / "data" / f"dataset_{num}"
        d.mkdir(parents=True)
        for n in names:
            (d / n).write_bytes(b"")
    real_listdir = os.listdir

    def fixed(path):
        base = os.path.basename(os.path.normpath(path))
        if base.startswith("dataset_") and base.split("_")[1] in g["listing"]:
            return list(g["listing"][base.split("_")[1]])
        return real_listdir(path)
    monkeypatch.setattr(os, "listdir", fixed)
    W.prepare(str(tmp_path), g["k_values"], g["num_datasets"])
    for rel, text in g["files"].items():
        assert (tmp_path / rel).read_text() == text, rel
    assert (tmp_path / "tmp").is_dir()


def test_cfg1_dag_through_standins(tmp_path):
    # BASELINE configs[0] shape, shrunk: 2 species x 1 genome, k = 21 (single-input complex)
    root = str(tmp_path)
    synth.write_dataset_tree(root, 2, 1, 6000)
    out = W.run(root, [21], 2, bin_dir=FAKE)
    assert out["processes"] == 2 * 2 + 3 * 2 + 2
    for rel in ("step_5/within_datasets_analysis.csv", "final_results_type1/across_datasets_analysis.csv",
                "step_1/k_21/dataset_1/sp1_g0.kmc_pre", "step_7/k_21/all_datasets.transformed.combined.transformed.combined.kmc_suf"):
        assert os.path.exists(os.path.join(root, rel)), rel
    rows = out["within"].splitlines()
    assert rows[0].startswith("group_num,k,percent_1_occ")
    assert rows[1].startswith("group_1,21,1.0,0.0,0.0,0.0,1.0,1.0,")     # one genome: all k-mers occur once
    # histogram file of a group equals the oracle's for that genome
    fa = O.read_fasta_bytes(os.path.join(root, "data/dataset_2/sp2_g0.fna.gz"))
    want = O.histogram_text(O.set_counts(O.build(fa, 21), 1), 65535)
    assert open(os.path.join(root, "step_4/k_21/dataset_2/dataset_2_k21_hist.txt")).read() == want


def test_failed_rule_raises_and_leaves_no_output(tmp_path):
    root = str(tmp_path)
    synth.write_dataset_tree(root, 1, 1, 2000)
    os.remove(os.path.join(root, "data/dataset_1/sp1_g0.fna.gz"))
    open(os.path.join(root, "data/dataset_1/sp1_g0.fna.gz"), "wb").write(b"\x1f\x8b broken")
    try:
        W.run(root, [21], 1, bin_dir=FAKE)
    except RuntimeError as e:
        assert "rule failed" in str(e)
    else:
        raise AssertionError("expected the build rule to fail")
    assert not os.path.exists(os.path.join(root, "step_1/k_21/dataset_1/sp1_g0.kmc_pre"))


def expected_type4_outputs(root, k, n):
    """accuracies_type_4 texts from the FASTA files by the oracle alone (no rule plumbing)."""
    from khoice_amd import merge_lists as ML
    from khoice_amd.workflow import exp_type_4 as W4
    from oracle import merge_oracle as MO
    unions, pivots = [], []
    for num in range(1, n + 1):
        sets = [O.set_counts(O.build(O.read_fasta_bytes(
            os.path.join(root, f"input_type4/rest_of_set/dataset_{num}/{g}.fna.gz")), k), 1)
            for g in W4.rest_of_set(root, num)]
        unions.append(O.set_counts(O.union_sum(sets, 5000), 1))
        pivots.append(O.build(O.read_fasta_bytes(os.path.join(root, f"input_type4/pivot/pivot_{num}.fna.gz")), k))
    rows, uniques = zip(*[MO.confusion_row(p, unions) for p in pivots])
    cm, cm_ucol = ML.assemble_matrices(rows, uniques, n)
    return ML.format_outputs(cm, cm_ucol, n, str(k))


def test_exp_type_4_dag_through_standins(tmp_path):
    """exp_type_4.smk rule by rule (build, set, union, D x D intersect, dumps, merge_lists)
    with oracle-backed stand-ins; the result equals the oracle's direct answer."""
    from khoice_amd.workflow import exp_type_4 as W4
    root = str(tmp_path)
    synth.write_type4_tree(root, 3, 2, 3000)
    out = W4.run(root, [9, 21], 3, bin_dir=FAKE, merge_cmd=os.path.join(FAKE, "merge_lists"))
    per_k = 3 * (2 * 2 + 1 + 3) + 3 * (1 + 3 * 3) + 1
    assert out["processes"] == 2 * per_k
    cat = ""
    for k in (21, 9):          # `cat values/*.csv`: k_21_... sorts before k_9_...
        want = expected_type4_outputs(root, k, 3)
        for rel, text in want.items():
            assert open(os.path.join(root, "accuracies_type_4", rel)).read() == text, rel
        cat += want[f"values/k_{k}_accuracy_values.csv"]
    assert open(out["accuracy_values"]).read() == cat
    # intermediates the rules delete are gone, the file lists name absolute dump paths
    assert not os.path.exists(os.path.join(root, "step_1_type_4/rest_of_set/k_21/dataset_1/sp1_g0.kmc_pre"))
    assert not os.path.exists(os.path.join(root, "text_dump_type_4/k_21/pivot/pivot_1.txt"))
    first = open(os.path.join(root, "filelists_type_4/k_21/intersections_filelist.txt")).readline().strip()
    assert first == os.path.abspath(root) + "/text_dump_type_4/k_21/intersection/pivot_1/pivot_1_intersect_dataset_1.txt"


def expected_type2_outputs(root, k_values, n, exp_dir):
    """Histogram texts and the two CSVs of experiment type 2 from the FASTA files by the oracle
    (sets as dicts) and the golden-pinned summariser; nothing from the rule plumbing."""
    from khoice_amd import summarize as S
    from khoice_amd.workflow import exp_type_2 as W2
    files = {}
    for k in k_values:
        unions, pivots = [], []
        for num in range(1, n + 1):
            sets = [O.set_counts(O.build(O.read_fasta_bytes(
                os.path.join(root, f"input_type_2/rest_of_set/dataset_{num}/{g}.fna.gz")), k), 1)
                for g in W2.rest_of_set(root, num)]
            unions.append(O.union_sum(sets, 5000))
            pivots.append(O.set_counts(O.build(O.read_fasta_bytes(
                os.path.join(root, f"input_type_2/pivot/dataset_{num}/pivot_{num}.fna.gz")), k), 1))
        group_sets = [O.set_counts(u, 1) for u in unions]
        for num in range(n):
            across = O.union_sum([group_sets[i] for i in range(n) if i != num], 5000)
            for scope, other in (("within", unions[num]), ("across", across)):
                # `simple` without -cs: the output takes the larger counter range of its operands, here
                # the -cs5000 unions (two counter bytes: 65535 histogram lines); parity unpinned
                for op, db in (("intersect", O.intersect(pivots[num], other, "sum", 5000)),
                               ("subtract", O.kmers_subtract(pivots[num], other))):
                    rel = f"{scope}_dataset_results_type_2/k_{k}/dataset_{num + 1}/{op}/dataset_{num + 1}_pivot_{op}_group.hist.txt"
                    files[rel] = O.histogram_text(db, 65535)
    for rel, text in files.items():
        os.makedirs(os.path.dirname(os.path.join(exp_dir, rel)), exist_ok=True)
        open(os.path.join(exp_dir, rel), "w").write(text)
    ks = [str(k) for k in k_values]
    cwd = os.getcwd()
    os.chdir(exp_dir)
    try:
        within = S.pivot_within_groups_csv(W2._hist_paths("within", ks, n), n, lambda d: len(W2.rest_of_set(root, int(d))))
        across = S.pivot_across_groups_csv(W2._hist_paths("across", ks, n), n)
    finally:
        os.chdir(cwd)
    return files, within, across


def test_exp_type_2_dag_through_standins(tmp_path):
    """exp_type_2.smk rule by rule with oracle-backed stand-ins == the oracle's direct answer."""
    from khoice_amd.workflow import exp_type_2 as W2
    root = str(tmp_path / "work")
    os.makedirs(root)
    synth.write_type2_tree(root, 3, 2, 3000)
    out = W2.run(root, [9, 21], 3, bin_dir=FAKE)
    assert out["processes"] == 2 * (3 * (2 * 2 + 2 + 1 + 4 + 1) + 3 * (1 + 4))
    files, within, across = expected_type2_outputs(root, [9, 21], 3, str(tmp_path / "expected"))
    for rel, text in files.items():
        assert open(os.path.join(root, rel)).read() == text, rel
    assert out["within"] == within and out["across"] == across
    assert open(os.path.join(root, "within_dataset_analysis_type_2/within_dataset_analysis.csv")).read() == within
    assert open(os.path.join(root, "across_dataset_analysis_type_2/across_dataset_analysis.csv")).read() == across
    ops = open(os.path.join(root, "complex_ops_type_2/across_groups/k_9/pivot_2/across_datasets_pivot_2.txt")).read()
    assert "dataset_2.transformed" not in ops and "set2 = within_databases_type_2/rest_of_set/k_9/dataset_3/" in ops
