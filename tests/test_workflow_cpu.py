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
