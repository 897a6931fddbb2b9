"""The drop-in boundary on a real MI355X: bin/kmc and bin/kmc_tools driven with exactly the
argv khoice's Snakemake rules use (exp_type_1.smk:156-259, exp_type_2.smk:354-393,
exp_type_4.smk:247-271), compared with the oracle; BASELINE configs[0] end to end."""
import os
import subprocess

import pytest

from khoice_amd import synth
from khoice_amd.workflow import exp_type_1 as W
from oracle import kmer_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin")


@pytest.fixture(scope="module", autouse=True)
def built():
    from khoice_amd import build as kbuild
    kbuild.build_library()
    kbuild.build_clis()


def sh(cmd, cwd, ok=True):
    env = dict(os.environ, PATH=BIN + os.pathsep + os.environ["PATH"])
    r = subprocess.run(["bash", "-c", "set -euo pipefail; " + cmd], cwd=cwd, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if ok:
        assert r.returncode == 0, r.stderr
    return r


def test_cfg1_two_species_one_genome_k21(tmp_path):
    """BASELINE configs[0]: 2 species x 1 genome (1 Mbp), k = 21, full step_1..step_9 DAG."""
    root = str(tmp_path)
    synth.write_dataset_tree(root, 2, 1, 1_000_000)
    out = W.run(root, [21], 2)            # real bin/kmc, bin/kmc_tools
    from oracle import c_oracle as CO
    dbs = []
    for s in (1, 2):
        fa = O.read_fasta_bytes(os.path.join(root, f"data/dataset_{s}/sp{s}_g0.fna.gz"))
        text = "\n".join(O.fasta_records(fa)).encode()
        dbs.append(CO.count(text, 21).set_counts(1))
    for s, d in zip((1, 2), dbs):
        h = CO.union_sum([d], 5000).histogram(65536)
        want = "".join(f"{c}\t{int(h[c])}\n" for c in range(1, 65536))
        assert open(os.path.join(root, f"step_4/k_21/dataset_{s}/dataset_{s}_k21_hist.txt")).read() == want
    h = CO.union_sum(dbs, 5000).histogram(65536)
    want = "".join(f"{c}\t{int(h[c])}\n" for c in range(1, 65536))
    assert open(os.path.join(root, "step_8/k_21/all_datasets_k21_hist.txt")).read() == want
    assert open(os.path.join(root, "final_results_type1/within_datasets_analysis.csv")).read() == out["within"]
    assert out["within"].splitlines()[1].startswith("group_1,21,1.0,0.0,0.0,0.0,1.0,1.0,")
    assert out["across"].splitlines()[1].startswith("full_group,21,")
    # the batched runner writes byte-identical results without launching a process
    root2 = str(tmp_path / "batched")
    os.makedirs(root2)
    synth.write_dataset_tree(root2, 2, 1, 1_000_000)
    out2 = W.run_batched(root2, [21], 2)
    assert out2["within"] == out["within"] and out2["across"] == out["across"]
    for rel in ("step_4/k_21/dataset_2/dataset_2_k21_hist.txt", "step_8/k_21/all_datasets_k21_hist.txt"):
        assert open(os.path.join(root, rel)).read() == open(os.path.join(root2, rel)).read()


def test_multi_genome_groups_and_k_sweep_match_oracle(tmp_path):
    root = str(tmp_path)
    synth.write_dataset_tree(root, 2, 3, 40_000)
    ks = [7, 21, 31, 41]
    out = W.run_batched(root, ks, 2)
    # expected CSVs: oracle histograms through the (golden-pinned) summariser
    exp = str(tmp_path / "expected")
    for k in ks:
        unions = []
        for s in (1, 2):
            sets = []
            for g in W.genomes_of(root, s):
                fa = O.read_fasta_bytes(os.path.join(root, f"data/dataset_{s}/{g}.fna.gz"))
                sets.append(O.set_counts(O.build(fa, k), 1))
            u = O.union_sum(sets, 5000)
            unions.append(O.set_counts(u, 1))
            p = os.path.join(exp, f"step_4/k_{k}/dataset_{s}/dataset_{s}_k{k}_hist.txt")
            os.makedirs(os.path.dirname(p), exist_ok=True)
            open(p, "w").write(O.histogram_text(u, 65535))
            assert open(os.path.join(root, f"step_4/k_{k}/dataset_{s}/dataset_{s}_k{k}_hist.txt")).read() == \
                O.histogram_text(u, 65535)
        p = os.path.join(exp, f"step_8/k_{k}/all_datasets_k{k}_hist.txt")
        os.makedirs(os.path.dirname(p), exist_ok=True)
        open(p, "w").write(O.histogram_text(O.union_sum(unions, 5000), 65535))
    os.makedirs(os.path.join(exp, "data"), exist_ok=True)
    for s in (1, 2):
        os.symlink(os.path.join(root, f"data/dataset_{s}"), os.path.join(exp, f"data/dataset_{s}"))
    want = W._csv_stage(exp, [str(k) for k in ks], 2)
    assert out["within"] == want["within"] and out["across"] == want["across"]
    # rule-per-process run of one k gives the same step_4 file
    root3 = str(tmp_path / "procs")
    os.makedirs(root3)
    synth.write_dataset_tree(root3, 2, 3, 40_000)
    W.run(root3, [31], 2)
    for s in (1, 2):
        rel = f"step_4/k_31/dataset_{s}/dataset_{s}_k31_hist.txt"
        assert open(os.path.join(root3, rel)).read() == open(os.path.join(root, rel)).read()


def test_exp_type_2_and_4_call_forms(tmp_path):
    """simple intersect -ocsum / kmers_subtract (exp_type_2.smk:363-379) and dump -s
    (exp_type_4.smk:255-257) with the reference's argv."""
    root = str(tmp_path)
    k = 21
    rng_recs = synth.genome_records(1, 0, 30_000), synth.genome_records(1, 1, 30_000)
    for name, recs in zip(("pivot", "other"), rng_recs):
        open(os.path.join(root, name + ".fa"), "wb").write(synth.fasta_bytes(recs))
    os.makedirs(os.path.join(root, "tmp"))
    sh(f"kmc -fm -m64 -k{k} -ci1 pivot.fa pivot tmp/", root)
    sh(f"kmc -fm -m64 -k{k} -ci1 other.fa other tmp/", root)
    sh("kmc_tools transform pivot set_counts 1 pivot.transformed", root)
    sh("kmc_tools transform other set_counts 1 other.transformed", root)
    sh("kmc_tools simple pivot.transformed other.transformed intersect inter -ocsum", root)
    sh("kmc_tools simple pivot.transformed other.transformed kmers_subtract  sub ", root)
    sh("kmc_tools transform inter histogram inter.hist.txt", root)
    sh("kmc_tools transform sub histogram sub.hist.txt", root)
    sh("kmc_tools transform pivot dump -s pivot.txt", root)
    sh("kmc_tools transform inter dump -s inter.txt", root)
    da = O.build(synth.fasta_bytes(rng_recs[0]), k)
    db = O.build(synth.fasta_bytes(rng_recs[1]), k)
    sa, sb = O.set_counts(da, 1), O.set_counts(db, 1)
    assert open(os.path.join(root, "pivot.txt")).read() == O.dump_sorted_text(da, k)
    assert open(os.path.join(root, "inter.txt")).read() == O.dump_sorted_text(O.intersect(sa, sb, "sum"), k)
    assert open(os.path.join(root, "inter.hist.txt")).read() == O.histogram_text(O.intersect(sa, sb, "sum"), 255)
    assert open(os.path.join(root, "sub.hist.txt")).read() == O.histogram_text(O.kmers_subtract(sa, sb), 255)
    # the reference's own invariants on these histograms (exp_type_2.smk:183-184)
    from khoice_amd import summarize as S
    sub = S.read_histogram_file(os.path.join(root, "sub.hist.txt"))
    inter = S.read_histogram_file(os.path.join(root, "inter.hist.txt"))
    assert inter[0] == 0 and sum(sub[1:]) == 0
    S.summarize_histogram_type2(sub, inter, 2, False, k)
    # src/merge_lists.py:14-33 consumes the dumps by `line.split()`: first/second column
    first = open(os.path.join(root, "pivot.txt")).readline().split()
    assert len(first) == 2 and len(first[0]) == k and first[1].isdigit()


def test_cli_errors_are_loud_and_leave_no_outputs(tmp_path):
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "tmp"))
    r = sh("kmc -fm -m64 -k21 -ci1 missing.fna.gz out tmp/", root, ok=False)
    assert r.returncode != 0 and "missing.fna.gz" in r.stderr
    assert not os.path.exists(os.path.join(root, "out.kmc_pre"))
    r = sh("kmc_tools transform nothing set_counts 1 out2", root, ok=False)
    assert r.returncode != 0 and not os.path.exists(os.path.join(root, "out2.kmc_pre"))
    r = sh("kmc -fm -b -k21 x y tmp/", root, ok=False)
    assert r.returncode != 0 and "not supported" in r.stderr
    open(os.path.join(root, "bad.kmc_pre"), "wb").write(b"KMCP not ours")
    open(os.path.join(root, "bad.kmc_suf"), "wb").write(b"KMCS")
    r = sh("kmc_tools transform bad histogram h.txt", root, ok=False)
    assert r.returncode != 0 and "khoice_amd database" in r.stderr


def test_resident_server_serves_the_same_dag(tmp_path, monkeypatch):
    """bin/khoice_server: the rule processes become thin clients of ONE engine context
    (SURVEY §8f next #1); results are byte-identical to stand-alone processes."""
    import time
    root = str(tmp_path / "direct")
    root2 = str(tmp_path / "served")
    for r in (root, root2):
        os.makedirs(r)
        synth.write_dataset_tree(r, 2, 2, 30_000)
    direct = W.run(root, [21, 31], 2)
    sock = str(tmp_path / "khoice.sock")
    srv = subprocess.Popen([os.path.join(BIN, "khoice_server"), sock], stderr=subprocess.PIPE, text=True)
    try:
        for _ in range(200):
            if os.path.exists(sock):
                break
            time.sleep(0.05)
        assert os.path.exists(sock), "server did not come up"
        monkeypatch.setenv("KHOICE_SERVER", sock)
        served = W.run(root2, [21, 31], 2)
        # errors travel back through the client
        r = sh("kmc_tools transform nothing histogram h.txt", root2, ok=False)
        assert r.returncode != 0 and "nothing" in r.stderr
    finally:
        subprocess.run([os.path.join(BIN, "khoice_server"), "--stop", sock], timeout=30)
        srv.wait(timeout=30)
    assert served["within"] == direct["within"] and served["across"] == direct["across"]
    assert "served" in srv.stderr.read()
    for rel in ("step_4/k_31/dataset_1/dataset_1_k31_hist.txt", "step_8/k_21/all_datasets_k21_hist.txt"):
        assert open(os.path.join(root, rel)).read() == open(os.path.join(root2, rel)).read()
    # a dead socket path falls back to a local engine
    monkeypatch.setenv("KHOICE_SERVER", str(tmp_path / "gone.sock"))
    sh("kmc_tools transform step_3/k_21/dataset_1/dataset_1.transformed.combined histogram again.txt", root2)
    assert open(os.path.join(root2, "again.txt")).read() == \
        open(os.path.join(root2, "step_4/k_21/dataset_1/dataset_1_k21_hist.txt")).read()


def test_cli_extras_filelist_cs_and_complex_operators(tmp_path):
    """Forms beyond khoice's seven: @file lists, -cs, `simple ... union`, `complex` with * and -."""
    root = str(tmp_path)
    k = 15
    recs = [synth.genome_records(3, g, 12_000) for g in range(3)]
    for i, r in enumerate(recs):
        open(os.path.join(root, f"g{i}.fa"), "wb").write(synth.fasta_bytes(r))
    os.makedirs(os.path.join(root, "tmp"))
    open(os.path.join(root, "list.txt"), "w").write("g0.fa\ng1.fa\n")
    sh(f"kmc -fm -k{k} -ci1 -cs3 @list.txt both tmp/", root)
    for i in range(3):
        sh(f"kmc -fm -k{k} -ci1 g{i}.fa s{i} tmp/", root)
    dbs = [O.build(synth.fasta_bytes(r), k) for r in recs]
    both = O.build(synth.fasta_bytes(recs[0]) + synth.fasta_bytes(recs[1]), k, cs=3)
    sh("kmc_tools transform both dump -s both.txt", root)
    assert open(os.path.join(root, "both.txt")).read() == O.dump_sorted_text(both, k)
    sh("kmc_tools simple s0 s1 union u01 -ocmax -cs7", root)
    sh("kmc_tools transform u01 dump -s u01.txt", root)
    assert open(os.path.join(root, "u01.txt")).read() == O.dump_sorted_text(O.union2(dbs[0], dbs[1], "max", cs=7), k)
    open(os.path.join(root, "ops.txt"), "w").write(
        "INPUT:\na = s0\nb = s1\nc = s2\nOUTPUT:\nres = (a * b) - c\nOUTPUT_PARAMS:\n-cs200\n")
    sh("kmc_tools complex ops.txt", root)
    sh("kmc_tools transform res dump -s res.txt", root)
    want = O.kmers_subtract(O.intersect(dbs[0], dbs[1], "min", cs=1 << 30), dbs[2])
    want = {c: min(n, 200) for c, n in want.items()}
    assert open(os.path.join(root, "res.txt")).read() == O.dump_sorted_text(want, k)
    # default -ci2 of kmc (khoice always passes -ci1): singletons are dropped
    sh(f"kmc -fm -k{k} g0.fa d2 tmp/", root)
    sh("kmc_tools transform d2 dump -s d2.txt", root)
    assert open(os.path.join(root, "d2.txt")).read() == O.dump_sorted_text(O.build(synth.fasta_bytes(recs[0]), k, ci=2), k)


def test_concurrent_processes_share_the_gpu(tmp_path):
    """`snakemake --cores N` runs N rule processes at once: independent contexts on one GPU, and
    concurrent clients of one server, must all produce the same databases as serial runs."""
    import time
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "tmp"))
    names = []
    for g in range(4):    # the GPU box admits 6 processes on the card; pytest itself is one
        name = f"g{g}"
        open(os.path.join(root, name + ".fa"), "wb").write(synth.fasta_bytes(synth.genome_records(5, g, 150_000)))
        names.append(name)
    jobs = " & ".join(f"kmc -fm -m64 -k31 -ci1 {n}.fa par_{n} tmp/ > /dev/null" for n in names)
    sh(f"{jobs} & wait", root)
    for n in names:
        sh(f"kmc -fm -m64 -k31 -ci1 {n}.fa ser_{n} tmp/ > /dev/null", root)
        for ext in (".kmc_pre", ".kmc_suf"):
            assert open(os.path.join(root, f"par_{n}{ext}"), "rb").read() == \
                open(os.path.join(root, f"ser_{n}{ext}"), "rb").read(), n
    sock = os.path.join(root, "s.sock")
    srv = subprocess.Popen([os.path.join(BIN, "khoice_server"), sock], stderr=subprocess.DEVNULL)
    try:
        for _ in range(200):
            if os.path.exists(sock):
                break
            time.sleep(0.05)
        jobs = " & ".join(f"KHOICE_SERVER={sock} kmc_tools transform ser_{n} set_counts 1 set_{n}" for n in names)
        sh(f"{jobs} & wait", root)
        ops = "INPUT:\n" + "".join(f"set{i + 1} = set_{n}\n" for i, n in enumerate(names)) + \
              "OUTPUT:\nunion = (" + " + ".join(f"set{i + 1}" for i in range(len(names))) + " )\nOUTPUT_PARAMS:\n-cs5000\n"
        open(os.path.join(root, "ops.txt"), "w").write(ops)
        sh(f"KHOICE_SERVER={sock} kmc_tools complex ops.txt && KHOICE_SERVER={sock} kmc_tools transform union histogram h.txt", root)
    finally:
        subprocess.run([os.path.join(BIN, "khoice_server"), "--stop", sock], timeout=30)
        srv.wait(timeout=30)
    dbs = [O.set_counts(O.build(open(os.path.join(root, n + ".fa"), "rb").read(), 31), 1) for n in names]
    assert open(os.path.join(root, "h.txt")).read() == O.histogram_text(O.union_sum(dbs, 5000), 65535)


def test_exp_type_4_rule_by_rule_and_batched_match_oracle(tmp_path):
    """exp_type_4.smk through bin/kmc, bin/kmc_tools and the merge_lists drop-in, and the batched
    runner (no dumps, no intersections): both equal the oracle's direct answer, byte for byte."""
    from khoice_amd.workflow import exp_type_4 as W4
    from tests.test_workflow_cpu import expected_type4_outputs
    root = str(tmp_path / "rules")
    os.makedirs(root)
    synth.write_type4_tree(root, 3, 2, 30_000)
    out = W4.run(root, [21], 3)
    root2 = str(tmp_path / "batched")
    os.makedirs(root2)
    synth.write_type4_tree(root2, 3, 2, 30_000)
    out2 = W4.run_batched(root2, [9, 21, 41], 3)
    assert out2["processes"] == 0
    for k in (9, 21, 41):
        want = expected_type4_outputs(root2, k, 3)
        for rel, text in want.items():
            assert open(os.path.join(root2, "accuracies_type_4", rel)).read() == text, (k, rel)
            if k == 21:
                assert open(os.path.join(root, "accuracies_type_4", rel)).read() == text, rel
    assert open(out["accuracy_values"]).read() == expected_type4_outputs(root, 21, 3)["values/k_21_accuracy_values.csv"]
    # the union histograms of the batched runner equal the rule's
    for num in (1, 2, 3):
        rel = f"unions_type_4/rest_of_set/k_21/dataset_{num}/dataset_{num}.hist.txt"
        assert open(os.path.join(root, rel)).read() == open(os.path.join(root2, rel)).read()


def test_exp_type_2_rule_by_rule_and_batched_match_oracle(tmp_path):
    """exp_type_2.smk (intersect -ocsum / kmers_subtract call sites, :354-380, :470-496) through
    bin/kmc + bin/kmc_tools and through the batched runner: histogram files and both CSVs equal
    the oracle's direct answer."""
    from khoice_amd.workflow import exp_type_2 as W2
    from tests.test_workflow_cpu import expected_type2_outputs
    root = str(tmp_path / "rules")
    os.makedirs(root)
    synth.write_type2_tree(root, 3, 2, 30_000)
    out = W2.run(root, [21], 3)
    files, within, across = expected_type2_outputs(root, [21], 3, str(tmp_path / "exp1"))
    for rel, text in files.items():
        assert open(os.path.join(root, rel)).read() == text, rel
    assert out["within"] == within and out["across"] == across
    root2 = str(tmp_path / "batched")
    os.makedirs(root2)
    synth.write_type2_tree(root2, 3, 2, 30_000)
    out2 = W2.run_batched(root2, [9, 21, 41], 3)
    files2, within2, across2 = expected_type2_outputs(root2, [9, 21, 41], 3, str(tmp_path / "exp2"))
    for rel, text in files2.items():
        assert open(os.path.join(root2, rel)).read() == text, rel
    assert out2["within"] == within2 and out2["across"] == across2 and out2["processes"] == 0
