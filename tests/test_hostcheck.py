"""CPU sanitizer leg: the product's host code — FASTA reader, database files, text outputs, the
`kmc` / `kmc_tools` argv forms and the `complex` operations-file parser (khoice_amd/csrc/kh_io.cpp, kh_cli.cpp) —
and the C restatement, built with -fsanitize=address,undefined against a host-only stand-in for the device
(tests/hostcheck/fake_engine.cpp) and driven through the reference's own call forms.  No GPU; never run on one."""
import gzip
import os
import shutil
import struct
import subprocess

import pytest

from khoice_amd import synth
from oracle import kmer_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
HC_DIR = os.path.join(HERE, "hostcheck")
FAKE = os.path.join(HERE, "fakebin")


@pytest.fixture(scope="module")
def hostcheck():
    if not (shutil.which("g++") and shutil.which("make")):
        pytest.skip("no host C++ toolchain")
    r = subprocess.run(["make", "-s", "-C", HC_DIR], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = os.path.join(HC_DIR, "_build", "hostcheck")
    assert os.path.exists(exe)
    return exe


@pytest.fixture(scope="module")
def san_bin(hostcheck, tmp_path_factory):
    """A PATH directory with `kmc` / `kmc_tools` that run the sanitized build."""
    d = tmp_path_factory.mktemp("sanbin")
    for tool in ("kmc", "kmc_tools"):
        p = d / tool
        p.write_text(f"#!/bin/sh\nexec {hostcheck} {tool} \"$@\"\n")
        p.chmod(0o755)
    return str(d)


def run(exe, *args, cwd=None):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, *args], capture_output=True, cwd=cwd, env=env)
    err = r.stderr.decode("latin-1")
    assert "AddressSanitizer" not in err and "runtime error:" not in err and "LeakSanitizer" not in err, err[-3000:]
    return r.returncode, r.stdout, err


def tree(root):
    out = {}
    for d, _, names in os.walk(root):
        for n in names:
            if n.endswith((".txt", ".csv")):
                p = os.path.join(d, n)
                out[os.path.relpath(p, root)] = open(p, "rb").read()
    return out


def test_exp_type_1_dag_under_sanitizers(san_bin, tmp_path):
    """exp_type_1.smk:156-308 rule by rule through the sanitized kmc / kmc_tools: same histogram files and CSVs as
    through the Python stand-ins (BASELINE configs[0] shape + a 3-genome group), and the oracle's histogram."""
    from khoice_amd.workflow import exp_type_1 as W
    roots = []
    for name, bin_dir in (("san", san_bin), ("py", FAKE)):
        root = str(tmp_path / name)
        os.makedirs(root)
        synth.write_dataset_tree(root, 2, 1, 6000)
        synth.write_dataset_tree(root, 3, 3, 4000)       # (adds / replaces: 3 groups of 3)
        W.run(root, [21, 33], 3, bin_dir=bin_dir)
        roots.append(root)
    a, b = tree(roots[0]), tree(roots[1])
    assert a.keys() == b.keys() and len(a) > 10
    for rel in a:
        assert a[rel] == b[rel], rel


def test_exp_type_2_dag_under_sanitizers(san_bin, tmp_path):
    """intersect -ocsum / kmers_subtract / multi-line complex files (exp_type_2.smk:289-554)."""
    from khoice_amd.workflow import exp_type_2 as W2
    outs = []
    for name, bin_dir in (("san", san_bin), ("py", FAKE)):
        root = str(tmp_path / name)
        os.makedirs(root)
        synth.write_type2_tree(root, 3, 2, 3000)
        outs.append((W2.run(root, [9, 41], 3, bin_dir=bin_dir), tree(root)))
    assert outs[0][0]["within"] == outs[1][0]["within"] and outs[0][0]["across"] == outs[1][0]["across"]
    assert outs[0][1] == outs[1][1]


def test_dump_and_file_round_trip_under_sanitizers(hostcheck, tmp_path):
    fa = tmp_path / "g.fna.gz"
    text = b">r1\nACGTNNACGTTGCA\nacgtacgtaa\r\n>r2\n" + b"ACGT" * 40 + b"\n>empty\n"
    with gzip.open(fa, "wb") as fh:
        fh.write(text)
    os.makedirs(tmp_path / "tmp")
    for k in (5, 33):
        rc, _, err = run(hostcheck, "kmc", "-fm", "-m64", f"-k{k}", "-ci1", str(fa), str(tmp_path / f"db{k}"), str(tmp_path / "tmp"))
        assert rc == 0, err
        rc, _, err = run(hostcheck, "kmc_tools", "transform", str(tmp_path / f"db{k}"), "dump", "-s", str(tmp_path / f"d{k}.txt"))
        assert rc == 0, err
        assert open(tmp_path / f"d{k}.txt").read() == O.dump_sorted_text(O.build(text, k), k)
        rc, _, err = run(hostcheck, "kmc_tools", "transform", str(tmp_path / f"db{k}"), "histogram", str(tmp_path / f"h{k}.txt"))
        assert rc == 0, err
        assert open(tmp_path / f"h{k}.txt").read() == O.histogram_text(O.build(text, k), 255)
    rc, out, _ = run(hostcheck, "read_fasta", str(fa))
    assert rc == 0 and out == b"ACGTNNACGTTGCAacgtacgtaa\n" + b"ACGT" * 40 + b"\n"   # (the separator in front of the last, empty record)


def test_malformed_inputs_fail_cleanly_under_sanitizers(hostcheck, tmp_path):
    """Truncated / corrupt database pairs, operation files and arguments: an error exit, never a sanitizer report."""
    fa = tmp_path / "g.fa"
    fa.write_bytes(b">r\n" + b"ACGTTGCATTGACC" * 30 + b"\n")
    os.makedirs(tmp_path / "tmp")
    db = str(tmp_path / "db")
    assert run(hostcheck, "kmc", "-k11", "-ci1", str(fa), db, str(tmp_path / "tmp"))[0] == 0
    pre, suf = open(db + ".kmc_pre", "rb").read(), open(db + ".kmc_suf", "rb").read()

    def variant(name, p, s):
        open(str(tmp_path / name) + ".kmc_pre", "wb").write(p)
        open(str(tmp_path / name) + ".kmc_suf", "wb").write(s)
        return str(tmp_path / name)
    bad = [variant("trunc_suf", pre, suf[: len(suf) // 2]), variant("trunc_pre", pre[:10], suf),
           variant("empty_pre", b"", suf), variant("bad_magic", b"KMCP" + pre[4:], suf),
           variant("suf_magic", pre, b"XXXXXXXX" + suf[8:]), variant("extra_tail", pre, suf + b"\0" * 5)]
    # a header that claims far more records than the file holds (the count sits behind the magic and the version)
    for off in range(8, min(len(pre), 64), 4):
        bad.append(variant(f"huge_{off}", pre[:off] + struct.pack("<Q", 1 << 60) + pre[off + 8:], suf))
    for b in bad:
        rc, _, err = run(hostcheck, "kmc_tools", "transform", b, "histogram", str(tmp_path / "h.txt"))
        if not os.path.basename(b).startswith("huge_"):
            assert rc != 0 and err, b
        assert not os.path.exists(str(tmp_path / "h.txt")) or rc == 0
        if os.path.exists(str(tmp_path / "h.txt")):
            os.remove(str(tmp_path / "h.txt"))
    ops = {
        "empty": "",
        "no_output": "INPUT:\nset1 = %s\n" % db,
        "undefined": "INPUT:\nset1 = %s\nOUTPUT:\n%s = (set1 + set2 )\n" % (db, tmp_path / "o"),
        "unbalanced": "INPUT:\nset1 = %s\nOUTPUT:\n%s = ((set1 + set1 \n" % (db, tmp_path / "o"),
        "trailing": "INPUT:\nset1 = %s\nOUTPUT:\n%s = (set1 ) set1\n" % (db, tmp_path / "o"),
        "missing_db": "INPUT:\nset1 = %s\nOUTPUT:\n%s = (set1 )\n" % (tmp_path / "nope", tmp_path / "o"),
        "outside": "hello\nINPUT:\n",
        "bad_param": "INPUT:\nset1 = %s\nOUTPUT:\n%s = (set1 )\nOUTPUT_PARAMS:\n-csabc\n" % (db, tmp_path / "o"),
        "operators": "INPUT:\nset1 = %s\nset2 = %s\nOUTPUT:\n%s = set1 * set2 - set1 ~ set2 + set1\nOUTPUT_PARAMS:\n-cs7\n" % (db, db, tmp_path / "okay"),
        "long_line": "INPUT:\nset1 = %s\nOUTPUT:\n%s = (%s)\n" % (db, tmp_path / "o2", " + ".join(["set1"] * 300)),
    }
    for name, text in ops.items():
        p = tmp_path / f"ops_{name}.txt"
        p.write_text(text)
        rc, _, err = run(hostcheck, "kmc_tools", "complex", str(p))
        assert (rc == 0) == (name in ("operators", "long_line")), (name, rc, err)
    for args in (["kmc"], ["kmc", "-k0", str(fa), db, "tmp"], ["kmc", "-k65", str(fa), db, "tmp"], ["kmc", "-kx", str(fa), db, "tmp"],
                 ["kmc", "-b", str(fa), db, "tmp"], ["kmc", "-k5", str(tmp_path / "none.fa"), db, "tmp"],
                 ["kmc_tools"], ["kmc_tools", "transform", db], ["kmc_tools", "transform", db, "set_counts", "0", db + "x"],
                 ["kmc_tools", "transform", db, "set_counts", "-3", db + "x"], ["kmc_tools", "simple", db, db, "intersect"],
                 ["kmc_tools", "simple", db, db, "intersect", db + "y", "-ocfoo"], ["kmc_tools", "frobnicate"],
                 ["hist_text", str(tmp_path / "big.txt"), "4294967295", "0", "1"]):
        rc, _, err = run(hostcheck, *args)
        assert rc != 0, args
