"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, exports
every symbol include/khoice_hip.h declares, fails loudly without a GPU, and its host
helpers (key mixing, FASTA ingest) behave.  No device compute here."""
import gzip
import os
import re

import numpy as np
import pytest

from khoice_amd import build as kbuild
from khoice_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    kbuild.build_library()
    return E.load_library()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "khoice_hip.h")).read()
    declared = set(re.findall(r"\b(kh_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"kh_ctx", "kh_set"}
    assert declared == set(E.ABI_SYMBOLS), declared ^ set(E.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_no_device_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(E.KhoiceError) as ei:
        E.Engine(0)
    assert "no CPU fallback" in str(ei.value) or "HIP" in str(ei.value)


@pytest.mark.parametrize("k", [1, 2, 3, 7, 15, 16, 21, 31, 32, 33, 34, 41, 48, 63, 64])
def test_mix_is_a_bijection_with_inverse(lib, k):
    rng = np.random.default_rng(k)
    w = 1 if k <= 32 else 2
    nbits = 2 * k
    seen = set()
    for _ in range(300):
        v = int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 1 << 62)) << 62) | (int(rng.integers(0, 16)) << 124)
        v &= (1 << nbits) - 1
        a = np.array([v & (2**64 - 1), v >> 64][:w], dtype=np.uint64)
        m = E.mix_host(k, a)
        mv = int(m[0]) | (int(m[1]) << 64 if w == 2 else 0)
        assert mv < (1 << nbits)
        back = E.unmix_host(k, m)
        assert (back == a).all()
        seen.add((v, mv))
    # small k: exhaustively a permutation
    if nbits <= 12:
        imgs = {int(E.mix_host(k, np.array([v], dtype=np.uint64))[0]) for v in range(1 << nbits)}
        assert imgs == set(range(1 << nbits))


def test_mix_top_bits_are_balanced_on_biased_input(lib):
    # AT-rich low-complexity keys must still spread over buckets
    rng = np.random.default_rng(5)
    k = 31
    nb = 64
    hist = np.zeros(nb, dtype=np.int64)
    for _ in range(20000):
        bases = rng.choice(4, size=k, p=[0.45, 0.05, 0.05, 0.45])
        v = 0
        for b in bases:
            v = (v << 2) | int(b)
        m = int(E.mix_host(k, np.array([v], dtype=np.uint64))[0])
        hist[((m >> (2 * k - 32)) * nb) >> 32] += 1
    assert hist.min() > 0.6 * hist.mean() and hist.max() < 1.5 * hist.mean()


def test_read_fasta_host(lib, tmp_path):
    text = b">r1 desc\nACGT\nNNAC\r\n\n>r2\nacgt\n>empty\n>r3\nTT\n"
    plain = tmp_path / "a.fa"
    plain.write_bytes(text)
    gz = tmp_path / "a.fna.gz"
    with gzip.open(gz, "wb") as fh:
        fh.write(text)
    eng = object.__new__(E.Engine)
    eng._lib = lib
    want = b"ACGTNNAC\nacgt\nTT"
    assert E.Engine.read_fasta(eng, str(plain)) == want
    assert E.Engine.read_fasta(eng, str(gz)) == want
    with pytest.raises(E.KhoiceError):
        E.Engine.read_fasta(eng, str(tmp_path / "missing.fa"))


def test_read_fasta_host_gives_the_oracles_records(lib, tmp_path):
    """The host FASTA reader against the oracle's independent one (oracle/kmer_oracle.py fasta_records, SURVEY
    App. A.1) — no GPU needed: CRLF line ends, a header at the end of the file, empty records, sequence before
    the first header, '>' inside a line, a header line straddling a 4 KB boundary, random structure bytes.  What
    has to agree is what the device will count: the non-empty runs of sequence, record by record."""
    import random
    from oracle import kmer_oracle as O
    rng = random.Random(11)
    cases = {
        "crlf": ">r1\r\nACGT\r\nGGNN\r\n>r2\r\nTT\r\n",
        "header_at_eof": ">r1\nACGT\n>r2",
        "empty_records": ">a\n>b\n>c\nACGT\n>d\n>e\nGGCC\n>f\n",
        "sequence_first": "ACGTAC\nGT\n>r1\nTTTT\n",
        "gt_inside_line": ">r1\nAC>GT\nA>\n>r2\n>>\nAC\n",
        "header_over_4k": ">r\n" + "A" * 4090 + "\n>" + "h" * 5000 + "\n" + "C" * 100 + "\n",
        "blank_lines": "\n\n>r1\n\nACGT\n\n\nTTGA\n\n>r2\n\n",
        "empty": "",
    }
    for i in range(20):
        n = rng.choice([1, 17, 4095, 4096, 4097, 30_000])
        cases[f"fuzz{i}"] = "".join(rng.choice("ACGTN>\n\n" if i % 2 else "ACGTacgtRY>\n") for _ in range(n))
    eng = object.__new__(E.Engine)
    eng._lib = lib
    for name, text in cases.items():
        for crlf in (False, True):
            data = (text.replace("\n", "\r\n") if crlf and "\r" not in text else text).encode()
            p = tmp_path / f"{name}_{int(crlf)}.fa"
            p.write_bytes(data)
            got = E.Engine.read_fasta(eng, str(p)).decode("latin-1")
            want = [r for r in O.fasta_records(data) if r]
            assert [r for r in got.split("\n") if r] == want, name
