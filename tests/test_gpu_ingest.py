"""Device-side FASTA ingest (kh_ingest_fasta: parallel inflate + stream-compaction kernels) against the
library's own CPU reader kh_read_fasta, byte for byte, AND against the oracle's independent reader
(oracle/kmer_oracle.py fasta_records: the k-mer database of every file must be the one the oracle builds from
the raw bytes), on inputs built to hit every rule of the -fm reader and every tile boundary."""
import gzip
import os
import random

import pytest

from oracle import kmer_oracle as O
from tests.util import random_dna, set_to_db

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from khoice_amd import build as kbuild
    from khoice_amd import engine as E
    kbuild.build_library()
    e = E.Engine(0)
    yield e
    e.close()


def cases():
    rng = random.Random(2024)
    seq = random_dna(rng, 30_000, "ACGTN")
    wrapped = "\n".join(seq[i:i + 70] for i in range(0, len(seq), 70))
    out = {
        "wrapped": ">r1 some description\n" + wrapped + "\n>r2\n" + wrapped[:5000] + "\n",
        "crlf": ">r1\r\n" + wrapped.replace("\n", "\r\n") + "\r\n>r2\r\nACGT\r\n",
        "no_trailing_newline": ">r1\n" + wrapped,
        "sequence_first": wrapped[:300] + "\n>r1\nACGTACGT\n",
        "consecutive_headers": ">a\n>b\n>c\nACGT\n>d\n>e\nGGCC\n>f\n",
        "blank_lines": "\n\n>r1\n\nACGT\n\n\nTTGA\n\n>r2\n\n",
        "gt_inside_line": ">r1\nAC>GT\nA>\n>r2\n>>\nAC\n",
        "long_header": ">" + "h" * 9000 + "\nACGT\n>" + "x" * 4095 + "\n" + seq[:100] + "\n",
        "unwrapped": ">r1\n" + seq + "\n>r2\n" + seq[::-1] + "\n",
        "empty": "",
        "header_only": ">nothing here",
        "newlines_only": "\n\n\n",
        "cr_before_header": "ACGT\n\r>hdr\nGG\n\r\rAC\n",
        "lower_iupac": ">r\nacgtnryswkm\nACGT\n",
        "tile_edges": ">r\n" + "A" * 4093 + "\n>" + "h" * 4094 + "\n" + "C" * 8190 + "\n>z\nG\n",
    }
    for i in range(12):       # fuzz: every structure byte, lengths straddling tiles
        n = rng.choice([1, 15, 16, 17, 4095, 4096, 4097, 8192, 20_000, 70_000])
        out[f"fuzz{i}"] = "".join(rng.choice("ACGTN>\n\n\r" if i % 2 else "ACGT>\n") for _ in range(n))
    return out


def test_device_clean_equals_cpu_reader(eng, tmp_path):
    paths, names = [], []
    for name, text in cases().items():
        for gz in (False, True):
            p = str(tmp_path / (name + (".fna.gz" if gz else ".fa")))
            if gz:
                with gzip.open(p, "wb") as fh:
                    fh.write(text.encode())
            else:
                with open(p, "wb") as fh:
                    fh.write(text.encode())
            paths.append(p)
            names.append(name + ("/gz" if gz else "/plain"))
    for threads in (1, 7):
        texts = eng.ingest_fasta(paths, threads=threads)
        assert len(texts.seqs) == len(paths)
        for i, (p, name) in enumerate(zip(paths, names)):
            want = eng.read_fasta(p)
            got = texts.download(i)
            assert got == want, (name, threads, len(got), len(want))
        texts.free()


def oracle_cases():
    """Inputs on which the oracle's reader and the product's are specified alike: line ends LF or CRLF (a CR that
    is not followed by LF is outside what either documents: the oracle keeps it as a run-breaking symbol, the
    product drops it — product-defined, covered by the byte comparison above)."""
    rng = random.Random(77)
    out = {name: text for name, text in cases().items() if "\r" not in text}
    out["crlf"] = cases()["crlf"]
    for i in range(16):       # fuzz without stray CRs: every structure byte, lengths straddling the 4 KB tiles
        n = rng.choice([1, 15, 16, 17, 4095, 4096, 4097, 8191, 8192, 8193, 20_000, 70_000])
        text = "".join(rng.choice("ACGTN>\n\n" if i % 2 else "ACGTacgtRY>\n") for _ in range(n))
        out[f"ofuzz{i}"] = text
        if i % 4 == 0:
            out[f"ofuzz{i}_crlf"] = text.replace("\n", "\r\n")
    return out


def test_device_clean_gives_the_oracles_kmers(eng, tmp_path):
    """f2 against an implementation that shares no code with the product: raw file bytes -> oracle FASTA reader
    -> oracle counting, versus file -> kh_ingest_fasta (device clean) -> kh_build_batch, and versus file ->
    kh_read_fasta -> kh_build_batch."""
    items = list(oracle_cases().items())
    paths = []
    for j, (name, text) in enumerate(items):
        p = str(tmp_path / (name + (".fna.gz" if j % 2 else ".fa")))
        with (gzip.open(p, "wb") if j % 2 else open(p, "wb")) as fh:
            fh.write(text.encode())
        paths.append(p)
    texts = eng.ingest_fasta(paths, threads=5)
    for k in (3, 12):
        dev_sets = eng.build_batch(texts.seqs, k)
        host_sets = eng.build_batch([eng.read_fasta(p) for p in paths], k)
        for (name, text), ds, hs in zip(items, dev_sets, host_sets):
            want = O.count_records(O.fasta_records(text.encode()), k)
            assert set_to_db(ds) == want, (name, k, "device clean")
            assert set_to_db(hs) == want, (name, k, "host reader")
    texts.free()


def test_ingested_texts_feed_the_fused_step(eng, tmp_path):
    """gz genomes -> kh_ingest_fasta -> kh_exp1_run on the resident texts == the same from host texts."""
    from khoice_amd import synth
    root = str(tmp_path)
    synth.write_dataset_tree(root, 3, 2, 150_000)
    names = [(num, g[:-len(".fna.gz")]) for num in (1, 2, 3)
             for g in sorted(os.listdir(os.path.join(root, f"data/dataset_{num}")))]
    paths = [os.path.join(root, f"data/dataset_{n}/{g}.fna.gz") for n, g in names]
    group_of = [n - 1 for n, _ in names]
    texts = eng.ingest_fasta(paths)
    host = [eng.read_fasta(p) for p in paths]
    a = eng.exp1_run(texts.seqs, group_of, 31, hist_len=64)
    b = eng.exp1_run(host, group_of, 31, hist_len=64)
    assert (a["within_hist"] == b["within_hist"]).all() and (a["across_hist"] == b["across_hist"]).all()
    assert (a["distinct_per_seq"] == b["distinct_per_seq"]).all()
    texts.free()
    with pytest.raises(Exception):
        eng.ingest_fasta([paths[0], os.path.join(root, "missing.fna.gz")])


def test_batched_runner_without_databases_matches_with_databases(tmp_path):
    """run_batched(keep_databases=False) — device ingest, fused step, histogram files written from
    arrays, CSV stage fed from memory — gives byte-identical CSVs and histogram files."""
    from khoice_amd import synth
    from khoice_amd.workflow import exp_type_1 as W
    outs = []
    for keep in (True, False):
        root = str(tmp_path / f"keep_{keep}")
        os.makedirs(root)
        synth.write_dataset_tree(root, 3, 2, 120_000)
        res = W.run_batched(root, [21, 31], 3, keep_databases=keep)
        hist = {}
        for k in (21, 31):
            for num in (1, 2, 3):
                rel = f"step_4/k_{k}/dataset_{num}/dataset_{num}_k{k}_hist.txt"
                hist[rel] = open(os.path.join(root, rel)).read()
            rel = f"step_8/k_{k}/all_datasets_k{k}_hist.txt"
            hist[rel] = open(os.path.join(root, rel)).read()
        outs.append((res["within"], res["across"], hist))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    assert outs[0][2] == outs[1][2]
