"""Experiment type 4 (src/merge_lists.py, feature level): host arithmetic and formatting of
khoice_amd.merge_lists, and the oracle restatement, against outputs produced by the
reference's own merge_lists.main (tests/golden/merge_lists.json, made by make_golden.py)."""
import json
import os

import numpy as np
import pytest

from khoice_amd import merge_lists as ML
from oracle import kmer_oracle as O
from oracle import merge_oracle as MO

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "merge_lists.json")))["cases"]


def parse_dump(text):
    return {O.encode(line.split()[0]): int(line.split()[1]) for line in text.splitlines()}


@pytest.mark.parametrize("case", GOLD, ids=lambda c: f"k{c['k']}_n{c['num_datasets']}")
def test_oracle_and_host_mirror_reproduce_reference_outputs_from_dumps(case):
    n, k = case["num_datasets"], case["k"]
    rows, uniques = [], []
    for p in range(n):
        pivot = parse_dump(case["pivot_dumps"][p])
        inter = [parse_dump(case["intersection_dumps"][p * n + d]) for d in range(n)]
        row, unique = MO.confusion_row(pivot, inter)
        rows.append(row)
        uniques.append(unique)
    cm, cm_ucol = ML.assemble_matrices(rows, uniques, n)
    assert ML.format_outputs(cm, cm_ucol, n, str(k)) == case["outputs"]


@pytest.mark.parametrize("case", GOLD, ids=lambda c: f"k{c['k']}_n{c['num_datasets']}")
def test_same_outputs_from_databases_without_dumps_or_intersections(case):
    """membership in the rest-of-set union == presence in the intersection dump."""
    n, k = case["num_datasets"], case["k"]
    unions = [O.set_counts(O.union_sum([O.set_counts(O.count_records([g], k), 1) for g in gs], 5000), 1)
              for gs in case["rest_of_set"]]
    rows, uniques = zip(*[MO.confusion_row(O.count_records([p], k), unions) for p in case["pivots"]])
    cm, cm_ucol = ML.assemble_matrices(rows, uniques, n)
    assert ML.format_outputs(cm, cm_ucol, n, str(k)) == case["outputs"]


def test_untouched_cells_print_like_the_reference():
    cm, cm_ucol = ML.assemble_matrices([[0.0, 2.5], [0.0, 0.0]], [0, 3], 2)
    assert cm_ucol == [[0, 2.5, 0], [0, 0, 0]]
    out = ML.format_outputs(cm, cm_ucol, 2, "9")
    assert out["confusion_matrix/k_9_confusion_matrix_with_unidentified.txt"] == "0,2.5,0\n0,0,0\n"
    assert out["confusion_matrix/k_9_confusion_matrix.txt"] == "0.0,2.5,0\n1.5,1.5,0\n"


@pytest.mark.parametrize("k", [5, 31, 32, 33, 63])
def test_read_dump_parses_what_dump_sorted_writes(tmp_path, k):
    import random
    rng = random.Random(k)
    db = O.count_records(["".join(rng.choice("ACGT") for _ in range(400)) + "A" * (k + 30)], k)
    p = tmp_path / "d.txt"
    p.write_text(O.dump_sorted_text(db, k))
    keys, counts = ML.read_dump(str(p), k)
    want = sorted(db)
    got = [int(r[0]) | (int(r[1]) << 64 if keys.shape[1] == 2 else 0) for r in keys]
    assert got == want and counts.tolist() == [db[c] for c in want]
    assert max(counts) > 9          # multi-digit counts
    empty = tmp_path / "e.txt"
    empty.write_text("")
    assert ML.read_dump(str(empty), k)[0].shape[0] == 0


def test_read_level_is_refused(tmp_path, capsys):
    f = tmp_path / "l.txt"
    f.write_text("")
    rc = ML.main(["-n", "1", "-p", str(f), "-i", str(f), "-o", str(tmp_path) + "/", "-k", "5", "-r", str(tmp_path)])
    assert rc == 1 and "read-level" in capsys.readouterr().err
