"""GPU parity for experiment type 4: kh_membership / kh_confusion_row through the C ABI against
the outputs of the reference's own merge_lists.main (tests/golden/merge_lists.json) and against
the oracle at sizes with two-word keys and more than 64 sets."""
import json
import os
import random

import numpy as np
import pytest

from khoice_amd import merge_lists as ML
from oracle import kmer_oracle as O
from oracle import merge_oracle as MO
from tests.util import random_dna

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "merge_lists.json")))["cases"]


@pytest.fixture(scope="module")
def eng():
    from khoice_amd import build as kbuild
    from khoice_amd import engine as E
    kbuild.build_library()
    e = E.Engine(0)
    yield e
    e.close()


def union_set(eng, genomes, k):
    sets = [eng.build(g.encode(), k).set_counts(1) for g in genomes]
    return eng.union_sum(sets, 5000).set_counts(1)


@pytest.mark.parametrize("case", GOLD, ids=lambda c: f"k{c['k']}_n{c['num_datasets']}")
def test_confusion_matrix_from_sets_equals_reference_output(eng, case):
    n, k = case["num_datasets"], case["k"]
    unions = [union_set(eng, gs, k) for gs in case["rest_of_set"]]
    pivots = [eng.build(p.encode(), k) for p in case["pivots"]]
    assert ML.confusion_from_sets(eng, pivots, [unions] * n, n, str(k)) == case["outputs"]


@pytest.mark.parametrize("case", GOLD[:3], ids=lambda c: f"k{c['k']}_n{c['num_datasets']}")
def test_drop_in_main_from_text_dumps_and_from_databases(eng, case, tmp_path):
    n, k = case["num_datasets"], case["k"]
    os.makedirs(tmp_path / "out" / "confusion_matrix")
    os.makedirs(tmp_path / "out" / "values")
    pl, il = [], []
    for p in range(n):
        f = tmp_path / f"pivot_{p + 1}.txt"
        f.write_text(case["pivot_dumps"][p])
        pl.append(str(f))
        for d in range(n):
            g = tmp_path / f"pivot_{p + 1}_intersect_dataset_{d + 1}.txt"
            g.write_text(case["intersection_dumps"][p * n + d])
            il.append(str(g))
    (tmp_path / "pivots.txt").write_text("".join(x + "\n" for x in pl))
    (tmp_path / "inters.txt").write_text("".join(x + "\n" for x in il))
    rc = ML.main(["-n", str(n), "-p", str(tmp_path / "pivots.txt"), "-i", str(tmp_path / "inters.txt"),
                  "-o", str(tmp_path / "out") + "/", "-k", str(k)])
    assert rc == 0
    for rel, text in case["outputs"].items():
        assert (tmp_path / "out" / rel).read_text() == text
    # databases instead of dumps (no intersections at all)
    os.makedirs(tmp_path / "db" / "confusion_matrix")
    os.makedirs(tmp_path / "db" / "values")
    up, pp = [], []
    for d, gs in enumerate(case["rest_of_set"]):
        union_set(eng, gs, k).save(str(tmp_path / f"union_{d}"))
        up.append(str(tmp_path / f"union_{d}"))
    for p, seq in enumerate(case["pivots"]):
        eng.build(seq.encode(), k).save(str(tmp_path / f"pv_{p}"))
        pp.append(str(tmp_path / f"pv_{p}"))
    ML.run_from_databases(eng, pp, up, str(k), str(tmp_path / "db") + "/")
    for rel, text in case["outputs"].items():
        assert (tmp_path / "db" / rel).read_text() == text


@pytest.mark.parametrize("k,nsets", [(13, 3), (31, 70), (41, 5), (63, 66)])
def test_membership_masks_match_oracle(eng, k, nsets):
    rng = random.Random(k * 100 + nsets)
    base = random_dna(rng, 60_000)
    texts = []
    for d in range(nsets):
        a = rng.randrange(0, 50_000)
        b = a + rng.randrange(200, 10_000 if nsets > 10 else 40_000)
        texts.append(base[a:b] + "N" + random_dna(rng, rng.randrange(0, 3_000)))
    texts[1] = ""                                      # an empty set
    pivot_text = base[5_000:45_000] + "N" + base[100:400] + "N" + random_dna(rng, 5_000)
    sets = [eng.build(t.encode(), k).set_counts(1) for t in texts]
    pivot = eng.build(pivot_text.encode(), k)
    keys, counts, masks = eng.membership(pivot, sets)
    pdb = O.count_records([pivot_text], k)
    dbs = [O.count_records([t], k) for t in texts]
    want_keys = sorted(pdb)
    got_keys = [int(r[0]) | ((int(r[1]) << 64) if keys.shape[1] == 2 else 0) for r in keys]
    assert got_keys == want_keys                       # dump order
    assert counts.tolist() == [pdb[c] for c in want_keys]
    assert masks.shape == (len(want_keys), (nsets + 63) // 64)
    step = max(1, len(want_keys) // 3000)
    for i in range(0, len(want_keys), step):
        want = 0
        for d, db in enumerate(dbs):
            if want_keys[i] in db:
                want |= 1 << d
        got = sum(int(masks[i, w]) << (64 * w) for w in range(masks.shape[1]))
        assert got == want
    row, unique = eng.confusion_row(pivot, sets)
    wrow, wunique = MO.confusion_row(pdb, dbs)
    assert unique == wunique and row.tolist() == wrow  # bit-exact float64 sums


def test_membership_errors_and_empty(eng):
    from khoice_amd.engine import KhoiceError
    a = eng.build(b"ACGTACGTTTGACCA", 5)
    b = eng.build(b"ACGTACGTTTGACCA", 7)
    with pytest.raises(KhoiceError):
        eng.confusion_row(a, [b])
    row, unique = eng.confusion_row(a, [])
    assert row.size == 0 and unique == int(a.download()[1].sum())
    e = eng.build(b"ACG", 5)
    row, unique = eng.confusion_row(e, [a])
    assert row.tolist() == [0.0] and unique == 0
