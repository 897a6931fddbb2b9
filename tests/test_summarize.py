"""CSV stage vs golden vectors captured from the reference's own functions
(tests/golden/make_golden.py; exp_type_1.smk:115-150,199-231,268-297;
exp_type_2.smk:171-216).  Bit-exact: floats compared with ==, CSV text with ==."""
import os

import pytest

from khoice_amd import summarize as S


def _dense(length, nonzero):
    h = [0] * length
    for i, v in nonzero:
        h[i] = v
    return h


def test_type1_matches_reference(golden):
    cases = golden("summarize_type1.json")
    assert len(cases) > 50
    for c in cases:
        h = _dense(c["len"], c["nonzero"])
        got = S.summarize_histogram_type1(h, c["n_members"], c["across"], c["k"])
        assert got == c["metrics"], c
        assert [repr(x) for x in got] == [repr(x) for x in c["metrics"]]


def test_type1_trailing_zero_bins_are_neutral(golden):
    c = golden("summarize_type1.json")[0]
    h = _dense(c["len"], c["nonzero"])
    a = S.summarize_histogram_type1(h, c["n_members"], c["across"], c["k"])
    b = S.summarize_histogram_type1(h + [0] * 1000, c["n_members"], c["across"], c["k"])
    assert a == b


def test_type2_matches_reference(golden):
    cases = golden("summarize_type2.json")
    assert len(cases) > 30
    for c in cases:
        sub = _dense(c["len"], [(0, c["sub0"])])
        inter = _dense(c["len"], c["inter_nonzero"])
        got = S.summarize_histogram_type2(sub, inter, c["n"], c["across"], c["k"])
        assert got == c["metrics"], c


def test_type1_error_behaviour():
    with pytest.raises(ZeroDivisionError):
        S.summarize_histogram_type1([0, 0, 0], 5, False, 31)
    with pytest.raises(AssertionError):
        S.summarize_histogram_type2([1, 0], [1, 0], 5, False, 31)
    with pytest.raises(AssertionError):
        S.summarize_histogram_type2([1, 1], [0, 0], 5, False, 31)


def test_csv_bytes_match_reference(golden, tmp_path, monkeypatch):
    g = golden("exp1_csv.json")
    monkeypatch.chdir(tmp_path)
    for path, nonzero in g["hists"].items():
        os.makedirs(os.path.dirname(path), exist_ok=True)
        h = _dense(g["hist_len"], nonzero)
        with open(path, "w") as fh:
            fh.write("".join(f"{i + 1}\t{v}\n" for i, v in enumerate(h)))
    members = {k: v for k, v in g["members"].items()}
    within = S.within_groups_csv(g["within_inputs"], g["num_datasets"],
                                 lambda num: members[str(num)])
    across = S.across_groups_csv(g["across_inputs"], g["num_datasets"])
    assert within == g["within_csv"]
    assert across == g["across_csv"]


def test_type2_csv_writers_match_reference_rule_bodies(tmp_path, monkeypatch):
    """exp_type_2.smk:404-438 and :521-554 run on the same histogram files (golden/exp2_csv.json)."""
    import json
    import os
    from khoice_amd import summarize as S
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "exp2_csv.json")))
    monkeypatch.chdir(tmp_path)
    for rel, nz in g["hists"].items():
        h = [0] * g["hist_len"]
        for i, v in nz:
            h[i] = v
        os.makedirs(os.path.dirname(rel), exist_ok=True)
        open(rel, "w").write("".join(f"{i + 1}\t{v}\n" for i, v in enumerate(h)))
    members = {int(a): b for a, b in g["members"].items()}
    assert S.pivot_within_groups_csv(g["within_inputs"], g["num_datasets"], lambda n: members[int(n)]) == g["within_csv"]
    assert S.pivot_across_groups_csv(g["across_inputs"], g["num_datasets"]) == g["across_csv"]
