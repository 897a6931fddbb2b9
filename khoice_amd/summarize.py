"""CSV stage of khoice experiment type 1 / 2.

Turns the histograms the engine produces into the `step_5` / `step_9` CSVs that
khoice's R scripts read.  This is the only floating-point code on the path and it is
deliberately kept in CPython: the reference computes these numbers with `int/int`
true division, left-to-right float accumulation, `round()` and `str(float)`, and the
CSV bytes must match it exactly.

Reference behaviour reproduced (function names are kept so call sites read alike):
  * summarize_histogram_type1      workflow/rules/exp_type_1.smk:115-150
  * within-group CSV writer        workflow/rules/exp_type_1.smk:199-231
  * across-group CSV writer        workflow/rules/exp_type_1.smk:268-297
  * summarize_histogram_type2      workflow/rules/exp_type_2.smk:171-216

Implementation notes (why this is bit-identical although written differently):
  * the reference accumulates with the builtin `sum` over *every* bin; on CPython
    3.10 (its pinned interpreter) that is a plain left-to-right double accumulation.
    We accumulate explicitly in the same order and skip empty bins, which adds an
    exact 0.0 and therefore cannot change any partial sum.  Being explicit also keeps
    us independent of CPython >= 3.12, whose `sum` switched to compensated summation.
  * the four percentage metrics divide an *integer* bin total by the integer number of
    distinct k-mers, so the order in which bins are added is irrelevant there.
"""
from __future__ import annotations

from typing import Optional, Callable, Iterable, List, Sequence

WITHIN_HEADER = ("group_num,k,percent_1_occ,percent_25_or_less,percent_25_to_75,"
                 "percent_75_or_more,unique_stat,unique_stat_norm,delta_frac,delta_frac_norm\n")
ACROSS_HEADER = ("group_num,k,percent_1_occ,percent_2_to_5,percent_5_to_20,percent_20_more,"
                 "unique_stat,unique_stat_norm,delta_frac,delta_frac_norm\n")

_ASSERT_MSG = "Issue occurred with histogram summarization"


def _bin_range_total(hist: Sequence[int], lo: int, hi: int) -> int:
    """Exact integer total of hist[lo:hi] with Python `range` clipping semantics."""
    hi = min(hi, len(hist))
    total = 0
    for i in range(max(lo, 0), hi):
        total += hist[i]
    return total


def _weighted_occurrence(hist: Sequence[int], distinct: int, members: int | None,
                         start: int = 0) -> float:
    """sum_i w_i * (hist[i] / distinct), accumulated in index order.
    w_i = i+1 (members is None) or (i+1)/members."""
    acc = 0.0
    for i in range(start, len(hist)):
        n = hist[i]
        if n == 0:
            continue
        w = (i + 1) if members is None else (i + 1) / members
        acc += w * (n / distinct)
    return acc


def summarize_histogram_type1(hist_counts: Sequence[int], num_dataset_members: int,
                              across_group_analysis: bool, k: int) -> List[float]:
    """hist_counts[i] = number of distinct k-mers whose counter is i+1.

    Returns [%1occ, %<=25%, %25-75%, %>=75%, unique_stat, unique_stat_norm, |set|/k]
    (exp_type_1.smk:115-150).  Raises ZeroDivisionError on an empty histogram and
    AssertionError when the four percentages do not add up to 1 +- 0.05, exactly as the
    reference does."""
    distinct = 0
    for n in hist_counts:
        distinct += n

    if across_group_analysis:
        first_cut, second_cut = 5, 20                        # exp_type_1.smk:133-134
    else:
        first_cut = max(int(0.25 * num_dataset_members), 1)   # exp_type_1.smk:129-130
        second_cut = max(int(0.75 * num_dataset_members), 1)

    shares = [
        round(hist_counts[0] / distinct, 3),
        round(_bin_range_total(hist_counts, 1, first_cut) / distinct, 3),
        round(_bin_range_total(hist_counts, first_cut, second_cut) / distinct, 3),
        round(_bin_range_total(hist_counts, second_cut, len(hist_counts)) / distinct, 3),
    ]
    # the reference sums the four rounded shares with builtin sum (int 0 start)
    drift = abs(((shares[0] + shares[1]) + shares[2]) + shares[3] - 1)
    assert drift < 0.05, _ASSERT_MSG

    return shares + [
        round(_weighted_occurrence(hist_counts, distinct, None), 4),
        round(_weighted_occurrence(hist_counts, distinct, num_dataset_members), 4),
        round(distinct / k, 4),
    ]


def summarize_histogram_type2(sub_counts: Sequence[int], inter_counts: Sequence[int],
                              num_genomes_in_dataset: int, across_group_analysis: bool,
                              k: int) -> List[float]:
    """Pivot-vs-group variant (exp_type_2.smk:171-216): `sub_counts` is the histogram of
    pivot \\ group (all mass in bin 1), `inter_counts` that of pivot & group with
    -ocsum counters (bin 1 empty)."""
    assert inter_counts[0] == 0, "intersection counts should have 0 unique kmers"
    assert _bin_range_total(sub_counts, 1, len(sub_counts)) == 0, \
        "all of kmers in sub_counts should be unique"

    distinct = 0
    for n in sub_counts:
        distinct += n
    for n in inter_counts:
        distinct += n

    if across_group_analysis:
        first_cut, second_cut = 3, 8                          # exp_type_2.smk:197-198
    else:
        first_cut = max(int(0.25 * num_genomes_in_dataset), 1)
        second_cut = max(int(0.75 * num_genomes_in_dataset), 1)

    shares = [
        round(sub_counts[0] / distinct, 3),
        round(_bin_range_total(inter_counts, 1, first_cut) / distinct, 3),
        round(_bin_range_total(inter_counts, first_cut, second_cut) / distinct, 3),
        round(_bin_range_total(inter_counts, second_cut, len(inter_counts)) / distinct, 3),
    ]
    drift = abs(((shares[0] + shares[1]) + shares[2]) + shares[3] - 1)
    assert drift < 0.05, _ASSERT_MSG

    # exp_type_2.smk:206-212: the pivot-only term first, then the sum over bins >= 2.
    # `1 * a / b` parses as (1*a)/b and `(1/n) * a / b` as ((1/n)*a)/b.
    stat = (1 * sub_counts[0]) / distinct
    stat += _weighted_occurrence(inter_counts, distinct, None, start=1)
    stat_norm = ((1 / num_genomes_in_dataset) * sub_counts[0]) / distinct
    stat_norm += _weighted_occurrence(inter_counts, distinct, num_genomes_in_dataset, start=1)

    return shares + [round(stat, 4), round(stat_norm, 4), round(distinct / k, 4)]


# --------------------------------------------------------------------------- CSV files
def read_histogram_file(path: str) -> List[int]:
    """`kmc_tools transform ... histogram` text: one `count<TAB>n` line per counter
    value starting at 1; only the second column is consumed (exp_type_1.smk:210-212)."""
    with open(path, "r") as fh:
        return [int(rec.split()[1]) for rec in fh.readlines()]


def _rows_to_csv(header: str, rows: Iterable[Sequence]) -> str:
    return header + "".join(",".join(str(x) for x in row) + "\n" for row in rows)


def within_groups_csv(hist_paths: Sequence[str], num_datasets: int,
                      members_of: Callable[[str], int], hists: Optional[dict] = None) -> str:
    """CSV text of step_5/within_datasets_analysis.csv (exp_type_1.smk:199-231).

    `hist_paths` are `step_4/k_{k}/dataset_{num}/dataset_{num}_k{k}_hist.txt` in the
    order Snakemake's expand() yields them; k and the group number are parsed from the
    path exactly as the rule does; `members_of(num_str)` is the genome count."""
    rows: List[list] = []
    for path in hist_paths:
        parts = path.split("/")
        k = parts[1][2:]
        num = parts[2].split("_")[1]
        # `hists`: histograms the caller still holds in memory (the batched runner just wrote these
        # files from them); same list the file would parse to
        hist = hists[path] if hists and path in hists else read_histogram_file(path)
        rows.append([f"group_{num}", k]
                    + summarize_histogram_type1(hist, members_of(num), False, int(k)))
    for g in range(1, num_datasets + 1):
        label = f"group_{g}"
        peak = max(r[8] for r in rows if r[0] == label)
        for r in rows:
            if r[0] == label:
                r.append(round(r[8] / peak, 4))
    return _rows_to_csv(WITHIN_HEADER, rows)


def across_groups_csv(hist_paths: Sequence[str], num_datasets: int, hists: Optional[dict] = None) -> str:
    """CSV text of step_9/across_datasets_analysis.csv (exp_type_1.smk:268-297)."""
    rows: List[list] = []
    for path in hist_paths:
        k = path.split("/")[1][2:]
        hist = hists[path] if hists and path in hists else read_histogram_file(path)
        rows.append(["full_group", k]
                    + summarize_histogram_type1(hist, num_datasets, True, int(k)))
    peak = max(r[8] for r in rows)
    for r in rows:
        r.append(round(r[8] / peak, 4))
    return _rows_to_csv(ACROSS_HEADER, rows)


# --------------------------------------------------------------------------- experiment type 2
PIVOT_WITHIN_HEADER = ("group_num,k,percent_1_occ,percent_25_or_less,percent_25_to_75,percent_75_or_more,"
                       "unique_stat,unique_stat_norm,delta_frac,delta_frac_norm\n")
PIVOT_ACROSS_HEADER = ("group_num,k,percent_1_occ,percent_2_to_3,percent_4_to_8,percent_9_more,"
                       "unique_stat,unique_stat_norm,delta_frac,delta_frac_norm\n")


def _pivot_rows(hist_paths: Sequence[str], num_datasets: int, members_of, across: bool) -> List[list]:
    """`hist_paths` alternate subtract / intersect histograms in the order of
    exp_type_2.smk:153-169 (dataset-major, then k); the group number and k are parsed from the
    path as the rules do (:408-409, :527-528); the last column divides by the group's maximum."""
    rows: List[list] = []
    for i in range(0, len(hist_paths), 2):
        parts = hist_paths[i].split("/")
        num = parts[2].split("_")[1]
        k = parts[1].split("_")[1]
        sub = read_histogram_file(hist_paths[i])
        inter = read_histogram_file(hist_paths[i + 1])
        rows.append([f"group_{num}", k] + summarize_histogram_type2(sub, inter, members_of(num), across, int(k)))
    for g in range(1, num_datasets + 1):
        label = f"group_{g}"
        peak = max(r[8] for r in rows if r[0] == label)
        for r in rows:
            if r[0] == label:
                r.append(round(r[8] / peak, 4))
    return rows


def pivot_within_groups_csv(hist_paths: Sequence[str], num_datasets: int,
                            members_of: Callable[[str], int]) -> str:
    """within_dataset_analysis_type_2/within_dataset_analysis.csv (exp_type_2.smk:404-438)."""
    return _rows_to_csv(PIVOT_WITHIN_HEADER, _pivot_rows(hist_paths, num_datasets, members_of, False))


def pivot_across_groups_csv(hist_paths: Sequence[str], num_datasets: int) -> str:
    """across_dataset_analysis_type_2/across_dataset_analysis.csv (exp_type_2.smk:521-554):
    the member count passed to the summariser is num_datasets (:536)."""
    return _rows_to_csv(PIVOT_ACROSS_HEADER, _pivot_rows(hist_paths, num_datasets, lambda _n: num_datasets, True))
