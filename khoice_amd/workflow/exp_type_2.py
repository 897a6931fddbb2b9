"""Experiment type 2 of khoice (pivot genome vs its group / vs the other groups), executed
without Snakemake.  Mirrors workflow/rules/exp_type_2.smk rule by rule (same directories,
`complex` files and shell strings; the intersect / kmers_subtract call sites of SURVEY §8 row a11)
and offers the batched form on one resident engine.

    run(work_root, k_values, num_datasets)        rule-per-process through kmc / kmc_tools on PATH
    run_batched(...)                              same histogram files and CSVs, no process launches

Inputs: input_type_2/{rest_of_set/dataset_N/*.fna.gz, pivot/dataset_N/pivot_N.fna.gz}
(staged out of DATABASE_ROOT at exp_type_2.smk:31-48 — data management, not k-mer work).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

from .. import summarize
from .exp_type_1 import REPO_BIN, _ops_text, _Shell


def rest_of_set(work_root: str, num: int) -> List[str]:
    d = os.path.join(work_root, f"input_type_2/rest_of_set/dataset_{num}")
    return [n.split(".fna.gz")[0] for n in os.listdir(d) if n.endswith(".fna.gz")]


def prepare(work_root: str, k_values: Sequence[str], num_datasets: int) -> None:
    """exp_type_2.smk:27-29,50-115: tmp/ and the `complex` operation files.  The across-group
    union of pivot p sums the group sets of every dataset but p's own (:99-101)."""
    os.makedirs(os.path.join(work_root, "tmp"), exist_ok=True)
    for k in k_values:
        for num in range(1, num_datasets + 1):
            d = os.path.join(work_root, f"complex_ops_type_2/within_groups/k_{k}/dataset_{num}")
            os.makedirs(d, exist_ok=True)
            ins = [f"genome_sets_type_2/rest_of_set/k_{k}/dataset_{num}/{g}.transformed" for g in rest_of_set(work_root, num)]
            with open(os.path.join(d, f"within_dataset_{num}.txt"), "w") as fd:
                fd.write(_ops_text(ins, f"within_databases_type_2/rest_of_set/k_{k}/dataset_{num}/dataset_{num}.transformed.combined"))
        for p in range(1, num_datasets + 1):
            d = os.path.join(work_root, f"complex_ops_type_2/across_groups/k_{k}/pivot_{p}")
            os.makedirs(d, exist_ok=True)
            ins = [f"within_databases_type_2/rest_of_set/k_{k}/dataset_{i}/dataset_{i}.transformed.combined.transformed"
                   for i in range(1, num_datasets + 1) if i != p]
            with open(os.path.join(d, f"across_datasets_pivot_{p}.txt"), "w") as fd:
                fd.write(_ops_text(ins, f"across_databases_type_2/k_{k}/pivot_{p}/all_datasets_pivot_{p}.transformed.combined.transformed.combined"))


# --- rules (exp_type_2.smk:289-507) -----------------------------------------------------------
def build_kmc_database_on_genome(sh, k, num, genome):
    pre = f"step_1_type_2/rest_of_set/k_{k}/dataset_{num}/{genome}"
    sh(f"kmc -fm -m64 -k{k} -ci1 input_type_2/rest_of_set/dataset_{num}/{genome}.fna.gz {pre} tmp/",
       [pre + ".kmc_pre", pre + ".kmc_suf"])


def build_kmc_database_on_pivot(sh, k, num):
    pre = f"step_1_type_2/pivot/k_{k}/dataset_{num}/pivot_{num}"
    sh(f"kmc -fm -m64 -k{k} -ci1 input_type_2/pivot/dataset_{num}/pivot_{num}.fna.gz {pre} tmp/",
       [pre + ".kmc_pre", pre + ".kmc_suf"])


def transform_genome_to_set(sh, k, num, genome):
    out = f"genome_sets_type_2/rest_of_set/k_{k}/dataset_{num}/{genome}.transformed"
    sh(f"kmc_tools transform step_1_type_2/rest_of_set/k_{k}/dataset_{num}/{genome} set_counts 1 {out}",
       [out + ".kmc_pre", out + ".kmc_suf"])


def transform_pivot_to_set(sh, k, num):
    out = f"genome_sets_type_2/pivot/k_{k}/dataset_{num}/pivot_{num}.transformed"
    sh(f"kmc_tools transform step_1_type_2/pivot/k_{k}/dataset_{num}/pivot_{num} set_counts 1 {out}",
       [out + ".kmc_pre", out + ".kmc_suf"])


def within_group_union(sh, k, num):
    out = f"within_databases_type_2/rest_of_set/k_{k}/dataset_{num}/dataset_{num}.transformed.combined"
    sh(f"kmc_tools complex complex_ops_type_2/within_groups/k_{k}/dataset_{num}/within_dataset_{num}.txt",
       [out + ".kmc_pre", out + ".kmc_suf"])


def _pivot_op(sh, k, num, scope, other, op):
    """`kmc_tools simple PIVOT OTHER intersect OUT -ocsum` / `... kmers_subtract OUT`
    (exp_type_2.smk:361-365,375-379,477-481,491-495)."""
    pivot = f"genome_sets_type_2/pivot/k_{k}/dataset_{num}/pivot_{num}.transformed"
    out = f"{scope}_dataset_results_type_2/k_{k}/dataset_{num}/{op}/dataset_{num}_pivot_{op}_group"
    verb = "intersect" if op == "intersect" else "kmers_subtract"
    tail = " -ocsum" if op == "intersect" else ""
    sh(f"kmc_tools simple {pivot} {other} {verb} {out}{tail}", [out + ".kmc_pre", out + ".kmc_suf"])
    sh(f"kmc_tools transform {out} histogram {out}.hist.txt", [out + ".hist.txt"])


def transform_rest_of_set_to_single_counts(sh, k, num):
    src = f"within_databases_type_2/rest_of_set/k_{k}/dataset_{num}/dataset_{num}.transformed.combined"
    sh(f"kmc_tools transform {src} set_counts 1 {src}.transformed", [src + ".transformed.kmc_pre", src + ".transformed.kmc_suf"])


def across_group_union_for_pivot(sh, k, num):
    out = f"across_databases_type_2/k_{k}/pivot_{num}/all_datasets_pivot_{num}.transformed.combined.transformed.combined"
    sh(f"kmc_tools complex complex_ops_type_2/across_groups/k_{k}/pivot_{num}/across_datasets_pivot_{num}.txt",
       [out + ".kmc_pre", out + ".kmc_suf"])


def _hist_lines(counter_max: int) -> int:
    """Lines `kmc_tools transform histogram` prints for a database: 2^(8 * counter bytes) - 1
    (same rule as bin/kmc_tools, kh_cli.cpp hist_lines)."""
    for lim in (0xFF, 0xFFFF, 0xFFFFFF):
        if counter_max <= lim:
            return lim
    return 0xFFFFFFFE


def _hist_paths(scope: str, k_values: Sequence[str], num_datasets: int) -> List[str]:
    """exp_type_2.smk:153-169: dataset-major, then k, then subtract before intersect."""
    return [f"{scope}_dataset_results_type_2/k_{k}/dataset_{num}/{op}/dataset_{num}_pivot_{op}_group.hist.txt"
            for num in range(1, num_datasets + 1) for k in k_values for op in ("subtract", "intersect")]


def _csv_stage(work_root: str, k_values: Sequence[str], num_datasets: int) -> Dict[str, str]:
    cwd = os.getcwd()
    os.chdir(work_root)
    try:
        within = summarize.pivot_within_groups_csv(_hist_paths("within", k_values, num_datasets), num_datasets,
                                                   lambda n: len(rest_of_set(".", int(n))))
        across = summarize.pivot_across_groups_csv(_hist_paths("across", k_values, num_datasets), num_datasets)
        for d, name, text in (("within_dataset_analysis_type_2", "within_dataset_analysis.csv", within),
                              ("across_dataset_analysis_type_2", "across_dataset_analysis.csv", across)):
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, name), "w") as fh:
                fh.write(text)
    finally:
        os.chdir(cwd)
    return {"within": within, "across": across}


def run(work_root: str, k_values: Sequence, num_datasets: int, bin_dir: Optional[str] = REPO_BIN):
    k_values = [str(k) for k in k_values]
    prepare(work_root, k_values, num_datasets)
    sh = _Shell(work_root, bin_dir)
    for k in k_values:
        for num in range(1, num_datasets + 1):
            for g in rest_of_set(work_root, num):
                build_kmc_database_on_genome(sh, k, num, g)
                transform_genome_to_set(sh, k, num, g)
            build_kmc_database_on_pivot(sh, k, num)
            transform_pivot_to_set(sh, k, num)
            within_group_union(sh, k, num)
            union = f"within_databases_type_2/rest_of_set/k_{k}/dataset_{num}/dataset_{num}.transformed.combined"
            for op in ("intersect", "subtract"):
                _pivot_op(sh, k, num, "within", union, op)
            transform_rest_of_set_to_single_counts(sh, k, num)
        for num in range(1, num_datasets + 1):
            across_group_union_for_pivot(sh, k, num)
            other = f"across_databases_type_2/k_{k}/pivot_{num}/all_datasets_pivot_{num}.transformed.combined.transformed.combined"
            for op in ("intersect", "subtract"):
                _pivot_op(sh, k, num, "across", other, op)
    out = _csv_stage(work_root, k_values, num_datasets)
    out["processes"] = sh.launched
    return out


def run_batched(work_root: str, k_values: Sequence, num_datasets: int, device: int = 0):
    """Same *.hist.txt files and CSVs from one resident engine: one batched build per k, one
    union per dataset, one union per pivot over the other datasets' group sets, and the
    intersect (-ocsum) / kmers_subtract pairs, all on sets that never leave HBM."""
    from concurrent.futures import ThreadPoolExecutor
    from .. import engine as E
    k_values = [str(k) for k in k_values]
    prepare(work_root, k_values, num_datasets)
    with E.Engine(device) as eng:
        paths, owner = [], []
        for num in range(1, num_datasets + 1):
            for g in rest_of_set(work_root, num):
                paths.append(os.path.join(work_root, f"input_type_2/rest_of_set/dataset_{num}/{g}.fna.gz"))
                owner.append(num - 1)
        pivot_paths = [os.path.join(work_root, f"input_type_2/pivot/dataset_{num}/pivot_{num}.fna.gz")
                       for num in range(1, num_datasets + 1)]
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as pool:
            texts = list(pool.map(eng.read_fasta, paths + pivot_paths))
        for k in k_values:
            ki = int(k)
            plain = eng.build_batch(texts, ki, ci=1, with_counts=False)
            genomes, pivots = plain[:len(paths)], plain[len(paths):]
            unions = [eng.union_sum([s for s, o in zip(genomes, owner) if o == num], 5000)
                      for num in range(num_datasets)]
            group_sets = [u.set_counts(1) for u in unions]
            for num in range(num_datasets):
                others = [group_sets[i] for i in range(num_datasets) if i != num]
                scopes = [("within", unions[num])]
                if others:
                    scopes.append(("across", eng.union_sum(others, 5000)))
                for scope, other in scopes:
                    for op in ("intersect", "subtract"):
                        res = (eng.intersect(pivots[num], other, "sum") if op == "intersect"
                               else eng.kmers_subtract(pivots[num], other))
                        d = os.path.join(work_root, f"{scope}_dataset_results_type_2/k_{k}/dataset_{num + 1}/{op}")
                        os.makedirs(d, exist_ok=True)
                        res.histogram_file(_hist_lines(res.counter_max()), os.path.join(d, f"dataset_{num + 1}_pivot_{op}_group.hist.txt"))
    out = _csv_stage(work_root, k_values, num_datasets)
    out["processes"] = 0
    return out
