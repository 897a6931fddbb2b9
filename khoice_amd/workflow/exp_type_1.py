"""Experiment type 1 of khoice, executed without Snakemake.

This mirrors workflow/rules/exp_type_1.smk rule by rule: the same directory layout, the same
`complex` operation files, and — most importantly — the same shell strings, so that whatever
`kmc` / `kmc_tools` are first on PATH get exactly the argv the reference gives KMC.  With
khoice_amd's bin/ on PATH the whole k-mer side runs on the MI355X.

    run(work_root, k_values=[...], num_datasets=N)    rule-per-process, like `snakemake --cores 1`
    run_batched(...)                                  same files, one resident engine, no
                                                      process launches (SURVEY §8f "next" #1)

The parse-time section (exp_type_1.smk:26-84) is `prepare()`; rules are the functions named
after them.  Only the CSV stage computes anything in Python (khoice_amd.summarize).
"""
from __future__ import annotations

import os
import shutil
import subprocess
from typing import Dict, List, Optional, Sequence

from .. import summarize

REPO_BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "bin")


def genomes_of(work_root: str, num: int) -> List[str]:
    """Base names of data/dataset_{num}/*.fna.gz in os.listdir order (exp_type_1.smk:44-47)."""
    out = []
    for name in os.listdir(os.path.join(work_root, "data", f"dataset_{num}")):
        if name.endswith(".fna.gz"):
            out.append(name.split(".fna.gz")[0])
    return out


def _ops_text(inputs: Sequence[str], output: str) -> str:
    """One `kmc_tools complex` definition: union of all inputs, counters saturating at 5000.
    Byte-compatible with what exp_type_1.smk:52-61 writes (note the blank before ')')."""
    lines = ["INPUT:"]
    lines += [f"set{i} = {path}" for i, path in enumerate(inputs, 1)]
    expr = "(" + " + ".join(f"set{i}" for i in range(1, len(inputs) + 1)) + " )"
    lines += ["OUTPUT:", f"{output} = {expr}", "OUTPUT_PARAMS:", "-cs5000"]
    return "\n".join(lines) + "\n"


def prepare(work_root: str, k_values: Sequence[str], num_datasets: int) -> None:
    """exp_type_1.smk:26-84: tmp/, complex_ops/{within_groups,across_groups}/..."""
    os.makedirs(os.path.join(work_root, "tmp"), exist_ok=True)
    for k in k_values:
        for num in range(1, num_datasets + 1):
            d = os.path.join(work_root, f"complex_ops/within_groups/k_{k}/dataset_{num}")
            os.makedirs(d, exist_ok=True)
            ins = [f"step_2/k_{k}/dataset_{num}/{g}.transformed" for g in genomes_of(work_root, num)]
            with open(os.path.join(d, f"within_dataset_{num}.txt"), "w") as fd:
                fd.write(_ops_text(ins, f"step_3/k_{k}/dataset_{num}/dataset_{num}.transformed.combined"))
        d = os.path.join(work_root, f"complex_ops/across_groups/k_{k}")
        os.makedirs(d, exist_ok=True)
        ins = [f"step_6/k_{k}/dataset_{i}/dataset_{i}.transformed.combined.transformed"
               for i in range(1, num_datasets + 1)]
        with open(os.path.join(d, "across_all_datasets.txt"), "w") as fd:
            fd.write(_ops_text(ins, f"step_7/k_{k}/all_datasets.transformed.combined.transformed.combined"))


class _Shell:
    """Runs rule shell strings the way Snakemake does: bash strict mode, cwd = WORK_ROOT."""

    def __init__(self, work_root: str, bin_dir: Optional[str]):
        self.cwd = work_root
        self.env = dict(os.environ)
        if bin_dir:
            self.env["PATH"] = bin_dir + os.pathsep + self.env.get("PATH", "")
        self.launched = 0

    def __call__(self, cmd: str, outputs: Sequence[str]):
        for o in outputs:   # Snakemake creates the parent directories of declared outputs
            os.makedirs(os.path.dirname(os.path.join(self.cwd, o)) or ".", exist_ok=True)
        self.launched += 1
        r = subprocess.run(["bash", "-c", "set -euo pipefail; " + cmd], cwd=self.cwd, env=self.env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            for o in outputs:   # a failed rule leaves no outputs behind
                try:
                    os.remove(os.path.join(self.cwd, o))
                except OSError:
                    pass
            raise RuntimeError(f"rule failed ({r.returncode}): {cmd}\n{r.stderr}")


# --- rules (names as in exp_type_1.smk:156-259) ---------------------------------------------
def build_kmc_database_on_genome(sh, k, num, genome):
    pre = f"step_1/k_{k}/dataset_{num}/{genome}"
    sh(f"kmc -fm -m64 -k{k} -ci1 data/dataset_{num}/{genome}.fna.gz {pre} tmp/",
       [pre + ".kmc_pre", pre + ".kmc_suf"])


def transform_genome_to_set(sh, k, num, genome):
    out = f"step_2/k_{k}/dataset_{num}/{genome}.transformed"
    sh(f"kmc_tools transform step_1/k_{k}/dataset_{num}/{genome} set_counts 1 {out}",
       [out + ".kmc_pre", out + ".kmc_suf"])


def within_group_union(sh, k, num):
    out = f"step_3/k_{k}/dataset_{num}/dataset_{num}.transformed.combined"
    sh(f"kmc_tools complex complex_ops/within_groups/k_{k}/dataset_{num}/within_dataset_{num}.txt",
       [out + ".kmc_pre", out + ".kmc_suf"])


def within_group_union_histogram(sh, k, num):
    out = f"step_4/k_{k}/dataset_{num}/dataset_{num}_k{k}_hist.txt"
    sh(f"kmc_tools transform step_3/k_{k}/dataset_{num}/dataset_{num}.transformed.combined histogram {out}", [out])


def build_group_kmer_set(sh, k, num):
    out = f"step_6/k_{k}/dataset_{num}/dataset_{num}.transformed.combined.transformed"
    sh(f"kmc_tools transform step_3/k_{k}/dataset_{num}/dataset_{num}.transformed.combined set_counts 1 {out}",
       [out + ".kmc_pre", out + ".kmc_suf"])


def across_group_union(sh, k):
    out = f"step_7/k_{k}/all_datasets.transformed.combined.transformed.combined"
    sh(f"kmc_tools complex complex_ops/across_groups/k_{k}/across_all_datasets.txt",
       [out + ".kmc_pre", out + ".kmc_suf"])


def across_group_union_histogram(sh, k):
    out = f"step_8/k_{k}/all_datasets_k{k}_hist.txt"
    sh(f"kmc_tools transform step_7/k_{k}/all_datasets.transformed.combined.transformed.combined histogram {out}",
       [out])


def _csv_stage(work_root: str, k_values: Sequence[str], num_datasets: int, hists: Optional[dict] = None) -> Dict[str, str]:
    """Rules within_group_union_analysis, across_group_union_analysis and
    copy_final_results_type1 (exp_type_1.smk:193-231, 262-308)."""
    cwd = os.getcwd()
    os.chdir(work_root)
    try:
        # expand(): the product iterates k_len outermost, num innermost (exp_type_1.smk:195)
        within_inputs = [f"step_4/k_{k}/dataset_{num}/dataset_{num}_k{k}_hist.txt"
                         for k in k_values for num in range(1, num_datasets + 1)]
        across_inputs = [f"step_8/k_{k}/all_datasets_k{k}_hist.txt" for k in k_values]
        members = {str(n): len(genomes_of(".", n)) for n in range(1, num_datasets + 1)}
        within = summarize.within_groups_csv(within_inputs, num_datasets, lambda n: members[str(n)], hists)
        across = summarize.across_groups_csv(across_inputs, num_datasets, hists)
        os.makedirs("step_5", exist_ok=True)
        os.makedirs("step_9", exist_ok=True)
        os.makedirs("final_results_type1", exist_ok=True)
        with open("step_5/within_datasets_analysis.csv", "w") as fh:
            fh.write(within)
        with open("step_9/across_datasets_analysis.csv", "w") as fh:
            fh.write(across)
        shutil.copyfile("step_5/within_datasets_analysis.csv", "final_results_type1/within_datasets_analysis.csv")
        shutil.copyfile("step_9/across_datasets_analysis.csv", "final_results_type1/across_datasets_analysis.csv")
    finally:
        os.chdir(cwd)
    return {"within": within, "across": across}


def run(work_root: str, k_values: Sequence, num_datasets: int, bin_dir: Optional[str] = REPO_BIN):
    """Target final_results_type1/*.csv, one process per rule instance (like the reference)."""
    k_values = [str(k) for k in k_values]
    prepare(work_root, k_values, num_datasets)
    sh = _Shell(work_root, bin_dir)
    for k in k_values:
        for num in range(1, num_datasets + 1):
            for g in genomes_of(work_root, num):
                build_kmc_database_on_genome(sh, k, num, g)
                transform_genome_to_set(sh, k, num, g)
            within_group_union(sh, k, num)
            within_group_union_histogram(sh, k, num)
            build_group_kmer_set(sh, k, num)
        across_group_union(sh, k)
        across_group_union_histogram(sh, k)
    out = _csv_stage(work_root, k_values, num_datasets)
    out["processes"] = sh.launched
    return out


def run_batched(work_root: str, k_values: Sequence, num_datasets: int, device: int = 0,
                keep_databases: bool = True, timings: Optional[dict] = None):
    """Same outputs as run(), produced by ONE resident engine: no process launches, no HIP
    re-initialisation, one batched build per k.  Intermediate databases (step_1..step_7) are
    still written when keep_databases is set, so a later Snakemake run finds them."""
    import time
    from .. import engine as E
    k_values = [str(k) for k in k_values]
    prepare(work_root, k_values, num_datasets)
    eng = E.Engine(device)
    t_start = time.perf_counter()
    try:
        genomes = {num: genomes_of(work_root, num) for num in range(1, num_datasets + 1)}
        group_of, names = [], []
        for num in range(1, num_datasets + 1):
            for g in genomes[num]:
                group_of.append(num - 1)
                names.append((num, g))
        if not keep_databases:
            # No intermediate database is wanted: the files are inflated on host threads into pinned
            # memory and cleaned on the device as they complete (kh_ingest_fasta), every k is ONE
            # fused kh_exp1_run on the resident texts, the step_4 / step_8 histogram files are written
            # from the returned arrays and the CSV stage consumes those arrays (not a re-parse of six
            # 65535-line files it has just written).
            import numpy as np
            paths = [os.path.join(work_root, f"data/dataset_{num}/{g}.fna.gz") for num, g in names]
            texts = eng.ingest_fasta(paths)
            t_ingest = time.perf_counter()
            hists = {}
            for k in k_values:
                res = eng.exp1_run(texts.seqs, group_of, int(k), cs=5000, hist_len=5001)
                def as_list(h):
                    # what the histogram file parses to, cut behind its last non-zero line (never
                    # shorter than KMC's 255 lines): the summariser only sums, trailing zeros add nothing
                    nz = np.flatnonzero(h[1:])
                    return h[1:max(256, int(nz[-1]) + 2 if nz.size else 0)].tolist()
                for num in range(1, num_datasets + 1):
                    os.makedirs(os.path.join(work_root, f"step_4/k_{k}/dataset_{num}"), exist_ok=True)
                    rel = f"step_4/k_{k}/dataset_{num}/dataset_{num}_k{k}_hist.txt"
                    eng.write_histogram_text(os.path.join(work_root, rel), res["within_hist"][num - 1], 65535)
                    hists[rel] = as_list(res["within_hist"][num - 1])
                os.makedirs(os.path.join(work_root, f"step_8/k_{k}"), exist_ok=True)
                rel = f"step_8/k_{k}/all_datasets_k{k}_hist.txt"
                eng.write_histogram_text(os.path.join(work_root, rel), res["across_hist"], 65535)
                hists[rel] = as_list(res["across_hist"])
            bases = texts.total_bases()
            texts.free()
            t_device = time.perf_counter()
            eng.close()
            out = _csv_stage(work_root, k_values, num_datasets, hists)
            out["processes"] = 0
            if timings is not None:
                timings.update(ingest_s=t_ingest - t_start, device_and_files_s=t_device - t_ingest,
                               csv_s=time.perf_counter() - t_device, bases=bases)
            return out
        # host ingest (inflate + FASTA parsing) runs in the library without the GIL: one thread
        # per file up to the core count
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as pool:
            texts = list(pool.map(
                lambda ng: eng.read_fasta(os.path.join(work_root, f"data/dataset_{ng[0]}/{ng[1]}.fna.gz")), names))
        t_ingest = time.perf_counter()
        for k in k_values:
            ki = int(k)
            counted = eng.build_batch(texts, ki, ci=1, with_counts=True)
            plain = [s.set_counts(1) for s in counted]
            if keep_databases:
                for (num, g), c, p in zip(names, counted, plain):
                    for d in (f"step_1/k_{k}/dataset_{num}", f"step_2/k_{k}/dataset_{num}"):
                        os.makedirs(os.path.join(work_root, d), exist_ok=True)
                    c.save(os.path.join(work_root, f"step_1/k_{k}/dataset_{num}/{g}"))
                    p.save(os.path.join(work_root, f"step_2/k_{k}/dataset_{num}/{g}.transformed"))
            group_sets = []
            for num in range(1, num_datasets + 1):
                members = [p for p, (n, _) in zip(plain, names) if n == num]
                union = eng.union_sum(members, 5000)
                for d in (f"step_3/k_{k}/dataset_{num}", f"step_4/k_{k}/dataset_{num}", f"step_6/k_{k}/dataset_{num}"):
                    os.makedirs(os.path.join(work_root, d), exist_ok=True)
                union.histogram_file(65535, os.path.join(work_root, f"step_4/k_{k}/dataset_{num}/dataset_{num}_k{k}_hist.txt"))
                gs = union.set_counts(1)
                group_sets.append(gs)
                if keep_databases:
                    union.save(os.path.join(work_root, f"step_3/k_{k}/dataset_{num}/dataset_{num}.transformed.combined"))
                    gs.save(os.path.join(work_root, f"step_6/k_{k}/dataset_{num}/dataset_{num}.transformed.combined.transformed"))
            across = eng.union_sum(group_sets, 5000)
            for d in (f"step_7/k_{k}", f"step_8/k_{k}"):
                os.makedirs(os.path.join(work_root, d), exist_ok=True)
            across.histogram_file(65535, os.path.join(work_root, f"step_8/k_{k}/all_datasets_k{k}_hist.txt"))
            if keep_databases:
                across.save(os.path.join(work_root, f"step_7/k_{k}/all_datasets.transformed.combined.transformed.combined"))
        t_device = time.perf_counter()
    finally:
        eng.close()
    out = _csv_stage(work_root, k_values, num_datasets)
    out["processes"] = 0
    if timings is not None:
        timings.update(ingest_s=t_ingest - t_start, device_and_files_s=t_device - t_ingest,
                       csv_s=time.perf_counter() - t_device, bases=sum(len(t) for t in texts))
    return out
