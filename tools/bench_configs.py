#!/usr/bin/env python3
"""Throughput of the single-GPU BASELINE.json configs next to bench.py's headline:

  cfg2  5 species x 5 genomes x 5 Mbp, k = 31: K1 build per genome (one call each) and batched
        (one call for all 25), K4 histogram (the CPU baseline lives in bench.py alone)
  cfg3  10 species x 10 genomes x 5 Mbp, k in {15, 21, 27, 31, 41}: fused within-group occurrence
        (steps 1-4 of exp_type_1.smk, `kh_exp1_run` without the across-group step)
  cfg4  20 species x 5 genomes x 5 Mbp, k = 31, steps 1-8 on ONE GPU (the config's 8-way sharding is the
        driver's to measure)
  cfg5  the shape of configs[4] ("all available genomes": groups far above the 64-genome mask): 5 species with
        10 / 70 / 200 / 70 / 10 genomes of 1 Mbp, k = 31 and k = 63, steps 1-8

Inputs are resident in HBM; every figure is the median of `--reps` timed repetitions after one
warm-up.  One JSON object on stdout (committed as profiles/rNN_configs.json).
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps, sync):
    fn()
    sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        sync()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--length", type=int, default=5_000_000)
    a = ap.parse_args()
    import torch
    from khoice_amd import engine as E
    from khoice_amd import synth
    torch.cuda.init()
    eng = E.Engine(0)
    out = {"genome_bp": a.length, "reps": a.reps}

    def resident(items):
        dev = [torch.from_numpy(np.frombuffer(t, dtype=np.uint8).copy()).cuda() for _, _, t in items]
        return dev, [(d.data_ptr(), d.numel()) for d in dev]

    # ---------------------------------------------------------------- cfg2
    items = synth.species_set(5, 5, a.length)
    dev, seqs = resident(items)
    t_batch, sets = timed(lambda: eng.build_batch(seqs, 31, ci=1, with_counts=True), a.reps, eng.sync)
    distinct = sum(len(s) for s in sets)
    t_single, _ = timed(lambda: [eng.build_batch([s], 31, ci=1, with_counts=True) for s in seqs], a.reps, eng.sync)
    u = eng.union_sum([s.set_counts(1) for s in sets[:5]], 5000)
    t_hist, _ = timed(lambda: u.histogram(5001), a.reps, eng.sync)
    cfg2 = {"workload": f"5 x 5 x {a.length} bp, k=31", "distinct_kmers": distinct,
            "k1_batched_ms": round(1e3 * t_batch, 3), "k1_batched_distinct_per_s": round(distinct / t_batch, 1),
            "k1_per_genome_ms_total": round(1e3 * t_single, 3),
            "k1_per_genome_distinct_per_s": round(distinct / t_single, 1),
            "k4_histogram_ms": round(1e3 * t_hist, 4), "k4_counters": len(u)}
    out["cfg2"] = cfg2
    del sets, u, dev
    eng.trim()

    # ---------------------------------------------------------------- cfg3
    items = synth.species_set(10, 10, a.length)
    dev, seqs = resident(items)
    group_of = [s - 1 for s, _, _ in items]
    rows = []
    for k in (15, 21, 27, 31, 41):
        r0 = eng.stats()["retries"]
        t, res = timed(lambda: eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=5001, across=False), max(2, a.reps // 2), eng.sync)
        d = int(res["distinct_per_seq"].sum())
        rows.append({"k": k, "ms": round(1e3 * t, 3), "replans_per_run": (eng.stats()["retries"] - r0) / (1 + max(2, a.reps // 2)),
                     "distinct_kmers": d, "distinct_per_s": round(d / t, 1),
                     "bases_per_s": round(sum(n for _, n in seqs) / t, 1)})
    out["cfg3"] = {"workload": f"10 x 10 x {a.length} bp, steps 1-4 (within-group occurrence)", "per_k": rows}
    del dev
    eng.trim()

    # ---------------------------------------------------------------- cfg4 (one GPU)
    items = synth.species_set(20, 5, a.length)
    dev, seqs = resident(items)
    group_of = [s - 1 for s, _, _ in items]
    r0 = eng.stats()["retries"]
    t, res = timed(lambda: eng.exp1_run(seqs, group_of, 31, cs=5000, hist_len=5001), max(2, a.reps // 2), eng.sync)
    d = int(res["distinct_per_seq"].sum())
    out["cfg4_one_gpu"] = {"workload": f"20 x 5 x {a.length} bp, k=31, steps 1-8", "ms": round(1e3 * t, 3), "distinct_kmers": d,
                           "distinct_per_s": round(d / t, 1), "replans_per_run": (eng.stats()["retries"] - r0) / (1 + max(2, a.reps // 2))}
    del dev
    eng.trim()

    # ---------------------------------------------------------------- cfg5 shape: groups above the 64-genome mask
    sizes = [10, 70, 200, 70, 10]
    length5 = min(a.length, 1_000_000)
    texts, group_of = [], []
    for g, n in enumerate(sizes):
        anc = synth.ancestor(g + 1, length5)
        for j in range(n):
            texts.append(synth.clean_text(synth.genome_records(g + 1, j, length5, anc)))
            group_of.append(g)
    dev = [torch.from_numpy(np.frombuffer(t, dtype=np.uint8).copy()).cuda() for t in texts]
    seqs = [(x.data_ptr(), x.numel()) for x in dev]
    rows = []
    for k in (31, 63):
        r0 = eng.stats()["retries"]
        eng.profile(True)
        eng.stats_reset()
        t, res = timed(lambda: eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=5001), 2, eng.sync)
        st = eng.stats()
        eng.profile(False)
        d = int(res["distinct_per_seq"].sum())
        rows.append({"k": k, "ms": round(1e3 * t, 3), "distinct_kmers": d, "distinct_per_s": round(d / t, 1),
                     "bases_per_s": round(sum(n for _, n in seqs) / t, 1), "replans_per_run": st["retries"] / 3,
                     "kernel_launches_per_run": {n: v["launches"] // 3 for n, v in st["kernels"].items() if v["launches"]}})
    out["cfg5_shape"] = {"workload": f"groups of {sizes} genomes x {length5} bp, steps 1-8", "genomes": len(texts), "per_k": rows}
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    main()
