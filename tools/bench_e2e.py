#!/usr/bin/env python3
"""End-to-end rate of experiment type 1 (gz FASTA on disk -> step_5/step_9 CSVs), reported next to
bench.py's device-resident figure as SURVEY.md §8d's timing protocol asks.  Not bench.py's
`value`: ingest (inflate + FASTA parsing on the host cores) and the H2D copy are inside.

    python tools/bench_e2e.py [--species 5 --genomes 5 --length 5000000 --k 31]
"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--species", type=int, default=5)
    ap.add_argument("--genomes", type=int, default=5)
    ap.add_argument("--length", type=int, default=5_000_000)
    ap.add_argument("--k", type=int, nargs="+", default=[31])
    ap.add_argument("--keep-databases", action="store_true")
    a = ap.parse_args()
    from khoice_amd import synth
    from khoice_amd.workflow import exp_type_1 as W
    with tempfile.TemporaryDirectory() as root:
        t0 = time.perf_counter()
        synth.write_dataset_tree(root, a.species, a.genomes, a.length)
        gen = time.perf_counter() - t0
        gz = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(os.path.join(root, "data")) for f in fs)
        W.run_batched(root, a.k[:1], a.species, keep_databases=False)      # warm: HIP start-up, page cache
        tm = {}
        t0 = time.perf_counter()
        W.run_batched(root, a.k, a.species, keep_databases=a.keep_databases, timings=tm)
        wall = time.perf_counter() - t0
    print(json.dumps({
        "workload": f"exp_type_1 end to end: {a.species} x {a.genomes} x {a.length} bp, k={a.k}, gz on disk -> CSV",
        "gz_bytes": gz, "bases": tm["bases"], "wall_s": round(wall, 3),
        "ingest_s": round(tm["ingest_s"], 3), "device_and_files_s": round(tm["device_and_files_s"], 3),
        "csv_s": round(tm["csv_s"], 3), "bases_per_s_per_k": round(tm["bases"] * len(a.k) / wall, 1),
        "host_cores": os.cpu_count(), "keep_databases": a.keep_databases, "generate_s": round(gen, 1)}))


if __name__ == "__main__":
    main()
