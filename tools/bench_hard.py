#!/usr/bin/env python3
"""The fused step on "hard" genomes (khoice_amd.synth.hard_species_set: GC 70 %, 50 copies of an insertion sequence, seven
rRNA-like operons, tandem repeats, homopolymer runs): time, which form ran, re-plans — next to the i.i.d. set of the
same shape.  python tools/bench_hard.py [--species 5 --genomes 5 --length 5000000] -> one JSON object."""
import argparse, json, os, statistics, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--species", type=int, default=5)
    ap.add_argument("--genomes", type=int, default=5)
    ap.add_argument("--length", type=int, default=5_000_000)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch
    from khoice_amd import engine as E
    from khoice_amd import synth
    torch.cuda.init()
    eng = E.Engine(0)
    out = {"shape": f"{a.species} x {a.genomes} x {a.length} bp", "rows": []}
    for name, items in (("iid", synth.species_set(a.species, a.genomes, a.length)),
                        ("hard", synth.hard_species_set(a.species, a.genomes, a.length))):
        dev = [torch.from_numpy(np.frombuffer(t, dtype=np.uint8).copy()).cuda() for _, _, t in items]
        seqs = [(d.data_ptr(), d.numel()) for d in dev]
        group_of = [s - 1 for s, _, _ in items]
        for k in (21, 31, 41):
            eng.exp1_run(seqs, group_of, k)
            eng.sync()
            eng.profile(True)
            eng.stats_reset()
            ts = []
            for _ in range(a.reps):
                t0 = time.perf_counter()
                res = eng.exp1_run(seqs, group_of, k)
                eng.sync()
                ts.append(time.perf_counter() - t0)
            st = eng.stats()
            eng.profile(False)
            out["rows"].append({"input": name, "k": k, "ms": round(1e3 * statistics.median(ts), 3),
                                "replans_per_run": st["retries"] / a.reps,
                                "launches_per_run": {n: v["launches"] / a.reps for n, v in st["kernels"].items() if v["launches"]},
                                "distinct": int(res["distinct_per_seq"].sum())})
        del dev
        eng.trim()
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    main()
