#!/usr/bin/env python3
"""bench.py — distinct canonical k-mers / second of the khoice experiment-type-1 hot path.

One step = the whole device side of workflow/rules/exp_type_1.smk:156-259 for k = 31 over a
synthetic 10-species x 5-genome x 5 Mbp set (BASELINE.json north_star / SURVEY.md §8d
"Headline"): per genome build + set (steps 1-2), per group union-sum + histogram
(steps 3-4), group sets (step 6), across-group union-sum + histogram (steps 7-8).
Inputs (cleaned sequence text, 1 byte per base) are resident in HBM before the timed region.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU; every rank owns its own 10 species (weak scaling); steps 1-6 need
no communication, step 7-8 exchanges minimizer records by slot range over RCCL (all-to-all; key sets by key
range outside 20 <= k <= 32) and all-reduces the histogram (khoice_amd/dist.py).

value = (sum over all genomes of all ranks of their distinct canonical k-mers) / step time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md §Chip-level parameters)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--species", type=int, default=10)
    ap.add_argument("--genomes", type=int, default=5)
    ap.add_argument("--length", type=int, default=5_000_000)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-species", type=int, default=0, help="species in the CPU baseline sample (0 = all)")
    return ap.parse_args()


def source_hash() -> str:
    """sha256 over the kernel / engine sources: ties a PMC traffic file to the code it was measured on
    (the GPU box has no .git to ask)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "khoice_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".cpp", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def kernel_algorithmic_bytes(name, st, k, nseq_bases):
    """Compulsory HBM bytes of ONE launch of a kernel class (read every input once, write
    every output once), from the engine's own counters for the timed steps."""
    w = 1 if k <= 32 else 2
    kb = 8 * w
    launches = max(1, st["kernels"][name]["launches"])
    steps_bases = st["bases"]
    kmers, distinct = st["kmers"], st["distinct"]
    if name == "extract_hist":
        total = steps_bases
    elif name == "extract_scatter":
        total = steps_bases + kmers * kb
    elif name == "bucket_sort_rle":
        total = kmers * kb + distinct * kb           # plain sets: no counter array written
    elif name == "setop":
        total = st["setop_in"] * kb + st["setop_out"] * (kb + 4)
    elif name == "union_tagged":
        total = st["setop_in"] * kb                  # reads every genome set once; writes histograms only
    # super-k-mer form (khoice_amd/csrc/kh_skm.hip, kh_skm2.hip): what each kernel has to move; a record is 16
    # bytes with one-word keys (k <= 32), 32 bytes with two-word keys
    elif name == "skm_scatter":
        total = steps_bases + st.get("skm_records", 0) * (16 * w)     # every base in, every record out
    elif name == "skm_regroup":
        total = st.get("skm_records", 0) * (32 * w)                    # every record in and out
    elif name == "skm_union":
        total = st.get("skm_records", 0) * (16 * w)                    # every record in; histograms out
    else:
        total = 0
    return total / launches


def self_launch(ngpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU)
    BEFORE this process has imported torch or touched HIP, relay rank 0's JSON line, and return
    non-zero if any rank failed.  Children are plain subprocesses of this script with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, i.e. exactly what torch.distributed.run does."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    procs = []
    for r in range(ngpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(ngpus),
                   LOCAL_WORLD_SIZE=str(ngpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = p.wait() or rc
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    return rc


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    import torch
    import torch.distributed as dist

    from khoice_amd import engine as E
    from khoice_amd import synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; khoice_amd has no CPU fallback")
    if os.environ.get("KHOICE_SHARE_GPU") == "1":   # rehearsal: every rank on device 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("KHOICE_BENCH_FORCE_DIST") == "1"   # rehearse the N>1 path on 1 GPU
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("KHOICE_DIST_BACKEND", "nccl")   # "gloo": rehearsal on a shared GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    # ---- synthetic inputs, resident in HBM
    t0 = time.time()
    items = synth.species_set(args.species, args.genomes, args.length,
                              first_species=1 + rank * args.species)
    group_of = [s - 1 - rank * args.species for s, _, _ in items]
    host = [np.frombuffer(t, dtype=np.uint8) for _, _, t in items]
    dev = [torch.from_numpy(h.copy()).cuda() for h in host]
    seqs = [(d.data_ptr(), d.numel()) for d in dev]
    total_bases = sum(d.numel() for d in dev)
    gen_s = time.time() - t0

    eng = E.Engine(local_rank)
    use_dist = world > 1 or force_dist
    if use_dist:
        from khoice_amd import dist as kdist

    def step():
        if not use_dist:
            return eng.exp1_run(seqs, group_of, args.k, cs=5000, hist_len=5001)
        return kdist.exp1_step(eng, seqs, group_of, args.k, cs=5000, hist_len=5001)

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    res = None
    for _ in range(args.warmup):
        res = step()
    eng.stats_reset()
    eng.profile(os.environ.get("KHOICE_BENCH_NOPROF") != "1")
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    dt = time.perf_counter() - t1
    st = eng.stats()          # collects the HIP events recorded on the engine's stream
    eng.profile(False)

    distinct_local = int(res["distinct_per_seq"].sum())
    if world > 1:
        cdev = "cpu" if dist.get_backend() == "gloo" else "cuda"
        tt = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        dd = torch.tensor([distinct_local], device=cdev, dtype=torch.int64)
        dist.all_reduce(dd, op=dist.ReduceOp.SUM)
        distinct_all = int(dd.item())
    else:
        distinct_all = distinct_local
    ms_per_step = 1e3 * dt / args.steps
    value = distinct_all / (dt / args.steps)

    # ---- roofline of the dominant kernel (HIP events on the engine's stream)
    kern = st["kernels"]
    hot = max((n for n in kern if n not in ("copy_in",)), key=lambda n: kern[n]["ms"])
    avg_ms = kern[hot]["ms"] / max(1, kern[hot]["launches"])
    alg_bytes = kernel_algorithmic_bytes(hot, st, args.k, total_bases)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM bytes per launch from the PMC passes (tools/collect_profiles.sh): only when that file was
    # collected on exactly these sources, otherwise null + traffic_stale
    traffic, traffic_stale = None, False
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("_source_sha") == source_hash():
                traffic = tj.get(hot, {}).get("bytes_per_launch")
            else:
                traffic_stale = True
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": hot, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_stale": traffic_stale,
                "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes)}
    # The dominant kernel of the super-k-mer form is bound by vector instruction issue and LDS round trips, not by
    # HBM: when the SQ counter file was collected on exactly these sources (tools/pmc_sq.sh), report its wave
    # instructions per second against the chip's MEASURED issue rate for the integer instructions these kernels are
    # made of (tools/micro/valu_issue.hip, profiles/r03_valu_issue.txt: 595 G wave-instructions/s for VOP3 / VOP1 /
    # 64-bit forms at 8 waves per SIMD on every CU = one per 4.1 cycles per SIMD; plain v_add_u32 / v_xor_b32 reach
    # 950-1050 G/s, which this mix cannot).
    issue = None
    spath = os.path.join(ROOT, "profiles", "sq_counters.json")
    if os.path.exists(spath):
        try:
            sj = json.load(open(spath))
            if sj.get("_source_sha") == source_hash() and hot in sj and avg_ms > 0:
                valu = sj[hot].get("SQ_INSTS_VALU", 0)
                peak = 595e9
                issue = {"bound": "valu_issue", "kernel": hot, "wave_instructions_per_launch": int(valu),
                         "achieved": round(valu / (avg_ms * 1e-3) / 1e9, 1), "peak": round(peak / 1e9, 1),
                         "unit": "G wave-instr/s", "frac": round(valu / (avg_ms * 1e-3) / peak, 4),
                         "sq_wait_any_share": round(sj[hot].get("SQ_WAIT_ANY", 0) / max(1, sj[hot].get("SQ_WAVE_CYCLES", 1)), 3)}
        except Exception:
            issue = None
    # whole K1 build against SURVEY §8d's compulsory figure N*1 + D*(8W+4)
    w = 1 if args.k <= 32 else 2
    # (the super-k-mer form builds no per-genome sets: its two partition kernels stand in for passes A-C)
    k1_ms = sum(kern[n]["ms"] for n in ("extract_hist", "bucket_plan", "extract_scatter", "bucket_sort_rle",
                                        "skm_scatter", "skm_regroup") if n in kern)
    k1_bytes = st["bases"] + st["distinct"] * (8 * w + 4)
    kernel_ms = {n: round(v["ms"] / args.steps, 4) for n, v in kern.items()}
    # whole path against SURVEY §8d's fused-path figure (read every base once, write every group
    # union and the across-group union once, read the group sets once):
    #   sum_g N_g + U (8W+4)  +  sum_g U_g 8W + V (8W+4)
    # where U = sum of the group unions' sizes, V = size of the across-group union.  The fused
    # engine path moves fewer bytes than that (no union is written when no set is requested), so
    # this prices the step against the reference DAG's compulsory traffic, not against its own.
    u_total = int(res["within_hist"][:, 1:].sum())
    v_total = int(res["across_hist"][1:].sum()) if res.get("across_hist") is not None else 0
    path_bytes = total_bases + u_total * (8 * w + 4) + u_total * 8 * w + v_total * (8 * w + 4)
    path_gbs = path_bytes / (ms_per_step * 1e-3) / 1e9
    path_roofline = None
    if world == 1:      # (at N > 1 the across-group histogram is global, the other terms per rank)
        path_roofline = {"bound": "hbm", "algorithmic_bytes_per_step": int(path_bytes),
                         "achieved": round(path_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(path_gbs / HBM_PEAK_GBS, 4),
                         "formula": "bases + U(8W+4) + U*8W + V(8W+4), SURVEY 8d"}

    out = {
        "metric": "distinct k-mers/sec (k=31, canonical)" if args.k == 31 else f"distinct k-mers/sec (k={args.k}, canonical)",
        "value": round(value, 1), "unit": "distinct k-mers/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u64" if args.k <= 32 else "u128",
        "data": "synthetic",
        "config": {"workload": f"exp_type_1 steps 1-8: {args.species} species x {args.genomes} genomes x "
                               f"{args.length} bp per GPU, k={args.k}, cs=5000",
                   "species_per_gpu": args.species, "genomes_per_species": args.genomes,
                   "genome_bp": args.length, "k": args.k, "bases_per_gpu": total_bases,
                   "distinct_kmers_per_step": distinct_all, "sharding": f"groups x{world}"},
        "roofline": roofline,
        "k1_build_roofline": {"algorithmic_bytes": int(k1_bytes / args.steps),
                              "ms": round(k1_ms / args.steps, 4),
                              "achieved": round(k1_bytes / (k1_ms * 1e-3) / 1e9, 1) if k1_ms else 0.0,
                              "frac": round(k1_bytes / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k1_ms else 0.0,
                              "unit": "GB/s"},
        "path_roofline": path_roofline,
        "issue_roofline": issue,
        "kernel_ms_per_step": kernel_ms,
        "replans": st["retries"], "order_fallbacks": st.get("order_fallbacks", 0),
        # N > 1: how steps 7-8 crossed the ranks ("records": every rank in the super-k-mer form, minimizer records
        # exchanged by slot range; "sets": key sets exchanged by key range) — a rank's step is then TWO passes over its
        # k-mers (within-group locally, across-group on exchanged data), not the one pass of the N = 1 line
        "exchange_form": (res.get("exchange", "sets") if use_dist else None),
        "setup_seconds": round(gen_s, 1),
    }

    # ---- CPU baseline: the C restatement (oracle/kh_oracle.c, kho_exp1) on the host cores.
    # All threads (genomes in parallel, surplus threads inside the radix passes) on the whole
    # workload, and ONE thread on one species of it; both checked against the GPU histograms.
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import c_oracle as CO
        ns = args.species if args.cpu_species <= 0 else min(args.cpu_species, args.species)
        sample = [(s, g, t) for s, g, t in items if s - 1 < ns]
        cseqs = [t for _, _, t in sample]
        cgroup = [s - 1 for s, _, _ in sample]
        cores = os.cpu_count() or 1
        c0 = time.perf_counter()
        cres = CO.exp1(cseqs, cgroup, args.k, cs=5000, hist_len=5001, nthreads=cores)
        cdt = time.perf_counter() - c0
        cd = int(cres["distinct_per_seq"].sum())
        ok = bool((cres["within_hist"] == res["within_hist"][:ns]).all())
        ok_across = bool((cres["across_hist"] == res["across_hist"]).all()) if ns == args.species else None
        one = [(s, g, t) for s, g, t in items if s == 1]
        o0 = time.perf_counter()
        ores = CO.exp1([t for _, _, t in one], [0] * len(one), args.k, cs=5000, hist_len=5001, nthreads=1)
        odt = time.perf_counter() - o0
        od = int(ores["distinct_per_seq"].sum())
        model = ""
        try:
            for ln in open("/proc/cpuinfo"):
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        out["cpu_baseline"] = {"value": round(cd / cdt, 1), "unit": "distinct k-mers/s",
                               "cores": int(cres["threads"]), "kind": "port", "cpu_model": model,
                               "sample": f"{ns} species x {args.genomes} genomes x {args.length} bp "
                                         f"({'the whole workload' if ns == args.species else 'of the same set'}), "
                                         f"oracle/kh_oracle.c kho_exp1, {cdt:.1f} s",
                               "one_thread": {"value": round(od / odt, 1), "cores": 1,
                                              "sample": f"species 1 ({len(one)} genomes), {odt:.1f} s",
                                              "within_hist_equal_to_gpu":
                                                  bool((ores["within_hist"][0] == res["within_hist"][0]).all())},
                               "within_hist_equal_to_gpu": ok, "across_hist_equal_to_gpu": ok_across}
    if rank == 0:
        print(json.dumps(out), flush=True)
    eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
