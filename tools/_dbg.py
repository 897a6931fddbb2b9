import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from khoice_amd import engine as E, synth
from test_gpu_scale import big_group_set
eng = E.Engine(0)
seqs, group_of = big_group_set([10, 70, 200, 300], 50_000)
order = np.random.default_rng(5).permutation(len(seqs))
seqs = [seqs[i] for i in order]; group_of = [group_of[i] for i in order]
sizes = [10, 70, 200, 70, 10]
seqs, group_of = big_group_set(sizes, 1_000_000)
dev = [torch.from_numpy(np.frombuffer(t, dtype=np.uint8).copy()).cuda() for t in seqs]
ptrs = [(x.data_ptr(), x.numel()) for x in dev]
for k in (31,):
    eng.exp1_run(ptrs, group_of, k, cs=5000, hist_len=5001); eng.sync()
    t0 = time.perf_counter(); eng.exp1_run(ptrs, group_of, k, cs=5000, hist_len=5001); eng.sync(); t1 = time.perf_counter()
    eng.profile(True); eng.stats_reset()
    eng.exp1_run(ptrs, group_of, k, cs=5000, hist_len=5001); eng.sync()
    st = eng.stats(); eng.profile(False)
    print("cfg5 k", k, "ms", round(1e3 * (t1 - t0), 2), "retries", st["retries"], "big", st["big_slots"])
    for n, v in sorted(st["kernels"].items(), key=lambda kv: -kv[1].get("ms", 0)):
        if v["launches"]: print("   ", n, v["launches"], round(v.get("ms", 0), 3))
