"""N>1 end to end on the one-GPU box: 2 ranks share device 0, collectives over gloo."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("ranks", [2, 3])
def test_ranks_share_one_gpu(ranks):
    """The exchange of minimizer records and of key sets, 2 and 3 ranks, against the single-process engine and
    the C restatement (tests/dist_rehearsal.py)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                        "--master-addr", "127.0.0.1", "--master-port", str(29571 + ranks),
                        os.path.join(HERE, "dist_rehearsal.py")],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "REHEARSAL_OK" in r.stdout, r.stdout[-3000:]


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it (what the driver runs): the parent
    starts the ranks itself, before touching HIP, and relays rank 0's line."""
    import json
    env = dict(os.environ, KHOICE_SHARE_GPU="1", KHOICE_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--species", "3", "--genomes", "2", "--length", "400000", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0


def test_c_level_rccl_exchange_world_1():
    """kh_comm_init / kh_across_exchange_histogram (the exchange behind the C ABI, RCCL loaded with
    dlopen) at world size 1: the local across-group set goes through the all-gather of bounds, the
    grouped send/recv (to itself), the counter-summing union and the all-reduce, and must give the
    step_8 histogram the fused step computed directly.  (More ranks need more GPUs than this box has.)"""
    from khoice_amd import build as kbuild
    from khoice_amd import engine as E
    from khoice_amd import synth
    kbuild.build_library()
    items = synth.species_set(4, 2, 150_000)
    seqs = [t for _, _, t in items]
    group_of = [s - 1 for s, _, _ in items]
    with E.Engine(0) as eng:
        for k in (31, 41):
            res = eng.exp1_run(seqs, group_of, k, cs=5000, hist_len=64, want_across_set=True)
            comm = eng.comm_init(0, 1, eng.comm_unique_id())
            try:
                h = eng.across_exchange_histogram(comm, res["across_set"], 5000, 64)
            finally:
                eng.comm_destroy(comm)
            assert (h == res["across_hist"]).all()
            assert int(h[2:].sum()) > 0
