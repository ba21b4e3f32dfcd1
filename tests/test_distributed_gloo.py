"""The N>1 path on CPU: two processes over gloo run the same sharding / barrier / max-over-ranks plumbing that
bench.py uses on RCCL, with the oracle's C restatement standing in for the GPU solve (checker-only use)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, batch, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import almpc_loader
    import mpc_oracle as mo
    import c_oracle as co
    pkg = almpc_loader.load_package()
    ranks = pkg.sharding.Ranks(backend="gloo")
    lo, hi = pkg.sharding.shard_range(batch, ranks.rank, ranks.world)
    p = mo.double_integrator()
    des = mo.design_shared(p)
    xi = mo.splitmix_normal(0x5EED0003, lo, hi - lo, 2) * np.array([3.0, 1.0])  # generator stream = instance index
    ranks.barrier()
    r = co.step_batch(p, des, xi, threads=1)
    elapsed = 1.0 + ranks.rank  # fake, to check the MAX reduction
    emax = ranks.max_over_ranks(elapsed)
    tot = ranks.sum_over_ranks(hi - lo)
    q.put((rank, lo, hi, r["u"][:, 0, 0].tolist(), emax, tot))
    ranks.close()


@pytest.mark.timeout(300)
def test_two_rank_shards_reassemble_the_batch(mo, co):
    import torch.multiprocessing as mp
    batch, world, port = 37, 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = sorted(q.get(timeout=240) for _ in range(world))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert [g[1:3] for g in got] == [(0, 19), (19, 37)]
    assert all(g[4] == 2.0 for g in got)       # max over ranks of the elapsed time
    assert all(g[5] == batch for g in got)     # every instance solved exactly once
    # shards glued together == the single-process solve of the whole batch
    p = mo.double_integrator()
    des = mo.design_shared(p)
    xi = mo.splitmix_normal(0x5EED0003, 0, batch, 2) * np.array([3.0, 1.0])
    full = co.step_batch(p, des, xi, threads=1)["u"][:, 0, 0]
    np.testing.assert_array_equal(np.array(got[0][3] + got[1][3]), full)


def _bench(extra, env_extra):
    import subprocess
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=600)


@pytest.mark.timeout(900)
def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` without a launcher starts two ranks, which rendezvous, all-reduce the rank count and the
    max-over-ranks time, and rank 0 prints ONE line with n_gpus = 2 (--rendezvous-only: the solver needs a GPU, the plumbing does not)."""
    import json
    r = _bench(["--gpus", "2", "--rendezvous-only"], {"ALMPC_DIST_BACKEND": "gloo", "ALMPC_NUM_DEVICES": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["global_batch"] == 8192
    assert out["max_over_ranks_check"] == 2.0 and out["value"] is None


@pytest.mark.timeout(300)
def test_bench_refuses_more_ranks_than_gpus():
    """Fewer GPUs than --gpus (none at all in the build container): non-zero exit, a message, no JSON line -- never a silent
    one-GPU number under an n_gpus it did not run."""
    import torch
    have = torch.cuda.device_count()
    r = _bench(["--gpus", str(have + 2), "--steps", "1", "--warmup", "0"], {"ALMPC_NUM_DEVICES": "0"})
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
