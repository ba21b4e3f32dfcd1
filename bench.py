#!/usr/bin/env python3
"""bench.py -- MPC steps/s of the HIP condensed-QP path on BASELINE.json's headline configuration.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (gradient, ADMM, polish, rollout) over ONE batch of 4096 quadrotor
instances (n=12, m=4, N=30; BASELINE.json configs[1]) per GPU; inputs (x0) are resident in HBM before the
timed region, results stay in HBM.  N > 1: one process per GPU (launched by torch.distributed.run), each rank
solves its own contiguous 4096-instance shard of the 4096*N batch (weak scaling, no data-path collective;
torch.distributed over RCCL is used only for the barrier and the max-over-ranks of the elapsed time).

`value` is 4096-instance batch steps per second over all GPUs (= instance-steps/s / 4096);
`instance_steps_per_s` is printed beside it.  The workload mixes the three amplitude classes of SURVEY.md
section 8d (s = 0.3 / 1.0 / 3.0, interleaved by instance index); per-class rates are in `classes`.

The oracle (oracle/) is used here only as the checker (`u_err_inf`) and as the `cpu_baseline` leg.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np  # noqa: E402

BATCH_PER_GPU = 4096
N_HORIZON, NX, NU = 30, 12, 4
NZ = NU * N_HORIZON
FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector = matrix peak (public spec)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
# algorithmic HBM bytes per instance-step (x0, refs, warm start in; u, x, warm start out)
ALG_BYTES_PER_INSTANCE_STEP = 11808
AMPLITUDES = (0.3, 1.0, 3.0)


def make_x0(wl, first_instance, count, amplitude=None, seed=0x5EED0002):
    """x0 of instances [first, first+count): amplitude class by instance index unless one is forced."""
    xi = wl.splitmix_normal(seed, first_instance, count, NX) * wl.QUADROTOR_X0_SCALE[None, :]
    if amplitude is None:
        amp = np.array(AMPLITUDES)[(np.arange(first_instance, first_instance + count) % 3)]
        return np.ascontiguousarray(xi * amp[:, None])
    return np.ascontiguousarray(xi * amplitude)


def time_steps(solver, opts, steps, barrier):
    barrier()
    solver.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        solver.calculate(opts, sync=False)
    solver.synchronize()
    barrier()
    return time.perf_counter() - t0


class c_stdout_to_stderr:
    """librccl prints a version banner on C stdout when NCCL_DEBUG asks for it (the GPU pool sets NCCL_DEBUG=VERSION), from whichever
    call first creates a communicator.  stdout of this job carries ONE JSON line: while communicators are made, file descriptor 1
    points at stderr, and C stdio is flushed before it is put back."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def api_path_figures(capi, solver, X0, opts, steps):
    """The step as the reference's API shapes it -- host x0 in (update_initialization!, src/main/computation_mpc.jl:17-29), host results
    out (calculate!, src/main/computation_mpc.jl:50-53) EVERY step -- through the pinned / copy-stream entry points:
      first_move   x0 (393 KB) up, u[:,1] + status (131 KB + 16 KB) down; request k+1 is enqueued before the results of k are waited for
      full         x0 up, x, e_x, u, e_u + status (32 MB) down into the handle's pinned slots (read in place: zero-copy views)
      sync_legacy  almpc_update_initialization + almpc_calculate + almpc_get_results (pageable, synchronous): the round-2 path
    PCIe-inclusive rates; `value` of the headline is the HBM-resident rate."""
    k = max(50, steps)
    out = {}
    solver.timing_set_stride(1 << 30)   # no event records in these loops
    # --- first move, pipelined depth 2.  The loops below call the C ABI directly with prebuilt arguments (what a compiled caller or a
    # Julia ccall does): the numpy-level wrappers of _capi.Solver cost ~10 us per call, which is half a step here.
    import ctypes
    L, h = solver.L, solver.h
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
    x0c = np.ascontiguousarray(X0, dtype=np.float64)
    x0p, op = x0c.ctypes.data_as(dp), ctypes.byref(opts)
    u0 = np.empty((X0.shape[0], NU)); st = np.empty(X0.shape[0], dtype=np.int32)
    u0p, stp = u0.ctypes.data_as(dp), st.ctypes.data_as(ip)
    mask = capi.WANT["u0"] | capi.WANT["status"]

    def ok(rc):
        if rc < 0:
            solver._check(rc)
        return rc

    def step_and_ask():
        ok(L.almpc_update_initialization_async(h, x0p))
        ok(L.almpc_calculate_async(h, op))
        return ok(L.almpc_get_results_async(h, mask))

    def wait(t):
        ok(L.almpc_get_results_wait(h, t, None, None, None, None, u0p, stp, None, None))
    for _ in range(3):
        wait(step_and_ask())
    best = float("inf")
    for _rep in range(3):
        t0 = time.perf_counter()
        prev = -1
        for _ in range(k):
            t = step_and_ask()
            if prev >= 0:
                wait(prev)
            prev = t
        wait(prev)
        best = min(best, time.perf_counter() - t0)
    ref_u0 = solver.get_results(want=("u",))["u"][:, :, 0]
    out["first_move"] = {"value": k / best, "unit": "batch-steps/s", "ms_per_step": 1e3 * best / k,
                         "bytes_up_per_step": int(x0c.nbytes), "bytes_down_per_step": int(u0.nbytes + st.nbytes),
                         "unsolved_last": int((st != 0).sum()), "matches_get_results": bool(np.array_equal(u0, ref_u0)),
                         "note": "host x0 in -> step -> host u[:,1] + status out, every step; depth-2 pipeline (the results of step k are "
                                 "waited for after step k+1 is enqueued); C ABI called with prebuilt arguments"}
    # zero-copy variant: x0 written in place into the handle's pinned slot (almpc_x0_staging), results read in place (almpc_host_results)
    slotp = dp()

    def step_zero(fill):
        ok(L.almpc_x0_staging(h, ctypes.byref(slotp)))
        if fill:   # (the benchmark's states do not change from step to step: both slots are filled once, before the timed loop; a real
            np.copyto(np.ctypeslib.as_array(slotp, shape=x0c.shape), x0c)   # caller's plant writes its new states here instead)
        ok(L.almpc_update_initialization_async(h, slotp))
        ok(L.almpc_calculate_async(h, op))
        return ok(L.almpc_get_results_async(h, mask))

    def wait_zero(t):
        ok(L.almpc_get_results_wait(h, t, None, None, None, None, None, None, None, None))
    for _ in range(4):
        wait_zero(step_zero(True))
    best = float("inf")
    for _rep in range(3):
        t0 = time.perf_counter()
        prev = -1
        for _ in range(k):
            t = step_zero(False)
            if prev >= 0:
                wait_zero(prev)
            prev = t
        wait_zero(prev)
        best = min(best, time.perf_counter() - t0)
    zview = solver.get_results_wait(prev, want=("u0", "status"), copy=False)
    out["first_move_zero_copy"] = {"value": k / best, "unit": "batch-steps/s", "ms_per_step": 1e3 * best / k,
                                   "matches_get_results": bool(np.array_equal(zview["u0"], ref_u0)) and bool((zview["status"] == 0).all()),
                                   "note": "as first_move without the two staging copies: x0 lives in the handle's pinned slot (almpc_x0_staging; the "
                                           "caller's write of its states is not part of the figure: it replaces the write into the caller's own "
                                           "array), u[:,1] and status are read in place (almpc_host_results)"}
    # the same without pipelining: every step waits for its own results (a closed loop whose plant lives on the host)
    best = float("inf")
    for _rep in range(3):
        t0 = time.perf_counter()
        for _ in range(k):
            wait(step_and_ask())
        best = min(best, time.perf_counter() - t0)
    out["first_move_serial"] = {"value": k / best, "unit": "batch-steps/s", "ms_per_step": 1e3 * best / k,
                                "note": "as first_move, but every step waits for its own u[:,1] before the next x0 goes up (host-side plant)"}
    # --- all four arrays into pinned memory
    want = ("x", "e_x", "u", "e_u", "status")
    kf = max(20, k // 5)
    for _ in range(2):
        solver.update_initialization_async(X0); solver.calculate(opts, sync=False)
        solver.get_results_wait(solver.get_results_async(want), want, copy=False)
    best = float("inf")
    for _rep in range(3):
        t0 = time.perf_counter()
        prev = None
        for _ in range(kf):
            solver.update_initialization_async(X0)
            solver.calculate(opts, sync=False)
            t = solver.get_results_async(want)
            if prev is not None:
                rf = solver.get_results_wait(prev, want, copy=False)
            prev = t
        rf = solver.get_results_wait(prev, want, copy=False)
        best = min(best, time.perf_counter() - t0)
    down = int(sum(rf[q].nbytes for q in want))
    gbs = (down + X0.nbytes) * kf / best / 1e9
    out["full"] = {"value": kf / best, "unit": "batch-steps/s", "ms_per_step": 1e3 * best / kf, "bytes_down_per_step": down,
                   "pcie_GBps": gbs, "pcie_frac_of_64GBps": gbs / 64.0,
                   "note": "x, e_x, u, e_u, status of every step into the handle's pinned slots (read in place); PCIe Gen5 x16 = 64 GB/s per "
                           "direction: this figure is the link, not the kernel"}
    # --- the synchronous pageable path of round 2
    ks = max(10, k // 10)
    best = float("inf")
    for _rep in range(2):
        t0 = time.perf_counter()
        for _ in range(ks):
            solver.update_initialization(X0)
            solver.calculate(opts)
            solver.get_results(want=("x", "e_x", "u", "e_u", "status"))
        best = min(best, time.perf_counter() - t0)
    out["sync_legacy_full"] = {"value": ks / best, "unit": "batch-steps/s", "ms_per_step": 1e3 * best / ks,
                               "note": "almpc_update_initialization + almpc_calculate + almpc_get_results into fresh pageable arrays"}
    return out


def single_process(args):
    """`--gpus N --single-process`: ONE process drives N devices through an almpc_group (one handle, one stream per device; enqueue on
    every device, then wait for each).  No launcher, no torch.distributed, no barrier: the elapsed time is this process's clock around
    K group steps."""
    import almpc_loader
    import importlib
    pkg = almpc_loader.load_package()
    capi = pkg._capi
    wl = importlib.import_module(pkg.__name__ + ".workloads")
    n = args.gpus
    fold = int(os.environ.get("ALMPC_NUM_DEVICES", "0"))   # test hook: several handles per device
    devices = [(i % fold) if fold > 0 else i for i in range(n)]
    seed = 0x5EED0003 if n == 8 else 0x5EED0002
    p = wl.quadrotor(N_HORIZON)
    batch = n * BATCH_PER_GPU
    X0 = make_x0(wl, 0, batch, seed=seed)
    rho = args.rho if args.rho is not None else (45.0 if args.rho_profile == "stiffness" else 0.1)
    try:
        g = capi.Group(NX, NU, N_HORIZON, batch, devices)
    except capi.AlmpcError as e:
        sys.stderr.write(f"bench.py --single-process: {e}\n")
        return 2
    g.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, rho=rho, rho_profile=args.rho_profile)
    g.set_reference(p.x_ref, p.u_ref)
    g.update_initialization(X0, resident=True)   # (HBM resident, as the N-process path: the timed steps do not pull x0 over the link)
    opts = capi.default_opts(rho=rho, max_iter=args.max_iter, check_every=args.max_iter, keep_warm_state=False)

    def run(k):
        g.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            g.calculate(opts, sync=False)
        g.synchronize()
        return time.perf_counter() - t0
    run(300)
    run(args.warmup)
    elapsed = run(args.steps)
    r = g.get_results(want=("u0", "status"))
    out = {"metric": "MPC steps/s (batch=4096, nx=12, nu=4, N=30)", "value": n * args.steps / elapsed,
           "unit": "batch-steps/s (one step = 4096 instance QP solves)", "instance_steps_per_s": batch * args.steps / elapsed,
           "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "configs[1] shards (configs[2] at 8 GPUs): hover-linearised quadrotor nx=12 nu=4 N=30, 4096 instances per "
                                  "GPU, shared model, cold start every step", "launch": "single process, almpc_group (one handle per device)",
                      "devices": devices, "global_batch": batch, "seed": hex(seed), "setup_ramp_steps": 300},
           "status_counts": np.bincount(r["status"], minlength=3).tolist()}
    g.close()
    if not args.no_relin:
        # BASELINE configs[3] through the group (almpc_group_relin_fnn_*): 1024 instances per device, re-linearised every step
        bq, nq, mq, Nq = 1024 * n, 4, 2, 20
        W_in, W_h, b_h, W_out = wl.synthetic_fnn_weights(nq, mq)
        f = pkg.Fnn(W_in, W_h, b_h, W_out, "relu")
        xr = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, Nq + 1)); ur = np.tile(np.array([0.1, -0.2])[:, None], (1, Nq))
        Xq = xr[:, 0][None, :] + wl.splitmix_normal(0x5EED0004, 0, bq, nq)
        Al, Bl = capi.fnn_linearize(f.W_in, f.W_h, f.b_h, f.W_out, xr[:, -1][None, :], ur[:, -1][None, :], act="relu", device=devices[0])
        Pq = capi.dare(Al[0], Bl[0], 100.0 * np.eye(nq), 0.1 * np.eye(mq))
        g3 = capi.Group(nq, mq, Nq, bq, devices)
        g3.relin_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, xr, ur, 100.0 * np.eye(nq), 0.1 * np.eye(mq), None, Pq, -np.ones(mq), np.ones(mq), act="relu")
        g3.update_initialization(Xq, resident=True)

        def run3(k):
            g3.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                g3.relin_fnn_step(None, sync=False)
            g3.synchronize()
            return time.perf_counter() - t0
        run3(50)
        k3 = 100
        el3 = min(run3(k3) for _ in range(3))
        r3 = g3.get_results(want=("status",))
        out["config3_fnn_relin"] = {"value": n * k3 / el3, "unit": "batch-steps/s (1024 instances per device, Fnn 4-2-16x2 relu, N=20, re-linearised every step)",
                                    "ms_per_step": 1e3 * el3 / k3, "global_batch": bq, "status_counts": np.bincount(r3["status"], minlength=3).tolist(),
                                    "launch": "single process, almpc_group_relin_fnn_step_async on every device, then almpc_group_synchronize"}
        g3.close()
    print(json.dumps(out))
    return 0


def spawn_ranks(n):
    """`python bench.py --gpus N` typed by hand (no launcher): start N ranks of this script under torch.distributed.run, one per
    GPU, relay their output (rank 0 prints the JSON line) and return the launcher's exit code.  Runs before anything in this
    process initialises HIP (the devices are counted by a short-lived child process); children are separate processes, nothing
    is exec'ed over this one."""
    import socket
    import subprocess
    fold = int(os.environ.get("ALMPC_NUM_DEVICES", "0"))   # test hook: fold ranks onto fewer devices (needs ALMPC_DIST_BACKEND=gloo)
    if fold <= 0:
        import almpc_loader
        have = almpc_loader.load_package().sharding.visible_gpu_count()   # counted in a throw-away child: no HIP in this process
        if have < n:
            sys.stderr.write(f"bench.py: --gpus {n} but only {have} GPU(s) visible on this node; refusing to fold ranks onto fewer "
                             f"devices (set ALMPC_NUM_DEVICES and ALMPC_DIST_BACKEND=gloo for a plumbing test)\n")
            return 2
    elif os.environ.get("ALMPC_DIST_BACKEND", "nccl") == "nccl":
        sys.stderr.write("bench.py: ALMPC_NUM_DEVICES folds ranks onto shared devices, which RCCL refuses: set ALMPC_DIST_BACKEND=gloo\n")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--max-iter", type=int, default=6,
                    help="ADMM iterations before the polish (= check interval); 6 is the tuned value for this workload "
                         "(sweep in DESIGN.md section 4), the library default is 25")
    ap.add_argument("--rho-profile", default="stiffness", choices=("scalar", "stiffness"),
                    help="ADMM penalty: OSQP's scalar rho, or rho_i = rho / (H'^-1)_ii (almpc_set_rho_profile)")
    ap.add_argument("--rho", type=float, default=None, help="rho (default 45 for the stiffness profile, 0.1 for scalar)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-classes", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true")
    ap.add_argument("--no-closed-loop", action="store_true")
    ap.add_argument("--no-batched-models", action="store_true", help="skip the secondary per-instance-model figure")
    ap.add_argument("--no-sqp", action="store_true", help="skip the secondary SQP (BASELINE configs[4]) figure")
    ap.add_argument("--relin-max-iter", type=int, default=25,
                    help="ADMM iterations before the exact finish in the configs[3] leg (= check interval; 25 = the library default).  On the "
                         "bench's batch 8..25 are within 1 % of each other (the step ends with the finish of its hardest instance), 4..6 are "
                         "twice as slow (DESIGN.md section 4)")
    ap.add_argument("--no-structured", action="store_true", help="skip the secondary structured-solve (N = 50) figures")
    ap.add_argument("--no-small-shared", action="store_true", help="skip the secondary small-shared-model figure (the reference's own test size, 65,536 instances)")
    ap.add_argument("--no-state-rows", action="store_true", help="skip the secondary state-row (state box + terminal equality, N = 30) figure")
    ap.add_argument("--no-relin", action="store_true", help="skip the secondary per-step re-linearisation (BASELINE configs[3]) figure")
    ap.add_argument("--no-api-path", action="store_true", help="skip the host-in / host-out figures (api_path_first_move, api_path_full)")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N from ONE process: an almpc_group with one handle per device (no launcher, no torch.distributed)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="plumbing check: spawn / rendezvous / barrier / reductions and the JSON line, no solver (value is null); "
                         "the only mode that runs without a GPU")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.single_process:
        raise SystemExit(single_process(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))   # nothing in this process has touched the GPU

    import almpc_loader
    import importlib
    pkg = almpc_loader.load_package()
    capi = pkg._capi
    wl = importlib.import_module(pkg.__name__ + ".workloads")  # synthetic inputs (product side)
    # oracle/ is test infrastructure: it is imported only by the checker legs below (u_err_inf, the NLP certificate) and by the
    # cpu_baseline leg, never for the inputs and never on the measured path
    mo = None
    with c_stdout_to_stderr():
        ranks = pkg.sharding.Ranks(backend="nccl")  # one process per GPU; RCCL only for barrier / max-reduce
        rank, local_rank, world = ranks.rank, ranks.local_rank, ranks.world
        if world > 1 and args.gpus != world:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        barrier, max_over_ranks = ranks.barrier, ranks.max_over_ranks
        # ranks that actually joined the job, counted by an all-reduce over the process group (RCCL on the GPU box)
        joined = int(round(ranks.sum_over_ranks(1.0)))
    if joined != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but {joined} rank(s) joined the process group")

    # configs[2]: 32,768 instances over 8 GPUs come from seed 0x5EED0003; every other world size shards the
    # configs[1] stream (seed 0x5EED0002), 4096 instances per GPU either way
    seed = 0x5EED0003 if world == 8 else 0x5EED0002
    p = wl.quadrotor(N_HORIZON)
    first, last = pkg.sharding.shard_range(world * BATCH_PER_GPU, rank, world)
    assert last - first == BATCH_PER_GPU
    if args.rendezvous_only:
        barrier()
        el = max_over_ranks(1.0 + rank)
        if rank == 0:
            print(json.dumps({"metric": "MPC steps/s (batch=4096, nx=12, nu=4, N=30)", "value": None, "n_gpus": world,
                              "rccl_ranks": joined, "global_batch": world * BATCH_PER_GPU, "seed": hex(seed),
                              "max_over_ranks_check": el, "backend": ranks.backend, "data": "none (rendezvous only)"}))
        ranks.close()
        return
    X0 = make_x0(wl, first, BATCH_PER_GPU, seed=seed)
    ndev = int(os.environ.get("ALMPC_NUM_DEVICES", "0"))  # test hook: fold ranks onto fewer devices (with ALMPC_DIST_BACKEND=gloo)
    dev_index = (local_rank % ndev) if ndev > 0 else local_rank
    solver = capi.Solver(NX, NU, N_HORIZON, BATCH_PER_GPU, device=dev_index, timing=True)
    rho = args.rho if args.rho is not None else (45.0 if args.rho_profile == "stiffness" else 0.1)
    design_kw = dict(rho=rho, rho_profile=args.rho_profile)
    solver.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, **design_kw)
    solver.set_reference(p.x_ref, p.u_ref)
    solver.update_initialization(X0)  # x0 resident in HBM from here on
    # the library's own RCCL communicator (almpc_comm_*: one rank per GPU; not possible when the test hook folds ranks onto one device)
    # It is not on the data path (the shards are independent) and it has only ever run as a one-rank communicator (one-GPU boxes), so
    # at N > 1 it must never take the measurement down: the collective init runs in a helper thread with a time limit, the ranks then
    # agree through the launcher's process group whether everybody has it, and the run goes on without it otherwise.
    lib_comm = None
    force_exit = False
    if ranks.backend != "gloo" and os.environ.get("ALMPC_BENCH_NO_LIB_COMM") != "1":
        import threading
        state = {}

        def _init():
            try:
                solver.comm_init(state["uid"], rank, world)
                state["ok"] = True
            except Exception as e:   # noqa: BLE001 (reported, not fatal)
                state["err"] = str(e)

        with c_stdout_to_stderr():
            try:
                uid0 = None
                if rank == 0:
                    try:
                        uid0 = capi.comm_unique_id()
                    except capi.AlmpcError as e:
                        state["err"] = str(e)
                state["uid"] = ranks.broadcast_bytes(uid0)   # (every rank takes part in the broadcast, whatever rank 0 got)
                if state["uid"] is None:
                    raise capi.AlmpcError(-4, state.get("err", "no RCCL id from rank 0"))
                th = threading.Thread(target=_init, daemon=True)
                th.start()
                th.join(timeout=float(os.environ.get("ALMPC_LIB_COMM_TIMEOUT", "90")))
                if th.is_alive():
                    state["err"] = "ncclCommInitRank did not return in time"
                    force_exit = True    # (a thread is stuck inside librccl: leave through os._exit at the end)
            except capi.AlmpcError as e:
                state["err"] = str(e)
        everybody = ranks.sum_over_ranks(1.0 if state.get("ok") else 0.0) >= world - 0.5
        lib_comm = True if everybody else f"unavailable ({state.get('err', 'another rank failed')})"
    # cold start every step (the headline workload): the ADMM state a warm start would need is not stored (ALMPC_OPT_NO_WARM_STATE);
    # the closed-loop leg below uses the default (state kept, warm_start = 1)
    opts = capi.default_opts(rho=rho, max_iter=args.max_iter, check_every=args.max_iter, keep_warm_state=False)
    opts_keep = capi.default_opts(rho=rho, max_iter=args.max_iter, check_every=args.max_iter)

    # HIP events on every 16th step of the timed region only: recording them on every step costs ~14 us/step of stream time
    # (short runs -- the driver's --steps 20 -- take at least FOUR samples: stride 5)
    TIMING_STRIDE = 16 if args.steps >= 64 else max(1, args.steps // 4)
    RAMP_STEPS = 300
    # setup, not part of the W warm-up steps: a fresh box starts with the GPU in a low power state and the code objects unloaded; a
    # few milliseconds of the workload bring clocks and caches to the steady state the metric is about
    solver.timing_set_stride(1 << 30)
    time_steps(solver, opts, RAMP_STEPS, barrier)
    solver.timing_set_stride(TIMING_STRIDE)
    time_steps(solver, opts, args.warmup, barrier)
    # The timed region (exactly --steps steps between barrier + synchronize on both sides, maximum over the ranks) is run
    # TIMED_REPEATS times back to back; `value` is the MEDIAN region, `value_runs` lists all of them.  (At the driver's --steps 20 one
    # region is 1.3 ms of wall clock: one sample.  A long region -- >= 256 steps -- is already an average and is timed once.)
    TIMED_REPEATS = 5 if args.steps < 256 else 1
    region_s, tsamp = [], []
    for _ in range(TIMED_REPEATS):
        solver.timing_reset(args.steps)
        region_s.append(max_over_ranks(time_steps(solver, opts, args.steps, barrier)))
        tsamp += [float(v) for v in solver.timing_samples()["polish_ms"]]   # the event pairs around k_step_fused inside the timed regions, one per sampled step
    elapsed = float(np.median(region_s))
    tsum = solver.timing_summary()
    res = solver.get_results(want=("u", "status", "iters", "polish_iters"))
    # more event pairs around the same kernel right after the timed region (same workload, every 2nd of 64 steps): the roofline's kernel
    # time is the MEDIAN of all pairs -- four samples of a 20-step run are too few, and their mean once came out above the step itself
    solver.timing_set_stride(2)
    solver.timing_reset(64)
    time_steps(solver, opts, 64, barrier)
    tsamp_after = solver.timing_samples()["polish_ms"]
    # The headline step is ONE kernel (k_step_fused: ADMM phase + polish of the same tile).  Its two phases are timed apart on
    # the two-kernel path of the same build (almpc_set_step_fusion(0): k_admm, k_polish<true>), outside the timed region.
    solver.set_step_fusion(False)
    solver.timing_set_stride(4)
    time_steps(solver, opts, 8, barrier)
    solver.timing_reset(64)
    time_steps(solver, opts, 64, barrier)
    tsum2 = solver.timing_summary()
    solver.set_step_fusion(True)
    solver.timing_set_stride(TIMING_STRIDE)
    solver.timing_reset(0)

    comm_line = None
    if lib_comm is True:   # collectives of the library itself, after the timed region: job-wide solve summary and the gathered first inputs
        import threading
        got = {}

        def _collect():
            try:
                got["summ"] = solver.comm_summary()
                got["u0_all"] = solver.comm_allgather_first_input()
            except Exception as e:   # noqa: BLE001
                got["err"] = str(e)

        th = threading.Thread(target=_collect, daemon=True)   # (time-boxed like the init: the measurement above must get printed)
        th.start()
        th.join(timeout=float(os.environ.get("ALMPC_LIB_COMM_TIMEOUT", "90")))
        if th.is_alive():
            got["err"] = "collective did not return in time"
            force_exit = True
        if "u0_all" in got:
            mine = solver.get_results(want=("u",))["u"][:, :, 0]
            comm_line = dict(got["summ"], first_input_gather_shape=list(got["u0_all"].shape),
                             first_input_gather_matches_local=bool(np.array_equal(got["u0_all"][rank], mine)))
        else:
            comm_line = {"error": got.get("err", "unknown")}
    elif lib_comm is not None:
        comm_line = {"error": lib_comm}
    inst_steps_per_s = pkg.sharding.aggregate_rate(BATCH_PER_GPU, args.steps, elapsed, world)
    out = {
        "metric": "MPC steps/s (batch=4096, nx=12, nu=4, N=30)",
        "value": inst_steps_per_s / BATCH_PER_GPU,
        "unit": "batch-steps/s (one step = 4096 instance QP solves)",
        "instance_steps_per_s": inst_steps_per_s,
        "n_gpus": world, "rccl_ranks": joined, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "value_runs": [world * args.steps / t for t in region_s],
        "value_note": f"median of {TIMED_REPEATS} timed region(s) of exactly {args.steps} steps each (barrier + synchronize on both sides, maximum over ranks); value_runs = every region",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[1]: hover-linearised quadrotor nx=12 nu=4 N=30, 4096 instances per GPU, shared model, "
                               "x0 amplitude classes 0.3/1.0/3.0 interleaved, cold start every step",
                   "batch_per_gpu": BATCH_PER_GPU, "global_batch": world * BATCH_PER_GPU, "seed": hex(seed),
                   "setup_ramp_steps": RAMP_STEPS, "setup_ramp_note": "untimed steps of the same workload before the W warm-up steps (clock ramp and "
                                                                     "code-object load of a fresh box); not part of `warmup`",
                   "admm_max_iter": int(opts.max_iter), "check_every": int(opts.check_every), "polish": int(opts.polish),
                   "rho": opts.rho, "rho_profile": args.rho_profile, "eps": opts.eps_abs, "warm_state_kept": False, "parallelism": f"instances sharded over {world} GPU(s), 16-instance tile per workgroup"},
    }
    if comm_line is not None:
        out["rccl_in_library"] = comm_line   # ranks / unsolved / max iterations all-reduced by libalmpc.so's own communicator
    if rank == 0:
        # ---- rooflines from the HIP events recorded inside the timed region (one event set per step on the stream the
        # kernels run on).  Two kernels per step: k_admm (FP64 MFMA bound) and k_polish (+ fused rollout; dependent
        # chains per instance, its only hardware roofline is HBM).  `roofline` is the one that took more time.
        # k_step_fused: the event pair right before / right after the kernel (the "polish" slot of the fused path), every 16th timed step
        tall = np.concatenate([np.asarray(tsamp, dtype=np.float64), np.asarray(tsamp_after, dtype=np.float64)])
        fused_ms = float(np.median(tall)) if len(tall) else tsum["polish_ms"] / max(1, tsum["steps"])
        kernel_time_source = "hip_events_median"
        if fused_ms > 1e3 * elapsed / args.steps:   # a kernel cannot take longer than the step it is: event overhead / noise
            fused_ms = 1e3 * elapsed / args.steps
            kernel_time_source = "ms_per_step"
        stage_ms = {k: tsum2[k] / max(1, tsum2["steps"]) for k in ("admm_ms", "polish_ms", "rollout_ms", "total_ms")}
        iters_total = int(res["iters"].astype(np.int64).sum())
        traffic, traffic_src = {}, None
        for nm in ("r5_hbm_traffic.json", "r4_hbm_traffic.json", "r3_hbm_traffic.json", "r2_hbm_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", nm)) as f:
                    traffic = json.load(f)
                traffic_src = f"profiles/{nm} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of the same command; not measured in this run)"
                break
            except OSError:
                continue
        # k_admm, algorithmic flops per launch: per instance and iteration one nz x nz product (2 nz^2) plus ~10 nz of
        # vector work; once per instance the gradient f' = F' e0 and v0 = V e0 (2 nz n each)
        flops = iters_total * (2 * NZ * NZ + 10 * NZ) + BATCH_PER_GPU * (2 * NZ * NX + 2 * NZ * NX)
        admm_tflops = flops / (stage_ms["admm_ms"] * 1e-3) / 1e12
        roof_admm = {"bound": "mfma", "kernel": "k_admm<8,30>", "achieved": admm_tflops, "peak": FP64_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": admm_tflops / FP64_PEAK_TFLOPS,
                     "traffic": traffic.get("k_admm", {}).get("hbm_bytes_per_launch"),
                     "avg_kernel_ms": stage_ms["admm_ms"], "admm_iters_per_launch": iters_total,
                     "note": "FP64 MFMA v_mfma_f64_16x16x4_f64; peak = public-spec FP64 78.6 TFLOP/s (71-76 measured with "
                             "tools/microbench/f64_pipes.hip); traffic = PMC bytes per launch from profiles/r2_hbm_traffic.json"}
        # k_polish (+ rollout), algorithmic bytes per launch: z, y, v0 in (3 nz doubles), u, e_u, x, e_x out
        pol_bytes = BATCH_PER_GPU * 8 * (3 * NZ + 2 * NZ + 2 * NX * (N_HORIZON + 1))
        pol_ms = stage_ms["polish_ms"] + stage_ms["rollout_ms"]
        pol_gbs = pol_bytes / (pol_ms * 1e-3) / 1e9
        roof_polish = {"bound": "hbm", "kernel": "k_polish<true> (active-set polish with G in LDS + fused rollout)", "achieved": pol_gbs,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": pol_gbs / HBM_PEAK_GBS,
                       "traffic": traffic.get("k_polish", {}).get("hbm_bytes_per_launch"), "avg_kernel_ms": pol_ms,
                       "algorithmic_bytes_per_launch": pol_bytes,
                       "note": "latency bound: one dependent active-set chain per instance (one wave each); HBM is the only "
                               "hardware roofline it touches"}
        # the kernel of the headline run: all flops of the ADMM phase and all bytes of the step over ITS duration
        roof_fused = {"bound": "mfma", "kernel": "k_step_fused<8,30> (ADMM phase + active-set polish + rollout in one kernel)",
                      "achieved": flops / (fused_ms * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac": flops / (fused_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                      "traffic": traffic.get("k_step_fused", {}).get("hbm_bytes_per_launch"), "traffic_source": traffic_src,
                      "avg_kernel_ms": fused_ms, "kernel_time_source": kernel_time_source,
                      "kernel_ms_samples": {"n": int(len(tall)), "n_in_timed_region": int(len(tsamp)),
                                            "min": float(tall.min()) if len(tall) else None,
                                            "median": float(np.median(tall)) if len(tall) else None,
                                            "max": float(tall.max()) if len(tall) else None,
                                            "in_timed_region": [float(v) for v in tsamp],
                                            "note": f"HIP event pairs around the kernel: every {TIMING_STRIDE}th step of the timed region + every 2nd of 64 "
                                                    "steps right after it; avg_kernel_ms = the median of all pairs, capped by ms_per_step"},
                      "hbm_achieved_GBps": BATCH_PER_GPU * ALG_BYTES_PER_INSTANCE_STEP / (fused_ms * 1e-3) / 1e9,
                      "note": "FP64 MFMA is the only unit this kernel can saturate: the ADMM phase runs at roofline_kernels[0].frac of it, the "
                              "polish phase is a latency-bound dependent chain per instance (roofline_kernels[1]); phases timed on the "
                              "two-kernel path of the same build"}
        out["roofline"] = roof_fused
        out["roofline_kernels"] = [roof_admm, roof_polish]
        hbm_gbs = BATCH_PER_GPU * ALG_BYTES_PER_INSTANCE_STEP / (fused_ms * 1e-3) / 1e9
        cold_bytes = 8 * (NX + 2 * NZ + 2 * NX * (N_HORIZON + 1))   # x0 in; u, e_u, x, e_x out (shared references, no warm-start state)
        out["hbm"] = {"algorithmic_bytes_per_instance_step": ALG_BYTES_PER_INSTANCE_STEP, "achieved_GBps": hbm_gbs,
                      "peak_GBps": HBM_PEAK_GBS, "frac": hbm_gbs / HBM_PEAK_GBS,
                      "cold_start_bytes_per_instance_step": cold_bytes,
                      "cold_start_achieved_GBps": BATCH_PER_GPU * cold_bytes / (fused_ms * 1e-3) / 1e9,
                      "cold_start_note": "what THIS workload must move (warm_state_kept false: no z / y state in or out, e_x / e_u written as "
                                         "calculate! returns them); the 11,808 B of SURVEY.md section 8d include 2 x 2 nz doubles of warm-start state",
                      "note": "whole step (all kernels), SURVEY.md section 8d byte count; the shared-model path is FP64-compute / "
                              "latency bound, not HBM bound"}
        out["stage_ms"] = dict(stage_ms, fused_step_ms=fused_ms, sampled_steps=tsum2["steps"],
                               note=f"fused_step_ms: HIP events around k_step_fused on every {TIMING_STRIDE}th step of the timed region; the "
                                    "other entries: the two-kernel path (step fusion off), 64 steps after the timed region")
        out["solver"] = {"status_counts": np.bincount(res["status"], minlength=3).tolist(),
                         "admm_iters_mean": float(res["iters"].mean()), "polish_iters_mean": float(res["polish_iters"].mean()),
                         "polish_iters_max": int(res["polish_iters"].max())}
        # ---- parity on a sample of this very run: exact (KKT-certified) optimum from the oracle
        import mpc_oracle as mo
        po = mo.make_problem(p.A, p.B, N_HORIZON, p.u_min, p.u_max)   # the oracle's statement of the same problem (P = its own DARE)
        nchk = 96
        err = 0.0
        for i in range(nchk):
            e = mo.solve_mpc_exact(po, X0[i])
            err = max(err, float(np.abs(res["u"][i] - e["u"]).max()))
        out["u_err_inf"] = err
        out["u_err_sample"] = f"first {nchk} instances of rank 0 vs oracle exact optimum"
        _, _, Hq, _ = mo.condense(po)
        out["cond_H"] = float(np.linalg.cond(Hq))

    solver2 = None
    if not args.no_pipelined:
        # Secondary figure: two INDEPENDENT batches in flight on two handles / HIP streams of this rank (Monte-Carlo use:
        # batches do not depend on each other), which lets one batch's ADMM fill the CUs left idle by the other's
        # active-set tail.  Not the headline `value` (a closed loop has one batch in flight).
        solver2 = capi.Solver(NX, NU, N_HORIZON, BATCH_PER_GPU, device=dev_index)
        solver2.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, **design_kw)
        solver2.set_reference(p.x_ref, p.u_ref)
        solver2.update_initialization(make_x0(wl, first + world * BATCH_PER_GPU, BATCH_PER_GPU, seed=seed))
        pair = (solver, solver2)
        solver.timing_reset(args.steps + args.warmup)
        for i in range(2 * max(5, args.warmup // 5)):
            pair[i & 1].calculate(opts, sync=False)
        solver.synchronize(); solver2.synchronize()
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            pair[i & 1].calculate(opts, sync=False)
        solver.synchronize(); solver2.synchronize()
        barrier()
        el2 = max_over_ranks(time.perf_counter() - t0)
        out["two_batches_in_flight"] = {"value": world * args.steps / el2, "unit": "batch-steps/s",
                                        "note": "independent batches alternated over two handles/streams per GPU"}

    if not args.no_closed_loop:
        # Secondary figure: receding-horizon closed loop on the device (x0 <- A x0 + B u[:,1], warm-started ADMM), the
        # mixed batch driven for 100 consecutive steps from its initial states (closed-loop variant)
        solver.update_initialization(X0)
        warm = capi.default_opts(rho=rho, max_iter=args.max_iter, check_every=args.max_iter, warm_start=1)
        solver.timing_reset(0)
        solver.timing_set_stride(1 << 30)  # no events in this section
        T = 100
        elc = float("inf")
        for _rep in range(3):   # best of three repetitions of the SAME 100 steps (x0 reset and the cold first step redone each time: an
            solver.update_initialization(X0)   # occasional ~20 ms host hiccup on this pool would otherwise swallow a 6 ms stretch)
            solver.calculate(opts_keep, sync=False)
            solver.advance_plant()
            barrier(); solver.synchronize()
            t0 = time.perf_counter()
            for _ in range(T):
                solver.calculate(warm, sync=False)
                solver.advance_plant()
            solver.synchronize(); barrier()
            elc = min(elc, max_over_ranks(time.perf_counter() - t0))
        rc_ = solver.get_results(want=("status", "polish_iters", "x"))
        out["closed_loop"] = {"value": world * T / elc, "unit": "batch-steps/s", "steps": T,
                              "redo": "gated: almpc_advance_plant enqueues the redo of what the step's finish left undecided behind the step (empty launches "
                                      "when it left nothing), so every plant step is driven by a decided input",
                              "status_counts_last": np.bincount(rc_["status"], minlength=4).tolist(),
                              "max_abs_position_last": float(np.abs(rc_["x"][:, :3, 0]).max())}
        solver.timing_set_stride(TIMING_STRIDE)
        solver.update_initialization(X0)

    if rank == 0 and world == 1 and not args.no_api_path:
        out["api_path"] = api_path_figures(capi, solver, X0, opts, args.steps)
        solver.timing_set_stride(TIMING_STRIDE)
        solver.update_initialization(X0)

    if not args.no_classes:
        # per-class rates (each class alone on the whole batch), short runs
        cls = {}
        for s_ in AMPLITUDES:
            solver.update_initialization(make_x0(wl, first, BATCH_PER_GPU, amplitude=s_, seed=seed))
            k = max(20, args.steps // 5)
            solver.timing_reset(3 * k + args.warmup)  # no event creation inside the timed loop
            time_steps(solver, opts, max(5, args.warmup // 5), barrier)
            # best of three bursts: a burst is a few milliseconds long, and an occasional ~15-30 ms hiccup of the host-side wait
            # (seen on this pool) would otherwise swallow the figure
            el = min(max_over_ranks(time_steps(solver, opts, k, barrier)) for _ in range(3))
            cls[str(s_)] = world * k / el
        out["classes"] = {"unit": "batch-steps/s", **cls}

    if rank == 0 and world == 1 and not args.no_batched_models:
        # Secondary figure: the per-instance-model regime (SURVEY.md section 8d "per-instance-model regime", BASELINE configs[3]
        # at the headline shape): every instance has its own (A_i, B_i) -> its own KKT inverse, which k_admm_inst streams from
        # HBM once per step.  This is the regime where the HBM roofline is the binding one.
        rng = np.random.default_rng(0)
        Ab = np.repeat(p.A[None], BATCH_PER_GPU, 0)
        Bb = np.repeat(p.B[None], BATCH_PER_GPU, 0) * (1.0 + 0.05 * rng.standard_normal((BATCH_PER_GPU, 1, 1)))
        sb = capi.Solver(NX, NU, N_HORIZON, BATCH_PER_GPU, device=dev_index, timing=True)
        t0 = time.perf_counter()
        P_lib = solver.get_design()["P"]   # the library's DARE of the nominal model, shared by the perturbed ones
        sb.design_batched(Ab, Bb, p.Q, p.R, p.S, P_lib, p.u_min, p.u_max, **design_kw)
        t_design_first = time.perf_counter() - t0    # includes the one-time allocation of ~2.5 GB of per-instance operands
        t_design = float("inf")
        for _ in range(3):                           # steady state: what a re-design (per-step re-linearisation) costs
            t0 = time.perf_counter()
            sb.design_batched(Ab, Bb, p.Q, p.R, p.S, P_lib, p.u_min, p.u_max, **design_kw)
            t_design = min(t_design, time.perf_counter() - t0)
        sb.set_reference(p.x_ref, p.u_ref)
        sb.update_initialization(X0)
        for _ in range(5):
            sb.calculate(opts)
        kb = 50
        elb = float("inf")
        for _rep in range(3):  # best of three bursts: a burst is only ~12 ms long and an occasional ~15 ms hiccup of the host-side wait
            sb.timing_reset(kb)  # (seen on this pool) would otherwise double the figure
            t0 = time.perf_counter()
            for _ in range(kb):
                sb.calculate(opts, sync=False)
            sb.synchronize()
            elb = min(elb, time.perf_counter() - t0)
        tb = sb.timing_summary()
        rb = sb.get_results(want=("status",))
        nzs_ = 16 * ((NZ + 15) // 16)
        # k_admm_inst<true>, algorithmic bytes per instance (round 5): the packed lower triangle of the symmetric M_i^-1 -- nz (nz + 1) / 2
        # entries + one pad per even column (packed_tri_doubles: 7,320 doubles at nz 120 against the 15,360 of the full nz x nzs layout
        # that rounds 1 - 4 streamed) --, F'_i and V_i (n x nzs each), d, rho, fS, v0S, x0 in; x, z, y, v0 out
        tri_ = (NZ * (NZ + 1) // 2 + (NZ + 1) // 2 + 1) & ~1 if NZ % 2 == 0 else NZ * nzs_
        admm_bytes = BATCH_PER_GPU * 8 * (tri_ + 2 * NX * nzs_ + 2 * nzs_ + 2 * NZ + NX + 4 * nzs_)
        admm_bytes_full = BATCH_PER_GPU * 8 * (NZ * nzs_ + 2 * NX * nzs_ + 2 * nzs_ + 2 * NZ + NX + 4 * nzs_)
        admm_ms_b = tb["admm_ms"] / max(1, tb["steps"])
        out["per_instance_models"] = {
            "value": kb / elb, "unit": "batch-steps/s (4096 instances, one model per instance)", "ms_per_step": 1e3 * elb / kb,
            "design_ms": 1e3 * t_design, "first_design_ms": 1e3 * t_design_first, "status_counts": np.bincount(rb["status"], minlength=3).tolist(),
            "stage_ms": {k: tb[k] / max(1, tb["steps"]) for k in ("admm_ms", "polish_ms", "total_ms")},
            "roofline": {"bound": "hbm", "kernel": "k_admm_inst<packed> (packed triangle of the KKT inverse streamed HBM -> LDS once per instance-step, gathered into registers)",
                         "algorithmic_bytes_per_launch_full_matrix_layout": admm_bytes_full,
                         "note": "half the stream of rounds 1 - 4 at the same kernel time: the kernel is no longer bound by HBM but by the per-instance chain "
                                 "(gather, gradient, K iterations of ~2.3 k cycles, hand-off) of two workgroups per CU -- frac is of the HBM peak all the same",
                         "achieved": admm_bytes / (admm_ms_b * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": admm_bytes / (admm_ms_b * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic.get("k_admm_inst", {}).get("hbm_bytes_per_launch") if rank == 0 else None,
                         "avg_kernel_ms": admm_ms_b, "algorithmic_bytes_per_launch": admm_bytes},
            "note": "models = the benchmark plant with B scaled per instance (+-5 %), shared P, same x0 and options as the headline run"}
        sb.close()

    if rank == 0 and world == 1 and not args.no_relin:
        # Secondary figure: BASELINE configs[3] -- Fnn model (4-2-16x2 relu), N = 20, batch = 1024, RE-LINEARISED EVERY STEP at each instance's
        # own state: k_fnn_jacobian -> per-instance condensed designs (H_i, scaling, two 40 x 40 inverses, V_i) -> k_admm_inst -> polish,
        # all on the handle's stream without host pointers (almpc_relin_fnn_*).
        b3, n3, m3, N3 = 1024, 4, 2, 20
        W_in, W_h, b_h, W_out = wl.synthetic_fnn_weights(n3, m3)
        A0, _ = capi.fnn_linearize(W_in, W_h, b_h, W_out, np.zeros((1, n3)), np.zeros((1, m3)), act="relu", device=dev_index)
        W_out = wl.scale_to_radius(W_out, A0[0])
        xr3 = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N3 + 1)); ur3 = np.tile(np.array([0.1, -0.2])[:, None], (1, N3))
        X03 = xr3[:, 0][None, :] + wl.splitmix_normal(0x5EED0004, 0, b3, n3)
        Q3, R3 = 100.0 * np.eye(n3), 0.1 * np.eye(m3)
        Al3, Bl3 = capi.fnn_linearize(W_in, W_h, b_h, W_out, xr3[:, -1][None, :], ur3[:, -1][None, :], act="relu", device=dev_index)
        P3 = capi.dare(Al3[0], Bl3[0], Q3, R3)   # terminal weight as the reference takes it: linearisation at the last reference
        # (no ALMPC_FLAG_TIMING on the measured handle: the four event records per step cost ~14 us of stream time at this step size;
        # the per-stage times come from a second handle with the flag)
        s3 = capi.Solver(n3, m3, N3, b3, device=dev_index)
        s3.relin_fnn_setup(W_in, W_h, b_h, W_out, xr3, ur3, Q3, R3, None, P3, -np.ones(m3), np.ones(m3), act="relu")
        s3.update_initialization(X03)
        s3t = capi.Solver(n3, m3, N3, b3, device=dev_index, timing=True)
        s3t.relin_fnn_setup(W_in, W_h, b_h, W_out, xr3, ur3, Q3, R3, None, P3, -np.ones(m3), np.ones(m3), act="relu")
        s3t.update_initialization(X03)
        o3 = capi.default_opts(max_iter=args.relin_max_iter, check_every=args.relin_max_iter)
        for _ in range(3):
            s3t.relin_fnn_step(o3)
        t3 = s3t.relin_fnn_timing()
        s3t.relin_fnn_advance()
        for _ in range(3):
            s3t.relin_fnn_step(capi.default_opts(warm_start=1))
        t3w = s3t.relin_fnn_timing()
        s3t.close()
        for _ in range(5):
            s3.relin_fnn_step(o3)
        k3, best3 = 50, float("inf")
        for _rep in range(3):
            t0 = time.perf_counter()
            for _ in range(k3):
                s3.relin_fnn_step(o3, sync=False)
            s3.synchronize()
            best3 = min(best3, time.perf_counter() - t0)
        r3 = s3.get_results(want=("status", "u", "polish_iters"))
        best3d, st3d = None, None
        if int(o3.max_iter) != int(capi.default_opts().max_iter):   # the same step at the library's default operating point, for comparison
            o3d = capi.default_opts()
            for _ in range(5):
                s3.relin_fnn_step(o3d)
            best3d = float("inf")
            for _rep in range(3):
                t0 = time.perf_counter()
                for _ in range(k3):
                    s3.relin_fnn_step(o3d, sync=False)
                s3.synchronize()
                best3d = min(best3d, time.perf_counter() - t0)
            st3d = s3.get_results(want=("status",))["status"]
        # the same pipeline in closed loop on the network itself (x0 <- fnn(x0, u[:,1]) on the device), warm steps: working-set guess
        # from the previous inputs shifted one stage, no ADMM phase, one inverse per design.  40 steps from X03, best of 3.
        o3w = capi.default_opts(warm_start=1)
        kcl3, bestcl3, st_cl3 = 40, float("inf"), None
        for _rep in range(3):
            s3.update_initialization(X03)
            s3.relin_fnn_step(o3)
            s3.synchronize()
            t0 = time.perf_counter()
            for _ in range(kcl3):
                s3.relin_fnn_advance()
                s3.relin_fnn_step(o3w, sync=False)
            s3.synchronize()
            bestcl3 = min(bestcl3, time.perf_counter() - t0)
            st_cl3 = s3.get_results(want=("status",))["status"]
        import mpc_oracle as mo   # checker: exact optimum of sampled instances' own QPs
        fo3 = mo.FnnModel(W_in, W_h, b_h, W_out, "relu")
        err3 = 0.0
        for i in range(0, b3, 64):
            if r3["status"][i] == 0:
                Ai, Bi = fo3.jacobian(X03[i], ur3[:, 0])
                pi = mo.make_problem(Ai, Bi, N3, -np.ones(m3), np.ones(m3), x_ref=xr3, u_ref=ur3, P=P3)
                err3 = max(err3, float(np.abs(r3["u"][i] - mo.solve_mpc_exact(pi, X03[i])["u"]).max()))
        out["config3_fnn_relin"] = {"value": k3 / best3, "unit": "batch-steps/s (1024 instances, Fnn 4-2-16x2 relu, N=20, re-linearised every step)",
                                    "ms_per_step": 1e3 * best3 / k3, "instance_steps_per_s": k3 * b3 / best3,
                                    "stage_ms": t3, "status_counts": np.bincount(r3["status"], minlength=3).tolist(),
                                    "polish_iters_max": int(r3["polish_iters"].max()), "u_err_inf_sampled": err3,
                                    "admm_max_iter": int(o3.max_iter),
                                    "library_default_K25": None if best3d is None else {"value": k3 / best3d, "ms_per_step": 1e3 * best3d / k3,
                                                                                        "status_counts": np.bincount(st3d, minlength=3).tolist()},
                                    "closed_loop_warm": {"value": kcl3 / bestcl3, "ms_per_step": 1e3 * bestcl3 / kcl3, "stage_ms": t3w,
                                                         "status_counts_last": np.bincount(st_cl3, minlength=3).tolist(),
                                                         "note": "plant = the network (almpc_relin_fnn_advance), opts.warm_start = 1: "
                                                                 "shifted-previous-inputs guess, no ADMM phase, one inverse per design"},
                                    "note": "one step = Jacobians at (x0_i, u_ref[:,1]) + per-instance condensed designs + ADMM + polish on the "
                                            "handle's stream (almpc_relin_fnn_step); stage_ms: HIP events of a step on a second handle "
                                            "created with ALMPC_FLAG_TIMING"}
        # roofline of the figure (per-instance-model regime): flops of one instance-step = A^k stack 2 n^3 N + Gamma
        # blocks 2 n^2 m N + Gamma' Qbar Gamma (block-triangular: ~ 2/3 nz nN nz) + two nz x nz inverses 2 * 2 nz^3 + V_i 2 nz^2 n +
        # K ADMM iterations 2 nz^2 each; bytes = 13,344-type vectors (here 8 (n + 2 nz + 2 n (N+1)) + 8 (n^2 + n m)) + the materialised
        # H_i, H'_i, G_i, M_i^-1 (written by the design, read by the step: 2 * 4 * 8 nz nzs)
        nz3, nzs3 = m3 * N3, 16 * ((m3 * N3 + 15) // 16)
        k_admm3 = int(o3.max_iter)
        fl3 = (2 * n3 ** 3 * N3 + 2 * n3 * n3 * m3 * N3 + (2.0 / 3.0) * 2 * nz3 * n3 * N3 * nz3 + 4.0 * nz3 ** 3 + 2 * nz3 * nz3 * n3
               + k_admm3 * 2 * nz3 * nz3)
        by3 = 8 * (n3 + 2 * nz3 + 2 * n3 * (N3 + 1)) + 8 * (n3 * n3 + n3 * m3) + 2 * 4 * 8 * nz3 * nzs3
        sec3 = best3 / k3
        out["config3_fnn_relin"]["roofline"] = {
            "bound": "hbm", "kernel": "k_step_inst_wave<48> + 2 x k_design_inverse_wave<48> (28 us) + k_design_instance_t + k_fnn_jacobian_w + 2 small design launches per step",
            "achieved": b3 * by3 / sec3 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b3 * by3 / sec3 / 1e9 / HBM_PEAK_GBS,
            "traffic": None, "algorithmic_bytes_per_instance_step": by3, "algorithmic_flops_per_instance_step": fl3,
            "fp64_tflops": b3 * fl3 / sec3 / 1e12, "fp64_frac": b3 * fl3 / sec3 / 1e12 / FP64_PEAK_TFLOPS,
            "note": "1024 instances of a 40 x 40 problem are 4 instances per CU: every kernel of the chain is bound by the latency of its own "
                    "dependent steps (40 pivots per inverse, the ADMM iterations, the active-set changes), far from both rooflines; the "
                    "floor of this shape is the sum of those chains (~0.15 ms: 40 pivots at ~1 k cycles per inverse, the ADMM iterations + the finish in "
                    "k_step_inst_wave), not bytes or flops"}
        s3.close()
        # The same pipeline beyond the condensed horizon (round 5): a quadrotor-size network (n 12, m 4, 24 x 2 tanh) at N = 50 --
        # m N = 200: no condensed handle exists -- on an ALMPC_FLAG_STRUCTURED handle: Jacobians -> k_sgains -> k_sdual, 256 instances
        n5, m5, N5, b5 = 12, 4, 50, 256
        Wi5, Wh5, bh5, Wo5 = wl.synthetic_fnn_weights(n5, m5, H=24, L=2, seed=0x5EED0044)
        A05, _ = capi.fnn_linearize(Wi5, Wh5, bh5, Wo5, np.zeros((1, n5)), np.zeros((1, m5)), act="tanh", device=dev_index)
        Wo5 = wl.scale_to_radius(Wo5, A05[0])
        Al5, Bl5 = capi.fnn_linearize(Wi5, Wh5, bh5, Wo5, np.zeros((1, n5)), np.zeros((1, m5)), act="tanh", device=dev_index)
        P5 = capi.dare(Al5[0], Bl5[0], 100.0 * np.eye(n5), 0.1 * np.eye(m5))
        X05 = 0.5 * wl.splitmix_normal(0x5EED0045, 0, b5, n5)
        s5 = capi.Solver(n5, m5, N5, b5, device=dev_index, structured=True)
        s5.relin_fnn_setup(Wi5, Wh5, bh5, Wo5, np.zeros((n5, N5 + 1)), np.zeros((m5, N5)), 100.0 * np.eye(n5), 0.1 * np.eye(m5), None, P5,
                           -0.3 * np.ones(m5), 0.3 * np.ones(m5), act="tanh")
        s5.update_initialization(X05)
        s5.relin_fnn_step()
        best5 = float("inf")
        for _rep in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                s5.relin_fnn_step(sync=False)
            s5.synchronize()
            best5 = min(best5, (time.perf_counter() - t0) / 10)
        r5 = s5.get_results(want=("u", "status", "polish_iters"))
        fo5 = mo.FnnModel(Wi5, Wh5, bh5, Wo5, "tanh")
        err5 = 0.0
        for i in range(0, b5, 32):
            Ai, Bi = fo5.jacobian(X05[i], np.zeros(m5))
            pi = mo.make_problem(Ai, Bi, N5, -0.3 * np.ones(m5), 0.3 * np.ones(m5), P=P5)
            err5 = max(err5, float(np.abs(r5["u"][i] - mo.solve_mpc_structured(pi, X05[i])["u"]).max()))
        bestw5 = float("inf")
        o5w = capi.default_opts(warm_start=1)
        for _rep in range(3):
            s5.update_initialization(X05)
            s5.relin_fnn_step()
            t0 = time.perf_counter()
            for _ in range(20):
                s5.relin_fnn_advance()
                s5.relin_fnn_step(o5w, sync=False)
            s5.synchronize()
            bestw5 = min(bestw5, (time.perf_counter() - t0) / 20)
        st5w = s5.get_results(want=("status",))["status"]
        s5.close()
        out["relin_structured_N50"] = {"value": 1.0 / best5, "unit": "batch-steps/s (256 instances, Fnn 12-4-24x2 tanh, N=50, re-linearised every step, stage-wise solve)",
                                       "ms_per_step": 1e3 * best5, "status_counts": np.bincount(r5["status"], minlength=4).tolist(),
                                       "working_set_changes_mean": float(r5["polish_iters"].mean()), "working_set_changes_max": int(r5["polish_iters"].max()),
                                       "u_err_inf_sampled": err5,
                                       "closed_loop_warm": {"ms_per_step": 1e3 * bestw5, "status_counts_last": np.bincount(st5w, minlength=4).tolist()},
                                       "note": "almpc_relin_fnn_step on an ALMPC_FLAG_STRUCTURED handle: k_fnn_jacobian_w -> k_sgains (Riccati recursion of "
                                               "every instance's unconstrained problem) -> k_sdual; m N = 200 is beyond the condensed kernels"}

    if rank == 0 and world == 1 and not args.no_structured:
        # Secondary figure: the structured (non-condensed) solve -- the benchmark plant over a horizon the condensed kernels cannot take
        # (N = 50: m N = 200 > 128), 4096 instances, mixed amplitudes.  k_sdual: dual active set in constraint space whose Ghat columns
        # are affine Riccati sweeps with the unconstrained gains (one wave per instance, two sweeps per working-set change).
        Ns = 50
        ps = wl.quadrotor(Ns)
        ss = capi.Solver(NX, NU, Ns, BATCH_PER_GPU, device=dev_index, structured=True)
        ss.design_shared(ps.A, ps.B, ps.Q, ps.R, ps.S, None, ps.u_min, ps.u_max)
        ss.set_reference(ps.x_ref, ps.u_ref)
        ss.update_initialization(X0)
        ss.calculate()

        def time_structured(sv, ks=5):
            best = float("inf")
            for _rep in range(3):
                t0 = time.perf_counter()
                for _ in range(ks):
                    sv.calculate(sync=False)
                sv.synchronize()
                best = min(best, time.perf_counter() - t0)
            return best / ks
        bests = time_structured(ss)
        ks = 1
        rs = ss.get_results(want=("status", "polish_iters", "u"))
        # horizon continuation (almpc_set_start_from): the condensed N = 30 step (the headline kernel) hands the N = 50 stage-wise solve
        # its working set; one "step" = both solves, from a cold start
        solver.update_initialization(X0)
        kc, bestc = 20, float("inf")
        for _rep in range(3):
            t0 = time.perf_counter()
            for _ in range(kc):
                solver.calculate(opts, sync=False)
                ss.start_from(solver)
                ss.calculate(sync=False)
                ss.synchronize()    # (one step at a time: the next condensed step does not run under this step's solve on the other stream)
            bestc = min(bestc, time.perf_counter() - t0)
        rcn = ss.get_results(want=("status", "polish_iters", "u"))
        import mpc_oracle as mo
        pso = mo.make_problem(ps.A, ps.B, Ns, ps.u_min, ps.u_max)
        errs = max(float(np.abs(rs["u"][i] - mo.solve_mpc_exact(pso, X0[i])["u"]).max()) for i in range(0, 96, 8))
        errc = max(float(np.abs(rcn["u"][i] - mo.solve_mpc_exact(pso, X0[i])["u"]).max()) for i in range(0, 96, 8))
        # the rows the reference's stage-wise form carries beyond the input box: state box on every stage (..linear.jl:62-70), terminal
        # equality (src/sub/design_mpc.jl:330-331), input-rate weight (src/sub/design_mpc.jl:423-446) -- same batch, same horizon
        xbox = np.array([3, 3, 3, 1.5, 1.5, 1.5, 0.3, 0.3, 0.3, 1.0, 1.0, 1.0])
        rows_out = {}
        for name, kwd in (("state_box", dict(xmin=-xbox, xmax=xbox)), ("terminal_equality", dict(terminal="equality")),
                          ("input_rate_weight_S5", dict(S=5.0 * np.eye(NU)))):
            sv = capi.Solver(NX, NU, Ns, BATCH_PER_GPU, device=dev_index, structured=True)
            sv.design_shared(ps.A, ps.B, ps.Q, ps.R, kwd.get("S", ps.S), None, ps.u_min, ps.u_max, xmin=kwd.get("xmin"), xmax=kwd.get("xmax"),
                             terminal=kwd.get("terminal", "none"))
            sv.set_reference(ps.x_ref, ps.u_ref)
            Xv = np.clip(X0, -0.99 * xbox, 0.99 * xbox) if "xmin" in kwd else X0
            sv.update_initialization(Xv)
            sv.calculate()
            tv = time_structured(sv, 3)
            rv = sv.get_results(want=("status", "polish_iters", "u"))
            pv = mo.make_problem(ps.A, ps.B, Ns, ps.u_min, ps.u_max, x_min=kwd.get("xmin"), x_max=kwd.get("xmax"), terminal=kwd.get("terminal", "none"),
                                 s=5.0 if "S" in kwd else 0.0)
            ev, ninf = 0.0, 0
            # sample: stride 7 mixes the three amplitude classes (instance index mod 3); plus the first four instances the kernel calls
            # infeasible, so that the verdicts are part of this line's own evidence
            sample = list(range(0, 96, 7)) + [int(i) for i in np.nonzero(rv["status"] == 3)[0][:4]]
            for i in sample:
                try:
                    ev = max(ev, float(np.abs(rv["u"][i] - mo.solve_mpc_exact(pv, Xv[i])["u"]).max()))
                except ValueError:   # infeasible by the oracle: the kernel must say so too
                    ninf += 1
                    ev = max(ev, 0.0 if rv["status"][i] == 3 else float("inf"))
            rows_out[name] = {"ms_per_step": 1e3 * tv, "status_counts": np.bincount(rv["status"], minlength=4).tolist(),
                              "working_set_changes_mean": float(rv["polish_iters"].mean()), "working_set_changes_max": int(rv["polish_iters"].max()),
                              "u_err_inf_sampled": ev, "sampled_infeasible_agree": ninf, "sampled_instances": len(sample),
                              "sampled_kernel_infeasible": int(sum(1 for i in sample if rv["status"][i] == 3))}
            sv.close()
        # k_sdual, algorithmic flops: a sweep is N stages x (backward + forward) x (n + m)^2 multiply-adds; an instance with c scan / step
        # iterations sweeps about 2 + 2 c times (unconstrained solution, confirmation, response + direction per change)
        fl_sweep = Ns * 2 * 2.0 * (NX + NU) ** 2
        fl = BATCH_PER_GPU * fl_sweep * (2.0 + 2.0 * float(rs["polish_iters"].mean()))
        out["structured_N50"] = {"value": ks / bests, "unit": "batch-steps/s (4096 instances, quadrotor nx=12 nu=4, N=50, stage-wise dual active-set solve)",
                                 "ms_per_step": 1e3 * bests / ks, "status_counts": np.bincount(rs["status"], minlength=3).tolist(),
                                 "working_set_changes_mean": float(rs["polish_iters"].mean()), "working_set_changes_max": int(rs["polish_iters"].max()),
                                 "u_err_inf_sampled": errs,
                                 "continued_from_N30": {"value": kc / bestc, "ms_per_step": 1e3 * bestc / kc,
                                                        "status_counts": np.bincount(rcn["status"], minlength=3).tolist(),
                                                        "working_set_changes_mean": float(rcn["polish_iters"].mean()),
                                                        "working_set_changes_max": int(rcn["polish_iters"].max()), "u_err_inf_sampled": errc,
                                                        "note": "one step = the condensed N = 30 step + almpc_set_start_from + the N = 50 stage-wise solve"},
                                 "with_rows": rows_out,
                                 "roofline": {"bound": "mfma", "kernel": "k_sdual<12,4,1> (FP64 vector pipeline: same 78.6 TFLOP/s peak as the matrix cores)", "achieved": fl / bests / 1e12, "peak": FP64_PEAK_TFLOPS,
                                              "unit": "TFLOP/s", "frac": fl / bests / 1e12 / FP64_PEAK_TFLOPS,
                                              "flop_model": "MODEL figure, not executed work: (2 + 2 c) sweeps per instance with c working-set changes is what the "
                                                            "sweep-only build (ALMPC_SDUAL_NO_GHAT=1) executes; the default build of a shared model takes a change's two "
                                                            "sweeps from the cached response table (a stream of |W| + 1 columns of TP doubles from L2) and sweeps only for "
                                                            "the start, the confirmation and refinements",
                                              "note": "one wave per instance, a stage of a sweep = 16 DPP-broadcast FMAs on a row of 16 lanes (~330 cycles measured with two "
                                                      "waves per SIMD): a dependent chain of N stages per sweep and two sweeps per working-set change -- latency bound, the "
                                                      "launch ends with its slowest instance (max changes); flops = sweeps x N x 4 (n + m)^2"}}
        ss.close()

    if rank == 0 and world == 1 and not args.no_small_shared:
        # Secondary figure: the reference's own test size (test/computation_mpc_test.jl:981-1054: quadruple-tank fixture, n 4, m 2, N 5 --
        # nz 10), library defaults.  One instance = the reference's call (latency), 512 and 65,536 Monte-Carlo initial states around the
        # reference (throughput).  Up to two instances per CU run ONE kernel per step, one wave per instance (k_step_inst_wave on the
        # shared operands, round 5); larger batches the tile path (k_admm's 16-instance MFMA tile + k_polish) -- each leg also on the
        # other path (ALMPC_NO_SHARED_WAVE / ALMPC_SHARED_WAVE_MAX_BATCH) for comparison.
        pq = wl.qtp_fixture()
        import mpc_oracle as mo
        pqo = mo.make_problem(pq.A, pq.B, pq.N, pq.u_min, pq.u_max, x_ref=pq.x_ref[:, 0], u_ref=pq.u_ref[:, 0])
        legs_q = {}
        for bq in (1, 512, 65536):
            Xq = 0.65 + 0.25 * wl.splitmix_normal(0x5EED0051, 0, bq, 4)
            if bq == 1:
                Xq[0] = 0.6          # the reference's own initial state (test/computation_mpc_test.jl:1040)
            row = {}
            for nm_, env_ in (("default_path", {}), ("two_launch_path", {"ALMPC_NO_SHARED_WAVE": "1"}), ("one_wave_per_instance", {"ALMPC_SHARED_WAVE_MAX_BATCH": "1000000000"})):
                os.environ.update(env_)
                try:
                    sq_ = capi.Solver(pq.n, pq.m, pq.N, bq, device=dev_index)
                    sq_.design_shared(pq.A, pq.B, pq.Q, pq.R, pq.S, None, pq.u_min, pq.u_max)
                    sq_.set_reference(pq.x_ref, pq.u_ref)
                    sq_.update_initialization(Xq)
                    for _ in range(5):
                        sq_.calculate()
                    best = float("inf")
                    for _rep in range(3):
                        t0 = time.perf_counter()
                        for _ in range(50):
                            sq_.calculate(sync=False)
                        sq_.synchronize()
                        best = min(best, (time.perf_counter() - t0) / 50)
                    rq = sq_.get_results(want=("u", "status"))
                    sq_.close()
                finally:
                    for k_ in env_:
                        os.environ.pop(k_, None)
                row[nm_] = {"us_per_step": 1e6 * best, "instance_steps_per_s": bq / best, "unsolved": int((rq["status"] != 0).sum())}
                if nm_ == "default_path":
                    row["u_err_inf_sampled"] = max(float(np.abs(rq["u"][i] - mo.solve_mpc_exact(pqo, Xq[i])["u"]).max()) for i in range(0, bq, 4099))
            legs_q["batch_%d" % bq] = row
        out["small_shared_qtp"] = dict(legs_q, workload="quadruple-tank fixture of the reference's tests, n 4, m 2, N 5 (nz 10), library defaults (rho 0.1, K <= 25); "
                                                        "batch 1: x0 = 0.6 as the reference's test")

    if rank == 0 and world == 1 and not args.no_state_rows:
        # Secondary figure: the reference's state box on every stage (mpc_state_constraint, ..linear.jl:62-70) + terminal equality
        # (src/sub/design_mpc.jl:330-331) on the condensed N = 30 handle: 4096 quadrotor instances, amplitude 1, the TIGHT box (3 x the
        # x0 scale), x0 clipped to 0.99 of it -- a third of the batch is infeasible, a dozen instances sit at the edge of feasibility
        # and leave the condensed finish undecided.  Two ways of driving it, each with its status counts:
        #   no_look          20 steps enqueued, ONE synchronous look at the end (the lazy redo of the undecided instances runs once);
        #   look_every_step  a receding-horizon caller: almpc_calculate (synchronous) + status read-back every step -- every step pays
        #                    the stage-wise redo of what its finish left undecided.
        xs_ = np.array([1, 1, 1, .5, .5, .5, .1, .1, .1, .1, .1, .1]) * 3.0
        Xr = np.clip(make_x0(wl, 0, BATCH_PER_GPU, amplitude=1.0), -0.99 * xs_, 0.99 * xs_)
        sr = capi.Solver(NX, NU, N_HORIZON, BATCH_PER_GPU, device=dev_index)
        sr.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=-xs_, xmax=xs_, rho=30.0, rho_profile="stiffness", terminal="equality")
        sr.set_reference(p.x_ref, p.u_ref)
        sr.update_initialization(Xr)
        orr = capi.default_opts(rho=30.0, max_iter=8, check_every=8)
        for _ in range(3):
            sr.calculate(orr)
        legs = {}
        best = float("inf")
        for _rep in range(3):
            t0 = time.perf_counter()
            for _ in range(20):
                sr.calculate(orr, sync=False)
            sr.synchronize()
            best = min(best, (time.perf_counter() - t0) / 20)
        legs["no_look"] = {"ms_per_step": 1e3 * best, "status_counts": np.bincount(sr.get_results(want=("status",))["status"], minlength=4).tolist()}
        best = float("inf")
        for _rep in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                sr.calculate(orr)
                st_ = sr.get_results(want=("status",))["status"]
            best = min(best, (time.perf_counter() - t0) / 10)
        legs["look_every_step"] = {"ms_per_step": 1e3 * best, "status_counts": np.bincount(st_, minlength=4).tolist()}
        # the same loop with the redo off: what the condensed finish alone leaves, and what the redo costs per look
        so = capi.Solver(NX, NU, N_HORIZON, BATCH_PER_GPU, device=dev_index, structured_fallback=False)
        so.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=-xs_, xmax=xs_, rho=30.0, rho_profile="stiffness", terminal="equality")
        so.set_reference(p.x_ref, p.u_ref)
        so.update_initialization(Xr)
        for _ in range(3):
            so.calculate(orr)
        best = float("inf")
        for _rep in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                so.calculate(orr)
                st0 = so.get_results(want=("status",))["status"]
            best = min(best, (time.perf_counter() - t0) / 10)
        legs["look_every_step_redo_off"] = {"ms_per_step": 1e3 * best, "status_counts": np.bincount(st0, minlength=4).tolist()}
        so.close()
        sr.close()
        out["state_rows_N30"] = dict(legs, unit="ms per step of 4096 instances", workload="quadrotor N = 30, state box 3 x the x0 scale on every stage + terminal equality, amplitude 1, x0 clipped to 0.99 of the box",
                                     status_legend="[solved, undecided (max_iter), non-finite, infeasible]")

    if rank == 0 and world == 1 and not args.no_sqp:
        # Secondary figure: BASELINE configs[4] -- the NLP of the reference's NonLinearProgramming branch for an Fnn model
        # (N = 50, batch = 256) through the device-resident SQP loop: one iteration = Jacobians of 256 x 50 stages, the LTV
        # condensed QP (H_i in LDS, two 100 x 100 inverses), ADMM + polish per instance, trajectory update.
        bq, nq, mq, Nq = 256, 4, 2, 50
        W_in, W_h, b_h, W_out = wl.synthetic_fnn_weights(nq, mq)
        A0, _ = capi.fnn_linearize(W_in, W_h, b_h, W_out, np.zeros((1, nq)), np.zeros((1, mq)), act="tanh", device=dev_index)
        W_out = wl.scale_to_radius(W_out, A0[0])   # spectral radius 0.95 at the origin
        f = pkg.Fnn(W_in, W_h, b_h, W_out, "tanh")
        xr = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, Nq + 1)); ur = np.tile(np.array([0.1, -0.2])[:, None], (1, Nq))
        X0q = xr[:, 0][None, :] + 0.6 * wl.splitmix_normal(0x5EED0005, 0, bq, nq)
        Al, Bl = capi.fnn_linearize(W_in, W_h, b_h, W_out, xr[:, -1][None, :], ur[:, -1][None, :], act="tanh", device=dev_index)
        Pq = capi.dare(Al[0], Bl[0], 100.0 * np.eye(nq), 0.1 * np.eye(mq))   # terminal weight as the reference takes it (last reference)
        sq = capi.Solver(nq, mq, Nq, bq, device=dev_index)
        sq.sqp_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, xr, ur, 100.0 * np.eye(nq), 0.1 * np.eye(mq), None, Pq, -np.ones(mq), np.ones(mq),
                         act="tanh")
        sq.sqp_fnn_start(X0q)
        sq.sqp_fnn_iterate(3)  # warm-up (module load, LDS attributes)
        best, its = float("inf"), 20
        for _rep in range(3):
            sq.sqp_fnn_start(X0q)
            t0 = time.perf_counter()
            st_, de_ = sq.sqp_fnn_iterate(its, step_rule="merit")
            best = min(best, time.perf_counter() - t0)
        st2_, de2_ = sq.sqp_fnn_iterate(its, step_rule="merit")   # 20 more from there: where the loop ends up
        rq = sq.get_results(want=("status", "u"))
        import mpc_oracle as mo   # checker: first-order certificate of the NLP itself
        fo = mo.FnnModel(W_in, W_h, b_h, W_out, "tanh")
        kkt = max(mo.nlp_kkt_residual(fo, X0q[i], rq["u"][i], xr, ur, 100.0 * np.eye(nq), 0.1 * np.eye(mq), np.zeros((mq, mq)), Pq,
                                      -np.ones(mq), np.ones(mq)) for i in range(0, bq, 32))
        out["sqp_fnn"] = {"value": its / best, "unit": "SQP iterations/s (256 instances, Fnn 4-2-16x2 tanh, N=50)",
                          "ms_per_iteration": 1e3 * best / its, "instance_iterations_per_s": its * bq / best,
                          "iterations": its, "step_inf_last": float(st_[-1]), "defect_inf_last": float(de_[-1]),
                          "step_inf_after_40": float(st2_[-1]), "defect_inf_after_40": float(de2_[-1]),
                          "nlp_kkt_residual_max_sampled": float(kkt),
                          "status_counts": np.bincount(rq["status"], minlength=3).tolist(),
                          "note": "merit-function step rule, start = the network's own rollout; the whole iteration runs on the handle's stream; "
                                  "the KKT residual is sampled after 40 iterations"}
        # per-kernel averages of the iteration from the committed rocprofv3 summary of the same command (profiles/): the iteration is a
        # chain of one-workgroup-per-instance launches, each latency bound at 256 instances -- there is no HBM or MFMA roofline it nears;
        # the inverse's FP64 rate (2 nz^3 flops per matrix) is stated against the vector FP64 peak for scale
        try:
            import csv
            ks, src = {}, None
            for cand in ("r5_sqp_kernel_stats.csv", "r4_sqp_kernel_stats.csv", "r3_sqp_kernel_stats.csv", "r2_sqp_kernel_stats.csv"):
                if os.path.exists(os.path.join(ROOT, "profiles", cand)):
                    src = cand
                    break
            with open(os.path.join(ROOT, "profiles", src)) as f:
                for row in csv.DictReader(f):
                    # (k_polish_sgl<1> / k_guess_iterate_ws: the iterations that start from a guess with the inverse of its working set;
                    # k_polish_sgl<0>: the first iteration after a start, which has an ADMM phase)
                    for short in ("k_design_ltv_reg", "k_design_inverse_c32", "k_design_inverse_t<4, 16", "k_polish_sgl<1>", "k_polish_sgl<0>", "k_sqp_step",
                                  "k_sqp_prepare", "k_guess_iterate_ws", "k_guess_iterate(", "k_fnn_jacobian_w", "k_design_scale", "k_fnn_rollout", "k_riccati_t"):
                        if short in row["Name"]:
                            key = short.rstrip("(") if short.startswith(("k_polish_sgl", "k_guess_iterate")) else short.split("<")[0]
                            ks[key] = round(float(row["AverageNs"]) / 1e3, 1)
            if ks:
                nzq = mq * Nq
                inv_us = ks.get("k_design_inverse_c32") or ks.get("k_design_inverse_t")
                out["sqp_fnn"]["kernel_us_avg"] = ks
                out["sqp_fnn"]["kernel_us_source"] = "profiles/" + src + " (rocprofv3 --kernel-trace --stats of tools/profile_sqp.py 256 50 20 25; not measured in this run)"
                if inv_us:
                    out["sqp_fnn"]["inverse_fp64_tflops"] = {"achieved": 2.0 * nzq ** 3 * bq / (inv_us * 1e-6) / 1e12, "peak_vector_fp64": 78.6,
                                                             "note": "256 matrices of 100 x 100: one workgroup (eight waves) each on 256 CUs, 100 dependent pivots"}
        except (OSError, KeyError, ValueError):
            pass
        # roofline of the figure: flops of one instance-iteration = Jacobians N (2 H (n + m) + 2 L H^2 + 2 n H) (n + m + 1) + the LTV
        # condensed Hessian ~ N n nz^2 (row blocks Gamma_k' Q Gamma_k accumulated stage by stage) + one nz x nz inverse 2 nz^3 (the
        # second one only in the first iteration) + v0 2 nz^2 + the active-set finish (~ 2 nz^2 per change)
        nzq = mq * Nq
        Hq_, Lq_ = int(np.asarray(W_in).shape[0]), len(W_h)
        flq = (Nq * (2 * Hq_ * (nq + mq) + 2 * Lq_ * Hq_ * Hq_ + 2 * nq * Hq_) * (nq + mq + 1) + Nq * nq * nzq * nzq + 2.0 * nzq ** 3
               + 2 * nzq * nzq + 10 * 2 * nzq * nzq)
        secq = best / its
        out["sqp_fnn"]["roofline"] = {
            "bound": "mfma", "kernel": "k_design_ltv_reg (~90 us: the longest of the 11 launches; the 100 x 100 Gauss-Jordan inverse k_design_inverse_c32<16,true> "
                                       "~51 us, the Jacobians ~40 us, k_guess_iterate_ws ~30 us, k_sqp_step ~27 us, k_polish_sgl<1> ~24 us)",
            "achieved": bq * flq / secq / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": bq * flq / secq / 1e12 / FP64_PEAK_TFLOPS,
            "traffic": None, "algorithmic_flops_per_instance_iteration": flq,
            "note": "FP64 vector pipeline (78.6 TFLOP/s peak, as the matrix cores); 256 instances are one workgroup per CU in every kernel of "
                    "the chain: each is bound by its own dependent steps (100 pivots at ~1.2 k cycles, N stage updates, the active-set changes), not by a throughput roofline"}
        sq.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ---- CPU baseline (kind "port": oracle/almpc_oracle.c, OpenMP over instances, all host cores) on a bounded
        # sample of the same workload: the same 4096 x0, same options, repeated until ~10 s of wall time
        import c_oracle
        import mpc_oracle as mo
        po = mo.make_problem(p.A, p.B, N_HORIZON, p.u_min, p.u_max)
        des = mo.design_shared(po, rho=opts.rho, sigma=opts.sigma, rho_profile=args.rho_profile)
        kw = dict(alpha=opts.alpha, eps_abs=opts.eps_abs, eps_rel=opts.eps_rel, max_iter=int(opts.max_iter),
                  check_every=int(opts.check_every), polish=bool(opts.polish), threads=pkg.sharding.host_cpu_share())
        c_oracle.step_batch(po, des, X0[:256], **kw)  # warm
        reps, t0, used = 0, time.perf_counter(), 1
        while True:
            r = c_oracle.step_batch(po, des, X0, **kw)
            used = r["threads"]
            reps += 1
            if time.perf_counter() - t0 > 12.0 or reps >= 100000:
                break
        cpu_el = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": reps / cpu_el, "unit": "batch-steps/s (one step = 4096 instance QP solves)",
                               "cores": int(used), "kind": "port",
                               "sample": f"{reps} passes over the same 4096 instances of this run ({cpu_el:.1f} s wall), oracle/almpc_oracle.c "
                                         f"(gcc -O3 -march=native, OpenMP): same ADMM+polish+rollout as the HIP path",
                               "instance_steps_per_s": reps * BATCH_PER_GPU / cpu_el}
        # secondary: single-thread latency of the exact active-set oracle (numpy), the truth the parity
        # tests use, on 12 instances of the same batch
        t0 = time.perf_counter()
        for i_ in range(0, 12 * 300, 300):
            mo.solve_mpc_exact(po, X0[i_])
        out["cpu_baseline"]["exact_oracle_ms_per_solve_1_thread"] = 1e3 * (time.perf_counter() - t0) / 12
    if force_exit:   # a helper thread is stuck inside librccl: print the measurement and leave without tearing anything down
        if rank == 0:
            print(json.dumps(out), flush=True)
        try:
            ranks.barrier()
        finally:
            os._exit(0)
    solver.close()
    if solver2 is not None:
        solver2.close()  # (freed only now: a large hipFree in the middle slows the launches that follow it)
    if rank == 0:
        print(json.dumps(out))
    ranks.close()


if __name__ == "__main__":
    main()
