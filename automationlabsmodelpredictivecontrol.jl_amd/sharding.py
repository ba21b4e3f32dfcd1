"""Multi-GPU split of a batch of independent MPC instances.

Instances never interact (the reference has a single-instance API, src/main/computation_mpc.jl:17-55), so the
batch is cut into contiguous shards, one per rank / GPU, with NO data-path collective.  torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) is used only for the barrier around
the timed region and for the max-over-ranks of the elapsed time."""
from __future__ import annotations

import os


def shard_range(batch: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of `batch` instances for `rank` of `world`: sizes differ by at most one,
    earlier ranks take the remainder."""
    if not (0 <= rank < world) or batch < 0:
        raise ValueError("need 0 <= rank < world and batch >= 0")
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class Ranks:
    """Process-group plumbing shared by bench.py and the gloo tests."""

    def __init__(self, backend=None, device_index=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = "cpu"
        self.backend = "none"
        if self.world > 1:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            import torch
            import torch.distributed as dist
            # ALMPC_DIST_BACKEND=gloo lets the multi-process plumbing be exercised on a box with fewer GPUs than ranks
            backend = os.environ.get("ALMPC_DIST_BACKEND") or backend or "nccl"
            self.backend = backend
            if backend == "nccl":
                idx = self.local_rank if device_index is None else device_index
                torch.cuda.set_device(idx)
                self.device = f"cuda:{idx}"
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", idx))
            else:
                dist.init_process_group(backend=backend)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def broadcast_bytes(self, payload, root: int = 0) -> bytes:
        """The bytes rank `root` holds, on every rank (used to hand the library's 128-byte RCCL id to the other ranks)."""
        if self.dist is None:
            return payload
        box = [payload if self.rank == root else None]
        self.dist.broadcast_object_list(box, src=root)
        return box[0]

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None


def visible_gpu_count() -> int:
    """GPUs a process started from here would see, WITHOUT initialising HIP in this process (a launcher must stay free to start
    its ranks): hipGetDeviceCount in a short-lived child process (it honours HIP_/ROCR_VISIBLE_DEVICES and the container's device
    cgroup, which a walk over /sys/class/kfd would not); the kfd topology is the fallback when no child can be started."""
    import subprocess
    import sys
    code = ("import ctypes\n"
            "n = ctypes.c_int(0)\n"
            "try:\n"
            "    lib = ctypes.CDLL('libamdhip64.so')\n"
            "    rc = lib.hipGetDeviceCount(ctypes.byref(n))\n"
            "    print(n.value if rc == 0 else 0)\n"
            "except OSError:\n"
            "    print(0)\n")
    try:
        out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120, text=True)
        return max(0, int(out.stdout.strip().splitlines()[-1]))
    except (OSError, ValueError, IndexError, subprocess.SubprocessError):
        pass
    count = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                count += 1
    except (OSError, ValueError):
        return 0
    return count


def aggregate_rate(units_per_rank: float, steps: int, elapsed_max: float, world: int) -> float:
    """Whole-job throughput: units all ranks processed / max-over-ranks time (weak scaling)."""
    return world * units_per_rank * steps / elapsed_max


def host_cpu_share() -> int:
    """CPUs this process may actually use: cgroup quota (cpu.max) if set, else the affinity mask.  On the GPU boxes
    `nproc` shows the whole host while the container is limited to a share of it."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]) + 0.5)))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, int(q / int(g.read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)
