"""Turn the raw rocprofv3 output directories of one round into the small summaries kept under profiles/.

    python tools/summarize_profiles.py <tag> <kernel_trace_dir> <pmc_fetch_dir> <pmc_write_dir> [<more pmc dirs> ...]

writes  profiles/<tag>_bench_kernel_stats.csv   (copy of the --stats kernel summary)
        profiles/<tag>_pmc_summary.csv          (per kernel / counter: dispatches, mean, min, max)
        profiles/<tag>_hbm_traffic.json         (HBM bytes per launch of the two step kernels, corrections as in
                                                 /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE x2 for 16-B-per-lane
                                                 coalesced reads on gfx950, WRITE_SIZE exact; both counters are in KB)
"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, pat):
    hits = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    if not hits:
        raise SystemExit(f"no {pat} under {d}")
    return hits[-1]


def main():
    tag, dtrace, dfetch, dwrite = sys.argv[1:5]
    out = os.path.join(ROOT, "profiles")
    shutil.copy(find(dtrace, "*kernel_stats.csv"), os.path.join(out, f"{tag}_bench_kernel_stats.csv"))
    rows = defaultdict(list)
    for d in (dfetch, dwrite):
        with open(find(d, "*counter_collection.csv")) as f:
            for r in csv.DictReader(f):
                rows[(r["Kernel_Name"], r["Counter_Name"], r["Workgroup_Size"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    with open(os.path.join(out, f"{tag}_pmc_summary.csv"), "w") as f:
        f.write("Kernel_Name,Counter,Workgroup_Size,Grid_Size,Dispatches,Mean,Min,Max\n")
        for (k, c, wg, g), v in sorted(rows.items()):
            f.write(f"\"{k}\",{c},{wg},{g},{len(v)},{sum(v) / len(v)},{min(v)},{max(v)}\n")
    traffic = {}
    for short, pat in (("k_admm", "k_admm<"), ("k_polish", "k_polish<true>"), ("k_admm_inst", "k_admm_inst"), ("k_step_fused", "k_step_fused")):
        ent = {}
        # a kernel launched at several sizes (k_admm_inst: the 4096-instance secondary figure and the 256-instance SQP loop):
        # the largest grid is the one the roofline entry is quoted on
        gmax = max([int(g) for (k, c, wg, g) in rows if pat in k], default=0)
        for (k, c, wg, g), v in rows.items():
            if pat in k and int(g) == gmax:
                ent[c] = (sum(v) / len(v), len(v))
        if "FETCH_SIZE" in ent and "WRITE_SIZE" in ent:
            fkb, fn = ent["FETCH_SIZE"]; wkb, wn = ent["WRITE_SIZE"]
            fetch_mult = 2.0 if short in ("k_polish", "k_admm_inst", "k_step_fused") else 1.0
            traffic[short] = {
                "FETCH_SIZE_KB_per_launch": fkb, "launches_FETCH_SIZE": fn,
                "WRITE_SIZE_KB_per_launch": wkb, "launches_WRITE_SIZE": wn,
                "hbm_bytes_per_launch": (fkb * fetch_mult + wkb) * 1024.0,
                "correction": ("FETCH_SIZE x2 (16-B-per-lane coalesced reads, MI355X_MICROARCH.md HBM section), WRITE_SIZE exact"
                               if fetch_mult == 2.0 else
                               "FETCH_SIZE raw (8-B-per-lane reads: uncalibrated width), WRITE_SIZE exact (16-B-per-lane stores)"),
            }
    # optional further --pmc passes (e.g. MfmaUtil MfmaFlopsF64 / SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE): per-kernel means
    extra = defaultdict(list)
    for d in sys.argv[5:]:
        with open(find(d, "*counter_collection.csv")) as f:
            for r in csv.DictReader(f):
                if "almpc::" in r["Kernel_Name"]:
                    extra[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    if extra:
        with open(os.path.join(out, f"{tag}_pmc_mfma_lds.csv"), "w") as f:
            f.write("Kernel_Name,Counter,Dispatches,Mean,Min,Max\n")
            for (k, c), v in sorted(extra.items()):
                f.write(f"\"{k}\",{c},{len(v)},{sum(v) / len(v)},{min(v)},{max(v)}\n")
    traffic["command"] = ("rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py "
                          "--steps 20 --warmup 5 --no-cpu-baseline --no-classes --no-pipelined --no-closed-loop")
    traffic["workload"] = "bench.py defaults: 4096 quadrotor instances, mixed amplitudes, stiffness rho profile (45), ADMM max_iter 6, polish on"
    with open(os.path.join(out, f"{tag}_hbm_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
