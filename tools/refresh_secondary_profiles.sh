# Secondary kernel summaries of a round (state rows, per-instance models / batched design, SQP loop, configs[3] pipeline): run from
# anywhere on the GPU box.  TAG names the round (default r3).
set -x
TAG=${TAG:-r5}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
for job in "state_rows tools/time_state_rows.py" "per_instance_models tools/time_batched.py" "batched_design tools/time_batched_design.py" "sqp tools/profile_sqp.py 256 50 20 25" "config3_relin tools/profile_relin.py 50" "structured tools/time_structured.py 50"; do
    set -- $job; name=$1; shift
    rm -rf $R/gpurun_out/prof_$name
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$name -- python3 $R/$@ > $R/gpurun_out/prof_$name.log 2>&1
    f=$(find $R/gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $R/gpurun_out/${TAG}_${name}_kernel_stats.csv
done
ls -la $R/gpurun_out/${TAG}_*_kernel_stats.csv
