# Round-end sequence on the GPU box: bench.py as the driver runs it, a kernel trace, four --pmc passes (each rocprofv3 under its own
# timeout, counters never combined with a trace), then tools/summarize_profiles.py.  TAG names the round (default r3).
set -x
TAG=${TAG:-r5}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf $O/${TAG}_trace $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_mfma $O/${TAG}_pmc_lds
timeout 1200 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-classes --no-pipelined --no-api-path --no-closed-loop --no-sqp --no-structured --no-state-rows --no-small-shared --no-relin > $O/${TAG}_bench_traced.json 2> /dev/null
B="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-classes --no-pipelined --no-closed-loop --no-sqp --no-structured --no-state-rows --no-small-shared --no-relin --no-api-path"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc MfmaUtil MfmaFlopsF64 --output-format csv -d $O/${TAG}_pmc_mfma -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/${TAG}_pmc_lds -- $B > /dev/null 2>&1
cd $R
python3 tools/summarize_profiles.py $TAG gpurun_out/${TAG}_trace gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write gpurun_out/${TAG}_pmc_mfma gpurun_out/${TAG}_pmc_lds 2>&1 | tail -3
cp profiles/${TAG}_*.csv profiles/${TAG}_hbm_traffic.json gpurun_out/ 2>/dev/null
tail -c 1500 gpurun_out/${TAG}_bench.json
