set -x
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/r2_trace $R/gpurun_out/r2_pmc_fetch $R/gpurun_out/r2_pmc_write $R/gpurun_out/r2_pmc_mfma $R/gpurun_out/r2_pmc_lds
timeout 900 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/r2_bench.json 2> $R/gpurun_out/r2_bench.err
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_trace -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-classes --no-pipelined > $R/gpurun_out/r2_bench_traced.json 2> /dev/null
B="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-classes --no-pipelined --no-closed-loop --no-sqp --no-structured --no-relin"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r2_pmc_fetch -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r2_pmc_write -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc MfmaUtil MfmaFlopsF64 --output-format csv -d $R/gpurun_out/r2_pmc_mfma -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/r2_pmc_lds -- $B > /dev/null 2>&1
cd $R
python3 tools/summarize_profiles.py r2 gpurun_out/r2_trace gpurun_out/r2_pmc_fetch gpurun_out/r2_pmc_write gpurun_out/r2_pmc_mfma gpurun_out/r2_pmc_lds 2>&1 | tail -3
cp profiles/r2_*.csv profiles/r2_hbm_traffic.json gpurun_out/ 2>/dev/null
tail -c 1500 gpurun_out/r2_bench.json
