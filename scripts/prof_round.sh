set -x
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r02prof
mkdir -p $O
# 1. kernel-trace stats of the driver's bench command
rocprofv3 --kernel-trace --stats -d $O/bench_stats --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_line_profiled.json 2> $O/bench_profiled.err
# 2. PMC passes: SpMM on the amazon-book graph (kbench: merge, h*h_N epilogue)
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_spmm_fetch --output-format csv -- python3 $R/scripts/kbench.py spmm --algos merge --rounds 5 > $O/pmc_spmm_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_spmm_write --output-format csv -- python3 $R/scripts/kbench.py spmm --algos merge --rounds 5 > $O/pmc_spmm_write.log 2>&1
# 3. PMC passes: SpMM on the HBM-resident power-law graph (plain operator)
export PROBE_MUL_SELF=0
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_pl_fetch --output-format csv -- python3 $R/scripts/hbm_probe.py redraw 1e7 2e8 3 > $O/pmc_pl_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_pl_write --output-format csv -- python3 $R/scripts/hbm_probe.py redraw 1e7 2e8 3 > $O/pmc_pl_write.log 2>&1
unset PROBE_MUL_SELF
# 4. PMC passes over the whole step (attention, softmax, ...)
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_step_fetch --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg > $O/pmc_step_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_step_write --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg > $O/pmc_step_write.log 2>&1
# drop the big traces, keep stats + counter CSVs
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
