O=$GRAFT_REPO_ROOT/gpurun_out/r03_f; mkdir -p $O; cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?" >> $O/bench_line.err
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-hbm-leg > $O/bench_50.json 2>/dev/null
python bench.py --dim 128 --no-cpu-baseline --no-hbm-leg > $O/bench_dim128.json 2>/dev/null
python bench.py --workload last-fm --no-cpu-baseline --no-hbm-leg > $O/bench_lastfm.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/bench_stats --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_line_profiled.json 2> $O/bench_profiled.err
rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_TAG_STALL TCC_BUSY -d $O/pmc5 --output-format csv -- python3 $R/scripts/placement_study.py pmc --sidecar $O/pmc5/sidecar.json > $O/pmc5.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_step_fetch --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg > $O/pmc_step_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_step_write --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg > $O/pmc_step_write.log 2>&1
cd $R
python scripts/placement_study.py report $O/pmc5 > $O/placement_report5.log 2>&1
find $O -name "*kernel_trace.csv" -size +5M -delete
find $O -name "*counter_collection.csv" -size +30M -delete
cat $O/placement_report5.log; tail -2 $O/bench_line.err
