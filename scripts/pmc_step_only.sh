# Developer tool: only the whole-step PMC passes of scripts/round4_runs.sh (attention, softmax, dense kernel traffic records).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_pmc_step; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_step_fetch --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg --no-graphs > $O/pmc_step_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_step_write --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg --no-graphs > $O/pmc_step_write.log 2>&1
cd $R
grep "pmc_traffic.py \$O/pmc_step_fetch" scripts/round4_runs.sh > /tmp/pmc_lines.sh
O=$O bash /tmp/pmc_lines.sh
find $O -name "*counter_collection.csv" -size +3M -delete
ls $O
