#!/bin/bash
# refresh of the round's bench line and CF-step evidence after the last CF-step change (subset of round6_runs.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_refresh; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?"
python scripts/kbench.py train --rounds 10 > $O/kbench_train.txt 2>&1
timeout 900 python examples/train_kgat.py --synthetic 1.0 --epochs 3 --log_json $O/epoch_measured.json > $O/train_epoch_measured.log 2>&1
timeout 600 python examples/train_kgat.py --planted --epochs 12 --lr 0.03 --batch_size 512 --batch_size_kg 512 --eval_before > $O/train_planted.log 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/bench_stats --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_line_profiled.json 2> $O/bench_profiled.err
rocprofv3 --kernel-trace --stats -d $O/trace_cf -o cf --output-format csv -- python3 $R/scripts/kbench.py train --rounds 20 > $O/trace_cf.txt 2>&1
cp $O/trace_cf/cf_kernel_stats.csv $O/cf_step_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete
grep -E "KGE|GNN|epoch" $O/train_epoch_measured.log | tail -3
