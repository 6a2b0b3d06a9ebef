O=$GRAFT_REPO_ROOT/gpurun_out/r03_a; mkdir -p $O; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?" >> $O/bench_line.err
python scripts/kbench.py softmax > $O/kb_softmax.log 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/bench_stats --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-leg > $O/bench_line_profiled.json 2> $O/bench_profiled.err
find $O -name "*kernel_trace.csv" -size +5M -delete
tail -3 $O/pytest_gpu.log; cat $O/bench_line.json | head -c 1500
