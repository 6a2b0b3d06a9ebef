# round-5 baseline run: GPU tests, bench line, kernel trace of the drop-in surface (VERDICT r4 task 6)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_base
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?" >> $O/bench_line.err
python scripts/surface_time.py > $O/surface_vs_fused.txt 2>&1
python scripts/kbench.py train --rounds 10 > $O/kbench_train.txt 2>&1
python scripts/kbench.py kg --rounds 30 > $O/kbench_kg.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/surface_trace --output-format csv -- python3 $R/scripts/surface_time.py > $O/surface_profiled.txt 2>&1
find $O -name "*kernel_trace.csv" -size +8M -delete
tail -3 $O/pytest_gpu.log
