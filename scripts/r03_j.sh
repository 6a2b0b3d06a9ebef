O=$GRAFT_REPO_ROOT/gpurun_out/r03_k; mkdir -p $O; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
python scripts/kbench.py softmax > $O/kb_softmax.log 2>&1
for i in 1 2; do python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-hbm-leg > $O/bench_50_$i.json 2>/dev/null; done
python bench.py --workload power-law --steps 5 --warmup 3 --no-cpu-baseline --no-hbm-leg > $O/bench_powerlaw.json 2>$O/bench_powerlaw.err
tail -5 $O/pytest_gpu.log; cat $O/kb_softmax.log
