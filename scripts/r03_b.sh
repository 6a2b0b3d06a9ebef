O=$GRAFT_REPO_ROOT/gpurun_out/r03_b; mkdir -p $O; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?" >> $O/bench_line.err
export KGAT_DIST_BACKEND=gloo KGAT_FORCE_DEVICE=0
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_2ranks.json 2> $O/bench_2ranks.err; echo "rc $?" >> $O/bench_2ranks.err
timeout 900 python bench.py --gpus 8 --steps 5 --warmup 2 > $O/bench_8ranks.json 2> $O/bench_8ranks.err; echo "rc $?" >> $O/bench_8ranks.err
timeout 900 python bench.py --gpus 8 --workload power-law --scale 0.1 --steps 3 --warmup 1 > $O/bench_8ranks_powerlaw.json 2> $O/bench_8ranks_powerlaw.err; echo "rc $?" >> $O/bench_8ranks_powerlaw.err
timeout 600 python examples/train_kgat.py --synthetic 0.02 --epochs 1 --max_iters 3 --grad_digest > $O/train_1gpu.log 2>&1
timeout 600 python examples/train_kgat.py --synthetic 0.02 --epochs 1 --max_iters 3 --grad_digest --gpus 2 > $O/train_2gpu.log 2>&1
unset KGAT_DIST_BACKEND KGAT_FORCE_DEVICE
python scripts/error_attribution.py > $O/error_attribution.log 2>&1
python scripts/error_attribution.py --seed 3 --nodes 400,600,500 --kg 15000 --uv 7000 > $O/error_attribution_big.log 2>&1
rocprofv3 -L > $O/counters.txt 2>&1
tail -3 $O/pytest_gpu.log; tail -2 $O/bench_2ranks.err $O/bench_8ranks.err $O/bench_8ranks_powerlaw.err; grep digest $O/train_*gpu.log; cat $O/error_attribution.log
