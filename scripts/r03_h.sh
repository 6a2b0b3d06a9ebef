O=$GRAFT_REPO_ROOT/gpurun_out/r03_h; mkdir -p $O; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
python scripts/micro/att_variants_ab.py --dim 128 --costs 64,20,600:64,12,700:64,8,900:64,12,1200:64,6,400 -- "-DKGAT_F128_PASSES=2" > $O/att128_ab.log 2>&1
python scripts/kbench.py softmax > $O/kb_softmax.log 2>&1
for i in 1 2; do python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-hbm-leg > $O/bench_50_$i.json 2>/dev/null; done
python bench.py --dim 128 --no-cpu-baseline --no-hbm-leg > $O/bench_dim128.json 2>/dev/null
tail -3 $O/pytest_gpu.log; cat $O/att128_ab.log $O/kb_softmax.log
