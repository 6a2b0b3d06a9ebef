O=$GRAFT_REPO_ROOT/gpurun_out/r03_g; mkdir -p $O; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "att_" > $O/pytest_att.log 2>&1; echo "pytest rc $?" >> $O/pytest_att.log
python -m pytest tests/test_gpu_configs.py -m gpu -q -x -s -k "config3" > $O/pytest_c3.log 2>&1; echo "pytest rc $?" >> $O/pytest_c3.log
python bench.py --dim 128 --no-cpu-baseline --no-hbm-leg > $O/bench_dim128.json 2>$O/bench_dim128.err
KGAT_ATT_FORM=folded python bench.py --dim 128 --no-cpu-baseline --no-hbm-leg > $O/bench_dim128_folded.json 2>/dev/null
python scripts/kbench.py att --dim 128 --rounds 10 > $O/kb_att128.log 2>&1
tail -3 $O/pytest_att.log $O/pytest_c3.log; tail -3 $O/bench_dim128.err
