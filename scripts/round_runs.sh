# The round's reference runs (developer tool): GPU tests, bench lines of every config, the
# rocprofv3 passes of scripts/prof_round.sh.  Writes under gpurun_out/<tag>/.
TAG=${1:-r02_final}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?" >> $O/bench_line.err
python bench.py --workload last-fm --no-cpu-baseline --no-hbm-leg > $O/bench_line_lastfm.json 2>/dev/null
python bench.py --dim 128 --no-cpu-baseline --no-hbm-leg > $O/bench_line_amazon_dim128.json 2>/dev/null
python bench.py --workload last-fm --dim 8 --layers 1 --no-cpu-baseline --no-hbm-leg > $O/bench_line_lastfm_dim8_1layer.json 2>/dev/null
python bench.py --workload power-law --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg > $O/bench_line_powerlaw_10M_200M.json 2>/dev/null
python scripts/surface_time.py > $O/surface_time.txt 2>&1
python scripts/kbench.py train --rounds 10 > $O/kbench_train.txt 2>&1
python scripts/kbench.py kg --rounds 30 > $O/kbench_kg.txt 2>&1
python scripts/shard_local_time.py 8 > $O/shard_local_time.txt 2>&1
bash scripts/prof_round.sh > $O/prof_round.log 2>&1
tail -3 $O/pytest_gpu.log
