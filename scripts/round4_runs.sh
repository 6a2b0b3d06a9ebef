# Round 4's reference runs (developer tool): GPU tests, smoke, bench lines of every config, the N > 1 path on one GPU
# over gloo, training logs, rocprofv3 kernel stats and PMC passes.  Writes under gpurun_out/<tag>/.
TAG=${1:-r04_final}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?" >> $O/bench_line.err
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-hbm-leg > $O/bench_line_50steps.json 2>/dev/null
python bench.py --workload last-fm --no-cpu-baseline --no-hbm-leg > $O/bench_line_lastfm.json 2>/dev/null
python bench.py --dim 128 --no-cpu-baseline --no-hbm-leg > $O/bench_line_amazon_dim128.json 2>/dev/null
python bench.py --workload last-fm --dim 8 --layers 1 --no-cpu-baseline --no-hbm-leg > $O/bench_line_lastfm_dim8_1layer.json 2>/dev/null
python bench.py --workload power-law --steps 5 --warmup 3 --no-cpu-baseline --no-hbm-leg > $O/bench_line_powerlaw_10M_200M.json 2>/dev/null
export KGAT_DIST_BACKEND=gloo KGAT_FORCE_DEVICE=0
timeout 900 python bench.py --gpus 8 --steps 5 --warmup 2 > $O/bench_line_8ranks_one_gpu_gloo.json 2> $O/bench_8ranks.err; echo "rc $?" >> $O/bench_8ranks.err
KGAT_EXCHANGE_CHUNKS=3 timeout 900 python bench.py --gpus 4 --steps 5 --warmup 2 > $O/bench_line_4ranks_one_gpu_gloo_chunked_exchange.json 2> $O/bench_4ranks.err; echo "rc $?" >> $O/bench_4ranks.err
timeout 600 python examples/train_kgat.py --synthetic 0.02 --epochs 1 --max_iters 3 --grad_digest > $O/train_1gpu.log 2>&1
timeout 600 python examples/train_kgat.py --synthetic 0.02 --epochs 1 --max_iters 3 --grad_digest --gpus 2 > $O/train_2gpu_gloo.log 2>&1
unset KGAT_DIST_BACKEND KGAT_FORCE_DEVICE
python scripts/surface_time.py > $O/surface_vs_fused.txt 2>&1
python scripts/kbench.py train --rounds 10 > $O/kbench_train.txt 2>&1
python scripts/kbench.py kg --rounds 30 > $O/kbench_kg.txt 2>&1
python scripts/shard_local_time.py 8 > $O/shard_local_time_8way.txt 2>&1
cd /tmp; export TMPDIR=/tmp
# kernel-trace stats of the driver's bench command
rocprofv3 --kernel-trace --stats -d $O/bench_stats --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_line_profiled.json 2> $O/bench_profiled.err
# PMC passes: the plain aggregation on the amazon-book graph (kbench: merge)
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_spmm_fetch --output-format csv -- python3 $R/scripts/kbench.py spmm --algos merge --rounds 5 > $O/pmc_spmm_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_spmm_write --output-format csv -- python3 $R/scripts/kbench.py spmm --algos merge --rounds 5 > $O/pmc_spmm_write.log 2>&1
# PMC passes: SpMM on the HBM-resident power-law graph (plain operator)
export PROBE_MUL_SELF=0
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_pl_fetch --output-format csv -- python3 $R/scripts/hbm_probe.py redraw 1e7 2e8 3 > $O/pmc_pl_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_pl_write --output-format csv -- python3 $R/scripts/hbm_probe.py redraw 1e7 2e8 3 > $O/pmc_pl_write.log 2>&1
unset PROBE_MUL_SELF
# PMC passes over the whole step (attention, softmax, dense kernel)
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_step_fetch --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg --no-graphs > $O/pmc_step_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_step_write --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg --no-graphs > $O/pmc_step_write.log 2>&1
cd $R
python scripts/pmc_traffic.py $O/pmc_spmm_fetch $O/pmc_spmm_write "spmm_merge2_kernel|spmm_finish_kernel" $O/pmc_spmm_traffic.json --sources kgat_spmm.hip,kgat_spmm_impl.h,kgat_common.h --workload "amazon-book-shaped CKG N=159251 E=3663302, D=64, plain update_all(u_mul_e, sum)" --command "rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE) --output-format csv -- python3 scripts/kbench.py spmm --algos merge --rounds 5" --algorithmic 1008516988 > /dev/null 2>&1
python scripts/pmc_traffic.py $O/pmc_pl_fetch $O/pmc_pl_write "spmm_merge2_kernel|spmm_finish_kernel" $O/pmc_spmm_traffic_powerlaw.json --sources kgat_spmm.hip,kgat_spmm_impl.h,kgat_common.h --workload "power-law CKG drawn on the device N=10000000 E=200000000, D=64, plain update_all(u_mul_e, sum)" --command "PROBE_MUL_SELF=0 rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE) --output-format csv -- python3 scripts/hbm_probe.py redraw 1e7 2e8 3" --algorithmic 55400000000 > /dev/null 2>&1
python scripts/pmc_traffic.py $O/pmc_step_fetch $O/pmc_step_write "att_fold_fused_kernel" $O/pmc_att_traffic.json --sources kgat_att_persistent.hip,kgat_att_common.h,kgat_common.h --workload "amazon-book-shaped CKG N=159251 E=3663302 R=41, d=k=64, fused form (bf16 / fp16 piece products), grouped-order logits" --command "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg --no-graphs" > /dev/null 2>&1
python scripts/pmc_traffic.py $O/pmc_step_fetch $O/pmc_step_write "softmax_local_kernel|softmax_cut_rows_kernel" $O/pmc_softmax_traffic.json --sources kgat_softmax.hip,kgat_common.h --workload "amazon-book-shaped CKG N=159251 E=3663302, grouped-order logits read through the position map" --command "same passes as pmc_att_traffic.json" --algorithmic 59249836 > /dev/null 2>&1
python scripts/pmc_traffic.py $O/pmc_step_fetch $O/pmc_step_write "bi_interaction_kernel<64, 64, 1" $O/pmc_bi_traffic.json --sources kgat_dense.hip,kgat_common.h --workload "amazon-book-shaped CKG N=159251, 64 -> 64 (kgat_bi_interaction_mul_deferred_f32: H, HN read - and the row offsets plus the tile partials of the rows the aggregation left -; h_out, normalised slice and ego block written, the last two non-temporal)" --command "same passes as pmc_att_traffic.json" --algorithmic 203841280 > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -size +5M -delete
find $O -name "*counter_collection.csv" -size +3M -delete
du -sh $O; tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log
