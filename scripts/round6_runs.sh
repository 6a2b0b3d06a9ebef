# The round's reference runs (developer tool): GPU tests, smoke, bench lines of every config, the N > 1 path on one
# GPU over gloo, the measured training epoch, rocprofv3 kernel stats (step only, training steps, KG phase, HBM leg) and
# PMC passes.  Writes under gpurun_out/<tag>/.  The library is built before the first profiler line.
TAG=${1:-r06_final}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo "build failed"; exit 1; }
if [ "$2" != "notests" ]; then
python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(seeds=(7, 8, 9))" > $O/smoke_three_seeds.txt 2>&1; echo "smoke rc $?" >> $O/smoke_three_seeds.txt
fi
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?" >> $O/bench_line.err
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-hbm-leg --no-train-leg > $O/bench_line_50steps.json 2>/dev/null
python bench.py --workload last-fm --no-cpu-baseline --no-hbm-leg > $O/bench_line_lastfm.json 2>/dev/null
python bench.py --dim 128 --no-cpu-baseline --no-hbm-leg > $O/bench_line_amazon_dim128.json 2>/dev/null
python bench.py --workload last-fm --dim 8 --layers 1 --no-cpu-baseline --no-hbm-leg > $O/bench_line_lastfm_dim8_1layer.json 2>/dev/null
python bench.py --workload power-law --steps 5 --warmup 3 --no-cpu-baseline --no-hbm-leg > $O/bench_line_powerlaw_10M_200M.json 2>/dev/null
export KGAT_DIST_BACKEND=gloo KGAT_FORCE_DEVICE=0
timeout 900 python bench.py --gpus 8 --steps 5 --warmup 2 > $O/bench_line_8ranks_one_gpu_gloo.json 2> $O/bench_8ranks.err; echo "rc $?" >> $O/bench_8ranks.err
timeout 600 python examples/train_kgat.py --synthetic 0.02 --epochs 1 --max_iters 3 --grad_digest > $O/train_1gpu.log 2>&1
timeout 600 python examples/train_kgat.py --synthetic 0.02 --epochs 1 --max_iters 3 --grad_digest --gpus 2 > $O/train_2gpu_gloo.log 2>&1
unset KGAT_DIST_BACKEND KGAT_FORCE_DEVICE
# the measured epoch: full amazon-book shape, three epochs, no iteration cap; and the planted-structure run
timeout 900 python examples/train_kgat.py --synthetic 1.0 --epochs 3 --log_json $O/epoch_measured.json > $O/train_epoch_measured.log 2>&1
timeout 600 python examples/train_kgat.py --planted --epochs 12 --lr 0.03 --batch_size 512 --batch_size_kg 512 --eval_before > $O/train_planted.log 2>&1
python scripts/kbench.py train --rounds 10 > $O/kbench_train.txt 2>&1
python scripts/micro/kg_phase_probe.py 400 > $O/kg_phase_probe.txt 2>&1
python scripts/micro/gather_vs_spmm_widths.py > $O/gather_vs_spmm_widths.txt 2>&1
cd /tmp; export TMPDIR=/tmp
# kernel-trace stats: the driver's bench command; the step only (+ the first-steps analysis); the training steps; the
# KG phase; the HBM-resident aggregation alone
rocprofv3 --kernel-trace --stats -d $O/bench_stats --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_line_profiled.json 2> $O/bench_profiled.err
rocprofv3 --kernel-trace --stats -d $O/step_stats -o step --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-leg --no-train-leg > $O/bench_line_profiled_step_only.json 2> $O/bench_profiled_step_only.err
cp $O/step_stats/step_kernel_stats.csv $O/bench_kernel_stats_step_only.csv
python3 $R/scripts/first_steps_trace.py $O/step_stats/step_kernel_trace.csv > $O/first_steps_trace.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_cf -o cf --output-format csv -- python3 $R/scripts/kbench.py train --rounds 20 > $O/trace_cf.txt 2>&1
cp $O/trace_cf/cf_kernel_stats.csv $O/cf_step_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $O/trace_kg -o kg --output-format csv -- python3 $R/scripts/micro/kg_phase_probe.py 400 > $O/trace_kg.txt 2>&1
cp $O/trace_kg/kg_kernel_stats.csv $O/kg_phase_kernel_stats.csv
export PROBE_MUL_SELF=0
rocprofv3 --kernel-trace --stats -d $O/trace_hbm -o hbm --output-format csv -- python3 $R/scripts/hbm_probe.py redraw 1e7 2e8 20 > $O/trace_hbm.txt 2>&1
cp $O/trace_hbm/hbm_kernel_stats.csv $O/hbm_leg_kernel_stats.csv
# PMC passes: SpMM on the amazon-book graph, the HBM-resident graph, the whole step
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_pl_fetch --output-format csv -- python3 $R/scripts/hbm_probe.py redraw 1e7 2e8 3 > $O/pmc_pl_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_pl_write --output-format csv -- python3 $R/scripts/hbm_probe.py redraw 1e7 2e8 3 > $O/pmc_pl_write.log 2>&1
unset PROBE_MUL_SELF
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_spmm_fetch --output-format csv -- python3 $R/scripts/kbench.py spmm --algos merge --rounds 5 > $O/pmc_spmm_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_spmm_write --output-format csv -- python3 $R/scripts/kbench.py spmm --algos merge --rounds 5 > $O/pmc_spmm_write.log 2>&1
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg --no-train-leg --no-graphs"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_step_fetch --output-format csv -- $B > $O/pmc_step_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_step_write --output-format csv -- $B > $O/pmc_step_write.log 2>&1
cd $R
python scripts/pmc_traffic.py $O/pmc_spmm_fetch $O/pmc_spmm_write "spmm_merge2_kernel|spmm_finish_kernel" $O/pmc_spmm_traffic.json --sources kgat_spmm.hip,kgat_spmm_impl.h,kgat_common.h --workload "amazon-book-shaped CKG N=159251 E=3663302, D=64, h*h_N epilogue" --command "rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE) --output-format csv -- python3 scripts/kbench.py spmm --algos merge --rounds 5" --algorithmic 1008516988 > /dev/null 2>&1
python scripts/pmc_traffic.py $O/pmc_pl_fetch $O/pmc_pl_write "spmm_merge2_kernel|spmm_finish_kernel" $O/pmc_spmm_traffic_powerlaw.json --sources kgat_spmm.hip,kgat_spmm_impl.h,kgat_common.h --workload "power-law CKG drawn on the device N=10000000 E=200000000, D=64, plain update_all(u_mul_e, sum)" --command "PROBE_MUL_SELF=0 rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE) --output-format csv -- python3 scripts/hbm_probe.py redraw 1e7 2e8 3" --algorithmic 55400000000 > /dev/null 2>&1
python scripts/pmc_traffic.py $O/pmc_step_fetch $O/pmc_step_write "att_fold_fused" $O/pmc_att_traffic.json --sources kgat_att_persistent.hip,kgat_att_common.h,kgat_common.h --workload "amazon-book-shaped CKG N=159251 E=3663302 R=41, d=k=64, fused form, grouped-order logits" --command "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg --no-train-leg --no-graphs" > /dev/null 2>&1
python scripts/pmc_traffic.py $O/pmc_step_fetch $O/pmc_step_write "softmax_local_kernel|softmax_cut_rows_kernel" $O/pmc_softmax_traffic.json --sources kgat_softmax.hip,kgat_common.h --workload "amazon-book-shaped CKG N=159251 E=3663302, grouped-order logits read through the position map" --command "same passes as pmc_att_traffic.json" --algorithmic 59249836 > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +3M -delete
du -sh $O; tail -3 $O/pytest_gpu.log 2>/dev/null; tail -2 $O/smoke_three_seeds.txt 2>/dev/null; grep -E "KGE|epoch" $O/train_epoch_measured.log | tail -4
