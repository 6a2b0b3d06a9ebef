# The round's reference runs (developer tool): GPU tests, smoke, bench lines of every config, the N > 1 path on one
# GPU over gloo, rocprofv3 kernel stats and PMC passes.  Writes under gpurun_out/<tag>/.  The library is built before
# the first profiler line (never from inside a profiled process).
TAG=${1:-r05_final}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo "build failed"; exit 1; }
if [ "$2" != "notests" ]; then
python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(seeds=(7, 8, 9))" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
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
python scripts/surface_time.py > $O/surface_vs_fused.txt 2>&1
python scripts/kbench.py train --rounds 10 > $O/kbench_train.txt 2>&1
python scripts/kbench.py kg --rounds 30 > $O/kbench_kg.txt 2>&1
python scripts/micro/kg_host_probe.py > $O/kg_host_probe.txt 2>&1
python scripts/micro/gather_vs_spmm_widths.py > $O/gather_vs_spmm_widths.txt 2>&1
cd /tmp; export TMPDIR=/tmp
# kernel-trace stats of the driver's bench command (the training leg runs inside it too)
rocprofv3 --kernel-trace --stats -d $O/bench_stats --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_line_profiled.json 2> $O/bench_profiled.err
rocprofv3 --kernel-trace --stats -d $O/bench_stats_step_only --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-leg --no-train-leg --no-graphs > $O/bench_line_profiled_step_only.json 2> $O/bench_profiled_step_only.err
# kernel traces of the training steps (launch counts, where the time goes)
rocprofv3 --kernel-trace --stats -d $O/trace_kg --output-format csv -- python3 $R/scripts/kbench.py kg --rounds 20 > $O/trace_kg.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_train --output-format csv -- python3 $R/scripts/kbench.py train --rounds 10 > $O/trace_train.txt 2>&1
# PMC passes: SpMM on the amazon-book graph, the whole step, issue counters
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_spmm_fetch --output-format csv -- python3 $R/scripts/kbench.py spmm --algos merge --rounds 5 > $O/pmc_spmm_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_spmm_write --output-format csv -- python3 $R/scripts/kbench.py spmm --algos merge --rounds 5 > $O/pmc_spmm_write.log 2>&1
export PROBE_MUL_SELF=0
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_pl_fetch --output-format csv -- python3 $R/scripts/hbm_probe.py redraw 1e7 2e8 3 > $O/pmc_pl_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_pl_write --output-format csv -- python3 $R/scripts/hbm_probe.py redraw 1e7 2e8 3 > $O/pmc_pl_write.log 2>&1
unset PROBE_MUL_SELF
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg --no-train-leg --no-graphs"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_step_fetch --output-format csv -- $B > $O/pmc_step_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_step_write --output-format csv -- $B > $O/pmc_step_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_i1 --output-format csv -- $B > $O/pmc_i1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/pmc_i2 --output-format csv -- $B > $O/pmc_i2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS -d $O/pmc_i3 --output-format csv -- $B > $O/pmc_i3.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM -d $O/pmc_i4 --output-format csv -- $B > $O/pmc_i4.log 2>&1
cd $R
python3 scripts/pmc_summary.py "att_fold_fused|spmm_merge2_kernel<16, 64, false, false, 0>|spmm_merge2_kernel<8, 32, false|softmax_local|bi_interaction_kernel<64, 64, 1" $O/pmc_i1 $O/pmc_i2 $O/pmc_i3 $O/pmc_i4 > $O/pmc_issue_counters.txt 2>&1
python scripts/pmc_traffic.py $O/pmc_spmm_fetch $O/pmc_spmm_write "spmm_merge2_kernel|spmm_finish_kernel" $O/pmc_spmm_traffic.json --sources kgat_spmm.hip,kgat_spmm_impl.h,kgat_common.h --workload "amazon-book-shaped CKG N=159251 E=3663302, D=64, h*h_N epilogue" --command "rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE) --output-format csv -- python3 scripts/kbench.py spmm --algos merge --rounds 5" --algorithmic 1008516988 > /dev/null 2>&1
python scripts/pmc_traffic.py $O/pmc_pl_fetch $O/pmc_pl_write "spmm_merge2_kernel|spmm_finish_kernel" $O/pmc_spmm_traffic_powerlaw.json --sources kgat_spmm.hip,kgat_spmm_impl.h,kgat_common.h --workload "power-law CKG drawn on the device N=10000000 E=200000000, D=64, plain update_all(u_mul_e, sum)" --command "PROBE_MUL_SELF=0 rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE) --output-format csv -- python3 scripts/hbm_probe.py redraw 1e7 2e8 3" --algorithmic 55400000000 > /dev/null 2>&1
python scripts/pmc_traffic.py $O/pmc_step_fetch $O/pmc_step_write "att_fold_fused" $O/pmc_att_traffic.json --sources kgat_att_persistent.hip,kgat_att_common.h,kgat_common.h --workload "amazon-book-shaped CKG N=159251 E=3663302 R=41, d=k=64, fused form, grouped-order logits" --command "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg --no-train-leg --no-graphs" > /dev/null 2>&1
python scripts/pmc_traffic.py $O/pmc_step_fetch $O/pmc_step_write "softmax_local_kernel|softmax_cut_rows_kernel" $O/pmc_softmax_traffic.json --sources kgat_softmax.hip,kgat_common.h --workload "amazon-book-shaped CKG N=159251 E=3663302, grouped-order logits read through the position map" --command "same passes as pmc_att_traffic.json" --algorithmic 59249836 > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -size +5M -delete
find $O -name "*counter_collection.csv" -size +3M -delete
du -sh $O; tail -3 $O/pytest_gpu.log 2>/dev/null; tail -2 $O/smoke.log 2>/dev/null
