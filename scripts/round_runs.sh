# The round's reference runs (developer tool): GPU tests, smoke, bench lines of every config, the N > 1
# path on one GPU over gloo, rocprofv3 kernel stats and PMC passes.  Writes under gpurun_out/<tag>/.
TAG=${1:-r03_final}
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
timeout 900 python bench.py --gpus 8 --workload power-law --scale 0.1 --steps 3 --warmup 2 > $O/bench_line_8ranks_powerlaw_scale0.1_gloo.json 2> $O/bench_8ranks_powerlaw.err; echo "rc $?" >> $O/bench_8ranks_powerlaw.err
timeout 600 python examples/train_kgat.py --synthetic 0.02 --epochs 1 --max_iters 3 --grad_digest > $O/train_1gpu.log 2>&1
timeout 600 python examples/train_kgat.py --synthetic 0.02 --epochs 1 --max_iters 3 --grad_digest --gpus 2 > $O/train_2gpu_gloo.log 2>&1
unset KGAT_DIST_BACKEND KGAT_FORCE_DEVICE
python scripts/surface_time.py > $O/surface_vs_fused.txt 2>&1
python scripts/kbench.py train --rounds 10 > $O/kbench_train.txt 2>&1
python scripts/kbench.py kg --rounds 30 > $O/kbench_kg.txt 2>&1
python scripts/shard_local_time.py 8 > $O/shard_local_time_8way.txt 2>&1
python scripts/att_products_check.py > $O/att_product_forms.txt 2>&1
python scripts/att_products_check.py --dim 128 >> $O/att_product_forms.txt 2>&1
python scripts/att_products_check.py --workload lastfm >> $O/att_product_forms.txt 2>&1
python scripts/error_attribution.py > $O/error_attribution.txt 2>&1
python scripts/error_attribution.py --seed 3 --nodes 400,600,500 --kg 15000 --uv 7000 >> $O/error_attribution.txt 2>&1
scripts/micro/build/gather_modes > $O/gather_modes.txt 2>&1
# experiments kept as records: XCD-contiguous tile mapping, wave roles in the attention kernels, softmax variants
AB_FLAG=-DKGAT_SPMM_XCD_REMAP=1 python scripts/micro/spmm_runlen_ab.py > $O/xcd_remap_ab.txt 2>&1
python scripts/micro/att_variants_ab.py --dim 64 --costs 64,38,1051 --rounds 10 -- "-DKGAT_ATT_XCD_REMAP=1" "-DKGAT_ATT_F16_SECOND=1" "-DKGAT_ATT_WAVE_ROLES" "-DKGAT_ATT_WAVE_ROLES -DKGAT_WS_ABLATE=1" "-DKGAT_ATT_WAVE_ROLES -DKGAT_WS_ABLATE=2" "-DKGAT_ATT_WAVE_ROLES -DKGAT_WS_PRODUCERS=4" > $O/att_wave_roles.txt 2>&1
python scripts/micro/att_variants_ab.py --dim 128 --costs 64,12,700 --rounds 8 -- "-DKGAT_ATT_XCD_REMAP=1" "-DKGAT_ATT_WAVE_ROLES" "-DKGAT_F128_PASSES=1" >> $O/att_wave_roles.txt 2>&1
python scripts/micro/att_ws_phases.py 64 >> $O/att_wave_roles.txt 2>&1
python scripts/micro/att_ws_phases.py 128 >> $O/att_wave_roles.txt 2>&1
python scripts/micro/softmax_ab.py "-DKGAT_SM_SMALL_EDGES=0" > $O/softmax_epl_ab.txt 2>&1
python scripts/placement_study.py modes --tries 12 > $O/placement_modes.txt 2>&1
cd /tmp; export TMPDIR=/tmp
# kernel-trace stats of the driver's bench command
rocprofv3 --kernel-trace --stats -d $O/bench_stats --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_line_profiled.json 2> $O/bench_profiled.err
# PMC passes: SpMM on the amazon-book graph (kbench: merge, h*h_N epilogue)
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_spmm_fetch --output-format csv -- python3 $R/scripts/kbench.py spmm --algos merge --rounds 5 > $O/pmc_spmm_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_spmm_write --output-format csv -- python3 $R/scripts/kbench.py spmm --algos merge --rounds 5 > $O/pmc_spmm_write.log 2>&1
# PMC passes: SpMM on the HBM-resident power-law graph (plain operator)
export PROBE_MUL_SELF=0
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_pl_fetch --output-format csv -- python3 $R/scripts/hbm_probe.py redraw 1e7 2e8 3 > $O/pmc_pl_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_pl_write --output-format csv -- python3 $R/scripts/hbm_probe.py redraw 1e7 2e8 3 > $O/pmc_pl_write.log 2>&1
unset PROBE_MUL_SELF
# PMC passes over the whole step (attention, softmax, ...)
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_step_fetch --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg > $O/pmc_step_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_step_write --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg > $O/pmc_step_write.log 2>&1
# the placement modes of the HBM-resident SpMM under counters
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE -d $O/pmc_place1 --output-format csv -- python3 $R/scripts/placement_study.py pmc --sidecar $O/pmc_place1/sidecar.json > $O/pmc_place1.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum -d $O/pmc_place2 --output-format csv -- python3 $R/scripts/placement_study.py pmc --sidecar $O/pmc_place2/sidecar.json > $O/pmc_place2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum -d $O/pmc_place3 --output-format csv -- python3 $R/scripts/placement_study.py pmc --sidecar $O/pmc_place3/sidecar.json > $O/pmc_place3.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum -d $O/pmc_place4 --output-format csv -- python3 $R/scripts/placement_study.py pmc --sidecar $O/pmc_place4/sidecar.json > $O/pmc_place4.log 2>&1
cd $R
python scripts/placement_study.py report $O/pmc_place1 $O/pmc_place2 $O/pmc_place3 $O/pmc_place4 > $O/placement_counters.txt 2>&1
# per-launch traffic records (bench.py reports them while the kernel sources' hash matches)
python scripts/pmc_traffic.py $O/pmc_spmm_fetch $O/pmc_spmm_write "spmm_merge2_kernel|spmm_finish_kernel" $O/pmc_spmm_traffic.json --sources kgat_spmm.hip,kgat_common.h --workload "amazon-book-shaped CKG N=159251 E=3663302, D=64, h*h_N epilogue" --command "rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE) --output-format csv -- python3 scripts/kbench.py spmm --algos merge --rounds 5" --algorithmic 1008516988 > /dev/null 2>&1
python scripts/pmc_traffic.py $O/pmc_pl_fetch $O/pmc_pl_write "spmm_merge2_kernel|spmm_finish_kernel" $O/pmc_spmm_traffic_powerlaw.json --sources kgat_spmm.hip,kgat_common.h --workload "power-law CKG drawn on the device N=10000000 E=200000000, D=64, plain update_all(u_mul_e, sum)" --command "PROBE_MUL_SELF=0 rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE) --output-format csv -- python3 scripts/hbm_probe.py redraw 1e7 2e8 3" --algorithmic 55400000000 > /dev/null 2>&1
python scripts/pmc_traffic.py $O/pmc_step_fetch $O/pmc_step_write "att_fold_fused_kernel" $O/pmc_att_traffic.json --sources kgat_att_persistent.hip,kgat_att_common.h,kgat_common.h --workload "amazon-book-shaped CKG N=159251 E=3663302 R=41, d=k=64, fused form (bf16-piece products), grouped-order logits" --command "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg" > /dev/null 2>&1
python scripts/pmc_traffic.py $O/pmc_step_fetch $O/pmc_step_write "softmax_local_kernel|softmax_cut_rows_kernel" $O/pmc_softmax_traffic.json --sources kgat_softmax.hip,kgat_common.h --workload "amazon-book-shaped CKG N=159251 E=3663302, grouped-order logits read through the position map" --command "same passes as pmc_att_traffic.json" --algorithmic 59249836 > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -size +5M -delete
find $O -name "*counter_collection.csv" -size +3M -delete
du -sh $O; tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log
