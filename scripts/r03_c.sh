O=$GRAFT_REPO_ROOT/gpurun_out/r03_c; mkdir -p $O; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
for i in 1 2; do
KGAT_ATT_SCATTER_CSR=1 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-hbm-leg > $O/bench_scatter_$i.json 2>/dev/null
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-hbm-leg > $O/bench_grouped_$i.json 2>/dev/null
done
scripts/micro/build/gather_modes > $O/gather_modes.log 2>&1
python scripts/placement_study.py modes --tries 12 > $O/placement_modes.log 2>&1
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $O/bench_stats --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-leg > $O/bench_line_profiled.json 2> $O/bench_profiled.err
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE -d $O/pmc1 --output-format csv -- python3 $R/scripts/placement_study.py pmc --sidecar $O/pmc1/sidecar.json > $O/pmc1.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum -d $O/pmc2 --output-format csv -- python3 $R/scripts/placement_study.py pmc --sidecar $O/pmc2/sidecar.json > $O/pmc2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum -d $O/pmc3 --output-format csv -- python3 $R/scripts/placement_study.py pmc --sidecar $O/pmc3/sidecar.json > $O/pmc3.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum -d $O/pmc4 --output-format csv -- python3 $R/scripts/placement_study.py pmc --sidecar $O/pmc4/sidecar.json > $O/pmc4.log 2>&1
cd $R
python scripts/placement_study.py report $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 > $O/placement_report.log 2>&1
find $O -name "*kernel_trace.csv" -size +5M -delete
find $O -name "*counter_collection.csv" -size +20M -delete
tail -3 $O/pytest_gpu.log; tail -3 $O/smoke.log; cat $O/gather_modes.log $O/placement_modes.log $O/placement_report.log
