# Developer tool (round 5): issue / wait counters of the KG step's kernels (autograd path: every kernel alone)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kgpmc; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
cd /tmp; export TMPDIR=/tmp
timeout 180 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/p1 --output-format csv -- python3 $R/scripts/micro/kg_host_probe.py > $O/p1.log 2>&1
timeout 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d $O/p2 --output-format csv -- python3 $R/scripts/micro/kg_host_probe.py > $O/p2.log 2>&1
timeout 180 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_MFMA -d $O/p3 --output-format csv -- python3 $R/scripts/micro/kg_host_probe.py > $O/p3.log 2>&1
cd $R
python3 scripts/pmc_summary.py "transr_wgrad_partial_kernel|transr_reduce_kernel|transr_sample_kernel|small_sort_kernel" $O/p1 $O/p2 $O/p3 2>&1 | head -70
find $O -name "*counter_collection.csv" -size +3M -delete
