#!/bin/bash
# Developer probe (round 6): issue / wait / MFMA-busy counters of the evaluation kernel (separate --pmc passes)
O=$GRAFT_REPO_ROOT/gpurun_out/eval_pmc; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo "build failed"; exit 1; }
cd /tmp; export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/scripts/micro/eval_probe.py"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/p1 --output-format csv -- $B > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/p2 --output-format csv -- $B > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS -d $O/p3 --output-format csv -- $B > $O/p3.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM -d $O/p4 --output-format csv -- $B > $O/p4.log 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py "eval_topk_kernel<4, 11>" $O/p1 $O/p2 $O/p3 $O/p4 > $O/summary.txt 2>&1
find $O -name "*counter_collection.csv" -size +3M -delete
cat $O/summary.txt; grep probe $O/p1.log
