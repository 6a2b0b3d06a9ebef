#!/bin/bash
# Developer probe (round 6): where the evaluation sweep's item fragments come from (L2 hit / miss, fabric reads) and
# what the memory pipeline stalls on - separate --pmc passes over scripts/micro/eval_probe.py.
O=$GRAFT_REPO_ROOT/gpurun_out/eval_mem; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo "build failed"; exit 1; }
cd /tmp; export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/scripts/micro/eval_probe.py"
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_HIT_sum" "TCC_MISS_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $O/m$i --output-format csv -- $B > $O/m$i.log 2>&1
  grep probe $O/m$i.log | cut -c1-200
done
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py "eval_topk_kernel<4, 11>" $O/m1 $O/m2 $O/m3 $O/m4 > $O/summary.txt 2>&1
find $O -name "*counter_collection.csv" -size +3M -delete
cat $O/summary.txt
