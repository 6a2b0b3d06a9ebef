# Developer tool (round 5): cache-path counters of the step's kernels (TA / TCP / TCC), separate --pmc passes.
TAG=${1:-mempath}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo "build failed"; exit 1; }
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o "TCP_[A-Z_0-9a-z]*\|TCC_[A-Z_0-9a-z]*\|TA_[A-Z_0-9a-z]*\|SQ_LDS[A-Z_0-9a-z]*" $O/avail.txt | sort -u > $O/avail_names.txt
wc -l $O/avail_names.txt
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-leg --no-train-leg"
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_HIT_sum" "TCC_MISS_sum TCC_READ_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS" "TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $set -d $O/m$i --output-format csv -- $B > $O/m$i.log 2>&1
  tail -2 $O/m$i.log | cut -c1-200
done
cd $R
python3 scripts/pmc_summary.py "att_fold_fused_kernel|spmm_merge2_kernel<16, 64, false, false, 0>|gather_probe" $O/m1 $O/m2 $O/m3 $O/m4 $O/m5 > $O/summary.txt 2>&1
find $O -name "*counter_collection.csv" -size +3M -delete
cat $O/summary.txt
