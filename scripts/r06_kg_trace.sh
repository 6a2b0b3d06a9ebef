#!/bin/bash
# kernel trace of the KG phase's iterations (per-kernel averages -> profiles/r06_kg_phase_kernel_stats.csv)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT


timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kgtrace -o kg --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/micro/kg_phase_probe.py 400 > $OUT/r06_kg_trace_stdout.txt 2>&1
find $OUT/kgtrace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r06_kg_phase_kernel_stats.csv
cut -d, -f1-4 $OUT/r06_kg_phase_kernel_stats.csv | head -12 | cut -c1-160; grep kg_phase $OUT/r06_kg_trace_stdout.txt
