#!/bin/bash
# Round 6 evidence for the evaluation kernel: kernel-trace stats of scripts/micro/eval_probe.py, the issue counters
# (separate --pmc passes), and the MFMA-sweep micro-benchmark.
O=$GRAFT_REPO_ROOT/gpurun_out/eval_evidence; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo "build failed"; exit 1; }
timeout 120 scripts/micro/build/eval_sweep_rate > $O/eval_sweep_rate.txt 2>&1
python3 scripts/micro/eval_probe.py 2>&1 | grep probe > $O/probe.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o ev --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/micro/eval_probe.py > $O/trace.log 2>&1
cp $O/trace/ev_kernel_stats.csv $O/eval_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete
cd $GRAFT_REPO_ROOT
bash scripts/micro/eval_pmc.sh > $O/pmc.txt 2>&1
cat $O/probe.txt; head -6 $O/eval_kernel_stats.csv | cut -c1-150; tail -22 $O/pmc.txt | head -18
