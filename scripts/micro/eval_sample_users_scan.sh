#!/bin/bash
# Developer probe (round 6): evaluation with and without the threshold sample over user / item counts.
# HISTORICAL: the threshold sample (kEvalSampleTiles) was removed from kgat_eval.hip after these scans (profiles/r06_eval_scan.txt);
# the script applies to the sources up to commit 'Evaluation: the prune selects the K-th best key by bisection'.
cd $GRAFT_REPO_ROOT
F=dgl-kgat_amd/csrc/kgat_eval.hip
cp $F /tmp/eval.orig
for st in 16 0; do
  cp /tmp/eval.orig $F
  sed -i "s/^constexpr int kEvalSampleTiles = [0-9]*;/constexpr int kEvalSampleTiles = $st;/" $F
  python3 -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1 || { echo "build failed"; tail -3 /tmp/build.log; continue; }
  for nu in 500 2000 8000 30000 70679 300000; do
    echo "sample tiles = $st, users = $nu: $(EVAL_PROBE_USERS=$nu python3 scripts/micro/eval_probe.py 2>&1 | grep probe)"
  done
  echo "sample tiles = $st, users = 70679, items 100000: $(EVAL_PROBE_ITEMS=100000 python3 scripts/micro/eval_probe.py 2>&1 | grep probe)"
  echo "sample tiles = $st: $(python3 -m pytest tests/test_gpu_eval.py -x -q -s -k 'full_size' 2>&1 | grep -E 'eval\]' )"
done
cp /tmp/eval.orig $F
python3 -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1
