#!/bin/bash
# HISTORICAL: the threshold sample (kEvalSampleTiles) was removed from kgat_eval.hip after these scans (profiles/r06_eval_scan.txt);
# the script applies to the sources up to commit 'Evaluation: the prune selects the K-th best key by bisection'.
# Developer probe (round 6): evaluation kernel - tiles of the sample segment (kEvalSampleTiles), rebuilt per variant
# on the GPU box; prints the full-size time.
cd $GRAFT_REPO_ROOT
F=dgl-kgat_amd/csrc/kgat_eval.hip
cp $F /tmp/eval.orig
for st in ${@:-16 32 64 128}; do
  cp /tmp/eval.orig $F
  sed -i "s/^constexpr int kEvalSampleTiles = [0-9]*;/constexpr int kEvalSampleTiles = $st;/" $F
  python3 -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1 || { echo "build failed"; tail -3 /tmp/build.log; continue; }
  echo "sample tiles = $st: $(python3 scripts/micro/eval_probe.py 2>&1 | grep probe)"
done
cp /tmp/eval.orig $F
python3 -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1
