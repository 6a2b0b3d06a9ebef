#!/bin/bash
# Developer probe (round 6): evaluation kernel - item tiles in flight per wavefront (kEvalNT) x tiles of the threshold
# sample launch, rebuilt per variant on the GPU box; prints the full-size launch time (tests/test_gpu_eval.py).
cd $GRAFT_REPO_ROOT
F=dgl-kgat_amd/csrc/kgat_eval.hip
cp $F /tmp/eval.orig
for nt in 2 3; do for st in 16 64; do
  cp /tmp/eval.orig $F
  sed -i "s/^constexpr int kEvalNT = [0-9]*;/constexpr int kEvalNT = $nt;/; s/^constexpr int kEvalSampleTiles = [0-9]*;/constexpr int kEvalSampleTiles = $st;/" $F
  python3 -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1 || { echo "build failed"; tail -3 /tmp/build.log; continue; }
  echo "kEvalNT = $nt, sample tiles = $st: $(python3 -m pytest tests/test_gpu_eval.py -x -q -s -k 'full_size or ties' 2>&1 | grep -E 'eval\]|passed|failed' | tr '\n' ' ')"
done; done
cp /tmp/eval.orig $F
