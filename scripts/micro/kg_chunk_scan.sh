#!/bin/bash
# Developer probe (round 6): samples per weight-gradient partial (kTrChunk) now that the partials are summed inside
# the Adam launch: rebuilds the library per value on the GPU box and times the KG phase's iteration.
cd $GRAFT_REPO_ROOT
for c in 64 32 16; do
  sed -i "s/^constexpr int kTrChunk = [0-9]*;/constexpr int kTrChunk = $c;/" dgl-kgat_amd/csrc/kgat_transr.hip
  python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  echo "kTrChunk = $c"
  python3 scripts/micro/kg_phase_probe.py 400 2>&1 | grep kg_phase | tail -2
done
