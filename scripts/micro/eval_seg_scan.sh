#!/bin/bash
# Developer probe (round 6): evaluation kernel - segment rows of the grid (count, equal or short-tailed) forced one by
# one instead of the plan's simulated choice, rebuilt per variant on the GPU box; prints the full-size time.
cd $GRAFT_REPO_ROOT
F=dgl-kgat_amd/csrc/kgat_eval.hip
cp $F /tmp/eval.orig
echo "the plan's own choice: $(python3 scripts/micro/eval_probe.py 2>&1 | grep probe)"
for v in ${@:-"4 0" "4 1" "5 0" "5 1" "6 1" "8 1"}; do
  set -- $v
  cp /tmp/eval.orig $F
  sed -i "s/for (int n = (int)seg; n <= (int)seg + 2 \&\& n <= max_seg \&\& n <= 64; ++n)/for (int n = $1; n <= $1; ++n)/; s/for (int tail = 0; tail <= 1; ++tail) {/for (int tail = $2; tail <= $2; ++tail) {/" $F
  python3 -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1 || { echo "build failed"; tail -3 /tmp/build.log; continue; }
  echo "rows = $1, short tail = $2: $(python3 scripts/micro/eval_probe.py 2>&1 | grep probe)"
done
cp /tmp/eval.orig $F
python3 -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1
