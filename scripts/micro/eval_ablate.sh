#!/bin/bash
# Developer probe (round 6): the evaluation launch taken apart (results are wrong, the times are the point), rebuilt on
# the GPU box per variant: A no score ever passes the filter (sweep + per-tile maximum + closing prunes of empty
# lists); B = A without the closing prunes and list stores; C = B with the filter reduced to one compare per tile.
cd $GRAFT_REPO_ROOT
F=dgl-kgat_amd/csrc/kgat_eval.hip
cp $F /tmp/eval.orig
build() { python3 -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1 || { echo "build failed"; tail -3 /tmp/build.log; }; }
echo "as shipped:        $(python3 scripts/micro/eval_probe.py 2>&1 | grep probe)"
sed -i 's/if (__ballot(m >= tau_s) != 0ull) {/if (__ballot(m == 1.2345e38f) != 0ull) {/' $F
build; echo "A no candidates:   $(python3 scripts/micro/eval_probe.py 2>&1 | grep probe)"
sed -i 's/    if (u0 + v >= n_users) break;/    if (u0 + v >= n_users || n_items > 0) break;/' $F
build; echo "B no closing:      $(python3 scripts/micro/eval_probe.py 2>&1 | grep probe)"
sed -i 's/      for (int r = 1; r < 16; ++r) m = fmaxf(m, acc\[t\]\[r\]);/      for (int r = 15; r < 16; ++r) m = fmaxf(m, acc[t][r]);/' $F
build; echo "C one-compare:     $(python3 scripts/micro/eval_probe.py 2>&1 | grep probe)"
cp /tmp/eval.orig $F
build
