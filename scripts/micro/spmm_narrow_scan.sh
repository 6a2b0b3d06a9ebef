#!/bin/bash
# Developer probe (round 6; VERDICT round 5, task 3): narrow rows (D = 32 / 16) - workgroup size and run length of the
# merge kernel's lane groups, rebuilt per variant on the GPU box, timed against the bare gather of the same rows.
cd $GRAFT_REPO_ROOT
H=dgl-kgat_amd/csrc/kgat_spmm_impl.h
cp $H /tmp/impl.h.orig
for thr in 2 8; do for div in 1 2 4; do
  cp /tmp/impl.h.orig $H
  sed -i "s/return (lpr <= 2 \&\& kSpmmThreads == 256) ? 128 : kSpmmThreads;/return (lpr <= $thr \&\& kSpmmThreads == 256) ? 128 : kSpmmThreads;/" $H
  sed -i "s/^constexpr int kSpmmMidDiv = [0-9]*;/constexpr int kSpmmMidDiv = $div;/" $H
  python3 -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1 || { echo "build failed"; tail -5 /tmp/build.log; continue; }
  echo "== 128-thread workgroups for LPR <= $thr, run length 64 / $div at LPR 8, 4"
  python3 scripts/micro/gather_vs_spmm_widths.py 2>&1 | grep "D="
done; done
cp /tmp/impl.h.orig $H
