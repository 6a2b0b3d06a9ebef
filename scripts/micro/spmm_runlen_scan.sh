#!/bin/bash
# Developer probe (round 6): narrow rows - run lengths that are not powers of two (the plain operator; the deferred
# finish assumes power-of-two tiles), rebuilt per variant on the GPU box, against the bare gather of the same rows.
cd $GRAFT_REPO_ROOT
H=dgl-kgat_amd/csrc/kgat_spmm_impl.h
cp $H /tmp/impl.h.orig
for v in ${@:-"32 16" "24 12" "20 12" "28 20" "40 24" "48 24"}; do
  set -- $v
  cp /tmp/impl.h.orig $H
  sed -i "s/  return (lpr == 8 || lpr == 4) ? run_len(lpr) \/ kSpmmMidDiv : run_len(lpr);/  return lpr == 8 ? $1 : (lpr == 4 ? $2 : run_len(lpr));/" $H
  grep -c "return lpr == 8 ? $1" $H
  python3 -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1 || { echo "build failed"; tail -5 /tmp/build.log; continue; }
  echo "== run length $1 at LPR 8 (D = 32), $2 at LPR 4 (D = 16)"
  python3 scripts/micro/gather_vs_spmm_widths.py 2>&1 | grep "D= 16\|D= 32"
done
cp /tmp/impl.h.orig $H
python3 -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1
