cd $GRAFT_REPO_ROOT
F=dgl-kgat_amd/csrc/kgat_eval.hip
cp $F /tmp/eval.orig
for wf in 4 2 1; do
  cp /tmp/eval.orig $F
  sed -i "s/  const int64_t want = slots \* 4;/  const int64_t want = slots * $wf;/" $F
  python3 -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1 || { echo "build failed"; continue; }
  for cfg in "8000 24915" "23566 48123" "45919 45538" "70679 24915"; do
    set -- $cfg
    echo "rounds wanted = $wf, users = $1, items = $2: $(EVAL_PROBE_USERS=$1 EVAL_PROBE_ITEMS=$2 python3 scripts/micro/eval_probe.py 2>&1 | grep probe)"
  done
done
cp /tmp/eval.orig $F
python3 -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1
