#!/bin/bash
# Developer probe (round 6): the evaluation launch over user counts (24,915 items, 176 columns, K = 20).
cd $GRAFT_REPO_ROOT
for nu in ${@:-500 2000 8000 30000 70679 300000}; do
  echo "users = $nu: $(EVAL_PROBE_USERS=$nu python3 scripts/micro/eval_probe.py 2>&1 | grep probe)"
done
