#!/bin/bash
mkdir -p gpurun_out
python scripts/micro/fused_pass_probe.py > gpurun_out/r06_fused_pass_probe.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06_fused_pass_probe.txt
for cfg in "0.01 30 0.1" "0.03 30 0.1" "0.03 30 0.0"; do
  set -- $cfg
  echo "== planted lr $1 epochs $2 dropout $3"
  python examples/train_kgat.py --planted --epochs $2 --lr $1 --dropout_rate $3 --eval_before --batch_size 2048 --batch_size_kg 1024 2>&1 | grep -E "test recall" | awk '{printf "%s ", $4} END {print ""}'
done
