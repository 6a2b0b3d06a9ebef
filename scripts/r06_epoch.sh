#!/bin/bash
# round 6: measured training epochs (full amazon-book shape) + planted-structure convergence sweeps
mkdir -p gpurun_out
python examples/train_kgat.py --synthetic 1.0 --epochs 3 --log_json gpurun_out/r06_epoch_full.json > gpurun_out/r06_epoch_full.log 2>&1
tail -25 gpurun_out/r06_epoch_full.log
for lr in 0.001 0.003 0.01; do
  echo "== planted lr $lr"
  python examples/train_kgat.py --planted --epochs 4 --lr $lr --eval_before --batch_size 2048 --batch_size_kg 1024 2>&1 | grep -v amdgpu.ids | grep -E "Epoch|test recall|valid recall|GNN|KGE"
done
