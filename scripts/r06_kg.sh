#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_train_ops.py -x -q 2>&1 | tail -15
python examples/train_kgat.py --synthetic 1.0 --epochs 3 --log_json gpurun_out/r06_epoch_kgphase.json > gpurun_out/r06_epoch_kgphase.log 2>&1
grep -v amdgpu.ids gpurun_out/r06_epoch_kgphase.log | tail -22
