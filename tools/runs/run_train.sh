#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_dataset.py -x -q -m gpu > gpurun_out/train_tests.log 2>&1
echo "exit $?" >> gpurun_out/train_tests.log
tail -15 gpurun_out/train_tests.log
