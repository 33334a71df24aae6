#!/bin/bash
# training step A/B of two library builds on one box: variants/lib_old.so vs variants/lib_new.so
for rep in 1 2 3; do for v in old new; do
GM_LIB_PATH=variants/lib_$v.so python - <<PY
import torch, sys
sys.path.insert(0, "/root/repo")
import bench
r = bench.extra_train(torch.device("cuda:0"), steps=10, warmup=3)
print("$v", round(r["value"], 2), "steps/s", round(r["ms"], 3), "ms")
PY
done; done
