#!/bin/bash
# training step: time + kernel budget
python - <<'PY'
import json, torch, sys
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
for i in range(2):
    r = bench.extra_train(dev, steps=10, warmup=3)
    print("train", round(r["value"], 2), "steps/s", round(r["ms"], 3), "ms", "loss", r["loss"])
PY
