#!/bin/bash
for sl in 512 640 768 1024 512 768 1024; do
GM_WGRAD_SLOTS=$sl python - <<'PY'
import os, torch, sys
sys.path.insert(0, "/root/repo")
import bench
r = bench.extra_train(torch.device("cuda:0"), steps=10, warmup=3)
print("slots", os.environ["GM_WGRAD_SLOTS"], "train", round(r["value"], 2), "steps/s", round(r["ms"], 3), "ms")
PY
done
