#!/bin/bash
# training step with / without the host-side edge_index range check (two blocking reductions per forward)
for rep in 1 2 3; do for v in check nocheck; do
V=$v python - <<'PY'
import os, torch, sys
sys.path.insert(0, "/root/repo")
import bench
import gnn_manip_amd.epd_gnn as E
if os.environ["V"] == "nocheck":
    E._check_edge_index = lambda *a: None
r = bench.extra_train(torch.device("cuda:0"), steps=10, warmup=3)
print(os.environ["V"], round(r["value"], 2), "steps/s", round(r["ms"], 3), "ms")
PY
done; done
