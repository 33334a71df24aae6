import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_train as T
dev = torch.device("cuda:0")
for seed in (105, 106, 107, 108, 109, 110, 111):
    try:
        T.test_reference_wiring_over_standalone_blocks_trains(dev, 800, 0.07, seed, 3)
        print(seed, "ok")
    except AssertionError as e:
        print(seed, "FAIL", str(e)[:160])
