"""Traceback of a failing graph capture of the train step.  python tools/capture_debug.py [train|train_full]"""
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from vmg_amd.data import synthetic_clip, synthetic_target  # noqa: E402
from vmg_amd.train import TrainStep  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "train"]
dev = torch.device("cuda", 0)
model = bench.build_model(dev, wl)
step = TrainStep(model, lr=2e-4, betas=(0.9, 0.99), aux=True, aux_ratio=0.005, distributed=False)
lrs = synthetic_clip(wl["batch"], wl["frames"], 64, 64, seed=1234, device=dev)
hrs = synthetic_target(lrs, seed=4321)
try:
    step.capture(lrs, hrs, warmup=2)
    print("capture ok")
except Exception:
    traceback.print_exc()
