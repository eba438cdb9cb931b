import sys, traceback
sys.path.insert(0, "/root/repo")
import torch, bench
from vmg_amd.data import synthetic_clip, synthetic_target
from vmg_amd.train import TrainStep
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
step = TrainStep(model, lr=2e-4, betas=(0.9, 0.99), aux=True, aux_ratio=0.005, distributed=False)
lrs = synthetic_clip(4, 7, 64, 64, seed=1234, device=dev)
hrs = synthetic_target(lrs, seed=4321)
try:
    step.capture(lrs, hrs, warmup=2)
    print("capture ok")
except Exception:
    traceback.print_exc()
