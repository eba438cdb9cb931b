import sys, os, ctypes, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from vmg_amd import hip
from vmg_amd.data import synthetic_clip, synthetic_target
from vmg_amd.train import TrainStep
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
m = bench.build_model(dev)
step = TrainStep(m, lr=2e-4, betas=(0.9, 0.99), aux=True, aux_ratio=0.005)
lrs = synthetic_clip(4, 7, 64, 64, seed=1, device=dev); hrs = synthetic_target(lrs, seed=2)
lib = hip.lib()
for _ in range(2): step(lrs, hrs)
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(5): step(lrs, hrs)
torch.cuda.synchronize(); print("eager ms/step", (time.perf_counter()-t0)/5*1e3)
hip.check(lib.vmg_prof_select_pixels(hip.ctx(), bench.K1_PIXELS), "sel")
hip.check(lib.vmg_prof_begin(hip.ctx(), 1, 16, 4096), "begin")
try:
    step.capture(lrs, hrs, warmup=1)
    print("captured")
except Exception as e:
    print("capture failed", type(e).__name__, str(e)[:300]); sys.exit(0)
for _ in range(3): step(lrs, hrs)
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(10): step(lrs, hrs)
torch.cuda.synchronize(); print("graph ms/step", (time.perf_counter()-t0)/10*1e3)
seen, n, ms = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_double(0.0)
rc = lib.vmg_prof_end(hip.ctx(), ctypes.byref(seen), ctypes.byref(n), ctypes.byref(ms))
print("prof_end rc", rc, "seen", seen.value, "samples", n.value, "avg us", ms.value / max(1, n.value) * 1e3)
