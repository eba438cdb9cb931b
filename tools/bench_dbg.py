import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
LOG = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench_dbg.log"), "a")
def log(*a):
    s = " ".join(str(x) for x in a); print(s, flush=True); LOG.write(s + "\n"); LOG.flush()
import bench
from vmg_amd.data import synthetic_clip, synthetic_target
from vmg_amd.train import TrainStep, charbonnier_edge_loss
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
t0 = time.time(); model = bench.build_model(dev); log("build", time.time() - t0)
lrs = synthetic_clip(B, 7, 64, 64, device=dev); hrs = synthetic_target(lrs)
ts = TrainStep(model)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    out = model(lrs); torch.cuda.synchronize(); t1 = time.time(); log(f"B={B} it{it} forward", t1 - t0)
    loss = charbonnier_edge_loss(out.float(), hrs.float()); torch.cuda.synchronize(); t2 = time.time(); log("  loss", t2 - t1, float(loss))
    loss.backward(); torch.cuda.synchronize(); t3 = time.time(); log("  backward", t3 - t2)
    ts.opt.step(); ts.opt.zero_grad(set_to_none=True); torch.cuda.synchronize(); t4 = time.time(); log("  opt", t4 - t3)
log("done")
