import os, sys, subprocess, socket, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.test_distributed_gpu import _local_grads
mode = sys.argv[1] if len(sys.argv) > 1 else "reducer"
tmp = tempfile.mkdtemp()
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
procs = []
for rank in range(2):
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_child.py"), mode, tmp], env=env))
for p in procs: p.wait()
res = [torch.load(os.path.join(tmp, f"rank{r}.pt")) for r in range(2)]
s0 = res[0]["log"][0]["state"]
g = [_local_grads(s0, 60 + r, mode) for r in range(2)]
gmax = max(float(v.abs().max()) for v in g[0].values())
rows = []
for n in g[0]:
    ref = (g[0][n] + g[1][n]) / 2
    got = res[0]["log"][0]["grads"][n]
    sc = max(float(ref.abs().max()), 1e-3 * gmax)
    rows.append((float((got - ref).abs().max()) / sc, n, float(ref.abs().max()), float((got - g[0][n]).abs().max()) / sc, float((got - g[1][n]).abs().max()) / sc, float((got - (g[0][n] + g[1][n])).abs().max()) / sc))
rows.sort(reverse=True)
for r in rows[:12]: print("err %.2e %s |ref| %.2e  vs g0 %.2e vs g1 %.2e vs sum %.2e" % (r[0], r[1], r[2], r[3], r[4], r[5]))
print("buckets", res[0]["buckets"])
