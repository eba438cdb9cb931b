"""Which forward tensors have several consumers in the train step's autograd graph (their gradients are summed by the autograd engine with torch add
kernels, one per extra consumer)?  Walks the graph from the loss and counts the edges into every (node, output) pair.
   python tools/fan_in_nodes.py [train|train_full|train_swin]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vmg_amd.data import synthetic_clip, synthetic_target
from vmg_amd.train import charbonnier_edge_loss_hip

dev = torch.device("cuda", 0)
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "train"]
model = bench.build_model(dev, wl)
lrs = synthetic_clip(wl["batch"], wl["frames"], wl["size"], wl["size"], seed=1234, device=dev)
hrs = synthetic_target(lrs, seed=4321)
out = model(lrs)
loss = charbonnier_edge_loss_hip(out.float(), hrs.float(), 1e-12, 0.005)
edges = collections.Counter()
consumers = collections.defaultdict(list)
seen, stack = set(), [loss.grad_fn]
while stack:
    n = stack.pop()
    if n is None or n in seen:
        continue
    seen.add(n)
    for nxt, idx in n.next_functions:
        if nxt is None:
            continue
        edges[(nxt, idx)] += 1
        consumers[(nxt, idx)].append(type(n).__name__)
        stack.append(nxt)
rows = collections.Counter()
for (node, idx), c in edges.items():
    if c > 1 and type(node).__name__ != "AccumulateGrad":
        shape = None
        try:
            shape = tuple(node._input_metadata[idx].shape)
        except Exception:
            pass
        rows[(type(node).__name__, idx, shape, c, tuple(sorted(collections.Counter(consumers[(node, idx)]).items())))] += 1
print("count  producer node [output]  shape  consumers  (consumer nodes)")
for (name, idx, shape, c, cons), k in sorted(rows.items(), key=lambda kv: -(kv[1] * (kv[0][3] - 1))):
    print("%5d  %s[%d]  %s  %d  %s" % (k, name, idx, shape, c, ", ".join("%s x%d" % (a, b) for a, b in cons)))
print("engine adds per backward pass:", sum((key[3] - 1) * k for key, k in rows.items()))
