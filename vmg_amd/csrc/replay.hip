// Replaying a captured train step WITHOUT the hipGraph executor.
//
// The benchmarked step is host-bound: ~2 000 launches, ~55 ms of Python per step for ~48 ms of GPU work.  Capturing the step into a
// hipGraph removes Python, but hipGraphLaunch on this ROCm is no faster than the eager step (490 vs 486 LR-frames/s): its executor
// spends per node what Python spent.  A stream capture of one stream is a linear list of kernel / memcpy / memset nodes whose
// arguments the graph owns; this file reads them out once (hipGraphGetNodes + the node parameter getters, topologically ordered)
// and replays them with plain hipLaunchKernel / hipMemcpy3DAsync / hipMemset*Async calls on the caller's stream: ~4 us of host time
// per node, in order, no graph executor.  Ranges of the list can be replayed separately, so that a data-parallel step can issue
// its gradient all-reduces between segments.  The graph must stay alive (torch.cuda.CUDAGraph(keep_graph=True)).
#include <vector>
#include <map>
#include <queue>

#include "common.h"

namespace {

struct ReplayOp {
  int type;  // 0 kernel, 1 memcpy, 2 memset
  hipKernelNodeParams k;
  hipMemcpy3DParms c;
  hipMemsetParams m;
};

struct Replay {
  std::vector<ReplayOp> ops;
  int skipped;
};

}  // namespace

extern "C" void* vmg_replay_build(void* graph_, int* n_ops, int* n_kernels) {
  hipGraph_t graph = (hipGraph_t)graph_;
  size_t n = 0;
  if (!graph || hipGraphGetNodes(graph, nullptr, &n) != hipSuccess || n == 0) {
    vmg_set_error("replay_build: hipGraphGetNodes failed or the graph is empty");
    return nullptr;
  }
  std::vector<hipGraphNode_t> nodes(n);
  if (hipGraphGetNodes(graph, nodes.data(), &n) != hipSuccess) {
    vmg_set_error("replay_build: hipGraphGetNodes failed");
    return nullptr;
  }
  // topological order, ties by creation order (a one-stream capture is a chain; forks from side streams are serialised)
  std::map<hipGraphNode_t, int> index;
  for (size_t i = 0; i < n; ++i) index[nodes[i]] = (int)i;
  std::vector<std::vector<int>> succ(n);
  std::vector<int> indeg(n, 0);
  for (size_t i = 0; i < n; ++i) {
    size_t nd = 0;
    if (hipGraphNodeGetDependencies(nodes[i], nullptr, &nd) != hipSuccess) { vmg_set_error("replay_build: hipGraphNodeGetDependencies failed"); return nullptr; }
    if (nd == 0) continue;
    std::vector<hipGraphNode_t> deps(nd);
    if (hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd) != hipSuccess) { vmg_set_error("replay_build: hipGraphNodeGetDependencies failed"); return nullptr; }
    for (size_t d = 0; d < nd; ++d) {
      auto it = index.find(deps[d]);
      if (it == index.end()) { vmg_set_error("replay_build: dependency outside the graph"); return nullptr; }
      succ[it->second].push_back((int)i);
      ++indeg[i];
    }
  }
  std::priority_queue<int, std::vector<int>, std::greater<int>> ready;
  for (size_t i = 0; i < n; ++i)
    if (indeg[i] == 0) ready.push((int)i);
  Replay* r = new Replay();
  r->skipped = 0;
  int kernels = 0;
  size_t done = 0;
  while (!ready.empty()) {
    const int i = ready.top();
    ready.pop();
    ++done;
    for (int s : succ[i])
      if (--indeg[s] == 0) ready.push(s);
    hipGraphNodeType ty;
    if (hipGraphNodeGetType(nodes[i], &ty) != hipSuccess) { vmg_set_error("replay_build: hipGraphNodeGetType failed"); delete r; return nullptr; }
    ReplayOp op;
    memset(&op, 0, sizeof(op));
    if (ty == hipGraphNodeTypeKernel) {
      op.type = 0;
      if (hipGraphKernelNodeGetParams(nodes[i], &op.k) != hipSuccess) { vmg_set_error("replay_build: hipGraphKernelNodeGetParams failed"); delete r; return nullptr; }
      if (!op.k.func || (!op.k.kernelParams && !op.k.extra)) { vmg_set_error("replay_build: kernel node %d has no function / arguments", i); delete r; return nullptr; }
      if (!op.k.kernelParams) { vmg_set_error("replay_build: kernel node %d was launched with an argument buffer (extra), not supported", i); delete r; return nullptr; }
      ++kernels;
    } else if (ty == hipGraphNodeTypeMemcpy) {
      op.type = 1;
      if (hipGraphMemcpyNodeGetParams(nodes[i], &op.c) != hipSuccess) { vmg_set_error("replay_build: hipGraphMemcpyNodeGetParams failed"); delete r; return nullptr; }
    } else if (ty == hipGraphNodeTypeMemset) {
      op.type = 2;
      if (hipGraphMemsetNodeGetParams(nodes[i], &op.m) != hipSuccess) { vmg_set_error("replay_build: hipGraphMemsetNodeGetParams failed"); delete r; return nullptr; }
      if (op.m.height > 1) { vmg_set_error("replay_build: 2-D memset node %d is not supported", i); delete r; return nullptr; }
      if (op.m.elementSize != 1 && op.m.elementSize != 2 && op.m.elementSize != 4) { vmg_set_error("replay_build: memset element size %u", op.m.elementSize); delete r; return nullptr; }
    } else if (ty == hipGraphNodeTypeEmpty || ty == hipGraphNodeTypeEventRecord || ty == hipGraphNodeTypeWaitEvent) {
      ++r->skipped;  // ordering only: the replay is in order on one stream anyway
      continue;
    } else {
      vmg_set_error("replay_build: node %d has type %d (kernel, memcpy, memset only)", i, (int)ty);
      delete r;
      return nullptr;
    }
    r->ops.push_back(op);
  }
  if (done != n) { vmg_set_error("replay_build: the graph has a cycle?"); delete r; return nullptr; }
  if (n_ops) *n_ops = (int)r->ops.size();
  if (n_kernels) *n_kernels = kernels;
  return r;
}

// kernel node `idx` of the list: function pointer and grid size (for callers that look for a marker kernel); returns 0 if not a kernel
extern "C" int vmg_replay_kernel_info(void* h, int idx, void** func, unsigned* grid_x, unsigned* block_x, void** first_arg) {
  Replay* r = (Replay*)h;
  if (!r || idx < 0 || idx >= (int)r->ops.size() || r->ops[idx].type != 0) return 0;
  const hipKernelNodeParams& k = r->ops[idx].k;
  if (func) *func = k.func;
  if (grid_x) *grid_x = k.gridDim.x;
  if (block_x) *block_x = k.blockDim.x;
  if (first_arg) *first_arg = k.kernelParams ? k.kernelParams[0] : nullptr;
  return 1;
}

extern "C" int vmg_replay_run(void* h, int first, int last, void* stream) {
  Replay* r = (Replay*)h;
  VMG_CHECK(r && first >= 0 && last <= (int)r->ops.size() && first <= last, "replay_run: bad range");
  hipStream_t st = (hipStream_t)stream;
  for (int i = first; i < last; ++i) {
    ReplayOp& op = r->ops[i];
    hipError_t e;
    if (op.type == 0) {
      e = hipLaunchKernel(op.k.func, op.k.gridDim, op.k.blockDim, op.k.kernelParams, op.k.sharedMemBytes, st);
    } else if (op.type == 1) {
      const hipMemcpy3DParms& c = op.c;
      if (!c.srcArray && !c.dstArray && c.extent.height <= 1 && c.extent.depth <= 1)  // 1-D: what a stream capture of hipMemcpyAsync records
        e = hipMemcpyAsync(c.dstPtr.ptr, c.srcPtr.ptr, c.extent.width, c.kind, st);
      else
        e = hipMemcpy3DAsync(&op.c, st);
      if (e != hipSuccess) {
        vmg_set_error("replay_run: memcpy op %d failed: %s (dst %p src %p extent %zu x %zu x %zu, kind %d, src pos %zu dst pos %zu)", i, hipGetErrorString(e),
                      c.dstPtr.ptr, c.srcPtr.ptr, c.extent.width, c.extent.height, c.extent.depth, (int)c.kind, c.srcPos.x, c.dstPos.x);
        return -2;
      }
    } else {
      if (op.m.elementSize == 1) e = hipMemsetD8Async((hipDeviceptr_t)op.m.dst, (unsigned char)op.m.value, op.m.width, st);
      else if (op.m.elementSize == 2) e = hipMemsetD16Async((hipDeviceptr_t)op.m.dst, (unsigned short)op.m.value, op.m.width, st);
      else e = hipMemsetD32Async((hipDeviceptr_t)op.m.dst, (int)op.m.value, op.m.width, st);
    }
    if (e != hipSuccess) {
      vmg_set_error("replay_run: op %d (type %d) failed: %s", i, op.type, hipGetErrorString(e));
      return -2;
    }
  }
  return 0;
}

extern "C" void vmg_replay_destroy(void* h) { delete (Replay*)h; }
