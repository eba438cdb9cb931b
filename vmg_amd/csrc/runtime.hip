// Error reporting and library identification for libvmg_hip.so.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void vmg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vmg_last_error(void) { return g_err; }
extern "C" int vmg_version(void) { return 100; }
extern "C" int vmg_max_lds_bytes(void) { return 160 * 1024; }

// ------------------------------------------------------------------------------------------------------
// Live kernel timing for bench.py's roofline object: HIP events recorded on the launch stream around every
// `stride`-th launch of the kernel class selected with vmg_prof_begin.  Reading the result synchronises.
// ------------------------------------------------------------------------------------------------------
#include <vector>
namespace {
struct Prof {
  int klass = 0, stride = 1;
  long long seen = 0, pixels = 0;  // pixels: only launches over exactly this many pixels are timed (0 = any)
  std::vector<hipEvent_t> ev;  // start/stop pairs
  size_t used = 0;
} g_prof;
}  // namespace

bool vmg_prof_before(int klass, long long pixels, hipStream_t st) {
  if (g_prof.klass == 0 || klass != g_prof.klass || (g_prof.pixels != 0 && pixels != g_prof.pixels)) return false;
  if ((g_prof.seen++ % g_prof.stride) != 0 || g_prof.used + 2 > g_prof.ev.size()) return false;
  (void)hipEventRecord(g_prof.ev[g_prof.used], st);
  return true;
}
void vmg_prof_after(hipStream_t st) {
  (void)hipEventRecord(g_prof.ev[g_prof.used + 1], st);
  g_prof.used += 2;
}

extern "C" int vmg_prof_begin(int klass, int stride, int max_samples) {
  VMG_CHECK(klass > 0 && stride > 0 && max_samples > 0, "prof_begin: bad arguments");
  g_prof.klass = klass; g_prof.stride = stride; g_prof.seen = 0; g_prof.used = 0;
  while (g_prof.ev.size() < (size_t)max_samples * 2) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) { vmg_set_error("prof_begin: hipEventCreate failed"); return -2; }
    g_prof.ev.push_back(e);
  }
  return 0;
}

extern "C" int vmg_prof_select_pixels(int64_t pixels) {
  VMG_CHECK(pixels >= 0, "prof_select_pixels: negative");
  g_prof.pixels = pixels;
  return 0;
}

extern "C" int vmg_prof_end(int64_t* launches_seen, int* samples, double* total_ms) {
  VMG_CHECK(launches_seen && samples && total_ms, "prof_end: null pointer");
  double tot = 0.0;
  int n = 0;
  for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
    if (hipEventSynchronize(g_prof.ev[i + 1]) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]) == hipSuccess) { tot += ms; ++n; }
  }
  *launches_seen = g_prof.seen; *samples = n; *total_ms = tot;
  g_prof.klass = 0; g_prof.used = 0;
  return 0;
}

// Event-pair interval of an empty one-wave kernel on `stream`: what the HIP events of vmg_prof_begin add on
// top of a kernel's own duration (dispatch + event latency).  bench.py subtracts it so that its live number agrees
// with rocprofv3's kernel timestamps.
__global__ void vmg_null_kernel(int* p) {
  if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = 0;
}

extern "C" double vmg_prof_null_interval_us(int reps, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
  double tot = 0.0;
  int n = 0;
  for (int i = 0; i < reps + 3; ++i) {
    hipLaunchKernelGGL(vmg_null_kernel, dim3(512, 1), dim3(256), 0, st, (int*)nullptr);  // a predecessor, as in the real chain
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(vmg_null_kernel, dim3(1, 1), dim3(64), 0, st, (int*)nullptr);
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    if (i >= 3 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) { tot += ms; ++n; }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return n ? tot / n * 1000.0 : -1.0;
}
