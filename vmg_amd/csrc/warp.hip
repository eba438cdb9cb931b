// Flow-guided sampling kernels of the trajectory recurrence (models/trajectory.py:71-116, 329-333, 414-417),
// channels-last, gfx950.  All are HBM-bound gathers: one thread per (pixel, 16-byte channel vector), consecutive
// lanes walk consecutive channel vectors of a pixel so every corner fetch is a coalesced row segment.
//
// Coordinate arithmetic restates flow_warp + F.grid_sample(align_corners=True) step by step in fp32 WITHOUT fused
// multiply-add contraction (explicit __f*_rn), so that the sampled positions -- and in particular the rounding of the
// nearest mode and the border clamp -- agree with the reference's CPU path bit for bit:
//     g  = x + flow_x;  gx = 2*g / max(W-1,1) - 1;  ix = ((gx + 1) / 2) * (W-1)        (same for y)
#include "common.h"

namespace {

__device__ __forceinline__ float unnorm_coord(float pix, float flow, int size) {
  const float g = __fadd_rn(pix, flow);
  const float denom = (float)(size - 1 > 1 ? size - 1 : 1);
  const float gn = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, g), denom), 1.0f);
  return __fmul_rn(__fdiv_rn(__fadd_rn(gn, 1.0f), 2.0f), (float)(size - 1));
}

__device__ __forceinline__ float clip_border(float v, int size, float* grad_mult) {
  // torch clip_coordinates(_set_grad): clamp to [0, size-1]; gradient 0 at or beyond the borders
  const float hi = (float)(size - 1);
  if (v <= 0.f) { if (grad_mult) *grad_mult = 0.f; return 0.f; }
  if (v >= hi) { if (grad_mult) *grad_mult = 0.f; return hi; }
  if (grad_mult) *grad_mult = 1.f;
  return v;
}

template <typename T>
struct V16 {
  static constexpr int N = 16 / sizeof(T);
  T v[N];
};

// ------------------------------------------------------------------------------------------ bilinear, border
template <typename T>
__global__ __launch_bounds__(256) void warp_bilinear_fwd_kernel(const T* __restrict__ x, const float* __restrict__ flow,
                                                                T* __restrict__ out, int N, int H, int W, int C) {
  constexpr int VN = V16<T>::N;
  const int nvec = C / VN;
  const long long total = (long long)N * H * W * nvec;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int v = (int)(i % nvec);
    const long long pix = i / nvec;
    const int px = (int)(pix % W);
    const int py = (int)((pix / W) % H);
    const long long n = pix / ((long long)W * H);
    const float fx = flow[pix * 2], fy = flow[pix * 2 + 1];
    const float ix = clip_border(unnorm_coord((float)px, fx, W), W, nullptr);
    const float iy = clip_border(unnorm_coord((float)py, fy, H), H, nullptr);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float x1f = x0f + 1.f, y1f = y0f + 1.f;  // weights in ATen's form: nw = (ix_se - ix) * (iy_se - iy), ...
    const float wnw = (x1f - ix) * (y1f - iy), wne = (ix - x0f) * (y1f - iy), wsw = (x1f - ix) * (iy - y0f), wse = (ix - x0f) * (iy - y0f);
    const T* base = x + n * H * W * C + v * VN;
    float acc[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) acc[e] = 0.f;
    auto corner = [&](int yy, int xx, float wgt) {
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const V16<T> t = *reinterpret_cast<const V16<T>*>(base + ((long long)yy * W + xx) * C);
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[e] += to_f32(t.v[e]) * wgt;
      }
    };
    corner(y0, x0, wnw);
    corner(y0, x1, wne);
    corner(y1, x0, wsw);
    corner(y1, x1, wse);
    V16<T> o;
#pragma unroll
    for (int e = 0; e < VN; ++e) o.v[e] = from_f32<T>(acc[e]);
    *reinterpret_cast<V16<T>*>(out + pix * C + v * VN) = o;
  }
}

// backward: dx_acc (zero-initialised, ALWAYS fp32) += scatter of dy; dflow = channel reductions.
// One WAVE per output pixel, lanes along channels: every atomic wave-instruction adds 256 contiguous bytes, the full-rate shape of
// global atomics on this chip (a lane-per-vector mapping strides the lanes by 32 B and ran 10x slower), and the flow gradient is a
// wave reduction written without atomics.  The accumulator is fp32 for bf16 tensors too: with border padding many output pixels clamp
// onto the same source pixels, and a bf16 running sum (global_atomic_pk_add_bf16, round 2) rounds to 8 bits at every add -- order-dependent,
// small addends swamped, 52-75 % of the elements different between two identical launches.  fp32 sums are rounded to bf16 ONCE by the caller.
typedef __attribute__((ext_vector_type(2))) __bf16 wp_bf16x2;
template <typename T>
__global__ __launch_bounds__(256) void warp_bilinear_bwd_kernel(const T* __restrict__ x, const float* __restrict__ flow,
                                                                const T* __restrict__ dy, float* __restrict__ dx_acc,
                                                                float* __restrict__ dflow, int N, int H, int W, int C) {
  const int lane = threadIdx.x & 63;
  const long long npix = (long long)N * H * W;
  for (long long pix = blockIdx.x * 4LL + (threadIdx.x >> 6); pix < npix; pix += (long long)gridDim.x * 4) {
    const int px = (int)(pix % W);
    const int py = (int)((pix / W) % H);
    const long long n = pix / ((long long)W * H);
    const float fx = flow[pix * 2], fy = flow[pix * 2 + 1];
    float gmx, gmy;
    const float ix = clip_border(unnorm_coord((float)px, fx, W), W, &gmx);
    const float iy = clip_border(unnorm_coord((float)py, fy, H), H, &gmy);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float x1f = x0f + 1.f, y1f = y0f + 1.f;
    const float tx = ix - x0f, ty = iy - y0f;
    const float wnw = (x1f - ix) * (y1f - iy), wne = (ix - x0f) * (y1f - iy), wsw = (x1f - ix) * (iy - y0f), wse = (ix - x0f) * (iy - y0f);
    const bool vnw = y0 >= 0 && y0 < H && x0 >= 0 && x0 < W, vne = y0 >= 0 && y0 < H && x1 >= 0 && x1 < W;
    const bool vsw = y1 >= 0 && y1 < H && x0 >= 0 && x0 < W, vse = y1 >= 0 && y1 < H && x1 >= 0 && x1 < W;
    const long long ib = n * H * W * C;
    const long long onw = ib + ((long long)y0 * W + x0) * C, one = ib + ((long long)y0 * W + x1) * C;
    const long long osw = ib + ((long long)y1 * W + x0) * C, ose = ib + ((long long)y1 * W + x1) * C;
    float gix = 0.f, giy = 0.f;
    if constexpr (sizeof(T) == 2) {
      // two passes of one channel per lane over each 128-channel block: lanes 0..63 add 256 contiguous bytes per atomic instruction
      for (int cb = 0; cb < C; cb += 128) {
        const int cp = cb + 2 * lane;          // this lane's channel PAIR for the loads (4-byte loads of dy and the four corners)
        const bool in = cp < C;                // (C is even: the host checks)
        const long long own = pix * C;
        const int cl = in ? cp : 0;
        const wp_bf16x2 gp = *reinterpret_cast<const wp_bf16x2*>(dy + own + cl);
        // the four corner reads are unconditional and issued together (a corner outside the image re-reads this pixel's own x and is dropped):
        // loads behind a branch are waited for one by one
        const wp_bf16x2 xnw = *reinterpret_cast<const wp_bf16x2*>(x + (vnw ? onw : own) + cl), xne = *reinterpret_cast<const wp_bf16x2*>(x + (vne ? one : own) + cl);
        const wp_bf16x2 xsw = *reinterpret_cast<const wp_bf16x2*>(x + (vsw ? osw : own) + cl), xse = *reinterpret_cast<const wp_bf16x2*>(x + (vse ? ose : own) + cl);
        const float g0 = in ? (float)gp[0] : 0.f, g1 = in ? (float)gp[1] : 0.f;
        // the loads hold channel PAIRS per lane; the atomics want one channel per lane, lanes along channels (256 contiguous bytes per
        // instruction): lane l adds channels cb + l and cb + 64 + l, fetched from the lanes that loaded them
        const int sl = lane >> 1;
        const float a0 = __shfl(g0, sl, 64), a1 = __shfl(g1, sl, 64), b0 = __shfl(g0, 32 + sl, 64), b1 = __shfl(g1, 32 + sl, 64);
        const float glo = (lane & 1) ? a1 : a0, ghi = (lane & 1) ? b1 : b0;
        const int clo = cb + lane, chi = cb + 64 + lane;
        auto corner = [&](long long o, float wgt, float sx, float sy, wp_bf16x2 xp) __attribute__((always_inline)) {
          if (clo < C) atomicAdd(dx_acc + o + clo, glo * wgt);
          if (chi < C) atomicAdd(dx_acc + o + chi, ghi * wgt);
          const float xv = (float)xp[0] * g0 + (float)xp[1] * g1;
          gix += xv * sx;
          giy += xv * sy;
        };
        if (vnw) corner(onw, wnw, -(1.f - ty), -(1.f - tx), xnw);
        if (vne) corner(one, wne, (1.f - ty), -tx, xne);
        if (vsw) corner(osw, wsw, -ty, (1.f - tx), xsw);
        if (vse) corner(ose, wse, ty, tx, xse);
      }
    } else {
      for (int c = lane; c < C; c += 64) {
        const float g = to_f32(dy[pix * C + c]);
        if (vnw) { atomicAdd(dx_acc + onw + c, g * wnw); const float xv = to_f32(x[onw + c]) * g; gix -= xv * (1.f - ty); giy -= xv * (1.f - tx); }
        if (vne) { atomicAdd(dx_acc + one + c, g * wne); const float xv = to_f32(x[one + c]) * g; gix += xv * (1.f - ty); giy -= xv * tx; }
        if (vsw) { atomicAdd(dx_acc + osw + c, g * wsw); const float xv = to_f32(x[osw + c]) * g; gix -= xv * ty; giy += xv * (1.f - tx); }
        if (vse) { atomicAdd(dx_acc + ose + c, g * wse); const float xv = to_f32(x[ose + c]) * g; gix += xv * ty; giy += xv * tx; }
      }
    }
    gix = wave_sum(gix);
    giy = wave_sum(giy);
    // d(ix)/d(flow_x) = (2/(W-1)) * ((W-1)/2) = 1 (0 where the border clamp is active)
    if (lane == 0) {
      dflow[pix * 2] = gix * gmx;
      dflow[pix * 2 + 1] = giy * gmy;
    }
  }
}

// ------------------------------------------------------------------------------------------ nearest, border (location maps)
// loc (N, K2, H, W) fp32 planes (pixel coordinates of tracked points, models/trajectory.py:321,332-333)
__global__ __launch_bounds__(256) void warp_nearest_planes_kernel(const float* __restrict__ loc, const float* __restrict__ flow,
                                                                  float* __restrict__ out, int N, int K2, int H, int W) {
  const long long total = (long long)N * H * W;
  for (long long pix = blockIdx.x * 256LL + threadIdx.x; pix < total; pix += (long long)gridDim.x * 256) {
    const int px = (int)(pix % W);
    const int py = (int)((pix / W) % H);
    const long long n = pix / ((long long)W * H);
    const float ix = clip_border(unnorm_coord((float)px, flow[pix * 2], W), W, nullptr);
    const float iy = clip_border(unnorm_coord((float)py, flow[pix * 2 + 1], H), H, nullptr);
    const int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);  // round half to even, as std::nearbyint in ATen
    const bool ok = xn >= 0 && xn < W && yn >= 0 && yn < H;
    for (int k = 0; k < K2; ++k) {
      const long long plane = (n * K2 + k) * H * W;
      out[plane + (long long)py * W + px] = ok ? loc[plane + (long long)yn * W + xn] : 0.f;
    }
  }
}

int grid_for(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

// ------------------------------------------------------------------------------------------ flow smoothing (models/function.py:1466-1478)
// reflect-pad the (n, c, H, W) flow planes on the right / bottom to a multiple of r, r x r mean, nearest x r, crop: out[y][x] = the mean of the padded plane over
// the r x r block that holds (y, x).  One thread per element, the block read directly through the reflection (the planes are a few hundred kB); the backward is the
// adjoint as a GATHER (an input pixel is its own padded position and at most one mirror image per axis): no atomics, no zero-fill.  torch spelled it pad + adaptive
// average pool + expand / reshape / crop copy: 3 launches forward and 4 backward per call on tiny tensors.
__device__ __forceinline__ int fs_reflect(int p, int n) { return p < n ? p : 2 * (n - 1) - p; }

__global__ void flow_smooth_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, long long planes, int H, int W, int r) {
  const long long total = planes * H * W;
  const float inv = 1.0f / (float)(r * r);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const float* pl = in + (i / ((long long)W * H)) * H * W;
    const int y0 = (y / r) * r, x0 = (x / r) * r;
    float s = 0.f;
    for (int dy = 0; dy < r; ++dy) {
      const float* row = pl + (long long)fs_reflect(y0 + dy, H) * W;
      for (int dx = 0; dx < r; ++dx) s += row[fs_reflect(x0 + dx, W)];
    }
    out[i] = s * inv;
  }
}

__global__ void flow_smooth_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, long long planes, int H, int W, int r) {
  const long long total = planes * H * W;
  const int hf = (H + r - 1) / r * r, wf = (W + r - 1) / r * r;
  const float inv = 1.0f / (float)(r * r);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const float* pl = dout + (i / ((long long)W * H)) * H * W;
    // padded positions that read input (y, x): itself and its mirror image 2 (n - 1) - p when that lies in the padding [n, nf)
    int py[2] = {y, 2 * (H - 1) - y}, px[2] = {x, 2 * (W - 1) - x};
    const int ny = (py[1] >= H && py[1] < hf) ? 2 : 1, nx = (px[1] >= W && px[1] < wf) ? 2 : 1;
    float acc = 0.f;
    for (int a = 0; a < ny; ++a) {
      const int y0 = (py[a] / r) * r;
      for (int b = 0; b < nx; ++b) {
        const int x0 = (px[b] / r) * r;
        float g = 0.f;  // sum of the output gradients of the block (its cropped part)
        for (int dy = 0; dy < r && y0 + dy < H; ++dy)
          for (int dx = 0; dx < r && x0 + dx < W; ++dx) g += pl[(long long)(y0 + dy) * W + x0 + dx];
        acc += g;
      }
    }
    din[i] = acc * inv;
  }
}

}  // namespace

extern "C" int vmg_warp_bilinear_fwd(int dtype, const void* x, const float* flow, void* out, int N, int H, int W, int C, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "warp_fwd: bad dtype");
  VMG_CHECK(x && flow && out && N > 0 && H > 0 && W > 0 && C > 0, "warp_fwd: bad arguments");
  const int vn = dtype == VMG_BF16 ? 8 : 4;
  VMG_CHECK(C % vn == 0, "warp_fwd: C must be a multiple of %d", vn);
  VMG_CHECK(((uintptr_t)x | (uintptr_t)out) % 16 == 0, "warp_fwd: pointers must be 16-byte aligned");
  const long long total = (long long)N * H * W * (C / vn);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VMG_BF16) hipLaunchKernelGGL(warp_bilinear_fwd_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16*)x, flow, (bf16*)out, N, H, W, C);
  else hipLaunchKernelGGL(warp_bilinear_fwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)x, flow, (float*)out, N, H, W, C);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_warp_bilinear_bwd(int dtype, const void* x, const float* flow, const void* dy, void* dx_acc, float* dflow, int N,
                                     int H, int W, int C, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "warp_bwd: bad dtype");
  VMG_CHECK(x && flow && dy && dx_acc && dflow && N > 0 && H > 0 && W > 0 && C > 0, "warp_bwd: bad arguments");
  const int vn = dtype == VMG_BF16 ? 8 : 4;
  (void)vn;
  const long long total = (long long)N * H * W * 64;  // one wave per pixel
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VMG_BF16) {
    VMG_CHECK(C % 2 == 0 && ((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx_acc) % 4 == 0, "warp_bwd: bf16 needs an even channel count and 4-byte aligned tensors");
    hipLaunchKernelGGL(warp_bilinear_bwd_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16*)x, flow, (const bf16*)dy, (float*)dx_acc, dflow, N, H, W, C);
  } else {
    hipLaunchKernelGGL(warp_bilinear_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)x, flow, (const float*)dy, (float*)dx_acc, dflow, N, H, W, C);
  }
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_warp_nearest_planes(const float* loc, const float* flow, float* out, int N, int K2, int H, int W, void* stream) {
  VMG_CHECK(loc && flow && out && N > 0 && K2 > 0 && H > 0 && W > 0, "warp_nearest: bad arguments");
  hipLaunchKernelGGL(warp_nearest_planes_kernel, dim3(grid_for((long long)N * H * W)), dim3(256), 0, (hipStream_t)stream, loc, flow, out, N, K2, H, W);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_flow_smooth(const float* in, float* out, int64_t planes, int H, int W, int r, int backward, void* stream) {
  VMG_CHECK(in && out && planes > 0 && H > 0 && W > 0 && r >= 1, "flow_smooth: bad arguments");
  const int hf = (H + r - 1) / r * r, wf = (W + r - 1) / r * r;
  VMG_CHECK(hf - H < H && wf - W < W, "flow_smooth: the reflected padding must be smaller than the plane (as F.pad(mode='reflect') demands)");
  const long long total = planes * H * W;
  if (backward) hipLaunchKernelGGL(flow_smooth_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, out, (long long)planes, H, W, r);
  else hipLaunchKernelGGL(flow_smooth_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, out, (long long)planes, H, W, r);
  VMG_LAUNCH_CHECK();
  return 0;
}
