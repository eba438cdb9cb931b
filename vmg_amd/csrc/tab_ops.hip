// HBM-bound kernels of the TAB token mixer (models/function.py:542-558 channel attention, :791-802 branch
// re-weighting and tanh gate): one grouped row reduction and one coefficient-broadcast elementwise kernel.
// Tensors are (G, R, C) channels-last views: G groups (frames for the channel attention, clips for the re-weighting),
// R rows (pixels) per group, C channels.  Per-(group, channel) coefficients and all reductions are fp32.
#include "common.h"

namespace {

template <typename T>
struct V16 {
  static constexpr int N = 16 / sizeof(T);
  T v[N];
};

// ------------------------------------------------------------------------------------------------ reduction
// mode 0: out[g,c] += scale * sum_r (a [+ b + c3])[g,r,c]      mode 1: out[g,c] += scale * sum_r a*b
// Threads run along 16-byte channel vectors (VN = 8 bf16 / 4 fp32 channels; VN = 2 when C is not a multiple of that),
// 256/tpr rows per block iteration; the row-parallel partials are combined in LDS in a fixed order and written to the block's row of a
// PARTIALS workspace [G][chunks][C]; group_reduce_final_kernel then adds the chunks of a group in index order.  No atomics: the pooled
// sums -- and with them the whole forward pass -- are the same bits on every run (round 2 added the block partials with float atomics,
// in arrival order).  The final kernel takes the launch slot of the zero-fill the atomic version needed.
template <typename T, int VN>
__global__ __launch_bounds__(256) void group_reduce_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c3,
                                                           float* __restrict__ out, int G, long long R, int C, int mode, float scale,
                                                           int chunks) {
  struct alignas(sizeof(T) * VN) Vec { T v[VN]; };
  __shared__ float red[VN * 256];
  const int g = blockIdx.x / chunks, ck = blockIdx.x - g * chunks;
  const int tpr = C / VN;           // threads per row
  const int rpb = 256 / tpr;        // rows per block iteration
  const int roff = threadIdx.x / tpr, cp = threadIdx.x - roff * tpr;
  const long long r0 = R * ck / chunks, r1 = R * (ck + 1) / chunks;
  float s[VN];
#pragma unroll
  for (int e = 0; e < VN; ++e) s[e] = 0.f;
  if (roff < rpb) {
    // FOUR rows per thread and iteration, every load unconditional (a row past the end re-reads the first one and is not added): one load in
    // flight per thread left the kernel at 2.5 TB/s
    const long long base = (long long)g * R * C + VN * cp;
    constexpr int U = 4;
    for (long long r = r0 + roff; r < r1; r += (long long)U * rpb) {
      Vec va[U], vb[U], vc[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long rr = r + (long long)u * rpb;
        ok[u] = rr < r1;
        const long long o = base + (ok[u] ? rr : r) * C;
        va[u] = *reinterpret_cast<const Vec*>(a + o);
        if (b) vb[u] = *reinterpret_cast<const Vec*>(b + o);
        if (c3) vc[u] = *reinterpret_cast<const Vec*>(c3 + o);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!ok[u]) continue;
        if (mode == 0) {
          if (b && c3) {
#pragma unroll
            for (int e = 0; e < VN; ++e) s[e] += to_f32(va[u].v[e]) + to_f32(vb[u].v[e]) + to_f32(vc[u].v[e]);
          } else if (b) {
#pragma unroll
            for (int e = 0; e < VN; ++e) s[e] += to_f32(va[u].v[e]) + to_f32(vb[u].v[e]);
          } else {
#pragma unroll
            for (int e = 0; e < VN; ++e) s[e] += to_f32(va[u].v[e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < VN; ++e) s[e] += to_f32(va[u].v[e]) * to_f32(vb[u].v[e]);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VN; ++e) red[e * 256 + threadIdx.x] = s[e];
  __syncthreads();
  // C threads, one per channel: channel c = VN * cp2 + e sums the rpb row partials of column thread cp2
  for (int c = threadIdx.x; c < C; c += 256) {
    const int cp2 = c / VN, e = c - cp2 * VN;
    float t = 0.f;
    for (int k = 0; k < rpb; ++k) t += red[e * 256 + k * tpr + cp2];
    out[((long long)g * chunks + ck) * C + c] = t;
  }
}

// out[g, i] = scale * sum over the group's chunks of partial[g][chunk][i]; `width` = C (or 3C for group_reduce3).  One block per (group, 16
// columns): 16 row groups x 16 columns of threads, row group q adds the chunks q, q + 16, ... in that order (64-byte reads, all issued
// up front), the 16 row-group sums are then added in index order through LDS -- a FIXED summation tree, independent of arrival order.
__global__ __launch_bounds__(256) void group_reduce_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int G, int chunks, int width,
                                                                 float scale) {
  __shared__ float sm[16][17];
  const int slabs = (width + 15) >> 4;
  const int g = blockIdx.x / slabs, c = (blockIdx.x - g * slabs) * 16 + (threadIdx.x & 15);
  const int q = threadIdx.x >> 4;
  float t = 0.f;
  if (c < width) {
    const float* p = partial + (long long)g * chunks * width + c;
#pragma unroll 4
    for (int k = q; k < chunks; k += 16) t += p[(long long)k * width];
  }
  sm[q][threadIdx.x & 15] = t;
  __syncthreads();
  if (q == 0 && c < width) {
    float u = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) u += sm[i][threadIdx.x & 15];
    out[(long long)g * width + c] = u * scale;
  }
}

// out (G, C, 3) += scale * sum_r a * {b0, b1, b2}: the three branch sums of the MorphFC re-weighting backward (d softmax-weights) from ONE
// pass over dy (reference: models/function.py:791-793 through autograd).  Same blocking as group_reduce_kernel.
template <typename T, int VN>
__global__ __launch_bounds__(256) void group_reduce3_kernel(const T* __restrict__ a, const T* __restrict__ b0, const T* __restrict__ b1,
                                                            const T* __restrict__ b2, float* __restrict__ out, int G, long long R, int C,
                                                            float scale, int chunks) {
  struct alignas(sizeof(T) * VN) Vec { T v[VN]; };
  __shared__ float red[3 * VN * 256];
  const int g = blockIdx.x / chunks, ck = blockIdx.x - g * chunks;
  const int tpr = C / VN;
  const int rpb = 256 / tpr;
  const int roff = threadIdx.x / tpr, cp = threadIdx.x - roff * tpr;
  const long long r0 = R * ck / chunks, r1 = R * (ck + 1) / chunks;
  float s[3][VN];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int e = 0; e < VN; ++e) s[k][e] = 0.f;
  if (roff < rpb) {
    const long long base = (long long)g * R * C + VN * cp;
    for (long long r = r0 + roff; r < r1; r += rpb) {
      const long long o = base + r * C;
      const Vec va = *reinterpret_cast<const Vec*>(a + o);
      const Vec v0 = *reinterpret_cast<const Vec*>(b0 + o), v1 = *reinterpret_cast<const Vec*>(b1 + o), v2 = *reinterpret_cast<const Vec*>(b2 + o);
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        const float d = to_f32(va.v[e]);
        s[0][e] += d * to_f32(v0.v[e]);
        s[1][e] += d * to_f32(v1.v[e]);
        s[2][e] += d * to_f32(v2.v[e]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int e = 0; e < VN; ++e) red[(k * VN + e) * 256 + threadIdx.x] = s[k][e];
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += 256) {
    const int c = i / 3, k = i - 3 * c;
    const int cp2 = c / VN, e = c - cp2 * VN;
    float t = 0.f;
    for (int j = 0; j < rpb; ++j) t += red[(k * VN + e) * 256 + j * tpr + cp2];
    out[(((long long)g * chunks + ck) * C + c) * 3 + k] = t;
  }
}

// ------------------------------------------------------------------------------------------------ elementwise
enum { OP_CA_FWD = 0, OP_CA_BWD = 1, OP_MIX_FWD = 2, OP_MIX_BWD = 3, OP_GATE_FWD = 4, OP_GATE_BWD = 5, OP_AFFINE2 = 6, OP_SCALE = 7, OP_GATE_RES_FWD = 8, OP_GATE_RES_BWD = 9 };

template <typename T>
__global__ __launch_bounds__(256) void tab_ew_kernel(int op, const T* __restrict__ p0, const T* __restrict__ p1, const T* __restrict__ p2,
                                                     const float* __restrict__ coef, const float* __restrict__ add, float s,
                                                     T* __restrict__ o0, T* __restrict__ o1, T* __restrict__ o2, long long rows,
                                                     long long R, int C) {
  constexpr int VN = V16<T>::N;
  const int nvec = C / VN;
  const long long total = rows * nvec;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long row = i / nvec;
    const int v = (int)(i - row * nvec);
    const long long g = row / R;
    const long long off = row * C + v * VN;
    const long long gc = g * C + v * VN;
    V16<T> a = *reinterpret_cast<const V16<T>*>(p0 + off);
    V16<T> r0, r1, r2;
    if (op == OP_CA_FWD) {  // (r * g + x) * s
      const V16<T> x = *reinterpret_cast<const V16<T>*>(p1 + off);
#pragma unroll
      for (int e = 0; e < VN; ++e) r0.v[e] = from_f32<T>((to_f32(a.v[e]) * coef[gc + e] + to_f32(x.v[e])) * s);
      *reinterpret_cast<V16<T>*>(o0 + off) = r0;
    } else if (op == OP_SCALE) {  // p0 * g * s, coef (G, C): the DropPath residual's gradient w.r.t. the dropped branch
#pragma unroll
      for (int e = 0; e < VN; ++e) r0.v[e] = from_f32<T>(to_f32(a.v[e]) * coef[gc + e] * s);
      *reinterpret_cast<V16<T>*>(o0 + off) = r0;
    } else if (op == OP_CA_BWD) {  // d_r = dy * s * g + add ; d_x = dy * s
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        const float d = to_f32(a.v[e]) * s;
        r0.v[e] = from_f32<T>(d * coef[gc + e] + add[gc + e]);
        r1.v[e] = from_f32<T>(d);
      }
      *reinterpret_cast<V16<T>*>(o0 + off) = r0;
      *reinterpret_cast<V16<T>*>(o1 + off) = r1;
    } else if (op == OP_MIX_FWD) {  // h*a0 + w*a1 + c*a2, coef (G, C, 3)
      const V16<T> w = *reinterpret_cast<const V16<T>*>(p1 + off);
      const V16<T> c = *reinterpret_cast<const V16<T>*>(p2 + off);
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        const float* k = coef + (gc + e) * 3;
        r0.v[e] = from_f32<T>(to_f32(a.v[e]) * k[0] + to_f32(w.v[e]) * k[1] + to_f32(c.v[e]) * k[2]);
      }
      *reinterpret_cast<V16<T>*>(o0 + off) = r0;
    } else if (op == OP_MIX_BWD) {  // d_k = dy * a_k + add
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        const float* k = coef + (gc + e) * 3;
        const float d = to_f32(a.v[e]), ad = add[gc + e];
        r0.v[e] = from_f32<T>(d * k[0] + ad);
        r1.v[e] = from_f32<T>(d * k[1] + ad);
        r2.v[e] = from_f32<T>(d * k[2] + ad);
      }
      *reinterpret_cast<V16<T>*>(o0 + off) = r0;
      *reinterpret_cast<V16<T>*>(o1 + off) = r1;
      *reinterpret_cast<V16<T>*>(o2 + off) = r2;
    } else if (op == OP_AFFINE2) {  // p0*k0 + p1*k1 + add, coef (G, C, 2), p1 optional; ReLU when s > 0.5 (GroupNorm(1)+ReLU and its backward)
      V16<T> q;
      if (p1) q = *reinterpret_cast<const V16<T>*>(p1 + off);
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        const float* k = coef + (gc + e) * 2;
        float t = to_f32(a.v[e]) * k[0] + add[gc + e];
        if (p1) t += to_f32(q.v[e]) * k[1];
        r0.v[e] = from_f32<T>(s > 0.5f ? fmaxf(t, 0.f) : t);
      }
      *reinterpret_cast<V16<T>*>(o0 + off) = r0;
    } else if (op == OP_GATE_RES_FWD) {  // res + (x + y) * tanh(y) * g: p0 = x, p1 = y, p2 = res, coef (G, C) -- the mixer's gate and the TAB residual
      const V16<T> y = *reinterpret_cast<const V16<T>*>(p1 + off);                       // (with its DropPath coefficient) in ONE pass
      const V16<T> r = *reinterpret_cast<const V16<T>*>(p2 + off);
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        const float yv = to_f32(y.v[e]);
        // (the gate is rounded to T first, as the two-pass form stored it: same bits)
        const float gate = to_f32(from_f32<T>((to_f32(a.v[e]) + yv) * tanhf(yv)));
        r0.v[e] = from_f32<T>((gate * coef[gc + e] + to_f32(r.v[e])) * s);
      }
      *reinterpret_cast<V16<T>*>(o0 + off) = r0;
    } else if (op == OP_GATE_RES_BWD) {  // p0 = dout, p1 = x, p2 = y: d = dout * g (rounded to T like the two-pass form), dx = d*t, dy = d*(t + (x+y)(1-t^2))
      const V16<T> x = *reinterpret_cast<const V16<T>*>(p1 + off);
      const V16<T> y = *reinterpret_cast<const V16<T>*>(p2 + off);
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        const float d = to_f32(from_f32<T>(to_f32(a.v[e]) * coef[gc + e] * s)), xv = to_f32(x.v[e]), yv = to_f32(y.v[e]);
        const float t = tanhf(yv);
        r0.v[e] = from_f32<T>(d * t);
        r1.v[e] = from_f32<T>(d * (t + (xv + yv) * (1.f - t * t)));
      }
      *reinterpret_cast<V16<T>*>(o0 + off) = r0;
      *reinterpret_cast<V16<T>*>(o1 + off) = r1;
    } else if (op == OP_GATE_FWD) {  // (x + y) * tanh(y): p0 = x, p1 = y
      const V16<T> y = *reinterpret_cast<const V16<T>*>(p1 + off);
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        const float yv = to_f32(y.v[e]);
        r0.v[e] = from_f32<T>((to_f32(a.v[e]) + yv) * tanhf(yv));
      }
      *reinterpret_cast<V16<T>*>(o0 + off) = r0;
    } else {  // OP_GATE_BWD: p0 = dy, p1 = x, p2 = y -> dx = dy*t ; dy_ = dy*(t + (x+y)*(1-t^2))
      const V16<T> x = *reinterpret_cast<const V16<T>*>(p1 + off);
      const V16<T> y = *reinterpret_cast<const V16<T>*>(p2 + off);
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        const float d = to_f32(a.v[e]), xv = to_f32(x.v[e]), yv = to_f32(y.v[e]);
        const float t = tanhf(yv);
        r0.v[e] = from_f32<T>(d * t);
        r1.v[e] = from_f32<T>(d * (t + (xv + yv) * (1.f - t * t)));
      }
      *reinterpret_cast<V16<T>*>(o0 + off) = r0;
      *reinterpret_cast<V16<T>*>(o1 + off) = r1;
    }
  }
}

// ------------------------------------------------------------------------------------------------ f x f max pooling
// The multi-scale skip of VMG (models/vmg.py:388-400): adaptive_max_pool2d to (H/4, W/4) = non-overlapping 4 x 4 windows when the
// size divides.  Forward keeps the winner's position inside its window (first maximum in row-major order, as ATen does);
// backward writes every input element once: the window's gradient at the winner, zero elsewhere (no atomics, no memset).
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ src, T* __restrict__ dst, unsigned char* __restrict__ idx, int N, int H,
                                                      int W, int C, int f) {
  const int Ho = H / f, Wo = W / f;
  if (!BWD) {
    const long long total = (long long)N * Ho * Wo * C;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
      const int c = (int)(i % C);
      long long t = i / C;
      const int xo = (int)(t % Wo);
      t /= Wo;
      const int yo = (int)(t % Ho);
      const long long n = t / Ho;
      float best = -INFINITY;
      int bi = 0;
      for (int iy = 0; iy < f; ++iy)
        for (int ix = 0; ix < f; ++ix) {
          const float v = to_f32(src[((n * H + yo * f + iy) * W + xo * f + ix) * C + c]);
          if (v > best || (v != v && best == best)) { best = v; bi = iy * f + ix; }
        }
      dst[i] = from_f32<T>(best);
      idx[i] = (unsigned char)bi;
    }
  } else {
    const long long total = (long long)N * H * W * C;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
      const int c = (int)(i % C);
      long long t = i / C;
      const int x = (int)(t % W);
      t /= W;
      const int y = (int)(t % H);
      const long long n = t / H;
      const long long o = ((n * Ho + y / f) * Wo + x / f) * C + c;
      const int local = (y % f) * f + (x % f);
      dst[i] = idx[o] == local ? src[o] : from_f32<T>(0.f);
    }
  }
}

// ---- squeeze-excite MLP on pooled vectors: CALayer.conv_du (reference models/function.py:542-558: 1x1 conv, ReLU, 1x1 conv, sigmoid)
// and Enhanced_MorphFCs_decay.reweight (models/function.py:791-793: Linear, GELU, Linear, softmax over the three branches), fp32.
// G pooled rows of C channels -> hidden Hd -> Co outputs.  The whole problem is a few hundred thousand MACs: one launch forward (a
// workgroup per row), two backward (rows, then parameters) replace chains of ~12 / ~25 elementwise and GEMM launches.  (A single
// 1024-thread workgroup doing all rows took 45 / 76 us: every phase is a chain of dependent L2 latencies.)
__device__ __forceinline__ float se_act(float x, int act) { return act == 1 ? fmaxf(x, 0.f) : 0.5f * x * (1.f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float se_dact(float x, int act) {
  return act == 1 ? (x > 0.f ? 1.f : 0.f) : 0.5f * (1.f + erff(x * 0.7071067811865476f)) + x * expf(-0.5f * x * x) * 0.3989422804014327f;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// pre (G, Hd) = W1 m + b1;  out (G, Co) = sigmoid(W2 act(pre) + b2)  (mode 0)  or softmax over consecutive triples (mode 1).
// One workgroup per pooled row; LDS: the row m (C floats) and act(pre) (Hd floats).
__global__ __launch_bounds__(1024) void se_mlp_fwd_kernel(const float* __restrict__ m, const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ pre,
                                                         float* __restrict__ out, int C, int Hd, int Co, int act1, int mode) {
  extern __shared__ float sm[];
  float* mrow = sm;       // C
  float* z1 = sm + C;     // Hd
  // (16 waves: the hidden units are a chain of dependent L2 latencies per wave -- 36 units take 3 rounds instead of 9 with 4 waves)
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nt = blockDim.x, nw = nt >> 6;
  for (int c = tid; c < C; c += nt) mrow[c] = m[(long long)g * C + c];
  __syncthreads();
  for (int j = wave; j < Hd; j += nw) {  // a wave per hidden unit: lanes stride over the C inputs (coalesced W1 row)
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += w1[(long long)j * C + c] * mrow[c];
    s = wave_sum(s) + (b1 ? b1[j] : 0.f);
    if (lane == 0) { pre[(long long)g * Hd + j] = s; z1[j] = se_act(s, act1); }
  }
  __syncthreads();
  if (mode == 0) {
    for (int o = tid; o < Co; o += nt) {
      float s = b2 ? b2[o] : 0.f;
      const float* wr = w2 + (long long)o * Hd;
#pragma unroll 12
      for (int j = 0; j < Hd; ++j) s += wr[j] * z1[j];
      out[(long long)g * Co + o] = 1.f / (1.f + expf(-s));
    }
  } else {
    for (int c = tid; c < Co / 3; c += nt) {
      float z[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int o = 3 * c + k;
        float s = b2 ? b2[o] : 0.f;
        const float* wr = w2 + (long long)o * Hd;
#pragma unroll 12
        for (int j = 0; j < Hd; ++j) s += wr[j] * z1[j];
        z[k] = s;
      }
      const float mx = fmaxf(z[0], fmaxf(z[1], z[2]));
      const float e0 = expf(z[0] - mx), e1 = expf(z[1] - mx), e2 = expf(z[2] - mx), inv = 1.f / (e0 + e1 + e2);
      float* orow = out + (long long)g * Co + 3 * c;
      orow[0] = e0 * inv; orow[1] = e1 * inv; orow[2] = e2 * inv;
    }
  }
}

// Backward, row part (one workgroup per pooled row): dz2 = d(out)/d(logits) applied to dout, dz1 = (dz2 W2) * act'(pre), dm = dm_scale * dz1 W1.
// dz2 (G, Co) and dz1 (G, Hd) go to the workspace for the parameter-gradient kernel.
__global__ __launch_bounds__(1024) void se_mlp_bwd_rows_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                              const float* __restrict__ pre, const float* __restrict__ w1,
                                                              const float* __restrict__ w2, float* __restrict__ dm, float* __restrict__ ws, int G,
                                                              int C, int Hd, int Co, int act1, int mode, float dm_scale) {
  extern __shared__ float sm[];
  float* dz2 = sm;        // Co
  float* dz1 = sm + Co;   // Hd
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nt = blockDim.x, nw = nt >> 6;
  float* gz2 = ws + (long long)g * Co;
  float* gz1 = ws + (long long)G * Co + (long long)g * Hd;
  const float* orow = out + (long long)g * Co;
  const float* drow = dout + (long long)g * Co;
  if (mode == 0) {
    for (int o = tid; o < Co; o += nt) { const float v = drow[o] * orow[o] * (1.f - orow[o]); dz2[o] = v; gz2[o] = v; }
  } else {
    for (int c = tid; c < Co / 3; c += nt) {
      const float a0 = orow[3 * c], a1 = orow[3 * c + 1], a2 = orow[3 * c + 2];
      const float d0 = drow[3 * c], d1 = drow[3 * c + 1], d2 = drow[3 * c + 2];
      const float sd = a0 * d0 + a1 * d1 + a2 * d2;
      const float v0 = a0 * (d0 - sd), v1 = a1 * (d1 - sd), v2 = a2 * (d2 - sd);
      dz2[3 * c] = v0; dz2[3 * c + 1] = v1; dz2[3 * c + 2] = v2;
      gz2[3 * c] = v0; gz2[3 * c + 1] = v1; gz2[3 * c + 2] = v2;
    }
  }
  __syncthreads();
  for (int j = wave; j < Hd; j += nw) {  // lanes stride over the Co outputs (column j of W2: a gather, served by L2)
    float s = 0.f;
    for (int o = lane; o < Co; o += 64) s += dz2[o] * w2[(long long)o * Hd + j];
    s = wave_sum(s) * se_dact(pre[(long long)g * Hd + j], act1);
    if (lane == 0) { dz1[j] = s; gz1[j] = s; }
  }
  __syncthreads();
  for (int c = tid; c < C; c += nt) {
    float s = 0.f;
#pragma unroll 12
    for (int j = 0; j < Hd; ++j) s += dz1[j] * w1[(long long)j * C + c];
    dm[(long long)g * C + c] = s * dm_scale;
  }
}

// Backward, parameter part: one thread per element of dw1 (Hd, C), dw2 (Co, Hd), db1, db2, summing over the G rows in order (deterministic).
__global__ __launch_bounds__(256) void se_mlp_bwd_params_kernel(const float* __restrict__ m, const float* __restrict__ pre, const float* __restrict__ ws,
                                                                float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                                                float* __restrict__ db2, int G, int C, int Hd, int Co, int act1, int accumulate) {
  const float* dz2 = ws;
  const float* dz1 = ws + (long long)G * Co;
  const int n1 = Hd * C, n2 = Co * Hd;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  if (i < n1) {
    const int j = i / C, c = i - j * C;
    for (int g = 0; g < G; ++g) s += dz1[g * Hd + j] * m[(long long)g * C + c];
    dw1[i] = accumulate ? dw1[i] + s : s;
  } else if (i < n1 + n2) {
    const int e = i - n1, o = e / Hd, j = e - o * Hd;
    for (int g = 0; g < G; ++g) s += dz2[(long long)g * Co + o] * se_act(pre[g * Hd + j], act1);
    dw2[e] = accumulate ? dw2[e] + s : s;
  } else if (i < n1 + n2 + Hd) {
    const int j = i - n1 - n2;
    for (int g = 0; g < G; ++g) s += dz1[g * Hd + j];
    db1[j] = accumulate ? db1[j] + s : s;
  } else if (i < n1 + n2 + Hd + Co) {
    const int o = i - n1 - n2 - Hd;
    for (int g = 0; g < G; ++g) s += dz2[(long long)g * Co + o];
    db2[o] = accumulate ? db2[o] + s : s;
  }
}

}  // namespace

extern "C" int vmg_se_mlp_fwd(const float* m, const float* w1, const float* b1, const float* w2, const float* b2, float* pre, float* out, int G,
                              int C, int Hd, int Co, int act1, int mode, void* stream) {
  VMG_CHECK(m && w1 && w2 && pre && out && G > 0 && C > 0 && Hd > 0 && Co > 0, "se_mlp_fwd: bad arguments");
  VMG_CHECK((act1 == 1 || act1 == 3) && (mode == 0 || (mode == 1 && Co % 3 == 0)), "se_mlp_fwd: act1 is ReLU (1) or GELU (3); mode 1 needs Co = 3 * channels");
  VMG_CHECK((C + Hd) * 4 <= 64 * 1024, "se_mlp_fwd: C + Hd too large");
  hipLaunchKernelGGL(se_mlp_fwd_kernel, dim3(G), dim3(1024), (C + Hd) * 4, (hipStream_t)stream, m, w1, b1, w2, b2, pre, out, C, Hd, Co, act1, mode);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_se_mlp_bwd(const float* dout, const float* out, const float* m, const float* pre, const float* w1, const float* w2, float* dm,
                              float* dw1, float* db1, float* dw2, float* db2, float* ws, int G, int C, int Hd, int Co, int act1, int mode,
                              float dm_scale, int accumulate, void* stream) {
  VMG_CHECK(dout && out && m && pre && w1 && w2 && dm && dw1 && db1 && dw2 && db2 && ws && G > 0 && C > 0 && Hd > 0 && Co > 0, "se_mlp_bwd: bad arguments");
  VMG_CHECK((act1 == 1 || act1 == 3) && (mode == 0 || (mode == 1 && Co % 3 == 0)), "se_mlp_bwd: act1 is ReLU (1) or GELU (3); mode 1 needs Co = 3 * channels");
  VMG_CHECK((Co + Hd) * 4 <= 64 * 1024, "se_mlp_bwd: Co + Hd too large");
  hipLaunchKernelGGL(se_mlp_bwd_rows_kernel, dim3(G), dim3(1024), (Co + Hd) * 4, (hipStream_t)stream, dout, out, pre, w1, w2, dm, ws, G, C, Hd, Co, act1,
                     mode, dm_scale);
  VMG_LAUNCH_CHECK();
  const int total = Hd * C + Co * Hd + Hd + Co;
  hipLaunchKernelGGL(se_mlp_bwd_params_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, m, pre, (const float*)ws, dw1, db1, dw2, db2, G,
                     C, Hd, Co, act1, accumulate);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_maxpool_fwd(int dtype, const void* x, void* y, unsigned char* idx, int N, int H, int W, int C, int f, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "maxpool_fwd: bad dtype");
  VMG_CHECK(x && y && idx && N > 0 && C > 0 && f > 0 && f <= 15 && H % f == 0 && W % f == 0 && H >= f && W >= f, "maxpool_fwd: the window must divide the image");
  const long long total = (long long)N * (H / f) * (W / f) * C;
  const int blocks = (int)(cdiv64(total, 256) > 8192 ? 8192 : cdiv64(total, 256));
  if (dtype == VMG_BF16) hipLaunchKernelGGL((maxpool_kernel<bf16, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, idx, N, H, W, C, f);
  else hipLaunchKernelGGL((maxpool_kernel<float, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, idx, N, H, W, C, f);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_maxpool_bwd(int dtype, const void* dy, const unsigned char* idx, void* dx, int N, int H, int W, int C, int f, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "maxpool_bwd: bad dtype");
  VMG_CHECK(dy && dx && idx && N > 0 && C > 0 && f > 0 && f <= 15 && H % f == 0 && W % f == 0, "maxpool_bwd: the window must divide the image");
  const long long total = (long long)N * H * W * C;
  const int blocks = (int)(cdiv64(total, 256) > 8192 ? 8192 : cdiv64(total, 256));
  if (dtype == VMG_BF16) hipLaunchKernelGGL((maxpool_kernel<bf16, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy, (bf16*)dx, const_cast<unsigned char*>(idx), N, H, W, C, f);
  else hipLaunchKernelGGL((maxpool_kernel<float, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (float*)dx, const_cast<unsigned char*>(idx), N, H, W, C, f);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t vmg_group_reduce_ws_bytes() { return (int64_t)1024 * 3 * 512 * sizeof(float); }  // [<= 1024 blocks][<= 3 * 512 sums]

static int gr_chunks(int G, int64_t R) {
  int chunks = (int)(1024 / G);
#ifdef VMG_DIAG
  { const char* e = getenv("VMG_GR_BLOCKS"); if (e && atoi(e) > 0) chunks = atoi(e) / G; }  // (tools/bench_group_reduce.py)
#endif
  if (chunks < 1) chunks = 1;
  if (chunks > R / 64) chunks = (int)(R / 64 > 0 ? R / 64 : 1);
  return chunks;
}

extern "C" int vmg_group_reduce(int dtype, const void* a, const void* b, const void* c3, float* out, int G, int64_t R, int C, int mode,
                                float scale, float* ws, int64_t ws_bytes, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "group_reduce: bad dtype");
  VMG_CHECK(a && out && ws && G > 0 && R > 0 && C > 0 && (C & 1) == 0 && C <= 512, "group_reduce: bad arguments (C even, <= 512)");
  VMG_CHECK(mode == 0 || (mode == 1 && b), "group_reduce: mode 1 needs b");
  const int chunks = gr_chunks(G, R);
  VMG_CHECK((int64_t)G * chunks * C * (int64_t)sizeof(float) <= ws_bytes, "group_reduce: workspace too small (vmg_group_reduce_ws_bytes)");
  hipStream_t st = (hipStream_t)stream;
  const bool al16 = ((uintptr_t)a % 16 == 0) && (!b || (uintptr_t)b % 16 == 0) && (!c3 || (uintptr_t)c3 % 16 == 0);
  if (dtype == VMG_BF16) {
    if (C % 8 == 0 && al16)
      hipLaunchKernelGGL((group_reduce_kernel<bf16, 8>), dim3(G * chunks), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (const bf16*)c3, ws,
                         G, (long long)R, C, mode, scale, chunks);
    else
      hipLaunchKernelGGL((group_reduce_kernel<bf16, 2>), dim3(G * chunks), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (const bf16*)c3, ws,
                         G, (long long)R, C, mode, scale, chunks);
  } else {
    if (C % 4 == 0 && al16)
      hipLaunchKernelGGL((group_reduce_kernel<float, 4>), dim3(G * chunks), dim3(256), 0, st, (const float*)a, (const float*)b, (const float*)c3,
                         ws, G, (long long)R, C, mode, scale, chunks);
    else
      hipLaunchKernelGGL((group_reduce_kernel<float, 2>), dim3(G * chunks), dim3(256), 0, st, (const float*)a, (const float*)b, (const float*)c3,
                         ws, G, (long long)R, C, mode, scale, chunks);
  }
  VMG_LAUNCH_CHECK();
  hipLaunchKernelGGL(group_reduce_final_kernel, dim3(G * cdiv(C, 16)), dim3(256), 0, st, (const float*)ws, out, G, chunks, C, scale);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_group_reduce3(int dtype, const void* a, const void* b0, const void* b1, const void* b2, float* out, int G, int64_t R, int C,
                                 float scale, float* ws, int64_t ws_bytes, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "group_reduce3: bad dtype");
  const int vn = dtype == VMG_BF16 ? 8 : 4;
  VMG_CHECK(a && b0 && b1 && b2 && out && ws && G > 0 && R > 0 && C > 0 && C % vn == 0 && C / vn <= 256 && C <= 512, "group_reduce3: bad arguments (C a multiple of %d, <= 512)", vn);
  VMG_CHECK((((uintptr_t)a | (uintptr_t)b0 | (uintptr_t)b1 | (uintptr_t)b2) % 16) == 0, "group_reduce3: pointers must be 16-byte aligned");
  const int chunks = gr_chunks(G, R);
  VMG_CHECK((int64_t)G * chunks * C * 3 * (int64_t)sizeof(float) <= ws_bytes, "group_reduce3: workspace too small (vmg_group_reduce_ws_bytes)");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VMG_BF16)
    hipLaunchKernelGGL((group_reduce3_kernel<bf16, 8>), dim3(G * chunks), dim3(256), 0, st, (const bf16*)a, (const bf16*)b0, (const bf16*)b1,
                       (const bf16*)b2, ws, G, (long long)R, C, scale, chunks);
  else
    hipLaunchKernelGGL((group_reduce3_kernel<float, 4>), dim3(G * chunks), dim3(256), 0, st, (const float*)a, (const float*)b0, (const float*)b1,
                       (const float*)b2, ws, G, (long long)R, C, scale, chunks);
  VMG_LAUNCH_CHECK();
  hipLaunchKernelGGL(group_reduce_final_kernel, dim3(G * cdiv(3 * C, 16)), dim3(256), 0, st, (const float*)ws, out, G, chunks, 3 * C, scale);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_tab_elementwise(int dtype, int op, const void* p0, const void* p1, const void* p2, const float* coef, const float* add,
                                   float s, void* o0, void* o1, void* o2, int64_t rows, int64_t R, int C, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "tab_elementwise: bad dtype");
  VMG_CHECK(op >= 0 && op <= 9 && p0 && o0 && rows > 0 && R > 0 && C > 0, "tab_elementwise: bad arguments");
  const int vn = dtype == VMG_BF16 ? 8 : 4;
  VMG_CHECK(C % vn == 0, "tab_elementwise: C must be a multiple of %d", vn);
  const long long total = rows * (C / vn);
  const int blocks = (int)(cdiv64(total, 256) > 8192 ? 8192 : cdiv64(total, 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VMG_BF16)
    hipLaunchKernelGGL(tab_ew_kernel<bf16>, dim3(blocks), dim3(256), 0, st, op, (const bf16*)p0, (const bf16*)p1, (const bf16*)p2, coef, add, s,
                       (bf16*)o0, (bf16*)o1, (bf16*)o2, (long long)rows, (long long)R, C);
  else
    hipLaunchKernelGGL(tab_ew_kernel<float>, dim3(blocks), dim3(256), 0, st, op, (const float*)p0, (const float*)p1, (const float*)p2, coef, add,
                       s, (float*)o0, (float*)o1, (float*)o2, (long long)rows, (long long)R, C);
  VMG_LAUNCH_CHECK();
  return 0;
}
