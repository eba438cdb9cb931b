// Trajectory window attention (LTAM_multi_head.forward_wins, models/trajectory.py:672-795, cal_pe :534-547) fused into
// one forward and one backward kernel, channels-last, gfx950.
//
// Per 2x2 window (wq = 4 queries) and head: keys/values are, for each key-frame j (oldest first), the 4 window
// positions' features GATHERED at the tracked locations (nearest, zeros padding, align_corners=True); q and the
// gathered keys are L2-normalised over ALL C channels (eps 1e-12) before the head split;
//     logit[q][j,k] = scale * <qn_q, kn_{j,k}>_head + decay_v[head]^(t-j) * rpe[head][q][k];   softmax over (j,k);  out = P V.
// The reference materialises gathers, normalisations, three rearranges, two batched matmuls and a softmax (about 20
// kernels and as many full-tensor round trips); here a workgroup stages an 8x8-pixel tile of q and, key-frame by
// key-frame, the gathered K and V rows in LDS and runs an online softmax.  This is HBM/gather-bound work (8 GMAC of
// 2 740 in the trajectory stage): no MFMA -- the contractions are 4x4xD per window.
//
// Thread mapping: 256 threads = 64 tile pixels x 4 heads (thread = pixel * 4 + head); a thread owns the D = C/4
// channel slice of its head for its pixel, as a query (forward, dq) and as a key/value position (dK, dV).
#include "common.h"

namespace {

constexpr int LT_TILE = 8;        // tile edge in pixels
constexpr int LT_PIX = 64;        // pixels per tile
constexpr int LT_HEADS = 4;       // heads (all shipped configs: traj_heads = 4)
constexpr int LT_MAX_T = 32;      // key-frames per call (cfg4: ceil(50/3) = 17)

struct LtamK {
  const char* q;
  const char* k[LT_MAX_T];
  const char* v[LT_MAX_T];
  const float* loc;    // (n, 2t, h, w)
  const float* rpe;    // (heads, wq, wq)
  const float* decay;  // (heads)
  char* out;           // (n,h,w,c)
  float* lse;          // (n,h,w,heads) log-sum-exp of the logits
  // backward
  const char* dout;
  char* dq;            // (n,h,w,c) T
  float* dk_acc[LT_MAX_T];  // fp32 (n,h,w,c) accumulators, zero-initialised
  float* dv_acc[LT_MAX_T];
  float* drpe;         // (heads, wq, wq) fp32, accumulated
  int n, h, w, c, t, wh, ww;
  float scale;
  int tiles_x, tiles_y;
};

__device__ __forceinline__ int nearest_index(float lx, float ly, int w, int h) {
  // grid = 2*l/max(size-1,1) - 1; ATen unnormalise ((g+1)/2)*(size-1); nearbyint; zeros padding -> -1
  const float dx = (float)(w - 1 > 1 ? w - 1 : 1), dy = (float)(h - 1 > 1 ? h - 1 : 1);
  const float gx = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, lx), dx), 1.0f);
  const float gy = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, ly), dy), 1.0f);
  const float ix = __fmul_rn(__fdiv_rn(__fadd_rn(gx, 1.0f), 2.0f), (float)(w - 1));
  const float iy = __fmul_rn(__fdiv_rn(__fadd_rn(gy, 1.0f), 2.0f), (float)(h - 1));
  const float xn = nearbyintf(ix), yn = nearbyintf(iy);
  if (!(xn >= 0.f && xn <= (float)(w - 1) && yn >= 0.f && yn <= (float)(h - 1))) return -1;
  return (int)yn * w + (int)xn;
}

__device__ __forceinline__ float quad_sum(float v) {  // sum over the 4 head-threads of one pixel (aligned lanes)
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}

// cooperative copy of one pixel row (c elements) per tile pixel into LDS; src_idx < 0 -> zeros
template <typename T>
__device__ __forceinline__ void stage_rows(T* dst, int row_stride, const T* src_base, const int* src_idx, int c, int tid) {
  constexpr int VN = 16 / sizeof(T);
  const int nvec = c / VN;
  for (int i = tid; i < LT_PIX * nvec; i += 256) {
    const int p = i / nvec, v = i - p * nvec;
    uint4 val = make_uint4(0, 0, 0, 0);
    const int s = src_idx[p];
    if (s >= 0) val = *reinterpret_cast<const uint4*>(src_base + (long long)s * c + v * VN);
    *reinterpret_cast<uint4*>(dst + p * row_stride + v * VN) = val;
  }
}

template <typename T, int D>
__global__ __launch_bounds__(256) void ltam_fwd_kernel(const LtamK a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int c = a.c, RS = c + 16 / sizeof(T);  // padded row stride (elements)
  T* qt = reinterpret_cast<T*>(smem);
  T* kt = qt + LT_PIX * RS;
  T* vt = kt + LT_PIX * RS;
  float* knorm = reinterpret_cast<float*>(vt + LT_PIX * RS);
  int* sidx = reinterpret_cast<int*>(knorm + LT_PIX);
  int* selfidx = sidx + LT_PIX;

  const int tid = threadIdx.x, p = tid >> 2, hd = tid & 3;
  const int tyi = p >> 3, txi = p & 7;
  int bid = blockIdx.x;
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y, n = bid / a.tiles_y;
  const int y = ty * LT_TILE + tyi, x = tx * LT_TILE + txi;
  const bool inside = y < a.h && x < a.w;
  const long long img = (long long)n * a.h * a.w;
  if (tid < LT_PIX) {
    const int yy = ty * LT_TILE + (tid >> 3), xx = tx * LT_TILE + (tid & 7);
    selfidx[tid] = (yy < a.h && xx < a.w) ? yy * a.w + xx : -1;
  }
  __syncthreads();
  stage_rows<T>(qt, RS, reinterpret_cast<const T*>(a.q) + img * c, selfidx, c, tid);
  __syncthreads();
  // normalised, scaled query slice in registers
  float qs[D];
  float ss = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) { qs[d] = to_f32(qt[p * RS + hd * D + d]); ss += qs[d] * qs[d]; }
  {
    const float inv = a.scale / fmaxf(sqrtf(quad_sum(ss)), 1e-12f);
#pragma unroll
    for (int d = 0; d < D; ++d) qs[d] *= inv;
  }
  const int wq = a.wh * a.ww;
  const int wy0 = (tyi / a.wh) * a.wh, wx0 = (txi / a.ww) * a.ww;  // window origin inside the tile
  const int qi = (tyi - wy0) * a.ww + (txi - wx0);
  const float dec = a.decay[hd];
  float m = -INFINITY, l = 0.f, acc[D];
#pragma unroll
  for (int d = 0; d < D; ++d) acc[d] = 0.f;

  for (int j = 0; j < a.t; ++j) {
    __syncthreads();  // previous key-frame's tiles are no longer read
    if (tid < LT_PIX) {
      int s = -1;
      if (selfidx[tid] >= 0) {
        const long long lb = ((long long)n * 2 * a.t + 2 * j) * a.h * a.w + selfidx[tid];
        s = nearest_index(a.loc[lb], a.loc[lb + (long long)a.h * a.w], a.w, a.h);
      }
      sidx[tid] = s;
    }
    __syncthreads();
    stage_rows<T>(kt, RS, reinterpret_cast<const T*>(a.k[j]) + img * c, sidx, c, tid);
    stage_rows<T>(vt, RS, reinterpret_cast<const T*>(a.v[j]) + img * c, sidx, c, tid);
    __syncthreads();
    {
      float s2 = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { const float kv = to_f32(kt[p * RS + hd * D + d]); s2 += kv * kv; }
      s2 = quad_sum(s2);
      if (hd == 0) knorm[p] = fmaxf(sqrtf(s2), 1e-12f);
    }
    __syncthreads();
    float pw = dec;  // decay^(t - j) by repeated multiplication (cal_pe's cumulative product)
    for (int e = 1; e < a.t - j; ++e) pw *= dec;
    for (int ki = 0; ki < wq; ++ki) {
      const int pk = (wy0 + ki / a.ww) * LT_TILE + wx0 + ki % a.ww;
      float dot = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) dot += qs[d] * to_f32(kt[pk * RS + hd * D + d]);
      const float logit = dot / knorm[pk] + pw * a.rpe[(hd * wq + qi) * wq + ki];
      const float mn = fmaxf(m, logit);
      const float corr = __expf(m - mn), pe = __expf(logit - mn);
      l = l * corr + pe;
#pragma unroll
      for (int d = 0; d < D; ++d) acc[d] = acc[d] * corr + pe * to_f32(vt[pk * RS + hd * D + d]);
      m = mn;
    }
  }
  if (inside) {
    const float inv = 1.f / l;
    T* o = reinterpret_cast<T*>(a.out) + (img + (long long)y * a.w + x) * c + hd * D;
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = from_f32<T>(acc[d] * inv);
    if (a.lse) a.lse[(img + (long long)y * a.w + x) * LT_HEADS + hd] = m + __logf(l);
  }
}

// Backward.  For every key-frame the logits are recomputed; thread (pixel, head) acts
//   as a QUERY: accumulates dqn (gradient of its normalised query slice) over all keys,
//   as a KEY/VALUE position: accumulates dkn, dv of ITS gathered row over the wq queries of its window,
// then the normalisation Jacobians are applied and dK/dV rows are scattered (float atomics) to the gather sources.
template <typename T, int D>
__global__ __launch_bounds__(256) void ltam_bwd_kernel(const LtamK a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int c = a.c, RS = c + 16 / sizeof(T);
  T* kt = reinterpret_cast<T*>(smem);
  T* vt = kt + LT_PIX * RS;
  T* dot_ = vt + LT_PIX * RS;  // dout tile (read by the key-side pass of window mates)
  float* knorm = reinterpret_cast<float*>(dot_ + LT_PIX * RS);
  float* qnorm = knorm + LT_PIX;
  float* lse = qnorm + LT_PIX;             // [64][4]
  float* delta = lse + LT_PIX * LT_HEADS;  // [64][4]
  float* drpe_s = delta + LT_PIX * LT_HEADS;  // [heads][wq][wq] <= 4*16*16... wq <= 16
  int* sidx = reinterpret_cast<int*>(drpe_s + LT_HEADS * 16 * 16);
  int* selfidx = sidx + LT_PIX;

  const int tid = threadIdx.x, p = tid >> 2, hd = tid & 3;
  const int tyi = p >> 3, txi = p & 7;
  int bid = blockIdx.x;
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y, n = bid / a.tiles_y;
  const int y = ty * LT_TILE + tyi, x = tx * LT_TILE + txi;
  const bool inside = y < a.h && x < a.w;
  const long long img = (long long)n * a.h * a.w;
  const int wq = a.wh * a.ww;
  if (tid < LT_PIX) {
    const int yy = ty * LT_TILE + (tid >> 3), xx = tx * LT_TILE + (tid & 7);
    selfidx[tid] = (yy < a.h && xx < a.w) ? yy * a.w + xx : -1;
  }
  for (int i = tid; i < LT_HEADS * wq * wq; i += 256) drpe_s[i] = 0.f;
  __syncthreads();
  stage_rows<T>(dot_, RS, reinterpret_cast<const T*>(a.dout) + img * c, selfidx, c, tid);
  __syncthreads();
  float qn[D], go[D];  // normalised query slice (unscaled), dout slice
  {
    float ss = 0.f, dl = 0.f;
    const long long own = (img + (long long)y * a.w + x) * c + hd * D;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      qn[d] = inside ? to_f32(reinterpret_cast<const T*>(a.q)[own + d]) : 0.f;
      go[d] = to_f32(dot_[p * RS + hd * D + d]);
      ss += qn[d] * qn[d];
      dl += inside ? go[d] * to_f32(reinterpret_cast<const T*>(a.out)[own + d]) : 0.f;
    }
    const float nr = fmaxf(sqrtf(quad_sum(ss)), 1e-12f);
    if (hd == 0) qnorm[p] = nr;
    const float inv = 1.f / nr;
#pragma unroll
    for (int d = 0; d < D; ++d) qn[d] *= inv;
    delta[p * LT_HEADS + hd] = dl;
    lse[p * LT_HEADS + hd] = inside ? a.lse[(img + (long long)y * a.w + x) * LT_HEADS + hd] : 0.f;
  }
  const int wy0 = (tyi / a.wh) * a.wh, wx0 = (txi / a.ww) * a.ww;
  const int myi = (tyi - wy0) * a.ww + (txi - wx0);  // this pixel's index inside its window (as query AND as key)
  const float dec = a.decay[hd];
  float dqn[D];
#pragma unroll
  for (int d = 0; d < D; ++d) dqn[d] = 0.f;
  for (int j = 0; j < a.t; ++j) {
    __syncthreads();
    if (tid < LT_PIX) {
      int s = -1;
      if (selfidx[tid] >= 0) {
        const long long lb = ((long long)n * 2 * a.t + 2 * j) * a.h * a.w + selfidx[tid];
        s = nearest_index(a.loc[lb], a.loc[lb + (long long)a.h * a.w], a.w, a.h);
      }
      sidx[tid] = s;
    }
    __syncthreads();
    stage_rows<T>(kt, RS, reinterpret_cast<const T*>(a.k[j]) + img * c, sidx, c, tid);
    stage_rows<T>(vt, RS, reinterpret_cast<const T*>(a.v[j]) + img * c, sidx, c, tid);
    __syncthreads();
    float kn[D];  // this thread's key slice (raw, then normalised)
    {
      float s2 = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { kn[d] = to_f32(kt[p * RS + hd * D + d]); s2 += kn[d] * kn[d]; }
      s2 = quad_sum(s2);
      const float nr = fmaxf(sqrtf(s2), 1e-12f);
      if (hd == 0) knorm[p] = nr;
      const float inv = 1.f / nr;
#pragma unroll
      for (int d = 0; d < D; ++d) kn[d] *= inv;
    }
    __syncthreads();
    float pw = dec;
    for (int e = 1; e < a.t - j; ++e) pw *= dec;

    // ---- query side: dqn += sum_k ds * scale * kn_k ; drpe
    {
      const float my_lse = lse[p * LT_HEADS + hd], my_delta = delta[p * LT_HEADS + hd];
      for (int ki = 0; ki < wq; ++ki) {
        const int pk = (wy0 + ki / a.ww) * LT_TILE + wx0 + ki % a.ww;
        const float inv_kn = 1.f / knorm[pk];
        float dot = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          dot += qn[d] * to_f32(kt[pk * RS + hd * D + d]);
          dp += go[d] * to_f32(vt[pk * RS + hd * D + d]);
        }
        const float logit = a.scale * dot * inv_kn + pw * a.rpe[(hd * wq + myi) * wq + ki];
        const float pr = __expf(logit - my_lse);
        const float ds = pr * (dp - my_delta);
        if (inside) atomicAdd(&drpe_s[(hd * wq + myi) * wq + ki], ds * pw);
        const float f = ds * a.scale * inv_kn;
#pragma unroll
        for (int d = 0; d < D; ++d) dqn[d] += f * to_f32(kt[pk * RS + hd * D + d]);
      }
    }
    // ---- key side: this thread's gathered row (p, key-frame j) against the wq queries of its window
    float dkn[D], dv[D];
#pragma unroll
    for (int d = 0; d < D; ++d) dkn[d] = dv[d] = 0.f;
    for (int qq = 0; qq < wq; ++qq) {
      const int pq = (wy0 + qq / a.ww) * LT_TILE + wx0 + qq % a.ww;
      if (selfidx[pq] < 0) continue;
      // query pq's normalised slice: recompute from the raw q tile would need its norm -> qnorm[] holds it
      const float inv_qn = 1.f / qnorm[pq];
      float dot = 0.f, dp = 0.f;
      const T* qraw = reinterpret_cast<const T*>(a.q) + (img + selfidx[pq]) * c + hd * D;
      const T* graw = dot_ + pq * RS + hd * D;
      float qv[D];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        qv[d] = to_f32(qraw[d]) * inv_qn;
        dot += qv[d] * kn[d];
        dp += to_f32(graw[d]) * to_f32(vt[p * RS + hd * D + d]);
      }
      const float logit = a.scale * dot + pw * a.rpe[(hd * wq + qq) * wq + myi];
      const float pr = __expf(logit - lse[pq * LT_HEADS + hd]);
      const float ds = pr * (dp - delta[pq * LT_HEADS + hd]);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        dkn[d] += ds * a.scale * qv[d];
        dv[d] += pr * to_f32(graw[d]);
      }
    }
    // normalisation Jacobian of the key row; then the rows are scattered to their gather sources.  The scatter goes
    // through LDS so that lanes run along CHANNELS: each atomic wave-instruction adds 64 consecutive floats (256
    // contiguous bytes = the full-rate shape of global float atomics here; a lane-per-head-slice scatter strides the
    // lanes by D floats and was ~10x slower).  The K/V tiles of this key-frame are dead by now and are reused as the
    // fp32 row buffer [64][c] (kt and vt are contiguous: 2 tiles hold 64*c floats for bf16 and fp32 alike).
    {
      float nd = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) nd += kn[d] * dkn[d];
      nd = quad_sum(nd);
      const float inv = 1.f / knorm[p];
      float* rowbuf = reinterpret_cast<float*>(kt);
      __syncthreads();  // every thread is done reading kt / vt of this key-frame
#pragma unroll
      for (int d = 0; d < D; ++d) rowbuf[p * c + hd * D + d] = (dkn[d] - kn[d] * nd) * inv;
      __syncthreads();
      for (int i = tid; i < LT_PIX * c; i += 256) {
        const int pp = i / c, ch = i - pp * c;
        const int sp = sidx[pp];
        if (sp >= 0 && selfidx[pp] >= 0) atomicAdd(a.dk_acc[j] + (img + sp) * c + ch, rowbuf[i]);
      }
      __syncthreads();
#pragma unroll
      for (int d = 0; d < D; ++d) rowbuf[p * c + hd * D + d] = dv[d];
      __syncthreads();
      for (int i = tid; i < LT_PIX * c; i += 256) {
        const int pp = i / c, ch = i - pp * c;
        const int sp = sidx[pp];
        if (sp >= 0 && selfidx[pp] >= 0) atomicAdd(a.dv_acc[j] + (img + sp) * c + ch, rowbuf[i]);
      }
    }
  }
  // query normalisation Jacobian
  {
    float nd = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) nd += qn[d] * dqn[d];
    nd = quad_sum(nd);
    if (inside) {
      const float inv = 1.f / qnorm[p];
      T* dq = reinterpret_cast<T*>(a.dq) + (img + (long long)y * a.w + x) * c + hd * D;
#pragma unroll
      for (int d = 0; d < D; ++d) dq[d] = from_f32<T>((dqn[d] - qn[d] * nd) * inv);
    }
  }
  __syncthreads();
  for (int i = tid; i < LT_HEADS * wq * wq; i += 256) atomicAdd(&a.drpe[i], drpe_s[i]);
}

template <typename T, int D>
int launch_ltam(const LtamK& k, bool backward, hipStream_t st) {
  const int RS = k.c + 16 / (int)sizeof(T);
  const int grid = k.n * k.tiles_y * k.tiles_x;
  if (!backward) {
    const int lds = 3 * LT_PIX * RS * (int)sizeof(T) + LT_PIX * 4 + 2 * LT_PIX * 4;
    auto fn = ltam_fwd_kernel<T, D>;
    static bool set[VMG_MAX_DEVICES] = {};  // the attribute is per device
    const int dev = vmg_current_device();
    if (!set[dev]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set[dev] = true; }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, st, k);
  } else {
    const int lds = 3 * LT_PIX * RS * (int)sizeof(T) + (2 * LT_PIX + 2 * LT_PIX * LT_HEADS + LT_HEADS * 256) * 4 + 2 * LT_PIX * 4;
    VMG_CHECK(lds <= 160 * 1024, "ltam_bwd: LDS request %d B exceeds 160 KiB", lds);
    auto fn = ltam_bwd_kernel<T, D>;
    static bool set[VMG_MAX_DEVICES] = {};  // the attribute is per device
    const int dev = vmg_current_device();
    if (!set[dev]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set[dev] = true; }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, st, k);
  }
  VMG_LAUNCH_CHECK();
  return 0;
}

template <typename T>
int dispatch_d(const LtamK& k, bool backward, hipStream_t st) {
  switch (k.c / LT_HEADS) {
    case 4: return launch_ltam<T, 4>(k, backward, st);
    case 8: return launch_ltam<T, 8>(k, backward, st);
    case 28: return launch_ltam<T, 28>(k, backward, st);
    case 36: return launch_ltam<T, 36>(k, backward, st);
  }
  vmg_set_error("ltam: head dim %d not instantiated (supported: 4, 8, 28, 36)", k.c / LT_HEADS);
  return -1;
}

int fill_common(LtamK& k, int dtype, const void* q, const void* const* keys, const void* const* vals, const float* loc, const float* rpe,
                const float* decay, int n, int h, int w, int c, int heads, int wh, int ww, int t, float scale) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "ltam: bad dtype");
  VMG_CHECK(q && keys && vals && loc && rpe && decay, "ltam: null pointer");
  VMG_CHECK(heads == LT_HEADS, "ltam: heads must be %d", LT_HEADS);
  VMG_CHECK(t >= 1 && t <= LT_MAX_T, "ltam: 1..%d key-frames", LT_MAX_T);
  VMG_CHECK(n > 0 && h > 0 && w > 0 && c % (LT_HEADS * (dtype == VMG_BF16 ? 8 : 4) / LT_HEADS) == 0 && c % LT_HEADS == 0, "ltam: bad shape");
  VMG_CHECK(wh >= 1 && ww >= 1 && LT_TILE % wh == 0 && LT_TILE % ww == 0 && h % wh == 0 && w % ww == 0 && wh * ww <= 16,
            "ltam: window %dx%d must divide 8 and the image", wh, ww);
  memset(&k, 0, sizeof(k));
  k.q = (const char*)q;
  for (int j = 0; j < t; ++j) {
    VMG_CHECK(keys[j] && vals[j], "ltam: null key/value %d", j);
    k.k[j] = (const char*)keys[j];
    k.v[j] = (const char*)vals[j];
  }
  k.loc = loc; k.rpe = rpe; k.decay = decay;
  k.n = n; k.h = h; k.w = w; k.c = c; k.t = t; k.wh = wh; k.ww = ww; k.scale = scale;
  k.tiles_x = cdiv(w, LT_TILE); k.tiles_y = cdiv(h, LT_TILE);
  return 0;
}

}  // namespace

extern "C" int vmg_ltam_fwd(int dtype, const void* q, const void* const* keys, const void* const* vals, const float* loc,
                            const float* rpe, const float* decay, void* out, float* lse, int n, int h, int w, int c, int heads, int wh,
                            int ww, int t, float scale, void* stream) {
  LtamK k;
  if (int rc = fill_common(k, dtype, q, keys, vals, loc, rpe, decay, n, h, w, c, heads, wh, ww, t, scale)) return rc;
  VMG_CHECK(out, "ltam_fwd: null output");
  k.out = (char*)out; k.lse = lse;
  return dtype == VMG_BF16 ? dispatch_d<bf16>(k, false, (hipStream_t)stream) : dispatch_d<float>(k, false, (hipStream_t)stream);
}

extern "C" int vmg_ltam_bwd(int dtype, const void* q, const void* const* keys, const void* const* vals, const float* loc,
                            const float* rpe, const float* decay, const void* out, const float* lse, const void* dout, void* dq,
                            float* const* dk_acc, float* const* dv_acc, float* drpe, int n, int h, int w, int c, int heads, int wh,
                            int ww, int t, float scale, void* stream) {
  LtamK k;
  if (int rc = fill_common(k, dtype, q, keys, vals, loc, rpe, decay, n, h, w, c, heads, wh, ww, t, scale)) return rc;
  VMG_CHECK(out && lse && dout && dq && dk_acc && dv_acc && drpe, "ltam_bwd: null pointer");
  k.out = (char*)out; k.lse = (float*)lse; k.dout = (const char*)dout; k.dq = (char*)dq; k.drpe = drpe;
  for (int j = 0; j < t; ++j) {
    VMG_CHECK(dk_acc[j] && dv_acc[j], "ltam_bwd: null accumulator %d", j);
    k.dk_acc[j] = dk_acc[j];
    k.dv_acc[j] = dv_acc[j];
  }
  return dtype == VMG_BF16 ? dispatch_d<bf16>(k, true, (hipStream_t)stream) : dispatch_d<float>(k, true, (hipStream_t)stream);
}
