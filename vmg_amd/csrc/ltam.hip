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
  void* dk_acc[LT_MAX_T];  // (n,h,w,c) fp32 accumulators, zero-initialised (or holding earlier calls' sums)
  void* dv_acc[LT_MAX_T];
  float* drpe;         // (heads, wq, wq) fp32, accumulated
  int n, h, w, c, t, wh, ww;
  float scale;
  int tiles_x, tiles_y;
  int dbg;  // diagnostics build only (env VMG_LTAM_DBG): 1 no scatter of dK / dV, 2 no key side, 4 no query side, 8 no gathers after the first key-frame
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

// cooperative copy of one pixel row (c = 4 D elements) per tile pixel into LDS; src_idx < 0 -> zeros.  Every load is UNCONDITIONAL (a row
// without a source reads the zero vector) and all of a thread's loads are issued before the first LDS store: loads behind a per-lane
// branch are each followed by s_waitcnt vmcnt(0), which serialised 4-5 gather latencies per tile and key-frame.
typedef __attribute__((ext_vector_type(4))) unsigned int lt_u32x4;  // (a native vector: the HIP uint4 struct behind a pointer select goes through scratch)
__device__ __attribute__((aligned(16))) unsigned int g_ltam_zero[4] = {0, 0, 0, 0};  // (not const: a constant-address-space pointer in the select turns the loads into flat_load)

template <typename T, int D>
struct RowCopy {
  static constexpr int VN = 16 / sizeof(T), NVEC = LT_HEADS * D / VN, TOTAL = LT_PIX * NVEC, NIT = (TOTAL + 255) / 256;
  lt_u32x4 r[NIT];
  __device__ __forceinline__ void fetch(const T* src_base, const int* src_idx, int tid) {
    constexpr int c = LT_HEADS * D;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = tid + k * 256;
      const int p = min(i / NVEC, LT_PIX - 1), v = i - (i / NVEC) * NVEC;
      const int s_ = src_idx[p];
      const bool ok = i < TOTAL && s_ >= 0;
      const T* src = ok ? src_base + (long long)s_ * c + v * VN : reinterpret_cast<const T*>(g_ltam_zero);
      r[k] = *reinterpret_cast<const lt_u32x4*>(src);
    }
  }
  __device__ __forceinline__ void store(T* dst, int row_stride, int tid) const {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = tid + k * 256;
      const int p = i / NVEC, v = i - p * NVEC;
      if (i < TOTAL) *reinterpret_cast<lt_u32x4*>(dst + p * row_stride + v * VN) = r[k];
    }
  }
};

// the D-channel slice of one head as floats, read with 8-byte (bf16) / 16-byte (fp32) vectors: D is a multiple of 4, rows and head slices
// start on those boundaries (c = 4 D; the LDS row stride is c + one 16-byte vector)
template <typename T, int D>
__device__ __forceinline__ void load_slice(const T* p, float (&o)[D]) {
  static_assert(D % 4 == 0, "head dim must be a multiple of 4");
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int i = 0; i < D / 4; ++i) {
      const bf16x4 t = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p) + 4 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[4 * i + e] = (float)t[e];
    }
  } else {
#pragma unroll
    for (int i = 0; i < D / 4; ++i) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + 4 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[4 * i + e] = t[e];
    }
  }
}

template <typename T, int D>
__global__ __launch_bounds__(256) void ltam_fwd_kernel(const LtamK a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int c = LT_HEADS * D, RS = c + 16 / sizeof(T);  // channels (== a.c: the dispatch picks D from it), padded row stride (elements)
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
  RowCopy<T, D> rk, rv;
  rk.fetch(reinterpret_cast<const T*>(a.q) + img * c, selfidx, tid);
  rk.store(qt, RS, tid);
  __syncthreads();
  // normalised, scaled query slice in registers
  float qs[D];
  float ss = 0.f;
  load_slice<T, D>(qt + p * RS + hd * D, qs);
#pragma unroll
  for (int d = 0; d < D; ++d) ss += qs[d] * qs[d];
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
    rk.fetch(reinterpret_cast<const T*>(a.k[j]) + img * c, sidx, tid);
    rv.fetch(reinterpret_cast<const T*>(a.v[j]) + img * c, sidx, tid);
    rk.store(kt, RS, tid);
    rv.store(vt, RS, tid);
    __syncthreads();
    {
      float s2 = 0.f, kv[D];
      load_slice<T, D>(kt + p * RS + hd * D, kv);
#pragma unroll
      for (int d = 0; d < D; ++d) s2 += kv[d] * kv[d];
      s2 = quad_sum(s2);
      if (hd == 0) knorm[p] = fmaxf(sqrtf(s2), 1e-12f);
    }
    __syncthreads();
    float pw = dec;  // decay^(t - j) by repeated multiplication (cal_pe's cumulative product)
    for (int e = 1; e < a.t - j; ++e) pw *= dec;
    for (int ki = 0; ki < wq; ++ki) {
      const int pk = (wy0 + ki / a.ww) * LT_TILE + wx0 + ki % a.ww;
      float dot = 0.f, kk[D], vv[D];
      load_slice<T, D>(kt + pk * RS + hd * D, kk);
      load_slice<T, D>(vt + pk * RS + hd * D, vv);
#pragma unroll
      for (int d = 0; d < D; ++d) dot += qs[d] * kk[d];
      const float logit = dot / knorm[pk] + pw * a.rpe[(hd * wq + qi) * wq + ki];
      const float mn = fmaxf(m, logit);
      const float corr = __expf(m - mn), pe = __expf(logit - mn);
      l = l * corr + pe;
#pragma unroll
      for (int d = 0; d < D; ++d) acc[d] = acc[d] * corr + pe * vv[d];
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
// Scatter-add of the tile's 64 gradient rows (fp32 in LDS, [64][C]) to their gather sources.  Lanes run along CHANNELS: every atomic
// wave-instruction adds 256 contiguous bytes -- the full-rate shape of global atomics here (a lane-per-head-slice scatter strides the lanes
// by D elements and was ~10x slower).  The accumulators are fp32 for every tensor dtype: a key-frame collects rows from up to six later
// frames and from several queries each; a bf16 running sum (packed bf16 atomics, round 2) rounded at every add, order-dependently.  The
// caller rounds the finished sums to bf16 once.
template <int C>
__device__ __forceinline__ void scatter_rows(void* acc, long long img, const float* rowbuf, const int* sidx, const int* selfidx, int tid, bool skip) {
  float* dst = reinterpret_cast<float*>(acc);
  for (int i = tid; i < LT_PIX * C; i += 256) {
    const int pp = i / C, ch = i - pp * C;
    const int sp = sidx[pp];
    if (sp >= 0 && selfidx[pp] >= 0 && !skip) atomicAdd(dst + (img + sp) * C + ch, rowbuf[i]);
  }
}

#ifdef VMG_DIAG
#define LT_ABL(bit) (a.dbg & (bit))
#else
#define LT_ABL(bit) false
#endif
template <typename T, int D>
__global__ __launch_bounds__(256) void ltam_bwd_kernel(const LtamK a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int c = LT_HEADS * D, RS = c + 16 / sizeof(T);
  T* kt = reinterpret_cast<T*>(smem);
  T* vt = kt + LT_PIX * RS;
  T* dot_ = vt + LT_PIX * RS;  // dout tile (read by the key-side pass of window mates)
  T* qt = dot_ + LT_PIX * RS;  // raw q tile (same)
  float* knorm = reinterpret_cast<float*>(qt + LT_PIX * RS);
  float* qnorm = knorm + LT_PIX;
  float* lse = qnorm + LT_PIX;             // [64][4]
  float* delta = lse + LT_PIX * LT_HEADS;  // [64][4]
  float* drpe_s = delta + LT_PIX * LT_HEADS;  // [heads][wq][wq]
  int* sidx = reinterpret_cast<int*>(drpe_s + LT_HEADS * a.wh * a.ww * a.wh * a.ww);
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
  RowCopy<T, D> rk, rv;
  rk.fetch(reinterpret_cast<const T*>(a.dout) + img * c, selfidx, tid);
  rv.fetch(reinterpret_cast<const T*>(a.q) + img * c, selfidx, tid);
  rk.store(dot_, RS, tid);
  rv.store(qt, RS, tid);
  __syncthreads();
  float qn[D], go[D];  // normalised query slice (unscaled), dout slice
  {
    float ss = 0.f, dl = 0.f, ov[D];
    const long long own = (img + (long long)(inside ? y * a.w + x : 0)) * c + hd * D;  // (tile pixels outside the image read pixel 0 and drop it)
    load_slice<T, D>(qt + p * RS + hd * D, qn);
    load_slice<T, D>(dot_ + p * RS + hd * D, go);
    load_slice<T, D>(reinterpret_cast<const T*>(a.out) + own, ov);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      ss += qn[d] * qn[d];
      dl += inside ? go[d] * ov[d] : 0.f;
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
    if (!(LT_ABL(8) && j > 0)) {
      rk.fetch(reinterpret_cast<const T*>(a.k[j]) + img * c, sidx, tid);
      rv.fetch(reinterpret_cast<const T*>(a.v[j]) + img * c, sidx, tid);
    }
    rk.store(kt, RS, tid);
    rv.store(vt, RS, tid);
    __syncthreads();
    float kn[D], vown[D];  // this thread's key slice (raw, then normalised) and value slice
    load_slice<T, D>(vt + p * RS + hd * D, vown);
    {
      float s2 = 0.f;
      load_slice<T, D>(kt + p * RS + hd * D, kn);
#pragma unroll
      for (int d = 0; d < D; ++d) s2 += kn[d] * kn[d];
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
    if (!LT_ABL(4)) {
      const float my_lse = lse[p * LT_HEADS + hd], my_delta = delta[p * LT_HEADS + hd];
      for (int ki = 0; ki < wq; ++ki) {
        const int pk = (wy0 + ki / a.ww) * LT_TILE + wx0 + ki % a.ww;
        const float inv_kn = 1.f / knorm[pk];
        float dot = 0.f, dp = 0.f, kk[D], vv[D];
        load_slice<T, D>(kt + pk * RS + hd * D, kk);
        load_slice<T, D>(vt + pk * RS + hd * D, vv);
#pragma unroll
        for (int d = 0; d < D; ++d) {
          dot += qn[d] * kk[d];
          dp += go[d] * vv[d];
        }
        const float logit = a.scale * dot * inv_kn + pw * a.rpe[(hd * wq + myi) * wq + ki];
        const float pr = __expf(logit - my_lse);
        const float ds = pr * (dp - my_delta);
        if (inside) atomicAdd(&drpe_s[(hd * wq + myi) * wq + ki], ds * pw);
        const float f = ds * a.scale * inv_kn;
#pragma unroll
        for (int d = 0; d < D; ++d) dqn[d] += f * kk[d];
      }
    }
    // ---- key side: this thread's gathered row (p, key-frame j) against the wq queries of its window
    float dkn[D], dv[D];
#pragma unroll
    for (int d = 0; d < D; ++d) dkn[d] = dv[d] = 0.f;
    for (int qq = 0; qq < wq; ++qq) {
      const int pq = (wy0 + qq / a.ww) * LT_TILE + wx0 + qq % a.ww;
      if (selfidx[pq] < 0 || LT_ABL(2)) continue;
      // query pq's normalised slice: recompute from the raw q tile would need its norm -> qnorm[] holds it
      const float inv_qn = 1.f / qnorm[pq];
      float dot = 0.f, dp = 0.f;
      float qv[D], gg[D];
      load_slice<T, D>(qt + pq * RS + hd * D, qv);
      load_slice<T, D>(dot_ + pq * RS + hd * D, gg);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        qv[d] *= inv_qn;
        dot += qv[d] * kn[d];
        dp += gg[d] * vown[d];
      }
      const float logit = a.scale * dot + pw * a.rpe[(hd * wq + qq) * wq + myi];
      const float pr = __expf(logit - lse[pq * LT_HEADS + hd]);
      const float ds = pr * (dp - delta[pq * LT_HEADS + hd]);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        dkn[d] += ds * a.scale * qv[d];
        dv[d] += pr * gg[d];
      }
    }
    // normalisation Jacobian of the key row; then the rows are scattered to their gather sources through LDS (scatter_rows).  The K/V
    // tiles of this key-frame are dead by now and are reused as the fp32 row buffer [64][c] (kt and vt are contiguous: 2 tiles hold
    // 64*c floats for bf16 and fp32 alike).
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
      scatter_rows<c>(a.dk_acc[j], img, rowbuf, sidx, selfidx, tid, LT_ABL(1));
      __syncthreads();
#pragma unroll
      for (int d = 0; d < D; ++d) rowbuf[p * c + hd * D + d] = dv[d];
      __syncthreads();
      scatter_rows<c>(a.dv_acc[j], img, rowbuf, sidx, selfidx, tid, LT_ABL(1));
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
    const int wq = k.wh * k.ww;
    // (bf16, 144 channels, 2 x 2 windows: 81 152 B -- two workgroups per CU)
    const int lds = 4 * LT_PIX * RS * (int)sizeof(T) + (2 * LT_PIX + 2 * LT_PIX * LT_HEADS + LT_HEADS * wq * wq) * 4 + 2 * LT_PIX * 4;
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
                            void* const* dk_acc, void* const* dv_acc, float* drpe, int n, int h, int w, int c, int heads, int wh,
                            int ww, int t, float scale, void* stream) {
  LtamK k;
  if (int rc = fill_common(k, dtype, q, keys, vals, loc, rpe, decay, n, h, w, c, heads, wh, ww, t, scale)) return rc;
  VMG_CHECK(out && lse && dout && dq && dk_acc && dv_acc && drpe, "ltam_bwd: null pointer");
#ifdef VMG_DIAG
  { const char* e = getenv("VMG_LTAM_DBG"); k.dbg = e ? atoi(e) : 0; }
#endif
  k.out = (char*)out; k.lse = (float*)lse; k.dout = (const char*)dout; k.dq = (char*)dq; k.drpe = drpe;
  for (int j = 0; j < t; ++j) {
    VMG_CHECK(dk_acc[j] && dv_acc[j], "ltam_bwd: null accumulator %d", j);
    k.dk_acc[j] = dk_acc[j];
    k.dv_acc[j] = dv_acc[j];
  }
  return dtype == VMG_BF16 ? dispatch_d<bf16>(k, true, (hipStream_t)stream) : dispatch_d<float>(k, true, (hipStream_t)stream);
}
