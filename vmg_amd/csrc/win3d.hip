// 3-D shifted-window attention of swin_3d.rWindowAttention (reference: models/swin_3d.py:167-252, window partition / roll / mask
// :55-118, :772-832): inside a (wt, 8, 8) window every time slice's 64 queries attend to the tokens of the OTHER slices, with a
// relative-position bias and, on shifted blocks, the -100 region mask.
//
// One workgroup = one (window, head), one thread = one token of the window (N = wt*64 threads).  Window partition, the cyclic
// roll of shifted blocks, the zero padding to window multiples, the mask and the bias gather are all INDEX ARITHMETIC here:
//   * q / kv are the Linear outputs on the un-partitioned (B, D, H, W, .) feature map (a per-token Linear commutes with any token
//     permutation); a token of a window is found by  shifted coordinate -> +shift mod padded size -> original coordinate;
//   * a padded position holds the Linear's bias (the reference pads zeros BEFORE the Linears: q = b_q, k/v = b_kv there);
//   * the bias is table[rel(q, k)][head] with rel computed from the two tokens' window coordinates (the head's column of the
//     table is staged in LDS); the mask is (region(q) != region(k)) ? -100 : 0 with the region from the shifted coordinate.
// K and V of the window (fp32, rows padded to float4) sit in LDS and are read as broadcasts; softmax is online (running max /
// sum in registers); the backward recomputes the probabilities from the saved log-sum-exp in two passes (queries, then keys),
// so no N x N matrix is ever stored.  HBM traffic = q, kv in, o out (and their gradients): the kernel is LDS/VALU work on tiny tiles.
#include "common.h"

namespace {

struct Win3dK {
  const char* q;    // (B, D, H, W, C)
  const char* kv;   // (B, D, H, W, 2C): k = [0, C), v = [C, 2C)
  const float* bq;  // (C) or null
  const float* bkv; // (2C) or null
  const float* table;  // (n_rel, heads)
  char* o;          // fwd: out (B, D, H, W, C)
  float* lse;       // (windows, heads, N)
  // backward
  const char* d_o;  // (B, D, H, W, C)
  const char* o_in; // saved output
  char* dq;         // (B, D, H, W, C)
  char* dkv;        // (B, D, H, W, 2C)
  float* dtable;    // (n_rel, heads), accumulated
  float* dbq;       // (C), accumulated: gradient reaching the bias through padded positions
  float* dbkv;      // (2C)
  int B, D, H, W, C, heads, d;
  int Dp, Hp, Wp, wt;
  int sd, sh, sw;   // shift (0 on unshifted blocks)
  int nwd, nwh, nww;  // windows per dimension
  float scale;
};

template <typename T>
__device__ __forceinline__ float ldf(const char* p, long long i) { return to_f32(reinterpret_cast<const T*>(p)[i]); }
template <typename T>
__device__ __forceinline__ void stf(char* p, long long i, float v) { reinterpret_cast<T*>(p)[i] = from_f32<T>(v); }

struct Tok {
  long long pix;  // pixel index in (B, D, H, W), -1 for a padded position
  int reg;        // region id of the shift mask
  int wd, wh, ww;
};

__device__ __forceinline__ Tok locate(const Win3dK& a, int win, int i) {
  Tok t;
  t.wd = i >> 6; t.wh = (i >> 3) & 7; t.ww = i & 7;
  const int bw = win % a.nww;
  int r = win / a.nww;
  const int bh = r % a.nwh;
  r /= a.nwh;
  const int bd = r % a.nwd;
  const int b = r / a.nwd;
  const int s0 = bd * a.wt + t.wd, s1 = bh * 8 + t.wh, s2 = bw * 8 + t.ww;  // coordinates in the rolled, padded volume
  // compute_mask (swin_3d.py:104-118): three slabs per dimension, [0, P-w), [P-w, P-shift), [P-shift, P)
  // (a dimension without shift is one region: the reference's last slice, [-0:], then covers everything)
  const int r0 = a.sd == 0 ? 2 : (s0 < a.Dp - a.wt ? 0 : (s0 < a.Dp - a.sd ? 1 : 2));
  const int r1 = a.sh == 0 ? 2 : (s1 < a.Hp - 8 ? 0 : (s1 < a.Hp - a.sh ? 1 : 2));
  const int r2 = a.sw == 0 ? 2 : (s2 < a.Wp - 8 ? 0 : (s2 < a.Wp - a.sw ? 1 : 2));
  t.reg = (r0 * 3 + r1) * 3 + r2;
  int p0 = s0 + a.sd, p1 = s1 + a.sh, p2 = s2 + a.sw;  // torch.roll(x, -shift): rolled[s] = padded[(s + shift) mod P]
  if (p0 >= a.Dp) p0 -= a.Dp;
  if (p1 >= a.Hp) p1 -= a.Hp;
  if (p2 >= a.Wp) p2 -= a.Wp;
  t.pix = (p0 < a.D && p1 < a.H && p2 < a.W) ? (((long long)b * a.D + p0) * a.H + p1) * a.W + p2 : -1;
  return t;
}

__device__ __forceinline__ int rel_index(int wt, int qd, int qh, int qw, int kd, int kh, int kw) {
  return ((qd - kd + wt - 1) * 15 + (qh - kh + 7)) * 15 + (qw - kw + 7);
}

// LDS layout (floats): A [N][DP], Bm [N][DP], tab [nrel], (bwd: dtab [nrel], lse [N], delta [N]), reg [N] ints
template <typename T, int NV, bool BWD>
__global__ __launch_bounds__(512) void win3d_kernel(const Win3dK a) {
  constexpr int DP = NV * 4;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = a.wt * 64, nrel = (2 * a.wt - 1) * 225;
  float* bufA = sm;
  float* bufB = bufA + N * DP;
  float* tab = bufB + N * DP;
  float* dtab = tab + nrel;
  float* slse = dtab + (BWD ? nrel : 0);
  float* sdel = slse + (BWD ? N : 0);
  int* sreg = reinterpret_cast<int*>(sdel + (BWD ? N : 0));
  const int i = threadIdx.x, win = blockIdx.x, head = blockIdx.y;
  const int d = a.d, c0 = head * d;
  const bool masked = (a.sd | a.sh | a.sw) != 0;
  const Tok me = locate(a, win, i);
  for (int r = i; r < nrel; r += N) {
    tab[r] = a.table[(long long)r * a.heads + head];
    if (BWD) dtab[r] = 0.f;
  }
  sreg[i] = me.reg;
  // this token's q (scaled), k, v
  float q[DP], kk[DP], vv[DP];
#pragma unroll
  for (int e = 0; e < DP; ++e) {
    float qe = 0.f, ke = 0.f, ve = 0.f;
    if (e < d) {
      if (me.pix >= 0) {
        qe = ldf<T>(a.q, me.pix * a.C + c0 + e);
        ke = ldf<T>(a.kv, me.pix * 2 * a.C + c0 + e);
        ve = ldf<T>(a.kv, me.pix * 2 * a.C + a.C + c0 + e);
      } else {
        qe = a.bq ? a.bq[c0 + e] : 0.f;
        ke = a.bkv ? a.bkv[c0 + e] : 0.f;
        ve = a.bkv ? a.bkv[a.C + c0 + e] : 0.f;
      }
    }
    q[e] = qe * a.scale; kk[e] = ke; vv[e] = ve;
  }
#pragma unroll
  for (int v4 = 0; v4 < NV; ++v4) {
    *reinterpret_cast<float4*>(bufA + i * DP + 4 * v4) = make_float4(kk[4 * v4], kk[4 * v4 + 1], kk[4 * v4 + 2], kk[4 * v4 + 3]);
    *reinterpret_cast<float4*>(bufB + i * DP + 4 * v4) = make_float4(vv[4 * v4], vv[4 * v4 + 1], vv[4 * v4 + 2], vv[4 * v4 + 3]);
  }
  __syncthreads();

  // logit of (query me, key j) given the key's K row in LDS
  auto logit = [&](const float* qv, const float* krow, int qd, int qh, int qw, int j, int qreg) {
    float s = 0.f;
#pragma unroll
    for (int v4 = 0; v4 < NV; ++v4) {
      const float4 k4 = *reinterpret_cast<const float4*>(krow + 4 * v4);
      s += qv[4 * v4] * k4.x + qv[4 * v4 + 1] * k4.y + qv[4 * v4 + 2] * k4.z + qv[4 * v4 + 3] * k4.w;
    }
    s += tab[rel_index(a.wt, qd, qh, qw, j >> 6, (j >> 3) & 7, j & 7)];
    if (masked && sreg[j] != qreg) s -= 100.f;
    return s;
  };

  if (!BWD) {
    float m = -INFINITY, l = 0.f, acc[DP];
#pragma unroll
    for (int e = 0; e < DP; ++e) acc[e] = 0.f;
    for (int j = 0; j < N; ++j) {
      if ((j >> 6) == me.wd) { j += 63; continue; }  // the query's own time slice is not a key
      const float s = logit(q, bufA + j * DP, me.wd, me.wh, me.ww, j, me.reg);
      const float mn = fmaxf(m, s);
      const float corr = __expf(m - mn), p = __expf(s - mn);
      l = l * corr + p;
#pragma unroll
      for (int v4 = 0; v4 < NV; ++v4) {
        const float4 v = *reinterpret_cast<const float4*>(bufB + j * DP + 4 * v4);
        acc[4 * v4] = acc[4 * v4] * corr + p * v.x;
        acc[4 * v4 + 1] = acc[4 * v4 + 1] * corr + p * v.y;
        acc[4 * v4 + 2] = acc[4 * v4 + 2] * corr + p * v.z;
        acc[4 * v4 + 3] = acc[4 * v4 + 3] * corr + p * v.w;
      }
      m = mn;
    }
    const float inv = 1.f / l;
    a.lse[((long long)win * a.heads + head) * N + i] = m + __logf(l);
    if (me.pix >= 0) {
#pragma unroll
      for (int e = 0; e < DP; ++e)
        if (e < d) stf<T>(a.o, me.pix * a.C + c0 + e, acc[e] * inv);
    }
    return;
  }

  // ------------------------------------------------------------------------------------------------ backward
  float go[DP], dqa[DP];
  float delta = 0.f;
#pragma unroll
  for (int e = 0; e < DP; ++e) {
    float g = 0.f, ov = 0.f;
    if (e < d && me.pix >= 0) {  // a padded query's output is dropped: no gradient enters there
      g = ldf<T>(a.d_o, me.pix * a.C + c0 + e);
      ov = ldf<T>(a.o_in, me.pix * a.C + c0 + e);
    }
    go[e] = g;
    delta += g * ov;
    dqa[e] = 0.f;
  }
  const float my_lse = a.lse[((long long)win * a.heads + head) * N + i];
  // pass 1, thread = query: dq, the table gradient
  for (int j = 0; j < N; ++j) {
    if ((j >> 6) == me.wd) { j += 63; continue; }
    const float s = logit(q, bufA + j * DP, me.wd, me.wh, me.ww, j, me.reg);
    const float p = __expf(s - my_lse);
    float dp = 0.f;
#pragma unroll
    for (int v4 = 0; v4 < NV; ++v4) {
      const float4 v = *reinterpret_cast<const float4*>(bufB + j * DP + 4 * v4);
      dp += go[4 * v4] * v.x + go[4 * v4 + 1] * v.y + go[4 * v4 + 2] * v.z + go[4 * v4 + 3] * v.w;
    }
    const float ds = p * (dp - delta);
#pragma unroll
    for (int v4 = 0; v4 < NV; ++v4) {
      const float4 k4 = *reinterpret_cast<const float4*>(bufA + j * DP + 4 * v4);
      dqa[4 * v4] += ds * k4.x; dqa[4 * v4 + 1] += ds * k4.y; dqa[4 * v4 + 2] += ds * k4.z; dqa[4 * v4 + 3] += ds * k4.w;
    }
    atomicAdd(&dtab[rel_index(a.wt, me.wd, me.wh, me.ww, j >> 6, (j >> 3) & 7, j & 7)], ds);
  }
  // dq = scale * sum_j ds * k_j  (q entered the logits scaled)
#pragma unroll
  for (int e = 0; e < DP; ++e) {
    if (e < d) {
      const float g = dqa[e] * a.scale;
      if (me.pix >= 0) stf<T>(a.dq, me.pix * a.C + c0 + e, g);
      else if (a.dbq) atomicAdd(&a.dbq[c0 + e], g);
    }
  }
  __syncthreads();  // everyone is done with K / V in LDS
  // pass 2, thread = key: stage the scaled queries and the output gradients, then dk, dv
#pragma unroll
  for (int v4 = 0; v4 < NV; ++v4) {
    *reinterpret_cast<float4*>(bufA + i * DP + 4 * v4) = make_float4(q[4 * v4], q[4 * v4 + 1], q[4 * v4 + 2], q[4 * v4 + 3]);
    *reinterpret_cast<float4*>(bufB + i * DP + 4 * v4) = make_float4(go[4 * v4], go[4 * v4 + 1], go[4 * v4 + 2], go[4 * v4 + 3]);
  }
  slse[i] = my_lse;
  sdel[i] = delta;
  __syncthreads();
  float dka[DP], dva[DP];
#pragma unroll
  for (int e = 0; e < DP; ++e) dka[e] = dva[e] = 0.f;
  for (int qi = 0; qi < N; ++qi) {
    if ((qi >> 6) == me.wd) { qi += 63; continue; }
    // logit(query qi, key me): the K row is this thread's own k
    float s = 0.f;
#pragma unroll
    for (int v4 = 0; v4 < NV; ++v4) {
      const float4 q4 = *reinterpret_cast<const float4*>(bufA + qi * DP + 4 * v4);
      s += q4.x * kk[4 * v4] + q4.y * kk[4 * v4 + 1] + q4.z * kk[4 * v4 + 2] + q4.w * kk[4 * v4 + 3];
    }
    s += tab[rel_index(a.wt, qi >> 6, (qi >> 3) & 7, qi & 7, me.wd, me.wh, me.ww)];
    if (masked && sreg[qi] != me.reg) s -= 100.f;
    const float p = __expf(s - slse[qi]);
    float dp = 0.f;
#pragma unroll
    for (int v4 = 0; v4 < NV; ++v4) {
      const float4 g4 = *reinterpret_cast<const float4*>(bufB + qi * DP + 4 * v4);
      dp += g4.x * vv[4 * v4] + g4.y * vv[4 * v4 + 1] + g4.z * vv[4 * v4 + 2] + g4.w * vv[4 * v4 + 3];
      dva[4 * v4] += p * g4.x; dva[4 * v4 + 1] += p * g4.y; dva[4 * v4 + 2] += p * g4.z; dva[4 * v4 + 3] += p * g4.w;
    }
    const float ds = p * (dp - sdel[qi]);
#pragma unroll
    for (int v4 = 0; v4 < NV; ++v4) {
      const float4 q4 = *reinterpret_cast<const float4*>(bufA + qi * DP + 4 * v4);
      dka[4 * v4] += ds * q4.x; dka[4 * v4 + 1] += ds * q4.y; dka[4 * v4 + 2] += ds * q4.z; dka[4 * v4 + 3] += ds * q4.w;
    }
  }
#pragma unroll
  for (int e = 0; e < DP; ++e) {
    if (e < d) {
      if (me.pix >= 0) {
        stf<T>(a.dkv, me.pix * 2 * a.C + c0 + e, dka[e]);
        stf<T>(a.dkv, me.pix * 2 * a.C + a.C + c0 + e, dva[e]);
      } else if (a.dbkv) {
        atomicAdd(&a.dbkv[c0 + e], dka[e]);
        atomicAdd(&a.dbkv[a.C + c0 + e], dva[e]);
      }
    }
  }
  __syncthreads();
  for (int r = i; r < nrel; r += N)
    if (dtab[r] != 0.f) atomicAdd(&a.dtable[(long long)r * a.heads + head], dtab[r]);
}

template <typename T, bool BWD>
int launch_win3d(const Win3dK& k, int nv, int lds, hipStream_t st) {
  const dim3 grid((unsigned)((long long)k.B * k.nwd * k.nwh * k.nww), k.heads), block(k.wt * 64);
#define W3_CASE(NVV)                                                                                           \
  case NVV: {                                                                                                  \
    auto fn = win3d_kernel<T, NVV, BWD>;                                                                       \
    static bool attr_set[VMG_MAX_DEVICES] = {};                                                                \
    const int dev = vmg_current_device();                                                                      \
    if (!attr_set[dev]) {                                                                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr_set[dev] = true;                                                                                    \
    }                                                                                                          \
    hipLaunchKernelGGL(fn, grid, block, lds, st, k);                                                           \
    break;                                                                                                     \
  }
  switch (nv) {
    W3_CASE(1) W3_CASE(2) W3_CASE(3) W3_CASE(4) W3_CASE(5) W3_CASE(6) W3_CASE(7) W3_CASE(8) W3_CASE(9) W3_CASE(10) W3_CASE(12) W3_CASE(14) W3_CASE(16)
    default:
      vmg_set_error("win3d_attn: head dimension %d is not instantiated (multiples of 4 up to 40, 48, 56, 64 after padding)", k.d);
      return -1;
  }
#undef W3_CASE
  VMG_LAUNCH_CHECK();
  return 0;
}

int win3d_prepare(Win3dK& k, int dtype, int B, int D, int H, int W, int C, int heads, int wt, int sd, int sh, int sw, bool bwd, int* nv, int* lds) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "win3d_attn: bad dtype");
  VMG_CHECK(B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && heads > 0 && C % heads == 0, "win3d_attn: bad shape");
  VMG_CHECK(wt >= 1 && wt <= 8, "win3d_attn: temporal window must be 1..8 (spatial window is 8 x 8)");
  k.B = B; k.D = D; k.H = H; k.W = W; k.C = C; k.heads = heads; k.d = C / heads; k.wt = wt;
  k.Dp = cdiv(D, wt) * wt; k.Hp = cdiv(H, 8) * 8; k.Wp = cdiv(W, 8) * 8;
  k.nwd = k.Dp / wt; k.nwh = k.Hp / 8; k.nww = k.Wp / 8;
  VMG_CHECK(sd >= 0 && sd < wt && sh >= 0 && sh < 8 && sw >= 0 && sw < 8, "win3d_attn: shifts must lie inside the window");
  k.sd = sd; k.sh = sh; k.sw = sw;
  k.scale = 1.0f / sqrtf((float)k.d);
  int n = (k.d + 3) / 4;
  if (n == 11) n = 12;
  if (n == 13) n = 14;
  if (n == 15) n = 16;
  VMG_CHECK(n <= 16, "win3d_attn: head dimension %d too large (<= 64)", k.d);
  *nv = n;
  const int N = wt * 64, nrel = (2 * wt - 1) * 225;
  *lds = (2 * N * n * 4 + nrel * (bwd ? 2 : 1) + (bwd ? 2 * N : 0) + N) * 4;
  VMG_CHECK(*lds <= 160 * 1024, "win3d_attn: window of %d tokens x head dimension %d needs %d B of LDS (> 160 KiB)", N, k.d, *lds);
  return 0;
}

}  // namespace

extern "C" int vmg_win3d_attn_fwd(int dtype, const void* q, const void* kv, const float* bq, const float* bkv, const float* table, void* out,
                                  float* lse, int B, int D, int H, int W, int C, int heads, int wt, int sd, int sh, int sw, void* stream) {
  VMG_CHECK(q && kv && table && out && lse, "win3d_attn_fwd: null pointer");
  Win3dK k;
  memset(&k, 0, sizeof(k));
  int nv, lds;
  if (win3d_prepare(k, dtype, B, D, H, W, C, heads, wt, sd, sh, sw, false, &nv, &lds)) return -1;
  k.q = (const char*)q; k.kv = (const char*)kv; k.bq = bq; k.bkv = bkv; k.table = table; k.o = (char*)out; k.lse = lse;
  return dtype == VMG_BF16 ? launch_win3d<bf16, false>(k, nv, lds, (hipStream_t)stream) : launch_win3d<float, false>(k, nv, lds, (hipStream_t)stream);
}

extern "C" int vmg_win3d_attn_bwd(int dtype, const void* q, const void* kv, const float* bq, const float* bkv, const float* table, const void* out,
                                  const float* lse, const void* d_out, void* dq, void* dkv, float* dtable, float* dbq, float* dbkv, int B, int D,
                                  int H, int W, int C, int heads, int wt, int sd, int sh, int sw, void* stream) {
  VMG_CHECK(q && kv && table && out && lse && d_out && dq && dkv && dtable, "win3d_attn_bwd: null pointer");
  Win3dK k;
  memset(&k, 0, sizeof(k));
  int nv, lds;
  if (win3d_prepare(k, dtype, B, D, H, W, C, heads, wt, sd, sh, sw, true, &nv, &lds)) return -1;
  k.q = (const char*)q; k.kv = (const char*)kv; k.bq = bq; k.bkv = bkv; k.table = table; k.o_in = (const char*)out; k.lse = const_cast<float*>(lse);
  k.d_o = (const char*)d_out; k.dq = (char*)dq; k.dkv = (char*)dkv; k.dtable = dtable; k.dbq = dbq; k.dbkv = dbkv;
  return dtype == VMG_BF16 ? launch_win3d<bf16, true>(k, nv, lds, (hipStream_t)stream) : launch_win3d<float, true>(k, nv, lds, (hipStream_t)stream);
}
