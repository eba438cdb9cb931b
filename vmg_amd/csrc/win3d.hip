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
  float* dtab_ws;   // null, or (windows, heads, n_rel): every workgroup's table gradient, summed in window order by win3d_dtable_reduce_kernel
  float* dbq;       // (C), accumulated: gradient reaching the bias through padded positions
  float* dbkv;      // (2C)
  int B, D, H, W, C, heads, d;
  int Dp, Hp, Wp, wt;
  int sd, sh, sw;   // shift (0 on unshifted blocks)
  int nwd, nwh, nww;  // windows per dimension
  float scale;
  int dm_lds;  // MFMA backward: the per-slice dS block is kept in LDS and the table gradient gathered from it (fits for wt <= 4)
  int dbg;  // diagnostics build only (env VMG_WIN3D_DBG): 1 no table-gradient LDS atomics, 2 no pass 2 (dK / dV), 4 no pass 1 (dQ, table)
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
{
    // (1 575 float atomics per workgroup onto the same 12 600 addresses from every window made the launch atomic-bound: 413 us at 1 024
    //  workgroups; with the workspace the partial goes out as plain coalesced stores and a second launch adds the windows in a fixed order)
    if (a.dtab_ws) a.dtab_ws[((long long)win * a.heads + head) * nrel + r] = dtab[r];
    else if (dtab[r] != 0.f) atomicAdd(&a.dtable[(long long)r * a.heads + head], dtab[r]);
  }
}

// =====================================================================================================================================
// MFMA form (bf16, head dimension even and <= 32; round 4 -- north_star asks for the QK^T / PV contractions on the matrix cores).
// One workgroup = one (window, head), four waves; per time slice, wave w owns the 16 queries 16w .. 16w+15 of the slice.  K and V rows of the
// window (32 dims, zero-padded; 72-byte rows) sit in LDS.  All products are taken TRANSPOSED so that no accumulator ever has to be
// re-laid out between two MFMAs:
//   S^T tile (16 keys x 16 queries) = K rows (A operand: lane = key row, 8 dims) x Q^T (B operand: lane = query column, 8 dims, from registers);
//     a lane then holds 4 keys (rows 4g .. 4g+3) of ONE query column: softmax statistics are a register loop + two shuffles across the lane groups;
//   O^T (16 dims x 16 queries) += V^T (A) x P^T (B): the B operand of lane group g wants 8 consecutive "k" of the query's column -- it takes the
//     4 + 4 probabilities the lane already holds from a PAIR of key tiles (the order of a sum is free), and the A operand reads V in that same key
//     order TRANSPOSED out of the row-major LDS rows (ds_read_b64_tr_b16: lane li of a 16-lane group fetches row li >> 2, dims 4 (li & 3) .. + 3 of a
//     4-key x 16-dim block and receives dim li of the 4 keys).
// The bias is tab[R(query) - K(key)] with both parts linear in the window coordinates (one add per element); the -100 mask compares region ids.
// Forward: one sweep over pairs of key tiles with an online softmax.
// Backward: pass 1, wave = 16 queries: S^T, dP^T = V rows x dO^T, dS^T = P (dP - delta), table gradient (LDS atomics), dQ^T += K^T x dS^T;
// pass 2, wave = 16 keys, everything with queries and keys swapped (Q and dO rows in LDS): dV^T += dO^T x P, dK^T += Q^T x dS.  fp32 inputs and
// other head dimensions keep the VALU kernel above.
constexpr int WM_ROW = 72;  // bytes of a token row in LDS: 32 bf16 dims + 8 B (18 dwords: the 16 rows of a tile start in 16 different banks)

__device__ __forceinline__ bf16x8 wm_tr_pair(const char* lo_p, const char* hi_p) {
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lo_p));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(hi_p));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// 8 bf16 of an LDS row (rows are 72 bytes: 8-byte aligned): two 8-byte accesses
__device__ __forceinline__ bf16x8 wm_ld8(const char* p) {
  const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p), hi = *reinterpret_cast<const bf16x4*>(p + 8);
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ void wm_st8(char* p, bf16x8 v) {
  *reinterpret_cast<bf16x4*>(p) = bf16x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<bf16x4*>(p + 8) = bf16x4{v[4], v[5], v[6], v[7]};
}

// dims 8g .. 8g+7 of one token's vector (element offset `off` in rows of `rowlen`): zero beyond d; a padded token holds the bias (or zero)
__device__ __forceinline__ bf16x8 wm_frag(const bf16* base, long long pix, int rowlen, int off, const float* bias, int boff, int g, int d, bool zero_pad) {
  bf16x8 f;
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const int dim = 8 * g + e;
    float v0 = 0.f, v1 = 0.f;
    if (dim < d) {  // (d is even: a pair is inside or outside together)
      if (pix >= 0) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
        const bf16x2_t pr = *reinterpret_cast<const bf16x2_t*>(base + pix * rowlen + off + dim);
        v0 = (float)pr[0]; v1 = (float)pr[1];
      } else if (bias && !zero_pad) {
        v0 = bias[boff + dim]; v1 = bias[boff + dim + 1];
      }
    }
    f[e] = (bf16)v0; f[e + 1] = (bf16)v1;
  }
  return f;
}

// one token's 32-dim row -> LDS (four 16-byte vectors)
__device__ __forceinline__ void wm_store_row(char* dst, const bf16* base, long long pix, int rowlen, int off, const float* bias, int boff, int d, bool zero_pad) {
#pragma unroll
  for (int g = 0; g < 4; ++g) wm_st8(dst + 16 * g, wm_frag(base, pix, rowlen, off, bias, boff, g, d, zero_pad));
}

__device__ __forceinline__ int wm_rel_q(int wt, int tok) { return ((tok >> 6) + wt - 1) * 225 + (((tok >> 3) & 7) + 7) * 15 + ((tok & 7) + 7); }
__device__ __forceinline__ int wm_rel_k(int tok) { return (tok >> 6) * 225 + ((tok >> 3) & 7) * 15 + (tok & 7); }

__global__ __launch_bounds__(256) void win3d_mfma_fwd_kernel(const Win3dK a) {
  extern __shared__ __attribute__((aligned(16))) char smc[];
  const int N = a.wt * 64, nrel = (2 * a.wt - 1) * 225;
  char* Ks = smc;
  char* Vs = Ks + N * WM_ROW;
  float* tab = reinterpret_cast<float*>(Vs + N * WM_ROW);
  int* sreg = reinterpret_cast<int*>(tab + nrel);
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int win = blockIdx.x, head = blockIdx.y, d = a.d, c0 = head * d;
  const bool masked = (a.sd | a.sh | a.sw) != 0;
  const bf16* qg = reinterpret_cast<const bf16*>(a.q);
  const bf16* kvg = reinterpret_cast<const bf16*>(a.kv);
  for (int r = tid; r < nrel; r += 256) tab[r] = a.table[(long long)r * a.heads + head];
  for (int i = tid; i < N; i += 256) {
    const Tok t = locate(a, win, i);
    sreg[i] = t.reg;
    wm_store_row(Ks + i * WM_ROW, kvg, t.pix, 2 * a.C, c0, a.bkv, c0, d, false);
    wm_store_row(Vs + i * WM_ROW, kvg, t.pix, 2 * a.C, a.C + c0, a.bkv, a.C + c0, d, false);
  }
  __syncthreads();
  const int NT = 4 * (a.wt - 1), DU = (d + 15) >> 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < a.wt; ++s) {
    const int qt = 64 * s + 16 * wave + c;
    const Tok me = locate(a, win, qt);
    const bf16x8 qf = wm_frag(qg, me.pix, a.C, c0, a.bq, c0, g, d, false);
    const int Rq = wm_rel_q(a.wt, qt);
    auto logits = [&](int tt, float (&x)[4]) __attribute__((always_inline)) {
      const bf16x8 kf = wm_ld8(Ks + (16 * tt + c) * WM_ROW + 16 * g);
      const f32x4 sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, zero4, 0, 0, 0);
      const int kt0 = 16 * tt + 4 * g;
      const int rk = Rq - wm_rel_k(kt0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = sv[r] * a.scale + tab[rk - r];
        if (masked && sreg[kt0 + r] != me.reg) v -= 100.f;
        x[r] = v;
      }
    };
    // online softmax over pairs of key tiles: the running maximum is made common to the four lane groups of a query column (they hold different
    // keys of it, and the PV product sums over all of them), the accumulators are rescaled when it moves
    float m = -INFINITY, l = 0.f;
    f32x4 oacc[2] = {zero4, zero4};
    for (int tp = 0; tp < NT; tp += 2) {
      const int tt0 = tp + (tp >= 4 * s ? 4 : 0), tt1 = tp + 1 + (tp + 1 >= 4 * s ? 4 : 0);
      float x0[4], x1[4];
      logits(tt0, x0);
      logits(tt1, x1);
      float pm = fmaxf(fmaxf(fmaxf(x0[0], x0[1]), fmaxf(x0[2], x0[3])), fmaxf(fmaxf(x1[0], x1[1]), fmaxf(x1[2], x1[3])));
      pm = fmaxf(pm, __shfl_xor(pm, 16, 64));
      pm = fmaxf(pm, __shfl_xor(pm, 32, 64));
      const float mn = fmaxf(m, pm), corr = __expf(m - mn);
      m = mn;
      l *= corr;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) oacc[u][r] *= corr;
      bf16x8 pf;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p0 = __expf(x0[r] - m), p1 = __expf(x1[r] - m);
        l += p0 + p1;
        pf[r] = (bf16)p0; pf[4 + r] = (bf16)p1;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u < DU) {
          const bf16x8 vf = wm_tr_pair(Vs + (16 * tt0 + 4 * g + (c >> 2)) * WM_ROW + (16 * u + 4 * (c & 3)) * 2,
                                       Vs + (16 * tt1 + 4 * g + (c >> 2)) * WM_ROW + (16 * u + 4 * (c & 3)) * 2);
          oacc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[u], 0, 0, 0);
        }
      }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    if (g == 0) a.lse[((long long)win * a.heads + head) * N + qt] = m + __logf(l);
    if (me.pix >= 0) {
      bf16* og = reinterpret_cast<bf16*>(a.o) + me.pix * a.C + c0;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const int dim = 16 * u + 4 * g + r;
          if (u < DU && dim < d) {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
            const bf16x2_t pr = {(bf16)(oacc[u][r] * inv), (bf16)(oacc[u][r + 1] * inv)};
            *reinterpret_cast<bf16x2_t*>(og + dim) = pr;
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void win3d_mfma_bwd_kernel(const Win3dK a) {
  extern __shared__ __attribute__((aligned(16))) char smc[];
  const int N = a.wt * 64, nrel = (2 * a.wt - 1) * 225;
  char* Ra = smc;                 // pass 1: K rows, pass 2: Q rows
  char* Rb = Ra + N * WM_ROW;     // pass 1: V rows, pass 2: dO rows
  float* tab = reinterpret_cast<float*>(Rb + N * WM_ROW);
  float* dtab = tab + nrel;
  float* slse = dtab + nrel;
  float* sdel = slse + N;
  int* sreg = reinterpret_cast<int*>(sdel + N);
  int* spix = sreg + N;  // pixel of every token (-1: padded); B * D * H * W < 2^31 (checked by the launcher)
  // The table gradient dtab[rel] = sum of dS over the (query, key) pairs at that relative position.  One LDS float atomic per element (what the
  // VALU kernel does) was 225 of this kernel's 430 us at the train_swin shape -- ds_add_f32 with 64 distinct addresses retires in ~300 cycles.
  // Instead the four waves store a slice's dS^T (64 queries x N - 64 keys, the bf16 values dQ is computed from) to LDS and every THREAD then owns table entries: it walks the
  // pairs of its entry (a 2-D diagonal of the block: (8 - |dh|) (8 - |dw|) of them) with plain reads.  Used when the block fits (wt <= 6).
  const int dm_row = (N - 64) * 2 + 16;
  char* Dm = a.dm_lds ? reinterpret_cast<char*>(spix + N) : nullptr;
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int win = blockIdx.x, head = blockIdx.y, d = a.d, c0 = head * d;
  const bool masked = (a.sd | a.sh | a.sw) != 0;
  const bf16* qg = reinterpret_cast<const bf16*>(a.q);
  const bf16* kvg = reinterpret_cast<const bf16*>(a.kv);
  const bf16* dog = reinterpret_cast<const bf16*>(a.d_o);
  const bf16* og = reinterpret_cast<const bf16*>(a.o_in);
  for (int r = tid; r < nrel; r += 256) {
    tab[r] = a.table[(long long)r * a.heads + head];
    dtab[r] = 0.f;
  }
  for (int i = tid; i < N; i += 256) {
    const Tok t = locate(a, win, i);
    sreg[i] = t.reg;
    spix[i] = (int)t.pix;
    wm_store_row(Ra + i * WM_ROW, kvg, t.pix, 2 * a.C, c0, a.bkv, c0, d, false);
    wm_store_row(Rb + i * WM_ROW, kvg, t.pix, 2 * a.C, a.C + c0, a.bkv, a.C + c0, d, false);
  }
  __syncthreads();
  const int NT = 4 * (a.wt - 1), DU = (d + 15) >> 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  // ---------------------------------------------------------------------------------------------------------------- pass 1: wave = 16 queries
#ifdef VMG_DIAG
  const int wt1 = (a.dbg & 4) ? 0 : a.wt, nkb = (a.dbg & 2) ? 0 : N / 16;
#else
  const int wt1 = a.wt, nkb = N / 16;
#endif
  for (int s = 0; s < wt1; ++s) {
    const int qt = 64 * s + 16 * wave + c;
    const long long qpix = spix[qt];  // (int -> long long: -1 stays -1)
    const int qreg = sreg[qt];
    const bf16x8 qf = wm_frag(qg, qpix, a.C, c0, a.bq, c0, g, d, false);
    const bf16x8 gof = wm_frag(dog, qpix, a.C, c0, nullptr, 0, g, d, true);  // a padded query's output is dropped: no gradient enters there
    const bf16x8 of = wm_frag(og, qpix, a.C, c0, nullptr, 0, g, d, true);
    float delta = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) delta += (float)gof[e] * (float)of[e];
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    const float my_lse = a.lse[((long long)win * a.heads + head) * N + qt];
    if (g == 0) { slse[qt] = my_lse; sdel[qt] = delta; }
    const int Rq = wm_rel_q(a.wt, qt);
    f32x4 dqa[2] = {zero4, zero4};
    for (int tp = 0; tp < NT; tp += 2) {
      bf16x8 dsf;
      int tts[2];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int t = tp + h2, tt = t + (t >= 4 * s ? 4 : 0);
        tts[h2] = tt;
        const bf16x8 kf = wm_ld8(Ra + (16 * tt + c) * WM_ROW + 16 * g);
        const bf16x8 vf = wm_ld8(Rb + (16 * tt + c) * WM_ROW + 16 * g);
        const f32x4 sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, zero4, 0, 0, 0);
        const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, gof, zero4, 0, 0, 0);
        const int kt0 = 16 * tt + 4 * g;
        const int rk = Rq - wm_rel_k(kt0);
        f32x4 dsv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = sv[r] * a.scale + tab[rk - r];
          if (masked && sreg[kt0 + r] != qreg) v -= 100.f;
          const float p = __expf(v - my_lse);
          const float ds = p * (dp[r] - delta);
          dsv[r] = ds;
#ifdef VMG_DIAG
          if (!(a.dbg & 1))
#endif
          if (!Dm) atomicAdd(&dtab[rk - r], ds);
          dsf[4 * h2 + r] = (bf16)ds;
        }
        if (Dm) *reinterpret_cast<bf16x4*>(Dm + (16 * wave + c) * dm_row + (16 * t + 4 * g) * 2) = bf16x4{(bf16)dsv[0], (bf16)dsv[1], (bf16)dsv[2], (bf16)dsv[3]};
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u < DU) {
          const bf16x8 ktf = wm_tr_pair(Ra + (16 * tts[0] + 4 * g + (c >> 2)) * WM_ROW + (16 * u + 4 * (c & 3)) * 2,
                                        Ra + (16 * tts[1] + 4 * g + (c >> 2)) * WM_ROW + (16 * u + 4 * (c & 3)) * 2);
          dqa[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf, dqa[u], 0, 0, 0);
        }
      }
    }
    // dq = scale * sum_k ds * k (the logits took q scaled); a padded query's gradient goes to the q bias
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const int dim = 16 * u + 4 * g + r;
        if (u < DU && dim < d) {
          const float g0 = dqa[u][r] * a.scale, g1 = dqa[u][r + 1] * a.scale;
          if (qpix >= 0) {
            const bf16x2_t pr = {(bf16)g0, (bf16)g1};
            *reinterpret_cast<bf16x2_t*>(reinterpret_cast<bf16*>(a.dq) + qpix * a.C + c0 + dim) = pr;
          } else if (a.dbq) {
            atomicAdd(&a.dbq[c0 + dim], g0);
            atomicAdd(&a.dbq[c0 + dim + 1], g1);
          }
        }
      }
    }
    if (Dm) {
      __syncthreads();  // the slice's dS^T block is complete
      // entry (i, hh, ww): key slice kd = the i-th slice other than s; relative position dh = hh - 7 = qh - kh, dw = ww - 7 = qw - kw
      for (int e = tid; e < (a.wt - 1) * 225; e += 256) {
        const int i = e / 225, rem = e - i * 225, hh = rem / 15, ww = rem - hh * 15;
        const int kd = i + (i >= s ? 1 : 0), dh = hh - 7, dw = ww - 7;
        const int qh_lo = dh > 0 ? dh : 0, qh_hi = dh < 0 ? 8 + dh : 8, qw_lo = dw > 0 ? dw : 0, qw_hi = dw < 0 ? 8 + dw : 8;
        float acc = 0.f;
        for (int qh = qh_lo; qh < qh_hi; ++qh) {
          const char* row = Dm + (qh * 8) * dm_row + (i * 64 + (qh - dh) * 8 - dw) * 2;  // + qw * (dm_row + 2)
          for (int qw = qw_lo; qw < qw_hi; ++qw) acc += (float)*reinterpret_cast<const bf16*>(row + qw * (dm_row + 2));
        }
        dtab[(s - kd + a.wt - 1) * 225 + rem] += acc;  // (one thread per entry and slice; slices are separated by the barriers)
      }
      __syncthreads();  // before the next slice overwrites the block
    }
  }
  __syncthreads();  // every wave is done with the K / V rows; lse / delta of all queries are in LDS
  // ---------------------------------------------------------------------------------------------------------------- pass 2: wave = 16 keys
  for (int i = tid; i < N; i += 256) {
    const long long pix = spix[i];
    wm_store_row(Ra + i * WM_ROW, qg, pix, a.C, c0, a.bq, c0, d, false);
    wm_store_row(Rb + i * WM_ROW, dog, pix, a.C, c0, nullptr, 0, d, true);
  }
  __syncthreads();
  for (int kb = wave; kb < nkb; kb += 4) {
    const int kt = 16 * kb + c, sk = kb >> 2;
    const long long kpix = spix[kt];
    const int kreg = sreg[kt];
    const bf16x8 kf = wm_frag(kvg, kpix, 2 * a.C, c0, a.bkv, c0, g, d, false);
    const bf16x8 vf = wm_frag(kvg, kpix, 2 * a.C, a.C + c0, a.bkv, a.C + c0, g, d, false);
    const int Kk = wm_rel_k(kt);
    f32x4 dka[2] = {zero4, zero4}, dva[2] = {zero4, zero4};
    for (int tp = 0; tp < NT; tp += 2) {
      bf16x8 pf, dsf;
      int tts[2];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int t = tp + h2, tt = t + (t >= 4 * sk ? 4 : 0);
        tts[h2] = tt;
        const bf16x8 qrow = wm_ld8(Ra + (16 * tt + c) * WM_ROW + 16 * g);
        const bf16x8 grow = wm_ld8(Rb + (16 * tt + c) * WM_ROW + 16 * g);
        const f32x4 sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qrow, kf, zero4, 0, 0, 0);   // rows: queries 16tt + 4g + r, column: key c
        const f32x4 dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(grow, vf, zero4, 0, 0, 0);
        const int q0 = 16 * tt + 4 * g;
        const int rq = wm_rel_q(a.wt, q0) - Kk;  // + r along the row of the window (q0 & 7 is 0 or 4)
        const f32x4 ls = *reinterpret_cast<const f32x4*>(slse + q0), dl = *reinterpret_cast<const f32x4*>(sdel + q0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = sv[r] * a.scale + tab[rq + r];
          if (masked && sreg[q0 + r] != kreg) v -= 100.f;
          const float p = __expf(v - ls[r]);
          pf[4 * h2 + r] = (bf16)p;
          dsf[4 * h2 + r] = (bf16)(p * (dp[r] - dl[r]));
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u < DU) {
          const int off0 = (16 * tts[0] + 4 * g + (c >> 2)) * WM_ROW + (16 * u + 4 * (c & 3)) * 2;
          const int off1 = (16 * tts[1] + 4 * g + (c >> 2)) * WM_ROW + (16 * u + 4 * (c & 3)) * 2;
          const bf16x8 gtf = wm_tr_pair(Rb + off0, Rb + off1);  // dO^T: dims x queries
          const bf16x8 qtf = wm_tr_pair(Ra + off0, Ra + off1);  // Q^T
          dva[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gtf, pf, dva[u], 0, 0, 0);
          dka[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, dsf, dka[u], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const int dim = 16 * u + 4 * g + r;
        if (u < DU && dim < d) {
          const float k0 = dka[u][r] * a.scale, k1 = dka[u][r + 1] * a.scale, v0 = dva[u][r], v1 = dva[u][r + 1];
          if (kpix >= 0) {
            bf16* dkv = reinterpret_cast<bf16*>(a.dkv) + kpix * 2 * a.C + c0 + dim;
            const bf16x2_t pk = {(bf16)k0, (bf16)k1}, pv = {(bf16)v0, (bf16)v1};
            *reinterpret_cast<bf16x2_t*>(dkv) = pk;
            *reinterpret_cast<bf16x2_t*>(dkv + a.C) = pv;
          } else if (a.dbkv) {
            atomicAdd(&a.dbkv[c0 + dim], k0); atomicAdd(&a.dbkv[c0 + dim + 1], k1);
            atomicAdd(&a.dbkv[a.C + c0 + dim], v0); atomicAdd(&a.dbkv[a.C + c0 + dim + 1], v1);
          }
        }
      }
    }
  }
  __syncthreads();
  for (int r = tid; r < nrel; r += 256)
{
    // (1 575 float atomics per workgroup onto the same 12 600 addresses from every window made the launch atomic-bound: 413 us at 1 024
    //  workgroups; with the workspace the partial goes out as plain coalesced stores and a second launch adds the windows in a fixed order)
    if (a.dtab_ws) a.dtab_ws[((long long)win * a.heads + head) * nrel + r] = dtab[r];
    else if (dtab[r] != 0.f) atomicAdd(&a.dtable[(long long)r * a.heads + head], dtab[r]);
  }
}

__global__ __launch_bounds__(256) void win3d_dtable_reduce_kernel(const float* __restrict__ ws, int nwin, int heads, int nrel, float* __restrict__ dtable) {
  const int r = blockIdx.x * 256 + threadIdx.x, head = blockIdx.y;
  if (r >= nrel) return;
  float acc = 0.f;
  for (int w = 0; w < nwin; ++w) acc += ws[((long long)w * heads + head) * nrel + r];  // fixed order: bit-reproducible
  dtable[(long long)r * heads + head] += acc;
}

static int g_win3d_variant = 1;  // 1: the MFMA kernels where they apply (bf16, even head dimension <= 32), 0: always the VALU kernel

template <bool BWD>
int launch_win3d_mfma(const Win3dK& k0, hipStream_t st) {
  Win3dK k = k0;
  const int N = k.wt * 64, nrel = (2 * k.wt - 1) * 225;
  int lds = 2 * N * WM_ROW + nrel * 4 * (BWD ? 2 : 1) + (BWD ? 2 * N * 4 : 0) + N * 4 + (BWD ? N * 4 : 0) + 16;
  if (BWD) {
    VMG_CHECK((long long)k.B * k.D * k.H * k.W < (1LL << 31), "win3d_attn (MFMA): too many tokens");
    const int dm = 64 * ((N - 64) * 2 + 16);
    k.dm_lds = lds + dm <= 160 * 1024 ? 1 : 0;  // (wt = 4: 79 KB in all, two workgroups per CU; wt = 8 does not fit and falls back to LDS atomics)
    if (k.dm_lds) lds += dm;
  }
  VMG_CHECK(lds <= 160 * 1024, "win3d_attn (MFMA): %d B of LDS", lds);
  const dim3 grid((unsigned)((long long)k.B * k.nwd * k.nwh * k.nww), k.heads);
  if (BWD) {
    static bool attr_b[VMG_MAX_DEVICES] = {};
    const int dev = vmg_current_device();
    if (!attr_b[dev]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(win3d_mfma_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_b[dev] = true; }
    hipLaunchKernelGGL(win3d_mfma_bwd_kernel, grid, dim3(256), lds, st, k);
  } else {
    static bool attr_f[VMG_MAX_DEVICES] = {};
    const int dev = vmg_current_device();
    if (!attr_f[dev]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(win3d_mfma_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_f[dev] = true; }
    hipLaunchKernelGGL(win3d_mfma_fwd_kernel, grid, dim3(256), lds, st, k);
  }
  VMG_LAUNCH_CHECK();
  return 0;
}

static bool win3d_mfma_ok(int dtype, const Win3dK& k) {
  return g_win3d_variant == 1 && dtype == VMG_BF16 && k.d <= 32 && (k.d % 2) == 0 && (k.C % 2) == 0 && k.wt >= 2;
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
  if (win3d_mfma_ok(dtype, k)) return launch_win3d_mfma<false>(k, (hipStream_t)stream);
  return dtype == VMG_BF16 ? launch_win3d<bf16, false>(k, nv, lds, (hipStream_t)stream) : launch_win3d<float, false>(k, nv, lds, (hipStream_t)stream);
}

extern "C" int64_t vmg_win3d_attn_bwd_ws_bytes(int B, int D, int H, int W, int heads, int wt) {
  if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || heads <= 0 || wt < 1 || wt > 8) return -1;
  return (int64_t)B * cdiv(D, wt) * cdiv(H, 8) * cdiv(W, 8) * heads * (2 * wt - 1) * 225 * 4;
}

extern "C" int vmg_win3d_attn_bwd(int dtype, const void* q, const void* kv, const float* bq, const float* bkv, const float* table, const void* out,
                                  const float* lse, const void* d_out, void* dq, void* dkv, float* dtable, float* dbq, float* dbkv, void* ws, int B, int D,
                                  int H, int W, int C, int heads, int wt, int sd, int sh, int sw, void* stream) {
  VMG_CHECK(q && kv && table && out && lse && d_out && dq && dkv && dtable, "win3d_attn_bwd: null pointer");
  Win3dK k;
  memset(&k, 0, sizeof(k));
  int nv, lds;
  if (win3d_prepare(k, dtype, B, D, H, W, C, heads, wt, sd, sh, sw, true, &nv, &lds)) return -1;
  k.q = (const char*)q; k.kv = (const char*)kv; k.bq = bq; k.bkv = bkv; k.table = table; k.o_in = (const char*)out; k.lse = const_cast<float*>(lse);
  k.d_o = (const char*)d_out; k.dq = (char*)dq; k.dkv = (char*)dkv; k.dtable = dtable; k.dbq = dbq; k.dbkv = dbkv; k.dtab_ws = (float*)ws;
#ifdef VMG_DIAG
  { const char* e = getenv("VMG_WIN3D_DBG"); k.dbg = e ? atoi(e) : 0; }
#endif
  int rc;
  if (win3d_mfma_ok(dtype, k)) rc = launch_win3d_mfma<true>(k, (hipStream_t)stream);
  else rc = dtype == VMG_BF16 ? launch_win3d<bf16, true>(k, nv, lds, (hipStream_t)stream) : launch_win3d<float, true>(k, nv, lds, (hipStream_t)stream);
  if (rc || !ws) return rc;
  const int nrel = (2 * wt - 1) * 225, nwin = k.B * k.nwd * k.nwh * k.nww;
  hipLaunchKernelGGL(win3d_dtable_reduce_kernel, dim3(cdiv(nrel, 256), heads), dim3(256), 0, (hipStream_t)stream, (const float*)ws, nwin, heads, nrel, dtable);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_win3d_variant(int v) {
  const int prev = g_win3d_variant;
  if (v == 0 || v == 1) g_win3d_variant = v;
  return prev;
}
