// HBM-bound kernels of the path: LayerNorm (fwd/bwd), activation backward, depth<->space.  All are one pass
// over the data with 16-byte vector accesses; statistics and parameter gradients are fp32.
#include "common.h"

namespace {

template <typename T>
struct Vec {  // 16-byte vector of T
  static constexpr int N = 16 / sizeof(T);
  T v[N];
};

// ----------------------------------------------------------------------------------------- act backward
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ ref, T* __restrict__ out, long long n,
                               int act, float slope, float alpha) {
  constexpr int V = Vec<T>::N;
  const long long nv = n / V;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    Vec<T> a = reinterpret_cast<const Vec<T>*>(dy)[i];
    Vec<T> r = reinterpret_cast<const Vec<T>*>(ref)[i];
    Vec<T> o;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float u = to_f32(r.v[e]);
      float d = 1.f;
      if (act == VMG_ACT_RELU) d = u > 0.f ? 1.f : 0.f;
      else if (act == VMG_ACT_LRELU) d = u > 0.f ? 1.f : slope;
      else if (act == VMG_ACT_GELU) d = gelu_erf_grad(u);
      o.v[e] = from_f32<T>(to_f32(a.v[e]) * d * alpha);
    }
    reinterpret_cast<Vec<T>*>(out)[i] = o;
  }
  // tail
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long long i = nv * V; i < n; ++i) {
      const float u = to_f32(ref[i]);
      float d = 1.f;
      if (act == VMG_ACT_RELU) d = u > 0.f ? 1.f : 0.f;
      else if (act == VMG_ACT_LRELU) d = u > 0.f ? 1.f : slope;
      else if (act == VMG_ACT_GELU) d = gelu_erf_grad(u);
      out[i] = from_f32<T>(to_f32(dy[i]) * d * alpha);
    }
}

// ----------------------------------------------------------------------------------------- depth <-> space
// PixelShuffle(2) order (models/vmg.py:380): channel co = c*4 + i*2 + j  <->  pixel (2y+i, 2x+j), channel c.
template <typename T, bool TO_DEPTH>
__global__ void pixel_shuffle_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H, int W, int c) {
  // one thread per (low-res pixel, c): moves 4 elements
  const long long total = (long long)N * H * W * c;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c);
    long long p = i / c;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int n = (int)(p / H);
    const long long lo = (((long long)n * H + y) * W + x) * (4LL * c) + 4 * cc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long hi = (((long long)n * 2 * H + 2 * y + (r >> 1)) * (2 * W) + 2 * x + (r & 1)) * c + cc;
      if (TO_DEPTH) out[lo + r] = in[hi];
      else out[hi] = in[lo + r];
    }
  }
}

// depth-to-space with 16-byte vectors (c a multiple of the vector): a thread reads the 4V consecutive low-res channels that hold V channels of
// the four sub-pixels and writes one vector to each of them.
template <typename T>
__global__ void pixel_shuffle_vec_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H, int W, int c) {
  constexpr int V = Vec<T>::N;
  const int cv = c / V;
  const long long total = (long long)N * H * W * cv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % cv);
    long long p = i / cv;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int n = (int)(p / H);
    const T* lo = in + (((long long)n * H + y) * W + x) * (4LL * c) + 4LL * v * V;
    Vec<T> g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const Vec<T> t = reinterpret_cast<const Vec<T>*>(lo)[k];
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const int q = k * V + j;  // low-res channel 4 * (v*V + q/4) + q%4
        g[q & 3].v[q >> 2] = t.v[j];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long hi = (((long long)n * 2 * H + 2 * y + (r >> 1)) * (2 * W) + 2 * x + (r & 1)) * c + v * V;
      *reinterpret_cast<Vec<T>*>(out + hi) = g[r];
    }
  }
}

// Backward of a conv with a PixelShuffle store and an activation: dpre[n, y, x, 4c + 2i + j] = dy[n, 2y + i, 2x + j, c] * alpha * act'(ref[same])
// in ONE pass (was: depth-to-space of dy, of the saved output, then act_bwd: 7 tensor passes over the HR map instead of 3).  A thread
// owns V consecutive channels of one low-res pixel group: four 16-byte loads per operand, four 16-byte stores of 4V consecutive channels.
template <typename T>
__global__ void pixel_unshuffle_actgrad_kernel(const T* __restrict__ dy, const T* __restrict__ ref, T* __restrict__ out, int N, int H, int W, int c,
                                               int act, float slope, float alpha) {
  constexpr int V = Vec<T>::N;
  const int cv = c / V;
  const long long total = (long long)N * H * W * cv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % cv);
    long long p = i / cv;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int n = (int)(p / H);
    float g[4][V];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long hi = (((long long)n * 2 * H + 2 * y + (r >> 1)) * (2 * W) + 2 * x + (r & 1)) * c + v * V;
      const Vec<T> a = *reinterpret_cast<const Vec<T>*>(dy + hi);
      Vec<T> u = a;
      if (ref) u = *reinterpret_cast<const Vec<T>*>(ref + hi);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        float d = 1.f;
        if (ref) {
          const float uf = to_f32(u.v[e]);
          if (act == VMG_ACT_RELU) d = uf > 0.f ? 1.f : 0.f;
          else if (act == VMG_ACT_LRELU) d = uf > 0.f ? 1.f : slope;
          else if (act == VMG_ACT_GELU) d = gelu_erf_grad(uf);
        }
        g[r][e] = to_f32(a.v[e]) * d * alpha;
      }
    }
    T* lo = out + (((long long)n * H + y) * W + x) * (4LL * c) + 4LL * v * V;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      Vec<T> o;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const int q = k * V + j;  // channel 4 * (v*V + q/4) + q%4 of the low-res pixel
        o.v[j] = from_f32<T>(g[q & 3][q >> 2]);
      }
      reinterpret_cast<Vec<T>*>(lo)[k] = o;
    }
  }
}

// ----------------------------------------------------------------------------------------- LayerNorm
// One row per group of G lanes (G = 16, 32 or 64 chosen from C); each lane keeps its vectors in registers.
constexpr int LN_MAXV = 4;  // vectors per lane -> C <= 64 * 4 * 8 = 2048 (bf16), 1024 (fp32)
constexpr int LN_SUB = 32;  // sub-accumulators of the backward's parameter gradients

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T, int V>
struct alignas(sizeof(T) * V) VecN {
  T v[V];
};

// Row map of the input side (UpdownkeepSampling, models/layers.py:785-793): the LayerNorm row is GATHERED from the feature map, so
// the space<->depth rearrangement costs no pass of its own.  H, W: grid of the LayerNorm rows; Cseg: channels per segment.
//   mode 0: rows are contiguous (M, C).
//   mode 1 ("down", 'n d c (h neih) (w neiw) -> n d h w (neiw neih c)'): row (n, h, w) has 4 segments of Cseg channels; segment
//           neiw*2 + neih comes from pixel (2h + neih, 2w + neiw) of the (2H, 2W, Cseg) input.
//   mode 2 ("up", 'n d (neiw neih c) h w -> n d (h neih) (w neiw) c'): row (n, y, x) of the (H, W) output grid is the Cseg-channel
//           slice (x%2)*2 + (y%2) of pixel (y/2, x/2) of the (H/2, W/2, 4*Cseg) input.
struct LnMap {
  int mode, H, W, Cseg;
};
__device__ __forceinline__ long long ln_src_elem(const LnMap& m, long long row, int c, int C) {
  if (m.mode == 0) return row * C + c;
  const long long hw = (long long)m.H * m.W;
  const long long n = row / hw;
  const int rem = (int)(row - n * hw);
  const int y = rem / m.W, x = rem - y * m.W;
  if (m.mode == 1) {
    const int sg = c / m.Cseg, cc = c - sg * m.Cseg;
    const int neiw = sg >> 1, neih = sg & 1;
    return (((n * 2 * m.H + 2 * y + neih) * (2 * m.W)) + 2 * x + neiw) * m.Cseg + cc;
  }
  const int sg = (x & 1) * 2 + (y & 1);
  return ((n * (m.H >> 1) + (y >> 1)) * (m.W >> 1) + (x >> 1)) * (4LL * m.Cseg) + sg * m.Cseg + c;
}

// NV = vectors per lane (1, 2 or 4): the register arrays are exactly as long as the row needs, the affine weights of a lane's channels
// are loaded once.
template <typename T, int V, int G, int NV, bool MAPPED>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, long long M,
                                                            int C, float eps, const LnMap map_arg) {
  // MAPPED = false: contiguous rows.  As a run-time mode the 64-bit divisions of the space<->depth maps were computed for every vector of
  // every row (the compiler hoists them out of the mode branch): 24 us instead of 12 for the plain LayerNorm at M = 114 688
  const LnMap map = MAPPED ? map_arg : LnMap{0, 0, 0, 0};
  const int nvec = C / V;
  const int gl = threadIdx.x % G;
  const long long groups_per_block = 256 / G;
  const long long stride = (long long)gridDim.x * groups_per_block;
  // the affine weights of a lane's channels, through LDS: read straight from memory they were 2 * NV * V four-byte load instructions per wave
  // (32 at C = 144), 16 cycles of address processing each -- with 24 waves per CU that prologue cost 10 of the kernel's 24 us
  extern __shared__ float ln_aff[];  // [2][C]
  for (int i = threadIdx.x; i < C; i += 256) {
    ln_aff[i] = w[i];
    ln_aff[C + i] = b[i];
  }
  __syncthreads();
  float wr[NV][V], br[NV][V];
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int c = (gl + k * G) * V + e;
      wr[k][e] = c < C ? ln_aff[c] : 0.f;
      br[k][e] = c < C ? ln_aff[C + c] : 0.f;
    }
  // STRAIGHT-LINE body, R rows per lane group and iteration: all loads of the R rows first (row indices clamped, so that no load sits behind
  // a branch), then the reductions, then the stores.  The earlier form -- the next row prefetched under `if (row + stride < M)` inside a
  // per-lane loop -- compiled to `s_waitcnt vmcnt(0)` at every join: each iteration waited for its own STORES, and the kernel ran at
  // 2.7 TB/s (24 us at M = 114 688, C = 144) where this form reaches 5.8 (tools/ubench/ln_probe.cpp)
  constexpr int R = 2;
  for (long long row = blockIdx.x * groups_per_block + threadIdx.x / G; row < M; row += R * stride) {
    VecN<T, V> buf[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long rr = row + r * stride < M ? row + r * stride : M - 1;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int vi = gl + k * G;
        if (vi < nvec) buf[r][k] = *reinterpret_cast<const VecN<T, V>*>(x + ln_src_elem(map, rr, vi * V, C));
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long rr = row + r * stride;
      const bool live = rr < M;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        if (gl + k * G < nvec) {
#pragma unroll
          for (int e = 0; e < V; ++e) s += to_f32(buf[r][k].v[e]);
        }
      }
      const float mu = group_sum<G>(s) / C;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        if (gl + k * G < nvec) {
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const float d = to_f32(buf[r][k].v[e]) - mu;
            q += d * d;
          }
        }
      }
      const float rs = rsqrtf(group_sum<G>(q) / C + eps);
      if (live && gl == 0 && mean) {
        mean[rr] = mu;
        rstd[rr] = rs;
      }
      T* yr = y + rr * C;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int vi = gl + k * G;
        if (live && vi < nvec) {
          VecN<T, V> o;
#pragma unroll
          for (int e = 0; e < V; ++e) o.v[e] = from_f32<T>((to_f32(buf[r][k].v[e]) - mu) * rs * wr[k][e] + br[k][e]);
          reinterpret_cast<VecN<T, V>*>(yr)[vi] = o;
        }
      }
    }
  }
}

struct LnMore {
  const void* p[4];
  int n;
};

// SPEC: 0 = `add` / `more` looked at at run time (a load behind a run-time branch, even a wave-uniform one, is followed by `s_waitcnt
// vmcnt(0)`: the generic form serialises its loads); 2 = add, no further gradients; 3 = add + four further gradients -- the two forms the
// TAB blocks use (functional.layer_norm_skip / layer_norm_fan), compiled without those branches
template <typename T, int V, int G, int NV, bool MAPPED, int SPEC>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ w, T* __restrict__ dx,
                                                            float* __restrict__ dw, float* __restrict__ db, long long M, int C,
                                                            const LnMap map_arg, const T* __restrict__ add, float* __restrict__ ws,
                                                            unsigned int* __restrict__ counter, const LnMore more) {
  const LnMap map = MAPPED ? map_arg : LnMap{0, 0, 0, 0};  // (see layernorm_fwd_kernel)
  const bool has_add = SPEC == 0 ? add != nullptr : true;
  const int nmore = SPEC == 0 ? more.n : (SPEC == 3 ? 4 : 0);
  // more: up to four FURTHER gradients of the LayerNorm output (contiguous rows, like dy): a normalised tensor that feeds several consumers
  // (the MorphFC mixer reads LN2(x) five times: H branch, W branch, RCAB conv, RCAB residual, tanh gate) collects one gradient per consumer;
  // they are summed here in fp32 on the way in -- autograd's pairwise adds cost three passes each over the (N, C) tensor.
  // add (optional, contiguous rows): a gradient that reaches the same tensor by a skip connection; dx = add + LayerNorm backward, so the
  // sum does not cost a pass of its own (the TAB residuals: x feeds the norm AND the residual add)
  const int nvec = C / V;
  const int gl = threadIdx.x % G;
  const long long groups_per_block = 256 / G;
  const long long stride = (long long)gridDim.x * groups_per_block;
  extern __shared__ float smw[];  // [4 waves][2][C]: the parameter-gradient sums at the end; first the affine weights on their way to the lanes
  for (int i = threadIdx.x; i < C; i += 256) smw[i] = w[i];  // (through LDS: see layernorm_fwd_kernel)
  __syncthreads();
  float pdw[NV][V], pdb[NV][V], wr[NV][V];
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      pdw[k][e] = pdb[k][e] = 0.f;
      const int c = (gl + k * G) * V + e;
      wr[k][e] = c < C ? smw[c] : 0.f;
    }
  // straight-line body like layernorm_fwd_kernel: the loads of R rows (clamped indices), the reductions, the stores.  The parameter-gradient
  // sums skip a clamped (repeated) row by a zero factor
  constexpr int R = 2;
  for (long long row = blockIdx.x * groups_per_block + threadIdx.x / G; row < M; row += R * stride) {
    VecN<T, V> bx[R][NV], bg[R][NV], ba[R][NV];
    float rmu[R], rrs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long rr = row + r * stride < M ? row + r * stride : M - 1;
      rmu[r] = mean[rr];
      rrs[r] = rstd[rr];
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int vi = gl + k * G;
        if (vi < nvec) {
          bx[r][k] = *reinterpret_cast<const VecN<T, V>*>(x + ln_src_elem(map, rr, vi * V, C));
          bg[r][k] = reinterpret_cast<const VecN<T, V>*>(dy + rr * C)[vi];
          if (has_add) ba[r][k] = reinterpret_cast<const VecN<T, V>*>(add + rr * C)[vi];  // (wave-uniform)
          if (nmore) {  // (wave-uniform)
            float acc[V];
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] = to_f32(bg[r][k].v[e]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (j < nmore) {
                const VecN<T, V> t = reinterpret_cast<const VecN<T, V>*>(reinterpret_cast<const T*>(more.p[j]) + rr * C)[vi];
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += to_f32(t.v[e]);
              }
            }
#pragma unroll
            for (int e = 0; e < V; ++e) bg[r][k].v[e] = from_f32<T>(acc[e]);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long rr = row + r * stride;
      const bool live = rr < M;
      const float keep = live ? 1.f : 0.f;
      const float mu = rmu[r], rs = rrs[r];
      float s1 = 0.f, s2 = 0.f;  // sum(g), sum(g * xhat) with g = dy * w
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        if (gl + k * G < nvec) {
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const float xh = (to_f32(bx[r][k].v[e]) - mu) * rs;
            const float d = to_f32(bg[r][k].v[e]) * keep;
            const float g = d * wr[k][e];
            s1 += g;
            s2 += g * xh;
            pdw[k][e] += d * xh;
            pdb[k][e] += d;
          }
        }
      }
      s1 = group_sum<G>(s1) / C;
      s2 = group_sum<G>(s2) / C;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int vi = gl + k * G;
        if (live && vi < nvec) {
          VecN<T, V> o;
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const float xh = (to_f32(bx[r][k].v[e]) - mu) * rs;
            const float g = to_f32(bg[r][k].v[e]) * wr[k][e];
            o.v[e] = from_f32<T>(rs * (g - s1 - xh * s2));
          }
          if (has_add) {
#pragma unroll
            for (int e = 0; e < V; ++e) o.v[e] = from_f32<T>(to_f32(o.v[e]) + to_f32(ba[r][k].v[e]));
          }
          *reinterpret_cast<VecN<T, V>*>(dx + ln_src_elem(map, rr, vi * V, C)) = o;
        }
      }
    }
  }
  // parameter gradients: block-level sum in LDS, then one float atomic per channel per block
#ifdef VMG_DIAG
  if (dw == nullptr) return;  // (tools/bench_ln.py, VMG_LN_DBG=1: the kernel without its parameter-gradient tail)
#endif
  // (LDS float atomics from all 256 threads cost 12 us here: the row groups of a wave are summed with shuffles, the four waves through LDS)
  __syncthreads();  // (every lane has taken its affine weights out of smw long ago; the barrier orders the slowest wave's reads before the sums)
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float a = pdw[k][e], b2 = pdb[k][e];
#pragma unroll
      for (int o = G; o < 64; o <<= 1) {
        a += __shfl_xor(a, o, 64);
        b2 += __shfl_xor(b2, o, 64);
      }
      const int vi = gl + k * G;
      if ((threadIdx.x & 63) < G && vi < nvec) {
        smw[(wv * 2) * C + vi * V + e] = a;
        smw[(wv * 2 + 1) * C + vi * V + e] = b2;
      }
    }
  }
  __syncthreads();
  float* sm = smw;  // [2][C]: the block's sums, in wave 0's slots
  for (int i = threadIdx.x; i < 2 * C; i += 256) sm[i] = smw[i] + smw[2 * C + i] + smw[4 * C + i] + smw[6 * C + i];
  __syncthreads();
#ifdef VMG_DIAG
  if (db == nullptr) return;  // (VMG_LN_DBG=2: with the block-level LDS sums, without the global ones)
#endif
  if (ws == nullptr) {  // (no workspace yet: one float atomic per channel and block -- 512 blocks queueing on 2C addresses cost 18 of 43 us at M = 114 688)
    for (int i = threadIdx.x; i < C; i += 256) {
      atomicAdd(&dw[i], sm[i]);
      atomicAdd(&db[i], sm[C + i]);
    }
    return;
  }
  // LN_SUB sub-accumulators instead of one: block b adds (device-scope float atomics, performed at the memory side) to sub-accumulator
  // b % LN_SUB -- 16 blocks per address instead of 512 -- and the LAST block to arrive collects them (atomic exchange with zero: read and
  // re-arm in one), adds the sums to dw / db and resets the arrival counter.  Only atomics carry data between blocks, so no L2
  // write-back / invalidate fence is needed (a __threadfence() per block made this kernel 7x slower): s_waitcnt vmcnt(0) before the
  // ticket orders a block's adds before its arrival.
  float* acc = ws + (size_t)(blockIdx.x % LN_SUB) * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(&acc[i], sm[i]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __shared__ unsigned int ticket;
  __syncthreads();
  if (threadIdx.x == 0) ticket = atomicAdd(counter, 1u);
  __syncthreads();
  if (ticket != gridDim.x - 1) return;
  const int nsub = min((int)gridDim.x, LN_SUB);
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float v[LN_SUB];
#pragma unroll
    for (int sb = 0; sb < LN_SUB; ++sb) v[sb] = sb < nsub ? atomicExch(&ws[(size_t)sb * 2 * C + i], 0.f) : 0.f;
    float t = 0.f;
#pragma unroll
    for (int sb = 0; sb < LN_SUB; ++sb) t += v[sb];
    if (i < C) dw[i] += t;
    else db[i - C] += t;
  }
  if (threadIdx.x == 0) atomicExch(counter, 0u);  // ready for the next launch on this stream
}

// per-device workspace of the backward's sub-accumulators (zero between launches) + the arrival counter: allocated on the first call outside a
// stream capture.  One per device: LayerNorm backward launches of a device must be stream-ordered (autograd runs them on the forward's stream).
static float* g_ln_ws[VMG_MAX_DEVICES] = {};
static unsigned int* g_ln_counter[VMG_MAX_DEVICES] = {};
constexpr size_t LN_WS_FLOATS = (size_t)LN_SUB * 2 * 2048;  // C <= 64 lanes * LN_MAXV vectors * 8 = 2048
static float* ln_workspace(hipStream_t st, unsigned int** counter) {
  const int dev = vmg_current_device();
  if (!g_ln_ws[dev]) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    char* p = nullptr;
    if (hipMalloc((void**)&p, LN_WS_FLOATS * 4 + 64) != hipSuccess || hipMemset(p, 0, LN_WS_FLOATS * 4 + 64) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    g_ln_ws[dev] = reinterpret_cast<float*>(p);
    g_ln_counter[dev] = reinterpret_cast<unsigned int*>(p + LN_WS_FLOATS * 4);
  }
  *counter = g_ln_counter[dev];
  return g_ln_ws[dev];
}

// lanes per row: the G in {16, 32, 64} that wastes the fewest lane slots (G * ceil(nvec / G)), the smaller G on a tie -- more rows per
// block and more vectors (= loads in flight) per lane: C = 144 bf16 (18 vectors) runs on 16 lanes x 2 vectors, not 32 x 1
int ln_group(int nvec) {
  int best = 64, slots = 1 << 30;
  for (int g = 64; g >= 16; g >>= 1) {
    const int nv = (nvec + g - 1) / g;
    if (nv > LN_MAXV) continue;
    const int nvp = nv == 3 ? 4 : nv;  // (instantiated: 1, 2, 4 vectors per lane)
    if (g * nvp <= slots) { slots = g * nvp; best = g; }
  }
  return best;
}


// ----------------------------------------------------------------------------------------- frame gather
// dst frame f = sum over k < nsrc of src frame idx[f * nsrc + k] (an index < 0 is skipped), frames = contiguous blocks of `fv` 16-byte
// vectors.  The recurrence (model.py: Trajectory_multi_head) runs its two direction sweeps as one batch: step j works on
// [frame t-1-j | frame j] of every clip.  That (t, 2n) arrangement is ONE gather from the batch-major (n, t) features (nsrc = 1) and its
// gradient ONE gather-add (nsrc = 2, fp32 sum, one rounding) -- torch spelled it transpose + flip + cat, three passes each way.
template <typename T, int NSRC>
__global__ __launch_bounds__(256) void frame_gather_kernel(const T* __restrict__ src, T* __restrict__ dst, const int* __restrict__ idx, long long fv,
                                                           int chunks) {
  constexpr int VN = 16 / (int)sizeof(T);
  typedef VecN<T, VN> Vec;
  const int f = blockIdx.x / chunks, ck = blockIdx.x - f * chunks;
  int s[NSRC];
#pragma unroll
  for (int k = 0; k < NSRC; ++k) s[k] = idx[f * NSRC + k];
  const long long v0 = fv * ck / chunks, v1 = fv * (ck + 1) / chunks;
  Vec* d = reinterpret_cast<Vec*>(dst) + (long long)f * fv;
  for (long long v = v0 + threadIdx.x; v < v1; v += 256) {
    if (NSRC == 1) {
      d[v] = reinterpret_cast<const Vec*>(src)[(long long)s[0] * fv + v];  // (the host rejects a negative index when nsrc = 1)
    } else {
      float acc[VN];
#pragma unroll
      for (int e = 0; e < VN; ++e) acc[e] = 0.f;
#pragma unroll
      for (int k = 0; k < NSRC; ++k) {
        const Vec a = reinterpret_cast<const Vec*>(src)[(long long)(s[k] < 0 ? 0 : s[k]) * fv + v];
        const float keep = s[k] < 0 ? 0.f : 1.f;
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[e] += to_f32(a.v[e]) * keep;
      }
      Vec o;
#pragma unroll
      for (int e = 0; e < VN; ++e) o.v[e] = from_f32<T>(acc[e]);
      d[v] = o;
    }
  }
}
// out = sum of 2..4 tensors of one dtype, fp32 sum, one rounding: the gradient of a tensor with several consumers (functional.fan_out) in ONE pass -- autograd
// sums them pairwise (k - 1 passes of 3 tensors each, a rounding per pass).
struct SumNK { const void* src[4]; int n; };
template <typename T>
__global__ __launch_bounds__(256) void sum_n_kernel(const SumNK k, T* __restrict__ out, long long nv) {
  constexpr int VN = 16 / (int)sizeof(T);
  typedef VecN<T, VN> Vec;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    float acc[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) acc[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < k.n) {  // (wave-uniform)
        const Vec a = reinterpret_cast<const Vec*>(k.src[j])[i];
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[e] += to_f32(a.v[e]);
      }
    }
    Vec o;
#pragma unroll
    for (int e = 0; e < VN; ++e) o.v[e] = from_f32<T>(acc[e]);
    reinterpret_cast<Vec*>(out)[i] = o;
  }
}

// fp32 scatter accumulators (flow-warp backward, the trajectory attention's key / value gradient banks) are rounded to the tensor dtype ONCE, when complete.
// This pass does the rounding, adds an optional further gradient (`add`: the frame's other uses, fp32 sum before the rounding) and leaves the accumulator
// ZERO behind -- so that the host can hand the same buffer to the next scatter without a fill pass (round 4: 25 fills + 25 casts + 8 adds per train step
// were torch kernels).
template <typename T>
__global__ __launch_bounds__(256) void cast_clear_kernel(float* __restrict__ acc, const T* __restrict__ add, T* __restrict__ out, long long n4) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 v = reinterpret_cast<const f32x4*>(acc)[i];
    if (add) {
      const VecN<T, 4> a = reinterpret_cast<const VecN<T, 4>*>(add)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += to_f32(a.v[e]);
    }
    VecN<T, 4> o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o.v[e] = from_f32<T>(v[e]);
    reinterpret_cast<VecN<T, 4>*>(out)[i] = o;
    reinterpret_cast<f32x4*>(acc)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// The step tensors of the lock-step recurrence, addressed through a POINTER LIST (the t step tensors are separate allocations: the residual chain
// of step j writes its own output).  Step j holds 2n frames: rows [0, n) = the backward sweep at frame t-1-j, rows [n, 2n) = the forward sweep at
// frame j of every clip.
//   mode 0: steps -> back, fwd (n, t) in frame order         (torch: t splits, two stacks over 2t views)
//   mode 1: back, fwd -> steps                               (its backward: t concatenations)
//   mode 2: steps -> a (n, t):  a[i, f] = steps[t-1-f][i] + steps[f][n+i]   (fp32 sum, one rounding: the backward of frame_gather's pairing, without the stack
//                                                                            autograd builds from the t step gradients first)
constexpr int PS_MAX_T = 64;
struct PairStepsK {
  void* step[PS_MAX_T];
  void* a;
  void* b;
  int n, t, chunks, mode;
  long long fv;
};
template <typename T>
__global__ __launch_bounds__(256) void pair_steps_kernel(const PairStepsK k) {
  constexpr int VN = 16 / (int)sizeof(T);
  typedef VecN<T, VN> Vec;
  const int f = blockIdx.x / k.chunks, ck = blockIdx.x - f * k.chunks;
  const long long v0 = k.fv * ck / k.chunks, v1 = k.fv * (ck + 1) / k.chunks;
  if (k.mode == 2) {
    const int i = f / k.t, fr = f - i * k.t;  // dst frame (clip i, frame fr)
    const Vec* s0 = reinterpret_cast<const Vec*>(k.step[k.t - 1 - fr]) + (long long)i * k.fv;
    const Vec* s1 = reinterpret_cast<const Vec*>(k.step[fr]) + (long long)(k.n + i) * k.fv;
    Vec* d = reinterpret_cast<Vec*>(k.a) + (long long)f * k.fv;
    for (long long v = v0 + threadIdx.x; v < v1; v += 256) {
      const Vec x = s0[v], y = s1[v];
      Vec o;
#pragma unroll
      for (int e = 0; e < VN; ++e) o.v[e] = from_f32<T>(to_f32(x.v[e]) + to_f32(y.v[e]));
      d[v] = o;
    }
    return;
  }
  const int j = f / (2 * k.n), r = f - j * 2 * k.n;  // step j, row r
  Vec* st = reinterpret_cast<Vec*>(k.step[j]) + (long long)r * k.fv;
  Vec* ot = r < k.n ? reinterpret_cast<Vec*>(k.a) + ((long long)r * k.t + (k.t - 1 - j)) * k.fv
                    : reinterpret_cast<Vec*>(k.b) + ((long long)(r - k.n) * k.t + j) * k.fv;
  const Vec* src = k.mode == 0 ? st : ot;
  Vec* dst = k.mode == 0 ? ot : st;
  for (long long v = v0 + threadIdx.x; v < v1; v += 256) dst[v] = src[v];
}
}  // namespace

extern "C" int vmg_act_bwd(int dtype, const void* dy, const void* ref, void* out, int64_t n, int act, float slope, float alpha,
                           void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "act_bwd: bad dtype");
  VMG_CHECK(dy && ref && out && n > 0, "act_bwd: bad arguments");
  VMG_CHECK(((uintptr_t)dy | (uintptr_t)ref | (uintptr_t)out) % 16 == 0, "act_bwd: pointers must be 16-byte aligned");
  const int blocks = (int)(cdiv64(n / 8 + 1, 256) > 2048 ? 2048 : cdiv64(n / 8 + 1, 256));
  if (dtype == VMG_BF16)
    hipLaunchKernelGGL(act_bwd_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy, (const bf16*)ref,
                       (bf16*)out, (long long)n, act, slope, alpha);
  else
    hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (const float*)ref,
                       (float*)out, (long long)n, act, slope, alpha);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_frame_gather(int dtype, const void* src, void* dst, const int* idx, int64_t frame_elems, int n_src_frames, int n_dst_frames, int nsrc,
                                void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "frame_gather: bad dtype");
  VMG_CHECK(src && dst && idx && frame_elems > 0 && n_src_frames > 0 && n_dst_frames > 0 && (nsrc == 1 || nsrc == 2), "frame_gather: bad arguments (nsrc 1 or 2)");
  const int vn = dtype == VMG_BF16 ? 8 : 4;
  VMG_CHECK(frame_elems % vn == 0 && ((uintptr_t)src | (uintptr_t)dst) % 16 == 0, "frame_gather: frames must be whole 16-byte vectors, tensors 16-byte aligned");
  (void)n_src_frames;  // (the index table lives on the device: the caller guarantees idx < n_src_frames; see kernels.frame_gather)
  const long long fv = frame_elems / vn;
  long long chunks = cdiv64(2048, n_dst_frames);  // ~2 048 blocks in all
  if (chunks > cdiv64(fv, 256)) chunks = cdiv64(fv, 256);
  if (chunks < 1) chunks = 1;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(n_dst_frames * chunks));
#define FG_LAUNCH(T, NS) hipLaunchKernelGGL((frame_gather_kernel<T, NS>), grid, dim3(256), 0, st, (const T*)src, (T*)dst, idx, fv, (int)chunks)
  if (dtype == VMG_BF16) { if (nsrc == 1) FG_LAUNCH(bf16, 1); else FG_LAUNCH(bf16, 2); }
  else { if (nsrc == 1) FG_LAUNCH(float, 1); else FG_LAUNCH(float, 2); }
#undef FG_LAUNCH
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_sum_n(int dtype, const void* const* srcs, int nsrc, void* out, int64_t n, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "sum_n: bad dtype");
  const int vn = dtype == VMG_BF16 ? 8 : 4;
  VMG_CHECK(srcs && out && nsrc >= 2 && nsrc <= 4 && n > 0 && n % vn == 0 && (uintptr_t)out % 16 == 0, "sum_n: 2..4 sources, whole 16-byte vectors, aligned tensors");
  SumNK k;
  k.n = nsrc;
  for (int j = 0; j < 4; ++j) {
    k.src[j] = j < nsrc ? srcs[j] : nullptr;
    VMG_CHECK(j >= nsrc || (srcs[j] && (uintptr_t)srcs[j] % 16 == 0), "sum_n: source %d null or unaligned", j);
  }
  const long long nv = n / vn;
  const int blocks = (int)(cdiv64(nv, 256) > 4096 ? 4096 : cdiv64(nv, 256));
  if (dtype == VMG_BF16) hipLaunchKernelGGL(sum_n_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, k, (bf16*)out, nv);
  else hipLaunchKernelGGL(sum_n_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, k, (float*)out, nv);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_cast_clear(int dtype, float* acc, const void* add, void* out, int64_t n, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "cast_clear: bad dtype");
  VMG_CHECK(acc && out && n > 0 && n % 4 == 0, "cast_clear: n must be a positive multiple of 4");
  VMG_CHECK((uintptr_t)acc % 16 == 0 && (uintptr_t)out % 8 == 0 && (uintptr_t)add % 8 == 0, "cast_clear: 16-byte aligned accumulator, 8-byte aligned tensors");
  VMG_CHECK(dtype == VMG_BF16 || ((uintptr_t)out % 16 == 0 && (uintptr_t)add % 16 == 0), "cast_clear: 16-byte aligned fp32 tensors");
  const long long n4 = n / 4;
  const int blocks = (int)(cdiv64(n4, 256) > 4096 ? 4096 : cdiv64(n4, 256));
  if (dtype == VMG_BF16) hipLaunchKernelGGL(cast_clear_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, acc, (const bf16*)add, (bf16*)out, n4);
  else hipLaunchKernelGGL(cast_clear_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, acc, (const float*)add, (float*)out, n4);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_pair_steps(int dtype, int mode, void* const* steps, void* a, void* b, int n, int t, int64_t frame_elems, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "pair_steps: bad dtype");
  VMG_CHECK(steps && a && (b || mode == 2) && n > 0 && t > 0 && t <= PS_MAX_T && frame_elems > 0 && mode >= 0 && mode <= 2, "pair_steps: bad arguments (t <= %d)", PS_MAX_T);
  const int vn = dtype == VMG_BF16 ? 8 : 4;
  VMG_CHECK(frame_elems % vn == 0 && ((uintptr_t)a | (uintptr_t)b) % 16 == 0, "pair_steps: frames must be whole 16-byte vectors, tensors 16-byte aligned");
  PairStepsK k;
  for (int j = 0; j < t; ++j) {
    VMG_CHECK(steps[j] && (uintptr_t)steps[j] % 16 == 0, "pair_steps: step %d: null or unaligned", j);
    k.step[j] = steps[j];
  }
  k.a = a; k.b = b; k.n = n; k.t = t; k.mode = mode;
  k.fv = frame_elems / vn;
  const long long frames = mode == 2 ? (long long)n * t : 2LL * n * t;
  long long chunks = cdiv64(2048, frames);  // ~2 048 blocks in all
  if (chunks > cdiv64(k.fv, 256)) chunks = cdiv64(k.fv, 256);
  if (chunks < 1) chunks = 1;
  k.chunks = (int)chunks;
  const dim3 grid((unsigned)(frames * chunks));
  if (dtype == VMG_BF16) hipLaunchKernelGGL(pair_steps_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, k);
  else hipLaunchKernelGGL(pair_steps_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, k);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_pixel_shuffle(int dtype, const void* in, void* out, int N, int H, int W, int c, int to_depth, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "pixel_shuffle: bad dtype");
  VMG_CHECK(in && out && N > 0 && H > 0 && W > 0 && c > 0, "pixel_shuffle: bad arguments");
  const long long total = (long long)N * H * W * c;
  const int blocks = (int)(cdiv64(total, 256) > 4096 ? 4096 : cdiv64(total, 256));
  hipStream_t st = (hipStream_t)stream;
  const int vn = dtype == VMG_BF16 ? 8 : 4;
  if (!to_depth && c % vn == 0 && ((uintptr_t)in | (uintptr_t)out) % 16 == 0) {
    const long long tv = (long long)N * H * W * (c / vn);
    const int bv = (int)(cdiv64(tv, 256) > 8192 ? 8192 : cdiv64(tv, 256));
    if (dtype == VMG_BF16) hipLaunchKernelGGL(pixel_shuffle_vec_kernel<bf16>, dim3(bv), dim3(256), 0, st, (const bf16*)in, (bf16*)out, N, H, W, c);
    else hipLaunchKernelGGL(pixel_shuffle_vec_kernel<float>, dim3(bv), dim3(256), 0, st, (const float*)in, (float*)out, N, H, W, c);
    VMG_LAUNCH_CHECK();
    return 0;
  }
  if (dtype == VMG_BF16) {
    if (to_depth) hipLaunchKernelGGL((pixel_shuffle_kernel<bf16, true>), dim3(blocks), dim3(256), 0, st, (const bf16*)in, (bf16*)out, N, H, W, c);
    else hipLaunchKernelGGL((pixel_shuffle_kernel<bf16, false>), dim3(blocks), dim3(256), 0, st, (const bf16*)in, (bf16*)out, N, H, W, c);
  } else {
    if (to_depth) hipLaunchKernelGGL((pixel_shuffle_kernel<float, true>), dim3(blocks), dim3(256), 0, st, (const float*)in, (float*)out, N, H, W, c);
    else hipLaunchKernelGGL((pixel_shuffle_kernel<float, false>), dim3(blocks), dim3(256), 0, st, (const float*)in, (float*)out, N, H, W, c);
  }
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_pixel_unshuffle_actgrad(int dtype, const void* dy, const void* ref, void* out, int N, int H, int W, int c, int act, float slope,
                                           float alpha, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "pixel_unshuffle_actgrad: bad dtype");
  const int V = dtype == VMG_BF16 ? 8 : 4;
  VMG_CHECK(dy && out && N > 0 && H > 0 && W > 0 && c > 0 && c % V == 0, "pixel_unshuffle_actgrad: channels must be a multiple of %d", V);
  VMG_CHECK((act == VMG_ACT_NONE) == (ref == nullptr), "pixel_unshuffle_actgrad: the activation needs its reference tensor (and only it)");
  VMG_CHECK((uintptr_t)dy % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)ref % 16 == 0, "pixel_unshuffle_actgrad: 16-byte aligned tensors");
  const long long total = (long long)N * H * W * (c / V);
  const int blocks = (int)(cdiv64(total, 256) > 8192 ? 8192 : cdiv64(total, 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VMG_BF16)
    hipLaunchKernelGGL(pixel_unshuffle_actgrad_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)dy, (const bf16*)ref, (bf16*)out, N, H, W, c, act, slope, alpha);
  else
    hipLaunchKernelGGL(pixel_unshuffle_actgrad_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)dy, (const float*)ref, (float*)out, N, H, W, c, act, slope, alpha);
  VMG_LAUNCH_CHECK();
  return 0;
}

template <typename T, int V>
static int ln_fwd_t(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, long long M, int C, float eps,
                    hipStream_t st, const LnMap map) {
  const int nvec = C / V;
  const int G = ln_group(nvec);
  VMG_CHECK(nvec <= G * LN_MAXV, "layernorm: C = %d too large", C);
  const long long rows_per_block = 256 / G;
  // several rows per lane group (the hoisted affine weights pay), and no more blocks than are RESIDENT at once (4 per CU at <= 128 registers):
  // with 1 536 blocks and 4-5 resident per CU a second, mostly empty round of blocks doubled the kernel's duration
  const int blocks = (int)(cdiv64(M, rows_per_block * 4) > 1024 ? 1024 : cdiv64(M, rows_per_block * 4));
  const int nv = (nvec + G - 1) / G;
#define LN_LAUNCH_M(GG, NV, MP) hipLaunchKernelGGL((layernorm_fwd_kernel<T, V, GG, NV, MP>), dim3(blocks), dim3(256), 2 * C * 4, st, (const T*)x, w, b, (T*)y, mean, rstd, M, C, eps, map)
#define LN_LAUNCH(GG, NV) do { if (map.mode == 0) LN_LAUNCH_M(GG, NV, false); else LN_LAUNCH_M(GG, NV, true); } while (0)
#define LN_LAUNCH_G(GG) do { if (nv == 1) LN_LAUNCH(GG, 1); else if (nv == 2) LN_LAUNCH(GG, 2); else LN_LAUNCH(GG, 4); } while (0)
  if (G == 16) LN_LAUNCH_G(16);
  else if (G == 32) LN_LAUNCH_G(32);
  else LN_LAUNCH_G(64);
#undef LN_LAUNCH_G
#undef LN_LAUNCH
#undef LN_LAUNCH_M
  VMG_LAUNCH_CHECK();
  return 0;
}

template <typename T, int V>
static int ln_bwd_t(const void* dy, const void* x, const float* mean, const float* rstd, const float* w, void* dx, float* dw, float* db,
                    long long M, int C, hipStream_t st, const LnMap map, const void* add, const LnMore more) {
  const int nvec = C / V;
  const int G = ln_group(nvec);
  VMG_CHECK(nvec <= G * LN_MAXV, "layernorm: C = %d too large", C);
  const long long rows_per_block = 256 / G;
  int blocks = (int)(cdiv64(M, rows_per_block * 8) > 512 ? 512 : cdiv64(M, rows_per_block * 8));  // (every block ends with 2C float atomics on the same addresses)
#ifdef VMG_DIAG
  { const char* e = getenv("VMG_LN_BLOCKS"); if (e && atoi(e) > 0) blocks = atoi(e); }
  { const char* e = getenv("VMG_LN_DBG"); if (e && atoi(e) == 1) dw = nullptr; if (e && atoi(e) == 2) db = nullptr; }
#endif
  const int lds = 8 * C * 4;  // [4 waves][2][C] floats
  const int nv = (nvec + G - 1) / G;
  unsigned int* counter = nullptr;
  float* ws = (blocks > LN_SUB && C <= 2048) ? ln_workspace(st, &counter) : nullptr;  // (null: plain float atomics on dw / db)
#define LN_LAUNCH_M(GG, NV, MP, SP) hipLaunchKernelGGL((layernorm_bwd_kernel<T, V, GG, NV, MP, SP>), dim3(blocks), dim3(256), lds, st, (const T*)dy, (const T*)x, mean, rstd, w, (T*)dx, dw, db, M, C, map, (const T*)add, ws, counter, more)
#define LN_LAUNCH(GG, NV) do { \
    if (map.mode != 0) LN_LAUNCH_M(GG, NV, true, 0); \
    else if constexpr (sizeof(T) == 2 && V == 8) { \
      if (add && more.n == 0) LN_LAUNCH_M(GG, NV, false, 2); else if (add && more.n == 4) LN_LAUNCH_M(GG, NV, false, 3); else LN_LAUNCH_M(GG, NV, false, 0); \
    } else LN_LAUNCH_M(GG, NV, false, 0); } while (0)
#define LN_LAUNCH_G(GG) do { if (nv == 1) LN_LAUNCH(GG, 1); else if (nv == 2) LN_LAUNCH(GG, 2); else LN_LAUNCH(GG, 4); } while (0)
  if (G == 16) LN_LAUNCH_G(16);
  else if (G == 32) LN_LAUNCH_G(32);
  else LN_LAUNCH_G(64);
#undef LN_LAUNCH_G
#undef LN_LAUNCH
#undef LN_LAUNCH_M
  VMG_LAUNCH_CHECK();
  return 0;
}

// widest vector (in elements) that divides C
static int ln_vec(int dtype, int C) {
  if (dtype == VMG_BF16) return C % 8 == 0 ? 8 : (C % 4 == 0 ? 4 : (C % 2 == 0 ? 2 : 1));
  return C % 4 == 0 ? 4 : (C % 2 == 0 ? 2 : 1);
}

static int ln_fwd_impl(int dtype, const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, int64_t M, int C, float eps,
                       void* stream, const LnMap map, int vmax) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "layernorm_fwd: bad dtype");
  VMG_CHECK(x && w && b && y && M > 0 && C > 0, "layernorm_fwd: bad arguments");
  VMG_CHECK(((uintptr_t)x | (uintptr_t)y) % 16 == 0, "layernorm_fwd: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int v = ln_vec(dtype, C);
  while (v > vmax) v >>= 1;
  if (dtype == VMG_BF16) {
    switch (v) {
      case 8: return ln_fwd_t<bf16, 8>(x, w, b, y, mean, rstd, M, C, eps, st, map);
      case 4: return ln_fwd_t<bf16, 4>(x, w, b, y, mean, rstd, M, C, eps, st, map);
      case 2: return ln_fwd_t<bf16, 2>(x, w, b, y, mean, rstd, M, C, eps, st, map);
      default: return ln_fwd_t<bf16, 1>(x, w, b, y, mean, rstd, M, C, eps, st, map);
    }
  }
  switch (v) {
    case 4: return ln_fwd_t<float, 4>(x, w, b, y, mean, rstd, M, C, eps, st, map);
    case 2: return ln_fwd_t<float, 2>(x, w, b, y, mean, rstd, M, C, eps, st, map);
    default: return ln_fwd_t<float, 1>(x, w, b, y, mean, rstd, M, C, eps, st, map);
  }
}

extern "C" int vmg_layernorm_fwd(int dtype, const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                                 int64_t M, int C, float eps, void* stream) {
  return ln_fwd_impl(dtype, x, w, b, y, mean, rstd, M, C, eps, stream, LnMap{0, 0, 0, 0}, 8);
}

// widest vector that divides the segment length of a space<->depth map (a vector must not straddle two segments)
static int ln_map_vmax(int dtype, int cseg) { return ln_vec(dtype, cseg); }
static int ln_map_check(int mode, int N, int H, int W, int cseg, int64_t* M, int* C) {
  VMG_CHECK((mode == 1 || mode == 2) && N > 0 && H > 0 && W > 0 && cseg > 0, "space_depth_ln: mode 1 (down) or 2 (up), positive sizes");
  VMG_CHECK(mode == 1 || (H % 2 == 0 && W % 2 == 0), "space_depth_ln (up): the output grid must be even");
  *M = (int64_t)N * H * W;
  *C = mode == 1 ? 4 * cseg : cseg;
  return 0;
}

extern "C" int vmg_space_depth_ln_fwd(int dtype, int mode, const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                                      int N, int H, int W, int cseg, float eps, void* stream) {
  int64_t M;
  int C;
  if (ln_map_check(mode, N, H, W, cseg, &M, &C)) return -1;
  return ln_fwd_impl(dtype, x, w, b, y, mean, rstd, M, C, eps, stream, LnMap{mode, H, W, cseg}, ln_map_vmax(dtype, cseg));
}

static int ln_bwd_impl(int dtype, const void* dy, const void* x, const float* mean, const float* rstd, const float* w, void* dx, float* dw,
                       float* db, int64_t M, int C, void* stream, const LnMap map, int vmax, const void* add = nullptr, const LnMore more = LnMore{{nullptr, nullptr, nullptr, nullptr}, 0}) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "layernorm_bwd: bad dtype");
  VMG_CHECK(dy && x && mean && rstd && w && dx && dw && db && M > 0 && C > 0, "layernorm_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int v = ln_vec(dtype, C);
  while (v > vmax) v >>= 1;
  if (dtype == VMG_BF16) {
    switch (v) {
      case 8: return ln_bwd_t<bf16, 8>(dy, x, mean, rstd, w, dx, dw, db, M, C, st, map, add, more);
      case 4: return ln_bwd_t<bf16, 4>(dy, x, mean, rstd, w, dx, dw, db, M, C, st, map, add, more);
      case 2: return ln_bwd_t<bf16, 2>(dy, x, mean, rstd, w, dx, dw, db, M, C, st, map, add, more);
      default: return ln_bwd_t<bf16, 1>(dy, x, mean, rstd, w, dx, dw, db, M, C, st, map, add, more);
    }
  }
  switch (v) {
    case 4: return ln_bwd_t<float, 4>(dy, x, mean, rstd, w, dx, dw, db, M, C, st, map, add, more);
    case 2: return ln_bwd_t<float, 2>(dy, x, mean, rstd, w, dx, dw, db, M, C, st, map, add, more);
    default: return ln_bwd_t<float, 1>(dy, x, mean, rstd, w, dx, dw, db, M, C, st, map, add, more);
  }
}

extern "C" int vmg_layernorm_bwd(int dtype, const void* dy, const void* x, const float* mean, const float* rstd, const float* w,
                                 void* dx, float* dw, float* db, int64_t M, int C, void* stream) {
  return ln_bwd_impl(dtype, dy, x, mean, rstd, w, dx, dw, db, M, C, stream, LnMap{0, 0, 0, 0}, 8);
}

extern "C" int vmg_layernorm_bwd_add(int dtype, const void* dy, const void* x, const float* mean, const float* rstd, const float* w,
                                     const void* add, void* dx, float* dw, float* db, int64_t M, int C, void* stream) {
  VMG_CHECK(!add || (uintptr_t)add % 16 == 0, "layernorm_bwd_add: add must be 16-byte aligned");
  return ln_bwd_impl(dtype, dy, x, mean, rstd, w, dx, dw, db, M, C, stream, LnMap{0, 0, 0, 0}, 8, add);
}

extern "C" int vmg_layernorm_bwd_multi(int dtype, int ndy, const void* const* dy, const void* x, const float* mean, const float* rstd, const float* w,
                                       const void* add, void* dx, float* dw, float* db, int64_t M, int C, void* stream) {
  VMG_CHECK(dy && ndy >= 1 && ndy <= 5, "layernorm_bwd_multi: 1 to 5 output gradients");
  LnMore more{{nullptr, nullptr, nullptr, nullptr}, ndy - 1};
  for (int i = 0; i < ndy; ++i) {
    VMG_CHECK(dy[i] && (uintptr_t)dy[i] % 16 == 0, "layernorm_bwd_multi: gradient %d is null or not 16-byte aligned", i);
    if (i) more.p[i - 1] = dy[i];
  }
  VMG_CHECK(!add || (uintptr_t)add % 16 == 0, "layernorm_bwd_multi: add must be 16-byte aligned");
  return ln_bwd_impl(dtype, dy[0], x, mean, rstd, w, dx, dw, db, M, C, stream, LnMap{0, 0, 0, 0}, 8, add, more);
}

extern "C" int vmg_space_depth_ln_bwd(int dtype, int mode, const void* dy, const void* x, const float* mean, const float* rstd, const float* w,
                                      void* dx, float* dw, float* db, int N, int H, int W, int cseg, void* stream) {
  int64_t M;
  int C;
  if (ln_map_check(mode, N, H, W, cseg, &M, &C)) return -1;
  return ln_bwd_impl(dtype, dy, x, mean, rstd, w, dx, dw, db, M, C, stream, LnMap{mode, H, W, cseg}, ln_map_vmax(dtype, cseg));
}
