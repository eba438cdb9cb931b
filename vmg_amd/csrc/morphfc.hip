// H- / W-branch of the MorphFC token mixer (reference: Enhanced_MorphFCs_decay.forward, models/function.py:763-786) with the token
// reshuffle folded into the GEMM's operand addressing: no token tensor is materialised on either side.
//
// The reference pads C -> Cp = chunk * S and the mixed axis to a multiple of `chunk`, takes `chunk` consecutive positions of the axis
// as a GROUP and turns it into `chunk` tokens: token (group, k) has features f = p*S + s <- x[position p of the group][channel k*S + s]
// (a transpose of (p, k) with S-channel vectors as elements -- S = 18 or 9 channels: neither 16-byte nor 4-byte granular), applies
// Linear(Cp, Cp) (+ ReLU, / Cp) and reshuffles back.  Here a wave owns a TILE of 16 tokens = 16 / chunk groups:
//   1. the groups' pixels (chunk pixels x C channels each, contiguous rows of the channels-last feature map) are copied into a
//      wave-private LDS block as they lie in HBM, 16 bytes per lane (zero rows / channels for the padding); in the data-gradient
//      form the ReLU mask of the forward output and the 1/Cp scale are applied on the way in;
//   2. the MFMA operand fragments (8 consecutive features of a token) are gathered from that block element by element
//      (ds_read_u16: a fragment crosses positions when S is not a multiple of 8);
//   3. weights (packed like a 1x1 convolution) are staged ONCE per workgroup in LDS and shared by its 8 waves;
//   4. bias / ReLU / scale, then the results are scattered element-wise into a second LDS block in PIXEL layout and leave as
//      whole 16-byte vectors of the output feature map (cropped to the real positions / channels).
// HBM-bound: one read of x and one write of the branch output, 2*N*C*2 bytes; the gather / scatter is LDS traffic.
#include "common.h"

namespace {

constexpr int MF_WAVES = 8;
constexpr int MF_MAXK = 5;  // 16-byte vectors of ONE group's pixel block per lane: chunk * C/8 <= 320
typedef __attribute__((ext_vector_type(4))) unsigned int mf_u32x4;  // (a native vector: the HIP uint4 struct behind a pointer select goes through scratch)

__device__ __attribute__((aligned(16))) unsigned int g_mf_trash[4];  // where the stores of lanes outside the image go

struct MorphK {
  const bf16* x;      // (BT, H, W, C)
  const bf16* mask;   // data-gradient form: the forward output h (same layout); x is multiplied by (mask > 0) * in_scale
  bf16* out;          // (BT, H, W, C)
  bf16* tok_out;      // null, or (ntiles * 16, Cp): the token matrix the GEMM multiplies (after the mask / scale), for the weight gradient
  const char* wpack;  // vmg_conv_pack image, ks = 1, one source of Cp channels, NCT tiles
  const float* bias;  // (Cp) or null
  int BT, H, W, C, Cp, chunk, S, axis;  // axis 0: groups along H, 1: along W
  int gpl;            // groups per line (= ceil(axis length / chunk))
  long long ngroups;  // BT * lines * gpl
  long long ntiles;
  int relu;
  float in_scale, out_scale;
  int ss;             // stage stride of the pack (bytes)
};

// EVEN: S is even -- a lane's features come in pairs that never straddle a position, so the gather reads and the scatter writes 32 bits
// at a time (half the LDS instructions and address arithmetic of the element-wise form; S = 18 in VMG-REDS-few_levels).
// MASK: the data-gradient form (a.mask given) -- a template parameter: loads behind a run-time branch, even a wave-uniform one, are each
// followed by s_waitcnt vmcnt(0).
template <int NK, int NCT, bool EVEN, bool MASK>
__global__ __launch_bounds__(MF_WAVES * 64, 1) void morph_linear_kernel(const MorphK a) {
  constexpr int COB = NCT * 16, KSB = 4 * COB * 16;  // bytes of one k-step of the pack
  constexpr int NST = (NK + 1) / 2;                  // stages (two k-steps each) of the pack
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rowb = a.Cp * 2;                 // bytes of a pixel row in the LDS blocks
  const int blkb = (16 * rowb + 15) & ~15;
  char* wl = smem;                                    // [NST][ss]
  float* lbias = reinterpret_cast<float*>(smem + NST * a.ss);  // [COB]
  char* xblk = smem + NST * a.ss + COB * 4 + wave * (2 * blkb + 16);  // [16 pixels][Cp] bf16
  char* oblk = xblk + blkb;
  // the wave's 16-byte tail: a ZERO word the gather reads for features >= Cp, and a TRASH word the scatter writes for them
  const int zoff = 2 * blkb, toff = 2 * blkb + 4;
  if (lane == 0) *reinterpret_cast<unsigned int*>(xblk + zoff) = 0u;
  // weights and bias -> LDS, once
  for (int i = tid * 16; i < NST * a.ss; i += MF_WAVES * 64 * 16) *reinterpret_cast<uint4*>(wl + i) = *reinterpret_cast<const uint4*>(a.wpack + i);
  for (int i = tid; i < COB; i += MF_WAVES * 64) lbias[i] = (a.bias && i < a.Cp) ? a.bias[i] : 0.f;
  __syncthreads();

  const int tok = lane & 15, kq = lane >> 4;
  const int ch = a.chunk, S = a.S;
  const int grp = tok / ch, kk = tok - grp * ch;  // this lane's token: group `grp` of the tile, channel chunk kk
  // LDS offsets (from xblk) of the lane's gather reads and scatter writes: they depend on the lane only, not on the tile.  Computed per
  // tile -- position / channel by division and carry, a branch per element for the padding -- the tile body was ~3 000 instructions for
  // 45 MFMAs and the kernel issue-bound at 2 TB/s (32 us per branch at N = 114 688); with the tables it is a few hundred.
  constexpr int GE = EVEN ? 4 : 8;   // gather reads per k-step (32-bit pairs / single elements)
  constexpr int SE = EVEN ? 2 : 4;   // scatter writes per output tile
  int goff[NK][GE], soff[NCT][SE];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    const int f0 = 32 * ks + 8 * kq;
#pragma unroll
    for (int j = 0; j < GE; ++j) {
      const int f = f0 + (EVEN ? 2 * j : j);
      const int p = f / S, sc = f - p * S;
      goff[ks][j] = p < ch ? (grp * ch + p) * rowb + (kk * S + sc) * 2 : zoff;  // features >= Cp multiply zero weights
    }
  }
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
    for (int j = 0; j < SE; ++j) {
      const int f = ct * 16 + kq * 4 + (EVEN ? 2 * j : j);
      const int p = f / S, sc = f - p * S;
      soff[ct][j] = f < a.Cp ? blkb + (grp * ch + p) * rowb + (kk * S + sc) * 2 : toff;  // (Cp even in the EVEN form: a pair is inside or outside together)
    }
  }
  const int vpp = a.C >> 3;      // 16-byte vectors per pixel (C % 8 == 0)
  const int G = 16 / ch;         // groups per tile (1 or 2: the host admits chunk 8 and 16 here)
  const int lines_len = a.axis == 0 ? a.H : a.W, lines = a.axis == 0 ? a.W : a.H;
  const int pos_stride = a.axis == 0 ? a.W : 1;  // pixels between consecutive positions of a group
  // per-lane constants of a group's copy: vector L = 64 k + lane of the [chunk pixels][vpp vectors] list (the same for every tile): its position
  // in the group, its byte offset from the group's first pixel in the feature map and in the LDS block.  A group's address is then one
  // wave-uniform 64-bit base + a 32-bit lane offset (the per-vector 64-bit products cost ~35 instructions per load before)
  int lp[MF_MAXK], xoff[MF_MAXK], lxo[MF_MAXK];
#pragma unroll
  for (int k = 0; k < MF_MAXK; ++k) {
    const int L = 64 * k + lane;
    const int p = (int)(((float)L + 0.5f) * (1.0f / (float)vpp));  // exact: L < 2^16
    const int v = L - p * vpp;
    lp[k] = L < ch * vpp ? p : -1;
    xoff[k] = lp[k] >= 0 ? (p * pos_stride * a.C + v * 8) * 2 : 0;
    lxo[k] = p * rowb + v * 16;
  }

  // ---- 1. pixel block -> LDS (zero-filled padding).  Group coordinates are wave-uniform; every load is UNCONDITIONAL (a lane outside the
  // image reads the group's first vector and zeroes it in registers): loads behind a per-lane branch are each followed by s_waitcnt
  // vmcnt(0), which made a tile cost ten serialised memory latencies (46 us per launch at N = 114 688).  (Tried: the loads of tile t+1
  // issued before tile t is multiplied, registers carried around the loop -- 29.8 us against 25.5: hipcc's waits at the loop top cover
  // the tile's stores as well.)  The stores are unconditional too: a lane outside the image writes a trash vector.
  struct Geo {
    long long gbyte[2];  // byte offset of the group's first pixel (0 for a group past the end)
    bool glive[2];
    int gpos0[2];
  };
  auto geometry = [&](long long tile, Geo& ge) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const unsigned gg = (unsigned)(tile * G + g);  // (the host checks ngroups < 2^31)
      const unsigned gi = gg % (unsigned)a.gpl, r = gg / (unsigned)a.gpl;
      const unsigned line = r % (unsigned)lines, bt = r / (unsigned)lines;
      ge.gpos0[g] = (int)gi * ch;
      const long long first = a.axis == 0 ? ((long long)bt * a.H + ge.gpos0[g]) * a.W + line : ((long long)bt * a.H + line) * a.W + ge.gpos0[g];
      ge.glive[g] = g < G && tile < a.ntiles && tile * G + g < a.ngroups;
      ge.gbyte[g] = ge.glive[g] ? first * a.C * 2 : 0;
    }
  };
  auto issue = [&](const Geo& ge, mf_u32x4 (&vx)[2][MF_MAXK], mf_u32x4 (&vm)[2][MF_MAXK]) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const char* gx = reinterpret_cast<const char*>(a.x) + ge.gbyte[g];
      const char* gm = reinterpret_cast<const char*>(a.mask) + ge.gbyte[g];
#pragma unroll
      for (int k = 0; k < MF_MAXK; ++k) {
        const bool okk = lp[k] >= 0 && ge.glive[g] && ge.gpos0[g] + lp[k] < lines_len;
        const int o = okk ? xoff[k] : 0;
        vx[g][k] = *reinterpret_cast<const mf_u32x4*>(gx + o);
        if constexpr (MASK) vm[g][k] = *reinterpret_cast<const mf_u32x4*>(gm + o);
      }
    }
  };
  for (long long tile = (long long)blockIdx.x * MF_WAVES + wave; tile < a.ntiles; tile += (long long)gridDim.x * MF_WAVES) {
    Geo gc;
    mf_u32x4 rx[2][MF_MAXK], rm[2][MF_MAXK];
    bool ok[2][MF_MAXK];
    geometry(tile, gc);
    issue(gc, rx, rm);
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int k = 0; k < MF_MAXK; ++k) ok[g][k] = lp[k] >= 0 && gc.glive[g] && gc.gpos0[g] + lp[k] < lines_len;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int k = 0; k < MF_MAXK; ++k) {
        mf_u32x4 val = ok[g][k] ? rx[g][k] : mf_u32x4{0u, 0u, 0u, 0u};
        if constexpr (MASK) {
          bf16x8 e = __builtin_bit_cast(bf16x8, val);
          const bf16x8 m = __builtin_bit_cast(bf16x8, rm[g][k]);
#pragma unroll
          for (int j = 0; j < 8; ++j) e[j] = (float)m[j] > 0.f ? (bf16)((float)e[j] * a.in_scale) : (bf16)0.f;
          val = __builtin_bit_cast(mf_u32x4, e);
        }
        if (lp[k] >= 0 && g < G) *reinterpret_cast<mf_u32x4*>(xblk + g * ch * rowb + lxo[k]) = val;
      }
    }
    if (a.Cp > a.C) {  // padded channels read as zero
      const int padc = a.Cp - a.C;
      for (int idx = lane; idx < 16 * padc; idx += 64) {
        const int pixl = idx / padc, c = a.C + idx - pixl * padc;
        *reinterpret_cast<bf16*>(xblk + pixl * rowb + c * 2) = (bf16)0.f;
      }
    }
    // (wave-private block: program order + hipcc's lgkmcnt waits are the only synchronisation needed)

    // ---- 2./3. gather the token fragments, multiply
    f32x4 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      bf16x8 tf;
      if constexpr (EVEN) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
        u32x4_t tw;
#pragma unroll
        for (int j = 0; j < 4; ++j) tw[j] = *reinterpret_cast<const unsigned int*>(xblk + goff[ks][j]);
        tf = __builtin_bit_cast(bf16x8, tw);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) tf[j] = *reinterpret_cast<const bf16*>(xblk + goff[ks][j]);
      }
      // the lane's fragment = features 32 ks + 8 kq .. + 7 of token (tile, tok): one 16-byte vector of the token matrix
      {  // (unconditional store, a trash vector for the lanes / launches without it: see above)
        bf16* tp = (a.tok_out && 32 * ks + 8 * kq < a.Cp) ? a.tok_out + (tile * 16 + tok) * a.Cp + 32 * ks + 8 * kq : reinterpret_cast<bf16*>(g_mf_trash);
        *reinterpret_cast<bf16x8*>(tp) = tf;
      }
      const char* wk = wl + (ks >> 1) * a.ss + (ks & 1) * KSB + kq * (COB * 16) + tok * 16;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wk + ct * 256);
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, tf, acc[ct], 0, 0, 0);
      }
    }
    // ---- 4. epilogue: lane holds output features f' = ct*16 + kq*4 + r of its token -> pixel (grp, p' = f'/S), channel kk*S + f'%S
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(lbias + ct * 16 + kq * 4);
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[ct][r] + bv[r];
        if (a.relu) v[r] = fmaxf(v[r], 0.f);
        v[r] *= a.out_scale;
      }
      if constexpr (EVEN) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x2_t pr = {(bf16)v[2 * j], (bf16)v[2 * j + 1]};
          *reinterpret_cast<bf16x2_t*>(xblk + soff[ct][j]) = pr;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<bf16*>(xblk + soff[ct][r]) = (bf16)v[r];
      }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      char* go = reinterpret_cast<char*>(a.out) + gc.gbyte[g];
#pragma unroll
      for (int k = 0; k < MF_MAXK; ++k) {
        char* dst = ok[g][k] ? go + xoff[k] : reinterpret_cast<char*>(g_mf_trash);
        *reinterpret_cast<mf_u32x4*>(dst) = *reinterpret_cast<const mf_u32x4*>(oblk + (lp[k] >= 0 ? g * ch * rowb + lxo[k] : 0));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// General path (any chunk / Cp, bf16 and fp32): the token matrix IS materialised, but by one gather kernel (and turned back by one scatter
// kernel) instead of torch's pad + transpose + reshape + permute + contiguous chain -- the full configuration's stages 1..3 (chunk 16 / 12 /
// 8 with Cp = 224 / 228 / 448, models/function.py:743-805) run 7 x 32^2 .. 7 x 8^2 pixels: launch count, not bandwidth, is what they cost.
// Token (group, k), feature f = p*S + s  <->  pixel (position p of the group), channel k*S + s; S = Cp / chunk.  The two kernels are each
// other's adjoint (every feature-map element appears in exactly one token feature; padding positions / channels read as zero and are
// dropped on the way back), so each is also the other's backward.
struct TokGeo {
  int BT, H, W, C, Cp, ld, chunk, S, axis, gpl;  // ld: row length of the token matrix in elements (>= Cp: zero-filled up to a multiple of 8)
  long long ngroups;
};
__device__ __forceinline__ long long tok_pixel(const TokGeo& g, long long group, int p, bool& inside) {
  const unsigned gi = (unsigned)(group % g.gpl);
  const long long r = group / g.gpl;
  const int lines = g.axis == 0 ? g.W : g.H;
  const int line = (int)(r % lines);
  const long long bt = r / lines;
  const int pos = (int)gi * g.chunk + p;
  inside = pos < (g.axis == 0 ? g.H : g.W);
  return g.axis == 0 ? (bt * g.H + pos) * g.W + line : (bt * g.H + line) * g.W + pos;
}

template <typename T>
__global__ __launch_bounds__(256) void morph_gather_kernel(const T* __restrict__ x, T* __restrict__ tok, const TokGeo g) {
  const long long total = g.ngroups * g.chunk * g.ld;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int f = (int)(i % g.ld);
    const long long row = i / g.ld;
    const int k = (int)(row % g.chunk);
    const long long group = row / g.chunk;
    T v = from_f32<T>(0.f);
    if (f < g.Cp) {
      const int p = f / g.S, c = k * g.S + (f - p * g.S);
      bool inside;
      const long long pix = tok_pixel(g, group, p, inside);
      if (inside && c < g.C) v = x[pix * g.C + c];
    }
    tok[i] = v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void morph_scatter_kernel(const T* __restrict__ tok, T* __restrict__ out, const TokGeo g) {
  const long long total = (long long)g.BT * g.H * g.W * g.C;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % g.C);
    const long long pix = i / g.C;
    const int xw = (int)(pix % g.W);
    const long long r = pix / g.W;
    const int yh = (int)(r % g.H);
    const long long bt = r / g.H;
    const int pos = g.axis == 0 ? yh : xw, line = g.axis == 0 ? xw : yh, lines = g.axis == 0 ? g.W : g.H;
    const int gi = pos / g.chunk, p = pos - gi * g.chunk;
    const int k = c / g.S, sc = c - k * g.S;
    const long long group = (bt * lines + line) * g.gpl + gi;
    out[i] = tok[(group * g.chunk + k) * g.ld + p * g.S + sc];
  }
}

// ---- the same two maps through an LDS tile, ONE WORKGROUP PER GROUP (chunk pixels of a line x all channels).  The element-wise kernels above read x with runs of S
// elements (S = Cp / chunk: 2 or 3 at 256 x 448, where the chunks are 64 / 112 pixels long) and pay six 64-bit divisions per element: 1.2 TB/s, 36 ms of the 319 ms
// Vimeo-size step.  Here the pixel rows come in as whole 16-byte vectors (a pixel's channels are contiguous), the permutation f = p * S + s <-> channel k * S + s
// happens between LDS and registers, and the token rows leave as 16-byte vectors; the 64-bit arithmetic is per group.
template <typename T>
__device__ __forceinline__ void tok_group_origin(const TokGeo& g, long long group, int& gi, long long& base, long long& step) {
  gi = (int)(group % g.gpl);
  const long long r = group / g.gpl;
  const int lines = g.axis == 0 ? g.W : g.H;
  const int line = (int)(r % lines);
  const long long bt = r / lines;
  // element offset of position `pos` of the line: base + pos * step
  base = g.axis == 0 ? (bt * g.H * g.W + line) * (long long)g.C : ((bt * g.H + line) * (long long)g.W) * g.C;
  step = g.axis == 0 ? (long long)g.W * g.C : (long long)g.C;
}

template <typename T>
__global__ __launch_bounds__(256) void morph_gather_lds_kernel(const T* __restrict__ x, T* __restrict__ tok, const TokGeo g) {
  constexpr int VN = 16 / (int)sizeof(T);
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  extern __shared__ __attribute__((aligned(16))) char tile_raw[];
  T* tile = reinterpret_cast<T*>(tile_raw);  // [chunk][ld] (row stride ld: a multiple of the vector): channels >= C and positions past the line are zero
  const int extent = g.axis == 0 ? g.H : g.W;
  const int cvec = g.C / VN, lvec = g.ld / VN;  // (the host admits C a multiple of VN here; ld is one by construction)
  for (long long group = blockIdx.x; group < g.ngroups; group += gridDim.x) {
    int gi;
    long long base, step;
    tok_group_origin<T>(g, group, gi, base, step);
    for (int i = threadIdx.x; i < g.chunk * lvec; i += 256) {
      const int p = i / lvec, v = i - p * lvec;
      const int pos = gi * g.chunk + p;
      u32x4 val = {0u, 0u, 0u, 0u};
      if (v < cvec && pos < extent) val = *reinterpret_cast<const u32x4*>(x + base + pos * step + v * VN);
      *reinterpret_cast<u32x4*>(tile + p * g.ld + v * VN) = val;
    }
    __syncthreads();
    T* rows = tok + group * g.chunk * (long long)g.ld;
    for (int i = threadIdx.x; i < g.chunk * lvec; i += 256) {
      const int k = i / lvec, v = i - k * lvec;
      int f = v * VN, p = f / g.S, sc = f - p * g.S;
      alignas(16) T o[VN];
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        o[e] = f < g.Cp ? tile[p * g.ld + k * g.S + sc] : from_f32<T>(0.f);
        ++f;
        if (++sc == g.S) { sc = 0; ++p; }
      }
      *reinterpret_cast<u32x4*>(rows + (long long)k * g.ld + v * VN) = *reinterpret_cast<const u32x4*>(o);
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void morph_scatter_lds_kernel(const T* __restrict__ tok, T* __restrict__ out, const TokGeo g) {
  constexpr int VN = 16 / (int)sizeof(T);
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  extern __shared__ __attribute__((aligned(16))) char tile_raw[];
  T* tile = reinterpret_cast<T*>(tile_raw);  // [chunk rows k][ld]
  const int extent = g.axis == 0 ? g.H : g.W;
  const int cvec = g.C / VN, lvec = g.ld / VN;
  for (long long group = blockIdx.x; group < g.ngroups; group += gridDim.x) {
    int gi;
    long long base, step;
    tok_group_origin<T>(g, group, gi, base, step);
    const T* rows = tok + group * g.chunk * (long long)g.ld;
    for (int i = threadIdx.x; i < g.chunk * lvec; i += 256)
      *reinterpret_cast<u32x4*>(tile + i * VN) = *reinterpret_cast<const u32x4*>(rows + (long long)i * VN);  // (the group's rows are contiguous)
    __syncthreads();
    for (int i = threadIdx.x; i < g.chunk * cvec; i += 256) {
      const int p = i / cvec, v = i - p * cvec;
      const int pos = gi * g.chunk + p;
      if (pos >= extent) continue;
      int c = v * VN, k = c / g.S, sc = c - k * g.S;
      alignas(16) T o[VN];
#pragma unroll
      for (int e = 0; e < VN; ++e) {
        o[e] = tile[k * g.ld + p * g.S + sc];
        if (++sc == g.S) { sc = 0; ++k; }
      }
      *reinterpret_cast<u32x4*>(out + base + pos * step + v * VN) = *reinterpret_cast<const u32x4*>(o);
    }
    __syncthreads();
  }
}

}  // namespace

static int tok_geo(TokGeo& g, int axis, int chunk, int BT, int H, int W, int C, int Cp, int ld) {
  VMG_CHECK(axis == 0 || axis == 1, "morph tokens: axis 0 (H) or 1 (W)");
  VMG_CHECK(BT > 0 && H > 0 && W > 0 && C > 0 && chunk > 0 && Cp >= C && Cp % chunk == 0 && ld >= Cp, "morph tokens: bad geometry (Cp >= C, chunk | Cp, ld >= Cp)");
  g.BT = BT; g.H = H; g.W = W; g.C = C; g.Cp = Cp; g.ld = ld; g.chunk = chunk; g.S = Cp / chunk; g.axis = axis;
  g.gpl = cdiv(axis == 0 ? H : W, chunk);
  g.ngroups = (long long)BT * (axis == 0 ? W : H) * g.gpl;
  return 0;
}

extern "C" int64_t vmg_morph_token_rows(int axis, int chunk, int BT, int H, int W) {
  if (chunk <= 0 || BT <= 0 || H <= 0 || W <= 0 || (axis != 0 && axis != 1)) return -1;
  return (int64_t)BT * (axis == 0 ? W : H) * cdiv(axis == 0 ? H : W, chunk) * chunk;
}

extern "C" int vmg_morph_tokens_gather(int dtype, int axis, int chunk, const void* x, void* tok, int BT, int H, int W, int C, int Cp, int ld, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "morph_tokens_gather: bad dtype");
  VMG_CHECK(x && tok, "morph_tokens_gather: null pointer");
  TokGeo g;
  if (int rc = tok_geo(g, axis, chunk, BT, H, W, C, Cp, ld)) return rc;
  const int es = dtype == VMG_BF16 ? 2 : 4, vn = 16 / es;
  const long long lds = (long long)chunk * ld * es;
  if (C % vn == 0 && ld % vn == 0 && lds <= 64 * 1024 && ((uintptr_t)x | (uintptr_t)tok) % 16 == 0) {  // the LDS-tiled form (one workgroup per group)
    const int gb = (int)(g.ngroups > 16384 ? 16384 : g.ngroups);
    if (dtype == VMG_BF16) hipLaunchKernelGGL(morph_gather_lds_kernel<bf16>, dim3(gb), dim3(256), (size_t)lds, (hipStream_t)stream, (const bf16*)x, (bf16*)tok, g);
    else hipLaunchKernelGGL(morph_gather_lds_kernel<float>, dim3(gb), dim3(256), (size_t)lds, (hipStream_t)stream, (const float*)x, (float*)tok, g);
    VMG_LAUNCH_CHECK();
    return 0;
  }
  const long long total = g.ngroups * chunk * ld;
  const int blocks = (int)(cdiv64(total, 256) > 8192 ? 8192 : cdiv64(total, 256));
  if (dtype == VMG_BF16) hipLaunchKernelGGL(morph_gather_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)tok, g);
  else hipLaunchKernelGGL(morph_gather_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)tok, g);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_morph_tokens_scatter(int dtype, int axis, int chunk, const void* tok, void* out, int BT, int H, int W, int C, int Cp, int ld, void* stream) {
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "morph_tokens_scatter: bad dtype");
  VMG_CHECK(tok && out, "morph_tokens_scatter: null pointer");
  TokGeo g;
  if (int rc = tok_geo(g, axis, chunk, BT, H, W, C, Cp, ld)) return rc;
  const int es = dtype == VMG_BF16 ? 2 : 4, vn = 16 / es;
  const long long lds = (long long)chunk * ld * es;
  if (C % vn == 0 && ld % vn == 0 && lds <= 64 * 1024 && ((uintptr_t)out | (uintptr_t)tok) % 16 == 0) {  // the LDS-tiled form (one workgroup per group)
    const int gb = (int)(g.ngroups > 16384 ? 16384 : g.ngroups);
    if (dtype == VMG_BF16) hipLaunchKernelGGL(morph_scatter_lds_kernel<bf16>, dim3(gb), dim3(256), (size_t)lds, (hipStream_t)stream, (const bf16*)tok, (bf16*)out, g);
    else hipLaunchKernelGGL(morph_scatter_lds_kernel<float>, dim3(gb), dim3(256), (size_t)lds, (hipStream_t)stream, (const float*)tok, (float*)out, g);
    VMG_LAUNCH_CHECK();
    return 0;
  }
  const long long total = (long long)BT * H * W * C;
  const int blocks = (int)(cdiv64(total, 256) > 8192 ? 8192 : cdiv64(total, 256));
  if (dtype == VMG_BF16) hipLaunchKernelGGL(morph_scatter_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)tok, (bf16*)out, g);
  else hipLaunchKernelGGL(morph_scatter_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)tok, (float*)out, g);
  VMG_LAUNCH_CHECK();
  return 0;
}

// the stage stride vmg_conv_pack uses for ks = 1 (conv_igemm.hip: stage_stride(1, ntb, 16))
static int morph_stage_stride(int nct) { return (2 * 4 * nct * 16 * 16 + 4095) / 4096 * 4096; }

extern "C" int64_t vmg_morphfc_token_rows(int axis, int chunk, int BT, int H, int W) {
  if (chunk <= 0 || 16 % chunk || BT <= 0 || H <= 0 || W <= 0) return -1;
  const int len = axis == 0 ? H : W, lines = axis == 0 ? W : H;
  const long long ngroups = (long long)BT * lines * cdiv(len, chunk);
  return cdiv64(ngroups, 16 / chunk) * 16;
}

extern "C" int vmg_morphfc_fwd(int axis, int chunk, const void* x, const void* relu_mask, const void* packed, const float* bias, void* out, void* tok_out,
                               int BT, int H, int W, int C, int Cp, int cout_tiles, int relu, float in_scale, float out_scale, void* stream) {
  VMG_CHECK(x && packed && out && BT > 0 && H > 0 && W > 0, "morphfc: bad arguments");
  VMG_CHECK(axis == 0 || axis == 1, "morphfc: axis 0 (H) or 1 (W)");
  VMG_CHECK((chunk == 8 || chunk == 16) && Cp % chunk == 0 && Cp >= C && C % 8 == 0 && chunk * (C / 8) <= 64 * MF_MAXK,
            "morphfc: chunk must be 8 or 16 and divide Cp; C a multiple of 8, at most %d", 64 * MF_MAXK * 8 / 16);
  VMG_CHECK(((uintptr_t)x | (uintptr_t)out | (uintptr_t)packed | (uintptr_t)relu_mask | (uintptr_t)tok_out) % 16 == 0, "morphfc: pointers must be 16-byte aligned");
  VMG_CHECK(!tok_out || Cp % 8 == 0, "morphfc: the token output needs Cp to be a multiple of 8");
  const int nk = (Cp + 31) / 32, nct = (Cp + 15) / 16;
  VMG_CHECK(cout_tiles == nct, "morphfc: the pack must hold all %d output tiles in one block (cout_tiles = %d given)", nct, cout_tiles);
  MorphK k;
  memset(&k, 0, sizeof(k));
  k.x = (const bf16*)x; k.mask = (const bf16*)relu_mask; k.out = (bf16*)out; k.tok_out = (bf16*)tok_out; k.wpack = (const char*)packed; k.bias = bias;
  k.BT = BT; k.H = H; k.W = W; k.C = C; k.Cp = Cp; k.chunk = chunk; k.S = Cp / chunk; k.axis = axis;
  const int len = axis == 0 ? H : W, lines = axis == 0 ? W : H;
  k.gpl = cdiv(len, chunk);
  k.ngroups = (long long)BT * lines * k.gpl;
  VMG_CHECK(k.ngroups < (1LL << 31) - 2, "morphfc: too many groups");
  const int G = 16 / chunk;
  k.ntiles = cdiv64(k.ngroups, G);
  k.relu = relu; k.in_scale = in_scale; k.out_scale = out_scale;
  k.ss = morph_stage_stride(nct);
  const int blkb = (16 * Cp * 2 + 15) & ~15;
  const int lds = ((nk + 1) / 2) * k.ss + nct * 16 * 4 + MF_WAVES * (2 * blkb + 16);
  VMG_CHECK(lds <= 160 * 1024, "morphfc: Cp = %d needs %d B of LDS (> 160 KiB): use the unfused path", Cp, lds);
  hipStream_t st = (hipStream_t)stream;
  const int ncu = vmg_cu_count(vmg_current_device());
  long long nwg = cdiv64(k.ntiles, MF_WAVES);
  if (nwg > ncu) nwg = ncu;  // one workgroup per CU (the LDS holds the weights): tiles are strided over the waves
#define MF_CASE(NK_, NCT_) MF_CASE2(NK_, NCT_, true, true) MF_CASE2(NK_, NCT_, false, true) MF_CASE2(NK_, NCT_, true, false) MF_CASE2(NK_, NCT_, false, false)
#define MF_CASE2(NK_, NCT_, EV_, MK_)                                                                               \
  if (nk == NK_ && nct == NCT_ && (k.S % 2 == 0) == EV_ && (relu_mask != nullptr) == MK_) {                         \
    auto fn = morph_linear_kernel<NK_, NCT_, EV_, MK_>;                                                             \
    static bool attr_set[VMG_MAX_DEVICES] = {};                                                                     \
    const int dev = vmg_current_device();                                                                           \
    if (!attr_set[dev]) {                                                                                           \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr_set[dev] = true;                                                                                         \
    }                                                                                                               \
    hipLaunchKernelGGL(fn, dim3((unsigned)nwg), dim3(MF_WAVES * 64), lds, st, k);                                   \
    VMG_LAUNCH_CHECK();                                                                                             \
    return 0;                                                                                                       \
  }
  MF_CASE(5, 9)   // Cp = 144
  MF_CASE(4, 7)   // Cp = 112
  MF_CASE(1, 1)   // Cp = 16 (test configurations)
  MF_CASE(1, 2)   // Cp = 32
  MF_CASE(2, 4)   // Cp = 64
#undef MF_CASE
#undef MF_CASE2
  vmg_set_error("morphfc: Cp = %d is not instantiated (144, 112, 64, 32, 16)", Cp);
  return -1;
}
