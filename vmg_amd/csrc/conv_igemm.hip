// Stride-1 "same" convolution (KS = 3 or 1; KS = 1 is every Linear on the path) as an MFMA implicit GEMM
// for gfx950, channels-last activations, with a fused epilogue.  See include/vmg_hip.h for the contract.
//
// Two kernels share the packed-weight layout, the halo copy (stage_halo) and the epilogues:
//   * conv_igemm_kernel  -- the four waves split the PIXELS of the tile; weights stream through an LDS ring.  Every shape,
//     bf16 and fp32 (the parity mode); described right below.
//   * conv_ksplit_kernel -- the four waves split the K LOOP, weights go global -> register; bf16, <= 5 cout tiles per
//     workgroup: the 144- / 288-channel 3x3 convs that dominate the path.  Described at its definition.
//
// conv_igemm_kernel (one workgroup = 4 waves = 256 threads):
//   * spatial tile of TH = 4*MT rows x 16 columns (KS = 3) or 64*MT consecutive rows of the (M, C) matrix
//     (KS = 1); wave w owns MT 16-pixel rows of it; blockIdx.y selects a block of NTB*16 output channels.
//   * K order = for each source tensor (a channel block of <= 160 channels; virtual concat = several sources):
//       for 32-channel block cbk: for tap (ky, kx).  One k-step = (cbk, tap) = 32 K values: lane group g of the MFMA
//       operand holds the 8-channel chunk 4*cbk + g.  The tap -- hence the LDS offset of the activation operand -- is
//       WAVE-UNIFORM per k-step (with one wave per SIMD every per-lane VALU instruction costs >= 4 cycles, and the
//       per-lane (tap, chunk) bookkeeping of a (tap, chunk)-ordered K cost as much as the MFMAs it fed).
//   * a STAGE = one (cbk, ky) row of 3 taps (3x3) or 2 channel blocks (1x1): a fully unrolled block in which the LDS
//     fragment reads of k-step j+1 are issued before the MFMAs of k-step j (two static register sets).
//   * the source's halo tile ((TH+2) x 18 pixels, all channels of the block) is staged once in LDS as one linear array of
//     16-byte vectors copied by LDS-DMA; the activation operand of every k-step is a plain 16-byte LDS read at
//     (pixel + tap offset).
//   * packed weights stream through an LDS ring of stages filled by global_load_lds (16 B/lane, 1 KiB per
//     instruction).  Stage images are padded to a multiple of 4 KiB so every wave issues the same number IPW of
//     instructions and "stage s has landed" is a COUNTED s_waitcnt vmcnt((RING-2)*IPW): with 3 slots the younger
//     stage stays in flight across the barrier.  hipcc does NOT drain LDS-DMA at __syncthreads(): the wait is explicit.
//   * MFMA: D[cout 16][pixel 16] += W[cout][k] * X[k][pixel]  (v_mfma_f32_16x16x32_bf16, or 8 x
//     v_mfma_f32_16x16x4_f32 per k-step for fp32), so a lane ends with 4 consecutive output channels of one
//     pixel -> 8-byte (bf16) / 16-byte (fp32) channels-last stores.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr int MAX_ISRC = 32;  // internal sources after channel-block splitting
constexpr int CBMAX = 160;    // max channels per internal source block

// ---- stage geometry shared by the packer, the host launcher and the kernel
__host__ __device__ constexpr int kstg(int ks) { return ks > 1 ? ks : 2; }  // k-steps per stage (a row of taps; 1x1: two channel blocks)
__host__ __device__ constexpr int stage_bytes(int ks, int ntb, int cb) { return kstg(ks) * 4 * ntb * 16 * cb; }
__host__ __device__ constexpr int stage_stride(int ks, int ntb, int cb) { return (stage_bytes(ks, ntb, cb) + 4095) / 4096 * 4096; }
// stages of one source block of `ch` channels: 32-channel blocks x KS tap rows (KS x KS) or pairs of blocks (1x1)
__host__ __device__ inline int stages_of(int ks, int ch) {
  const int nb = (ch / 8 + 3) / 4;
  return ks > 1 ? ks * nb : (nb + 1) / 2;
}

struct ConvK {
  const char* src[MAX_ISRC];
  long long src_ps[MAX_ISRC];
  short src_ch[MAX_ISRC];
  short src_nst[MAX_ISRC];  // stages of this source
  int src_pixb[MAX_ISRC];  // LDS pixel stride in bytes
  int nsrc;
  const char* wpack;
  const float* bias;
  char* out;
  long long out_ps;
  char* out_pre;
  const char* res;
  long long res_ps;
  const char* aux;
  long long aux_ps;
  int N, H, W, Cout, tiles_x, tiles_y;
  long long M;
  int act;
  float slope, alpha;
  int actgrad, ps;
  int nstages;     // total stages over all sources
  int halo_bytes;  // LDS bytes reserved for the halo tile
  int ring;        // weight-streaming kernel: LDS ring slots (4 when they fit beside the halo tile, else 3)
  int vec8;        // 1: out / res / aux / out_pre rows are 16-byte aligned with strides % 8 == 0, Cout % 16 == 0, no PixelShuffle
  unsigned long long* stamps;  // diagnostics: per-wave s_memrealtime stamps (8 per wave) when non-null (vmg_conv_debug_stamps)
  int dbg;         // ablation bits (env VMG_CONV_DBG, diagnostics only): 1 no weight DMA, 4 no halo staging, 8 no store, 16 return at once, 32 no main loop, 128 no XCD-aware tile order
};

template <typename T>
struct Frag;  // 8 consecutive K elements of one row/column of an MFMA operand
template <>
struct Frag<bf16> {
  bf16x8 v;
  __device__ __forceinline__ void load(const char* p) { v = *reinterpret_cast<const bf16x8*>(p); }
};
template <>
struct Frag<float> {
  float4 lo, hi;
  __device__ __forceinline__ void load(const char* p) {
    lo = *reinterpret_cast<const float4*>(p);
    hi = *reinterpret_cast<const float4*>(p + 16);
  }
};
__device__ __forceinline__ f32x4 mma(const Frag<bf16>& w, const Frag<bf16>& x, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, x.v, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(const Frag<float>& w, const Frag<float>& x, f32x4 c) {
  // 8 x v_mfma_f32_16x16x4_f32: lane group g supplies element j of chunk (4*kstep + g) in MFMA j (A and B alike)
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.x, x.lo.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.y, x.lo.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.z, x.lo.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.lo.w, x.lo.w, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.x, x.hi.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.y, x.hi.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.z, x.hi.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(w.hi.w, x.hi.w, c, 0, 0, 0);
  return c;
}

__device__ uint4 g_conv_zero16;  // 16 zero bytes: the source of out-of-image lanes of the halo copy
__device__ uint4 g_wstat_trash;  // where conv_wstat_kernel's items outside the image store (their stores stay unconditional: counted waits)

// diagnostics: wave-level time stamps (100 MHz constant clock) at phase boundaries of the k-split kernel
// (compiled in only with -DVMG_DIAG: tools/conv_timeline.py, tools/conv_ablate.py; the shipped library carries neither the
// stamps nor the ablation bits)
#ifdef VMG_DIAG
#define VMG_DBG(a, bit) ((a).dbg & (bit))
__device__ __forceinline__ void conv_stamp(const ConvK& a, int wave, int slot) {
  if (a.stamps) {
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) a.stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8 + slot] = t;
  }
}
#else
#define VMG_DBG(a, bit) 0
__device__ __forceinline__ void conv_stamp(const ConvK&, int, int) {}
#endif

// Consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2.  Giving XCD j the j-th contiguous
// eighth of the tile list makes spatially neighbouring tiles (which share halo rows) and both cout blocks of a tile
// meet in the same L2 (speed only: nothing depends on the placement).  bit 128 of dbg disables it.
__device__ __forceinline__ int xcd_remap(int bid, int nblk, int dbg) {
#ifndef VMG_DIAG
  dbg = 0;
#endif
  if ((nblk & 7) != 0 || (dbg & 128)) return bid;
  return (bid & 7) * (nblk >> 3) + (bid >> 3);
}

// ---- stage the halo tile of source s in LDS: (THH rows) x (TWH pixels) x ch channels (all 256 threads; no barriers
// inside -- the caller brackets it)
// LDS-DMA as an asm statement (see conv_wgrad.hip::glds16_hidden): with the builtin, hipcc waits vmcnt(0) in front of every later ds_read of
// the issuing wave, i.e. for the whole weight ring in flight; the kernels below count their copies by hand.
__device__ __forceinline__ void glds16_asm(const char* gsrc, char* lds_dst) {
  unsigned keep;
  const unsigned ldst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(lds_dst));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(ldst) : "memory");
}

// the same with a wave-uniform 64-bit base (SGPR pair) + a 32-bit lane offset: no per-lane 64-bit address arithmetic
__device__ __forceinline__ void glds16_asm_s(const char* sbase, int voff, char* lds_dst) {
  unsigned keep;
  const unsigned ldst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(lds_dst));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldst) : "memory");
}

template <typename T, int KS, int MT>
__device__ __forceinline__ void stage_halo(const ConvK& a, int s, char* halo, int n, int ty, int tx, long long m0, int tid) {
  constexpr int ES = ElemTraits<T>::ES;
  constexpr int TH = 4 * MT;
  constexpr int TWH = 16 + KS - 1;
  constexpr int THH = TH + KS - 1;
  const int lane = tid & 63, wave = tid >> 6;
  const int ch = a.src_ch[s], pixb = a.src_pixb[s];
      const char* sp = a.src[s];
      const long long ps_b = a.src_ps[s] * ES;
      const int vpp = ch * ES / 16;  // 16-byte vectors per pixel
      const float inv_vpp = 1.0f / (float)vpp;
      if (pixb == ch * ES && !VMG_DBG(a, 4)) {
        // Dense LDS pixel stride: the tile is ONE linear array of THH*TWH*vpp 16-byte vectors, copied by LDS-DMA (no
        // registers, 1 KiB per instruction).  Wave w issues instructions w, w+4, ... -- the same count on every wave --
        // and every lane always loads: a lane whose pixel lies outside the image (or past the end of the tile: the LDS
        // region is padded to a multiple of 1 KiB) reads a 16-byte zero constant instead.  No branches, no exec masking.
        const int row_vecs = TWH * vpp, total = THH * row_vecs;
        const int L0 = wave * 64 + lane;
        int r = (int)(((float)L0 + 0.5f) * (1.0f / (float)row_vecs));  // exact: L0 < 2^16, row_vecs <= 720
        int rem = L0 - r * row_vecs;
        const int dr = 256 / row_vecs, drem = 256 - dr * row_vecs;
        // tile origin as a wave-uniform 64-bit pointer (may lie before the tensor for border tiles: only in-image lanes use
        // it); per lane a 32-bit offset -- (r*W + p)*ps_b + 16v stays far below 2^31 for r <= THH, W <= 8192
        const int y0 = (KS > 1) ? ty * TH - KS / 2 : 0, x0 = (KS > 1) ? tx * 16 - KS / 2 : 0;
        const char* origin = (KS > 1) ? sp + (((long long)n * a.H + y0) * a.W + x0) * ps_b : sp + m0 * ps_b;
        const int ps32 = (int)ps_b, rowstep = ((KS > 1) ? a.W : 16) * ps32;
        const long long mleft = a.M - m0;  // KS == 1: pixels from the tile start to the end of the matrix
        for (int i0 = wave * 64; i0 < total; i0 += 256) {
          const int p = (int)(((float)rem + 0.5f) * inv_vpp);  // exact for these ranges (rem < 2^16, vpp <= 40)
          const int v = rem - p * vpp;
          bool ok = r < THH;
          if (KS > 1) ok = ok && (unsigned)(y0 + r) < (unsigned)a.H && (unsigned)(x0 + p) < (unsigned)a.W;
          else ok = ok && (long long)(r * 16 + p) < mleft;
          const int off = r * rowstep + p * ps32 + v * 16;
          const char* gp = ok ? origin + off : reinterpret_cast<const char*>(&g_conv_zero16);
          glds16_asm(gp, halo + i0 * 16);
          rem += drem;
          r += dr;
          if (rem >= row_vecs) { rem -= row_vecs; ++r; }
        }
        conv_stamp(a, wave, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        // Padded LDS pixel stride: through registers.  Loads are UNCONDITIONAL (out-of-image lanes read the tensor's
        // first vector and select zero afterwards): a load behind a per-element branch makes hipcc wait vmcnt(0) per
        // element; this way a batch of NB loads is in flight before the first LDS write.
        const int total = VMG_DBG(a, 4) ? 0 : THH * TWH * vpp;
        constexpr int NB = 8;
        for (int i0 = tid; i0 < total; i0 += 256 * NB) {
          uint4 val[NB];
          int dsto[NB];
          bool inb[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const int i = i0 + b * 256;
            const int ic = i < total ? i : total - 1;
            const int p = (int)(((float)ic + 0.5f) * inv_vpp);
            const int v = ic - p * vpp;
            dsto[b] = i < total ? p * pixb + v * 16 : -1;
            long long goff;
            if (KS > 1) {
              const int r = p / TWH, c = p - r * TWH;
              const int y = ty * TH + r - KS / 2, x = tx * 16 + c - KS / 2;
              inb[b] = y >= 0 && y < a.H && x >= 0 && x < a.W;
              goff = (((long long)n * a.H + y) * a.W + x) * ps_b + v * 16;
            } else {
              const long long m = m0 + p;
              inb[b] = m < a.M;
              goff = m * ps_b + v * 16;
            }
            val[b] = *reinterpret_cast<const uint4*>(sp + (inb[b] ? goff : 0));
          }
#pragma unroll
          for (int b = 0; b < NB; ++b)
            if (dsto[b] >= 0) *reinterpret_cast<uint4*>(halo + dsto[b]) = inb[b] ? val[b] : make_uint4(0, 0, 0, 0);
        }
      }
}

// 4 consecutive elements kept in their storage type (operands of the epilogue fetched before the main loop)
template <typename T>
struct Raw4;
template <>
struct Raw4<bf16> {
  bf16x4 v;
  __device__ __forceinline__ void load(const bf16* p) { v = *reinterpret_cast<const bf16x4*>(p); }
  __device__ __forceinline__ void get(float o[4]) const { o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3]; }
};
template <>
struct Raw4<float> {
  float4 v;
  __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void get(float o[4]) const { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
};

// pixel of lane px in row `row` of the workgroup's tile
template <int KS>
__device__ __forceinline__ bool conv_row_pixel(const ConvK& a, int row, int TH, int n, int ty, int tx, long long m0, int px, long long& pix) {
  if (KS > 1) {
    const int y = ty * TH + row, x = tx * 16 + px;
    pix = ((long long)n * a.H + y) * a.W + x;
    return (y < a.H) && (x < a.W);
  }
  pix = m0 + row * 16 + px;
  return pix < a.M;
}
__device__ __forceinline__ bool conv_epilogue_fast(const ConvK& a) {
  const bool vec_ok = ((a.Cout & 3) == 0) && ((a.out_ps & 3) == 0) && (!a.res || (a.res_ps & 3) == 0) && (!a.aux || (a.aux_ps & 3) == 0);
  return vec_ok && ((a.Cout & 15) == 0);
}

// ---- epilogue of one 16-pixel row (row `row` of the workgroup's tile): acc[ct] holds, per lane, output channels
// cb*COB + ct*16 + 4g .. +3 of pixel px.  bias, activation, scale, activation-gradient mask, residual, PixelShuffle.
// PRE: the residual / activation-gradient operands of the fast path were fetched earlier into pre_res / pre_aux, and
// `lbias` (LDS) holds the block's COB bias values.
template <typename T, int KS, int NTB, bool PRE = false>
__device__ __forceinline__ void conv_epilogue_row(const ConvK& a, const f32x4 (&acc)[NTB], int row, int TH, int cb, int nt_real, int n, int ty,
                                                  int tx, long long m0, int px, int g, const Raw4<T>* pre_res = nullptr,
                                                  const Raw4<T>* pre_aux = nullptr, const float* lbias = nullptr) {
  constexpr int COB = NTB * 16;
  T* out = reinterpret_cast<T*>(a.out);
  T* out_pre = reinterpret_cast<T*>(a.out_pre);
  const T* res = reinterpret_cast<const T*>(a.res);
  const T* aux = reinterpret_cast<const T*>(a.aux);
  const bool vec_ok = ((a.Cout & 3) == 0) && ((a.out_ps & 3) == 0) && (!res || (a.res_ps & 3) == 0) &&
                      (!aux || (a.aux_ps & 3) == 0);
  const bool fast = vec_ok && ((a.Cout & 15) == 0);  // every real tile of this block is complete: straight-line path
  // ReLU / LeakyReLU / none as one select: t > 0 ? t : t * neg
  const float neg = a.act == VMG_ACT_RELU ? 0.f : (a.act == VMG_ACT_LRELU ? a.slope : 1.f);
  {
    long long pix;
    bool valid;
    int y = 0, x = 0;
    if (KS > 1) {
      y = ty * TH + row;
      x = tx * 16 + px;
      valid = (y < a.H) && (x < a.W);
      pix = ((long long)n * a.H + y) * a.W + x;
    } else {
      pix = m0 + row * 16 + px;
      valid = pix < a.M;
      if (a.ps) {
        x = (int)(pix % a.W);
        const long long t = pix / a.W;
        y = (int)(t % a.H);
        n = (int)(t / a.H);
      }
    }
    if (!valid || VMG_DBG(a, 8)) return;
    if (fast) {
#pragma unroll
      for (int ct = 0; ct < NTB; ++ct) {
        if (ct < nt_real) {
          const int co0 = cb * COB + ct * 16 + g * 4;
          float v[4] = {acc[ct][0], acc[ct][1], acc[ct][2], acc[ct][3]};
          if (a.bias) {
            const float4 bv = PRE ? *reinterpret_cast<const float4*>(lbias + ct * 16 + g * 4) : *reinterpret_cast<const float4*>(a.bias + co0);
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
          }
          if (out_pre) store4(out_pre + pix * a.out_ps + co0, v);
          if (a.act == VMG_ACT_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]) * a.alpha;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (v[r] > 0.f ? v[r] : v[r] * neg) * a.alpha;
          }
          if (aux) {
            float u[4];
            if (PRE) pre_aux[ct].get(u);
            else load4(aux + pix * a.aux_ps + co0, u);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float d = a.actgrad == 3 ? gelu_erf_grad(u[r]) : (u[r] > 0.f ? 1.f : (a.actgrad == 2 ? a.slope : 0.f));
              v[r] *= d;
            }
          }
          if (res) {
            float u[4];
            if (PRE) pre_res[ct].get(u);
            else load4(res + pix * a.res_ps + co0, u);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += u[r];
          }
          if (a.ps) {
            // torch PixelShuffle(2): channel co = c*4 + i*2 + j -> out[n, 2y+i, 2x+j, c]
            const int c = co0 >> 2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const long long op = ((long long)n * (2 * a.H) + (2 * y + (r >> 1))) * (2 * a.W) + (2 * x + (r & 1));
              out[op * a.out_ps + c] = from_f32<T>(v[r]);
            }
          } else {
            store4(out + pix * a.out_ps + co0, v);
          }
        }
      }
      return;
    }
    // general path (Cout not a multiple of 16, or unaligned strides): element-wise with bounds checks
    for (int ct = 0; ct < nt_real; ++ct) {
      const int co0 = cb * COB + ct * 16 + g * 4;
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + r;
        if (co >= a.Cout) continue;
        float t = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < NTB; ++c2)  // static register index
          if (c2 == ct) t = r == 0 ? acc[c2][0] : (r == 1 ? acc[c2][1] : (r == 2 ? acc[c2][2] : acc[c2][3]));
        if (a.bias) t += a.bias[co];
        if (out_pre) out_pre[pix * a.out_ps + co] = from_f32<T>(t);
        if (a.act == VMG_ACT_GELU) t = gelu_erf(t) * a.alpha;
        else t = (t > 0.f ? t : t * neg) * a.alpha;
        if (aux) {
          const float u = to_f32(aux[pix * a.aux_ps + co]);
          t *= a.actgrad == 3 ? gelu_erf_grad(u) : (u > 0.f ? 1.f : (a.actgrad == 2 ? a.slope : 0.f));
        }
        if (res) t += to_f32(res[pix * a.res_ps + co]);
        if (a.ps) {
          const long long op = ((long long)n * (2 * a.H) + (2 * y + ((co & 3) >> 1))) * (2 * a.W) + (2 * x + (co & 1));
          out[op * a.out_ps + (co >> 2)] = from_f32<T>(t);
        } else {
          out[pix * a.out_ps + co] = from_f32<T>(t);
        }
      }
    }
  }
}

template <typename T, int KS, int MT, int NTB, bool DEEP>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvK a) {
  constexpr int ES = ElemTraits<T>::ES, CB = ElemTraits<T>::CHUNKB;
  constexpr int TH = 4 * MT;
  constexpr int TWH = 16 + KS - 1;
  constexpr int THH = TH + KS - 1;
  constexpr int COB = NTB * 16;
  constexpr int KSTG = kstg(KS), RING = DEEP ? 3 : 2;
  constexpr int SS = stage_stride(KS, NTB, CB);  // stage image size in LDS and in the packed weights
  constexpr int IPW = SS / 4096;                 // global_load_lds instructions per wave per stage
  constexpr int KK = KS * KS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wbuf = smem + a.halo_bytes;
  if (VMG_DBG(a, 16)) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, g = lane >> 4;
  const int cb = blockIdx.y;
  // 16-channel tiles this block really has (the last block of a layer may be short: its MFMAs are skipped, wave-uniform)
  const int nt_real = min(NTB, (a.Cout - cb * COB + 15) >> 4);
  int n = 0, ty = 0, tx = 0;
  long long m0 = 0;
  if (KS > 1) {
    int bid = xcd_remap(blockIdx.x, gridDim.x, a.dbg);
    tx = bid % a.tiles_x;
    int r = bid / a.tiles_x;
    ty = r % a.tiles_y;
    n = r / a.tiles_y;
  } else {
    m0 = (long long)blockIdx.x * (64 * MT);
  }

  f32x4 acc[NTB][MT];
#pragma unroll
  for (int ct = 0; ct < NTB; ++ct)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[ct][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nstages = a.nstages;
  const char* wsrc = a.wpack + (long long)cb * nstages * SS + (wave * 1024 + lane * 16);
  auto issue_w = [&](int stage) {
    if (VMG_DBG(a, 1)) return;
    const char* gsrc = wsrc + (long long)stage * SS;
    char* dst = wbuf + (stage % RING) * SS + wave * 1024;
#pragma unroll
    for (int i = 0; i < IPW; ++i) glds16_asm(gsrc + i * 4096, dst + i * 4096);
  };

#pragma unroll
  for (int i = 0; i < RING - 1; ++i)
    if (i < nstages) issue_w(i);
  int gst = 0;  // global stage index
  for (int s = 0; s < a.nsrc; ++s) {
    const int ch = a.src_ch[s], pixb = a.src_pixb[s];
    const int CH = ch >> 3;  // 8-channel chunks of this source
    __syncthreads();         // everyone is done reading the previous halo tile
    const bool after_restage = s > 0;  // LDS-DMA of the halo is younger than the weight stages in flight: see the stage wait
    stage_halo<T, KS, MT>(a, s, halo, n, ty, tx, m0, tid);
    __syncthreads();

    // per-lane pixel base addresses in the halo tile
    const char* pixp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = wave * MT + mt;
      pixp[mt] = halo + ((KS > 1) ? (row * TWH + px) * pixb : (row * 16 + px) * pixb);
    }

    const int nst_s = (VMG_DBG(a, 32) ? 0 : a.src_nst[s]);
    int cbk = 0, ky = 0;  // 3x3: stage = (channel block cbk, tap row ky); 1x1: stage = channel blocks 2*sl, 2*sl+1
    for (int sl = 0; sl < nst_s; ++sl, ++gst) {
      // Stage gst must have landed, and every wave must be past stage gst-1 before its slot is refilled.
      if (RING > 2 && gst + RING - 2 < nstages && !(after_restage && sl == 0))
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * IPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (gst + RING - 1 < nstages) issue_w(gst + RING - 1);

      // activation-operand offsets of the KSTG k-steps (zero-weight padding chunks read the last real chunk)
      int boff[KSTG];
      if (KS > 1) {
        const int base = ky * (TWH * pixb) + min(4 * cbk + g, CH - 1) * CB;
#pragma unroll
        for (int j = 0; j < KSTG; ++j) boff[j] = base + j * pixb;
        if (++ky == KS) { ky = 0; ++cbk; }
      } else {
#pragma unroll
        for (int j = 0; j < KSTG; ++j) boff[j] = min(4 * (2 * sl + j) + g, CH - 1) * CB;
      }
      const char* wslot = wbuf + (gst % RING) * SS + g * (COB * CB) + px * CB;

      // ONE straight-line body: every tile of the block is computed (padding tiles of a short last block have zero
      // weights).  Two variants of this body -- e.g. a branch-free one and one that skips padding tiles -- make hipcc
      // merge the accumulators of both paths with v_accvgpr_mov shuffles around every MFMA (4 copies per MFMA measured).
      Frag<T> xf[2][MT], wf[2][NTB];
      auto load_frags = [&](int j, int set) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xf[set][mt].load(pixp[mt] + boff[j]);
#pragma unroll
        for (int ct = 0; ct < NTB; ++ct) wf[set][ct].load(wslot + j * (4 * COB * CB) + ct * 16 * CB);
      };
      load_frags(0, 0);
#pragma unroll
      for (int j = 0; j < KSTG; ++j) {
        if (j + 1 < KSTG) load_frags(j + 1, (j + 1) & 1);
#pragma unroll
        for (int ct = 0; ct < NTB; ++ct)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[ct][mt] = mma(wf[j & 1][ct], xf[j & 1][mt], acc[ct][mt]);
      }
    }
  }
  // drain the ring before the epilogue's ordinary loads / exit
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---------------------------------------------------------------------------------------- epilogue
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    f32x4 r[NTB];
#pragma unroll
    for (int ct = 0; ct < NTB; ++ct) r[ct] = acc[ct][mt];
    conv_epilogue_row<T, KS, NTB>(a, r, wave * MT + mt, TH, cb, nt_real, n, ty, tx, m0, px, g);
  }
}

// ---- epilogue of one 16-pixel row through LDS (bf16, ConvK::vec8): the accumulators (per lane 4 channels of a pixel in
// each of NTB tiles) are written to a wave-private LDS patch as fp32 [pixel][channel] and read back as items of
// (pixel, 8 consecutive channels), one per lane, so that every global access of the epilogue -- residual, activation
// -gradient operand, pre-activation and output stores -- is 16 bytes per lane and a pixel's channels leave as whole
// lines.  The 8-byte stores of the register layout were store-issue bound (16 separate 32-byte segments per instruction).
template <int NTB>
struct EpiLds {
  static constexpr int COB = NTB * 16, C8 = 2 * NTB, PSTR = COB * 4 + 16, NIT = (16 * C8 + 63) / 64;
  static constexpr int BYTES = 16 * PSTR;
};

template <int KS, int NTB>
__device__ __forceinline__ void conv_epilogue_prefetch8(const ConvK& a, int row, int TH, int cb, int n, int ty, int tx, long long m0, int lane,
                                                        bf16x8 (&pre_res)[EpiLds<NTB>::NIT], bf16x8 (&pre_aux)[EpiLds<NTB>::NIT]) {
  using E = EpiLds<NTB>;
  const bf16* res = reinterpret_cast<const bf16*>(a.res);
  const bf16* aux = reinterpret_cast<const bf16*>(a.aux);
#pragma unroll
  for (int it = 0; it < E::NIT; ++it) {
    const int j = it * 64 + lane, pxi = j / E::C8, c8 = j - pxi * E::C8;
    long long pix;
    const bool valid = conv_row_pixel<KS>(a, row, TH, n, ty, tx, m0, pxi & 15, pix);
    const int co = cb * E::COB + c8 * 8;
    if (j < 16 * E::C8 && valid && co < a.Cout) {
      if (res) pre_res[it] = *reinterpret_cast<const bf16x8*>(res + pix * a.res_ps + co);
      if (aux) pre_aux[it] = *reinterpret_cast<const bf16x8*>(aux + pix * a.aux_ps + co);
    }
  }
}

template <int KS, int NTB>
__device__ __forceinline__ void conv_epilogue_lds8(const ConvK& a, const f32x4 (&acc)[NTB], char* stg, int row, int TH, int cb, int n, int ty, int tx,
                                                   long long m0, int lane, const bf16x8 (&pre_res)[EpiLds<NTB>::NIT],
                                                   const bf16x8 (&pre_aux)[EpiLds<NTB>::NIT], const float* lbias) {
  using E = EpiLds<NTB>;
  const int px = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ct = 0; ct < NTB; ++ct) *reinterpret_cast<f32x4*>(stg + px * E::PSTR + (ct * 16 + g * 4) * 4) = acc[ct];
  bf16* out = reinterpret_cast<bf16*>(a.out);
  bf16* out_pre = reinterpret_cast<bf16*>(a.out_pre);
  const float neg = a.act == VMG_ACT_RELU ? 0.f : (a.act == VMG_ACT_LRELU ? a.slope : 1.f);
#pragma unroll
  for (int it = 0; it < E::NIT; ++it) {
    const int j = it * 64 + lane, pxi = j / E::C8, c8 = j - pxi * E::C8;
    long long pix;
    const bool valid = conv_row_pixel<KS>(a, row, TH, n, ty, tx, m0, pxi & 15, pix);
    const int co = cb * E::COB + c8 * 8;
    if (!(j < 16 * E::C8 && valid && co < a.Cout) || VMG_DBG(a, 8)) continue;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + pxi * E::PSTR + c8 * 32);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + pxi * E::PSTR + c8 * 32 + 16);
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    if (a.bias) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(lbias + c8 * 8), b1 = *reinterpret_cast<const f32x4*>(lbias + c8 * 8 + 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] += b0[r]; v[4 + r] += b1[r]; }
    }
    if (out_pre) {
      bf16x8 t;
#pragma unroll
      for (int r = 0; r < 8; ++r) t[r] = (bf16)v[r];
      *reinterpret_cast<bf16x8*>(out_pre + pix * a.out_ps + co) = t;
    }
    if (a.act == VMG_ACT_GELU) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = gelu_erf(v[r]) * a.alpha;
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = (v[r] > 0.f ? v[r] : v[r] * neg) * a.alpha;
    }
    if (a.aux) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float u = (float)pre_aux[it][r];
        v[r] *= a.actgrad == 3 ? gelu_erf_grad(u) : (u > 0.f ? 1.f : (a.actgrad == 2 ? a.slope : 0.f));
      }
    }
    if (a.res) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += (float)pre_res[it][r];
    }
    bf16x8 t;
#pragma unroll
    for (int r = 0; r < 8; ++r) t[r] = (bf16)v[r];
    *reinterpret_cast<bf16x8*>(out + pix * a.out_ps + co) = t;
  }
}

// ================================================================================================ K-split variant
// Same decomposition of the OUTPUT (64 pixels x NTB*16 channels per workgroup, epilogue row w on wave w), but the four
// waves split the K loop instead of the pixels: every wave accumulates the whole 64 x COB tile over a quarter of the
// k-steps and the partial sums are exchanged through LDS at the end (reduce-scatter: wave q ends with row q).
//   * per k-step a wave issues 4 activation reads (LDS) + NTB weight reads for 4*NTB MFMAs (v4: 1 + NTB reads for NTB
//     MFMAs -- that kernel is LDS-bandwidth bound, its 4 waves all read the same weight fragments);
//   * a wave's weight fragments are used by that wave only, so they go global -> VGPR directly (1 KiB contiguous per
//     16-lane group in the packed layout), prefetched two k-steps ahead in three static register sets; no weight ring,
//     no barrier and no manual waitcnt in the main loop;
//   * rows are visited in the rotated order (wave + i) & 3 so that the wave's own row has the STATIC register index 0;
//   * a wave's k-range is padded to a multiple of 3 (the unroll); padding steps read a 16-byte zero slot in LDS as their
//     activation operand;
//   * residual / activation-gradient operands and the bias are fetched before the main loop.
template <typename T, int KS, int NTB>
__global__ __launch_bounds__(256, (NTB <= 3 ? 3 : 2)) void conv_ksplit_kernel(const ConvK a) {
  constexpr int ES = ElemTraits<T>::ES, CB = ElemTraits<T>::CHUNKB;
  constexpr int TH = 4;
  constexpr int TWH = 16 + KS - 1;
  constexpr int COB = NTB * 16;
  constexpr int SS = stage_stride(KS, NTB, CB);
  constexpr int KSB = 4 * COB * CB;  // bytes of one k-step in a stage image
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;                     // halo tile; after the main loop: reduce scratch, 12 slots of NTB KiB
  char* zslot = smem + a.halo_bytes;     // 16 zero bytes
  float* lbias = reinterpret_cast<float*>(zslot + 16);
  if (VMG_DBG(a, 16)) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, g = lane >> 4;
  const int cb = blockIdx.y;
  const int nt_real = min(NTB, (a.Cout - cb * COB + 15) >> 4);
  int n = 0, ty = 0, tx = 0;
  long long m0 = 0;
  if (KS > 1) {
    int bid = xcd_remap(blockIdx.x, gridDim.x, a.dbg);
    tx = bid % a.tiles_x;
    int r = bid / a.tiles_x;
    ty = r % a.tiles_y;
    n = r / a.tiles_y;
  } else {
    m0 = (long long)blockIdx.x * 64;
  }
  conv_stamp(a, wave, 0);
  if (tid < 4) reinterpret_cast<int*>(zslot)[tid] = 0;
  // bias: fetched now, written to LDS once the halo has landed (no wait on it here)
  const float bias_r = (tid < COB && a.bias && cb * COB + tid < a.Cout) ? a.bias[cb * COB + tid] : 0.f;

  // ---- early fetch of the epilogue operands of this wave's own row (16-byte path only)
  constexpr bool BF = std::is_same<T, bf16>::value;
  const bool vec8 = BF && a.vec8 != 0;
  bf16x8 pre_res[EpiLds<NTB>::NIT], pre_aux[EpiLds<NTB>::NIT];
  if (vec8) conv_epilogue_prefetch8<KS, NTB>(a, wave, TH, cb, n, ty, tx, m0, lane, pre_res, pre_aux);

  f32x4 acc[NTB][4];
#pragma unroll
  for (int ct = 0; ct < NTB; ++ct)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[ct][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const char* wlane = a.wpack + (long long)cb * a.nstages * SS + g * (COB * CB) + px * CB;
  int st0 = 0;
  for (int s = 0; s < a.nsrc; ++s) {
    const int ch = a.src_ch[s], pixb = a.src_pixb[s];
    const int CH = ch >> 3, nb = (CH + 3) >> 2;
    if (s > 0) __syncthreads();  // everyone is done reading the previous halo tile
    conv_stamp(a, wave, 1);
    stage_halo<T, KS, 1>(a, s, halo, n, ty, tx, m0, tid);
    if (s == 0 && tid < COB) lbias[tid] = bias_r;
    conv_stamp(a, wave, 3);
    __syncthreads();
    conv_stamp(a, wave, 4);

    const int nks = VMG_DBG(a, 32) ? 0 : (KS > 1 ? 9 * nb : nb);  // k-steps of this source
    const int per = ((nks + 3) / 4 + 2) / 3 * 3;                 // per wave, padded to the unroll
    const int k0 = wave * per;
    const int rowb = TWH * pixb;  // LDS bytes per tile row (KS == 1: TWH = 16 pixels)
    const char* pixp = halo + px * pixb;
    int rowoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rowoff[i] = ((wave + i) & 3) * rowb;

    Frag<T> wf[3][NTB], xf[3][4];
    // The wave walks k-steps k0, k0+1, ... with TWO cursors (weights run two steps ahead of the activations).  A cursor
    // is advanced like an odometer -- (kx, ky, channel block) and a running byte offset into the packed weights -- because
    // decoding a flat k index (divisions by 9 and 3, 64-bit products) for every load cost ~80 scalar instructions per
    // k-step, more than everything else in the loop.  Past the last real k-step a cursor stays on it (`real` turns false).
    struct Cur { int k, cbk, ky, kx; long long woff; };
    auto cur_at = [&](int k) {
      Cur c;
      c.k = k;
      const int kc = min(k, nks - 1);
      if (KS > 1) {
        c.cbk = kc / 9;
        const int tap = kc - 9 * c.cbk;
        c.ky = tap / 3;
        c.kx = tap - 3 * c.ky;
        c.woff = (long long)(st0 + c.cbk * 3 + c.ky) * SS + c.kx * KSB;
      } else {
        c.cbk = kc; c.ky = 0; c.kx = 0;
        c.woff = (long long)(st0 + (kc >> 1)) * SS + (kc & 1) * KSB;
      }
      return c;
    };
    auto advance = [&](Cur& c) {
      ++c.k;
      if (c.k < nks) {
        if (KS > 1) {
          c.woff += KSB;
          if (++c.kx == 3) {
            c.kx = 0;
            c.woff += SS - 3 * KSB;
            if (++c.ky == 3) { c.ky = 0; ++c.cbk; }
          }
        } else {
          c.woff += (c.k & 1) ? KSB : SS - KSB;
          c.cbk = c.k;
        }
      }
    };
    Cur cw = cur_at(k0), cx = cw;
    auto load_w = [&](int set) {  // weights of the cursor's k-step (a padding step multiplies them by zeros)
      const char* p = wlane + cw.woff;
#pragma unroll
      for (int ct = 0; ct < NTB; ++ct) wf[set][ct].load(p + ct * 16 * CB);
      advance(cw);
    };
    auto load_x = [&](int set) {
      const bool real = cx.k < nks;
      const int boff = (KS > 1 ? cx.ky * rowb + cx.kx * pixb : 0) + min(4 * cx.cbk + g, CH - 1) * CB;
      const char* base = real ? pixp + boff : zslot;
#pragma unroll
      for (int i = 0; i < 4; ++i) xf[set][i].load(base + (real ? rowoff[i] : 0));
      advance(cx);
    };
    auto mfma_set = [&](int set) {
#pragma unroll
      for (int ct = 0; ct < NTB; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[ct][i] = mma(wf[set][ct], xf[set][i], acc[ct][i]);
    };
    // (Asking for the first two k-steps' weights BEFORE the halo tile is staged -- they do not depend on it -- was measured at one clip per GPU,
    // M = 8 192, round 4: 9.66 us per launch against 9.47; the halo's own loads then queue behind them.)
    if (nks > 0) {
      load_w(0);
      load_w(1);
      load_x(0);
      for (int i = 0; i < per; i += 3) {
        load_w(2);
        load_x(1);
        mfma_set(0);
        load_w(0);
        load_x(2);
        mfma_set(1);
        load_w(1);
        load_x(0);
        mfma_set(2);
      }
    }
    st0 += a.src_nst[s];
  }

  // ---- reduce-scatter of the four partial tiles: slot (q, j) = row q's partial from wave (q + j) & 3, j = 1..3
  conv_stamp(a, wave, 5);
  __syncthreads();  // all waves are done with the halo tile
  char* scratch = smem;
#pragma unroll
  for (int i = 1; i < 4; ++i) {
    const int q = (wave + i) & 3, j = 4 - i;
    char* dst = scratch + ((q * 3 + (j - 1)) * NTB) * 1024 + lane * 16;
#pragma unroll
    for (int ct = 0; ct < NTB; ++ct) *reinterpret_cast<f32x4*>(dst + ct * 1024) = acc[ct][i];
  }
  __syncthreads();
  f32x4 own[NTB];
#pragma unroll
  for (int ct = 0; ct < NTB; ++ct) own[ct] = acc[ct][0];
#pragma unroll
  for (int j = 1; j < 4; ++j) {
    const char* src = scratch + ((wave * 3 + (j - 1)) * NTB) * 1024 + lane * 16;
#pragma unroll
    for (int ct = 0; ct < NTB; ++ct) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(src + ct * 1024);
      own[ct] += t;
    }
  }
  conv_stamp(a, wave, 6);
  if (vec8) {
    // staging patch = the scratch slots only this wave reads (its reads above have completed: their data is in `own`)
    static_assert(EpiLds<NTB>::BYTES <= 3 * NTB * 1024, "epilogue staging patch must fit the wave's own scratch slots");
    conv_epilogue_lds8<KS, NTB>(a, own, scratch + (wave * 3 * NTB) * 1024, wave, TH, cb, n, ty, tx, m0, lane, pre_res, pre_aux, lbias);
  } else {
    conv_epilogue_row<T, KS, NTB, false>(a, own, wave, TH, cb, nt_real, n, ty, tx, m0, px, g);
  }
  conv_stamp(a, wave, 7);
}

// ================================================================================================ wave-autonomous 1x1
// 1x1 convolutions / Linears with a small K (<= 160 input channels, one source), bf16: HBM-bound (33 MB in, 33 MB out, 2 us of
// MFMA at M = 114 688) but the general kernel spent 36 us on them -- 1 792 workgroups of 64 rows, each streaming the 41 KiB
// weight block through the LDS ring for 45 MFMAs per wave.  Here EVERY WAVE IS ITS OWN PIPELINE: it keeps the whole weight
// block of its output-channel block resident in registers (5 k-steps x NTB fragments), walks over consecutive 16-row tiles
// of the (M, C) matrix, copies tile t+1 into a wave-private LDS buffer by LDS-DMA while it multiplies tile t, and runs the
// 16-byte epilogue through a wave-private LDS patch.  No workgroup barrier anywhere; one counted s_waitcnt per tile.
template <int NTB, int NS = 1>
__global__ __launch_bounds__(256, NS == 1 ? 2 : 1) void linear_wres_kernel(const ConvK a, int ntiles, int tiles_per_wave) {
  using T = bf16;
  // NS = 1: one source of <= 160 channels = 5 k-steps; a 16-row tile = <= 320 vectors = 5 x 1 KiB.  NS = 2: a source of <= 320 channels that the
  // pack splits into two equal blocks (Mlp_cnn.fc2: 288 = 2 x 144): 10 k-steps, the tile is still ONE dense list of 16 x (2 CH) vectors.
  constexpr int KS = 1, CB = 16, NKS = 5, NK = NKS * NS, NDMA = NK;
  constexpr int COB = NTB * 16;
  constexpr int SS = stage_stride(KS, NTB, CB);
  constexpr int KSB = 4 * COB * CB;
  constexpr int XBUF = NDMA * 1024;
  constexpr int WAVE_LDS = 2 * XBUF + ((EpiLds<NTB>::BYTES + 1023) & ~1023);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, g = lane >> 4;
  const int cb = blockIdx.y;
  char* xbuf = smem + wave * WAVE_LDS;
  char* patch = xbuf + 2 * XBUF;
  const int CH = a.src_ch[0] >> 3, nb = (CH + 3) >> 2;  // chunks / k-steps per block
  const int vpp = CH * NS;                               // 16-byte vectors per row
  const int pixb = vpp * 16;                             // (dense rows in LDS)
  const long long ps_b = a.src_ps[0] * 2;
  const int t0 = (blockIdx.x * 4 + wave) * tiles_per_wave, t1 = min(ntiles, t0 + tiles_per_wave);
  if (t0 >= t1) return;  // (whole wave; nothing below synchronises across waves)

  // ---- resident weights: every k-step of this output-channel block (k-steps past the real ones are zero fragments)
  Frag<T> wres[NK][NTB];
  {
    const char* wlane = a.wpack + (long long)cb * a.nstages * SS + g * (COB * CB) + px * CB;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int sb = k / NKS, kk = k - sb * NKS;
      const int kc = min(kk, nb - 1);
      const char* p = wlane + (long long)(sb * a.src_nst[0] + (kc >> 1)) * SS + (kc & 1) * KSB;
#pragma unroll
      for (int ct = 0; ct < NTB; ++ct) {
        wres[k][ct].load(p + ct * 16 * CB);
        if (kk >= nb) wres[k][ct].v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
  }
  // ---- lane constants of the tile copy: vector L = 64 i + lane of the [16 rows][vpp] list
  int c_off[NDMA];
  int c_row[NDMA];  // row of the vector, or 16 for lanes past the end of the list
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int L = 64 * i + lane;
    const int r = (int)(((float)L + 0.5f) * (1.0f / (float)vpp));
    c_row[i] = L < 16 * vpp ? r : 16;
    c_off[i] = (int)(r * ps_b) + (L - r * vpp) * 16;
  }
  auto issue = [&](int t, int buf) {
    const long long m0 = (long long)t * 16;
    const char* origin = a.src[0] + m0 * ps_b;
    const long long left = a.M - m0;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const bool ok = c_row[i] < 16 && c_row[i] < left;
      const char* gp = ok ? origin + c_off[i] : reinterpret_cast<const char*>(&g_conv_zero16);
      __builtin_amdgcn_global_load_lds(GLB_PTR(gp), LDS_PTR(xbuf + buf * XBUF + i * 1024), 16, 0, 0);
    }
  };
  int boff[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int sb = k / NKS, kk = k - sb * NKS;
    boff[k] = px * pixb + (sb * CH + min(4 * kk + g, CH - 1)) * CB;
  }

  issue(t0, 0);
  int buf = 0;
  for (int t = t0; t < t1; ++t, buf ^= 1) {
    const long long m0 = (long long)t * 16;
    bf16x8 pre_res[EpiLds<NTB>::NIT], pre_aux[EpiLds<NTB>::NIT];
    conv_epilogue_prefetch8<KS, NTB>(a, 0, 4, cb, 0, 0, 0, m0, lane, pre_res, pre_aux);
    if (t + 1 < t1) {
      issue(t + 1, buf ^ 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");  // tile t (and its epilogue operands) landed; tile t+1 stays in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    f32x4 acc[NTB];
#pragma unroll
    for (int ct = 0; ct < NTB; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* xb = xbuf + buf * XBUF;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      Frag<T> xf;
      xf.load(xb + boff[k]);
#pragma unroll
      for (int ct = 0; ct < NTB; ++ct) acc[ct] = mma(wres[k][ct], xf, acc[ct]);
    }
    conv_epilogue_lds8<KS, NTB>(a, acc, patch, 0, 4, cb, 0, 0, 0, m0, lane, pre_res, pre_aux, a.bias ? a.bias + cb * COB : nullptr);
  }
}

// ================================================================================================ weights-stationary 3x3 (few channels, many pixels)
// The HR head (HRconv 64 -> 64 and conv_last's data gradient 8 -> 64 on 28 x 256 x 256 = 1.8 M pixels; models/vmg.py:629-632) on the general
// kernel: 28 672 workgroups of 64 pixels, each streaming the layer's 74 KiB of weights through its LDS ring -- 2.1 GB of L2 -> LDS traffic for
// 470 MB of activations, ~400 us per launch (1.2 TB/s, 13 % of the matrix roof).  Here the weights never move: ONE workgroup per CU keeps the
// whole packed weight block of its output-channel block in LDS and walks over 128-pixel tiles (8 rows x 16 columns, like conv_ws_kernel).
// Eight waves with fixed roles:
//   * waves 0..3, the CONSUMERS, multiply tile t: wave w owns pixel rows 2w, 2w+1 x all NCT channel tiles.  The weight fragments of the first
//     RW = min(NCT, 2) channel tiles of EVERY k-step live in the wave's registers (144 VGPRs at 64 -> 64), the others are read from LDS: per
//     k-step 2 activation + (NCT - RW) weight reads for 2 NCT MFMAs.  (All weight fragments from LDS -- the same 4 KiB read by four waves
//     per k-step -- made the loop LDS-bound: 2.9 us per tile for 1.2 us of MFMAs.)  A finished tile goes to an LDS patch [128 px][COB fp32];
//   * waves 4..7 are HELPERS, one per SIMD beside a consumer.  Each copies a quarter of the halo tile of tile t+1 (10 x 18 pixels x <= 64
//     channels, LDS pixel stride 144 bytes = 8 chunks + 1 pad for conflict-free 16-byte reads) by LDS-DMA into the second of two halo
//     buffers, and turns a quarter of the patch of tile t-1 into (pixel, 8 channels) items -- bias, activation, mask, residual: every
//     epilogue option of the general kernel, 16-byte stores, NCT items per lane -- while the consumers multiply tile t (the epilogue on the
//     consumers cost 2 us per tile; on three store waves beside one loader 2.5 us);
//   * two workgroup barriers per tile: X_t "halo t has landed, patch t-1 is written, the other halo buffer is free" and Y_t "patch t-1 is
//     drained" (before the consumers overwrite it).
// Tiles are dealt to the XCDs in contiguous bands (workgroup b belongs to XCD b % 8), so that the halo overlap of neighbouring tiles is read
// through one L2.  HBM-bound: (23 KiB in + 16 KiB out) per tile.
// FULL = false: the epilogue is bias + ReLU / LeakyReLU / none + scale only (what the HR head uses); the full item code (GELU, mask, residual,
// pre-activation output) is ~5 000 instructions that the helpers have to branch around.
template <int NCT, int NB, bool FULL>
__global__ __launch_bounds__(512, 1) void conv_wstat_kernel(const ConvK a, int ntiles) {
  using T = bf16;
  constexpr int KS = 3, CB = 16, TH = 8, TWH = 18, THH = 10;
  constexpr int COB = NCT * 16, KSB = 4 * COB * CB;
  constexpr int SS = stage_stride(KS, NCT, CB);  // a stage of the pack = the three taps of one (channel block, tap row)
  constexpr int NK = 9 * NB, RW = NCT < 2 ? NCT : 2;
  constexpr int PIXB = 144;
  constexpr int HVEC = THH * TWH * 9, NDMA = (HVEC + 63) / 64, HB = NDMA * 1024;
  constexpr int PSTR = COB * 4 + 16, C8 = 2 * NCT, NITEM = 128 * C8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, g = lane >> 4;
  const int cb = blockIdx.y;
  const int CH = a.src_ch[0] >> 3;
  constexpr int wbytes = 3 * NB * SS;
  char* wl = smem;
  char* halo = smem + wbytes;                                // [2][HB]
  float* lbias = reinterpret_cast<float*>(halo + 2 * HB);   // [COB]
  char* patch = halo + 2 * HB + COB * 4;                     // [128][PSTR]

  // ---- the tiles of this workgroup: XCD band, then strided over the band's workgroups
  const int per_xcd = gridDim.x >> 3;  // (the host launches a multiple of 8 workgroups)
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int band = (ntiles + 7) >> 3;
  const int band0 = xcd * band, band1 = min(ntiles, band0 + band);
  auto tile_of = [&](int it) { return band0 + it * per_xcd + local; };  // < band1 while the workgroup has work
  auto barrier = [&]() {  // a bare barrier behind an LDS wait: __syncthreads() would also wait for the store waves' global stores
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // ---- prologue: the weight block and the bias -> LDS (all waves)
  {
    const char* wblk = a.wpack + (long long)cb * wbytes;
    for (int i = tid * 16; i < wbytes; i += 512 * 16) *reinterpret_cast<uint4*>(wl + i) = *reinterpret_cast<const uint4*>(wblk + i);
    for (int i = tid; i < COB; i += 512) lbias[i] = (a.bias && cb * COB + i < a.Cout) ? a.bias[cb * COB + i] : 0.f;
  }

  if (wave >= 4) {
    // ================================================================ helpers: quarter of the halo copy + quarter of the previous tile's items
    const int h = wave - 4;
    // copy: vector v = 64 i + lane of the halo image [THH*TWH pixels][9 slots], helper h takes instructions i = h, h + 4, ...
    constexpr int NDH = (NDMA + 3) / 4;
    int lpos[NDH], loff[NDH];
    const long long ps_b = a.src_ps[0] * 2;
#pragma unroll
    for (int k = 0; k < NDH; ++k) {
      const int v = 64 * (h + 4 * k) + lane;
      const int p = v / 9, c = v - p * 9;
      const int hy = p / TWH, hx = p - hy * TWH;
      lpos[k] = (v < HVEC && c < CH) ? (hy << 8) | hx : -1;
      loff[k] = (hy * a.W + hx) * (int)ps_b + c * 16;
    }
    int lfast[NDH];  // the offsets of the fast path: a slot nobody reads (pad slot, chunk >= CH, past the tile) copies the tile's first vector
#pragma unroll
    for (int k = 0; k < NDH; ++k) lfast[k] = lpos[k] >= 0 ? loff[k] : 0;
    auto issue_halo = [&](int t, int buf) {
      const int tx = t % a.tiles_x, r = t / a.tiles_x;
      const int ty = r % a.tiles_y, n = r / a.tiles_y;
      const int y0 = ty * TH - 1, x0 = tx * 16 - 1;
      const char* origin = a.src[0] + (((long long)n * a.H + y0) * a.W + x0) * ps_b;  // (may lie before the tensor: only in-image lanes use it)
      char* dst = halo + buf * HB;
      if (VMG_DBG(a, 1)) return;  // (ablation bits of the diagnostics build: 1 no halo copies, 32 no K loop, 64 no epilogue)
      if (y0 >= 0 && x0 >= 0 && y0 + THH <= a.H && x0 + TWH <= a.W) {  // the whole halo tile inside the image (wave-uniform): base + lane offset
#pragma unroll
        for (int k = 0; k < NDH; ++k) {
          if (h + 4 * k >= NDMA) break;  // (wave-uniform)
          glds16_asm_s(origin, lfast[k], dst + (h + 4 * k) * 1024);
        }
        return;
      }
#pragma unroll
      for (int k = 0; k < NDH; ++k) {
        if (h + 4 * k >= NDMA) break;  // (wave-uniform)
        const bool ok = lpos[k] >= 0 && (unsigned)(y0 + (lpos[k] >> 8)) < (unsigned)a.H && (unsigned)(x0 + (lpos[k] & 255)) < (unsigned)a.W;
        const char* gp = ok ? origin + loff[k] : reinterpret_cast<const char*>(&g_conv_zero16);
        glds16_asm(gp, dst + (h + 4 * k) * 1024);
      }
    };
    // items: item j = 64 h + lane + 256 i of the [128 pixels][C8] list, i < NCT -- pixel, channel group, patch and output offsets do not
    // depend on the tile
    bf16* out = reinterpret_cast<bf16*>(a.out);
    bf16* out_pre = FULL ? reinterpret_cast<bf16*>(a.out_pre) : nullptr;
    const bf16* res = FULL ? reinterpret_cast<const bf16*>(a.res) : nullptr;
    const bf16* aux = FULL ? reinterpret_cast<const bf16*>(a.aux) : nullptr;
    const float neg = a.act == VMG_ACT_RELU ? 0.f : (a.act == VMG_ACT_LRELU ? a.slope : 1.f);
    int ipos[NCT], ipatch[NCT], ico[NCT];
#pragma unroll
    for (int i = 0; i < NCT; ++i) {
      const int j = 64 * h + lane + 256 * i;
      const int p = j / C8, c8 = j - p * C8;
      ipos[i] = ((p >> 4) << 8) | (p & 15);
      ipatch[i] = p * PSTR + c8 * 32;
      ico[i] = c8 * 8;
    }
    const int nstore = NCT * (out_pre ? 2 : 1);  // global stores a helper issues per tile, all of them unconditional (see below)
    // the fast path of the simple epilogue on a tile that lies inside the image: byte offset of the item from the tile's first output vector,
    // the bias of the item's 8 channels in registers
    int ioff[NCT];
    float ibias[NCT][8];
#pragma unroll
    for (int i = 0; i < NCT; ++i) {
      ioff[i] = (((ipos[i] >> 8) * a.W + (ipos[i] & 255)) * (int)a.out_ps + ico[i]) * 2;
#pragma unroll
      for (int q = 0; q < 8; ++q) ibias[i][q] = (a.bias && cb * COB + ico[i] + q < a.Cout) ? a.bias[cb * COB + ico[i] + q] : 0.f;
    }

    if (tile_of(0) < band1) issue_halo(tile_of(0), 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int it = 0;; ++it) {
      barrier();  // X_it: halo it has landed, patch it-1 is written (after the last tile: the drain's X)
      const bool more = tile_of(it) < band1;
      if (more && tile_of(it + 1) < band1) issue_halo(tile_of(it + 1), (it + 1) & 1);
      if (it > 0 && !VMG_DBG(a, 64)) {
        const int t = tile_of(it - 1);
        const int tx = t % a.tiles_x, r = t / a.tiles_x;
        const int ty = r % a.tiles_y, n = r / a.tiles_y;
        const long long pix0 = ((long long)n * a.H + ty * TH) * a.W + tx * 16;
        if (!FULL && ty * TH + TH <= a.H && tx * 16 + 16 <= a.W && cb * COB + COB <= a.Cout && !VMG_DBG(a, 8)) {  // (wave-uniform)
          char* obase = reinterpret_cast<char*>(out + pix0 * a.out_ps + cb * COB);
#pragma unroll
          for (int i = 0; i < NCT; ++i) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(patch + ipatch[i]);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(patch + ipatch[i] + 16);
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            bf16x8 tq;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              if (a.bias) v[q] += ibias[i][q];
              v[q] = (v[q] > 0.f ? v[q] : v[q] * neg) * a.alpha;  // (the same operations as below)
              tq[q] = (bf16)v[q];
            }
            *reinterpret_cast<bf16x8*>(obase + ioff[i]) = tq;
          }
        } else
#pragma unroll
        for (int i = 0; i < NCT; ++i) {
          const int y = ty * TH + (ipos[i] >> 8), x = tx * 16 + (ipos[i] & 255);
          const int co = cb * COB + ico[i];
          const bool valid = y < a.H && x < a.W && co < a.Cout && !VMG_DBG(a, 8);
          const long long pix = valid ? pix0 + (long long)(ipos[i] >> 8) * a.W + (ipos[i] & 255) : 0;  // (an item outside the image reads pixel 0 and stores to a trash vector)
          const f32x4 lo = *reinterpret_cast<const f32x4*>(patch + ipatch[i]);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(patch + ipatch[i] + 16);
          float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          // (the item arithmetic of conv_epilogue_lds8, operation for operation: both kernels give the same bits)
          if (a.bias) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(lbias + ico[i]), b1 = *reinterpret_cast<const f32x4*>(lbias + ico[i] + 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] += b0[q]; v[4 + q] += b1[q]; }
          }
          const int cov = valid ? co : 0;
          if (out_pre) {
            bf16x8 tq;
#pragma unroll
            for (int q = 0; q < 8; ++q) tq[q] = (bf16)v[q];
            bf16* dstp = valid ? out_pre + pix * a.out_ps + co : reinterpret_cast<bf16*>(&g_wstat_trash);
            *reinterpret_cast<bf16x8*>(dstp) = tq;
          }
          if (FULL && a.act == VMG_ACT_GELU) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = gelu_erf(v[q]) * a.alpha;
          } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (v[q] > 0.f ? v[q] : v[q] * neg) * a.alpha;
          }
          if (aux) {
            const bf16x8 au = *reinterpret_cast<const bf16x8*>(aux + pix * a.aux_ps + cov);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const float u = (float)au[q];
              v[q] *= a.actgrad == 3 ? gelu_erf_grad(u) : (u > 0.f ? 1.f : (a.actgrad == 2 ? a.slope : 0.f));
            }
          }
          if (res) {
            const bf16x8 rv = *reinterpret_cast<const bf16x8*>(res + pix * a.res_ps + cov);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += (float)rv[q];
          }
          bf16x8 tq;
#pragma unroll
          for (int q = 0; q < 8; ++q) tq[q] = (bf16)v[q];
          bf16* dst = valid ? out + pix * a.out_ps + co : reinterpret_cast<bf16*>(&g_wstat_trash);
          *reinterpret_cast<bf16x8*>(dst) = tq;
        }
      }
      if (!more) break;
      barrier();  // Y_it: patch it-1 is drained
      // the halo copy must have landed before X_{it+1}; it is OLDER than this tile's stores, which may stay in flight: a counted wait
      if (it > 0 && !VMG_DBG(a, 64)) {
        if (nstore == NCT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NCT) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    return;
  }

  // ================================================================ consumers
  barrier();  // X_0 (the weights are in LDS, the first halo tile has landed)
  // the register-resident weight fragments: channel tiles 0..RW-1 of every k-step
  Frag<T> wreg[NK][RW];
  const char* wlane = wl + g * (COB * CB) + px * CB;
#pragma unroll
  for (int j = 0; j < NK; ++j)
#pragma unroll
    for (int ct = 0; ct < RW; ++ct) wreg[j][ct].load(wlane + (j / 3) * SS + (j % 3) * KSB + ct * 16 * CB);
  for (int it = 0;; ++it) {
    const int t = tile_of(it);
    if (t >= band1) break;  // (workgroup-uniform; the other roles leave at the same X)
    const int buf = it & 1;
    f32x4 acc[2][NCT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[mt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* hb = halo + buf * HB + ((2 * wave) * TWH + px) * PIXB;
    if (!VMG_DBG(a, 32)) {
      Frag<T> xf[2][2], wf[2][NCT > RW ? NCT - RW : 1];
      auto load_frags = [&](int j, int set) {
        const int cbk = j / 9, tap = j - cbk * 9, ky = tap / 3, kx = tap - ky * 3;
        const char* xc = hb + min(4 * cbk + g, CH - 1) * CB;  // (chunks past CH meet zero weights)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) xf[set][mt].load(xc + ((mt + ky) * TWH + kx) * PIXB);
#pragma unroll
        for (int ct = RW; ct < NCT; ++ct) wf[set][ct - RW].load(wlane + (j / 3) * SS + (j % 3) * KSB + ct * 16 * CB);
      };
      load_frags(0, 0);
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        if (j + 1 < NK) load_frags(j + 1, (j + 1) & 1);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt][ct] = mma(ct < RW ? wreg[j][ct < RW ? ct : 0] : wf[j & 1][ct >= RW ? ct - RW : 0], xf[j & 1][mt], acc[mt][ct]);
      }
    }
    barrier();  // Y_it: the store waves have drained patch it-1
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) *reinterpret_cast<f32x4*>(patch + ((2 * wave + mt) * 16 + px) * PSTR + (ct * 16 + g * 4) * 4) = acc[mt][ct];
    barrier();  // X_{it+1}: patch it is written (and halo it+1 has landed)
  }
}

// ================================================================================================ weight-streaming variant
// conv_ws_kernel: bf16 3x3, ONE workgroup per 128-pixel tile (8 rows x 16 columns) and per block of NCT*16 output channels
// (144 or 112: all of them for the convs of the recurrent chains), seven waves with fixed roles:
//   * waves 4..6 are LOADERS: they stream the packed weights through an LDS ring of R slots (4 when it fits, else 3) by LDS-DMA,
//     one stage (= 3 k-steps = 3*NCT KiB; loader wave l copies k-step l) per barrier interval, ahead of the stage being consumed,
//     with COUNTED s_waitcnt vmcnt; stage / piece order is rotated per workgroup;
//   * waves 0..3 are CONSUMERS, one per SIMD: they first copy the halo tile (10 x 18 pixels, all channels of a source block; an
//     inline-asm LDS-DMA the compiler does not see as one), then wave w accumulates pixel rows 2w, 2w+1 x all NCT channel tiles (2*NCT MFMAs
//     per k-step from 2 activation + NCT weight fragments read from LDS, three static fragment sets, the reads of the next
//     k-step interleaved one per MFMA gap);
//   * the epilogue is done by ALL seven waves on (pixel, 8 channels) items out of an LDS patch placed in the ring.
// Why: the K-split kernel above gives every 64-pixel x 48-channel workgroup its own copy of a third of the weights (124 KiB) and
// of the halo (31 KiB) -- 238 MB through L2 -> CU per launch at M = 32 768, which is what bounds it (MFMA 23 % busy).  Here a CU
// pulls the 373 KiB of weights ONCE for 128 pixels x 144 channels: 108 MB per launch, 256 workgroups = one per CU.
// K order per source block of CH 8-channel chunks: for each full 32-channel block: 9 taps (one tap per k-step, lane group g =
// chunk 4*blk + g); then the remainder chunks R = CH % 4, several TAPS per k-step (R = 2: lane groups 0,1 = tap 2j, groups
// 2,3 = tap 2j+1), so that 144 channels cost 42 k-steps instead of 45.  ws_slot() below is the single definition of that
// order for the packer and the kernel.
template <int NCT>
struct WsGeo {
  static constexpr int COB = NCT * 16;
  static constexpr int KSB = COB * 64;  // bytes of one k-step image [4 lane groups][COB][8] bf16 = NCT KiB
  static constexpr int STG = 3 * KSB;   // a stage = 3 k-steps
  static constexpr int TH = 8, TWH = 18, THH = 10;
  static constexpr int NLW = 3;         // loader waves
  static constexpr int THREADS = (4 + NLW) * 64;
  static constexpr int NIT = EpiLds<NCT>::NIT;
};
__host__ __device__ inline int ws_rem_ksteps(int R) { return R == 0 ? 0 : (R == 1 ? 3 : (R == 2 ? 6 : 9)); }
__host__ __device__ inline int ws_ksteps(int CH) { return 9 * (CH / 4) + ws_rem_ksteps(CH % 4); }  // always a multiple of 3
// (tap, chunk) of lane group g in k-step j of a block of CH chunks; tap = -1: empty slot (zero weights)
__host__ __device__ inline void ws_slot(int CH, int j, int g, int& tap, int& chunk) {
  const int nfull = CH / 4, R = CH % 4;
  if (j < 9 * nfull) { tap = j % 9; chunk = 4 * (j / 9) + g; return; }
  const int jr = j - 9 * nfull;
  if (R == 1) { tap = 4 * jr + g; chunk = 4 * nfull; }
  else if (R == 2) { tap = 2 * jr + (g >> 1); chunk = 4 * nfull + (g & 1); }
  else { tap = (g == 3) ? -1 : jr; chunk = 4 * nfull + (g == 3 ? 0 : g); }
  if (tap > 8) tap = -1;
}

// diagnostics build only: stamp slot `slot` (0..31) of wave `wave` of this workgroup (100 MHz clock)
__device__ __forceinline__ void ws_stamp(const ConvK& a, int wave, int slot) {
#ifdef VMG_DIAG
  if (a.stamps && slot < 32) {
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) a.stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 32 + slot] = t;
  }
#endif
}

// (Tried and dropped: the consumers storing their accumulators directly -- a lane holds 4 consecutive channels of a pixel, one 8-byte store
// per tile, residual prefetched in the same layout -- instead of the LDS patch below: 22.6 us per launch by events instead of 21.1; the
// 32-byte runs per pixel cost more than the patch and the item pass.)
// The epilogue works on ITEMS = (pixel of the tile, 8 consecutive output channels): item j = pixel j / C8, channel group j % C8,
// thread `tid` of the 448 owns items tid, tid + 448, ...  The consumers spill their accumulators to an LDS patch [128 pixels][COB]
// fp32 (pixel stride COB*4 + 16 bytes: conflict-free 16-byte writes), then ALL seven waves turn items into 16-byte global
// stores -- whole 288-byte pixel rows leave as contiguous runs.  The residual (or, without one, the activation-gradient) operand
// of a thread's items is fetched at kernel start, so its latency is hidden by the main loop.
template <int NCT>
struct WsItems {
  static constexpr int C8 = 2 * NCT, TOTAL = 128 * C8, THREADS = WsGeo<NCT>::THREADS, NIT = (TOTAL + THREADS - 1) / THREADS;
  static constexpr int PSTR = NCT * 64 + 16;  // bytes per pixel in the patch
  __device__ static __forceinline__ void decode(int j, int& pxl, int& c8) {
    pxl = NCT == 9 ? (j * 3641) >> 16 : (NCT == 7 ? (j * 4682) >> 16 : (NCT == 8 ? j >> 4 : (j * 10923) >> 16));  // j / 18, j / 14, j / 16, j / 6 (exact for j < 4000)
    c8 = j - pxl * C8;
  }
};

template <int NCT>
__global__ __launch_bounds__(WsGeo<NCT>::THREADS, (NCT <= 3 ? 4 : 2)) void conv_ws_kernel(const ConvK a) {
  using G = WsGeo<NCT>;
  using I = WsItems<NCT>;
  using T = bf16;
  constexpr int COB = G::COB, KSB = G::KSB, STG = G::STG;
  static_assert(128 * I::PSTR <= 3 * STG, "the epilogue patch must fit the ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* ring = smem + a.halo_bytes;
  const int R = a.ring;  // 4 slots: a slot is refilled two barriers after its last read, no wait needed; 3: one barrier, explicit wait
  float* lbias = reinterpret_cast<float*>(ring + R * STG);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = blockIdx.y;
  const int bid = xcd_remap(blockIdx.x, gridDim.x, 0);
  const int tx = bid % a.tiles_x;
  const int rr = bid / a.tiles_x;
  const int ty = rr % a.tiles_y, n = rr / a.tiles_y;
  const int nstages = a.nstages;
  ws_stamp(a, wave, 0);

  // ---- every wave: its share of the first halo tile (instruction q of the [row][pixel][vector] list goes to wave q % 7)
  auto issue_halo = [&](int s, int first, int step, auto hidden) {
    constexpr bool HIDDEN = decltype(hidden)::value;
    const int ch = a.src_ch[s];
    const char* sp = a.src[s];
    const int ps32 = (int)(a.src_ps[s] * 2);
    const int vpp = ch >> 3, row_vecs = G::TWH * vpp, total = G::THH * row_vecs;
    const float inv_row = 1.0f / (float)row_vecs, inv_vpp = 1.0f / (float)vpp;
    const int y0 = ty * G::TH - 1, x0 = tx * 16 - 1;
    const char* origin = sp + (((long long)n * a.H + y0) * a.W + x0) * (long long)ps32;
    const int rowstep = a.W * ps32;
    const int nq = a.halo_bytes >> 10;
    for (int q = first; q < nq; q += step) {
      const int L = q * 64 + lane;
      const int r = (int)(((float)L + 0.5f) * inv_row);  // exact: L < 2^16
      const int rem = L - r * row_vecs;
      const int p = (int)(((float)rem + 0.5f) * inv_vpp);
      const int v = rem - p * vpp;
      const bool ok = (L < total) & ((unsigned)(y0 + r) < (unsigned)a.H) & ((unsigned)(x0 + p) < (unsigned)a.W);
      const long long off = ok ? (long long)(r * rowstep + p * ps32 + v * 16) : (reinterpret_cast<const char*>(&g_conv_zero16) - origin);
      if (HIDDEN) {
        // consumer waves: the LDS-DMA is an asm statement, so that hipcc does not know this wave has written LDS behind its back
        // -- with the builtin it treats every later ds_read as possibly aliasing and waits lgkmcnt(0) in front of each MFMA
        // group (measured: 0.72 us per stage instead of 0.45).  M0 is saved and restored inside the statement; the copies are
        // counted by hand (the s_waitcnt before B_0).
        unsigned keep;
        const char* gp = origin + off;
        const unsigned ldst = (unsigned)(uintptr_t)LDS_PTR(halo + q * 1024);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gp), "s"(ldst) : "memory");
      } else {
        __builtin_amdgcn_global_load_lds(GLB_PTR(origin + off), LDS_PTR(halo + q * 1024), 16, 0, 0);
      }
    }
  };

  // ---- every thread: the epilogue operand of its items (residual, else activation-gradient operand), fetched now
  const bf16* esrc = reinterpret_cast<const bf16*>(a.res ? a.res : a.aux);
  const long long eps = a.res ? a.res_ps : a.aux_ps;
  const long long tile_pix = ((long long)n * a.H + ty * G::TH) * a.W + tx * 16;
  bf16x8 pre[I::NIT];
  auto prefetch = [&]() {
    if (!esrc) return;
    const bf16* ebase = esrc + tile_pix * eps + cb * COB;
#pragma unroll
    for (int it = 0; it < I::NIT; ++it) {
      int pxl, c8;
      I::decode(it * I::THREADS + tid, pxl, c8);
      const int row = pxl >> 4, col = pxl & 15;
      const bool valid = (pxl < 128) & (ty * G::TH + row < a.H) & (tx * 16 + col < a.W) & (cb * COB + c8 * 8 < a.Cout);
      const int off = valid ? (row * a.W + col) * (int)eps + c8 * 8 : 0;  // (invalid items read the tile's first vector: always in range)
      pre[it] = *reinterpret_cast<const bf16x8*>(ebase + off);
    }
  };

  // Every workgroup walks the stages of a source in its own ROTATED order (first stage = bid % stages; the sum is order-free),
  // and a loader copies the pieces of its k-step in a rotated order too: neighbouring CUs do not ask the L2 for the same lines at
  // the same moment.
  if (wave >= 4) {
    // ------------------------------------------------------------------------------------------------ loader waves
    const int lw = wave - 4;
    const char* wsrc = a.wpack + ((long long)cb * nstages) * STG + lw * KSB + lane * 16;
    const int prot = bid % NCT;
    int slot_i = 0;  // ring slot of the next stage to issue
    int issued = 0;
    // issue cursor: source, first stage of the source, its stage count, rotated stage index within it
    int is = 0, ist0 = 0, inst = a.src_nst[0], isl = 0, irot = bid % a.src_nst[0];
    auto issue_stage = [&]() {
      const char* gsrc = wsrc + (long long)(ist0 + irot) * STG;
      char* dst = ring + slot_i * STG + lw * KSB;
#pragma unroll
      for (int i = 0; i < NCT; ++i) {
        int ii = i + prot;
        ii = ii >= NCT ? ii - NCT : ii;
        __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc + ii * 1024), LDS_PTR(dst + ii * 1024), 16, 0, 0);
      }
      ++issued;
      slot_i = slot_i + 1 == R ? 0 : slot_i + 1;
      irot = irot + 1 == inst ? 0 : irot + 1;
      if (++isl == inst && issued < nstages) {
        ++is;
        ist0 += inst;
        inst = a.src_nst[is];
        isl = 0;
        irot = bid % inst;
      }
    };
    {
      const int c = tid - 256;  // the block's bias goes to LDS (read by the consumers behind B_0)
      if (c < COB) lbias[c] = (a.bias && cb * COB + c < a.Cout) ? a.bias[cb * COB + c] : 0.f;
    }
    prefetch();  // oldest in the queue: never counted by the waits below
    asm volatile("" ::: "memory");
    issue_stage();  // (the consumers, idle until B_0, copy the first halo tile)
    issue_stage();  // (every source has >= 3 stages)
    ws_stamp(a, wave, 1);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(NCT) : "memory");  // stage 0 has landed (stage 1 may be in flight), the bias is written
    ws_stamp(a, wave, 2);
    __builtin_amdgcn_s_barrier();  // B_0
    int t = 0;
    for (int s = 0; s < a.nsrc; ++s) {
      const int nst = a.src_nst[s];
      for (int sl = 0; sl < nst; ++sl, ++t) {
        // just behind B_t: the consumers have finished stage t-1, its slot takes stage t+2
        if (issued < nstages) issue_stage();
        const bool last = t + 1 == nstages, sw = !last && sl + 1 == nst;
        if (sw) {
          __builtin_amdgcn_s_barrier();  // X: the consumers are done with the halo tile of source s
          issue_halo(s + 1, lw, G::NLW, std::false_type{});
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (last || issued == nstages) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCT) : "memory");  // stage t+1 has landed; stage t+2 stays in flight
        }
        ws_stamp(a, wave, 4 + t);      // the loader is ready for B_{t+1}
        __builtin_amdgcn_s_barrier();  // B_{t+1}, or E behind the last stage
      }
    }
  } else {
    // ------------------------------------------------------------------------------------------------ consumer waves
    const int px = lane & 15, g = lane >> 4;
    issue_halo(0, wave, 4, std::true_type{});
    asm volatile("" ::: "memory");  // the operand loads below stay YOUNGER than the halo pieces: the counted wait before B_0 relies on it
    prefetch();

    f32x4 acc[NCT][2];
    Frag<T> wf[3][NCT], xf[3][2];

    // cursor over (source, stage of the source); csl counts the stages done, crot is the rotated stage index being consumed
    int cs = 0, csl = 0, cnst = a.src_nst[0], crot = bid % cnst;
    int CH = a.src_ch[0] >> 3, nfull3 = 3 * (CH >> 2), pixb = a.src_pixb[0], rowb = G::TWH * pixb;
    const char* xb = halo + ((2 * wave) * G::TWH + px) * pixb;
    int slot = 0;
    int xo[3];
    auto stage_offsets = [&]() {
      if (crot < nfull3) {
        const int cbk = (crot * 171) >> 9, ky = crot - 3 * cbk;  // crot / 3, crot % 3 (exact below 512)
        const int base = ky * rowb + cbk * 64 + g * 16;
        xo[0] = base; xo[1] = base + pixb; xo[2] = base + 2 * pixb;
      } else {  // remainder chunks (CH % 4 == 2: the host admits 0 and 2 only): lane groups 0,1 = tap 2j, groups 2,3 = tap 2j + 1
        const int nfull = CH >> 2, jr0 = 3 * (crot - nfull3);
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
          int tap = 2 * (jr0 + jj) + (g >> 1);
          tap = tap > 8 ? 8 : tap;          // empty slots carry zero weights: any valid address
          const int tyy = (tap * 11) >> 5;   // tap / 3 for 0..8
          xo[jj] = tyy * rowb + (tap - 3 * tyy) * pixb + (4 * nfull + (g & 1)) * 16;
        }
      }
    };
    auto advance = [&]() {  // to the next stage (same source)
      ++csl;
      crot = crot + 1 == cnst ? 0 : crot + 1;
      slot = slot + 1 == R ? 0 : slot + 1;
    };
    auto next_source = [&]() {
      ++cs;
      csl = 0; cnst = a.src_nst[cs]; crot = bid % cnst;
      CH = a.src_ch[cs] >> 3; nfull3 = 3 * (CH >> 2); pixb = a.src_pixb[cs]; rowb = G::TWH * pixb;
      xb = halo + ((2 * wave) * G::TWH + px) * pixb;
      slot = slot + 1 == R ? 0 : slot + 1;
    };
    auto rd = [&](int set, int jj) {
      // activation fragments first: LDS reads return in order, so the first MFMA of the next k-step waits for 3 of the 11
      // reads only and the later ones have its predecessors as cover
      xf[set][0].load(xb + xo[jj]);
      xf[set][1].load(xb + rowb + xo[jj]);
      const char* wp = ring + slot * STG + jj * KSB + g * (COB * 16) + px * 16;
#ifdef VMG_WS_ABL_ONE_FRAG
      wf[set][0].load(wp);  // ablation build (tools/ws_lds_ablate.sh): ONE weight fragment per k-step instead of NCT -- LDS read traffic 11 -> 3 KiB per wave and k-step
#else
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) wf[set][ct].load(wp + ct * 256);
#endif
    };
    auto mm = [&](int set) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
#ifdef VMG_WS_ABL_ONE_FRAG
        constexpr int cw = 0;  // (wrong sums, same MFMA count: is the K loop bound by the LDS reads?)
#else
        const int cw = ct;
#endif
        acc[ct][0] = mma(wf[set][cw], xf[set][0], acc[ct][0]);
        acc[ct][1] = mma(wf[set][cw], xf[set][1], acc[ct][1]);
      }
    };
    // one phase = the MFMAs of one k-step with the fragment reads of the next one slotted into the gaps between them (an MFMA
    // holds the SIMD's issue port for half of its 16 cycles: one LDS read per gap is free), fenced so that hipcc neither starts
    // the next k-step's MFMAs early nor bunches the reads.  Measured per stage (3 k-steps, 54 MFMAs per wave; 0.54 us would be
    // the matrix pipe alone at the ~1.6 GHz the chip holds under this load): reads bunched in front of the MFMAs 0.92 us,
    // interleaved 0.70 us.
    auto phase_fence = [&]() {
#pragma unroll
      for (int i = 0; i < NCT + 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NCT - (NCT + 2), 0);
      __builtin_amdgcn_sched_barrier(0);
    };

    if (esrc) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(I::NIT) : "memory");  // this wave's halo share has landed; the operand prefetch may fly on
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ws_stamp(a, wave, 1);
    __builtin_amdgcn_s_barrier();  // B_0: halo and the first stage are in LDS
    asm volatile("" ::: "memory");
    ws_stamp(a, wave, 2);
    // the accumulators start from the bias (a lane holds output channels ct*16 + 4g .. +3 of its pixels)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(lbias + ct * 16 + g * 4);
      acc[ct][0] = bv;
      acc[ct][1] = bv;
    }
    stage_offsets();
    rd(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    int t = 0;
    for (int s = 0; s < a.nsrc; ++s) {
      for (int sl = 0; sl + 1 < cnst; ++sl, ++t) {  // every stage of the source but its last: one straight-line body
        rd(1, 1);
        mm(0);
        phase_fence();
        rd(2, 2);
        mm(1);
        phase_fence();
        if (R == 3) __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) (a builtin: hipcc's wait bookkeeping sees it): with 3 slots the
                                                         // loaders refill stage t's slot right behind the barrier
        ws_stamp(a, wave, 3 + t);                        // the consumer is ready for B_{t+1}
        __builtin_amdgcn_s_barrier();        // B_{t+1}
        asm volatile("" ::: "memory");
        advance();
        stage_offsets();
        rd(0, 0);
        mm(2);
        phase_fence();
      }
      // the last stage of the source
      rd(1, 1);
      mm(0);
      phase_fence();
      rd(2, 2);
      mm(1);
      phase_fence();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      ws_stamp(a, wave, 3 + t);
      __builtin_amdgcn_s_barrier();  // X: done with this source's halo tile (the loaders bring the next one) / E behind the last stage
      asm volatile("" ::: "memory");
      mm(2);
      __builtin_amdgcn_sched_barrier(0);
      ++t;
      if (s + 1 < a.nsrc) {
        __builtin_amdgcn_s_barrier();  // B_{t}: the next source's halo tile has landed
        asm volatile("" ::: "memory");
        next_source();
        stage_offsets();
        rd(0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    ws_stamp(a, wave, 28);
    // accumulators -> patch (the ring is idle: every consumer is past E, the loaders have nothing in flight)
    char* prow = ring + ((2 * wave) * 16 + px) * I::PSTR + g * 16;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      *reinterpret_cast<f32x4*>(prow + ct * 64) = acc[ct][0];
      *reinterpret_cast<f32x4*>(prow + 16 * I::PSTR + ct * 64) = acc[ct][1];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
  ws_stamp(a, wave, 29);
  __builtin_amdgcn_s_barrier();  // F: the patch is complete
  asm volatile("" ::: "memory");

  // ---- items -> global memory, all seven waves
  {
    bf16* out = reinterpret_cast<bf16*>(a.out) + tile_pix * a.out_ps + cb * COB;
    bf16* out_pre = a.out_pre ? reinterpret_cast<bf16*>(a.out_pre) + tile_pix * a.out_ps + cb * COB : nullptr;
    const bf16* auxb = a.aux ? reinterpret_cast<const bf16*>(a.aux) + tile_pix * a.aux_ps + cb * COB : nullptr;
#pragma unroll
    for (int it = 0; it < I::NIT; ++it) {
      int pxl, c8;
      I::decode(it * I::THREADS + tid, pxl, c8);
      const int row = pxl >> 4, col = pxl & 15;
      const bool valid = (pxl < 128) & (ty * G::TH + row < a.H) & (tx * 16 + col < a.W) & (cb * COB + c8 * 8 < a.Cout);
      if (!valid) continue;
      const char* pp = ring + pxl * I::PSTR + c8 * 32;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(pp), hi = *reinterpret_cast<const f32x4*>(pp + 16);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};  // (bias included: the accumulators started from it)
      const int opix = row * a.W + col;
      if (out_pre) {
        bf16x8 tq;
#pragma unroll
        for (int r = 0; r < 8; ++r) tq[r] = (bf16)v[r];
        *reinterpret_cast<bf16x8*>(out_pre + opix * (int)a.out_ps + c8 * 8) = tq;
      }
      // wave-uniform branches: each activation costs only what it needs (ReLU: one max per element)
      if (a.act == VMG_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = gelu_erf(v[r]);
      } else if (a.act == VMG_ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
      } else if (a.act == VMG_ACT_LRELU) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * a.slope;
      }
      if (a.alpha != 1.0f) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] *= a.alpha;
      }
      if (auxb) {
        bf16x8 u = pre[it];
        if (a.res) u = *reinterpret_cast<const bf16x8*>(auxb + opix * (int)a.aux_ps + c8 * 8);  // both operands: aux is read here
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float uf = (float)u[r];
          v[r] *= a.actgrad == 3 ? gelu_erf_grad(uf) : (uf > 0.f ? 1.f : (a.actgrad == 2 ? a.slope : 0.f));
        }
      }
      if (a.res) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)pre[it][r];
      }
      bf16x8 tq;
#pragma unroll
      for (int r = 0; r < 8; ++r) tq[r] = (bf16)v[r];
      __builtin_nontemporal_store(tq, reinterpret_cast<bf16x8*>(out + opix * (int)a.out_ps + c8 * 8));
    }
  }
  ws_stamp(a, wave, 31);
}

// ------------------------------------------------------------------------------------------- packing
struct PackK {
  const float* w;
  char* out;
  int O, I, ks, o0, on, nsrc, transpose_flip, cob, ncb, nstages, ss_elems;  // ss_elems = stage stride in elements
  int groups;  // > 1: w is the (O, I, ks, ks) weight of a GROUPED convolution (I = channels per group); the pack is the dense block-diagonal operator
  int fast3;   // 3x3, one group: the item form below (pack3_item) applies; `units` = its work items + padding vectors
  long long units;
  short src_off[MAX_ISRC], src_ch[MAX_ISRC], src_st0[MAX_ISRC], src_nst[MAX_ISRC];
};

// one weight value of the (possibly grouped) operator: forward packs ask (output channel oc, K-side input channel kc), data-gradient packs
// (output = input channel oc, K-side output channel kc, mirrored tap).  groups > 1: the dense block-diagonal image of the grouped convolution
// -- zero wherever the two channels belong to different groups -- so that ALL groups run as ONE convolution launch (round 4: the full
// configuration's Mlp_cnn.fc1, n_groups = 4, was four launches per call and direction on 28 -> 168 .. 112 -> 672 channel slices).
__device__ __forceinline__ float pack_weight_value(const PackK& p, int oc, int kc, int tap, int KK) {
  if (p.groups <= 1) {
    if (!p.transpose_flip) return p.w[((long long)oc * p.I + kc) * KK + tap];
    return p.w[((long long)kc * p.I + oc) * KK + (KK - 1 - tap)];
  }
  const int og = p.O / p.groups;
  if (!p.transpose_flip) {
    const int kl = kc - (oc / og) * p.I;
    return (kl >= 0 && kl < p.I) ? p.w[((long long)oc * p.I + kl) * KK + tap] : 0.f;
  }
  const int g = oc / p.I;  // oc: dense input channel
  return (kc / og == g) ? p.w[((long long)kc * p.I + (oc - g * p.I)) * KK + (KK - 1 - tap)] : 0.f;
}

// one packed element of [cout block][stage][stage image]; a stage image is [k-step j][chunk g][co][8] followed by zero padding up to
// the 4-KiB-aligned stage stride
// EIGHT consecutive elements (e = 0..7: the 8 K-side channels of one chunk -- one 16-byte vector of the packed image) per call: the index
// arithmetic (five divisions) is done once per vector instead of once per element and the result leaves as one 16-byte store (bf16; two for
// fp32).  Round 4: the per-element form ran at 0.4 TB/s -- 372 us per launch for the full configuration's weights, 2 % of its step.
template <typename T>
__device__ __forceinline__ void pack_std_vec8(const PackK& p, long long i8) {
  // (32-bit index arithmetic: the host admits packs below 2^31 elements only -- a 64-bit division costs ~150 instructions, and there were four per vector)
  const unsigned i = (unsigned)i8 * 8u;
  const int KK = p.ks * p.ks, KSTG = kstg(p.ks);
  const int body = KSTG * 4 * p.cob * 8;
  const unsigned r = i / (unsigned)p.ss_elems;
  const int within = (int)(i - r * (unsigned)p.ss_elems);
  const int cb = (int)(r / (unsigned)p.nstages);
  const int stage = (int)(r - (unsigned)cb * (unsigned)p.nstages);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (within < body) {
    int t = within >> 3;
    const int co = t % p.cob;
    t /= p.cob;
    const int g = t & 3, j = t >> 2;
    int s = 0;
    while (s + 1 < p.nsrc && stage >= p.src_st0[s + 1]) ++s;
    const int sl = stage - p.src_st0[s];
    const int CH = p.src_ch[s] >> 3;
    int tap, q;
    if (p.ks > 1) { const int cbk = sl / p.ks, ky = sl - p.ks * cbk; tap = ky * p.ks + j; q = 4 * cbk + g; }
    else { tap = 0; q = 4 * (2 * sl + j) + g; }
    const int col = cb * p.cob + co;  // output channel within [0, on)
    if (q < CH && col < p.on) {
      const int kc = p.src_off[s] + q * 8;  // first K-side channel of the chunk
      const int oc = p.o0 + col;            // output-side channel
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = pack_weight_value(p, oc, kc + e, tap, KK);
    }
  }
  if constexpr (std::is_same<T, bf16>::value) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.out) + i) = o;
  } else {
    f32x4* o = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + i);
    o[0] = f32x4{v[0], v[1], v[2], v[3]};
    o[1] = f32x4{v[4], v[5], v[6], v[7]};
  }
}

template <typename T>
__global__ void conv_pack_kernel(const PackK p) {
  const long long total8 = (long long)p.ncb * p.nstages * p.ss_elems / 8;  // (the stage stride is 4-KiB aligned: whole vectors)
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) pack_std_vec8<T>(p, i);
}

// weight-streaming layout: [cout block][stage][k-step jj 0..2][lane group g][co][8] bf16, K slots by ws_slot()
__device__ __forceinline__ void pack_ws_vec8(const PackK& p, long long i8) {  // (eight elements per call, see pack_std_vec8)
  unsigned t = (unsigned)i8;
  const int co = (int)(t % (unsigned)p.cob); t /= (unsigned)p.cob;
  const int g = (int)(t & 3); t >>= 2;
  const int jj = (int)(t % 3u); t /= 3u;
  const int cb = (int)(t / (unsigned)p.nstages);
  const int stage = (int)(t - (unsigned)cb * (unsigned)p.nstages);
  int s = 0;
  while (s + 1 < p.nsrc && stage >= p.src_st0[s + 1]) ++s;
  const int j = (stage - p.src_st0[s]) * 3 + jj;
  int tap, chunk;
  ws_slot(p.src_ch[s] >> 3, j, g, tap, chunk);
  const int col = cb * p.cob + co;
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)0.f;
  if (tap >= 0 && col < p.on) {
    const int kc = p.src_off[s] + chunk * 8;
    const int oc = p.o0 + col;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)pack_weight_value(p, oc, kc + e, tap, 9);
  }
  *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.out) + i8 * 8) = o;
}

// ---- 3x3, one group: ONE THREAD PER (cout block, 32-channel block, chunk g, output channel co) -- an "item" = the 8 K-side channels x 9 taps of one chunk.
// The vector form above reads a packed vector's 8 channels with a stride of 9 floats (and the lanes of a wave sit 4 KB apart): every 64-byte line of the weight is
// touched by nine different instructions, 4 bytes at a time -- 0.9 TB/s.  An item reads its 72 floats as 18 aligned 16-byte loads (forward packs: 288 contiguous
// bytes; data-gradient packs: eight 36-byte tap rows, adjacent lanes adjacent rows) and writes nine 16-byte vectors, adjacent lanes adjacent vectors.
__device__ __forceinline__ int pack3_blocks(const PackK& p, int s) { return ((p.src_ch[s] >> 3) + 3) >> 2; }

template <typename T, bool WS>
__device__ __forceinline__ void pack3_store(const PackK& p, long long elem, const float (&v)[8]) {
  if constexpr (std::is_same<T, bf16>::value) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.out) + elem) = o;
  } else {
    f32x4* o = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + elem);
    o[0] = f32x4{v[0], v[1], v[2], v[3]};
    o[1] = f32x4{v[4], v[5], v[6], v[7]};
  }
}

template <typename T, bool WS>
__device__ __forceinline__ void pack3_item(const PackK& p, long long item, long long items) {
  const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (item >= items) {  // (standard layout only) the zero padding between a stage's body and its 4-KiB-aligned stride
    const int body8 = 3 * 4 * p.cob, pad8 = p.ss_elems / 8 - body8;
    const unsigned r = (unsigned)(item - items);
    const unsigned stage = r / (unsigned)pad8;
    pack3_store<T, WS>(p, (long long)stage * p.ss_elems + (long long)(body8 + (int)(r - stage * (unsigned)pad8)) * 8, zero8);
    return;
  }
  unsigned t = (unsigned)item / (unsigned)p.cob;  // (32-bit index arithmetic, see pack_std_vec8)
  const int co = (int)((unsigned)item - t * (unsigned)p.cob);
  const int g = (int)(t & 3);
  t >>= 2;
  int nblk = 0;
  for (int s = 0; s < p.nsrc; ++s) nblk += pack3_blocks(p, s);
  const int cb = (int)(t / (unsigned)nblk), blk = (int)(t - (unsigned)cb * (unsigned)nblk);
  int s = 0, b0 = 0;
  while (s + 1 < p.nsrc && blk >= b0 + pack3_blocks(p, s)) { b0 += pack3_blocks(p, s); ++s; }
  const int cbk = blk - b0, CH = p.src_ch[s] >> 3, nfull = CH >> 2;
  const bool rem = WS && cbk == nfull;  // the weight-streaming layout's remainder block (CH % 4 == 2: two chunks, two taps per k-step)
  if (rem && g >= 2) return;
  const int q = 4 * cbk + g, col = cb * p.cob + co;
  const bool valid = q < CH && col < p.on;
  float v[9][8];  // [tap][channel of the chunk]
  if (valid) {
    const int kc = p.src_off[s] + q * 8, oc = p.o0 + col;
    if (!p.transpose_flip) {
      const f32x4* src = reinterpret_cast<const f32x4*>(p.w + ((long long)oc * p.I + kc) * 9);  // 72 contiguous floats, 16-byte aligned (kc, I multiples of 8; the host checks w)
      float buf[72];
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        const f32x4 x = src[i];
        buf[4 * i] = x[0]; buf[4 * i + 1] = x[1]; buf[4 * i + 2] = x[2]; buf[4 * i + 3] = x[3];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) v[tap][e] = buf[e * 9 + tap];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float* src = p.w + ((long long)(kc + e) * p.I + oc) * 9;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) v[tap][e] = src[8 - tap];
      }
    }
  } else {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[tap][e] = 0.f;
  }
  const int st0 = p.src_st0[s];
  if (!WS) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - 3 * ky;
      pack3_store<T, WS>(p, ((long long)cb * p.nstages + st0 + cbk * 3 + ky) * p.ss_elems + (long long)((kx * 4 + g) * p.cob + co) * 8, v[tap]);
    }
  } else {
    auto slot = [&](int j, int gg) { return ((((long long)cb * p.nstages + st0 + j / 3) * 3 + j % 3) * 4 + gg) * p.cob * 8 + (long long)co * 8; };
    if (!rem) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) pack3_store<T, WS>(p, slot(cbk * 9 + tap, g), v[tap]);
    } else {  // chunk 4 nfull + g (g = 0, 1): k-step 9 nfull + tap / 2, lane group g | (tap & 1) << 1; the empty slots of its column are zero
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) pack3_store<T, WS>(p, slot(9 * nfull + (tap >> 1), g | ((tap & 1) << 1)), v[tap]);
      pack3_store<T, WS>(p, slot(9 * nfull + 4, g | 2), zero8);
      pack3_store<T, WS>(p, slot(9 * nfull + 5, g), zero8);
      pack3_store<T, WS>(p, slot(9 * nfull + 5, g | 2), zero8);
    }
  }
}

template <typename T, bool WS>
__global__ void pack3_kernel(const PackK p) {
  const long long items = (long long)p.units - (WS ? 0 : (long long)p.ncb * p.nstages * (p.ss_elems / 8 - 3 * 4 * p.cob));
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < p.units; i += (long long)gridDim.x * blockDim.x) pack3_item<T, WS>(p, i, items);
}

__global__ void convws_pack_kernel(const PackK p) {
  const long long total8 = (long long)p.ncb * p.nstages * 3 * 4 * p.cob;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) pack_ws_vec8(p, i);
}

// A PLAN of packs run by one launch (every weight of a network after an optimizer step): entry e owns blocks [blk0, blk0 + nblk).
struct PackEntry {
  PackK p;
  long long total;  // elements
  int kind;         // 0: vmg_conv_pack layout, 1: vmg_convws_pack layout
  int dtype;
  int blk0, nblk;
};

__global__ __launch_bounds__(256) void pack_batch_kernel(const PackEntry* __restrict__ plan, int n) {
  __shared__ PackEntry ent;
  // the entry that owns this block: a table of one int per block behind the n entries (round 4: a binary search over the entries' block ranges was ten DEPENDENT
  // global loads by one thread, ~10 us before the block's first useful instruction)
  const int which = reinterpret_cast<const int*>(plan + n)[blockIdx.x];
  {
    const int* src = reinterpret_cast<const int*>(plan + which);
    int* dst = reinterpret_cast<int*>(&ent);
    for (int k = threadIdx.x; k < (int)(sizeof(PackEntry) / 4); k += 256) dst[k] = src[k];
  }
  __syncthreads();
  const long long stride = (long long)ent.nblk * 256, total8 = ent.total / 8;  // (every layout is made of whole 8-element vectors)
  long long i = (long long)((int)blockIdx.x - ent.blk0) * 256 + threadIdx.x;
  if (ent.p.fast3) {
    const long long units = ent.p.units, items = units - (ent.kind == 1 ? 0 : (long long)ent.p.ncb * ent.p.nstages * (ent.p.ss_elems / 8 - 3 * 4 * ent.p.cob));
    if (ent.kind == 1) { for (; i < units; i += stride) pack3_item<bf16, true>(ent.p, i, items); }
    else if (ent.dtype == VMG_BF16) { for (; i < units; i += stride) pack3_item<bf16, false>(ent.p, i, items); }
    else { for (; i < units; i += stride) pack3_item<float, false>(ent.p, i, items); }
    return;
  }
  if (ent.kind == 1) { for (; i < total8; i += stride) pack_ws_vec8(ent.p, i); }
  else if (ent.dtype == VMG_BF16) { for (; i < total8; i += stride) pack_std_vec8<bf16>(ent.p, i); }
  else { for (; i < total8; i += stride) pack_std_vec8<float>(ent.p, i); }
}

// internal channel-block splitting: the SAME rule for packing and for the conv call
int expand_sources(int nsrc, const int* off, const int* ch, short* xoff, short* xch, int* parent) {
  int k = 0;
  for (int s = 0; s < nsrc; ++s) {
    int parts = 1;
    while (ch[s] / parts > CBMAX || ch[s] % (8 * parts) != 0) {
      ++parts;
      if (parts > ch[s] / 8) return -1;
    }
    const int c = ch[s] / parts;
    for (int j = 0; j < parts; ++j) {
      if (k >= MAX_ISRC) return -1;
      xoff[k] = (short)((off ? off[s] : 0) + j * c);
      xch[k] = (short)c;
      if (parent) parent[k] = s;
      ++k;
    }
  }
  return k;
}

// smallest LDS pixel stride (bytes) >= ch*ES that keeps the 16-pixel ds_read_b128 lane groups conflict-free
int choose_pixb(int ch, int es, int cb) {
  static const int grp[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                 {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                 {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                 {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  int best = ch * es, bestc = 1 << 30;
  for (int pad = 0; pad <= 240; pad += 16) {
    const int pixb = ch * es + pad;
    int worst = 0;
    for (int gi = 0; gi < 4; ++gi) {
      int cnt[16] = {0};
      for (int j = 0; j < 16; ++j) {
        const int l = grp[gi][j];
        const int addr = (l & 15) * pixb + (l >> 4) * cb;
        int c = ++cnt[(addr >> 4) & 15];
        if (c > worst) worst = c;
      }
    }
    if (worst < bestc) { bestc = worst; best = pixb; }
    if (worst == 1) break;
  }
  return best;
}

template <typename T, int KS, int MT, int NTB, bool DEEP>
int launch_conv(const ConvK& k, int ncb, hipStream_t st) {
  constexpr int CB = ElemTraits<T>::CHUNKB;
  const int lds = k.halo_bytes + (DEEP ? 3 : 2) * stage_stride(KS, NTB, CB);
  VMG_CHECK(lds <= 160 * 1024, "conv: LDS request %d B exceeds 160 KiB", lds);
  auto fn = conv_igemm_kernel<T, KS, MT, NTB, DEEP>;
  static bool attr_set[VMG_MAX_DEVICES] = {};  // the attribute is per device (one code object per device)
  const int dev = vmg_current_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = true;
  }
  long long nblk = (KS > 1) ? (long long)k.N * k.tiles_y * k.tiles_x : cdiv64(k.M, 64 * MT);
  VMG_CHECK(nblk > 0 && nblk < (1ll << 31), "conv: bad grid %lld", nblk);
  // the dominant kernel class of the path: bf16 3x3, one source of 144 channels, 144 outputs (trajectory chains)
  const bool prof = (KS == 3 && sizeof(T) == 2 && k.nsrc == 1 && k.src_ch[0] == 144 && k.Cout == 144) && vmg_prof_before(VMG_PROF_CONV3X3, k.M, st);
  hipLaunchKernelGGL(fn, dim3((unsigned)nblk, ncb), dim3(256), lds, st, k);
  if (prof) vmg_prof_after(st);
  VMG_LAUNCH_CHECK();
  return 0;
}

template <typename T, int KS, int MT, bool DEEP>
int dispatch_ntb(const ConvK& k, int ntb, int ncb, hipStream_t st) {
  switch (ntb) {
    case 1: return launch_conv<T, KS, MT, 1, DEEP>(k, ncb, st);
    case 3: return launch_conv<T, KS, MT, 3, DEEP>(k, ncb, st);
    case 4: return launch_conv<T, KS, MT, 4, DEEP>(k, ncb, st);
    case 5: return launch_conv<T, KS, MT, 5, DEEP>(k, ncb, st);
    case 7: return launch_conv<T, KS, MT, 7, DEEP>(k, ncb, st);
    case 8: return launch_conv<T, KS, MT, 8, DEEP>(k, ncb, st);
    case 9: return launch_conv<T, KS, MT, 9, DEEP>(k, ncb, st);
  }
  vmg_set_error("conv: cout_tiles must be 1, 3, 4, 5, 7, 8 or 9 (got %d)", ntb);
  return -1;
}

// 7x7 (SPyNet's convs, models/vmg.py:126-173): the pixel-split kernel with 1, 2 or 4 output-channel tiles per workgroup
template <typename T>
int dispatch_ks7(const ConvK& k, int ntb, int ncb, hipStream_t st) {
  switch (ntb) {
    case 1: return launch_conv<T, 7, 1, 1, false>(k, ncb, st);
    case 2: return launch_conv<T, 7, 1, 2, false>(k, ncb, st);
    case 4: return launch_conv<T, 7, 1, 4, false>(k, ncb, st);
  }
  vmg_set_error("conv (7x7): cout_tiles must be 1, 2 or 4 (got %d)", ntb);
  return -1;
}

template <typename T, int KS, int NTB>
int launch_ksplit(const ConvK& k, int ncb, int halo_total, hipStream_t st) {
  const int scratch = 12 * NTB * 1024;
  const int lds = (halo_total > scratch ? halo_total : scratch) + 16 + NTB * 16 * 4;
  VMG_CHECK(lds <= 160 * 1024, "conv (k-split): LDS request %d B exceeds 160 KiB", lds);
  auto fn = conv_ksplit_kernel<T, KS, NTB>;
  static bool attr_set[VMG_MAX_DEVICES] = {};  // the attribute is per device (one code object per device)
  const int dev = vmg_current_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = true;
  }
  long long nblk = (KS > 1) ? (long long)k.N * k.tiles_y * k.tiles_x : cdiv64(k.M, 64);
  VMG_CHECK(nblk > 0 && nblk < (1ll << 31), "conv: bad grid %lld", nblk);
  ConvK kk = k;
  kk.halo_bytes = lds - 16 - NTB * 16 * 4;  // zero slot and bias sit behind max(halo, scratch)
  const bool prof = (KS == 3 && sizeof(T) == 2 && k.nsrc == 1 && k.src_ch[0] == k.Cout && (k.Cout == 144 || k.Cout == 112)) && vmg_prof_before(VMG_PROF_CONV3X3, k.M, st);
  hipLaunchKernelGGL(fn, dim3((unsigned)nblk, ncb), dim3(256), lds, st, kk);
  if (prof) vmg_prof_after(st);
  VMG_LAUNCH_CHECK();
  return 0;
}

template <typename T, int KS>
int dispatch_ksplit(const ConvK& k, int ntb, int ncb, int halo_total, hipStream_t st) {
  switch (ntb) {
    case 1: return launch_ksplit<T, KS, 1>(k, ncb, halo_total, st);
    case 3: return launch_ksplit<T, KS, 3>(k, ncb, halo_total, st);
    case 4: return launch_ksplit<T, KS, 4>(k, ncb, halo_total, st);
    case 5: return launch_ksplit<T, KS, 5>(k, ncb, halo_total, st);
  }
  vmg_set_error("conv (k-split): cout_tiles must be 1, 3, 4 or 5 (got %d)", ntb);
  return -1;
}

template <int NTB, int NS = 1>
int launch_linear_wres(const ConvK& k, int ncb, hipStream_t st) {
  constexpr int WAVE_LDS = 2 * 5 * NS * 1024 + ((EpiLds<NTB>::BYTES + 1023) & ~1023);
  const int lds = 4 * WAVE_LDS;
  auto fn = linear_wres_kernel<NTB, NS>;
  static bool attr_set[VMG_MAX_DEVICES] = {};
  const int dev = vmg_current_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = true;
  }
  const int ncu = vmg_cu_count(dev);
  const long long ntiles = cdiv64(k.M, 16);
  VMG_CHECK(ntiles > 0 && ntiles < (1ll << 31), "conv: bad tile count %lld", ntiles);
  // two workgroups per CU over all output-channel blocks; at least 4 tiles per wave so the resident weights pay off
  constexpr int WG_PER_CU = NS == 1 ? 2 : 1;
  long long waves = (long long)(WG_PER_CU * ncu / ncb > 0 ? WG_PER_CU * ncu / ncb : 1) * 4;
  long long tpw = cdiv64(ntiles, waves);
  if (tpw < 4) tpw = 4;
  const long long nwg = cdiv64(cdiv64(ntiles, tpw), 4);
  hipLaunchKernelGGL(fn, dim3((unsigned)nwg, ncb), dim3(256), lds, st, k, (int)ntiles, (int)tpw);
  VMG_LAUNCH_CHECK();
  return 0;
}

template <int NCT, int NB, bool FULL>
int launch_wstat2(const ConvK& k, int ncb, hipStream_t st) {
  constexpr int SS = stage_stride(3, NCT, 16);
  constexpr int HB = ((10 * 18 * 9 + 63) / 64) * 1024;
  constexpr int lds = 3 * NB * SS + 2 * HB + NCT * 16 * 4 + 128 * (NCT * 64 + 16);
  static_assert(lds <= 160 * 1024, "conv_wstat_kernel: LDS");
  VMG_CHECK(k.nstages == 3 * NB, "conv (weights-stationary): %d stages in the pack, %d expected", k.nstages, 3 * NB);
  auto fn = conv_wstat_kernel<NCT, NB, FULL>;
  static bool attr_set[VMG_MAX_DEVICES] = {};
  const int dev = vmg_current_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = true;
  }
  const long long ntiles = (long long)k.N * k.tiles_y * k.tiles_x;
  VMG_CHECK(ntiles > 0 && ntiles < (1ll << 30), "conv: bad tile count %lld", ntiles);
  int nwg = vmg_cu_count(dev) & ~7;  // one workgroup per CU, a multiple of the 8 XCDs
  if (nwg < 8) nwg = 8;
  while (nwg > 8 && (long long)(nwg >> 3) > cdiv64(ntiles, 8)) nwg -= 8;  // fewer tiles than workgroups
  hipLaunchKernelGGL(fn, dim3((unsigned)nwg, ncb), dim3(512), lds, st, k, (int)ntiles);
  VMG_LAUNCH_CHECK();
  return 0;
}

template <int NCT, int NB>
int launch_wstat(const ConvK& k, int ncb, hipStream_t st) {
  const bool full = k.res || k.aux || k.out_pre || k.act == VMG_ACT_GELU;
  return full ? launch_wstat2<NCT, NB, true>(k, ncb, st) : launch_wstat2<NCT, NB, false>(k, ncb, st);
}

template <int NCT>
int launch_ws(const ConvK& k, int ncb, hipStream_t st) {
  using G = WsGeo<NCT>;
  ConvK kk = k;
  // NCT = 3 (48-channel blocks, three workgroups per tile): three ring slots, <= 80 KiB in all, so that TWO workgroups share a CU -- one's halo
  // burst and store tail run under the other's K loop.  For launches with several tiles per CU (functional.choose_tiling).
  kk.ring = NCT <= 3 ? 3 : ((k.halo_bytes + 4 * G::STG + G::COB * 4 <= 160 * 1024) ? 4 : 3);
  const int lds = k.halo_bytes + kk.ring * G::STG + G::COB * 4;
  VMG_CHECK(lds <= 160 * 1024, "conv (weight-streaming): LDS request %d B exceeds 160 KiB", lds);
  auto fn = conv_ws_kernel<NCT>;
  static bool attr_set[VMG_MAX_DEVICES] = {};
  const int dev = vmg_current_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = true;
  }
  const long long nblk = (long long)k.N * k.tiles_y * k.tiles_x;
  VMG_CHECK(nblk > 0 && nblk < (1ll << 31), "conv: bad grid %lld", nblk);
  const bool prof = (k.nsrc == 1 && k.src_ch[0] == k.Cout && (k.Cout == 144 || k.Cout == 112)) && vmg_prof_before(VMG_PROF_CONV3X3, k.M, st);
  hipLaunchKernelGGL(fn, dim3((unsigned)nblk, ncb), dim3(G::THREADS), lds, st, kk);
  if (prof) vmg_prof_after(st);
  VMG_LAUNCH_CHECK();
  return 0;
}

}  // namespace

#ifdef VMG_DIAG
static unsigned long long* g_conv_stamps = nullptr;
#endif
extern "C" int vmg_conv_debug_stamps(void* buf) {  // diagnostics builds only (tools/conv_timeline.py); buf: 8 * 4 * workgroups uint64, or null
#ifdef VMG_DIAG
  g_conv_stamps = (unsigned long long*)buf;
  return 0;
#else
  (void)buf;
  vmg_set_error("conv_debug_stamps: this library was built without -DVMG_DIAG (VMG_DIAG=1 python -m vmg_amd.build)");
  return -1;
#endif
}

extern "C" int64_t vmg_conv_pack_bytes(int dtype, int ks, int on, int nsrc, const int* src_ch, int cout_tiles) {
  short xoff[MAX_ISRC], xch[MAX_ISRC];
  const int n = expand_sources(nsrc, nullptr, src_ch, xoff, xch, nullptr);
  if (n < 0 || cout_tiles <= 0 || (ks != 1 && ks != 3 && ks != 7)) return -1;
  int64_t nst = 0;
  for (int s = 0; s < n; ++s) nst += stages_of(ks, xch[s]);
  const int cob = cout_tiles * 16;
  const int ncb = (on + cob - 1) / cob;
  return (int64_t)ncb * nst * stage_stride(ks, cout_tiles, dtype == VMG_BF16 ? 16 : 32);
}

static int fill_pack_std(PackK& p, long long& total, int dtype, const float* w, int O, int I, int ks, int o0, int on, int nsrc, const int* src_off,
                         const int* src_ch, int transpose_flip, int cout_tiles, void* packed) {
  VMG_CHECK(w && packed, "conv_pack: null pointer");
  VMG_CHECK(ks == 1 || ks == 3 || ks == 7, "conv_pack: ks must be 1, 3 or 7");
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "conv_pack: bad dtype");
  VMG_CHECK(nsrc >= 1 && nsrc <= 4, "conv_pack: nsrc must be 1..4");
  VMG_CHECK(!(transpose_flip & 1) || nsrc == 1, "conv_pack: data-gradient packing takes one K slice");
  VMG_CHECK(cout_tiles >= 1 && cout_tiles <= 9, "conv_pack: cout_tiles must be 1..9");
  memset(&p, 0, sizeof(p));
  const int n = expand_sources(nsrc, src_off, src_ch, p.src_off, p.src_ch, nullptr);
  VMG_CHECK(n > 0, "conv_pack: channel slices must be multiples of 8 (and split into <= %d blocks)", MAX_ISRC);
  int st = 0;
  for (int s = 0; s < n; ++s) {
    p.src_st0[s] = (short)st;
    p.src_nst[s] = (short)stages_of(ks, p.src_ch[s]);
    st += p.src_nst[s];
  }
  // bounds of the slices against the weight tensor
  const int groups = (transpose_flip >> 8) > 1 ? (transpose_flip >> 8) : 1;  // (bits 8.. of the flag word: see vmg_hip.h)
  transpose_flip &= 1;
  VMG_CHECK(groups == 1 || (O % groups == 0), "conv_pack: output channels must divide into the groups");
  const int Id = I * groups;  // dense input channels
  const int kdim = transpose_flip ? O : Id, odim = transpose_flip ? Id : O;
  for (int s = 0; s < n; ++s) VMG_CHECK(p.src_off[s] >= 0 && p.src_off[s] + p.src_ch[s] <= kdim, "conv_pack: K slice out of range");
  VMG_CHECK(o0 >= 0 && o0 + on <= odim, "conv_pack: output slice out of range");
  p.w = w; p.out = (char*)packed; p.O = O; p.I = I; p.ks = ks; p.o0 = o0; p.on = on; p.nsrc = n;
  p.transpose_flip = transpose_flip; p.groups = groups; p.cob = cout_tiles * 16; p.ncb = (on + p.cob - 1) / p.cob; p.nstages = st;
  const int es = dtype == VMG_BF16 ? 2 : 4;
  p.ss_elems = stage_stride(ks, cout_tiles, es * 8) / es;
  total = (long long)p.ncb * p.nstages * p.ss_elems;
  VMG_CHECK(total < (1ll << 31), "conv_pack: pack of %lld elements (the index arithmetic is 32-bit)", total);
  if (ks == 3 && groups == 1 && (transpose_flip || (uintptr_t)w % 16 == 0) && (transpose_flip || I % 4 == 0)) {  // the item form (pack3_item)
    int nblk = 0;
    for (int s = 0; s < n; ++s) nblk += ((p.src_ch[s] >> 3) + 3) >> 2;
    p.fast3 = 1;
    p.units = (long long)p.ncb * nblk * 4 * p.cob + (long long)p.ncb * p.nstages * (p.ss_elems / 8 - 3 * 4 * p.cob);
  }
  return 0;
}

extern "C" int vmg_conv_pack(int dtype, const float* w, int O, int I, int ks, int o0, int on, int nsrc, const int* src_off,
                             const int* src_ch, int transpose_flip, int cout_tiles, void* packed, void* stream) {
  PackK p;
  long long total = 0;
  const int rc = fill_pack_std(p, total, dtype, w, O, I, ks, o0, on, nsrc, src_off, src_ch, transpose_flip, cout_tiles, packed);
  if (rc) return rc;
  if (p.fast3) {
    const int fb = (int)((p.units + 255) / 256 > 4096 ? 4096 : (p.units + 255) / 256);
    if (dtype == VMG_BF16) hipLaunchKernelGGL((pack3_kernel<bf16, false>), dim3(fb), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((pack3_kernel<float, false>), dim3(fb), dim3(256), 0, (hipStream_t)stream, p);
    VMG_LAUNCH_CHECK();
    return 0;
  }
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if (dtype == VMG_BF16) hipLaunchKernelGGL(conv_pack_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(conv_pack_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  VMG_LAUNCH_CHECK();
  return 0;
}

static int ws_plan(int nsrc, const int* src_off, const int* src_ch, PackK& p) {
  const int n = expand_sources(nsrc, src_off, src_ch, p.src_off, p.src_ch, nullptr);
  if (n <= 0) return -1;
  int st = 0;
  for (int s = 0; s < n; ++s) {
    p.src_st0[s] = (short)st;
    p.src_nst[s] = (short)(ws_ksteps(p.src_ch[s] >> 3) / 3);
    st += p.src_nst[s];
  }
  p.nsrc = n;
  p.nstages = st;
  return n;
}

extern "C" int64_t vmg_convws_pack_bytes(int on, int nsrc, const int* src_ch, int cout_tiles) {
  PackK p;
  memset(&p, 0, sizeof(p));
  if (cout_tiles <= 0 || cout_tiles > 9 || on <= 0 || nsrc < 1 || nsrc > 4 || ws_plan(nsrc, nullptr, src_ch, p) <= 0) return -1;
  const int cob = cout_tiles * 16;
  return (int64_t)cdiv(on, cob) * p.nstages * 3 * (int64_t)cob * 64;
}

static int fill_pack_ws(PackK& p, long long& total, const float* w, int O, int I, int o0, int on, int nsrc, const int* src_off, const int* src_ch,
                        int transpose_flip, int cout_tiles, void* packed) {
  VMG_CHECK(w && packed, "convws_pack: null pointer");
  VMG_CHECK(cout_tiles == 3 || cout_tiles == 7 || cout_tiles == 8 || cout_tiles == 9, "convws_pack: cout_tiles must be 3, 7, 8 or 9");
  VMG_CHECK(nsrc >= 1 && nsrc <= 4, "convws_pack: nsrc must be 1..4");
  VMG_CHECK(!(transpose_flip & 1) || nsrc == 1, "convws_pack: data-gradient packing takes one K slice");
  const int groups = (transpose_flip >> 8) > 1 ? (transpose_flip >> 8) : 1;
  transpose_flip &= 1;
  VMG_CHECK(groups == 1 || (O % groups == 0), "convws_pack: output channels must divide into the groups");
  memset(&p, 0, sizeof(p));
  VMG_CHECK(ws_plan(nsrc, src_off, src_ch, p) > 0, "convws_pack: channel slices must be multiples of 8 (and split into <= %d blocks)", MAX_ISRC);
  const int Id = I * groups;
  const int kdim = transpose_flip ? O : Id, odim = transpose_flip ? Id : O;
  for (int s = 0; s < p.nsrc; ++s) VMG_CHECK(p.src_off[s] >= 0 && p.src_off[s] + p.src_ch[s] <= kdim, "convws_pack: K slice out of range");
  VMG_CHECK(o0 >= 0 && on > 0 && o0 + on <= odim, "convws_pack: output slice out of range");
  p.w = w; p.out = (char*)packed; p.O = O; p.I = I; p.ks = 3; p.o0 = o0; p.on = on; p.transpose_flip = transpose_flip; p.groups = groups;
  p.cob = cout_tiles * 16; p.ncb = cdiv(on, p.cob);
  total = (long long)p.ncb * p.nstages * 3 * 4 * p.cob * 8;
  VMG_CHECK(total < (1ll << 31), "convws_pack: pack of %lld elements (the index arithmetic is 32-bit)", total);
  bool rem_ok = true;  // the item form knows the remainder blocks of 0 or 2 chunks (all the conv kernel admits)
  for (int s = 0; s < p.nsrc; ++s) rem_ok = rem_ok && (((p.src_ch[s] >> 3) & 3) == 0 || ((p.src_ch[s] >> 3) & 3) == 2);
  if (groups == 1 && rem_ok && (transpose_flip || ((uintptr_t)w % 16 == 0 && I % 4 == 0))) {
    int nblk = 0;
    for (int s = 0; s < p.nsrc; ++s) nblk += ((p.src_ch[s] >> 3) + 3) >> 2;
    p.fast3 = 1;
    p.units = (long long)p.ncb * nblk * 4 * p.cob;
  }
  return 0;
}

extern "C" int vmg_convws_pack(const float* w, int O, int I, int o0, int on, int nsrc, const int* src_off, const int* src_ch,
                               int transpose_flip, int cout_tiles, void* packed, void* stream) {
  PackK p;
  long long total = 0;
  const int rc = fill_pack_ws(p, total, w, O, I, o0, on, nsrc, src_off, src_ch, transpose_flip, cout_tiles, packed);
  if (rc) return rc;
  if (p.fast3) {
    const int fb = (int)((p.units + 255) / 256 > 4096 ? 4096 : (p.units + 255) / 256);
    hipLaunchKernelGGL((pack3_kernel<bf16, true>), dim3(fb), dim3(256), 0, (hipStream_t)stream, p);
    VMG_LAUNCH_CHECK();
    return 0;
  }
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(convws_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_pack_entry_bytes(void) { return (int)sizeof(PackEntry); }

extern "C" int vmg_pack_entry(void* entry, int kind, int dtype, const float* w, int O, int I, int ks, int o0, int on, int nsrc, const int* src_off,
                              const int* src_ch, int transpose_flip, int cout_tiles, void* packed, int blk0) {
  VMG_CHECK(entry && (kind == 0 || kind == 1) && blk0 >= 0, "pack_entry: bad arguments");
  PackEntry e;
  memset(&e, 0, sizeof(e));
  int rc;
  if (kind == 1) {
    VMG_CHECK(ks == 3 && dtype == VMG_BF16, "pack_entry: the weight-streaming layout is bf16 3x3");
    rc = fill_pack_ws(e.p, e.total, w, O, I, o0, on, nsrc, src_off, src_ch, transpose_flip, cout_tiles, packed);
  } else {
    rc = fill_pack_std(e.p, e.total, dtype, w, O, I, ks, o0, on, nsrc, src_off, src_ch, transpose_flip, cout_tiles, packed);
  }
  if (rc) return rc;
  e.kind = kind; e.dtype = dtype; e.blk0 = blk0;
  long long nb = e.p.fast3 ? (e.p.units + 255) / 256 : (e.total + 256LL * 16 - 1) / (256LL * 16);  // an item (72 elements) / ~16 elements per thread
  e.nblk = (int)(nb < 1 ? 1 : (nb > 64 ? 64 : nb));
  memcpy(entry, &e, sizeof(e));
  return e.nblk;
}

extern "C" int vmg_pack_run(const void* plan_dev, int n, int total_blocks, void* stream) {
  VMG_CHECK(plan_dev && n > 0 && total_blocks > 0, "pack_run: bad arguments");
  static_assert(sizeof(PackEntry) % 4 == 0, "the block-owner table sits behind the entries");
  hipLaunchKernelGGL(pack_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const PackEntry*)plan_dev, n);
  VMG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmg_conv_fwd(const vmg_conv_desc* d, void* stream) {
  VMG_CHECK(d != nullptr, "conv_fwd: null descriptor");
  VMG_CHECK(d->ks == 1 || d->ks == 3 || d->ks == 7, "conv_fwd: ks must be 1, 3 or 7");
  VMG_CHECK(d->dtype == VMG_F32 || d->dtype == VMG_BF16, "conv_fwd: bad dtype");
  VMG_CHECK(d->nsrc >= 1 && d->nsrc <= 4, "conv_fwd: nsrc must be 1..4");
  VMG_CHECK(d->N > 0 && d->H > 0 && d->W > 0 && d->Cout > 0, "conv_fwd: bad shape");
  VMG_CHECK(d->packed && d->out, "conv_fwd: null packed weights / output");
  VMG_CHECK(!d->pixel_shuffle || (d->Cout % 4 == 0), "conv_fwd: pixel_shuffle needs Cout %% 4 == 0");
  VMG_CHECK(!(d->actgrad != 0 && d->aux == nullptr), "conv_fwd: actgrad without aux");
  const int es = d->dtype == VMG_BF16 ? 2 : 4, cbytes = es * 8;
  ConvK k;
  memset(&k, 0, sizeof(k));
  short xoff[MAX_ISRC];
  int parent[MAX_ISRC];
  const int n = expand_sources(d->nsrc, nullptr, d->src_ch, xoff, k.src_ch, parent);
  VMG_CHECK(n > 0, "conv_fwd: source channel counts must be multiples of 8");
  int mt = d->mt;
  if (d->dtype == VMG_F32) mt = 1;
  const long long M = (long long)d->N * d->H * d->W;
  const int ntb = d->cout_tiles;
  VMG_CHECK(ntb > 0, "conv_fwd: cout_tiles must be positive");
  const int ncb = cdiv(d->Cout, ntb * 16);
  if (d->ks == 7) mt = 1;
  if (mt == 0) mt = 1;  // measured (tools/bench_conv.py): one 16-pixel row per wave keeps 2 workgroups per CU and wins everywhere
  VMG_CHECK(mt == 1 || mt == 2, "conv_fwd: mt must be 1 or 2");
  const int TH = 4 * mt, TWH = 16 + d->ks - 1, THH = TH + d->ks - 1;
  int kt = 0, halo = 0;
  for (int s = 0; s < n; ++s) {
    const int par = parent[s];
    VMG_CHECK(d->src[par] != nullptr, "conv_fwd: null source %d", par);
    VMG_CHECK(d->src_ps[par] >= d->src_ch[par] && (d->src_ps[par] * es) % 16 == 0, "conv_fwd: source pixel stride must be >= channels and 16-byte aligned");
    VMG_CHECK(((uintptr_t)d->src[par]) % 16 == 0, "conv_fwd: source pointer must be 16-byte aligned");
    k.src[s] = (const char*)d->src[par] + (long long)xoff[s] * es;
    k.src_ps[s] = d->src_ps[par];
    k.src_nst[s] = (short)stages_of(d->ks, k.src_ch[s]);
    k.src_pixb[s] = choose_pixb(k.src_ch[s], es, cbytes);
    kt += k.src_nst[s];
    const int hb = THH * TWH * k.src_pixb[s];
    if (hb > halo) halo = hb;
  }
  k.nsrc = n;
  k.wpack = (const char*)d->packed; k.bias = d->bias;
  k.out = (char*)d->out; k.out_ps = d->out_ps; k.out_pre = (char*)d->out_pre;
  k.res = (const char*)d->res; k.res_ps = d->res_ps; k.aux = (const char*)d->aux; k.aux_ps = d->aux_ps;
  k.N = d->N; k.H = d->H; k.W = d->W; k.Cout = d->Cout; k.M = M;
  k.tiles_x = cdiv(d->W, 16); k.tiles_y = cdiv(d->H, TH);
  k.act = d->act; k.slope = d->slope; k.alpha = d->alpha; k.actgrad = d->aux ? d->actgrad : 0; k.ps = d->pixel_shuffle;
#ifdef VMG_DIAG
  {
    const char* e = getenv("VMG_CONV_DBG");  // ablation bits for tools/conv_ablate.py; read per call so one process can sweep
    k.dbg = e ? atoi(e) : 0;
    k.stamps = g_conv_stamps;
  }
#endif
  {
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    k.vec8 = d->dtype == VMG_BF16 && !d->pixel_shuffle && (d->Cout % 16) == 0 && (d->out_ps % 8) == 0 && al(d->out) &&
             (!d->out_pre || al(d->out_pre)) && (!d->res || ((d->res_ps % 8) == 0 && al(d->res))) &&
             (!d->aux || ((d->aux_ps % 8) == 0 && al(d->aux)));
  }
  k.nstages = kt; k.halo_bytes = (halo + 1023) & ~1023;  // the LDS-DMA copy writes whole 1-KiB pieces
  VMG_CHECK(d->out_ps >= (d->pixel_shuffle ? d->Cout / 4 : d->Cout), "conv_fwd: out pixel stride too small");
  hipStream_t st = (hipStream_t)stream;
  if (d->deep == 3) {
    // weight-streaming kernel: 8 x 16 pixel tiles, dense halo, its own packed layout (vmg_convws_pack)
    VMG_CHECK(d->dtype == VMG_BF16 && d->ks == 3 && k.vec8 && (ntb == 3 || ntb == 7 || ntb == 8 || ntb == 9), "conv_fwd: the weight-streaming kernel is bf16 3x3 with 16-byte aligned rows, cout_tiles 3, 7, 8 or 9");
    for (int s = 0; s < n; ++s) VMG_CHECK(k.src_ch[s] % 16 == 0, "conv_fwd: the weight-streaming kernel takes channel blocks that are multiples of 16 (got %d)", k.src_ch[s]);
    int kt3 = 0, halo3 = 0;
    for (int s = 0; s < n; ++s) {
      k.src_pixb[s] = k.src_ch[s] * 2;
      k.src_nst[s] = (short)(ws_ksteps(k.src_ch[s] >> 3) / 3);
      kt3 += k.src_nst[s];
      const int hb = 10 * 18 * k.src_pixb[s];
      if (hb > halo3) halo3 = hb;
    }
    k.nstages = kt3;
    k.halo_bytes = (halo3 + 1023) & ~1023;  // whole 1-KiB LDS-DMA pieces
    k.tiles_y = cdiv(d->H, 8);
    return ntb == 9 ? launch_ws<9>(k, ncb, st) : (ntb == 7 ? launch_ws<7>(k, ncb, st) : (ntb == 8 ? launch_ws<8>(k, ncb, st) : launch_ws<3>(k, ncb, st)));
  }
  if (d->deep == 6 && d->dtype == VMG_BF16 && d->ks == 3 && mt == 1 && k.vec8 && n == 1 && (ntb == 1 || ntb == 3 || ntb == 4) && k.src_ch[0] <= 64 && k.src_ch[0] % 8 == 0 &&
      k.src_ps[0] % 8 == 0 && d->W <= 255 * 16 && (long long)d->W * k.src_ps[0] * 2 * 10 < (1ll << 31)) {
    // weights-stationary kernel (a hint: anything it does not cover takes the general kernel below); its tiles are 8 rows high
    ConvK kk = k;
    kk.tiles_y = cdiv(d->H, 8);
    const int nb = (k.src_ch[0] / 8 + 3) / 4;  // 32-channel blocks: 1 or 2
    switch (ntb * 2 + (nb - 1)) {
      case 2: return launch_wstat<1, 1>(kk, ncb, st);
      case 3: return launch_wstat<1, 2>(kk, ncb, st);
      case 6: return launch_wstat<3, 1>(kk, ncb, st);
      case 7: return launch_wstat<3, 2>(kk, ncb, st);
      case 8: return launch_wstat<4, 1>(kk, ncb, st);
      case 9: return launch_wstat<4, 2>(kk, ncb, st);
    }
  }
  if (d->deep == 4 && d->dtype == VMG_BF16 && d->ks == 1 && k.vec8 && (ntb == 3 || ntb == 5)) {
    // (a hint: anything it does not cover -- padded LDS stride, several sources, unaligned rows -- takes the general kernel below)
    if (n == 1 && k.src_ch[0] <= 160 && k.src_pixb[0] == k.src_ch[0] * 2) return ntb == 3 ? launch_linear_wres<3>(k, ncb, st) : launch_linear_wres<5>(k, ncb, st);
    // one wide source that the pack split into two equal blocks (288 = 2 x 144)
    if (n == 2 && ntb == 3 && d->nsrc == 1 && k.src_ch[0] == k.src_ch[1] && k.src_ch[0] <= 160 && k.src[1] == k.src[0] + k.src_ch[0] * 2 &&
        k.src_nst[0] == k.src_nst[1])
      return launch_linear_wres<3, 2>(k, ncb, st);
  }
  if (d->ks == 7) return d->dtype == VMG_BF16 ? dispatch_ks7<bf16>(k, ntb, ncb, st) : dispatch_ks7<float>(k, ntb, ncb, st);
  if (d->deep == 2) {
    VMG_CHECK(d->dtype == VMG_BF16 && mt == 1, "conv_fwd: the k-split variant is bf16, mt = 1");
    return d->ks == 3 ? dispatch_ksplit<bf16, 3>(k, ntb, ncb, k.halo_bytes, st) : dispatch_ksplit<bf16, 1>(k, ntb, ncb, k.halo_bytes, st);
  }
  const bool deep = d->deep == 1;
  if (d->dtype == VMG_BF16) {
    if (d->ks == 3) {
      if (deep) return mt == 2 ? dispatch_ntb<bf16, 3, 2, true>(k, ntb, ncb, st) : dispatch_ntb<bf16, 3, 1, true>(k, ntb, ncb, st);
      return mt == 2 ? dispatch_ntb<bf16, 3, 2, false>(k, ntb, ncb, st) : dispatch_ntb<bf16, 3, 1, false>(k, ntb, ncb, st);
    }
    if (deep) return mt == 2 ? dispatch_ntb<bf16, 1, 2, true>(k, ntb, ncb, st) : dispatch_ntb<bf16, 1, 1, true>(k, ntb, ncb, st);
    return mt == 2 ? dispatch_ntb<bf16, 1, 2, false>(k, ntb, ncb, st) : dispatch_ntb<bf16, 1, 1, false>(k, ntb, ncb, st);
  }
  if (d->ks == 3) return dispatch_ntb<float, 3, 1, false>(k, ntb, ncb, st);
  return dispatch_ntb<float, 1, 1, false>(k, ntb, ncb, st);
}

// ------------------------------------------------------------------------------------------------ residual chains
static void chain_base(const vmg_chain_desc* c, vmg_conv_desc& d) {
  memset(&d, 0, sizeof(d));
  d.dtype = c->dtype; d.ks = 3; d.N = c->N; d.H = c->H; d.W = c->W; d.Cout = c->C; d.alpha = 1.f;
  d.nsrc = 1; d.src_ps[0] = c->C; d.src_ch[0] = c->C; d.out_ps = c->C; d.cout_tiles = c->cout_tiles; d.deep = c->deep; d.mt = 1;
}

extern "C" int vmg_resblock_chain_fwd(const vmg_chain_desc* c, void* stream) {
  VMG_CHECK(c && c->nblk >= 0 && c->nsrc >= 1 && c->nsrc <= 4 && c->y && c->packed0, "resblock_chain_fwd: bad descriptor");
  VMG_CHECK(c->nblk == 0 || (c->t && c->packed1 && c->packed2 && c->bias1 && c->bias2), "resblock_chain_fwd: null block arrays");
  vmg_conv_desc d;
  chain_base(c, d);
  d.nsrc = c->nsrc;
  for (int s = 0; s < c->nsrc; ++s) { d.src[s] = c->src[s]; d.src_ps[s] = c->src_ps[s]; d.src_ch[s] = c->src_ch[s]; }
  d.packed = c->packed0; d.bias = c->bias0; d.out = c->y[0]; d.act = VMG_ACT_LRELU; d.slope = c->slope0;
  d.cout_tiles = c->cout_tiles0; d.deep = c->deep0;
  int rc = vmg_conv_fwd(&d, stream);
  if (rc) return rc;
  for (int k = 0; k < c->nblk; ++k) {
    chain_base(c, d);
    d.src[0] = c->y[k]; d.packed = c->packed1[k]; d.bias = c->bias1[k]; d.out = c->t[k]; d.act = VMG_ACT_RELU;
    if ((rc = vmg_conv_fwd(&d, stream))) return rc;
    chain_base(c, d);
    d.src[0] = c->t[k]; d.packed = c->packed2[k]; d.bias = c->bias2[k]; d.out = c->y[k + 1]; d.alpha = c->r_scaling;
    d.res = c->y[k]; d.res_ps = c->C;
    if ((rc = vmg_conv_fwd(&d, stream))) return rc;
  }
  return 0;
}

extern "C" int vmg_resblock_chain_bwd(const vmg_chain_desc* c, void* stream) {
  VMG_CHECK(c && c->nblk >= 0 && c->g_y, "resblock_chain_bwd: bad descriptor");
  VMG_CHECK(c->nblk == 0 || (c->t && c->g_t && c->packed1 && c->packed2), "resblock_chain_bwd: null block arrays");
  vmg_conv_desc d;
  for (int k = c->nblk - 1; k >= 0; --k) {
    // g_t[k] = r * dgrad2(g_y[k+1]) * relu'(t_k)
    chain_base(c, d);
    d.src[0] = c->g_y[k + 1]; d.packed = c->packed2[k]; d.out = c->g_t[k]; d.alpha = c->r_scaling; d.aux = c->t[k]; d.aux_ps = c->C; d.actgrad = 1;
    int rc = vmg_conv_fwd(&d, stream);
    if (rc) return rc;
    // g_y[k] = g_y[k+1] + dgrad1(g_t[k])
    chain_base(c, d);
    d.src[0] = c->g_t[k]; d.packed = c->packed1[k]; d.out = c->g_y[k]; d.res = c->g_y[k + 1]; d.res_ps = c->C;
    if ((rc = vmg_conv_fwd(&d, stream))) return rc;
  }
  return 0;
}
