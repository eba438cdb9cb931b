// Weight / bias gradient of the stride-1 "same" convolution (KS = 3 or 1), channels-last, gfx950.
//
//   dW[o][i][tap] += scale * sum_pixels dY[p][o] * X[p + tap][i]          (fp32, OIHW, atomically accumulated)
//
// GEMM view: M = output channels, N = (tap, input channel), K = pixels.  Both operands are stored pixel-major
// (channels-last), but the MFMA wants 8 consecutive K (= pixels) per lane, i.e. the TRANSPOSE of what is in
// memory.  gfx950's ds_read_b64_tr_b16 does that transpose on the LDS read path for 16-bit data: tiles are
// staged in LDS exactly as they lie in HBM ([pixel][channel], 16-byte vector copies) and each operand fragment
// is two transposed reads.  The tap shift is a row (pixel) offset of the transposed read, so one staged X tile
// with a 1-pixel halo serves all 9 taps.  fp32 uses v_mfma_f32_16x16x4_f32, whose operands are single
// elements (plain ds_read_b32).
//
// Work split: blockIdx.x = block of CT*16 output channels, blockIdx.y = block of IT*16 input channels,
// blockIdx.z = K split.  A K unit is 32 consecutive pixels of one image row.  Each of the 4 waves walks its own
// units through a wave-private LDS tile (no workgroup barriers in the main loop; next unit's global loads are in
// flight while the current one is multiplied).  At the end the 4 waves' accumulators are summed with LDS float
// atomics into a [co][ci][tap] image = the OIHW order, and written out with coalesced global float atomics.
#include <stdlib.h>
#include <type_traits>

#include "common.h"

namespace {

constexpr int WG_MAX_PAIRS = 16;  // (x, dy) pairs summed by one launch (uses of a shared weight)
__device__ __attribute__((aligned(256))) unsigned int g_zero_buf[64];  // 256 zero bytes: load / DMA source for out-of-image lanes

// LDS-DMA (1 KiB per wave: 16 bytes per lane from a per-lane global address into a lane-linear LDS block) as an asm statement: when a
// wave issues the BUILTIN, hipcc treats every later ds_read of that wave as possibly aliasing the copy and puts s_waitcnt vmcnt(0) in
// front of it -- the copies in flight (the whole prefetch ring) are then waited for before every fragment read, and copies and MFMAs
// run one after the other (measured on the 3x3 kernel: copies alone 39 us, MFMAs + reads 52 us, together 89 us).  Hidden in asm the
// copies are counted by hand (the s_waitcnt vmcnt(N) before the barriers); M0 is saved and restored inside the statement.
__device__ __forceinline__ void glds16_hidden(const char* gsrc, char* lds_dst) {
  unsigned keep;
  const unsigned ldst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(lds_dst));  // (wave-uniform by construction)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(ldst) : "memory");
}

// The gx * gy workgroups of one K slab stream the SAME dY / X lines.  The hardware hands workgroup L of a 1-D grid to XCD L % 8 (each XCD
// has its own L2), so the logical index is permuted: the blocks of a slab are consecutive on ONE XCD, run at the same time and
// share the lines through that L2 instead of fetching them gx * gy times from HBM.
struct SlabBlock { int x, y, z; };
__device__ __forceinline__ SlabBlock xcd_slab_block(int gx, int gy, int S) {
  const int T = gx * gy * S, L = blockIdx.x;
  const int xcd = L & 7, j = L >> 3, rem = T & 7;
  const int w = xcd * (T >> 3) + (xcd < rem ? xcd : rem) + j;
  SlabBlock b;
  b.x = w % gx;
  const int t = w / gx;
  b.y = t % gy;
  b.z = t / gy;
  return b;
}

struct WgradK {
  const char* x[WG_MAX_PAIRS];
  const char* dy[WG_MAX_PAIRS];
  int npairs;
  long long Upair;  // K units per pair
  long long x_ps;
  int Cin;
  long long dy_ps;
  int Cout;
  float* dW;
  int I_total, o0, i0;
  float* db;
  float scale;
  int N, H, W, SEG;
  int HB;  // row blocks per image (KS > 1: a K unit is UR rows x 32 pixels)
  long long M, U;
  int S;
  int vec_x, vec_dy;  // 16-byte vector loads are legal for the operand (pixel stride, channel count and base address are vector multiples)
  float* slab;  // non-null: every workgroup STORES its partial tile to slab[z][y][x][tile] (a reduce kernel sums the K splits in a fixed
                // order) instead of adding it to dW with float atomics -- with few (co, ci) tiles and many K splits the atomics all hit the
                // same few addresses and serialise (7x7: 131 us of a 140 us launch)
};

// UR: image rows per K unit (a unit = UR rows x 32 pixels).  The X tile with its halo is (UR + KS - 1) x (32 + KS - 1) pixels: for 7x7
// one row per unit re-loads 7 rows of halo per 32 pixels; four rows per unit cut the X traffic per pixel 2.8-fold.
template <typename T, int KS, int CT, int IT, int UR = 1>
struct WgradCfg {
  static constexpr int ES = ElemTraits<T>::ES;
  static constexpr int KK = KS * KS;
  static constexpr int XR = KS + UR - 1, XW = 32 + KS - 1;
  static constexpr int UP = UR * 32;  // pixels per unit
  static constexpr int DYC = CT * 16, XC = IT * 16;
  static constexpr int DY_RS = DYC * ES + 16, X_RS = XC * ES + 16;  // LDS row (= pixel) strides in bytes
  static constexpr int DY_BYTES = UP * DY_RS, X_BYTES = XR * XW * X_RS;
  static constexpr int WAVE_BYTES = (DY_BYTES + X_BYTES + 15) & ~15;
  static constexpr int RED_RS = XC * KK + 2;  // floats per output-channel row of the reduction image
  static constexpr int RED_BYTES = (DYC * RED_RS + DYC) * 4;
  // the reduction image takes over the waves' tiles once the main loop is done (behind a barrier)
  static constexpr int LDS_BYTES = 4 * WAVE_BYTES > RED_BYTES ? 4 * WAVE_BYTES : RED_BYTES;
  static constexpr int VPL = 16 / ES;                        // elements per 16-byte vector
  static constexpr int DY_VPP = DYC / VPL, X_VPP = XC / VPL;  // vectors per pixel
  static constexpr int DY_NV = (UP * DY_VPP + 63) / 64, X_NV = (XR * XW * X_VPP + 63) / 64;  // vectors per lane
};

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int rs, int k0, int c0, int lane) {
  // operand fragment of v_mfma_f32_16x16x32_bf16 from a [pixel][channel] LDS tile: lane l gets channel c0+(l&15),
  // pixels k0 + 8*(l>>4) + 0..7.  Two transposed 4x16 block reads (cdna guide T10): lane i = 4q+p of a 16-lane
  // group supplies the address of block row q, columns 4p..4p+3.
  const int i = lane & 15, g = lane >> 4;
  const char* p0 = tile + (k0 + 8 * g + (i >> 2)) * rs + (c0 + 4 * (i & 3)) * 2;
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * rs));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <typename T, int KS, int CT, int IT, int UR = 1>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradK a) {
  using C = WgradCfg<T, KS, CT, IT, UR>;
  constexpr int ES = C::ES, KK = C::KK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* dyt = smem + wave * C::WAVE_BYTES;
  char* xt = dyt + C::DY_BYTES;
  float* red = reinterpret_cast<float*>(smem);  // (valid behind the barrier that follows the main loop)
  float* redb = red + C::DYC * C::RED_RS;

  const int ob = blockIdx.x * C::DYC, ib = blockIdx.y * C::XC;  // channel offsets of this block
  const long long u_lo = a.U * blockIdx.z / a.S, u_hi = a.U * (blockIdx.z + 1) / a.S;
  const bool do_bias = (a.db != nullptr) && (blockIdx.y == 0);

  f32x4 acc[CT][KK][IT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int t = 0; t < KK; ++t)
#pragma unroll
      for (int it = 0; it < IT; ++it) acc[ct][t][it] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  uint4 rdy[C::DY_NV], rx[C::X_NV];
  const char* zsrc = reinterpret_cast<const char*>(g_zero_buf);

  auto load_unit = [&](long long ug) {
    // which (x, dy) pair, then pixel coordinates of the unit inside it
    const int pair = (int)(ug / a.Upair);
    const long long u = ug - (long long)pair * a.Upair;
    const char* xbase = a.x[pair];
    const char* dybase = a.dy[pair];
    int n = 0, y = 0, x0 = 0;
    long long m0 = 0;
    if (KS > 1) {
      const int seg = (int)(u % a.SEG);
      const long long r = u / a.SEG;
      y = (int)(r % a.HB) * UR;
      n = (int)(r / a.HB);
      x0 = seg * 32;
    } else {
      m0 = u * 32;
    }
    // Every load is UNCONDITIONAL: a lane outside the image (or past the channels) reads the zero buffer instead.  Loads behind a
    // per-lane branch are each followed by an s_waitcnt vmcnt(0) -- ten serialised memory latencies per unit, which made even the
    // 2 x 2-pixel SPyNet level cost 100 us.
    auto dy_src = [&](int k, bool& ok, int& c) -> const char* {
      const int idx = lane + 64 * k;
      const int p = idx / C::DY_VPP, v = idx - p * C::DY_VPP;
      c = ob + v * C::VPL;
      ok = (idx < C::UP * C::DY_VPP);
      long long pix;
      if (KS > 1) {
        const int rr = p >> 5, col = p & 31;
        ok = ok && (x0 + col < a.W) && (y + rr < a.H);
        pix = ((long long)n * a.H + y + rr) * a.W + x0 + col;
      } else { ok = ok && (m0 + p < a.M); pix = m0 + p; }
      return dybase + (pix * a.dy_ps + c) * ES;
    };
    auto x_src = [&](int k, bool& ok, int& c) -> const char* {
      const int idx = lane + 64 * k;
      const int p = idx / C::X_VPP, v = idx - p * C::X_VPP;
      c = ib + v * C::VPL;
      ok = (idx < C::XR * C::XW * C::X_VPP);
      long long pix;
      if (KS > 1) {
        const int r = p / C::XW, col = p - r * C::XW;
        const int yy = y + r - KS / 2, xx = x0 + col - KS / 2;
        ok = ok && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        pix = ((long long)n * a.H + yy) * a.W + xx;
      } else {
        ok = ok && (m0 + p < a.M);
        pix = m0 + p;
      }
      return xbase + (pix * a.x_ps + c) * ES;
    };
    if (a.vec_dy) {
#pragma unroll
      for (int k = 0; k < C::DY_NV; ++k) {
        bool ok; int c;
        const char* src = dy_src(k, ok, c);
        rdy[k] = *reinterpret_cast<const uint4*>((ok && c < a.Cout) ? src : zsrc);  // (a vector that starts inside the channels ends inside the pixel stride: channels past Cout are computed and dropped)
      }
    } else {
#pragma unroll
      for (int k = 0; k < C::DY_NV; ++k) {
        bool ok; int c;
        const char* src = dy_src(k, ok, c);
        T tmp[C::VPL];
#pragma unroll
        for (int e = 0; e < C::VPL; ++e) tmp[e] = *reinterpret_cast<const T*>((ok && c + e < a.Cout) ? src + e * ES : zsrc);
        rdy[k] = *reinterpret_cast<const uint4*>(tmp);
      }
    }
    if (a.vec_x) {
#pragma unroll
      for (int k = 0; k < C::X_NV; ++k) {
        bool ok; int c;
        const char* src = x_src(k, ok, c);
        rx[k] = *reinterpret_cast<const uint4*>((ok && c < a.Cin) ? src : zsrc);
      }
    } else {
#pragma unroll
      for (int k = 0; k < C::X_NV; ++k) {
        bool ok; int c;
        const char* src = x_src(k, ok, c);
        T tmp[C::VPL];
#pragma unroll
        for (int e = 0; e < C::VPL; ++e) tmp[e] = *reinterpret_cast<const T*>((ok && c + e < a.Cin) ? src + e * ES : zsrc);
        rx[k] = *reinterpret_cast<const uint4*>(tmp);
      }
    }
  };

  auto store_unit = [&]() {
#pragma unroll
    for (int k = 0; k < C::DY_NV; ++k) {
      const int idx = lane + 64 * k;
      const int p = idx / C::DY_VPP, v = idx - p * C::DY_VPP;
      if (idx < C::UP * C::DY_VPP) *reinterpret_cast<uint4*>(dyt + p * C::DY_RS + v * 16) = rdy[k];
    }
#pragma unroll
    for (int k = 0; k < C::X_NV; ++k) {
      const int idx = lane + 64 * k;
      const int p = idx / C::X_VPP, v = idx - p * C::X_VPP;
      if (idx < C::XR * C::XW * C::X_VPP) *reinterpret_cast<uint4*>(xt + p * C::X_RS + v * 16) = rx[k];
    }
  };

  long long u = u_lo + wave;
  if (u < u_hi) load_unit(u);
  while (u < u_hi) {
    store_unit();
    const long long un = u + 4;
    if (un < u_hi) load_unit(un);
    // wave-private tile: program order + the compiler's lgkmcnt waits are the only synchronisation needed
#pragma unroll
    for (int rr = 0; rr < UR; ++rr) {
      if constexpr (ES == 2) {
        bf16x8 af[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) af[ct] = tr_frag(dyt, C::DY_RS, rr * 32, ct * 16, lane);
#pragma unroll
        for (int t = 0; t < KK; ++t) {
          const int ky = t / KS, kx = t % KS;
#pragma unroll
          for (int it = 0; it < IT; ++it) {
            const bf16x8 bfg = tr_frag(xt + (rr + ky) * C::XW * C::X_RS, C::X_RS, kx, it * 16, lane);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
              acc[ct][t][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct], bfg, acc[ct][t][it], 0, 0, 0);
          }
        }
      } else {
        const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float af[CT];
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) af[ct] = *reinterpret_cast<const float*>(dyt + (rr * 32 + 4 * j + g) * C::DY_RS + (ct * 16 + l15) * 4);
#pragma unroll
          for (int t = 0; t < KK; ++t) {
            const int ky = t / KS, kx = t % KS;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
              const float bv = *reinterpret_cast<const float*>(xt + ((rr + ky) * C::XW + 4 * j + g + kx) * C::X_RS + (it * 16 + l15) * 4);
#pragma unroll
              for (int ct = 0; ct < CT; ++ct)
                acc[ct][t][it] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ct], bv, acc[ct][t][it], 0, 0, 0);
            }
          }
        }
      }
    }
    if (do_bias && lane < C::DYC) {
      float s = 0.f;
#pragma unroll 8
      for (int p = 0; p < C::UP; ++p) s += to_f32(*reinterpret_cast<const T*>(dyt + p * C::DY_RS + lane * ES));
      bsum += s;
    }
    u = un;
  }

  // ---- cross-wave reduction in LDS, laid out [co][ci][tap] = OIHW order of this block
  __syncthreads();  // every wave is done with its tiles: the reduction image reuses that memory
  for (int i = tid; i < C::DYC * C::RED_RS + C::DYC; i += 256) red[i] = 0.f;
  __syncthreads();
  {
    const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int t = 0; t < KK; ++t)
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            atomicAdd(&red[(ct * 16 + 4 * g + r) * C::RED_RS + (it * 16 + l15) * KK + t], acc[ct][t][it][r]);
    if (do_bias && lane < C::DYC) atomicAdd(&redb[lane], bsum);
  }
  __syncthreads();
  constexpr int ROW = C::XC * KK;
  if (a.slab) {
    constexpr int TILE = C::DYC * ROW + C::DYC;  // floats per workgroup: the tile, then the bias sums
    float* dst = a.slab + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * TILE;
    for (int i = tid; i < C::DYC * ROW; i += 256) {
      const int co = i / ROW, rem = i - co * ROW;
      dst[i] = red[co * C::RED_RS + rem];
    }
    if (tid < C::DYC) dst[C::DYC * ROW + tid] = do_bias ? redb[tid] : 0.f;
    return;
  }
  for (int i = tid; i < C::DYC * ROW; i += 256) {
    const int co = i / ROW, rem = i - co * ROW;
    const int ci = rem / KK;
    if (ob + co < a.Cout && ib + ci < a.Cin)
      atomicAdd(&a.dW[((long long)(a.o0 + ob + co) * a.I_total + (a.i0 + ib)) * KK + rem], red[co * C::RED_RS + rem] * a.scale);
  }
  if (do_bias && tid < C::DYC && ob + tid < a.Cout) atomicAdd(&a.db[a.o0 + ob + tid], redb[tid] * a.scale);
}

// sums the K splits of the slab form in a fixed order and adds the result to dW / db (one thread per element: no atomics)
__global__ void wgrad_slab_reduce_kernel(const float* __restrict__ slab, int S, int gx, int gy, int DYC, int XC, int KK, int Cout, int Cin,
                                         float* __restrict__ dW, int I_total, int o0, int i0, float* __restrict__ db, float scale) {
  const int ROW = XC * KK, TILE = DYC * ROW + DYC;
  const long long total = (long long)gx * gy * TILE;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(i % TILE);
    const int bxy = (int)(i / TILE);
    const int bx = bxy % gx, by = bxy / gx;
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += slab[((size_t)z * gy * gx + bxy) * TILE + e];
    if (e < DYC * ROW) {
      const int co = e / ROW, rem = e - co * ROW, ci = rem / KK;
      if (bx * DYC + co < Cout && by * XC + ci < Cin)
        dW[((long long)(o0 + bx * DYC + co) * I_total + (i0 + by * XC)) * KK + rem] += s * scale;
    } else if (db && by == 0) {
      const int co = e - DYC * ROW;
      if (bx * DYC + co < Cout) db[o0 + bx * DYC + co] += s * scale;
    }
  }
}

template <typename T, int KS, int CT, int IT, int UR = 1>
int launch_wgrad(WgradK k, hipStream_t st) {
  using C = WgradCfg<T, KS, CT, IT, UR>;
  static_assert(C::LDS_BYTES <= 160 * 1024, "wgrad LDS");
  const int gx = cdiv(k.Cout, C::DYC), gy = cdiv(k.Cin, C::XC);
  long long s = 1024 / ((long long)gx * gy);  // ~4 workgroups per CU in flight
  if (s > k.U / 32) s = k.U / 32;           // >= 8 K units per wave, or the float-atomic epilogue dominates
  constexpr long long TILE = C::DYC * C::XC * C::KK + C::DYC;
  if (k.slab) {  // the K splits must fit the workspace
    const long long cap = k.M /* = workspace floats here, see wgrad_impl */ / ((long long)gx * gy * TILE);
    if (s > cap) s = cap;
  }
  if (s < 1) s = 1;
  if (s > 65535) s = 65535;
  k.S = (int)s;
  auto fn = conv_wgrad_kernel<T, KS, CT, IT, UR>;
  static bool attr_set[VMG_MAX_DEVICES] = {};  // the attribute is per device
  const int dev = vmg_current_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[dev] = true;
  }
  const long long ws_floats = k.slab ? k.M : 0;
  k.M = (long long)k.N * k.H * k.W;
  if (k.slab && (long long)gx * gy * k.S * TILE > ws_floats) k.slab = nullptr;  // (workspace too small even for one split: atomics)
  hipLaunchKernelGGL(fn, dim3(gx, gy, k.S), dim3(256), C::LDS_BYTES, st, k);
  VMG_LAUNCH_CHECK();
  if (k.slab) {
    const long long total = (long long)gx * gy * TILE;
    const int rb = (int)(cdiv64(total, 256) > 4096 ? 4096 : cdiv64(total, 256));
    hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3(rb), dim3(256), 0, st, (const float*)k.slab, k.S, gx, gy, C::DYC, C::XC, C::KK, k.Cout, k.Cin, k.dW,
                       k.I_total, k.o0, k.i0, k.db, k.scale);
    VMG_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace

static int wgrad_impl(int dtype, int ks, int npairs, const void* const* x, const void* const* dy, int N, int H, int W, int64_t x_ps,
                      int Cin, int64_t dy_ps, int Cout, float* dW, int I_total, int o0, int i0, float* db, float scale, void* stream,
                      float* ws = nullptr, int64_t ws_bytes = 0) {
  VMG_CHECK(ks == 1 || ks == 3 || ks == 7, "conv_wgrad: ks must be 1, 3 or 7");
  VMG_CHECK(dtype == VMG_F32 || dtype == VMG_BF16, "conv_wgrad: bad dtype");
  VMG_CHECK(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv_wgrad: bad shape");
  VMG_CHECK(npairs >= 1 && npairs <= WG_MAX_PAIRS && x && dy && dW, "conv_wgrad: 1..%d (x, dy) pairs", WG_MAX_PAIRS);
  VMG_CHECK(x_ps >= Cin && dy_ps >= Cout && i0 >= 0 && i0 + Cin <= I_total && o0 >= 0, "conv_wgrad: bad strides / slices");
  const int es = dtype == VMG_BF16 ? 2 : 4, vpl = 16 / es;
  WgradK k;
  memset(&k, 0, sizeof(k));
  // 16-byte vector loads need a pixel stride that is a multiple of the vector (then a vector that starts inside the channel range ends
  // inside the pixel's stride: a zero-padded or wider tensor behind a channel slice, e.g. the 3-channel output gradient of conv_last
  // padded to 8) and aligned bases; channel counts need not be multiples of 8
  k.vec_x = (x_ps % vpl == 0);
  k.vec_dy = (dy_ps % vpl == 0);
  for (int p = 0; p < npairs; ++p) {
    VMG_CHECK(x[p] && dy[p], "conv_wgrad: null pointer in pair %d", p);
    k.x[p] = (const char*)x[p];
    k.dy[p] = (const char*)dy[p];
    k.vec_x = k.vec_x && ((uintptr_t)x[p] % 16 == 0);
    k.vec_dy = k.vec_dy && ((uintptr_t)dy[p] % 16 == 0);
  }
  k.npairs = npairs;
  k.x_ps = x_ps; k.Cin = Cin; k.dy_ps = dy_ps; k.Cout = Cout;
  k.dW = dW; k.I_total = I_total; k.o0 = o0; k.i0 = i0; k.db = db; k.scale = scale;
  k.N = N; k.H = H; k.W = W; k.M = (long long)N * H * W;
  k.SEG = cdiv(W, 32);
  const int ur = 1;  // rows per K unit (must match the launch below; 2 and 4 rows spilled registers in the 49-tap instantiations)
  k.HB = cdiv(H, ur);
  k.Upair = ks > 1 ? (long long)N * k.HB * k.SEG : cdiv64(k.M, 32);
  k.U = k.Upair * npairs;
  hipStream_t st = (hipStream_t)stream;
  if (ks == 7) {  // 49 taps x one 16 x 16 tile per workgroup; K splits reduced through the workspace when there is one
    if (ws && ws_bytes >= (1 << 20)) {
      k.slab = ws;
      k.M = ws_bytes / 4;  // (launch_wgrad reads the workspace size from here and restores M)
    }
    return dtype == VMG_BF16 ? launch_wgrad<bf16, 7, 1, 1, 1>(k, st) : launch_wgrad<float, 7, 1, 1, 1>(k, st);
  }
  if (dtype == VMG_BF16 && ks == 3 && Cout <= 16) {  // (conv_last: 64 -> 3 on 1.8 M pixels; 64 input channels per workgroup: whole 128-byte pixel rows of X)
    // (only without a workspace: vmg_conv_wgrad_batched_ws sends this shape to the three-wave instance of conv_wgrad7_kernel, 122 us
    //  instead of 470-560 here, where ~650 address instructions per K unit and one wave per SIMD leave the loads exposed)
    return Cin >= 64 ? launch_wgrad<bf16, 3, 1, 4>(k, st) : launch_wgrad<bf16, 3, 1, 1>(k, st);
  }
  if (dtype == VMG_BF16) return ks == 3 ? launch_wgrad<bf16, 3, 3, 1>(k, st) : launch_wgrad<bf16, 1, 3, 3>(k, st);
  return ks == 3 ? launch_wgrad<float, 3, 3, 1>(k, st) : launch_wgrad<float, 1, 3, 3>(k, st);
}

extern "C" int vmg_conv_wgrad(int dtype, int ks, int N, int H, int W, const void* x, int64_t x_ps, int Cin, const void* dy,
                              int64_t dy_ps, int Cout, float* dW, int I_total, int o0, int i0, float* db, float scale,
                              void* stream) {
  return wgrad_impl(dtype, ks, 1, &x, &dy, N, H, W, x_ps, Cin, dy_ps, Cout, dW, I_total, o0, i0, db, scale, stream);
}

extern "C" int vmg_conv_wgrad_batched(int dtype, int ks, int npairs, const void* const* x, const void* const* dy, int N, int H, int W,
                                      int64_t x_ps, int Cin, int64_t dy_ps, int Cout, float* dW, int I_total, int o0, int i0,
                                      float* db, float scale, void* stream) {
  return wgrad_impl(dtype, ks, npairs, x, dy, N, H, W, x_ps, Cin, dy_ps, Cout, dW, I_total, o0, i0, db, scale, stream);
}

// =====================================================================================================
// Large-tile kernels (bf16): shared LDS tiles, slab reduction.
//
// The v1 kernel above gives every workgroup a 48 x 16 x 9 output tile, so dY is re-read 9x and X 3x from L2/HBM (1.2 GB
// for the batched gradient of one recurrent conv) and the float-atomic epilogue scales with the number of K splits.
// Here a workgroup owns ALL 144 output channels x 48 input channels x 9 taps (3x3) or 144 x 144 (1x1).  Per 32-pixel K
// unit the dY tile [32][144] and the X tile ([3][34][48] with its halo, or [32][144]) are filled by LDS-DMA
// (global_load_lds, per-lane source, lane-linear destination; out-of-image lanes read a zero buffer, so every wave issues
// the same number of DMA instructions per unit and the wait can be COUNTED) and read transposed.  Partial results go to
// per-slab workspaces in the accumulator's native layout with coalesced float4 stores; a second kernel sums the slabs in
// a fixed order (bitwise reproducible, unlike atomics) and adds them into the OIHW gradient.  The bias gradient is one
// more MFMA per co tile against a ones operand.
// =====================================================================================================
namespace {


constexpr int W3_MAX_PROBS = 8;  // weight-gradient problems of one shape served by one launch (vmg_conv_wgrad3_multi)
struct Wgrad2K {
  const char* x[W3_MAX_PROBS * WG_MAX_PAIRS];   // [problem][pair]
  const char* dy[W3_MAX_PROBS * WG_MAX_PAIRS];
  int npairs, nprob;
  long long Upair, U;  // K units per pair / per problem
  long long x_ps, dy_ps;
  int Cin, Cout;
  float* slab;       // [problem][S][ciblk][coblk][8 waves][36 tiles][64][4]
  int N, H, W, SEG, S;
  int has_bias;
  int gx, gy;        // co / ci blocks (the grid is 1-D: gx * gy * S * nprob workgroups, see xcd_slab_block)
  int dbg;           // diagnostics build only (env VMG_WGRAD_DBG): 1 no copies after the prologue, 2 no MFMAs, 4 no fragment reads
};
struct Wgrad3Out {   // per problem: where the reduce kernel adds the sums
  float* dW[W3_MAX_PROBS];
  float* db[W3_MAX_PROBS];
  float scale[W3_MAX_PROBS];
};

constexpr int W2_WAVES = 9, W2_THREADS = 576;  // the 1x1 kernel's workgroup
constexpr int W2_DYC = 144, W2_XC = 48, W2_XR = 3, W2_XW = 34;
constexpr int W2_XVEC = W2_XR * W2_XW * (W2_XC / 8);        // 612 vectors in the X tile

// Sum of element i over the S slabs, in a FIXED order (deterministic): the block's 4 waves each take the slabs
// s = w, w+4, ... (coalesced 256-byte rows, 4 loads in flight), the 4 partial sums are combined through LDS as
// ((p0 + p1) + (p2 + p3)).  Call with blockDim = 256; lane l of every wave works on element i0 + l; returns the sum on
// wave 0 (other waves get an unspecified value).
__device__ __forceinline__ float slab_sum_4waves(const float* __restrict__ slab, long long per_s, int S, long long i, bool valid, float* red) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (valid) {
    int s = w;
    for (; s + 12 < S; s += 16) {
      a0 += slab[(long long)s * per_s + i];
      a1 += slab[(long long)(s + 4) * per_s + i];
      a2 += slab[(long long)(s + 8) * per_s + i];
      a3 += slab[(long long)(s + 12) * per_s + i];
    }
    for (; s < S; s += 4) a0 += slab[(long long)s * per_s + i];
  }
  red[w * 64 + l] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  const float r = (red[l] + red[64 + l]) + (red[128 + l] + red[192 + l]);
  __syncthreads();
  return r;
}

// =====================================================================================================
// 3x3: the tile (144 co x 48 ci x 9 taps per workgroup) on EIGHT waves, two per SIMD.
// (A nine-wave 3 x 3 layout of the same tile put three waves on one SIMD -- its MFMA share bounded every unit -- and left
// 170 registers per wave, too few for a second fragment set: LDS reads and MFMAs ran one after the other; 161 us vs 124 us
// for the 7-use gradient of a recurrent conv.)  Wave (h, q) owns co tiles
// 5h..5h+4 (h = 1: four real tiles) x fragment columns 7q..7q+6 of the 27 (ci tile, tap) columns = 35 accumulator
// tiles; with 256 registers the X fragment of the NEXT unit is read into the registers of the fragment just consumed,
// so LDS reads run under the MFMAs.  Per unit the [dY tile | X tile] pair is one linear list of 1188 16-byte vectors
// copied by LDS-DMA, three 1-KiB instructions per wave (lanes past the list or outside the image read a zero buffer):
// the wait is COUNTED.  Six LDS unit slots, one barrier per two units; slab partials + ordered reduction.
// =====================================================================================================
constexpr int W3_WAVES = 8, W3_THREADS = 512, W3_SLOTS = 3;
constexpr int W3_DYV = 576, W3_VECS = 576 + W2_XVEC;            // 1188 vectors per unit
constexpr int W3_BUF = W3_SLOTS * W3_WAVES * 1024;               // 24576 B per buffer (every wave issues 3 full instructions)
constexpr int W3_TILES = 36;                                     // 35 accumulator tiles + 1 bias tile per wave
constexpr long long W3_WG_FLOATS = (long long)W3_WAVES * W3_TILES * 256;

__device__ __forceinline__ bf16x8 tr_read(const char* p0, int rs) {
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + 4 * rs));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

typedef __attribute__((ext_vector_type(4))) unsigned int w3_u32x4;
// LDS-DMA through a buffer resource: 16 bytes per lane from (descriptor base + SGPR offset + per-lane 32-bit offset) into the lane-linear 1 KiB
// block at lds_dst; a lane whose offset is >= num_records (2^31 here) gets zeros.  Hidden from hipcc's wait bookkeeping like glds16_hidden; the
// s_nop covers the M0 write and SGPRs the compiler's SALU may have written just before the statement (descriptor, offset).
__device__ __forceinline__ void bufdma16_hidden(unsigned voff, w3_u32x4 rsrc, unsigned soff, char* lds_dst) {
  unsigned keep;
  const unsigned ldst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(lds_dst));
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(ldst) : "memory");
}
__device__ __forceinline__ w3_u32x4 w3_rsrc(const char* base) {  // raw buffer (stride 0) of 2^31 bytes at `base` (wave-uniform)
  const unsigned long long p = (unsigned long long)(uintptr_t)base;
  w3_u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32)) & 0xFFFFu;
  r[2] = 0x80000000u;
  r[3] = 0x00020000u;
  return r;
}

__global__ __launch_bounds__(W3_THREADS) void conv_wgrad3_kernel(const Wgrad2K a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wave >> 2, q = wave & 3;
  SlabBlock blk = xcd_slab_block(a.gx, a.gy, a.S * a.nprob);
  const int prob = blk.z / a.S;  // several problems of one shape share the launch: fewer, longer K slabs per problem at a full chip
  const int zz = blk.z;          // slab index over all problems (the slab workspace is [problem][S]...)
  blk.z -= prob * a.S;
  const char* const* xs = a.x + prob * WG_MAX_PAIRS;
  const char* const* dys = a.dy + prob * WG_MAX_PAIRS;
  const int ob = blk.x * W2_DYC, ib = blk.y * W2_XC;
  const long long u_lo = a.U * blk.z / a.S, u_hi = a.U * (blk.z + 1) / a.S;
  const char* zsrc = reinterpret_cast<const char*>(g_zero_buf);

  f32x4 acc[5][7];
#pragma unroll
  for (int c = 0; c < 5; ++c)
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[c][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 accb[5];
#pragma unroll
  for (int c = 0; c < 5; ++c) accb[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16 one = (bf16)1.0f;
  const bf16x8 ones = {one, one, one, one, one, one, one, one};

  // ---- copy instructions.  The kernel is bound by INSTRUCTION ISSUE, not by memory (with every copy an L2 hit it runs at the same
  // speed) nor by the matrix pipe: per 32-pixel unit a wave has 35 MFMAs (560 cycles) and, in the first version, ~450 other instructions
  // (per-lane 64-bit address arithmetic of the copies, 64-bit loop counters, slot rotation, fragment double buffers): 0.44 us of loop
  // skeleton + 0.57 us of copy issue + 0.5 us of MFMAs per unit.  Hence: a lane keeps a 64-bit POINTER per copy instruction and adds a
  // wave-uniform delta per unit; the loop is unrolled over the six LDS slots (static slot addresses, static barrier parity, 32-bit
  // counters); one fragment set, re-read behind the last MFMA that uses it.
  // Lane constants of the three copy instructions: vector L of the [dY | X] list.  Chunk sl * 8 + wave < 9 is a dY chunk, the rest X
  // (wave-uniform); abk = a | b << 8 | kind << 16, kind 0 dY (a = pixel), 1 X (a = row, b = column), 2 padding (reads the zero buffer).
  int s_abk[W3_SLOTS];
  const char* s_ptr[W3_SLOTS];  // source of the NEXT unit to be issued
  const int nun = (int)(u_hi - u_lo);
  int ipair = (int)(u_lo / a.Upair), iseg, iy, in_;
  {
    const long long uu = u_lo - (long long)ipair * a.Upair;
    iseg = (int)(uu % a.SEG);
    const long long r = uu / a.SEG;
    iy = (int)(r % a.H);
    in_ = (int)(r / a.H);
  }
  {
    const long long row0 = ((long long)in_ * a.H + iy) * a.W + iseg * 32;
#pragma unroll
    for (int sl = 0; sl < W3_SLOTS; ++sl) {
      const int L = (sl * W3_WAVES + wave) * 64 + lane;
      if (L < W3_DYV) {
        const int pp = L / 18, v = L - pp * 18;
        s_abk[sl] = pp | (((ob + v * 8 + 8 <= a.Cout) ? 0 : 2) << 16);
        s_ptr[sl] = dys[ipair] + ((row0 + pp) * a.dy_ps + ob + v * 8) * 2;
      } else if (L < W3_VECS) {
        const int vec = L - W3_DYV, pp = vec / 6, v = vec - pp * 6;
        const int rr = pp / W2_XW, col = pp - rr * W2_XW;
        s_abk[sl] = rr | (col << 8) | (((ib + v * 8 + 8 <= a.Cin) ? 1 : 2) << 16);
        s_ptr[sl] = xs[ipair] + ((row0 + (long long)(rr - 1) * a.W + (col - 1)) * a.x_ps + ib + v * 8) * 2;
      } else {
        s_abk[sl] = 2 << 16;
        s_ptr[sl] = zsrc;
      }
    }
  }
  const long long pair_pix = (long long)a.N * a.H * a.W;
  const int row_adv = a.W - (a.SEG - 1) * 32;  // pixels from the last segment of a row to the start of the next row
  auto issue = [&](int buf) __attribute__((always_inline)) {
    const int x0 = iseg * 32;
    char* dst = smem + buf * W3_BUF;
#pragma unroll
    for (int sl = 0; sl < W3_SLOTS; ++sl) {
      const int abk = s_abk[sl], kind = abk >> 16, sa = abk & 255, sb = (abk >> 8) & 255;
      bool ok;
      if (sl * W3_WAVES + wave < W3_DYV / 64) ok = (kind == 0) & ((unsigned)(x0 + sa) < (unsigned)a.W);
      else ok = (kind == 1) & ((unsigned)(iy + sa - 1) < (unsigned)a.H) & ((unsigned)(x0 + sb - 1) < (unsigned)a.W);
      glds16_hidden(ok ? s_ptr[sl] : zsrc, dst + (sl * W3_WAVES + wave) * 1024);
    }
    // odometer + the wave-uniform pointer deltas to the next unit
    long long adv = 32;  // pixels
    long long rebase_dy = 0, rebase_x = 0;
    if (++iseg == a.SEG) {
      iseg = 0;
      adv = row_adv;
      if (++iy == a.H) {
        iy = 0;
        if (++in_ == a.N) {
          in_ = 0;
          if (ipair + 1 < a.npairs) {  // next (x, dy) pair: back to pixel 0 of other tensors
            rebase_dy = dys[ipair + 1] - dys[ipair] - pair_pix * a.dy_ps * 2;
            rebase_x = xs[ipair + 1] - xs[ipair] - pair_pix * a.x_ps * 2;
          }
          ++ipair;
        }
      }
    }
    const long long d_dy = adv * a.dy_ps * 2 + rebase_dy, d_x = adv * a.x_ps * 2 + rebase_x;
#pragma unroll
    for (int sl = 0; sl < W3_SLOTS; ++sl) s_ptr[sl] += (sl * W3_WAVES + wave < W3_DYV / 64) ? d_dy : d_x;
  };

  // ---- fragment addresses (within a buffer): dY rows are 288 B, X rows 96 B; lane i = 4qq+pp of a 16-lane group supplies
  // block row qq, columns 4pp..4pp+3 of a transposed 4x16 block read
  const int li = lane & 15, lg = lane >> 4;
  const int a_off = (8 * lg + (li >> 2)) * (W2_DYC * 2) + (h * 5 * 16 + 4 * (li & 3)) * 2;  // + c * 32 for co tile c of this wave
  int b_off[7];
#pragma unroll
  for (int jj = 0; jj < 7; ++jj) {
    const int col = min(q * 7 + jj, 26), itile = col / 9, tap = col - 9 * itile, ky = tap / 3, kx = tap - 3 * ky;
    b_off[jj] = W3_DYV * 16 + ky * W2_XW * (W2_XC * 2) + (kx + 8 * lg + (li >> 2)) * (W2_XC * 2) + (itile * 16 + 4 * (li & 3)) * 2;
  }

  // SIX unit slots, one barrier per TWO units.  Unit j lives in slot j % 6; iteration j multiplies unit j (fragments already in
  // registers) and reads unit j+1's fragments.  At the barrier of an even j every wave has (a) waited for its share of units
  // j+1 and j+2 (units j+3, j+4 stay in flight: the wait is counted) and (b) drained its reads of unit j; afterwards units
  // j+5 and j+6 are issued into the slots of units j-1 and j.
  for (int j = 0; j < 5; ++j)
    if (j < nun) issue(j);
  {
    const int later = nun > 3 ? (nun > 4 ? 2 : 1) : 0;  // units 3, 4 may stay in flight
    if (later == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  bf16x8 af[5], bfg[7];
  if (nun > 0) {
#pragma unroll
    for (int c = 0; c < 5; ++c) af[c] = tr_read(smem + a_off + c * 32, W2_DYC * 2);
#pragma unroll
    for (int jj = 0; jj < 7; ++jj) bfg[jj] = tr_read(smem + b_off[jj], W2_XC * 2);
  }
  const bool do_bias = a.has_bias && q == 0 && blk.y == 0;
  // one unit; SU = its slot (compile-time), j = its index.  Returns false when the slab is finished.
  auto unit = [&](int j, auto su_c) __attribute__((always_inline)) -> bool {
    constexpr int SU = decltype(su_c)::value, SN = (SU + 1) % 6, S5 = (SU + 5) % 6;
    constexpr bool SYNC = (SU & 1) == 0;  // slots alternate with the unit index: even units synchronise
    if (j >= nun) return false;
    if (SYNC) {
      const int later = (j + 3 < nun ? 1 : 0) + (j + 4 < nun ? 1 : 0);
      if (later == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of unit j are complete
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    const char* src = smem + ((j + 1 < nun) ? SN : SU) * W3_BUF;  // (last unit: re-read its own slot; the values are not used)
    // units j+5, j+6 -> the slots of units j-1 and j (free since this unit's barrier).  The two waves of a SIMD (h = 0, 1) are kept in
    // step by the barriers: h = 0 issues its copies BEFORE its MFMAs, h = 1 after them, so that one wave's address arithmetic runs
    // under the other wave's MFMAs instead of both leaving the matrix pipe idle at the same time.
    if (SYNC && h == 0) {
#ifdef VMG_DIAG
      if (!(a.dbg & 1))
#endif
      {
        if (j + 5 < nun) issue(S5);
        if (j + 6 < nun) issue(SU);
      }
    }
    if (do_bias) {
#pragma unroll
      for (int c = 0; c < 5; ++c) accb[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c], ones, accb[c], 0, 0, 0);
    }
#pragma unroll
    for (int jj = 0; jj < 7; ++jj) {
#ifdef VMG_DIAG
      if (!(a.dbg & 2))
#endif
      {
#pragma unroll
        for (int c = 0; c < 5; ++c) acc[c][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c], bfg[jj], acc[c][jj], 0, 0, 0);
      }
#ifdef VMG_DIAG
      if (!(a.dbg & 4))
#endif
      bfg[jj] = tr_read(src + b_off[jj], W2_XC * 2);
    }
#ifdef VMG_DIAG
    if (!(a.dbg & 4))
#endif
    {
#pragma unroll
      for (int c = 0; c < 5; ++c) af[c] = tr_read(src + a_off + c * 32, W2_DYC * 2);  // (behind the last MFMA that reads af[c])
    }
    if (SYNC && h == 1) {
#ifdef VMG_DIAG
      if (!(a.dbg & 1))
#endif
      {
        if (j + 5 < nun) issue(S5);
        if (j + 6 < nun) issue(SU);
      }
    }
    return true;
  };
  for (int base = 0; base < nun; base += 6) {
    if (!unit(base + 0, std::integral_constant<int, 0>{})) break;
    if (!unit(base + 1, std::integral_constant<int, 1>{})) break;
    if (!unit(base + 2, std::integral_constant<int, 2>{})) break;
    if (!unit(base + 3, std::integral_constant<int, 3>{})) break;
    if (!unit(base + 4, std::integral_constant<int, 4>{})) break;
    if (!unit(base + 5, std::integral_constant<int, 5>{})) break;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // slab store: native accumulator layout, one float4 per lane per tile (fully coalesced)
  float* sl = a.slab + ((((long long)zz * a.gy + blk.y) * a.gx + blk.x) * W3_WAVES + wave) * (W3_TILES * 256);
#pragma unroll
  for (int c = 0; c < 5; ++c)
#pragma unroll
    for (int j = 0; j < 7; ++j) *reinterpret_cast<f32x4*>(sl + ((c * 7 + j) * 64 + lane) * 4) = acc[c][j];
  // bias tile: rows 4g..4g+3 of co tile c live in every column of D; column l15 < 5 of the slot carries co tile l15
  if (a.has_bias && q == 0 && blk.y == 0) {
    f32x4 pack = f32x4{0.f, 0.f, 0.f, 0.f};
    if (li < 5) pack = li == 0 ? accb[0] : (li == 1 ? accb[1] : (li == 2 ? accb[2] : (li == 3 ? accb[3] : accb[4])));
    *reinterpret_cast<f32x4*>(sl + (35 * 64 + lane) * 4) = pack;
  }
}

__global__ __launch_bounds__(W3_THREADS) void conv_wgrad3b_kernel(const Wgrad2K a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wave >> 2, q = wave & 3;
  SlabBlock blk = xcd_slab_block(a.gx, a.gy, a.S * a.nprob);
  const int prob = blk.z / a.S;  // several problems of one shape share the launch: fewer, longer K slabs per problem at a full chip
  const int zz = blk.z;          // slab index over all problems (the slab workspace is [problem][S]...)
  blk.z -= prob * a.S;
  const char* const* xs = a.x + prob * WG_MAX_PAIRS;
  const char* const* dys = a.dy + prob * WG_MAX_PAIRS;
  const int ob = blk.x * W2_DYC, ib = blk.y * W2_XC;
  const long long u_lo = a.U * blk.z / a.S, u_hi = a.U * (blk.z + 1) / a.S;
  const char* zsrc = reinterpret_cast<const char*>(g_zero_buf);

  f32x4 acc[5][7];
#pragma unroll
  for (int c = 0; c < 5; ++c)
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[c][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // The bias gradient rides on the tile's PADDING: wave q = 3 owns fragment columns 21..27 of 27, its seventh column is a duplicate that the
  // reduce kernel drops.  With a ones operand in its place that accumulator column is sum_p dY[p][co] -- no accumulators of its own (20
  // registers) and no extra MFMAs (round 2's kernel issues five more per unit on the q = 0 waves of the first ci block: 80 instead of 70 on SIMD 0).
  const bf16 one = (bf16)1.0f;
  const bf16x8 ones = {one, one, one, one, one, one, one, one};
  const bool bias_col = q == 3;

  // ---- copy instructions (round 4).  conv_wgrad3_kernel above keeps a 64-bit POINTER per lane and copy instruction, selects it against a
  // zero buffer with per-unit bounds tests and advances it by wave-uniform 64-bit deltas: ~60 instructions per unit and wave, issued as ONE
  // block before or after the unit's 35 MFMAs -- while a wave is in that block it feeds the matrix pipe nothing (the unit cost MFMAs + block:
  // 0.92 us where the MFMAs alone take 0.50).  Here the copies are BUFFER loads (`buffer_load_dwordx4 ... offen lds`): the tensor base sits in a
  // resource descriptor (SGPRs), the unit's first pixel in an SGPR byte offset that advances by precomputed 32-bit deltas, and a lane's offset
  // inside the [dY | X] list is a CONSTANT 32-bit register.  Which lanes fall outside the image depends on four wave-uniform facts only (first /
  // last row, first / last segment of a row): a lane carries its static flags, a unit its bits, and a flagged lane asks for offset 2^31 -- beyond
  // num_records, so the hardware writes zeros.  3 VALU per copy instruction instead of ~12, no zero buffer, no 64-bit lane arithmetic; and the
  // issue block of a unit shrinks from ~60 to ~25 instructions.
  unsigned voff[W3_SLOTS], flg[W3_SLOTS];
  const int nun = (int)(u_hi - u_lo);
  int ipair = (int)(u_lo / a.Upair), iseg, iy, in_;
  {
    const long long uu = u_lo - (long long)ipair * a.Upair;
    iseg = (int)(uu % a.SEG);
    const long long r = uu / a.SEG;
    iy = (int)(r % a.H);
    in_ = (int)(r / a.H);
  }
  const int wlast = a.W - (a.SEG - 1) * 32;  // pixels of a row's last segment
#pragma unroll
  for (int sl = 0; sl < W3_SLOTS; ++sl) {
    const int L = (sl * W3_WAVES + wave) * 64 + lane;
    if (L < W3_DYV) {
      const int pp = L / 18, v = L - pp * 18;
      voff[sl] = (unsigned)((pp * a.dy_ps + ob + v * 8) * 2);
      flg[sl] = 32u | (pp >= wlast ? 8u : 0u) | ((ob + v * 8 + 8 <= a.Cout) ? 0u : 16u);
    } else if (L < W3_VECS) {
      const int vec = L - W3_DYV, pp = vec / 6, v = vec - pp * 6;
      const int rr = pp / W2_XW, col = pp - rr * W2_XW;
      voff[sl] = (unsigned)(((long long)(rr * a.W + col) * a.x_ps + ib + v * 8) * 2);  // from the descriptor's base = one row and one pixel before the unit
      flg[sl] = 32u | (rr == 0 ? 1u : 0u) | (rr == 2 ? 2u : 0u) | (col == 0 ? 4u : 0u) | (col - 1 >= wlast ? 8u : 0u) | ((ib + v * 8 + 8 <= a.Cin) ? 0u : 16u);
    } else {
      voff[sl] = 0u;
      flg[sl] = 48u;
    }
  }
  const long long halo_x = (long long)(a.W + 1) * a.x_ps * 2;
  w3_u32x4 rs_dy = w3_rsrc(dys[ipair]), rs_x = w3_rsrc(xs[ipair] - halo_x);
  unsigned soff_dy, soff_x;
  {
    const long long row0 = ((long long)in_ * a.H + iy) * a.W + iseg * 32;
    soff_dy = (unsigned)(row0 * a.dy_ps * 2);
    soff_x = (unsigned)(row0 * a.x_ps * 2);
  }
  const unsigned a32_dy = (unsigned)(32 * a.dy_ps * 2), a32_x = (unsigned)(32 * a.x_ps * 2);
  const unsigned arow_dy = (unsigned)(wlast * a.dy_ps * 2), arow_x = (unsigned)(wlast * a.x_ps * 2);  // last segment of a row -> first pixel of the next row
  // Every copy instruction is UNCONDITIONAL: a unit past the end of the slab is issued as a DUMMY -- bit 32, which every lane carries, is set in
  // the unit's bits and all 64 lanes ask for offset 2^31 (zeros land in a free slot, no memory is touched).  Every wave then issues exactly six
  // copies per even iteration and every counted wait is vmcnt(6).
  unsigned ubits = 0;
  auto issue_slot = [&](int sl, int buf, unsigned dummy) __attribute__((always_inline)) {
    if (sl == 0) ubits = dummy | 16u | (iy == 0 ? 1u : 0u) | (iy == a.H - 1 ? 2u : 0u) | (iseg == 0 ? 4u : 0u) | (iseg == a.SEG - 1 ? 8u : 0u);
    const bool isdy = sl * W3_WAVES + wave < W3_DYV / 64;  // wave-uniform: chunks 0..8 of the list are dY
    const unsigned vo = (flg[sl] & ubits) ? 0x80000000u : voff[sl];
    bufdma16_hidden(vo, isdy ? rs_dy : rs_x, isdy ? soff_dy : soff_x, smem + buf * W3_BUF + (sl * W3_WAVES + wave) * 1024);
  };
  auto advance = [&]() __attribute__((always_inline)) {  // the odometer: to the unit after the one just issued
    unsigned d_dy = a32_dy, d_x = a32_x;
    if (++iseg == a.SEG) {
      iseg = 0;
      d_dy = arow_dy; d_x = arow_x;
      if (++iy == a.H) {
        iy = 0;
        if (++in_ == a.N) {
          in_ = 0;
          ++ipair;
          if (ipair < a.npairs) {  // next (x, dy) pair: other tensors, from their pixel 0
            rs_dy = w3_rsrc(dys[ipair]);
            rs_x = w3_rsrc(xs[ipair] - halo_x);
            soff_dy = 0u - d_dy;
            soff_x = 0u - d_x;
          }
        }
      }
    }
    soff_dy += d_dy;
    soff_x += d_x;
  };
  auto issue = [&](int buf, bool live) __attribute__((always_inline)) {
#pragma unroll
    for (int sl = 0; sl < W3_SLOTS; ++sl) issue_slot(sl, buf, live ? 0u : 32u);
    if (live) advance();
  };

  // ---- fragment addresses (within a buffer): dY rows are 288 B, X rows 96 B; lane i = 4qq+pp of a 16-lane group supplies
  // block row qq, columns 4pp..4pp+3 of a transposed 4x16 block read
  const int li = lane & 15, lg = lane >> 4;
  const int a_off = (8 * lg + (li >> 2)) * (W2_DYC * 2) + (h * 5 * 16 + 4 * (li & 3)) * 2;  // + c * 32 for co tile c of this wave
  int b_off[7];
#pragma unroll
  for (int jj = 0; jj < 7; ++jj) {
    const int col = min(q * 7 + jj, 26), itile = col / 9, tap = col - 9 * itile, ky = tap / 3, kx = tap - 3 * ky;
    b_off[jj] = W3_DYV * 16 + ky * W2_XW * (W2_XC * 2) + (kx + 8 * lg + (li >> 2)) * (W2_XC * 2) + (itile * 16 + 4 * (li & 3)) * 2;
  }

  // SIX unit slots, one barrier per TWO units.  Unit j lives in slot j % 6; iteration j multiplies unit j (fragments already in
  // registers) and reads unit j+1's fragments.  At the barrier of an even j every wave has (a) waited for its share of units
  // j+1 and j+2 (units j+3, j+4 stay in flight: the wait is counted) and (b) drained its reads of unit j; afterwards units
  // j+5 and j+6 are issued into the slots of units j-1 and j.
  for (int j = 0; j < 5; ++j) issue(j, j < nun);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // units 3, 4 (real or dummy) may stay in flight
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  bf16x8 af[5], bfg[7];
  if (nun > 0) {
#pragma unroll
    for (int c = 0; c < 5; ++c) af[c] = tr_read(smem + a_off + c * 32, W2_DYC * 2);
#pragma unroll
    for (int jj = 0; jj < 7; ++jj) bfg[jj] = tr_read(smem + b_off[jj], W2_XC * 2);
    if (bias_col) bfg[6] = ones;
  }
  // one unit; SU = its slot (compile-time), j = its index.  Returns false when the slab is finished.
  auto unit = [&](int j, auto su_c) __attribute__((always_inline)) -> bool {
    constexpr int SU = decltype(su_c)::value, SN = (SU + 1) % 6, S5 = (SU + 5) % 6;
    constexpr bool SYNC = (SU & 1) == 0;  // slots alternate with the unit index: even units synchronise
    if (j >= nun) return false;
    if (SYNC) {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // units j + 1, j + 2 have landed; j + 3, j + 4 (real or dummy) stay in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of unit j are complete
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    const char* src = smem + ((j + 1 < nun) ? SN : SU) * W3_BUF;  // (last unit: re-read its own slot; the values are not used)
    // units j+5, j+6 -> the slots of units j-1 and j (free since this unit's barrier), as in conv_wgrad3_kernel: h = 0 issues its copies BEFORE
    // its MFMAs, h = 1 after them.  (Issuing one unit's copies in EVERY iteration, or the copy instructions between the MFMA columns, made hipcc
    // spill the accumulators -- 3 KB of scratch per lane, 25x slower; the same happens to conv_wgrad3_kernel with that change alone.)
#ifdef VMG_DIAG
    const bool nocopy = a.dbg & 1;
#else
    constexpr bool nocopy = false;
#endif
    if (SYNC && h == 0) {
      issue(S5, j + 5 < nun && !nocopy);
      issue(SU, j + 6 < nun && !nocopy);
    }
#pragma unroll
    for (int jj = 0; jj < 7; ++jj) {
#ifdef VMG_DIAG
      if (!(a.dbg & 2))
#endif
      {
#pragma unroll
        for (int c = 0; c < 5; ++c) acc[c][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c], bfg[jj], acc[c][jj], 0, 0, 0);
      }
#ifdef VMG_DIAG
      if (!(a.dbg & 4))
#endif
      if (jj < 6 || !bias_col) bfg[jj] = tr_read(src + b_off[jj], W2_XC * 2);  // (q = 3: column 6 stays the ones operand; a wave-uniform skip of two reads)
    }
#ifdef VMG_DIAG
    if (!(a.dbg & 4))
#endif
    {
#pragma unroll
      for (int c = 0; c < 5; ++c) af[c] = tr_read(src + a_off + c * 32, W2_DYC * 2);  // (behind the last MFMA that reads af[c])
    }
    if (SYNC && h == 1) {
      issue(S5, j + 5 < nun && !nocopy);
      issue(SU, j + 6 < nun && !nocopy);
    }
    return true;
  };
  for (int base = 0; base < nun; base += 6) {
    if (!unit(base + 0, std::integral_constant<int, 0>{})) break;
    if (!unit(base + 1, std::integral_constant<int, 1>{})) break;
    if (!unit(base + 2, std::integral_constant<int, 2>{})) break;
    if (!unit(base + 3, std::integral_constant<int, 3>{})) break;
    if (!unit(base + 4, std::integral_constant<int, 4>{})) break;
    if (!unit(base + 5, std::integral_constant<int, 5>{})) break;
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");  // (the dummy copies of the last iterations too: no LDS-DMA outlives the loop)
  // slab store: native accumulator layout, one float4 per lane per tile (fully coalesced)
  float* sl = a.slab + ((((long long)zz * a.gy + blk.y) * a.gx + blk.x) * W3_WAVES + wave) * (W3_TILES * 256);
#pragma unroll
  for (int c = 0; c < 5; ++c)
#pragma unroll
    for (int j = 0; j < 7; ++j) *reinterpret_cast<f32x4*>(sl + ((c * 7 + j) * 64 + lane) * 4) = acc[c][j];
  // (bias: column 6 of the q = 3 waves, see above; slot 35 of the wave's slab stays unused)
}

__global__ __launch_bounds__(256) void conv_wgrad3_reduce_kernel(const float* __restrict__ slab0, int S, int gx, int gy, int Cin, int Cout,
                                                                 const Wgrad3Out outs, int I_total, int o0, int i0, int bias_in_pad) {
  __shared__ float red[256];
  const long long per_s = (long long)gy * gx * W3_WG_FLOATS;
  const float* __restrict__ slab = slab0 + (long long)blockIdx.y * S * per_s;  // blockIdx.y = problem
  float* __restrict__ dW = outs.dW[blockIdx.y];
  float* __restrict__ db = outs.db[blockIdx.y];
  const float scale = outs.scale[blockIdx.y];
  for (long long e0 = blockIdx.x * 64LL; e0 < per_s; e0 += (long long)gridDim.x * 64) {
    const long long i = e0 + (threadIdx.x & 63);
    const float sum = slab_sum_4waves(slab, per_s, S, i, true, red);
    if (threadIdx.x >= 64) continue;
    const int r = (int)(i & 3);
    const int lane = (int)((i >> 2) & 63);
    long long qq = i >> 8;
    const int tile = (int)(qq % W3_TILES);
    qq /= W3_TILES;
    const int wave = (int)(qq % W3_WAVES);
    qq /= W3_WAVES;
    const int coblk = (int)(qq % gx), ciblk = (int)(qq / gx);
    const int h = wave >> 2, q = wave & 3;
    const int g = lane >> 4, l15 = lane & 15;
    if (tile < 35) {
      const int c = tile / 7, j = tile - c * 7;
      const int cot = h * 5 + c, col = q * 7 + j;
      if (cot < 9 && col < 27) {
        const int itile = col / 9, t = col - itile * 9;
        const int co = coblk * W2_DYC + cot * 16 + 4 * g + r;
        const int ci = ciblk * W2_XC + itile * 16 + l15;
        if (co < Cout && ci < Cin) dW[((long long)(o0 + co) * I_total + (i0 + ci)) * 9 + t] += sum * scale;
      }
    } else if (!bias_in_pad && db && q == 0 && ciblk == 0 && l15 < 5) {
      const int cot = h * 5 + l15;
      const int co = coblk * W2_DYC + cot * 16 + 4 * g + r;
      if (cot < 9 && co < Cout) db[o0 + co] += sum * scale;
    }
    if (bias_in_pad && db && tile < 35 && q == 3 && ciblk == 0 && l15 == 0 && tile % 7 == 6) {  // conv_wgrad3b_kernel: the padding column of the q = 3 waves
      const int cot = h * 5 + tile / 7;
      const int co = coblk * W2_DYC + cot * 16 + 4 * g + r;
      if (cot < 9 && co < Cout) db[o0 + co] += sum * scale;
    }
  }
}

// =====================================================================================================
// v2 for 1x1 convolutions / Linears (bf16): dW[co][ci] = sum_p dY[p][co] * X[p][ci], a GEMM whose K is the pixel list.
// Workgroup = 9 waves (cg, it); output tile 144 co x 144 ci, wave (cg, it) owns co tiles 3cg..3cg+2 x ci tiles 3it..3it+2
// (9 accumulator tiles + the bias tile for it == 0).  A unit = 32 consecutive pixels of the flat pixel list: both tiles
// ([32][144] bf16, 9 KiB each) arrive by LDS-DMA exactly as they lie in HBM, one 16-byte vector per thread and tile, and
// are read TRANSPOSED (ds_read_b64_tr_b16).  Three LDS buffers, one barrier per unit: at the top of iteration u a wave waits for its own share of unit u, and the
// barrier also tells it that nobody reads unit u-1's buffer any more, so unit u+2 may overwrite it; the LDS
// footprint (54 KiB) and register use allow two workgroups per CU, whose barrier bubbles overlap.  The pixel list is
// cut into S slabs; partial tiles go to the workspace and the reduce kernel adds them to dW in slab order.
struct Lgrad2K {
  const char* x[W3_MAX_PROBS * WG_MAX_PAIRS];   // [problem][pair]
  const char* dy[W3_MAX_PROBS * WG_MAX_PAIRS];
  int npairs, nprob;
  long long Mpair;   // pixels per pair
  long long Upair, U;  // 32-pixel units per pair / in total
  long long x_ps, dy_ps;
  int Cin, Cout;
  float* slab;  // [problem][S][ciblk][coblk][9 waves][10 tiles][64][4]
  int S, has_bias;
  int gx, gy;   // co / ci blocks (1-D grid of gx * gy * S workgroups, see xcd_slab_block)
};
constexpr int L2_TILE_BYTES = 32 * 144 * 2;  // 9216
constexpr int L2_BUF = 2 * L2_TILE_BYTES;    // dY tile + X tile
constexpr int L2_TILES = 10;
constexpr long long L2_WG_FLOATS = (long long)W2_WAVES * L2_TILES * 256;

__global__ __launch_bounds__(W2_THREADS, 2) void linear_wgrad2_kernel(const Lgrad2K a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cg = wave / 3, it = wave - cg * 3;
  SlabBlock blk = xcd_slab_block(a.gx, a.gy, a.S * a.nprob);
  const int prob = blk.z / a.S, zz = blk.z;  // (several problems of one shape per launch, as in conv_wgrad3_kernel)
  blk.z -= prob * a.S;
  const char* const* xs = a.x + prob * WG_MAX_PAIRS;
  const char* const* dys = a.dy + prob * WG_MAX_PAIRS;
  const int ob = blk.x * 144, ib = blk.y * 144;
  const long long u_lo = a.U * blk.z / a.S, u_hi = a.U * (blk.z + 1) / a.S;
  const char* zsrc = reinterpret_cast<const char*>(g_zero_buf);

  f32x4 acc[3][3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[c][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 accb[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const bf16 one = (bf16)1.0f;
  const bf16x8 ones = {one, one, one, one, one, one, one, one};

  // lane constants: thread tid moves vector v (8 channels) of pixel p of both tiles
  const int p_l = tid / 18, v_l = tid - p_l * 18;
  const bool c_dy = ob + v_l * 8 + 8 <= a.Cout, c_x = ib + v_l * 8 + 8 <= a.Cin;
  const int off_dy = (int)((p_l * a.dy_ps + ob + v_l * 8) * 2), off_x = (int)((p_l * a.x_ps + ib + v_l * 8) * 2);
  int ipair = (int)(u_lo / a.Upair);
  long long ipix = (u_lo - (long long)ipair * a.Upair) * 32;  // first pixel (within the pair) of the unit to issue next
  auto issue = [&](int buf) {
    const bool pok = ipix + p_l < a.Mpair;
    char* dyt = smem + buf * L2_BUF;
    const char* s0 = (pok && c_dy) ? dys[ipair] + ipix * a.dy_ps * 2 + off_dy : zsrc;
    glds16_hidden(s0, dyt + wave * 1024);
    const char* s1 = (pok && c_x) ? xs[ipair] + ipix * a.x_ps * 2 + off_x : zsrc;
    glds16_hidden(s1, dyt + L2_TILE_BYTES + wave * 1024);
    ipix += 32;
    if (ipix >= a.Mpair) { ipix = 0; ++ipair; }
  };

  if (u_lo < u_hi) issue(0);
  if (u_lo + 1 < u_hi) issue(1);
  int buf = 0;
  for (long long u = u_lo; u < u_hi; ++u) {
    if (u + 1 < u_hi) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // unit u landed; unit u+1 (2 instructions per wave) stays in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* dyt = smem + buf * L2_BUF;
    const char* xt = dyt + L2_TILE_BYTES;
    bf16x8 af[3], bfg[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) af[c] = tr_frag(dyt, 288, 0, (cg * 3 + c) * 16, lane);
#pragma unroll
    for (int j = 0; j < 3; ++j) bfg[j] = tr_frag(xt, 288, 0, (it * 3 + j) * 16, lane);
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c], bfg[j], acc[c][j], 0, 0, 0);
    if (a.has_bias && it == 0 && blk.y == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) accb[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[c], ones, accb[c], 0, 0, 0);
    }
    if (u + 2 < u_hi) issue(buf == 0 ? 2 : buf - 1);  // (behind the MFMAs: the copy's address arithmetic runs in their shadow)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    buf = buf == 2 ? 0 : buf + 1;
  }
  float* sl = a.slab + ((((long long)zz * a.gy + blk.y) * a.gx + blk.x) * W2_WAVES + wave) * (L2_TILES * 256);
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4*>(sl + ((c * 3 + j) * 64 + lane) * 4) = acc[c][j];
  if (a.has_bias && it == 0 && blk.y == 0) {
    f32x4 pack = f32x4{0.f, 0.f, 0.f, 0.f};
    const int l15 = lane & 15;
    if (l15 < 3) pack = l15 == 0 ? accb[0] : (l15 == 1 ? accb[1] : accb[2]);
    *reinterpret_cast<f32x4*>(sl + (9 * 64 + lane) * 4) = pack;
  }
}

__global__ __launch_bounds__(256) void linear_wgrad2_reduce_kernel(const float* __restrict__ slab0, int S, int gx, int gy, int Cin, int Cout,
                                                                   const Wgrad3Out outs, int I_total, int o0, int i0) {
  __shared__ float red[256];
  const long long per_s = (long long)gy * gx * L2_WG_FLOATS;
  const float* __restrict__ slab = slab0 + (long long)blockIdx.y * S * per_s;  // blockIdx.y = problem
  float* __restrict__ dW = outs.dW[blockIdx.y];
  float* __restrict__ db = outs.db[blockIdx.y];
  const float scale = outs.scale[blockIdx.y];
  for (long long e0 = blockIdx.x * 64LL; e0 < per_s; e0 += (long long)gridDim.x * 64) {
    const long long i = e0 + (threadIdx.x & 63);
    const float sum = slab_sum_4waves(slab, per_s, S, i, true, red);
    if (threadIdx.x >= 64) continue;
    const int r = (int)(i & 3);
    const int lane = (int)((i >> 2) & 63);
    long long q = i >> 8;
    const int tile = (int)(q % L2_TILES);
    q /= L2_TILES;
    const int wave = (int)(q % W2_WAVES);
    q /= W2_WAVES;
    const int coblk = (int)(q % gx), ciblk = (int)(q / gx);
    const int cg = wave / 3, it = wave - cg * 3;
    const int g = lane >> 4, l15 = lane & 15;
    if (tile < 9) {
      const int c = tile / 3, j = tile - c * 3;
      const int co = coblk * 144 + (cg * 3 + c) * 16 + 4 * g + r;
      const int ci = ciblk * 144 + (it * 3 + j) * 16 + l15;
      if (co < Cout && ci < Cin) dW[(long long)(o0 + co) * I_total + (i0 + ci)] += sum * scale;
    } else if (db && it == 0 && ciblk == 0 && l15 < 3) {
      const int co = coblk * 144 + (cg * 3 + l15) * 16 + 4 * g + r;
      if (co < Cout) db[o0 + co] += sum * scale;
    }
  }
}

// =====================================================================================================
// 7x7 (SPyNet's ConvModules, reference models/vmg.py:126-173), bf16: dW[co][ci][ky][kx] = sum_p dY[p][co] * X[p + (ky-3, kx-3)][ci].
// Workgroup = SEVEN waves, wave ky owns tap row ky: CT output-channel tiles x 7 kx = 7*CT accumulator tiles of one 16-channel ci
// block (grid.x = ci blocks, grid.y = K slabs).  A K unit = W7_R image rows x 32 pixels: the dY tile (R x 32 pixels x CT*16
// channels) and the X tile with its 3-pixel halo ((R + 6) x 38 pixels x 16 channels) are fetched by all 448 threads with
// UNCONDITIONAL 16-byte loads (lanes outside the image / past the channels read a zero buffer) one unit ahead into registers,
// written to one of two LDS buffers, one barrier per unit.  Per image row a wave reads CT dY fragments and 7 X fragments (the
// same staged row serves all kx: the fragment's first pixel is kx) transposed with ds_read_b64_tr_b16 for 7*CT MFMAs.  Every
// (co, ci, tap) lives in exactly one wave: no reduction inside the workgroup; slab partials in the accumulators' native layout
// (coalesced float4 stores) and a second kernel sums the slabs in a fixed order (deterministic) into dW / db.  The bias
// gradient is one more MFMA per co tile against a ones vector on wave 0 of ci block 0.
// The same kernel with KS = 3 (three waves, 8 rows per unit) takes the 3x3 convs with <= 16 output channels (conv_last: 64 -> 3 on the 1.8 M
// HR pixels of a batch), where the 144-channel tile of conv_wgrad3_kernel would be 9/10 padding.
constexpr int W7_X_RS = 48;  // LDS pixel stride of the X tile: 16 channels + 16 bytes (conflict-free transposed reads)
template <int KS>
struct W7Geo {
  static constexpr int R = KS == 7 ? 4 : 8;  // image rows per K unit (3x3 with 4 rows and 768 workgroups: 140 us instead of 122 on conv_last)
  static constexpr int THREADS = KS * 64, XW = 32 + KS - 1, XR = R + KS - 1;
  static constexpr int XVEC = XR * XW * 2;  // 16-byte vectors of the X tile (7x7: 760)
  static constexpr int NX = (XVEC + THREADS - 1) / THREADS;
  static constexpr int X_BYTES = ((NX * THREADS + 1) / 2) * W7_X_RS;  // room for every thread's NX vectors: the staging stores are unconditional
};
template <int CT, int KS = 7>
struct W7Cfg {
  using G = W7Geo<KS>;
  static constexpr int DY_RS = CT * 32 + 16, DYVEC = G::R * 32 * CT * 2;
  static constexpr int NX = G::NX, NDY = (DYVEC + G::THREADS - 1) / G::THREADS;
  static constexpr int DY_BYTES = ((NDY * G::THREADS + CT * 2 - 1) / (CT * 2)) * DY_RS;  // (same: room for the padding vectors)
  static constexpr int BUF = (G::X_BYTES + DY_BYTES + 15) & ~15;
  static constexpr int WG_FLOATS = (KS * KS * CT + CT) * 256;  // tiles [ky][ct][kx], then the CT bias tiles
};

struct Wgrad7K {
  const char* x[WG_MAX_PAIRS];
  const char* dy[WG_MAX_PAIRS];
  int npairs;
  int Upair, U;       // K units per pair / in total
  long long x_ps, dy_ps;
  int Cin, Cout;      // true channel counts
  int vec_dy;         // dY can be read as 16-byte vectors (else element by element: the 2-channel flow head)
  float* slab;        // [S][ci blocks][WG_FLOATS]
  int N, H, W, SEG, HB, S;
  int has_bias;
  int gx;             // ci blocks (1-D grid of gx * S workgroups, see xcd_slab_block)
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int CT, int KS>
__global__ __launch_bounds__(W7Geo<KS>::THREADS) void conv_wgrad7_kernel(const Wgrad7K a) {
  using C = W7Cfg<CT, KS>;
  using G = W7Geo<KS>;
  constexpr int W7_THREADS = G::THREADS, W7_XW = G::XW, W7_XVEC = G::XVEC, W7_X_BYTES = G::X_BYTES, W7_RU = G::R, HALO = KS / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int ky = __builtin_amdgcn_readfirstlane(tid >> 6);
  const SlabBlock blk = xcd_slab_block(a.gx, 1, a.S);
  const int ib = blk.x * 16;
  const int u_lo = (int)((long long)a.U * blk.z / a.S), u_hi = (int)((long long)a.U * (blk.z + 1) / a.S);
  const bool do_bias = a.has_bias && blk.x == 0 && ky == 0;
  const char* zsrc = reinterpret_cast<const char*>(g_zero_buf);

  f32x4 acc[CT][KS], accb[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    accb[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) acc[ct][kx] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const __bf16 one = (__bf16)1.0f;
  const bf16x8 ones = {one, one, one, one, one, one, one, one};

  u32x4 rx[C::NX], rdy[C::NDY];  // (a native vector type: the HIP uint4 struct is copied with memcpy and stays in scratch)
  auto fetch = [&](int ug) __attribute__((always_inline)) {
    const int pair = ug / a.Upair;
    int u = ug - pair * a.Upair;
    const int seg = u % a.SEG;
    u /= a.SEG;
    const int hb = u % a.HB, n = u / a.HB;
    const int y0 = hb * W7_RU, x0 = seg * 32;
    const char* xb = a.x[pair];
    const char* db_ = a.dy[pair];
#pragma unroll
    for (int k = 0; k < C::NX; ++k) {
      const int idx = tid + k * W7_THREADS;
      const int pp = idx >> 1, v = idx & 1;
      const int r = pp / W7_XW, col = pp - r * W7_XW;
      const int yy = y0 + r - HALO, xx = x0 + col - HALO, c = ib + v * 8;
      const bool ok = idx < W7_XVEC && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W && c + 8 <= a.Cin;
      const long long off = ((((long long)n * a.H + yy) * a.W + xx) * a.x_ps + c) * 2;
      rx[k] = *reinterpret_cast<const u32x4*>(ok ? xb + off : zsrc);
    }
    if (a.vec_dy) {
#pragma unroll
      for (int k = 0; k < C::NDY; ++k) {
        const int idx = tid + k * W7_THREADS;
        const int pp = idx / (CT * 2), v = idx - pp * (CT * 2);
        const int rr = pp >> 5, col = pp & 31;
        const bool ok = idx < C::DYVEC && y0 + rr < a.H && x0 + col < a.W && v * 8 < a.Cout;  // (a vector that starts inside the channels ends inside the pixel stride)
        const long long off = ((((long long)n * a.H + y0 + rr) * a.W + x0 + col) * a.dy_ps + v * 8) * 2;
        rdy[k] = *reinterpret_cast<const u32x4*>(ok ? db_ + off : zsrc);
      }
    } else {
#pragma unroll
      for (int k = 0; k < C::NDY; ++k) {
        const int idx = tid + k * W7_THREADS;
        const int pp = idx / (CT * 2), v = idx - pp * (CT * 2);
        const int rr = pp >> 5, col = pp & 31;
        const bool ok = idx < C::DYVEC && y0 + rr < a.H && x0 + col < a.W;
        const long long off = ((((long long)n * a.H + y0 + rr) * a.W + x0 + col) * a.dy_ps + v * 8) * 2;
        unsigned short tmp[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) tmp[e] = *reinterpret_cast<const unsigned short*>((ok && v * 8 + e < a.Cout) ? db_ + off + 2 * e : zsrc);
        rdy[k] = u32x4{tmp[0] | (unsigned)tmp[1] << 16, tmp[2] | (unsigned)tmp[3] << 16, tmp[4] | (unsigned)tmp[5] << 16, tmp[6] | (unsigned)tmp[7] << 16};
      }
    }
  };
  auto stage = [&](char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < C::NX; ++k) {
      const int idx = tid + k * W7_THREADS;
      *reinterpret_cast<u32x4*>(buf + (idx >> 1) * W7_X_RS + (idx & 1) * 16) = rx[k];
    }
#pragma unroll
    for (int k = 0; k < C::NDY; ++k) {
      const int idx = tid + k * W7_THREADS;
      const int pp = idx / (CT * 2), v = idx - pp * (CT * 2);
      *reinterpret_cast<u32x4*>(buf + W7_X_BYTES + pp * C::DY_RS + v * 16) = rdy[k];
    }
  };

  int which = 0;
  if (u_lo < u_hi) fetch(u_lo);
  for (int u = u_lo; u < u_hi; ++u) {
    char* buf = smem + which * C::BUF;
    stage(buf);
    __syncthreads();  // the unit is complete in `buf`; every wave has finished the unit before the previous one (the other buffer is free again after the NEXT barrier)
    if (u + 1 < u_hi) fetch(u + 1);
    const char* xt = buf;
    const char* dyt = buf + W7_X_BYTES;
#pragma unroll
    for (int rr = 0; rr < W7_RU; ++rr) {
      bf16x8 af[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) af[ct] = tr_frag(dyt, C::DY_RS, rr * 32, ct * 16, lane);
      const char* xrow = xt + (rr + ky) * (W7_XW * W7_X_RS);
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const bf16x8 bfg = tr_frag(xrow, W7_X_RS, kx, 0, lane);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct][kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct], bfg, acc[ct][kx], 0, 0, 0);
      }
      if (do_bias) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) accb[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct], ones, accb[ct], 0, 0, 0);
      }
    }
    which ^= 1;
  }

  float* sl = a.slab + ((long long)blk.z * a.gx + blk.x) * C::WG_FLOATS;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int kx = 0; kx < KS; ++kx)
      *reinterpret_cast<f32x4*>(sl + (((ky * CT + ct) * KS + kx) * 64 + lane) * 4) = acc[ct][kx];
  if (ky == 0) {  // (written by every ci block so that the reduce kernel never reads uninitialised memory)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) *reinterpret_cast<f32x4*>(sl + ((KS * KS * CT + ct) * 64 + lane) * 4) = accb[ct];
  }
}

__global__ __launch_bounds__(256) void conv_wgrad7_reduce_kernel(const float* __restrict__ slab, int S, int gx, int CT, int KS, int Cin, int Cout,
                                                                 float* __restrict__ dW, int I_total, int o0, int i0,
                                                                 float* __restrict__ db, float scale) {
  __shared__ float red[256];
  const int KK = KS * KS, wgf = (KK * CT + CT) * 256;
  const long long per_s = (long long)gx * wgf;
  for (long long e0 = blockIdx.x * 64LL; e0 < per_s; e0 += (long long)gridDim.x * 64) {
    const long long i = e0 + (threadIdx.x & 63);
    const float sum = slab_sum_4waves(slab, per_s, S, i, true, red);
    if (threadIdx.x >= 64) continue;
    const int ciblk = (int)(i / wgf);
    const int e = (int)(i - (long long)ciblk * wgf);
    const int r = e & 3, lane = (e >> 2) & 63, tile = e >> 8;
    const int g = lane >> 4, l15 = lane & 15;
    if (tile < KK * CT) {
      const int kx = tile % KS, ct = (tile / KS) % CT, ky = tile / (KS * CT);
      const int co = ct * 16 + 4 * g + r, ci = ciblk * 16 + l15;
      if (co < Cout && ci < Cin) dW[((long long)(o0 + co) * I_total + (i0 + ci)) * KK + ky * KS + kx] += sum * scale;
    } else if (db && ciblk == 0 && l15 == 0) {
      const int co = (tile - KK * CT) * 16 + 4 * g + r;
      if (co < Cout) db[o0 + co] += sum * scale;
    }
  }
}

template <int CT, int KS = 7>
int launch_wgrad7(Wgrad7K k, float* dW, int I_total, int o0, int i0, float* db, float scale, int64_t ws_bytes, hipStream_t st) {
  using C = W7Cfg<CT, KS>;
  const int gx = cdiv(k.Cin, 16);
  long long S = 512 / gx;            // ~two rounds of workgroups over the 256 CUs
  if (S > k.U / 2) S = k.U / 2;      // >= 2 units per workgroup
  const long long cap = ws_bytes / ((long long)gx * C::WG_FLOATS * 4);
  if (S > cap) S = cap;
  if (S < 1) S = 1;
  if (cap < 1) return 1;  // (workspace too small: the caller falls back)
  k.S = (int)S; k.gx = gx;
  auto fn = conv_wgrad7_kernel<CT, KS>;
  static bool attr7[VMG_MAX_DEVICES] = {};  // per instantiation and device
  const int dev = vmg_current_device();
  if (!attr7[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * C::BUF);
    attr7[dev] = true;
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)(gx * S)), dim3(W7Geo<KS>::THREADS), 2 * C::BUF, st, k);
  VMG_LAUNCH_CHECK();
  const long long per_s = (long long)gx * C::WG_FLOATS;
  const int rb = (int)(cdiv64(per_s, 64) > 8192 ? 8192 : cdiv64(per_s, 64));
  hipLaunchKernelGGL(conv_wgrad7_reduce_kernel, dim3(rb), dim3(256), 0, st, (const float*)k.slab, (int)S, gx, CT, KS, k.Cin, k.Cout, dW, I_total, o0, i0, db, scale);
  VMG_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// Shared by the single and the multi-problem entry points.  Returns 1 when the workspace cannot hold the slabs (caller falls back).
static int g_w3_variant = 1;  // 1: conv_wgrad3b_kernel (buffer-addressed copies between the MFMA columns), 0: conv_wgrad3_kernel
extern "C" int vmg_conv_wgrad3_variant(int v) {
  const int prev = g_w3_variant;
  if (v == 0 || v == 1) g_w3_variant = v;
  return prev;
}

static int launch_wgrad3(int nprob, int npairs, const void* const* x, const void* const* dy, int N, int H, int W, int64_t x_ps, int Cin, int64_t dy_ps,
                         int Cout, float* const* dW, int I_total, int o0, int i0, float* const* db, const float* scales, void* ws, int64_t ws_bytes,
                         void* stream) {
  const int cin8 = (Cin + 7) & ~7, cout8 = (Cout + 7) & ~7;
  Wgrad2K k;
  memset(&k, 0, sizeof(k));
  Wgrad3Out outs;
  memset(&outs, 0, sizeof(outs));
  bool any_bias = false;
  for (int q = 0; q < nprob; ++q) {
    for (int p = 0; p < npairs; ++p) {
      k.x[q * WG_MAX_PAIRS + p] = (const char*)x[q * npairs + p];
      k.dy[q * WG_MAX_PAIRS + p] = (const char*)dy[q * npairs + p];
    }
    outs.dW[q] = dW[q];
    outs.db[q] = db ? db[q] : nullptr;
    outs.scale[q] = scales[q];
    any_bias = any_bias || outs.db[q] != nullptr;
  }
  k.npairs = npairs; k.nprob = nprob; k.x_ps = x_ps; k.dy_ps = dy_ps; k.Cin = cin8; k.Cout = cout8;  // bounds of the 8-channel vector loads; the reduce kernel keeps the true counts
  k.N = N; k.H = H; k.W = W; k.SEG = cdiv(W, 32);
  k.Upair = (long long)N * H * k.SEG; k.U = k.Upair * npairs;
  k.has_bias = any_bias;
  const int gx = cdiv(Cout, W2_DYC), gy = cdiv(Cin, W2_XC);
  long long S = 256 / ((long long)gx * gy * nprob);  // one workgroup per CU over all problems
  if (S > k.U / 8) S = k.U / 8;
  if (S < 1) S = 1;
  const long long need = (long long)nprob * S * gx * gy * W3_WG_FLOATS * 4;
  if (need > ws_bytes) return 1;
  hipStream_t st = (hipStream_t)stream;
  k.S = (int)S; k.slab = (float*)ws; k.gx = gx; k.gy = gy;
#ifdef VMG_DIAG
  { const char* e = getenv("VMG_WGRAD_DBG"); k.dbg = e ? atoi(e) : 0; }
#endif
  static bool attr3[VMG_MAX_DEVICES] = {};  // the attribute is per device
  const int dev3 = vmg_current_device();
  if (!attr3[dev3]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr3[dev3] = true;
  }
  // the buffer-addressed variant needs every byte of a pair's tensors (plus the X tile's reach of two rows and 34 pixels) below 2^31 from the base
  int bias_in_pad = 0;
  const long long reach = ((long long)N * H * W + 2LL * W + 40) * (x_ps > dy_ps ? x_ps : dy_ps) * 2;
  if (g_w3_variant == 1 && reach < (1LL << 31)) {
    static bool attr3b[VMG_MAX_DEVICES] = {};
    if (!attr3b[dev3]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad3b_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr3b[dev3] = true;
    }
    hipLaunchKernelGGL(conv_wgrad3b_kernel, dim3((unsigned)(gx * gy * S * nprob)), dim3(W3_THREADS), 6 * W3_BUF, st, k);
    bias_in_pad = 1;
  } else {
    hipLaunchKernelGGL(conv_wgrad3_kernel, dim3((unsigned)(gx * gy * S * nprob)), dim3(W3_THREADS), 6 * W3_BUF, st, k);
  }
  VMG_LAUNCH_CHECK();
  const long long per3 = (long long)gy * gx * W3_WG_FLOATS;
  const int rb3 = (int)(cdiv64(per3, 64) > 8192 ? 8192 : cdiv64(per3, 64));
  hipLaunchKernelGGL(conv_wgrad3_reduce_kernel, dim3(rb3, nprob), dim3(256), 0, st, (const float*)ws, (int)S, gx, gy, Cin, Cout, outs, I_total, o0, i0, bias_in_pad);
  VMG_LAUNCH_CHECK();
  return 0;
}

// 1x1 / Linear weight gradients: one or several problems of one shape per launch.  Returns 1 when the large-tile path does not apply
// (workspace too small, tiny problem): the caller falls back.
static int launch_lgrad2(int nprob, int npairs, const void* const* x, const void* const* dy, long long Mpair, int64_t x_ps, int Cin, int64_t dy_ps,
                         int Cout, float* const* dW, int I_total, int o0, int i0, float* const* db, const float* scales, void* ws,
                         int64_t ws_bytes, void* stream) {
  Lgrad2K k;
  memset(&k, 0, sizeof(k));
  Wgrad3Out outs;
  memset(&outs, 0, sizeof(outs));
  k.Mpair = Mpair;
  k.Upair = (k.Mpair + 31) / 32; k.U = k.Upair * npairs;
  const int gx = cdiv(Cout, 144), gy = cdiv(Cin, 144);
  long long S = 512 / ((long long)gx * gy * nprob);  // two workgroups per CU over all problems
  if (S > k.U / 8) S = k.U / 8;
  if (S < 1) S = 1;
  const long long need = (long long)nprob * S * gx * gy * L2_WG_FLOATS * 4;
  if (need > ws_bytes || k.U < 64) return 1;  // (tiny problems: the v1 kernel's single launch is cheaper)
  bool any_bias = false;
  for (int q = 0; q < nprob; ++q) {
    for (int p = 0; p < npairs; ++p) {
      k.x[q * WG_MAX_PAIRS + p] = (const char*)x[q * npairs + p];
      k.dy[q * WG_MAX_PAIRS + p] = (const char*)dy[q * npairs + p];
    }
    outs.dW[q] = dW[q];
    outs.db[q] = db ? db[q] : nullptr;
    outs.scale[q] = scales[q];
    any_bias = any_bias || outs.db[q] != nullptr;
  }
  k.npairs = npairs; k.nprob = nprob; k.x_ps = x_ps; k.dy_ps = dy_ps; k.Cin = Cin; k.Cout = Cout; k.S = (int)S; k.slab = (float*)ws;
  k.has_bias = any_bias;
  k.gx = gx; k.gy = gy;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(linear_wgrad2_kernel, dim3((unsigned)(gx * gy * S * nprob)), dim3(W2_THREADS), 3 * L2_BUF, st, k);
  VMG_LAUNCH_CHECK();
  const long long per_s = (long long)gy * gx * L2_WG_FLOATS;
  const int rb = (int)(cdiv64(per_s, 64) > 8192 ? 8192 : cdiv64(per_s, 64));
  hipLaunchKernelGGL(linear_wgrad2_reduce_kernel, dim3(rb, nprob), dim3(256), 0, st, (const float*)ws, (int)S, gx, gy, Cin, Cout, outs, I_total, o0, i0);
  VMG_LAUNCH_CHECK();
  return 0;
}

// Several 1x1 / Linear weight gradients of ONE shape in one launch (the token mixers and projections of a TAB stage): x / dy hold
// nprob * npairs pointers [problem][pair] over M pixels each, dW (O_total, I_total) / db one pointer and scales one factor per problem.
extern "C" int vmg_linear_wgrad2_multi(int nprob, int npairs, const void* const* x, const void* const* dy, int64_t M, int64_t x_ps, int Cin,
                                       int64_t dy_ps, int Cout, float* const* dW, int I_total, int o0, int i0, float* const* db,
                                       const float* scales, void* ws, int64_t ws_bytes, void* stream) {
  VMG_CHECK(nprob >= 1 && nprob <= W3_MAX_PROBS && npairs >= 1 && npairs <= WG_MAX_PAIRS && x && dy && dW && ws && scales, "linear_wgrad2_multi: 1..%d problems, 1..%d pairs",
            W3_MAX_PROBS, WG_MAX_PAIRS);
  VMG_CHECK(M > 0 && (x_ps % 8 == 0) && (dy_ps % 8 == 0) && (Cin % 8 == 0) && (Cout % 8 == 0) && x_ps >= Cin && dy_ps >= Cout && i0 >= 0 && i0 + Cin <= I_total && o0 >= 0,
            "linear_wgrad2_multi: bf16 with 8-channel vectors only");
  for (int p = 0; p < nprob * npairs; ++p)
    VMG_CHECK(x[p] && dy[p] && ((uintptr_t)x[p] % 16 == 0) && ((uintptr_t)dy[p] % 16 == 0), "linear_wgrad2_multi: null or unaligned pointer %d", p);
  for (int p = 0; p < nprob; ++p) VMG_CHECK(dW[p], "linear_wgrad2_multi: null gradient pointer %d", p);
  const int rc = launch_lgrad2(nprob, npairs, x, dy, M, x_ps, Cin, dy_ps, Cout, dW, I_total, o0, i0, db, scales, ws, ws_bytes, stream);
  VMG_CHECK(rc != 1, "linear_wgrad2_multi: problem too small for the large-tile kernel or workspace too small (use vmg_conv_wgrad_batched_ws)");
  return rc;
}

extern "C" int64_t vmg_conv_wgrad_ws_bytes(void) { return 320LL * W3_WG_FLOATS * 4; }  // up to 320 workgroups of slabs (~94 MB)

extern "C" int vmg_conv_wgrad_batched_ws(int dtype, int ks, int npairs, const void* const* x, const void* const* dy, int N, int H, int W,
                                         int64_t x_ps, int Cin, int64_t dy_ps, int Cout, float* dW, int I_total, int o0, int i0,
                                         float* db, float scale, void* ws, int64_t ws_bytes, void* stream) {
  if (ws && dtype == VMG_BF16 && ks == 1 && npairs >= 1 && npairs <= WG_MAX_PAIRS && x && dy && dW && (x_ps % 8 == 0) && (dy_ps % 8 == 0) &&
      (Cin % 8 == 0) && (Cout % 8 == 0) && x_ps >= Cin && dy_ps >= Cout && N > 0 && H > 0 && W > 0 && i0 >= 0 && i0 + Cin <= I_total && o0 >= 0) {
    bool al = true;
    for (int p = 0; al && p < npairs; ++p) al = x[p] && dy[p] && ((uintptr_t)x[p] % 16 == 0) && ((uintptr_t)dy[p] % 16 == 0);
    if (al) {
      const int rc = launch_lgrad2(1, npairs, x, dy, (long long)N * H * W, x_ps, Cin, dy_ps, Cout, &dW, I_total, o0, i0, &db, &scale, ws, ws_bytes, stream);
      if (rc <= 0) return rc;
    }
  }
  // 7x7 with <= 64 output channels and 3x3 with <= 16: one wave per tap row (conv_wgrad7_kernel)
  const int rows_u = ks == 7 ? W7Geo<7>::R : W7Geo<3>::R;
  if (ws && dtype == VMG_BF16 && ((ks == 7 && Cout <= 64) || (ks == 3 && Cout <= 16)) && npairs >= 1 && npairs <= WG_MAX_PAIRS && x && dy && dW &&
      (x_ps % 8 == 0) && (Cin % 8 == 0) && x_ps >= Cin && dy_ps >= Cout && N > 0 && H > 0 && W > 0 && i0 >= 0 && i0 + Cin <= I_total && o0 >= 0 &&
      (long long)npairs * N * cdiv(H, rows_u) * cdiv(W, 32) < (1LL << 30)) {
    bool al = true, vdy = (dy_ps % 8 == 0);  // (channels past Cout inside the last vector are computed and dropped)
    for (int p = 0; al && p < npairs; ++p) {
      al = x[p] && dy[p] && ((uintptr_t)x[p] % 16 == 0) && ((uintptr_t)dy[p] % 2 == 0);
      vdy = vdy && ((uintptr_t)dy[p] % 16 == 0);
    }
    if (al) {
      Wgrad7K k;
      memset(&k, 0, sizeof(k));
      for (int p = 0; p < npairs; ++p) { k.x[p] = (const char*)x[p]; k.dy[p] = (const char*)dy[p]; }
      k.npairs = npairs; k.x_ps = x_ps; k.dy_ps = dy_ps; k.Cin = Cin; k.Cout = Cout; k.vec_dy = vdy ? 1 : 0; k.slab = (float*)ws;
      k.N = N; k.H = H; k.W = W; k.SEG = cdiv(W, 32); k.HB = cdiv(H, rows_u);
      k.Upair = N * k.HB * k.SEG; k.U = k.Upair * npairs;
      k.has_bias = db != nullptr;
      hipStream_t st = (hipStream_t)stream;
      const int ct = cdiv(Cout, 16);
      const int rc = ks == 3   ? launch_wgrad7<1, 3>(k, dW, I_total, o0, i0, db, scale, ws_bytes, st)
                     : ct == 1 ? launch_wgrad7<1>(k, dW, I_total, o0, i0, db, scale, ws_bytes, st)
                     : ct == 2 ? launch_wgrad7<2>(k, dW, I_total, o0, i0, db, scale, ws_bytes, st)
                               : launch_wgrad7<4>(k, dW, I_total, o0, i0, db, scale, ws_bytes, st);
      if (rc <= 0) return rc;
    }
  }
  // the large-tile path needs bf16, 3x3, 16-byte aligned 8-channel vectors; anything else takes the v1 kernel
  // (a channel count that is not a multiple of 8 is fine when the pixel stride has room for the whole last vector -- the
  // zero-padded input of the 3-channel stem conv, a slice of a wider tensor: the extra channels are computed and dropped)
  const int cin8 = (Cin + 7) & ~7, cout8 = (Cout + 7) & ~7;
  bool ok = ws && dtype == VMG_BF16 && ks == 3 && npairs >= 1 && npairs <= WG_MAX_PAIRS && x && dy && (x_ps % 8 == 0) && (dy_ps % 8 == 0) &&
            x_ps >= cin8 && dy_ps >= cout8 && Cout > 16;  // (Cout <= 16: the 144-channel tile would be 9/10 padding; the 16-channel tile of the v1 kernel)
  for (int p = 0; ok && p < npairs; ++p) ok = x[p] && dy[p] && ((uintptr_t)x[p] % 16 == 0) && ((uintptr_t)dy[p] % 16 == 0);
  if (!ok) return wgrad_impl(dtype, ks, npairs, x, dy, N, H, W, x_ps, Cin, dy_ps, Cout, dW, I_total, o0, i0, db, scale, stream, (float*)ws, ws_bytes);
  VMG_CHECK(N > 0 && H > 0 && W > 0 && dW && x_ps >= Cin && dy_ps >= Cout && i0 >= 0 && i0 + Cin <= I_total && o0 >= 0, "conv_wgrad: bad arguments");
  const int rc = launch_wgrad3(1, npairs, x, dy, N, H, W, x_ps, Cin, dy_ps, Cout, &dW, I_total, o0, i0, &db, &scale, ws, ws_bytes, stream);
  if (rc == 1) return wgrad_impl(dtype, ks, npairs, x, dy, N, H, W, x_ps, Cin, dy_ps, Cout, dW, I_total, o0, i0, db, scale, stream);
  return rc;
}

// Several weight gradients of ONE shape (the 30 equal convs of a recurrent residual chain all complete at the same moment of the
// backward pass) in one launch: x / dy hold nprob * npairs pointers [problem][pair], dW / db one pointer per problem.  With one problem per
// launch every conv cuts its pixels into ~85 K slabs to fill the chip and writes 85 x 864 KB of partial sums that a second kernel reads
// back (40 us of a 100 us gradient); with 8 problems per launch a problem needs only ~10 slabs.
extern "C" int vmg_conv_wgrad3_multi(int nprob, int npairs, const void* const* x, const void* const* dy, int N, int H, int W, int64_t x_ps, int Cin,
                                     int64_t dy_ps, int Cout, float* const* dW, int I_total, int o0, int i0, float* const* db, const float* scales,
                                     void* ws, int64_t ws_bytes, void* stream) {
  VMG_CHECK(nprob >= 1 && nprob <= W3_MAX_PROBS && npairs >= 1 && npairs <= WG_MAX_PAIRS && x && dy && dW && ws && scales, "conv_wgrad3_multi: 1..%d problems, 1..%d pairs",
            W3_MAX_PROBS, WG_MAX_PAIRS);
  const int cin8 = (Cin + 7) & ~7, cout8 = (Cout + 7) & ~7;
  VMG_CHECK(N > 0 && H > 0 && W > 0 && (x_ps % 8 == 0) && (dy_ps % 8 == 0) && x_ps >= cin8 && dy_ps >= cout8 && i0 >= 0 && i0 + Cin <= I_total && o0 >= 0,
            "conv_wgrad3_multi: bf16 3x3 with 8-channel vectors only");
  for (int p = 0; p < nprob * npairs; ++p)
    VMG_CHECK(x[p] && dy[p] && ((uintptr_t)x[p] % 16 == 0) && ((uintptr_t)dy[p] % 16 == 0), "conv_wgrad3_multi: null or unaligned pointer %d", p);
  for (int p = 0; p < nprob; ++p) VMG_CHECK(dW[p], "conv_wgrad3_multi: null gradient pointer %d", p);
  const int rc = launch_wgrad3(nprob, npairs, x, dy, N, H, W, x_ps, Cin, dy_ps, Cout, dW, I_total, o0, i0, db, scales, ws, ws_bytes, stream);
  VMG_CHECK(rc != 1, "conv_wgrad3_multi: the workspace is too small");
  return rc;
}
