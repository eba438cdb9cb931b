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
__device__ __attribute__((aligned(16))) unsigned int g_mf_zero[4] = {0, 0, 0, 0};  // (not const: a constant-address-space pointer in the select turns the loads into flat_load)

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
template <int NK, int NCT, bool EVEN>
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
  char* xblk = smem + NST * a.ss + COB * 4 + wave * 2 * blkb;  // [16 pixels][Cp] bf16
  char* oblk = xblk + blkb;
  // weights and bias -> LDS, once
  for (int i = tid * 16; i < NST * a.ss; i += MF_WAVES * 64 * 16) *reinterpret_cast<uint4*>(wl + i) = *reinterpret_cast<const uint4*>(a.wpack + i);
  for (int i = tid; i < COB; i += MF_WAVES * 64) lbias[i] = (a.bias && i < a.Cp) ? a.bias[i] : 0.f;
  __syncthreads();

  const int tok = lane & 15, kq = lane >> 4;
  const int ch = a.chunk, S = a.S;
  const int grp = tok / ch, kk = tok - grp * ch;  // this lane's token: group `grp` of the tile, channel chunk kk
  // per k-step: (position, channel-in-chunk) of the first of the lane's 8 features
  int p0[NK], s0[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    const int f0 = 32 * ks + 8 * kq;
    p0[ks] = f0 / S;
    s0[ks] = f0 - p0[ks] * S;
  }
  const int vpp = a.C >> 3;      // 16-byte vectors per pixel (C % 8 == 0)
  const int G = 16 / ch;         // groups per tile (1 or 2: the host admits chunk 8 and 16 here)
  const int lines_len = a.axis == 0 ? a.H : a.W, lines = a.axis == 0 ? a.W : a.H;
  const int pos_stride = a.axis == 0 ? a.W : 1;  // pixels between consecutive positions of a group
  // per-lane constants of a group's copy: vector L = 64 k + lane of the [chunk pixels][vpp vectors] list (the same for every tile)
  int lp[MF_MAXK], lv[MF_MAXK];
#pragma unroll
  for (int k = 0; k < MF_MAXK; ++k) {
    const int L = 64 * k + lane;
    const int p = (int)(((float)L + 0.5f) * (1.0f / (float)vpp));  // exact: L < 2^16
    lp[k] = L < ch * vpp ? p : -1;
    lv[k] = L - p * vpp;
  }
  const bf16* zsrc = reinterpret_cast<const bf16*>(g_mf_zero);

  for (long long tile = (long long)blockIdx.x * MF_WAVES + wave; tile < a.ntiles; tile += (long long)gridDim.x * MF_WAVES) {
    // ---- 1. pixel block -> LDS (zero-filled padding).  Group coordinates are wave-uniform; every load is UNCONDITIONAL (a lane outside
    // the image reads the zero vector) and all of a tile's loads are issued before the first one is used: loads behind a per-lane branch
    // are each followed by s_waitcnt vmcnt(0), which made a tile cost ten serialised memory latencies (46 us per launch at N = 114 688).
    long long gbase[2];  // first pixel of the group, or -1
    int gpos0[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const unsigned gg = (unsigned)(tile * G + g);  // (the host checks ngroups < 2^31)
      const unsigned gi = gg % (unsigned)a.gpl, r = gg / (unsigned)a.gpl;
      const unsigned line = r % (unsigned)lines, bt = r / (unsigned)lines;
      gpos0[g] = (int)gi * ch;
      const long long first = a.axis == 0 ? ((long long)bt * a.H + gpos0[g]) * a.W + line : ((long long)bt * a.H + line) * a.W + gpos0[g];
      gbase[g] = (g < G && tile * G + g < a.ngroups) ? first : -1;
    }
    mf_u32x4 rx[2][MF_MAXK], rm[2][MF_MAXK];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int k = 0; k < MF_MAXK; ++k) {
        const bool ok = lp[k] >= 0 && gbase[g] >= 0 && gpos0[g] + lp[k] < lines_len;
        const long long off = (gbase[g] + (long long)lp[k] * pos_stride) * a.C + lv[k] * 8;
        rx[g][k] = *reinterpret_cast<const mf_u32x4*>(ok ? a.x + off : zsrc);
        if (a.mask) rm[g][k] = *reinterpret_cast<const mf_u32x4*>(ok ? a.mask + off : zsrc);  // (wave-uniform branch)
      }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int k = 0; k < MF_MAXK; ++k) {
        mf_u32x4 val = rx[g][k];
        if (a.mask) {
          bf16x8 e = __builtin_bit_cast(bf16x8, val);
          const bf16x8 m = __builtin_bit_cast(bf16x8, rm[g][k]);
#pragma unroll
          for (int j = 0; j < 8; ++j) e[j] = (float)m[j] > 0.f ? (bf16)((float)e[j] * a.in_scale) : (bf16)0.f;
          val = __builtin_bit_cast(mf_u32x4, e);
        }
        if (lp[k] >= 0 && g < G) *reinterpret_cast<mf_u32x4*>(xblk + (g * ch + lp[k]) * rowb + lv[k] * 16) = val;
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
      int p = p0[ks], s = s0[ks];
      if constexpr (EVEN) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
        u32x4_t tw;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool in = p < ch;  // features >= Cp multiply zero weights
          tw[j] = in ? *reinterpret_cast<const unsigned int*>(xblk + (grp * ch + p) * rowb + (kk * S + s) * 2) : 0u;
          s += 2;
          if (s == S) { s = 0; ++p; }
        }
        tf = __builtin_bit_cast(bf16x8, tw);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool in = p < ch;  // features >= Cp multiply zero weights
          tf[j] = in ? *reinterpret_cast<const bf16*>(xblk + (grp * ch + p) * rowb + (kk * S + s) * 2) : (bf16)0.f;
          if (++s == S) { s = 0; ++p; }
        }
      }
      // the lane's fragment = features 32 ks + 8 kq .. + 7 of token (tile, tok): one 16-byte vector of the token matrix
      if (a.tok_out && 32 * ks + 8 * kq < a.Cp) *reinterpret_cast<bf16x8*>(a.tok_out + (tile * 16 + tok) * a.Cp + 32 * ks + 8 * kq) = tf;
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
      const int f0 = ct * 16 + kq * 4;
      int p = f0 / S, s = f0 - p * S;
      if constexpr (EVEN) {
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          if (f0 + r < a.Cp) {  // (Cp is even: the pair is inside or outside together)
            float v0 = acc[ct][r] + lbias[f0 + r], v1 = acc[ct][r + 1] + lbias[f0 + r + 1];
            if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
            const bf16x2_t pr = {(bf16)(v0 * a.out_scale), (bf16)(v1 * a.out_scale)};
            *reinterpret_cast<bf16x2_t*>(oblk + (grp * ch + p) * rowb + (kk * S + s) * 2) = pr;
          }
          s += 2;
          if (s == S) { s = 0; ++p; }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (f0 + r < a.Cp) {
            float v = acc[ct][r] + lbias[f0 + r];
            if (a.relu) v = fmaxf(v, 0.f);
            *reinterpret_cast<bf16*>(oblk + (grp * ch + p) * rowb + (kk * S + s) * 2) = (bf16)(v * a.out_scale);
          }
          if (++s == S) { s = 0; ++p; }
        }
      }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int k = 0; k < MF_MAXK; ++k) {
        if (lp[k] < 0 || gbase[g] < 0 || gpos0[g] + lp[k] >= lines_len) continue;
        const long long off = (gbase[g] + (long long)lp[k] * pos_stride) * a.C + lv[k] * 8;
        *reinterpret_cast<mf_u32x4*>(a.out + off) = *reinterpret_cast<const mf_u32x4*>(oblk + (g * ch + lp[k]) * rowb + lv[k] * 16);
      }
    }
  }
}

}  // namespace

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
  const int lds = ((nk + 1) / 2) * k.ss + nct * 16 * 4 + MF_WAVES * 2 * blkb;
  VMG_CHECK(lds <= 160 * 1024, "morphfc: Cp = %d needs %d B of LDS (> 160 KiB): use the unfused path", Cp, lds);
  hipStream_t st = (hipStream_t)stream;
  const int ncu = vmg_cu_count(vmg_current_device());
  long long nwg = cdiv64(k.ntiles, MF_WAVES);
  if (nwg > ncu) nwg = ncu;  // one workgroup per CU (the LDS holds the weights): tiles are strided over the waves
#define MF_CASE(NK_, NCT_) MF_CASE2(NK_, NCT_, true) MF_CASE2(NK_, NCT_, false)
#define MF_CASE2(NK_, NCT_, EV_)                                                                                    \
  if (nk == NK_ && nct == NCT_ && (k.S % 2 == 0) == EV_) {                                                          \
    auto fn = morph_linear_kernel<NK_, NCT_, EV_>;                                                                  \
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
