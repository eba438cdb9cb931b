/*
 * vmg_hip.h -- C-ABI of libvmg_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the VMG
 * per-frame forward/backward hot path.
 *
 * The reference (EasyVision-Ton/VMG @ 2024_10_08) has no native code: every op below replaces a torch-op
 * call site of the reference's Python modules (file:line cited per entry, relative to the reference root).
 * The host side (the vmg_amd python package, an nn.Module mirror of models/vmg.py) binds these with ctypes; INTEGRATION.md
 * shows the stub.
 *
 * Conventions
 *   - every entry returns 0 on success, <0 on error; vmg_last_error() gives the text (thread-local).
 *   - all tensor arguments are raw DEVICE pointers; features are channels-last: (N, H, W, C), C fastest.
 *   - `dtype`: VMG_F32 (0) or VMG_BF16 (1) = storage type of activations and packed weights; accumulation,
 *     bias, statistics and gradients of parameters are always fp32.
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, never synchronised; no allocation
 *     happens inside (packed weights / workspaces are caller-provided), so calls are hipGraph-capturable.
 *   - strides named *_ps are PIXEL strides in elements (>= channel count), which lets a call read or write a
 *     channel slice of a wider tensor (virtual concat / split without copies).
 */
#ifndef VMG_HIP_H
#define VMG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VMG_F32 0
#define VMG_BF16 1

/* activation codes of the conv / linear epilogue */
#define VMG_ACT_NONE 0
#define VMG_ACT_RELU 1
#define VMG_ACT_LRELU 2 /* slope argument */
#define VMG_ACT_GELU 3  /* erf form, torch nn.GELU() default */

const char* vmg_last_error(void);
int vmg_version(void);

/* ------------------------------------------------------------------------------------------------
 * Per-device context.  The library keeps no process-global mutable state besides one handle per device:
 * the handle owns the live profiler below and the device facts the launchers cache.  vmg_create(device) returns the
 * device's handle (creating it on first use, reference-counted), vmg_destroy releases it.  A handle is used by one host
 * thread at a time.  The kernel entry points themselves are stateless: they take raw pointers and a stream and use the
 * CURRENT device (hipGetDevice) -- all pointers of a call must belong to it.
 * ---------------------------------------------------------------------------------------------- */
typedef struct vmg_ctx vmg_ctx;
vmg_ctx* vmg_create(int device);   /* NULL on error (vmg_last_error) */
int vmg_destroy(vmg_ctx* ctx);
int vmg_ctx_device(const vmg_ctx* ctx);
/* number of bytes of dynamic LDS the largest kernel instance requests (diagnostics) */
int vmg_max_lds_bytes(void);

/* ------------------------------------------------------------------------------------------------
 * Weight packing for the implicit-GEMM convolution.
 *
 * w      : fp32 weights in checkpoint layout OIHW (O, I, KS, KS); nn.Linear (O, I) is KS = 1.
 * Forward packing (transpose_flip = 0): output channels = O[o0 : o0+on); the K dimension is the list of
 *   `nsrc` input-channel slices [src_off[s], src_off[s]+src_ch[s]) of I, in that order, each slice a
 *   multiple of 8 channels (one slice per tensor of a virtual channel concat).
 * Data-gradient packing (transpose_flip = 1): output channels = I[o0 : o0+on), K = O[src_off[0] : +src_ch[0])
 *   with taps mirrored -- conv(dY, pack) is then dX of the forward conv (stride 1, pad KS/2).
 * The packed image is [cout_block][stage][k-step][4 K-chunks][co in block][8] elements of `dtype` (a stage = 3 taps of a
 * 32-channel block for 3x3, two 32-channel blocks for 1x1; stages padded to 4 KiB); its size in bytes is returned by
 * vmg_conv_pack_bytes.  `cout_tiles` (1, 3, 4, 5, 7, 8 or 9) is the number of 16-channel tiles per block and must be the
 * value later passed to vmg_conv_fwd.
 * ---------------------------------------------------------------------------------------------- */
int64_t vmg_conv_pack_bytes(int dtype, int ks, int on, int nsrc, const int* src_ch, int cout_tiles);
int vmg_conv_pack(int dtype, const float* w, int O, int I, int ks, int o0, int on, int nsrc, const int* src_off,
                  const int* src_ch, int transpose_flip, int cout_tiles, void* packed, void* stream);

/* Packing for the weight-streaming 3x3 kernel (vmg_conv_fwd with deep = 3; bf16 only): same slicing arguments as
 * vmg_conv_pack.  Image: [cout block][stage][3 k-steps][4 lane groups][co in block][8] bf16, a k-step = one tap of a 32-channel
 * block, or -- for the last 16 channels of a block that is not a multiple of 32 -- two taps of 16 channels; cout_tiles 9 (blocks
 * of 144 output channels) or 7 (112).  Channel slices must split into blocks of <= 160 channels that are multiples of 16. */
int64_t vmg_convws_pack_bytes(int on, int nsrc, const int* src_ch, int cout_tiles);
int vmg_convws_pack(const float* w, int O, int I, int o0, int on, int nsrc, const int* src_off, const int* src_ch, int transpose_flip,
                    int cout_tiles, void* packed, void* stream);

/* ------------------------------------------------------------------------------------------------
 * vmg_conv_fwd -- stride-1 "same" convolution (KS = 3 or 1) as MFMA implicit GEMM with fused epilogue.
 *
 * Replaces: nn.Conv2d 3x3 call sites models/trajectory.py:32,185-186 (recurrent residual chains),
 *   models/function.py:567 (RCAB), :57 (Mlp_cnn.fc1), :1314 (local_cnn), models/vmg.py:378-382 (head),
 *   and with KS = 1 every nn.Linear / 1x1 conv on the path (function.py:64,631-636,647; trajectory.py:271,517;
 *   layers.py:769; swin_3d.py:143-149).  With a data-gradient pack it is also the backward-data pass.
 *
 *   v   = sum_s conv(src[s]) + bias                       (fp32 accumulate)
 *   pre = v                                              -> out_pre if non-null
 *   v   = act(v) * alpha
 *   v  *= act'(aux)   if actgrad != 0  (1: aux > 0 ? 1 : 0; 2: aux > 0 ? 1 : slope; 3: gelu'(aux))
 *   v  += res         if res != null
 *   out = v           (pixel_shuffle = 1: stored to (N, 2H, 2W, Cout/4) in torch PixelShuffle(2) order)
 *
 * For KS = 1 the tensor is treated as (M = N*H*W, C) rows; H, W are only used for pixel_shuffle.
 * ---------------------------------------------------------------------------------------------- */
typedef struct vmg_conv_desc {
  int dtype, ks, cout_tiles; /* cout_tiles as used for packing */
  int N, H, W, Cout;
  int nsrc;
  const void* src[4];
  int64_t src_ps[4];
  int src_ch[4];
  const void* packed;
  const float* bias; /* may be null */
  void* out;
  int64_t out_ps;
  void* out_pre; /* may be null; same layout as out */
  const void* res;
  int64_t res_ps; /* may be null */
  const void* aux;
  int64_t aux_ps; /* may be null */
  int act;
  float slope, alpha;
  int actgrad;
  int pixel_shuffle;
  int mt;   /* 0 = auto; 16-pixel tiles per wave (1 or 2) */
  int deep; /* kernel variant: 0 = 2-slot weight ring, 1 = 3-slot counted-wait ring, 2 = K split over the 4 waves with
              * weights read global->register (bf16, cout_tiles <= 5, mt 1), 3 = weight-streaming 3x3 (bf16; 128-pixel tiles x 144 or
              * 112 output channels per workgroup, loader waves stream the vmg_convws_pack image through an LDS ring; cout_tiles 9 or 7,
              * 16-byte aligned rows, no pixel_shuffle), 4 = 1x1 with every wave its own pipeline: weights
              * resident in registers, 16-row tiles through wave-private LDS (bf16, one dense source of <= 160 channels,
              * cout_tiles 3 or 5, 16-byte aligned output rows), 6 = weights-stationary 3x3 (bf16, one source of <= 64 channels,
              * cout_tiles 1, 3 or 4: a persistent workgroup per CU keeps the packed weights in LDS and walks over 128-pixel tiles;
              * the HR head, models/vmg.py:629-632).  4 and 6 are hints: a call they do not cover runs on the general kernel */
} vmg_conv_desc;

int vmg_conv_fwd(const vmg_conv_desc* d, void* stream);
/* ------------------------------------------------------------------------------------------------
 * vmg_resblock_chain_fwd / _bwd -- ResidualBlocksWithInputConv (reference: models/trajectory.py:16-52, 165-221), the 1 + 2*nblk
 * convolutions of one recurrence step, enqueued by ONE call (no host work between the launches):
 *     y_0 = lrelu_slope0(conv0(cat(src)) + b0);   t_k = relu(conv1_k(y_k) + b1_k);   y_{k+1} = y_k + r * (conv2_k(t_k) + b2_k)
 * A persistent single-launch form is not used: every layer boundary is an all-neighbour exchange (3x3 halo), and on MI355X a
 * grid-wide seam inside a launch costs more than the kernel boundary it replaces.
 * All tensors (N, H, W, C) channels-last, contiguous, dtype bf16 or fp32; packed weights from vmg_conv_pack / vmg_convws_pack with
 * cout_tiles / deep as for vmg_conv_fwd (conv0 may use its own: cout_tiles0 / deep0).  y[0..nblk] and t[0..nblk-1] are outputs (kept for
 * the backward).  Backward: g_y[nblk] = gradient of y_nblk on entry;  g_t[k] = r * dgrad2_k(g_y[k+1]) * relu'(t_k);
 * g_y[k] = g_y[k+1] + dgrad1_k(g_t[k]);  all g_y / g_t are outputs (the weight gradients take them as operands, vmg_conv_wgrad*).
 * ---------------------------------------------------------------------------------------------- */
typedef struct vmg_chain_desc {
  int dtype, N, H, W, C, nblk, nsrc;
  const void* src[4];
  int64_t src_ps[4];
  int src_ch[4];
  const void* packed0; /* forward: conv0's pack; backward: unused */
  const float* bias0;
  float slope0;
  int cout_tiles0, deep0;
  const void* const* packed1; /* nblk packs of conv1 (forward packs in _fwd, data-gradient packs in _bwd) */
  const float* const* bias1;
  const void* const* packed2;
  const float* const* bias2;
  float r_scaling;
  int cout_tiles, deep;
  void* const* y; /* nblk + 1 tensors */
  void* const* t; /* nblk tensors */
  void* const* g_y; /* backward only: nblk + 1 tensors, g_y[nblk] given */
  void* const* g_t; /* backward only: nblk tensors */
} vmg_chain_desc;
int vmg_resblock_chain_fwd(const vmg_chain_desc* d, void* stream);
int vmg_resblock_chain_bwd(const vmg_chain_desc* d, void* stream);

/* ---- replaying a captured step without the hipGraph executor (vmg_amd.train.TrainStep.capture).  hipGraphLaunch on ROCm 7 costs per node
 * what the eager Python step costs; a one-stream capture is a linear list of kernel / memcpy / memset nodes, which vmg_replay_build reads
 * out of the hipGraph_t (it must stay alive) and vmg_replay_run re-issues, op first .. last-1, with plain launches on `stream`.
 * vmg_replay_kernel_info lets the caller find marker kernels (segment boundaries of a data-parallel step). */
void* vmg_replay_build(void* hip_graph, int* n_ops, int* n_kernels);
int vmg_replay_kernel_info(void* replay, int idx, void** func, unsigned* grid_x, unsigned* block_x, void** first_arg);
int vmg_replay_run(void* replay, int first, int last, void* stream);
void vmg_replay_destroy(void* replay);

/* ---- batched packing: after an optimizer step every weight needs its packs rebuilt -- ~390 launches of 4 us in VMG-REDS-few_levels.
 * A PLAN is an array of vmg_pack_entry_bytes()-sized opaque entries in DEVICE memory, each holding the arguments of one
 * vmg_conv_pack (kind 0) / vmg_convws_pack (kind 1) call.  vmg_pack_entry fills one entry in HOST memory (the caller copies the array to
 * the device once; it stays valid while the weight and pack buffers stay where they are) and returns the number of 256-thread blocks the
 * entry wants (> 0; negative: error); blk0 = sum of the block counts of the entries before it.  vmg_pack_run repacks all of them in ONE
 * launch of total_blocks blocks; BEHIND the n entries plan_dev holds total_blocks int32: the index of the entry that owns each block (entry i owns blocks
 * [blk0_i, blk0_i + its count)). */
int vmg_pack_entry_bytes(void);
int vmg_pack_entry(void* entry, int kind, int dtype, const float* w, int O, int I, int ks, int o0, int on, int nsrc, const int* src_off,
                   const int* src_ch, int transpose_flip, int cout_tiles, void* packed, int blk0);
int vmg_pack_run(const void* plan_dev, int n, int total_blocks, void* stream);

/* diagnostics: when buf is non-null the k-split variant writes 8 wave-level 100-MHz time stamps per wave
 * (uint64[workgroups][4][8]) at its phase boundaries; null switches it off */
int vmg_conv_debug_stamps(void* buf);

/* ------------------------------------------------------------------------------------------------
 * vmg_conv_wgrad -- weight (and bias) gradient of the same convolution, accumulated into fp32 OIHW.
 *   dW[o][i][ky][kx] += scale * sum_{n,y,x} dY[n,y,x,o] * X[n,y+ky-KS/2,x+kx-KS/2,i]
 *   db[o]            += scale * sum dY[n,y,x,o]                      (if db != null)
 * dW addresses the (O_total, I_total, KS, KS) gradient of the full parameter; the call covers output channels
 * [o0, o0+Cout) and input channels [i0, i0+Cin).  fp32 atomics: call-to-call results are summed in
 * arrival order (see DESIGN.md "determinism").
 * ---------------------------------------------------------------------------------------------- */
int vmg_conv_wgrad(int dtype, int ks, int N, int H, int W, const void* x, int64_t x_ps, int Cin, const void* dy,
                   int64_t dy_ps, int Cout, float* dW, int I_total, int o0, int i0, float* db, float scale,
                   void* stream);
/* The same, summed over `npairs` (1..16) pairs (x[p], dy[p]) of identical shape and strides in ONE launch: the uses of a
 * weight that the recurrence shares over frames and directions (models/trajectory.py:361,448 call one module 2T times).
 * x and dy are HOST arrays of device pointers. */
int vmg_conv_wgrad_batched(int dtype, int ks, int npairs, const void* const* x, const void* const* dy, int N, int H, int W,
                           int64_t x_ps, int Cin, int64_t dy_ps, int Cout, float* dW, int I_total, int o0, int i0, float* db,
                           float scale, void* stream);

/* The same with a caller-provided workspace (vmg_conv_wgrad_ws_bytes() bytes, reusable across calls on one stream): bf16 3x3
 * gradients then take the large-tile kernel (144 x 48 x 9 outputs per workgroup, partial slabs + ordered reduction: bitwise
 * reproducible, no atomics); every other case falls through to vmg_conv_wgrad_batched. */
int64_t vmg_conv_wgrad_ws_bytes(void);
int vmg_conv_wgrad_batched_ws(int dtype, int ks, int npairs, const void* const* x, const void* const* dy, int N, int H, int W,
                              int64_t x_ps, int Cin, int64_t dy_ps, int Cout, float* dW, int I_total, int o0, int i0, float* db,
                              float scale, void* ws, int64_t ws_bytes, void* stream);
/* Several weight gradients of ONE shape in one launch (bf16, 3x3, 8-channel vectors, 16-byte aligned operands): x / dy hold
 * nprob * npairs pointers [problem][pair] (nprob <= 8, npairs <= 16), dW / db one pointer and scales one factor per problem (db or db[i]
 * may be null): dW_i += scales[i] * sum over pairs.
 * The 30 equal convolutions of a recurrent residual chain (reference: models/trajectory.py:16-52) complete at the same moment of the
 * backward pass; served one per launch each cuts its pixels into ~85 K slabs to fill the chip and moves 2 x 73 MB of partial sums,
 * served eight per launch a problem needs ~10 slabs. */
int vmg_conv_wgrad3_multi(int nprob, int npairs, const void* const* x, const void* const* dy, int N, int H, int W, int64_t x_ps, int Cin,
                          int64_t dy_ps, int Cout, float* const* dW, int I_total, int o0, int i0, float* const* db, const float* scales, void* ws,
                          int64_t ws_bytes, void* stream);
/* Tuning knob (A/B measurements, tests): which kernel serves the batched 3x3 weight gradients -- 1 (default) conv_wgrad3b_kernel, whose tile
 * copies are buffer loads with constant per-lane offsets issued between the MFMA columns; 0 conv_wgrad3_kernel (round 2: per-lane 64-bit pointers,
 * the copies as one block per unit).  Same tiles, same slab order, same bits.  Returns the previous value; any other argument only queries. */
int vmg_conv_wgrad3_variant(int variant);
/* The same for 1x1 convolutions / Linears (bf16; Cin, Cout and the pixel strides multiples of 8; at least 2 048 pixels per problem): x / dy
 * hold nprob * npairs pointers over M pixels each, dW (O_total, I_total). */
int vmg_linear_wgrad2_multi(int nprob, int npairs, const void* const* x, const void* const* dy, int64_t M, int64_t x_ps, int Cin, int64_t dy_ps,
                            int Cout, float* const* dW, int I_total, int o0, int i0, float* const* db, const float* scales, void* ws,
                            int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise / normalisation kernels (HBM-bound, one pass, 16-byte vectors).
 *
 * vmg_act_bwd        out = dy * alpha * act'(ref)   ref = output for RELU/LRELU, pre-activation for GELU.
 *                    Backward of the activations fused in the conv epilogue (function.py:72,632; trajectory.py:33,188).
 * vmg_pixel_shuffle  to_depth = 0: (N,H,W,4c) -> (N,2H,2W,c) in torch nn.PixelShuffle(2) channel order
 *                    (models/vmg.py:380,629-630); to_depth = 1: the inverse (its backward).
 * vmg_pixel_unshuffle_actgrad  the backward of "conv -> PixelShuffle(2) -> activation" in front of the conv's gradients, one pass:
 *                    out[n,y,x,4c+2i+j] = dy[n,2y+i,2x+j,c] * alpha * act'(ref[n,2y+i,2x+j,c]); dy / ref (N,2H,2W,c), out (N,H,W,4c);
 *                    ref = null with VMG_ACT_NONE.  c a multiple of the 16-byte vector (8 bf16 / 4 fp32).  models/vmg.py:629-630.
 * vmg_frame_gather   dst frame f = sum_{k < nsrc} src frame idx[f*nsrc + k] (nsrc 1 or 2; idx: device int32, < 0 = no term; fp32 sum, one
 *                    rounding); frames = contiguous blocks of frame_elems elements (whole 16-byte vectors).  The two direction sweeps of
 *                    the recurrence (models/trajectory.py:323-392, 407-477) run as one batch: step j works on [frame t-1-j | frame j] of
 *                    every clip -- that arrangement of the (n, t) features is one gather, its gradient one gather-add.
 * vmg_sum_n          out = the sum of 2..4 tensors of one dtype (fp32 sum, one rounding): the gradient of a tensor with several consumers -- the hidden
 *                    state of the recurrence feeds the next step's warp, the attention memory and the output (models/trajectory.py:323-392).
 * vmg_cast_clear     out = (T)(acc + add) (add optional), acc = 0: the ONE rounding of a finished fp32 scatter accumulator (flow-warp backward,
 *                    models/trajectory.py:95-116; the key / value frames of the trajectory attention, :672-795) that also leaves the accumulator ready for
 *                    its next use.  n a multiple of 4.
 * vmg_pair_steps     the same pairing through a POINTER LIST of the t step tensors (2n frames each; the recurrence's step outputs / step
 *                    gradients are separate allocations): mode 0 steps -> a = backward-sweep features, b = forward-sweep features, both (n, t)
 *                    in frame order (trajectory.py:394-395, 479: `feats_.insert(0, ...)` / `append` + stack); mode 1 the inverse (its
 *                    backward); mode 2 a[i, f] = steps[t-1-f][i] + steps[f][n+i] (the gradient of the pairing, fp32 sum, one rounding).  t <= 64.
 * vmg_layernorm_fwd  y = (x - mean) * rstd * w + b over the last dim C of (M, C) rows, eps inside the sqrt; mean / rstd
 *                    (fp32, M each) are written when non-null.  nn.LayerNorm at function.py:1164,1195; layers.py:768-775;
 *                    swin_3d.py:717,741.
 * vmg_layernorm_bwd  dx, and dw += sum dy * xhat, db += sum dy (fp32 atomics).
 * ---------------------------------------------------------------------------------------------- */
int vmg_act_bwd(int dtype, const void* dy, const void* ref, void* out, int64_t n, int act, float slope, float alpha,
                void* stream);
int vmg_pixel_shuffle(int dtype, const void* in, void* out, int N, int H, int W, int c, int to_depth, void* stream);
int vmg_sum_n(int dtype, const void* const* srcs, int nsrc, void* out, int64_t n, void* stream);
int vmg_cast_clear(int dtype, float* acc, const void* add, void* out, int64_t n, void* stream);
int vmg_pair_steps(int dtype, int mode, void* const* steps, void* a, void* b, int n, int t, int64_t frame_elems, void* stream);
int vmg_frame_gather(int dtype, const void* src, void* dst, const int* idx, int64_t frame_elems, int n_src_frames, int n_dst_frames, int nsrc,
                     void* stream);
int vmg_pixel_unshuffle_actgrad(int dtype, const void* dy, const void* ref, void* out, int N, int H, int W, int c, int act, float slope,
                                float alpha, void* stream);
int vmg_layernorm_fwd(int dtype, const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, int64_t M,
                      int C, float eps, void* stream);
int vmg_layernorm_bwd(int dtype, const void* dy, const void* x, const float* mean, const float* rstd, const float* w, void* dx,
                      float* dw, float* db, int64_t M, int C, void* stream);
/* the same with dx = add + LayerNorm backward (add: contiguous (M, C) in the activation dtype, or null): the gradient that reaches the
 * normalised tensor through a skip connection (TAB: x feeds norm2 / norm3 AND the residual, models/function.py:1212-1217) is summed here
 * instead of by a separate pass */
int vmg_layernorm_bwd_add(int dtype, const void* dy, const void* x, const float* mean, const float* rstd, const float* w, const void* add,
                          void* dx, float* dw, float* db, int64_t M, int C, void* stream);
/* The same with ndy (1..5) gradients of the LayerNorm OUTPUT, summed in fp32 inside the kernel (a normalised tensor read by several consumers --
 * the MorphFC mixer reads LayerNorm(x) five times -- gets one gradient per consumer; autograd would add them pairwise, three passes per add). */
int vmg_layernorm_bwd_multi(int dtype, int ndy, const void* const* dy, const void* x, const float* mean, const float* rstd, const float* w,
                            const void* add, void* dx, float* dw, float* db, int64_t M, int C, void* stream);

/* ---- UpdownkeepSampling (reference: models/layers.py:777-798): the space<->depth rearrangement fused into the LayerNorm that follows it.
 * The LayerNorm rows are GATHERED from the feature map (forward) and their gradient is scattered back (backward); no rearranged
 * copy exists.  mode 1 = "down" ('n d c (h neih) (w neiw) -> n d h w (neiw neih c)': x is (N, 2H, 2W, cseg), rows (N*H*W, 4*cseg));
 * mode 2 = "up" ('n d (neiw neih c) h w -> n d (h neih) (w neiw) c': x is (N, H/2, W/2, 4*cseg), rows (N*H*W, cseg)).  y, dy: contiguous rows;
 * the Linear of the module is vmg_conv_fwd (KS = 1) on y.  w, b, mean, rstd, dw, db as in vmg_layernorm_fwd / _bwd. */
int vmg_space_depth_ln_fwd(int dtype, int mode, const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, int N, int H,
                           int W, int cseg, float eps, void* stream);
int vmg_space_depth_ln_bwd(int dtype, int mode, const void* dy, const void* x, const float* mean, const float* rstd, const float* w, void* dx,
                           float* dw, float* db, int N, int H, int W, int cseg, void* stream);

/* ------------------------------------------------------------------------------------------------
 * TAB token-mixer tail (models/function.py:542-558 channel attention, :791-802 branch re-weighting + tanh gate).
 * Tensors are (G, R, C) channels-last views (G groups of R rows); per-(group, channel) quantities are fp32.
 *   vmg_group_reduce     mode 0: out[g,c] = scale * sum_r (a [+ b + c3])   mode 1: out[g,c] = scale * sum_r a*b.  Two launches, NO atomics:
 *                        block partials go to the workspace ws (vmg_group_reduce_ws_bytes() bytes of device memory, reusable by the next
 *                        call on the stream) and are added in a fixed order -- the same bits on every run
 *   vmg_tab_elementwise  op 0 CA_FWD   o0 = (p0 * coef[g,c] + p1) * s
 *                        op 1 CA_BWD   o0 = p0 * s * coef[g,c] + add[g,c];  o1 = p0 * s
 *                        op 2 MIX_FWD  o0 = p0*coef[g,c,0] + p1*coef[g,c,1] + p2*coef[g,c,2]
 *                        op 3 MIX_BWD  o_k = p0 * coef[g,c,k] + add[g,c]   (k = 0,1,2)
 *                        op 4 GATE_FWD o0 = (p0 + p1) * tanh(p1)
 *                        op 5 GATE_BWD p0 = dy, p1 = x, p2 = y:  o0 = dy*tanh(y);  o1 = dy*(tanh(y) + (x+y)*(1 - tanh(y)^2))
 *                        op 7 SCALE    o0 = p0 * coef[g,c] * s   (gradient of the DropPath residual w.r.t. the dropped branch, function.py:1212-1217)
 * ---------------------------------------------------------------------------------------------- */
int64_t vmg_group_reduce_ws_bytes(void);
int vmg_group_reduce(int dtype, const void* a, const void* b, const void* c3, float* out, int G, int64_t R, int C, int mode, float scale,
                     float* ws, int64_t ws_bytes, void* stream);
/* out (G, C, 3) = scale * sum over the R rows of group g of a * {b0, b1, b2} (fp32): the three branch sums of the MorphFC re-weighting
 * backward from one pass over the gradient (reference: models/function.py:791-793 through autograd).  Same two-launch ordered scheme. */
int vmg_group_reduce3(int dtype, const void* a, const void* b0, const void* b1, const void* b2, float* out, int G, int64_t R, int C, float scale,
                      float* ws, int64_t ws_bytes, void* stream);
int vmg_tab_elementwise(int dtype, int op, const void* p0, const void* p1, const void* p2, const float* coef, const float* add, float s,
                        void* o0, void* o1, void* o2, int64_t rows, int64_t R, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Flow-guided sampling of the trajectory recurrence (models/trajectory.py:71-116 flow_warp, :329-333, :414-417).
 * flow: (N,H,W,2) fp32 pixel offsets (x then y); coordinates follow flow_warp + F.grid_sample(align_corners=True).
 *   vmg_warp_bilinear_fwd   out[n,y,x,:] = bilinear sample of x at (x + flow_x, y + flow_y), border padding.
 *   vmg_warp_bilinear_bwd   dx_acc ((N,H,W,C), ALWAYS fp32, caller-zeroed) += scatter of dy (float atomics, also for bf16 tensors -- the
 *                           caller rounds the sums to bf16 once; C even for bf16); dflow (fp32, (N,H,W,2)) = d/dflow, every element written.
 *   vmg_warp_nearest_planes advects the tracked-location maps (N,K2,H,W) fp32 with nearest sampling, border padding.
 * ---------------------------------------------------------------------------------------------- */
int vmg_warp_bilinear_fwd(int dtype, const void* x, const float* flow, void* out, int N, int H, int W, int C, void* stream);
int vmg_warp_bilinear_bwd(int dtype, const void* x, const float* flow, const void* dy, void* dx_acc, float* dflow, int N, int H,
                          int W, int C, void* stream);
int vmg_warp_nearest_planes(const float* loc, const float* flow, float* out, int N, int K2, int H, int W, void* stream);
/* vmg_flow_smooth: the flow smoothing in front of a trajectory stage (models/function.py:1466-1478: reflect-pad right / bottom to a multiple of r, r x r mean,
 * nearest x r, crop) on `planes` fp32 planes of H x W; backward = 0: out = smoothed planes, backward = 1: `in` is the output gradient, `out` the input gradient
 * (a gather: deterministic, no zero-fill).  The padding must be smaller than the plane, as F.pad(mode='reflect') demands. */
int vmg_flow_smooth(const float* in, float* out, int64_t planes, int H, int W, int r, int backward, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Trajectory window attention = LTAM_multi_head.forward_wins without the output projection
 * (models/trajectory.py:672-774; relative position term cal_pe :534-547).
 *   q (n,h,w,c); keys[j], vals[j] (n,h,w,c) for key-frame j = 0 (oldest) .. t-1; loc (n,2t,h,w) fp32 tracked pixel
 *   coordinates (x plane then y plane per key-frame); rpe (heads, wh*ww, wh*ww) fp32; decay (heads) fp32.
 *   out (n,h,w,c); lse (n,h,w,heads) fp32 log-sum-exp, kept for the backward pass.
 * Backward: dq (n,h,w,c); dk_acc[j], dv_acc[j] (n,h,w,c) caller-zeroed FP32 accumulators for every tensor dtype (gradients are
 * scattered to the gathered source pixels with float atomics; several calls may add into one accumulator; the caller rounds the
 * finished sums to the tensors' dtype once); drpe (heads, wq, wq) fp32 accumulated.  heads must be 4.
 * ---------------------------------------------------------------------------------------------- */
int vmg_ltam_fwd(int dtype, const void* q, const void* const* keys, const void* const* vals, const float* loc, const float* rpe,
                 const float* decay, void* out, float* lse, int n, int h, int w, int c, int heads, int wh, int ww, int t, float scale,
                 void* stream);
int vmg_ltam_bwd(int dtype, const void* q, const void* const* keys, const void* const* vals, const float* loc, const float* rpe,
                 const float* decay, const void* out, const float* lse, const void* dout, void* dq, void* const* dk_acc,
                 void* const* dv_acc, float* drpe, int n, int h, int w, int c, int heads, int wh, int ww, int t, float scale,
                 void* stream);

/* ------------------------------------------------------------------------------------------------
 * Live kernel timing (bench.py roofline object): HIP events are recorded on the launch stream around every
 * `stride`-th launch of kernel class `klass` (1 = bf16 conv3x3 144->144, the trajectory-chain kernel, forward
 * and data-gradient alike) until `max_samples` pairs are used.  vmg_prof_end synchronises those events and
 * returns the number of launches seen, the samples taken and their summed duration in milliseconds.
 * ---------------------------------------------------------------------------------------------- */
int vmg_prof_begin(vmg_ctx* ctx, int klass, int stride, int max_samples);
int vmg_prof_end(vmg_ctx* ctx, int64_t* launches_seen, int* samples, double* total_ms);
/* restrict the timed launches to those over exactly `pixels` pixels (N*H*W); 0 = any size */
int vmg_prof_select_pixels(vmg_ctx* ctx, int64_t pixels);
/* event-pair interval (microseconds) around an empty one-wave kernel: the dispatch + event latency that the intervals
 * above contain on top of the kernel's own duration (synchronises; call outside the timed region). */
double vmg_prof_null_interval_us(int reps, void* stream);

/* ---- MorphFC H- / W-branch (reference: Enhanced_MorphFCs_decay.forward, models/function.py:763-786), bf16 --------------------------------------
 * out = reshuffle^-1( act( reshuffle(x) W^T + b ) ) * out_scale with the token reshuffle (pad C -> Cp and the mixed axis to a multiple of `chunk`;
 * token (group, k) feature p*S + s <- position p of the group, channel k*S + s; S = Cp / chunk) folded into the GEMM's operand addressing: x, out
 * are the channels-last (BT, H, W, C) feature maps, no token tensor exists.  axis 0 mixes along H, 1 along W.  `packed`: vmg_conv_pack image of
 * the (Cp, Cp) weight (ks = 1, one source of Cp channels, cout_tiles = ceil(Cp / 16)); the caller packs AFTER applying the retention decay
 * W <- W * Gamma (function.py:766-768).  relu_mask (may be null): the data-gradient form -- x is multiplied by (relu_mask > 0) * in_scale on the way
 * in (relu_mask = the forward output, `packed` = the data-gradient pack, relu = 0).  chunk 8 or 16; Cp in {144, 112, 64, 32, 16}; other
 * shapes (e.g. Cp = 228, chunk 12 of the full configuration) take the general path: gather kernel + vmg_conv_fwd.
 * tok_out (may be null): the token matrix the GEMM multiplied, (vmg_morphfc_token_rows(...), Cp) bf16, rows = (group, k) in the reference's
 * order, after the mask / scale -- the operands of the weight gradient dW = dpre_tokens^T x_tokens (forward call: tokens of x; data-gradient
 * call: tokens of dy * relu'(y) * in_scale), written by the lanes that hold them as MFMA fragments: no gather pass for the weight gradient. */
int vmg_morphfc_fwd(int axis, int chunk, const void* x, const void* relu_mask, const void* packed, const float* bias, void* out, void* tok_out,
                    int BT, int H, int W, int C, int Cp, int cout_tiles, int relu, float in_scale, float out_scale, void* stream);
int64_t vmg_morphfc_token_rows(int axis, int chunk, int BT, int H, int W);
/* General MorphFC path (any chunk / Cp, bf16 and fp32; the full configuration's chunk 12, Cp 224 / 228 / 448 -- models/function.py:749-750,
 * 763-764, 772, 776-777, 785): the token matrix of the reference's pad + rearrange chain written by ONE gather kernel, and turned back into the
 * feature map (cropped) by ONE scatter kernel; the Linear in between is vmg_conv_fwd (ks = 1).  x / out: (BT, H, W, C) channels-last; tok:
 * (vmg_morph_token_rows(...), ld) with ld >= Cp (features [Cp, ld) are written as zeros: ld = Cp rounded up to 8 serves the convolution
 * kernel's 16-byte vectors).  Row (group, k) feature p*S + s <-> pixel (position p of the group), channel k*S + s, S = Cp / chunk; groups in the
 * order (bt, line, group along the mixed axis).  The two calls are each other's adjoint, i.e. each other's backward. */
int64_t vmg_morph_token_rows(int axis, int chunk, int BT, int H, int W);
int vmg_morph_tokens_gather(int dtype, int axis, int chunk, const void* x, void* tok, int BT, int H, int W, int C, int Cp, int ld, void* stream);
int vmg_morph_tokens_scatter(int dtype, int axis, int chunk, const void* tok, void* out, int BT, int H, int W, int C, int Cp, int ld, void* stream);

/* ---- 3-D shifted-window attention (reference: models/swin_3d.py:167-252 rWindowAttention.attention, :55-118 window partition / mask, :772-832 block) ----
 * q (B, D, H, W, C) and kv (B, D, H, W, 2C; k then v) are the outputs of the q / kv Linears on the UN-partitioned feature map; window partition
 * into (wt, 8, 8) windows, the zero padding to window multiples (a padded position holds the Linear's bias bq / bkv, as in the reference where
 * zeros are padded before the Linears; pass null for bias-free Linears), the cyclic roll by (sd, sh, sw) of shifted blocks, the -100 region mask
 * and the relative-position bias gather (table (n_rel, heads) for the (wt, 8, 8) window) are index arithmetic inside the kernel.  Every time
 * slice's queries attend to the tokens of the other slices of their window.  out (B, D, H, W, C); lse (windows, heads, wt*64) fp32 is kept for the
 * backward, which writes dq, dkv and accumulates (+=) dtable and, for gradient reaching the biases through padded positions, dbq / dbkv (may be null).
 * ws (may be null): vmg_win3d_attn_bwd_ws_bytes(...) bytes -- every (window, head) workgroup's table gradient goes there with plain stores and a
 * second launch adds the windows in a fixed order (bit-reproducible); without it the workgroups add into dtable with float atomics (1 575 per
 * workgroup onto the same addresses: the launch is then bound by them). */
int vmg_win3d_attn_fwd(int dtype, const void* q, const void* kv, const float* bq, const float* bkv, const float* table, void* out, float* lse,
                       int B, int D, int H, int W, int C, int heads, int wt, int sd, int sh, int sw, void* stream);
int64_t vmg_win3d_attn_bwd_ws_bytes(int B, int D, int H, int W, int heads, int wt);
int vmg_win3d_attn_bwd(int dtype, const void* q, const void* kv, const float* bq, const float* bkv, const float* table, const void* out,
                       const float* lse, const void* d_out, void* dq, void* dkv, float* dtable, float* dbq, float* dbkv, void* ws, int B, int D, int H, int W,
                       int C, int heads, int wt, int sd, int sh, int sw, void* stream);
/* Tuning knob (tests, A/B): 1 (default) the MFMA kernels for bf16 tensors with an even head dimension <= 32 (QK^T, PV and their gradients on
 * v_mfma_f32_16x16x32_bf16), 0 the VALU kernel (one thread per token) for everything.  fp32 tensors always take the VALU kernel.  Returns the
 * previous value; any other argument only queries. */
int vmg_win3d_variant(int variant);

/* ---- multi-scale skip (MDSC; reference: models/vmg.py:388-400, 519, 525): adaptive_max_pool2d to (H/f, W/f) as non-overlapping f x f
 * windows (f must divide H and W; f = 4 in the model).  idx (N, H/f, W/f, C) bytes: position of the winner inside its window.
 * The 1x1 conv that follows is vmg_conv_fwd (KS = 1); GroupNorm(1, C) + ReLU is two vmg_group_reduce calls (sum, sum of squares) and
 * vmg_tab_elementwise op 6 (p0 * k0 + p1 * k1 + add with per-(sample, channel) coefficients, ReLU when s > 0.5). */
int vmg_maxpool_fwd(int dtype, const void* x, void* y, unsigned char* idx, int N, int H, int W, int C, int f, void* stream);
int vmg_maxpool_bwd(int dtype, const void* dy, const unsigned char* idx, void* dx, int N, int H, int W, int C, int f, void* stream);

/* ---- squeeze-excite MLPs on pooled (G, C) fp32 vectors, one launch each way.
 * Replaces CALayer.conv_du (reference: models/function.py:542-558; act1 = VMG_ACT_RELU, mode 0: out = sigmoid(W2 act(W1 m + b1) + b2))
 * and Enhanced_MorphFCs_decay.reweight + softmax (models/function.py:791-793; act1 = VMG_ACT_GELU, mode 1: Co = 3 * channels, out =
 * softmax over each channel's three consecutive logits).  w1 (Hd, C), w2 (Co, Hd) row-major; pre (G, Hd) is the saved pre-activation.
 * _bwd takes dout = dL/d out and WRITES dm (G, C) = dm_scale * dL/dm; the parameter gradients dw1, db1, dw2, db2 are written
 * (accumulate = 0) or added to what the buffers hold (accumulate = 1: straight into param.grad); ws: G * (Co + Hd) floats of scratch. */
int vmg_se_mlp_fwd(const float* m, const float* w1, const float* b1, const float* w2, const float* b2, float* pre, float* out, int G, int C,
                   int Hd, int Co, int act1, int mode, void* stream);
int vmg_se_mlp_bwd(const float* dout, const float* out, const float* m, const float* pre, const float* w1, const float* w2, float* dm,
                   float* dw1, float* db1, float* dw2, float* db2, float* ws, int G, int C, int Hd, int Co, int act1, int mode, float dm_scale,
                   int accumulate, void* stream);

/* ---- SPyNet pyramid pieces (reference: models/vmg.py:39-123), channels-last -----------------------------------------------
 * vmg_avgpool2_nhwc: F.avg_pool2d(x, 2, 2) of (n, h, w, c) -> (n, h/2, w/2, c)  (:66-70).
 * vmg_upsample2x_ac_fwd: y (n, 2h, 2w, c) = scale * F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True), fp32 (the
 * flow between pyramid levels, scale = 2, :97-102); _bwd: dx = scale * U^T dy (dx is overwritten).
 * The 7x7 convolutions of SPyNetBasicModule (:126-173) are vmg_conv_fwd / vmg_conv_wgrad with ks = 7, the warps vmg_warp_bilinear_*. */
int vmg_avgpool2_nhwc(int dtype, const void* x, void* y, int n, int h, int w, int c, void* stream);
int vmg_upsample2x_ac_fwd(const float* x, float* y, int n, int h, int w, int c, float scale, void* stream);
int vmg_upsample2x_ac_bwd(const float* dy, float* dx, int n, int h, int w, int c, float scale, void* stream);
/* The glue of one SPyNet pyramid level (models/vmg.py:72-85) on this port's 8-channel pixels (RGB in channels 0..2):
 * vmg_spy_operand_fwd  out (npix, 8) = [ref[..., 0:3] | warped[..., 0:3] | (T) up (npix, 2) fp32]  -- `torch.cat([ref, warp(supp, flow_up), flow_up], 1)`
 * vmg_spy_operand_bwd  dwarped (npix, 8) = [dx8[..., 3:6] | 0 x 5], dup (npix, 2) fp32 = dx8[..., 6:8]
 * vmg_spy_flow_add     out fp32 = up fp32 + (float) res (n elements)                               -- `flow = flow_up + basic_module(...)` */
int vmg_spy_operand_fwd(int dtype, const void* ref, const void* warped, const float* up, void* out, int64_t npix, void* stream);
int vmg_spy_operand_bwd(int dtype, const void* dx8, void* dwarped, float* dup, int64_t npix, void* stream);
int vmg_spy_flow_add(int dtype, const float* up, const void* res, float* out, int64_t n, void* stream);
/* vmg_spy_prep: SPyNet's input normalisation (models/vmg.py:104-106): img (n, 3, h, w) fp32, mean / std (3) fp32 on the device -> out (n, h, w, 8) in `dtype`,
 * channels 0..2 = (img - mean) / std, zeros behind. */
int vmg_spy_prep(int dtype, const float* img, const float* mean, const float* stdv, void* out, int64_t n, int h, int w, void* stream);

/* ---- sliding-window inference accumulators (reference: tools/Tester.py:107-177, :249-250) --------------------------
 * vmg_tile_accumulate: for a tile `patch` (planes, ph, pw; dtype 0 = f32, 1 = bf16) placed at (oh, ow) of the fp32
 * canvases E and Wt (planes, EH, EW):  E += patch * m,  Wt += m, where m drops the first `top` / last `bottom` rows and
 * the first `left` / last `right` columns of the tile (replaces Tester.test_image's four in-place border zeroings of
 * out_patch and out_patch_mask and its two slice adds, tools/Tester.py:126-139).
 * vmg_tile_finalize: q = E / Wt (tools/Tester.py:139); out_f32 (optional) = q; out_u8 (optional) =
 * round-half-even(clamp(q, 0, 1) * 255) (tools/Tester.py:249-250). */
int vmg_tile_accumulate(int dtype, const void* patch, float* E, float* Wt, int64_t planes, int ph, int pw, int EH, int EW, int oh, int ow,
                        int top, int bottom, int left, int right, void* stream);
int vmg_tile_finalize(const float* E, const float* Wt, float* out_f32, unsigned char* out_u8, int64_t n, void* stream);

/* ---- AdamW over a flat fp32 segment (reference: torch.optim.AdamW as set up in tools/Trainer.py:86-105) -----------------
 * p, g, m, v: parameter, gradient, exp_avg, exp_avg_sq (n floats each, 16-byte aligned).  hyper (DEVICE memory, 4 floats):
 * lr, weight_decay, 1 - beta1^t, sqrt(1 - beta2^t).  torch's update order and formula, no amsgrad. */
int vmg_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float beta1, float beta2, float eps,
                   void* stream);

/* ---- fp8 (OCP e4m3) 3x3 convolution on the block-scaled matrix instruction (BASELINE configs[4] "fp8 MFMA weights"; the
 * reference computes the same convolutions in fp32: models/trajectory.py:16-52, 165-221) ------------------------------------------------
 * Q8 RECORD of a pixel with C channels: [32*ceil(C/32) bytes e4m3, zero beyond C][16 bytes: an E8M0 scale byte per 32-channel block, rest
 * zero]; value = e4m3 * 2^(scale - 127); vmg_q8_record_bytes(C) bytes per pixel (176 for C = 144, 144 for C = 112).
 *   vmg_q8_quantize     bf16 rows (M pixels, pixel stride x_ps elements) -> records (block scale: the power of two mapping the block's largest
 *                       magnitude into (224, 448]).
 *   vmg_convq8_pack     fp32 (O, I, 3, 3) -> e4m3 k-step images of the instruction's A operand + one E8M0 scale byte per output channel
 *                       (transpose_flip: the data-gradient form); vmg_convq8_pack_bytes(Cout, Cin) bytes.  144 -> 144 and 112 -> 112.
 *   vmg_convq8_fwd      out = [res +] alpha * act(conv3x3(src records) + bias), fp32 accumulation; written as bf16 rows (out, optional) and / or
 *                       as records for the next convolution (outq, optional; quantised from the bf16-rounded values when both are written). */
typedef struct vmg_convq8_desc {
  int N, H, W, Cin, Cout;
  const void* src;     /* (N,H,W) records, vmg_q8_record_bytes(Cin) bytes each */
  const void* packed;  /* vmg_convq8_pack image */
  const float* bias;   /* (Cout) or null */
  void* out;           /* bf16 (N,H,W,out_ps) or null */
  int64_t out_ps;
  void* outq;          /* records (N,H,W) x vmg_q8_record_bytes(Cout), or null */
  const void* res;     /* bf16 residual or null */
  int64_t res_ps;
  int act;             /* 0 none, 1 relu, 2 leaky relu (slope) */
  float slope, alpha;
} vmg_convq8_desc;
int vmg_q8_record_bytes(int C);
int vmg_q8_quantize(const void* x, int64_t x_ps, void* out, int64_t M, int C, void* stream);
int64_t vmg_convq8_pack_bytes(int Cout, int Cin);
int vmg_convq8_pack(const float* w, int O, int I, int transpose_flip, void* packed, void* stream);
int vmg_convq8_fwd(const vmg_convq8_desc* d, void* stream);
/* The fp8 part of ResidualBlocksWithInputConv (models/trajectory.py:16-52, 165-221) from one call: per block t_k = relu(conv1(q)),
 * y_{k+1} = y_k + r * conv2(q(t_k)).  q0: records of y[0]; y: nblk + 1 bf16 tensors (y[0] given); t: nblk bf16 tensors or null (not kept);
 * qa, qb: scratch record buffers (N*H*W * vmg_q8_record_bytes(C) bytes each; qa may equal q0). */
typedef struct vmg_chainq8_desc {
  int N, H, W, C, nblk;
  const void* q0;
  void* qa;
  void* qb;
  void* const* packed1;
  void* const* bias1;
  void* const* packed2;
  void* const* bias2;
  void* const* y;
  void* const* t;
  float r_scaling;
} vmg_chainq8_desc;
int vmg_resblock_chain_fwd_q8(const vmg_chainq8_desc* c, void* stream);

/* ---- clip_grad_norm_ over a flat fp32 gradient buffer (reference: torch.nn.utils.clip_grad_norm_(parameters, max_norm, norm_type=2) as
 * called in tools/Trainer.py:141-143, 166-167 when train.if_grad_clip is set) -------------------------------------------------
 * g: n floats, 16-byte aligned; workspace: vmg_grad_clip_ws_bytes() bytes of device memory; norm_out (DEVICE, 2 floats): the total
 * L2 norm and the applied coefficient min(1, max_norm / (norm + 1e-6)).  Two launches, fixed summation order (bit-reproducible). */
int64_t vmg_grad_clip_ws_bytes(void);
int vmg_grad_clip_norm(float* g, int64_t n, float max_norm, void* workspace, float* norm_out, void* stream);

/* ---- Charbonnier + edge loss (reference: utils/loss.py:22-79, CharbonnierLoss(eps, if_aux_loss=True, aux_ratio)) ---------
 * x, y: (planes, H, W) fp32 images, planes = B*T*3.  fwd: a1 (planes, ceil(H/2), ceil(W/2)) scratch, ld (planes, H, W) = the
 * Laplacian of x - y (kept for the backward), partial: 2 floats per block (vmg_charbonnier_edge_blocks of them): sums of
 * sqrt(d^2 + eps) and sqrt(ld^2 + eps); loss = (sum0 + aux_ratio * sum1) / (planes*H*W).
 * bwd: dx = gs1 * d / sqrt(d^2 + eps) + gs2 * lap^T(ld / sqrt(ld^2 + eps));  gs1 = dL / n, gs2 = dL * aux_ratio / n. */
int vmg_charbonnier_edge_blocks(int64_t planes, int H, int W);
int vmg_charbonnier_edge_fwd(const float* x, const float* y, float* a1, float* ld, float* partial, int64_t planes, int H, int W, float eps,
                             void* stream);
int vmg_charbonnier_edge_bwd(const float* x, const float* y, const float* ld, float* u, float* dx, int64_t planes, int H, int W, float eps,
                             float gs1, float gs2, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VMG_HIP_H */
