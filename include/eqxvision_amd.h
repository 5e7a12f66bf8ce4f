/* eqxvision_amd -- C ABI of the MI355X (gfx950) forward-pass engine.
 *
 * The reference (paganpasta/eqxvision) has NO native/FFI interface: its hot path is
 * `jax.vmap(net, axis_name="batch")(images, key=keys)` (README.md:37-40) over
 * `eqx.Module.__call__`s that bottom out in equinox/jax ops.  This header is the boundary the
 * replacement sits behind; each entry point names the reference op(s) it replaces.  The Python
 * host (`eqxvision_amd/_lib.py`) binds it with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch .data_ptr()), dense, caller-owned;
 *   - activations are NHWC ("pixel-major, channel-contiguous") or row-major [rows][features];
 *     the only NCHW tensors are user images entering `mv_conv2d_nchw_fwd` and the layout
 *     converters;
 *   - dtype codes: MV_F32 / MV_BF16.  bf16 = bf16 storage, fp32 accumulate, fp32 epilogue;
 *   - all work is enqueued on `stream` (a hipStream_t; NULL = default stream), asynchronous;
 *   - return 0 on success, negative MV_E_* on argument errors, positive = hipError_t;
 *     `mv_last_error()` gives a thread-local message.  Nothing aborts or throws across the ABI.
 */
#ifndef EQXVISION_AMD_H
#define EQXVISION_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MV_ABI_VERSION 2   /* 2 (round 6): mv_bottleneck_strip_* gone, mv_bn_train_dz_coef_f32 / mv_device_status / the layer-1 plan (rc0, rc, res, sub)
                            * chain entries and the LayerNorm-fold Linear pair (lnout / lnin) added, every scratch consumer keeps its data behind a 4096-byte sync header */

typedef void* mv_stream_t; /* hipStream_t */

enum { MV_F32 = 0, MV_BF16 = 1 };
/* Activations.  0-2 are fused into every GEMM / convolution epilogue.  3-6 (jax.nn.hard_swish / hard_sigmoid / sigmoid / silu:
 * mobilenetv3.py:57,72, efficientnet.py:117, lraspp.py:102) are implemented by mv_conv2d_nhwc_fwd (routed to the streaming kernel or
 * the one tile kernel whose epilogues have them), mv_conv2d_nchw_fwd (the image-entry kernels: the SiLU / hard-swish stems), the
 * depthwise convolution, mv_se_scale_fwd and the element-wise entries (mv_eltwise_fwd, mv_add_fwd, mv_channel_affine_fwd); every
 * other matrix-core entry refuses them with MV_E_INVALID. */
enum { MV_ACT_NONE = 0, MV_ACT_RELU = 1, MV_ACT_GELU_TANH = 2, MV_ACT_HARD_SWISH = 3, MV_ACT_HARD_SIGMOID = 4, MV_ACT_SIGMOID = 5,
       MV_ACT_SILU = 6 };
enum { MV_OK = 0, MV_E_INVALID = -1, MV_E_UNSUPPORTED = -2, MV_E_OOM = -3 };
/* mv_set_flag names (per calling thread; 0 = the tuned default).  Policy since round 4: ONE "no_<kernel>" switch per fused / special
 * kernel (in-process A/B with tools/ab_flag.py and the tests' cross-checks against the un-fused path), the forcing switches the tests
 * use to reach every tile shape, and nothing else -- switches whose A/B is settled are deleted together with the losing branch.
 *   cross-check     "force_generic" (every op on the simple VALU kernels)
 *   GEMM core       "igemm8" (2 / 3 / 4 = force 256x256 / 128x256 / 256x128 tiles), "no_igemm8", "no_i8_lin" (generic main loop for dense
 *                   1x1 / Linear shapes), "no_splitk", "splitk_min_nk" (tests: split short reductions too), "igemm2_tile" (1 / 2 / 3),
 *                   "no_igemm2", "igemm_tile" (1 / 2), "no_skinny", "no_tuned", "no_dual", and the per-shape choice
 *                   "ov:<M>:<C>:<K>:<R>:<S>:<stride>" / "ovh:<M>:<N>:<K>:1:1:1" / "ovd:<M>:<C1>:<C2>:<K>:<stride>:1" (tools/tune_tiles.py)
 *   streaming 1x1   "no_stream", "no_stream_narrow", "no_chain", "no_chain_stream", "no_dual_chain", "no_ln_stream", "ln_stream_192",
 *                   "no_chain_rc" (the layer-1 plan that leaves the first block output un-written: mv_conv1x1_chain_rc*_supported say no),
 *                   "no_chain_rc0" (host: first boundary on mv_conv1x1_dual_chain_fwd with y = NULL instead of mv_conv1x1_chain_rc0_fwd),
 *                   "no_chain_res" (last boundary on chain1x1's form), "no_chain_sub" (block output written whole, not sub-sampled),
 *                   "no_swin_precise" (host: plain bf16 block Linears for Swin widths off the fused kernels)
 *   whole blocks    "no_bneck_tail",
 *                   "no_ln_mlp", "ln_mlp_waves" (8 / 12 / 16), "no_ln_mlp_stream", "no_swin_block_attn", "swin_c96_shared" (the
 *                   two-windows-per-workgroup kernel at C = 96), "no_patch_merge_ln", "no_patch4_ln", "no_fc_stream"
 *   entry / misc    "stem_v0", "no_stem_pool", "no_stem_pool11", "no_patch_f32out", "no_ln_slim", "no_grouped64", "no_dwconv",
 *                   "dwconv_generic", "dwconv_no_tile", "dwconv_tile3", "no_oddc", "no_se_fused", "se_fused_always", "eltwise_scalar",
 *                   "affine_scalar", "dropout_x8", "dropout_scalar", "no_f32_mfma" (fp32 contractions back on the VALU kernel), "no_f32_lds" (only the direct fp32 matrix-core kernel), "no_attn_f32_lds" (fp32 attention back on the one-wave-per-query kernels),
 *                   "wgrad_valu" / "dgrad_valu" (the one-thread-per-element gradient kernels: cross-check)
 * The debug build (EQV_PROF=1 python -m eqxvision_amd.build -> libeqxvision_amd_prof.so) additionally reads "prof_hi" / "prof_lo"
 * (a device buffer for per-wave phase stamps), "*_prof", "i8_skew", "i8_ablate", "strip_skew", "ln_mlp_dbg", "rc_dbg" (chain_rc ablations): none of them exists in
 * the product library. */

int mv_abi_version(void);
const char* mv_last_error(void);
int mv_set_flag(const char* name, int value);   /* flags are per calling THREAD, like mv_last_error */
int mv_get_flag(const char* name);
int mv_flags_epoch(void);                        /* hash of this thread's current switch settings (0 = all default): recorded launch lists key on it */
/* Device status word: kernels that detect a broken protocol at run time record it here instead of trapping (a trap poisons the whole
 * HIP context, graph replays and other streams included).  bit 0: a split-K block gave up waiting for its partner's partial sums
 * (dirty / shared scratch; the tile it wrote is incomplete).  Writes the word to *status, clears it if `clear`; SYNCHRONISES the
 * device -- call it where the host synchronises anyway (after a forward, at the end of a test). */
int mv_device_status(int clear, unsigned* status);
/* name of the kernel variant the last call on this thread dispatched to (for tests/bench) */
const char* mv_last_kernel(void);

/* eqx.nn.Conv2d (+ eqx.experimental.BatchNorm inference + relu + residual add), reference call
 * sites resnet.py:144-162, conv_norm_activation.py:60-85, alexnet.py:44-55.
 *   y[n,ho,wo,k] = act( scale[k] * sum_{r,s,c} x[n, ho*sh-ph+r*dh, wo*sw-pw+s*dw, c] * w[k,r,s,c]
 *                       + shift[k] + residual[n,ho,wo,k] )
 * x NHWC [N,H,W,C]; w KRSC [K][R][S][C/groups] (same dtype as x); scale/shift fp32 [K] or NULL
 * (=1 / =0); residual (out_dtype, NHWC like y) or NULL. */
int mv_conv2d_nhwc_fwd(const void* x, const void* w, const float* scale, const float* shift,
                       const void* residual, void* y,
                       int N, int H, int W, int C, int K, int R, int S,
                       int sh, int sw, int ph, int pw, int dh, int dw, int groups,
                       int act, int in_dtype, int out_dtype, mv_stream_t stream);

/* Same contraction for a raw NCHW image batch (the network entry: resnet.py:243-251 stem,
 * alexnet.py:44 conv1, patch_embed.py:60-62/79-82, swin.py:705-711).  x NCHW [N,C,H,W] of
 * x_dtype; w OIHW [K][C][R][S] of out_dtype; y NHWC rows of out_dtype.
 * Token mode (ViT): if tok_stride > 0 the output row of pixel p of image n is
 * n*tok_stride + tok_offset + p and `pos` (fp32 [tok_stride][K]) row tok_offset+p is added
 * (vit.py:269: concat(cls, x) + pos_embed).  Otherwise tok_stride = tok_offset = 0, pos NULL. */
int mv_conv2d_nchw_fwd(const void* x, const void* w, const float* scale, const float* shift, void* y,
                       int N, int C, int H, int W, int K, int R, int S,
                       int sh, int sw, int ph, int pw, int act, int x_dtype, int out_dtype,
                       int tok_stride, int tok_offset, const float* pos, mv_stream_t stream);

/* The same entry with bf16 operands and an fp32 result: the ViT patch embedding (patch_embed.py:60-62, vit.py:268-269) whose token
 * rows START the fp32 residual stream -- they are not rounded to bf16 on the way (and the cast pass disappears).  w OIHW bf16, y
 * fp32 rows; non-overlapping patches only (stride = kernel, no padding, S % 8 == 0): ask the _supported entry. */
int mv_conv2d_nchw_f32out_supported(int C, int H, int W, int K, int R, int S, int sh, int sw, int ph, int pw, int x_dtype);
int mv_conv2d_nchw_f32out_fwd(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int C,
                              int H, int W, int K, int R, int S, int sh, int sw, int ph, int pw, int act, int x_dtype,
                              int tok_stride, int tok_offset, const float* pos, mv_stream_t stream);

/* The ResNet network entry in one launch (resnet.py:243-254: conv1 -> bn1 -> relu -> maxpool):
 * x NCHW [N,C,H,W] of x_dtype, w OIHW, folded BN scale/shift, y NHWC [N][Po][Qo][K] with
 * (Ho, Wo) the convolution's and (Po, Qo) the MaxPool2d(pool_k, pool_s, pool_p) output size.  The conv map
 * never reaches HBM.  mv_stem_conv_pool_supported() says whether the configuration has the path
 * (C=3, K=64, 7x7/2 pad 3, pool 3/2 pad 1, ReLU, bf16 out); otherwise mv_conv2d_nchw_fwd +
 * mv_maxpool2d_nhwc_fwd. */
int mv_stem_conv_pool_supported(int C, int K, int R, int S, int sh, int sw, int ph, int pw,
                                int pool_k, int pool_s, int pool_p, int act, int x_dtype, int out_dtype,
                                int64_t in_elems);
int mv_stem_conv_pool_fwd(const void* x, const void* w, const float* scale, const float* shift, void* y,
                          int N, int C, int H, int W, int K, int R, int S, int sh, int sw, int ph, int pw,
                          int pool_k, int pool_s, int pool_p, int act, int x_dtype, int out_dtype,
                          mv_stream_t stream);

/* Two chained pointwise layers of consecutive ResNet bottlenecks in one launch (resnet.py:144-162: block i's
 * conv3 -> bn3 -> + identity -> relu, then block i+1's conv1 -> bn1 -> relu; blocks chained by nn.Sequential, :330-333):
 *   y [M,K]  = relu(scale3[k] * (x[M,C] . w3[K,C]^T) + shift3[k] + residual[M,K])
 *   t1[M,N2] = relu(scale1[n] * (y[M,K] . w1[N2,K]^T) + shift1[n])
 * All operands bf16, NHWC rows; y is rounded to bf16 before the second product, exactly as the un-fused pair
 * (two mv_conv2d_nhwc_fwd calls) would see it.  mv_conv1x1_chain_supported() says whether the shape has the path:
 * C=64, K=256, N2=64 or 128, M >= 8192 (both weight matrices resident in LDS), or C=128, K=512, N2=128, M >= 16384
 * (ResNet-50 layer2: the weights streamed through LDS in 32-channel chunks of y).  y must not alias x, residual or t1. */
int mv_conv1x1_chain_supported(int64_t M, int C, int K, int N2, int dtype);
int mv_conv1x1_chain_fwd(const void* x, const void* w3, const float* scale3, const float* shift3,
                         const void* residual, void* y, const void* w1, const float* scale1,
                         const float* shift1, void* t1, int64_t M, int C, int K, int N2, int dtype,
                         mv_stream_t stream);

/* The tail of an identity ResNet bottleneck, CU-resident per image (resnet.py:144-162: conv2 3x3 -> bn2 -> relu -> conv3 1x1 ->
 * bn3 -> + identity -> relu; stride 1, no downsample branch), one launch, one workgroup per image:
 *   t2[B,H,W,width] = relu(scale2 * conv3x3_pad1(t1[B,H,W,width], w2) + shift2)      (stays in LDS, rounded to bf16)
 *   y [B,H,W,cout]  = relu(scale3 * (t2 . w3[cout,width]^T) + shift3 + residual[B,H,W,cout])
 * Weights in FRAGMENT ORDER (prepared once by the caller, eqxvision_amd/ops.py:prep_bneck_tail): a wave owns 32 output
 * channels and fetches each k16-step of them as one 1 KB piece,
 *   w2f[wave 0..7][tap r*3+s][j 0..width/16-1][lane 0..63][e 0..7] = w2[k = 32*wave + lane%32][r][s][c = 16*j + 8*(lane/32) + e]
 *   w3f[chunk 0..cout/256-1][wave][j][lane][e]                     = w3[k = 256*chunk + 32*wave + lane%32][c = 16*j + 8*(lane/32) + e]
 * (w2 in KRSC, w3 [cout][width]).  Supported: bf16, 14x14 maps, width 256, cout 1024 (ResNet-50/101/152 layer3).  y must
 * not alias t1 or residual. */
int mv_bottleneck_tail_supported(int H, int W, int width, int cout, int dtype);
int mv_bottleneck_tail_fwd(const void* t1, const void* w2f, const float* scale2, const float* shift2, const void* w3f,
                           const float* scale3, const float* shift3, const void* residual, void* y, int B, int H, int W,
                           int width, int cout, int dtype, mv_stream_t stream);

/* conv3 + BN and the downsample conv + BN of a stage's first bottleneck (resnet.py:144-162, 295-303) as ONE GEMM over
 * the concatenated reduction: both add into the same output, so
 *   y[N,Ho,Wo,K] = act(scale[k] * (x[N,Ho,Wo,C1] . wcat[k, 0:C1] + x2[N, s*ho, s*wo, C2] . wcat[k, C1:C1+C2]) + shift[k])
 * with x2 the block input read at pixel stride s (1 or 2) and the two BatchNorm scales folded into wcat's rows by the
 * caller (scale = NULL, shift = shift3 + shift_d).  The identity map is neither written nor read.  C1, C2 % 64 == 0. */
int mv_conv1x1_dual_supported(int64_t M, int C1, int C2, int K, int dtype);
int mv_conv1x1_dual_fwd(const void* x, const void* x2, const void* wcat, const float* scale, const float* shift,
                        void* y, int N, int Ho, int Wo, int C1, int H2, int W2, int C2, int stride2, int K,
                        int act, int dtype, mv_stream_t stream);

/* The same chain for the FIRST bottleneck of a stage whose identity is a 1x1 stride-1 convolution of the block input
 * (downsample = conv + BN, resnet.py:295-303; ResNet-50 layer1): conv3 and the downsample conv accumulate into one
 * output, so they are one GEMM over the concatenated reduction,
 *   y [M,K]  = relu(scale[k] * ([x | x2][M,C1+C2] . wcat[K,C1+C2]^T) + shift[k]),   t1 as above,
 * with wcat = [scale3 * W3 | scale_d * W_d] (the two BatchNorm scales folded into the bf16 weight rows by the caller),
 * shift = shift3 + shift_d, scale = NULL.  The identity map is never materialised.  C1 = C2 = 64, K = 256, N2 = 64. */
int mv_conv1x1_dual_chain_supported(int64_t M, int C1, int C2, int K, int N2, int dtype);
int mv_conv1x1_dual_chain_fwd(const void* x, const void* x2, const void* wcat, const float* scale,
                              const float* shift, void* y, const void* w1, const float* scale1,
                              const float* shift1, void* t1, int64_t M, int C1, int C2, int K, int N2,
                              int dtype, mv_stream_t stream);

/* ---- a bottleneck stage whose block outputs are not all written (round 6; resnet.py:144-162, 295-303, 330-333).  Between the fused
 * boundaries above the 256-channel block output y_i is still written once and read once; in a stage of three bottlenecks
 * (ResNet-50 / 101 / 152 layer1) most of that traffic can go:
 *   (1) mv_conv1x1_dual_chain_fwd with y = NULL: block 0 stores only t1 of block 1 -- y0 is recomputed by (2);
 *   (2) mv_conv1x1_chain_rc_fwd: block 1's boundary WITHOUT y0 in memory.  It reads t2 (block 1's conv2 output), t2_prev (block 0's)
 *       and x0 (the stage input), recomputes  y0 = relu([t2_prev | x0] . wcat0^T + shift0)  exactly as (1) computed it (bf16-rounded),
 *       then  y = relu(t2 . (scale1 w3)^T + shift1 + y0)  and  t1 = relu(y . (scaleN w1n)^T + shiftN).  Every BatchNorm scale is folded
 *       into the bf16 weight rows by the caller (as wcat of (1) always was); the shifts enter through the matrix pipe.
 *       wfrag: the three (scaled) weight matrices in MFMA fragment order, 128 fragments of 1 KB -- per 32-channel chunk c of y: 8 of
 *       wcat0[32c.., :] (k16-steps of [t2_prev | x0]), 4 of scale1 w3[32c.., :], 4 of scaleN w1n[:, 32c..] as (k-step s, 32-row
 *       tile a2) with the reduction index in accumulator order (slot 8 fh + i <-> channel 32 c + 8 (2 s + i / 4) + 4 fh + i % 4); a
 *       fragment is [lane = 32 fh + r][8 bf16] = W[row0 + r][k0 + 8 fh ..].  shifts: 18 rows of 64 32-bit words -- rows 0-7 shift0
 *       of chunk c, 8-15 shift1 of chunk c, 16-17 shiftN of tile a2 -- word r < 32 = bf16 hi(v) | bf16 lo(v) << 16 of v = shift[row r]
 *       (hi + lo = v to 16 mantissa bits), words 32-63 zero.  C = 64, K = 256, N2 = 64, M >= 8192;
 *   (3) mv_conv1x1_chain_sub_fwd: the LAST block's boundary when the only other consumer of y is a stride-2 pointwise convolution
 *       (the next stage's downsample branch): as mv_conv1x1_chain_fwd with N2 = 128, but y_sub = [N][H/2][W/2][K] holds only the
 *       pixels with even (h, w) -- the strided consumer reads it with stride 1 (mv_conv1x1_dual_fwd, stride2 = 1).
 *   (1') mv_conv1x1_chain_rc0_fwd: (1) rebuilt in (2)'s style -- t1 = relu(relu([t2 | x0] . wcat0^T + shift0) . (scaleN w1n)^T + shiftN),
 *       wfrag = 96 fragments (per chunk: the 8 of wcat0, the 4 of scaleN w1n, as in (2)), shifts = 10 rows (0-7 shift0, 8-9 shiftN);
 *       nothing is staged for a store, so twelve waves per CU fit instead of six.  Same shapes as (2).
 *   (3') mv_conv1x1_chain_res_fwd: mv_conv1x1_chain_fwd / (3) in (2)'s style, for the boundary whose identity comes from memory:
 *       y = relu(t2 . (scale3 w3)^T + shift3 + residual)  (all pixels, or with sub = 2 only those with even (h, w), compactly),
 *       t1 = relu(y . (scaleN w1n)^T + shiftN).  wfrag = 8 x (4 + 2 N2 / 32) fragments (per chunk c: 4 of scale3 w3[32c.., :], then the
 *       scaled w1n fragments as (k-step s, 32-row tile a2) in accumulator order), shifts = 8 rows of shift3 + N2 / 32 rows of shiftN,
 *       both as in (2).  C = 64, K = 256, N2 = 128, N H W >= 8192. */
int mv_conv1x1_chain_res_supported(int N, int H, int W, int C, int K, int N2, int sub, int dtype);
int mv_conv1x1_chain_res_fwd(const void* t2, const void* residual, const void* wfrag, const void* shifts, void* y, void* t1, int N, int H,
                             int W, int C, int K, int N2, int sub, int dtype, mv_stream_t stream);
int mv_conv1x1_chain_rc_supported(int64_t M, int C, int K, int N2, int dtype);
int mv_conv1x1_chain_rc0_fwd(const void* t2, const void* x0, const void* wfrag, const void* shifts, void* t1, int64_t M, int C, int K,
                             int N2, int dtype, mv_stream_t stream);
int mv_conv1x1_chain_rc_fwd(const void* t2, const void* t2_prev, const void* x0, const void* wfrag, const void* shifts, void* y,
                            void* t1, int64_t M, int C, int K, int N2, int dtype, mv_stream_t stream);
int mv_conv1x1_chain_sub_supported(int N, int H, int W, int C, int K, int N2, int dtype);
int mv_conv1x1_chain_sub_fwd(const void* x, const void* w3, const float* scale3, const float* shift3, const void* residual,
                             void* y_sub, const void* w1, const float* scale1, const float* shift1, void* t1, int N, int H, int W,
                             int C, int K, int N2, int dtype, mv_stream_t stream);

/* eqx.nn.Linear under vmap / Linear2d (vit.py:64,74; mlps.py:60-64; resnet.py:356;
 * extensions_2d.py:31-50):  y[M,N] = act(scale[n]*(x[M,K] . w[N,K]^T) + shift[n] + residual[M,N]).
 * in_dtype = out_dtype = MV_F32 with M <= 1024 (the classifier heads: resnet.py:356, vit.py:273, swin.py:771) runs on the
 * exact-fp32 MFMA: the logits then carry no bf16 rounding of the pooled features or of the head weights. */
int mv_linear_fwd(const void* x, const void* w, const float* scale, const float* shift,
                  const void* residual, void* y, int64_t M, int N, int K,
                  int act, int in_dtype, int out_dtype, mv_stream_t stream);

/* The same Linear / entry convolution with SPLIT-PRECISION weights: the fp32 weight of the reference is carried as two bf16
 * terms, w ~ w_hi + w_lo (w_hi = bf16(w), w_lo = bf16(w - w_hi): ~16 mantissa bits), and the kernel accumulates both products
 * in fp32.  For the few layers whose weight rounding dominates the bf16 logit error -- layers that PRODUCE the residual stream
 * instead of adding a correction to it (Swin patch embedding swin.py:705-711 and patch merging swin.py:61-65): 1.4e-2 -> 8e-3
 * on swin_t.  mv_linear_split_fwd: w_hi_lo is [N][2K] = [w_hi row | w_lo row]; otherwise as mv_linear_fwd (bf16 x).
 * mv_conv2d_nchw_split_fwd: as mv_conv2d_nchw_fwd without token mode. */
int mv_linear_split_supported(int64_t M, int N, int K, int dtype);
int mv_linear_split_fwd(const void* x, const void* w_hi_lo, const float* scale, const float* shift, const void* residual,
                        void* y, int64_t M, int N, int K, int act, int in_dtype, int out_dtype, mv_stream_t stream);
int mv_conv2d_nchw_split_fwd(const void* x, const void* w_hi, const void* w_lo, const float* scale, const float* shift,
                             void* y, int N, int C, int H, int W, int K, int R, int S, int sh, int sw, int ph, int pw,
                             int act, int x_dtype, int out_dtype, mv_stream_t stream);

/* x * scale broadcast over the pixels of each image: the last line of SqueezeExcitation.__call__ (layers/squeeze.py:60),
 * `x * scale_activation(fc2(...))`.  x, y NHWC [N, HW, C]; s [N, C]; all of `dtype`. */
int mv_channel_scale_nhwc_fwd(const void* x, const void* s, void* y, int N, int64_t HW, int C, int dtype, mv_stream_t stream);

/* SqueezeExcitation's scale vector in one launch (layers/squeeze.py:47-60): scale[n, c] = act2(b2 + w2 . act1(b1 + w1 . mean_hw x[n]))
 * for an NHWC bf16 map x[N, HW, C]; w1 [S][C] and w2t [S][C] bf16 (the first 1x1 convolution's weights, the second's TRANSPOSED:
 * both read along C), b1 / b2 fp32 or NULL; scale
 * [N][C] bf16.  act1 / act2: any MV_ACT_* (relu / silu and sigmoid / hard_sigmoid in the reference's models).  Follow with
 * mv_channel_scale_nhwc_fwd.  Flag "no_se_fused": pool, two B-row GEMMs and activation passes as separate launches. */
int mv_se_scale_supported(int C, int S, int dtype);
int mv_se_scale_fwd(const void* x, const void* w1, const float* b1, const void* w2t, const float* b2, void* scale, int N, int64_t HW,
                    int C, int S, int act1, int act2, int dtype, mv_stream_t stream);

/* Depthwise Conv2d (groups == in_channels == out_channels: mobilenetv2.py:58-68 `ConvNormActivation(hidden, hidden,
 * groups=hidden)`) with the folded BatchNorm and the activation in the same pass.  w_rsc: the (C, 1, R, S) filters re-laid
 * by the caller to [R][S][C] (channel-contiguous taps); x, y NHWC bf16; scale / shift fp32 [C] or NULL. */
int mv_dwconv2d_supported(int C, int K, int groups, int R, int S, int in_dtype, int out_dtype);
int mv_dwconv2d_nhwc_fwd(const void* x, const void* w_rsc, const float* scale, const float* shift, void* y, int N, int H, int W,
                         int C, int R, int S, int sh, int sw, int ph, int pw, int dh, int dw, int act, int in_dtype, int out_dtype,
                         mv_stream_t stream);

/* Grouped Conv2d on the matrix cores (ResNeXt conv2: resnet.py:17-27 `groups`, :440-471 `groups=32, width_per_group=4 | 8`; RegNet:
 * regnet.py:49-70, group widths 8 ... 264), for C == K and Cg = C / groups with Cg % 8 == 0 or Cg | 64.  A tile of 64 output
 * channels n0 .. n0+63 reads the contiguous WINDOW of input channels of the groups it touches, starting at (n0 / Cg) * Cg.  The CALLER
 * expands the [K][R][S][Cg] filters to w64 [K][R][S][win], win = mv_conv2d_grouped64_window(C, groups) (the widest window, a
 * multiple of 64): w64[k][r][s][(k / Cg) * Cg - ((k / 64 * 64) / Cg) * Cg + i] = w[k][r][s][i], 0 elsewhere (for Cg | 64 this is the
 * block-diagonal 64 x 64 tile).  Everything else as mv_conv2d_nhwc_fwd. */
int mv_conv2d_grouped64_supported(int C, int K, int R, int S, int groups, int in_dtype, int out_dtype);
int mv_conv2d_grouped64_window(int C, int groups);
int mv_conv2d_nhwc_grouped64_fwd(const void* x, const void* w64, const float* scale, const float* shift, const void* residual,
                                 void* y, int N, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw, int dh,
                                 int dw, int groups, int act, int in_dtype, int out_dtype, mv_stream_t stream);

/* LayerNorm folded into the Linear that consumes it (swin.py:572-578 `attn(norm1(x))` / `mlp(norm2(x))` feeding qkv / fc1;
 * extensions_2d.py:9-50), for short rows (K = 96: Swin stage 0, where the LayerNorm launch is a 115 MB round trip; K = 192 only
 * behind the "ln_stream_192" switch -- it loses there, see stream1x1.hip):
 *   y[m,:] = act( w . n(x[m,:]) + bias ),   n(x) = (x - mean) * rsqrt(var + eps)   (biased var)
 * with the LayerNorm affine folded by the caller (w = W . diag(gamma) bf16 [N][K], bias = b + W . beta fp32).
 * x is MV_F32 (the fp32 residual stream) or MV_BF16; y is MV_BF16. */
int mv_ln_linear_supported(int64_t M, int N, int K, int x_dtype, int out_dtype);
int mv_ln_linear_fwd(const void* x, const void* w, const float* bias, void* y, int64_t M, int N, int K, float eps, int act,
                     int x_dtype, int out_dtype, mv_stream_t stream);

/* The MLP half of a pre-norm block in ONE launch (swin.py:572-578 `x + stochastic_depth(mlp(norm2(x)))` in inference,
 * mlps.py:54-66, extensions_2d.py:9-28), for narrow rows whose two weight matrices fit in LDS (C = 96, hidden = 384: Swin
 * stage 0, where the hidden activations are 154 MB per 64 images):
 *   y[m,:] = x[m,:] + w2 . gelu_tanh( w1 . n(x[m,:]) + b1 ) + b2,   n(x) = (x - mean) * rsqrt(var + eps)   (biased var)
 * The LayerNorm affine is folded by the caller: w1 = W1 . diag(gamma) (bf16 [hidden][C]), b1 = b1 + W1 . beta (fp32);
 * w2 bf16 [C][hidden], b2 fp32.  x and y share x_dtype (MV_F32 = the fp32 residual stream, or MV_BF16); not in place. */
int mv_ln_mlp_supported(int64_t M, int C, int hidden, int x_dtype);
int mv_ln_mlp_fwd(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, int64_t M, int C,
                  int hidden, float eps, int x_dtype, mv_stream_t stream);

/* The ATTENTION half of a Swin block in ONE launch, one workgroup per window (swin.py:572-578 first line, 90-255, 342-366:
 * x + proj(shifted_window_attention(qkv(LayerNorm(x)))); inference, no dropout):
 *   x, y: MV_F32 [B][Hf][Wf][C] (the fp32 residual stream), not in place.  Cyclic shift, window partition / reverse and the shift
 *   mask (-100) are index arithmetic; q / k / v and the attention output never leave LDS.
 * The caller folds the LayerNorm affine into the qkv weights (W . diag(gamma), b + W . beta) and hands the weights over in
 * FRAGMENT ORDER (a wave fetches each k16-step of a 32-channel tile as one 1 KB piece; prepared by eqxvision_amd/ops.py):
 *   wqkv_f[group g 0..2][tile t 0..3*HG-1][j 0..C/16-1][lane 0..63][e 0..7] = Wqkv'[row(g,t) + lane%32][16*j + 8*(lane/32) + e]
 *       with HG = heads/3 heads per group and row(g,t) = (t/HG)*C + 32*(HG*g + t%HG): q, k, v of heads HG*g .. HG*g+HG-1;
 *       bqkv[g][t][0..31] = the matching bias slice
 *   wp_f[tile t 0..C/32-1][j][lane][e] = Wproj[32*t + lane%32][16*j + 8*(lane/32) + e];   bp[C]
 *   bias64[heads][64][64] fp32: relative-position bias bias[h][query][key] padded to 64 x 64, -1e30 on key columns >= 49
 * Supported: heads of 32 channels, 7 x 7 windows, C = 384 / 192 / 96 (swin_t / swin_s stages 2 / 1 / 0: one / two / four windows
 * per workgroup; the windows of an image must divide by that). */
int mv_swin_block_attn_supported(int Hf, int Wf, int C, int heads, int wsh, int wsw, int x_dtype);
int mv_swin_block_attn_fwd(const void* x, const void* wqkv_f, const float* bqkv, const void* wp_f, const float* bp,
                           const float* bias64, void* y, int B, int Hf, int Wf, int C, int heads, int wsh, int wsw, int shh,
                           int shw, float eps, int x_dtype, mv_stream_t stream);

/* The same MLP half for rows too wide for the weights to live in LDS (Swin stages 1 / 2: C = 192 / 384, hidden = 4 C): both weight
 * matrices are STREAMED from L2 straight into registers, 128 / 64 token rows per workgroup, the hidden activations pass through LDS in chunks
 * of 256 units and never reach HBM.  Same formula and folding as mv_ln_mlp_fwd; the weights come in FRAGMENT ORDER (prepared once
 * by the caller, eqxvision_amd/ops.py:ln_mlp): a wave fetches each k16-step of its 32 output units as one 1 KB piece,
 *   w1f[chunk 0..hidden/256-1][wave 0..7][j 0..C/16-1][lane 0..63][e 0..7] = w1[u = 256*chunk + 32*wave + lane%32][c = 16*j + 8*(lane/32) + e]
 *   w2f[chunk][tile 0..C/32-1][j 0..15][lane][e]                        = w2[c = 32*tile + lane%32][u = 256*chunk + 16*j + 8*(lane/32) + e]
 * x, y: MV_F32 (the fp32 residual stream), not in place. */
int mv_ln_mlp_stream_supported(int64_t M, int C, int hidden, int x_dtype);
int mv_ln_mlp_stream_fwd(const void* x, const void* w1f, const float* b1, const void* w2f, const float* b2, void* y, int64_t M,
                         int C, int hidden, float eps, int x_dtype, mv_stream_t stream);

/* eqx.nn.Linear over FEW rows with a big weight matrix -- the AlexNet / VGG classifiers (alexnet.py:62-70: Linear(9216, 4096),
 * Linear(4096, 4096), Linear(4096, classes), each behind `jax.vmap` = M rows) -- y[M][N] = act(x[M][K] . w^T + bias).  The layer is
 * bound by streaming w: w_frag is w in MFMA fragment order, prepared once by the caller (eqxvision_amd/ops.py:fc_fragments),
 *   w_frag[tile 0..ceil(N/32)-1][step 0..K/16-1][lane 0..63][e 0..7] = w[32*tile + lane%32][16*step + 8*(lane/32) + e]  (zero rows past N);
 * the reduction is split over the CUs and the fp32 partial sums are added in a fixed order (bit-reproducible) by a second launch.
 * x: MV_BF16 [M][K]; y: MV_BF16 or MV_F32 [M][N]; bias fp32 [N] or NULL; workspace: device memory of at least
 * mv_fc_stream_workspace(M, N, K) bytes, owned by the caller, free again when the call has completed on `stream`. */
int mv_fc_stream_supported(int64_t M, int N, int K, int in_dtype, int out_dtype);
int64_t mv_fc_stream_workspace(int64_t M, int N, int K);
int mv_fc_stream_fwd(const void* x, const void* w_frag, const float* bias, void* y, void* workspace, int64_t workspace_bytes, int64_t M,
                     int N, int K, int act, int in_dtype, int out_dtype, mv_stream_t stream);

/* Scratch memory for the NEXT launch this host thread issues on `stream`.  The library never allocates device memory; a Conv2d /
 * Linear whose output has too few tiles to fill the 256 CUs and whose reduction is long (ResNet layer 4: resnet.py:144-162 at
 * 7 x 7; Swin stage 3: swin.py:148-156, 280-300 at 49 tokens per image) is run as a two-way split over the reduction when the caller
 * provides room for the fp32 partial sums: the two halves of a tile run on two CUs, the half that finishes second adds the other's
 * partial tile to its own (a + b == b + a: the result does not depend on which one that is -- bit-reproducible) and runs the
 * usual epilogue.  mv_splitk_scratch_bytes(M, N, K_reduction) = the bytes such a launch wants for an [M] x [N] output reduced over
 * K_reduction = R * S * C (+ C2) elements, 0 when it would not split.  Protocol: the first 4096 bytes must be ZERO when the memory is
 * first handed over (the kernel leaves them zero again); the memory must stay valid until the launch has completed on `stream`, and
 * must not be shared with a launch that can run concurrently (another stream / another branch of a captured graph).  The hand-over
 * is consumed by the launch that uses it; one that does not split ignores it and LEAVES IT PENDING for the next entry on this
 * thread and stream, so a host that does not want that withdraws it (bytes = 0 / ptr = NULL).  Every consumer -- the backward
 * entries too (mv_colsum_f32, the weight gradient) -- keeps its data behind the first 4096 bytes and those stay zero, so a pending
 * offer is always a valid hand-over for whichever entry takes it.  Without scratch every launch runs un-split (same entry points,
 * same results to within fp32 summation order). */
int mv_set_scratch(void* ptr, int64_t bytes, mv_stream_t stream);
int64_t mv_splitk_scratch_bytes(int64_t M, int64_t N, int64_t K_reduction);

/* jax.image.resize(x, shape, method="bilinear") for up-sampling (segmentation/_utils.py:52-58: logits -> input resolution;
 * deeplabv3.py:66-72: the pooled ASPP branch back to the feature size): half-pixel centres, out-of-range taps dropped and the
 * rest renormalised (== clamped taps for the 2-tap kernel).  x NHWC [N,h,w,C]; y NHWC [N,H,W,C] or, with out_nchw, NCHW
 * [N,C,H,W] (the layout the reference returns).  H >= h and W >= w, else MV_E_UNSUPPORTED (the antialiased down-sampling
 * window is not on the path). */
int mv_resize_bilinear_nhwc_fwd(const void* x, void* y, int N, int h, int w, int C, int H, int W, int in_dtype, int out_dtype,
                                int out_nchw, mv_stream_t stream);

/* rows x row_bytes strided copy (jnp.concatenate of NHWC maps along channels, deeplabv3.py:132-136: one call per source with
 * dst offset by the channels already placed).  Sizes and pitches in bytes, multiples of 2. */
int mv_copy_rows(const void* src, void* dst, int64_t rows, int64_t row_bytes, int64_t src_pitch, int64_t dst_pitch,
                 mv_stream_t stream);

/* eqx.nn.MaxPool2d (resnet.py:254, alexnet.py:46,49,56): -inf padding, floor output size. */
int mv_maxpool2d_nhwc_fwd(const void* x, void* y, int N, int H, int W, int C,
                          int kh, int kw, int sh, int sw, int ph, int pw, int dtype, mv_stream_t stream);

/* eqx.nn.AdaptiveAvgPool2d (resnet.py:283, alexnet.py:59, swin.py:757), equinox chunking rule. */
int mv_adaptive_avgpool2d_nhwc_fwd(const void* x, void* y, int N, int H, int W, int C, int oh, int ow,
                                   int in_dtype, int out_dtype, mv_stream_t stream);

/* eqx.nn.LayerNorm under vmap / LayerNorm2d (vit.py:149,154,272; extensions_2d.py:9-28):
 * M rows of C, biased variance, gamma/beta fp32 [C] or NULL.  Row i of x starts at element
 * i*x_row_stride (0 = dense = C), so e.g. only the cls row of every image can be normalised
 * (vit.py:272-273 uses x[0] only); y is dense [M][C]. */
int mv_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                     int64_t M, int C, int64_t x_row_stride, float eps, int in_dtype, int out_dtype,
                     mv_stream_t stream);

/* _VitAttention core (vit.py:65-73): qkv [B,N,3,H,dh] (the qkv Linear output, channel order
 * [q|k|v][head][dh]) -> out [B,N,H*dh] = merge_heads(softmax(q k^T * scale) v);
 * probs (fp32 [B,H,N,N]) optional (NULL) -- the `attn` the reference returns. */
int mv_mha_fwd(const void* qkv, void* out, float* probs, int B, int N, int H, int dh, float scale,
               int dtype, mv_stream_t stream);

/* The same two reference lines (vit.py:64 qkv = self.qkv(x); vit.py:65-66 reshape + transpose to
 * [3, H, N, dh]) with the transposed layout produced by the projection itself: the GEMM epilogue writes
 * y HEAD-MAJOR [B][3*H][tokens][dh] (rows m = b*tokens + t, column n -> group n/dh), so that every head's
 * q/k/v block is contiguous for mv_mha_heads_fwd (128-byte pieces at a 3*H*dh stride are served by HBM
 * at about half rate -- tools/ubench/stride_read.hip).  mv_linear_heads_supported() says whether this
 * shape has the path (bf16, dh = 64, M a multiple of tokens, large enough for the 256-row GEMM);
 * otherwise use mv_linear_fwd + mv_mha_fwd. */
int mv_linear_heads_supported(int64_t M, int N, int K, int tokens, int dh, int dtype);
int mv_linear_heads_fwd(const void* x, const void* w, const float* scale, const float* shift, void* y,
                        int64_t M, int N, int K, int tokens, int dh, int dtype, mv_stream_t stream);
int mv_mha_heads_fwd(const void* qkv_head_major, void* out, float* probs, int B, int N, int H, int dh,
                     float scale, int dtype, mv_stream_t stream);

/* _VitBlock (vit.py:139-157) in inference: x = x + proj(attn(norm1(x))); x = x + fc2(gelu(fc1(norm2(x)))).  The LayerNorm
 * between a residual-producing Linear (proj, fc2) and the next Linear (fc1, the next block's qkv) is folded into the two
 * GEMM epilogues -- no LayerNorm launch, no second read of the fp32 rows:
 *     LN(y) . W^T + b  =  rstd * (y . W'^T - mean * colsum(W')) + b',     W' = W . diag(gamma),  b' = b + W . beta
 *   mv_linear_lnout_fwd   y[M][N] = residual + x[M][K] . w[N][K]^T + shift, the residual STREAM kept as two bf16 planes:
 *                         hi = bf16(y), lo = bf16(y - hi) -- the four bytes per value of fp32, ~16 mantissa bits (2^-17 relative per
 *                         store, against the 2^-9 of the operands); the high plane IS the next Linear's operand, so the fold
 *                         costs no extra copy of the rows.  Forms:
 *                           res fp32 rows (res_lo NULL)      -> planes (y, y_lo) + stats     the first block (token rows are fp32)
 *                           res planes (res, res_lo)         -> planes (y, y_lo) + stats
 *                           res planes (res, res_lo)         -> fp32 rows y (y_lo, stats NULL)   the stream leaves the split form
 *                         stats[ceil(N/256)][M][2] fp32 = per row and 256-column piece j (its last one: what is left of N) the pair
 *                         (sum, sum of squared deviations from the piece's own mean) -- merged by the consumer with Chan's
 *                         formula, so no E[x^2] - mean^2 cancellation.
 *   mv_linear_lnin_fwd    y[M][N] bf16 = act(rstd[m] * (x[M][K] . w_folded[N][K]^T - mean[m] * colsum[n]) + shift[n]), x = the
 *                         producer's HIGH plane (un-normalised rows), stats = its table ([ceil(K/256)][M][2]), (mean, rstd) with the
 *                         biased variance and `eps` as eqx.nn.LayerNorm; colsum[n] = sum_k float(w_folded[n][k]) (of the
 *                         bf16-ROUNDED values), shift = b'.  tokens > 0: head-major output as mv_linear_heads_fwd.  K <= 768.
 * The operand is rounded to bf16 BEFORE the mean is removed (the un-fused path rounds LN(y)): the relative error of the two is
 * the same while |mean| <~ std of a row, and grows with |mean| / std beyond -- the host uses the pair for the ViT blocks'
 * residual stream only (switch: no_ln_fold).  Both need the 256 x 256 GEMM tile to be the dispatch's choice for the shape. */
int mv_linear_lnout_supported(int64_t M, int N, int K, int dtype);
int mv_linear_lnout_fwd(const void* x, const void* w, const float* shift, const void* res, const void* res_lo, void* y,
                        void* y_lo, float* stats, int64_t M, int N, int K, int dtype, mv_stream_t stream);
int mv_linear_lnin_supported(int64_t M, int N, int K, int tokens, int dh, int dtype);
int mv_linear_lnin_fwd(const void* x, const float* stats, const void* w_folded, const float* colsum, const float* shift,
                       void* y, int64_t M, int N, int K, float eps, int act, int tokens, int dh, int dtype,
                       mv_stream_t stream);

/* The same attention core with the reference's LIVE attention dropout (vit.py:71, training mode: attn =
 * attn_drop(softmax(...), key)): every probability is kept (and divided by keep_prob) or zeroed by word
 * ((h * N + i) * N + j) of the sample's Threefry-2x32 stream -- jax.random.bernoulli(key, keep_prob,
 * (1, H, N, N)) bit for bit -- before it multiplies V; probs (optional) receives the dropped matrix, which
 * is what the reference returns.  keys: B x 2 uint32 (one key per sample); qkv token-major (head_major = 0,
 * as mv_mha_fwd) or head-major (1, as mv_mha_heads_fwd).  bf16, dh 32 / 64, N <= 256, else MV_E_UNSUPPORTED. */
int mv_mha_dropout_fwd(const void* qkv, int head_major, void* out, float* probs, const uint32_t* keys,
                       float keep_prob, int B, int N, int H, int dh, float scale, int dtype,
                       mv_stream_t stream);

/* _shifted_window_attention core (swin.py:123-250) on the qkv Linear2d output:
 * qkv NHWC [B,Hf,Wf,3*C] -> out NHWC [B,Hf,Wf,C]; cyclic shift, window partition/reverse and
 * the shift mask are folded into addressing; bias fp32 [heads][ws*ws][ws*ws] (table[index]). */
int mv_swin_window_attn_fwd(const void* qkv, const float* bias, void* out, int B, int Hf, int Wf, int C,
                            int heads, int ws_h, int ws_w, int shift_h, int shift_w,
                            int dtype, mv_stream_t stream);

/* The reference's two `_func_dropout` calls inside `_shifted_window_attention` (swin.py:17-20, :227, :233 -- applied in
 * EVERY mode, the function has no inference switch; both draw from the SAME key):
 * mv_swin_window_attn_dropout_fwd: the attention core with the probabilities (num_windows, heads, n, n) of every sample dropped
 * by jax.random.bernoulli(key, keep_prob, that shape) between the softmax and P . V (MFMA path only: bf16, 32 channels per
 * head, <= 64 tokens per window, else MV_E_UNSUPPORTED);
 * mv_dropout_windows_fwd: Dropout of the projection's output, an NHWC map here, in the LOGICAL order (num_windows, n, C) of the
 * shifted, window-partitioned map it has in the reference at that point.  keys: [B][2] uint32 on the device. */
int mv_swin_window_attn_dropout_fwd(const void* qkv, const float* bias, void* out, const void* keys, float keep_prob, int B,
                                    int Hf, int Wf, int C, int heads, int ws_h, int ws_w, int shift_h, int shift_w,
                                    int dtype, mv_stream_t stream);
int mv_dropout_windows_fwd(const void* x, const void* keys, void* y, int B, int Hf, int Wf, int C, int ws_h, int ws_w,
                           int shift_h, int shift_w, float keep_prob, int dtype, mv_stream_t stream);

/* _patch_merging_pad (swin.py:23-31): NHWC [B,H,W,C] -> [B,H/2,W/2,4C], channel blocks
 * [x(0::2,0::2) | x(1::2,0::2) | x(0::2,1::2) | x(1::2,1::2)], zero pad odd H/W. */
int mv_patch_merge_gather_nhwc(const void* x, void* y, int B, int H, int W, int C, int dtype, mv_stream_t stream);

/* Swin patch embedding + its LayerNorm in one launch (swin.py:705-711: LayerNorm2d(Conv2d(3, K, kernel 4, stride 4)(x))): x MV_F32
 * NCHW [B][3][H][W] -> y MV_F32 NHWC [B][H/4][W/4][K] (the fp32 residual stream).  w_hi [K][48] bf16 (OIHW flattened), w_lo = the low
 * halves of split-precision weights (hi + lo ~ the fp32 weight) or NULL, bias [K] or NULL.  K = 96 or 128; H, W multiples of 4. */
int mv_patch4_ln_supported(int C, int H, int W, int K, int x_dtype);
int mv_patch4_ln_fwd(const void* x, const void* w_hi, const void* w_lo, const float* bias, const float* gamma, const float* beta,
                     void* y, int B, int C, int H, int W, int K, float eps, int x_dtype, mv_stream_t stream);

/* Swin patch merging's gather + its LayerNorm (swin.py:23-31, 61-65: `norm(_patch_merging_pad(x))`) in one pass: row (b, i, j) of y
 * is the LayerNorm over the 4 C channels [x[2i,2j] | x[2i+1,2j] | x[2i,2j+1] | x[2i+1,2j+1]] of the fp32 NHWC map x; y [B][H/2][W/2][4C]
 * in out_dtype.  H, W even, C % 4 == 0, C <= 384. */
int mv_patch_merge_ln_supported(int H, int W, int C, int x_dtype);
int mv_patch_merge_ln_fwd(const void* x, const float* gamma, const float* beta, void* y, int B, int H, int W, int C, float eps,
                          int x_dtype, int out_dtype, mv_stream_t stream);

/* vit.py:269 row 0 of every image: tokens[b,0,:] = cls[:] + pos[0,:] (fp32 inputs). */
int mv_vit_cls_pos_fwd(const float* cls, const float* pos, void* tokens, int B, int tok_stride, int D,
                       int dtype, mv_stream_t stream);

/* jax.nn.relu / jax.nn.gelu (tanh form) / residual add as standalone ops (unfused call sites) */
int mv_eltwise_fwd(const void* x, void* y, int64_t n, int act, int dtype, mv_stream_t stream);
int mv_add_fwd(const void* a, const void* b, void* y, int64_t n, int act, int dtype, mv_stream_t stream);
/* per-channel affine y = act(x*scale[c] + shift[c]) over rows of C (stand-alone BatchNorm inference) */
int mv_channel_affine_fwd(const void* x, const float* scale, const float* shift, void* y,
                          int64_t rows, int C, int act, int dtype, mv_stream_t stream);
/* The same with the block's identity added before the activation: y = act(x * scale[c] + shift[c] + residual) -- the tail of a residual
 * block whose BatchNorm is in training mode (resnet.py:155-160 / :88-91); C must be a multiple of 8. */
int mv_channel_affine_res_fwd(const void* x, const float* scale, const float* shift, const void* residual, void* y, int64_t rows,
                              int C, int act, int dtype, mv_stream_t stream);

/* layout / dtype plumbing at the user boundary */
int mv_nchw_to_nhwc(const void* x, void* y, int N, int C, int H, int W, int in_dtype, int out_dtype, mv_stream_t stream);
int mv_nhwc_to_nchw(const void* x, void* y, int N, int C, int H, int W, int in_dtype, int out_dtype, mv_stream_t stream);
int mv_cast(const void* x, void* y, int64_t n, int in_dtype, int out_dtype, mv_stream_t stream);

/* ---- backward half of a training step (reference tests/test_grads.py:35-47: eqx.filter_value_and_grad +
 * optax.adam + eqx.apply_updates on the classification models).  fp32 throughout.  The contractions that are a forward
 * contraction on other operands (Linear dgrad / wgrad, the four products of attention's backward) go through mv_linear_fwd on
 * operands transposed with mv_transpose2d_f32 (eqxvision_amd/grad.py); these entries are the gradient kernels with no forward twin. */
/* Conv2d (eqx.nn.Conv2d incl. grouped / depthwise): dx[N,H,W,C] from dy[N,Ho,Wo,K] and w [K][R][S][C/groups]; dw in the same
 * layout from x and dy.  Both run on the fp32 matrix cores (dgrad: groups = 1).  The weight gradient's reduction runs over all
 * N Ho Wo output positions: with scratch on offer (mv_set_scratch, any size from 4096 + 2 x the gradient's bytes up) the positions are split
 * over blocks and the parts added in a fixed order (bit-reproducible); without it one block walks all positions of its tile. */
int mv_conv2d_dgrad_nhwc_f32(const float* dy, const float* w_krsc, float* dx, int N, int H, int W, int C, int K, int R, int S,
                             int sh, int sw, int ph, int pw, int dh, int dw, int groups, mv_stream_t stream);
int mv_conv2d_wgrad_nhwc_f32(const float* x, const float* dy, float* dw_krsc, int N, int H, int W, int C, int K, int R, int S,
                             int sh, int sw, int ph, int pw, int dh, int dw, int groups, mv_stream_t stream);
/* dx = dy * act'(ref), ref = the activation's INPUT: every MV_ACT_* (relu, gelu-tanh, hard_swish, hard_sigmoid, sigmoid, silu). */
int mv_act_bwd_f32(const float* dy, const float* ref, float* dx, int64_t n, int act, mv_stream_t stream);
/* eqx.nn.MaxPool2d backward: dy goes to the first maximum of each window (x = the forward input). */
int mv_maxpool2d_bwd_nhwc_f32(const float* x, const float* dy, float* dx, int N, int H, int W, int C, int kh, int kw, int sh, int sw,
                              int ph, int pw, mv_stream_t stream);
/* AdaptiveAvgPool2d((1,1)) backward: dx[n,h,w,c] = dy[n,c] / (H W). */
int mv_avgpool_global_bwd_nhwc_f32(const float* dy, float* dx, int N, int HW, int C, mv_stream_t stream);
/* Attention backward (vit.py:64-73, the op of mv_mha_fwd with the probabilities kept): dqkv[B,N,3,H,dh] from qkv (same layout),
 * probs[B,H,N,N] and dout[B,N,H*dh]; ds_scratch: B*H*N*N floats owned by the caller (the softmax gradient, written by the first of the
 * two launches and read by the second).  Every sum runs in index order. */
int mv_mha_bwd_f32(const float* qkv, const float* probs, const float* dout, float* ds_scratch, float* dqkv, int B, int N, int H, int dh,
                   float scale, mv_stream_t stream);
/* out[c] = sum over the M rows of a[m,c] * (b ? b[m,c] : 1): bias / beta gradients (b = NULL), gamma gradients (b = x_hat).  From 512
 * rows up it takes 4096 bytes + ceil(M / 256) (<= 512) x C floats of scratch when on offer (mv_set_scratch): rows split over blocks, fixed-order finish. */
int mv_colsum_f32(const float* a, const float* b, float* out, int64_t M, int C, mv_stream_t stream);
/* y = x * s[b, c] (SqueezeExcitation's multiply, DropPath): ds[b, c] = sum over the image's HW positions of g * x; dx is the
 * forward multiply applied to g (mv_channel_scale_nhwc_fwd). */
int mv_channel_scale_bwd_f32(const float* g, const float* x, float* ds, int B, int HW, int C, mv_stream_t stream);
/* BatchNorm gamma gradient when the normalisation used the RUNNING statistics (the reference's training branch; the state is not
 * differentiated): dgamma[c] = (sum dy z - mean[c] sum dy) / sqrt(var[c] + eps), from the two column sums. */
int mv_bn_dgamma_f32(const float* sum_dy_z, const float* sum_dy, const float* mean, const float* var, float eps, float* dgamma, int C,
                     mv_stream_t stream);
/* eqx.experimental.BatchNorm TRAINING branch, input gradient through the batch statistics.  The layer normalises with
 * running' = a * batch + (1 - a) * running (a = 1 on the first call, 1 - momentum afterwards) and the batch mean / variance are
 * differentiable functions of z, so   dz = scale * dy + A[c] + B[c] * z   with scale = gamma * rstd(running'),
 *   B[c] = -(a / n) * scale * rstd^2 * (sum dy z - mean'[c] sum dy),   A[c] = -(a / n) * scale * sum dy - B[c] * batch_mean[c],
 * batch_mean = sum_z / n.  The three column sums are over ALL n rows of the data-parallel batch (summed over ranks by the caller);
 * n = *count when count != NULL (device scalar: ragged shards), else rows. */
int mv_bn_train_dz_coef_f32(const float* sum_dy, const float* sum_dy_z, const float* sum_z, const float* mean, const float* var,
                            const float* scale, const float* count, float rows, float a, float eps, float* A, float* B, int C,
                            mv_stream_t stream);
/* eqx.nn.LayerNorm backward over M rows of C (x = the forward input): dx, and dy_xhat = dy * x_hat (its column sums = dgamma;
 * the column sums of dy = dbeta). */
int mv_layernorm_bwd_f32(const float* x, const float* gamma, const float* dy, float* dx, float* dy_xhat, int64_t M, int C, float eps,
                         mv_stream_t stream);
/* softmax backward for p = softmax(scale * s) row-wise: ds = scale * p * (dp - sum_j dp_j p_j). */
int mv_softmax_bwd_f32(const float* p, const float* dp, float* ds, int64_t rows, int cols, float scale, mv_stream_t stream);
/* optax.softmax_cross_entropy(logits, target).mean() (tests/test_grads.py:41): loss_rows[B], loss_mean[1], dlogits = d mean / d logits. */
int mv_softmax_xent_f32(const float* logits, const float* target, float* loss_rows, float* loss_mean, float* dlogits, int B, int K,
                        mv_stream_t stream);
/* optax.adam (tests/test_grads.py:52): m, v updated in place, update = -lr * (m / bias_corr1) / (sqrt(v / bias_corr2) + eps) with
 * bias_corr = 1 - beta^t computed by the caller; eqx.apply_updates adds `update` to the parameter (mv_add_fwd). */
int mv_adam_step_f32(const float* grad, float* m, float* v, float* update, int64_t n, float lr, float b1, float b2, float eps,
                     float bias_corr1, float bias_corr2, mv_stream_t stream);
/* Backward of mv_swin_window_attn_fwd (swin.py:123-250), fp32: dqkv[B,Hf,Wf,3C] from qkv, the relative-position bias
 * [heads][n][n] and dout[B,Hf,Wf,C]; g_windows [B * windows][heads][n * n] receives the gradient w.r.t. the attention logits of
 * every window (its column sum over the windows = the bias gradient).  Windows of <= 64 tokens. */
int mv_swin_window_attn_bwd_f32(const float* qkv, const float* bias, const float* dout, float* dqkv, float* g_windows, int B, int Hf,
                                int Wf, int C, int heads, int ws_h, int ws_w, int shift_h, int shift_w, mv_stream_t stream);
/* out[t][c] = sum of src[m][c] over the rows m with index[m] == t (t < T): the gradient of a table gathered by `index`
 * (swin.py:34-43: relative_position_bias_table[relative_position_index]). */
int mv_scatter_rows_sum_f32(const float* src, const int* index, float* out, int M, int C, int T, mv_stream_t stream);
/* Backward of mv_patch_merge_gather_nhwc (swin.py:23-31) for even H, W: dx[B,H,W,C] from dy[B,H/2,W/2,4C]. */
int mv_patch_merge_gather_bwd_f32(const float* dy, float* dx, int B, int H, int W, int C, mv_stream_t stream);
/* y[C][R] = x[R][C]^T, rows of x x_row_stride elements apart (0 = dense). */
int mv_transpose2d_f32(const float* x, float* y, int R, int C, int64_t x_row_stride, mv_stream_t stream);

/* hipGraph capture of a whole forward (replaces the reference's eqx.filter_jit executable) */
int mv_graph_begin_capture(mv_stream_t stream);
int mv_graph_end_capture(mv_stream_t stream, void** graph_exec);
int mv_graph_launch(void* graph_exec, mv_stream_t stream);
int mv_graph_destroy(void* graph_exec);

/* Training-mode Dropout (eqx.nn.Dropout with inference = False: alexnet.py:63-68, vit.py:39-53, mlps.py:43-52 before
 * tree_inference): y = where(bernoulli(key_b, keep_prob, x_b.shape), x_b / keep_prob, 0) for every sample b, keys [B][2]
 * uint32 on the device (one jax.random key per sample, as under vmap).  The mask is JAX's bit stream: Threefry-2x32 words in
 * jax.random's counter layout, word i for element i of the LOGICAL single-sample array in row-major order -- the reference's
 * (C,H,W) when chw_logical != 0 (x is NHWC [B][per_sample / C][C] here), else the physical order ((N,D) rows, (D,) vectors). */
int mv_dropout_fwd(const void* x, const void* keys, void* y, int B, int64_t per_sample, int C, int chw_logical, float keep_prob,
                   int dtype, mv_stream_t stream);

/* jax.random.split(key, num) for R keys at once, on the device (the reference derives one key per TOKEN for the vmapped
 * transformer MLP, vit.py:155: R = batch * tokens keys per block is host work worth a launch): keys [R][2] uint32 ->
 * out [R][num][2] (child_major = 0: the children of a key together, e.g. the per-token keys of every sample) or [num][R][2]
 * (child_major = 1: what a vmapped split returns, child i of every key contiguous). */
int mv_prng_split(const void* keys, void* out, int64_t R, int num, int child_major, mv_stream_t stream);

/* DropPath's draws (drop_path.py:51-61, training mode) as the scale mv_channel_scale_nhwc_fwd multiplies by: out [B][C] =
 * bernoulli(key_b, keep_prob) / keep_prob for the whole sample (local = 0, mode "global") or bernoulli(key_b, keep_prob, (C,))[c] /
 * keep_prob (local = 1, mode "local": one draw per entry of the sample's first axis), in `dtype`. */
int mv_drop_path_noise(const void* keys, void* out, int B, int C, int local, float keep_prob, int dtype, mv_stream_t stream);

/* Per-channel batch moments of rows x[rows][C] (an NHWC map or a row matrix): the statistics of eqx.experimental.BatchNorm's
 * TRAINING branch (reference resnet.py:132-136 / :252 / :301 with the model not in inference mode):
 *   out[c] = sum over rows of (x[r][c] - shift[c])        (squared = 0; shift may be NULL)
 *   out[c] = sum over rows of (x[r][c] - shift[c])^2      (squared = 1)
 * Two passes like the reference (mean, then the mean of squared deviations from the cross-replica mean); the caller divides by
 * the global row count after mv_allreduce_sum_f32.  workspace: mv_channel_moments_ws(C) floats.  Deterministic summation order. */
int mv_channel_moments_ws(int C);
int mv_channel_moments_supported(int64_t rows, int C, int dtype);
int mv_channel_moments_fwd(const void* x, const float* shift, float* out, float* workspace, int64_t rows, int C, int squared,
                           int dtype, mv_stream_t stream);
/* The rest of the training-mode BatchNorm update on the device, so that a step has no host round trip: mean = sum / n after the
 * first pass (n: the global row count, from the host, or -- ragged shards -- the all-reduced count on the device); after the second:
 * var = sqdev / n, running statistics in place (first call: = batch; else (1 - momentum) * batch + momentum * running, momentum 0.99
 * in the reference), and the scale / shift the normalisation applies (mv_channel_affine_fwd): weight / sqrt(running_var + eps),
 * bias - running_mean * scale -- the UPDATED running statistics, as eqx.experimental.BatchNorm does. */
/* Steady state (running statistics exist): ONE pass and ONE all-reduce of 2 x C floats per layer.  Both moments are taken about the
 * running mean of the previous steps -- the same on every rank, so the ranks' sums add up, and within a fraction of sigma of the batch
 * mean, so var = S2 / n - (S1 / n)^2 loses nothing: out[0:C] = sum (x - shift), out[C:2C] = sum (x - shift)^2 (workspace:
 * 2 * mv_channel_moments_ws(C) floats); mv_bn_ema_fold1_fwd then does what mv_bn_mean_fwd + mv_bn_ema_fold_fwd do for the two-pass
 * form (which the first step of a BatchNorm, with no running statistics yet, still takes -- the reference's literal order). */
int mv_channel_moments2_fwd(const void* x, const float* shift, float* out, float* workspace, int64_t rows, int C, int dtype,
                            mv_stream_t stream);
int mv_bn_ema_fold1_fwd(const float* sums, const float* count_dev, float count_host, float* run_mean, float* run_var,
                        const float* weight, const float* bias, float* scale, float* shift, float momentum, float eps, int C,
                        mv_stream_t stream);
int mv_bn_mean_fwd(const float* sum, const float* count_dev, float count_host, float* mean, int C, mv_stream_t stream);
int mv_bn_ema_fold_fwd(const float* sqdev, const float* mean, const float* count_dev, float count_host, float* run_mean,
                       float* run_var, const float* weight, const float* bias, float* scale, float* shift, float momentum, float eps,
                       int first_time, int C, mv_stream_t stream);

/* The path's one collective: the batch axis of `jax.vmap(net, axis_name="batch")(images)` (README.md:37-40)
 * shards over the GPUs of a node, one process per GPU; rank r runs images[r*B/W:(r+1)*B/W] and ONE all-gather of the fp32 logits
 * rebuilds the (B, classes) array the single-device vmap returns.  RCCL over xGMI (librccl is dlopen'ed by the first call;
 * $EQV_RCCL_LIB overrides the search).  Bootstrap: rank 0 calls mv_comm_unique_id and ships the MV_COMM_ID_BYTES bytes to
 * the other ranks out of band (the Python host uses a torch.distributed gloo store); every rank then calls mv_comm_init
 * with its HIP device already selected.  mv_allgather is enqueued on `stream` like every other call: recv[r*bytes : (r+1)*bytes]
 * = rank r's send buffer.  One communicator per process. */
#define MV_COMM_ID_BYTES 128
int mv_comm_unique_id(void* out, size_t out_bytes);
int mv_comm_init(int rank, int nranks, const void* unique_id);
int mv_comm_size(void); /* 0 = no communicator */
int mv_comm_rank(void);
int mv_allgather(const void* send, void* recv, size_t bytes_per_rank, mv_stream_t stream);
int mv_allreduce_sum_f32(void* buf, size_t count, mv_stream_t stream);   /* in place, sum over ranks (training-mode BatchNorm moments) */
int mv_comm_destroy(void);

/* HIP-event timing helpers on the caller's stream (bench.py's live roofline measurement) */
int mv_event_create(void** ev);
int mv_event_record(void* ev, mv_stream_t stream);
int mv_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on stop */
int mv_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* EQXVISION_AMD_H */
