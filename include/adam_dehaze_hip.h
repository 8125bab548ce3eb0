/*
 * adam_dehaze_hip.h -- C ABI of libadamdehaze_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary of the ADAM-Dehaze hot path.  The reference has no FFI, plugin registry or
 * operator table: its boundary is Python (factory functions + nn.Module.forward).
 * Each entry point below therefore replaces a group of ATen calls made by one reference Module /
 * function (cited per entry as file:line under /root/reference); the Python host in
 * adam-dehaze_amd/ binds them with ctypes and keeps the reference's class / factory names.
 *
 * Conventions
 *   - every function returns 0 on success, a negative ADH_E_* code otherwise (the Python host raises
 *     RuntimeError); nothing is allocated inside, nothing synchronises: work is enqueued on `stream`
 *     (a hipStream_t passed as void*), so calls are graph-capturable.
 *   - activations are fp32 NHWC: element (n,y,x,c) of a tensor with `cstride` floats per pixel is at
 *     ptr[((n*H + y)*W + x)*cstride + c]; `cstride` >= C lets a tensor be a channel slice of a wider
 *     buffer (zero-copy skip concatenation).  Pointers and cstrides must be multiples of 4 floats.
 *   - images at the model boundary are the reference's fp32 NCHW [N,3,H,W].
 *   - "packed weights": float4 wp[(tap*KQ + kq)*NcP + n] holds k = 4*kq..4*kq+3 of output column n,
 *     KQ = ceil(K/8)*2, NcP = Nc rounded up to 32 (zero padded).
 */
#ifndef ADAM_DEHAZE_HIP_H
#define ADAM_DEHAZE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADH_OK 0
#define ADH_E_ARG (-1)      /* bad argument (shape / alignment / null) */
#define ADH_E_LAUNCH (-2)   /* hipLaunch failed */
#define ADH_E_UNSUPPORTED (-3)

#define ADH_ACT_NONE 0
#define ADH_ACT_RELU 1
#define ADH_ACT_SIGMOID 2
#define ADH_ACT_TANH 3

/* Gather-form convolution descriptor shared by the forward / dgrad / wgrad kernels.
 * Virtual output grid VH x VW; virtual pixel (vy,vx) reads input pixel
 * (vy*in_sy + dy, vx*in_sx + dx) for tap (ty,tx): dy = dy0 + ty*dstep_y, dx = dx0 + tx*dstep_x
 * (zero outside the image) and writes output pixel (vy*out_sy + out_oy, vx*out_sx + out_ox).
 * This one form covers Conv2d k3/k7/k1 s1, Conv2d k4 s2, each output-parity class of
 * ConvTranspose2d k4 s2 p1, and the data-gradient of all of them. */
typedef struct adh_conv_desc {
    const float* in;        /* NHWC [N][IH][IW][in_cstride], channels [0,Cin) used          */
    float* out;             /* NHWC [N][OH][OW][out_cstride]; wgrad: grad wrt this (read)   */
    const float* wp;        /* packed weights (fwd/dgrad); wgrad: unused                    */
    const float* scale;     /* per-output-channel multiplier or NULL (=1)                   */
    const float* shift;     /* per-output-channel addend (bias / folded BN) or NULL (=0)    */
    const float* residual;  /* NHWC tensor added before the activation, or NULL             */
    float* stats;           /* NULL, or [n_blocks][2][NcP] per-block sum / sum-of-squares of
                               the value after scale/shift (train-mode BatchNorm statistics) */
    int32_t N, IH, IW, Cin, in_cstride;
    int32_t OH, OW, Cout, out_cstride, res_cstride;
    int32_t VH, VW;
    int32_t in_sy, in_sx;
    int32_t out_sy, out_sx, out_oy, out_ox;
    int32_t KH, KW;
    int32_t dy0, dx0, dstep_y, dstep_x;
    int32_t act;            /* ADH_ACT_NONE / ADH_ACT_RELU applied last                      */
    int32_t NcP;            /* padded output-channel count of wp / stats                     */
} adh_conv_desc;

/* Source-layout description used to pack weights for, and to scatter weight gradients from, the
 * gather form: element (tap (ty,tx), k, n) lives at
 *   src[tap_off0 + ty*tap_off_sy + tx*tap_off_sx + k*stride_k + n*stride_n].
 * Conv2d OIHW forward: k=ci (stride KH*KW), n=co (stride Cin*KH*KW); its dgrad swaps k and n;
 * ConvTranspose2d [Cin][Cout][kh][kw] likewise (models/dehazing/base_model.py:11-13,
 * medium_intensity.py:53,63). */
typedef struct adh_wlayout {
    int32_t K, Nc;
    int32_t KHt, KWt;
    int32_t tap_off0, tap_off_sy, tap_off_sx;
    int32_t stride_k, stride_n;
} adh_wlayout;

/* ---- library / device ------------------------------------------------------------------- */
int adh_version(void);
/* bytes of dynamic LDS the conv kernels need for `d` (diagnostic; >0) */
int adh_conv_lds_bytes(const adh_conv_desc* d);
/* number of thread blocks adh_conv_forward launches for `d` (= rows of d->stats) */
int adh_conv_num_blocks(const adh_conv_desc* d);

/* ---- weights ------------------------------------------------------------------------------ */
/* pack `src` (layout L) into wp; wp has KHt*KWt*KQ*NcP float4.  Replaces nothing in the reference
 * (ATen keeps OIHW); it is the one-time re-layout for the MFMA kernels. */
int adh_pack_weights(void* stream, const float* src, const adh_wlayout* L, float* wp);

/* ---- convolution (base_model.py:11-13 Conv2d, medium_intensity.py:53,63 ConvTranspose2d) ---- */
/* out = act(scale*gather_conv(in, wp) + shift + residual); optional BN statistics partials. */
int adh_conv_forward(void* stream, const adh_conv_desc* d);
/* Winograd F(2x2,3x3) path for 3x3 stride-1 pad-1 convolutions (and their data gradients): same descriptor
 * and fused epilogue, d->wp packed by adh_pack_weights_wino (U = G g G^T per (k, n), [16][K/4][NcP][4]).
 * adh_conv_wino_supported returns 1 when `d` is eligible (KH=KW=3, unit strides, dy0=dx0=-1, Cin % 16 == 0). */
int adh_conv_wino_supported(const adh_conv_desc* d);
int adh_conv_wino_forward(void* stream, const adh_conv_desc* d);
/* rows of d->stats adh_conv_wino_forward writes (one per 8x32 output region); adh_conv_num_blocks is the count for
 * adh_conv_forward, whose kernels use other region shapes */
int adh_conv_wino_num_blocks(const adh_conv_desc* d);
int adh_pack_weights_wino(void* stream, const float* src, const adh_wlayout* L, float* wp);
/* Winograd F(3x3,2x2) path for the 2x2-tap gather forms (KH = KW = 2, dstep == in_s: a parity class of
 * ConvTranspose2d k4 s2 or of a k4 s2 data gradient) and for KH = KW = 4, in_s = 2, dstep = 1 (Conv2d k4 s2 / the data
 * gradient of ConvTranspose2d k4 s2, as four input-parity classes); Cin % 16 == 0.  d->wp packed by
 * adh_pack_weights_wino32 (L = the 2x2 layout, or the 4x4 layout for the four-class form):
 * [classes][16][K/4][NcP][4] floats.  Same fused epilogue; statistics rows = adh_conv_wino32_num_blocks. */
int adh_conv_wino32_supported(const adh_conv_desc* d);
int adh_conv_wino32_num_blocks(const adh_conv_desc* d);
int adh_conv_wino32_forward(void* stream, const adh_conv_desc* d);
/* descs[0 .. n-1] (n <= 4): the single-class launches of ONE layer -- the four output-parity classes of a ConvTranspose2d k4 s2 p1
 * or of the data gradient of a Conv2d k4 s2 p1 -- as one grid.  They may differ in wp, out_oy / out_ox, dy0 / dx0 and stats only
 * (ADH_E_UNSUPPORTED otherwise: launch them one by one).  Same results as n calls of adh_conv_wino32_forward; one partial last
 * round of workgroups instead of n. */
int adh_conv_wino32_forward_multi(void* stream, const adh_conv_desc* descs, int n);
int adh_pack_weights_wino32(void* stream, const float* src, const adh_wlayout* L, float* wp);
/* Opt-in bf16 x 3 contraction of the same launches (round 4; host switch ADH_CONTRACT=bf16x3; see adh_conv_wino43_forward_bf16x3):
 * same descriptors, same epilogue, same statistics rows; d->wp from adh_pack_weights_wino32_bf16x3 (ncls * 16 * Kp * NcP * 6
 * bytes, Kp = K rounded up to 16, NcP = Nc rounded up to 32).  Replaces the same ATen calls as adh_conv_wino32_forward
 * (/root/reference models/dehazing/medium_intensity.py:53,63 ConvTranspose2d k4 s2; base_model.py:11-13 Conv2d k4 s2). */
int adh_conv_wino32_forward_bf16x3(void* stream, const adh_conv_desc* d);
int adh_conv_wino32_forward_multi_bf16x3(void* stream, const adh_conv_desc* descs, int n);
int adh_pack_weights_wino32_bf16x3(void* stream, const float* src, const adh_wlayout* L, void* wp);

/* Winograd F(4x4,3x3) on fp32 MFMA (conv_wino43.hip): the same 3x3 stride-1 pad-1 forms as adh_conv_wino_forward at
 * 1/4 of the direct algorithm's MFMA work (F(2x2,3x3): 4/9); needs Cin % 16 == 0.  Replaces the same ATen conv2d calls
 * (/root/reference models/dehazing/base_model.py:11-13,26-41).  Weights: adh_pack_weights_wino43 ->
 * [36][K/4][NcP][4] floats (U = G g G^T).  fp32 throughout; transform constants up to 8 make its rounding error
 * ~10x the direct kernel's (still ~1e-6 relative).  Statistics rows = adh_conv_wino43_num_blocks. */
int adh_conv_wino43_supported(const adh_conv_desc* d);
int adh_conv_wino43_num_blocks(const adh_conv_desc* d);
int adh_conv_wino43_forward(void* stream, const adh_conv_desc* d);
/* The data gradient of such a layer together with the BatchNorm-backward sums of the layer that PRODUCED its input (one
 * launch instead of the data gradient + adh_bn_bwd_reduce over the same tensor): d describes the data-gradient convolution,
 * whose output is the producer's output gradient g; d->residual / res_cstride = the producer's raw convolution output y (not
 * added to anything), d->scale / d->shift = the producer's forward BatchNorm scale / shift (its ReLU mask is
 * m = [fma(y, scale, shift) > 0], as in adh_bn_apply), bn_mean = its batch mean, d->act = ADH_ACT_NONE.  g is stored unchanged;
 * d->stats gets [adh_conv_wino43_num_blocks(d)][2][NcP] rows of (sum g m, sum g m (y - mean)) for
 * adh_bn_bwd_finalize_centered.  For a ConvBlock whose only consumer is a 3x3 convolution -- the first half of every
 * ResidualBlock (/root/reference models/dehazing/base_model.py:4-24,26-41). */
int adh_conv_wino43_dgrad_bnred(void* stream, const adh_conv_desc* d, const float* bn_mean);
int adh_pack_weights_wino43(void* stream, const float* src, const adh_wlayout* L, float* wp);
/* Launch shape of the F(4x4,3x3) kernels (library-wide; returns the previous setting).  1 (default): persistent -- one workgroup per
 * CU walks a static share of the regions and stages the next region's first chunk during the current one's last contraction
 * (DESIGN 4.17).  0: one workgroup per region, for processes whose other streams hold CUs while these kernels run (the
 * data-parallel step: gradient all-reduces beside the backward pass) -- a static share would make the launch wait for the
 * workgroups that start late.  No counterpart in the reference (ATen picks its own launch shapes). */
int adh_conv_wino43_set_persistent(int on);
/* Opt-in (round 4; host switch ADH_CONTRACT=bf16x3, the default stays the fp32 MFMA): the same two launches with the contraction
 * on v_mfma_f32_32x32x16_bf16.  Both operands are split EXACTLY into three bf16 planes (x = hi + mid + lo, 3 x 8 significant
 * bits; the transformed input inside the kernel, U at pack time) and the six significant cross terms are accumulated in fp32:
 * the three dropped ones sit at 2^-24 of a product, measured error against fp64 0.7 - 0.9x the fp32 MFMA path's
 * (profiles/r03_micro_bf16split.txt, profiles/r04_bf16x3_error.txt).  Same descriptor, same epilogue, same statistics rows as
 * the fp32 entry points; d->wp from adh_pack_weights_wino43_bf16x3 (36 * Kp * NcP * 6 bytes, Kp = K rounded up to 16, NcP = Nc
 * rounded up to 32; laid out per channel group as the launch picks it from NcP).  Replaces the same ATen conv2d calls
 * (/root/reference models/dehazing/base_model.py:11-13,26-41). */
int adh_conv_wino43_forward_bf16x3(void* stream, const adh_conv_desc* d);
int adh_conv_wino43_dgrad_bnred_bf16x3(void* stream, const adh_conv_desc* d, const float* bn_mean);
int adh_pack_weights_wino43_bf16x3(void* stream, const float* src, const adh_wlayout* L, void* wp);

/* Weight gradient of the 3x3 s1 p1 convolutions with few channels (conv_wgrad_small.hip, MFMA 16x16x4: one tile = one tap
 * x 16 input channels x up to 16 output channels): channel strides (in, out) = (8, 16), (16, 16) or (48, 8), Cout <= 16 --
 * the guidance branch and the output convolution of the Complex model
 * (/root/reference models/dehazing/high_intensity.py:74-90).  adh_conv_wgrad_small_slabs(d) = number of slabs
 * [9][KP][NcP] the launch writes (0: not one of its shapes); adh_wgrad_reduce_small adds them (one wave per element). */
int adh_conv_wgrad_small_slabs(const adh_conv_desc* d);
int adh_conv_wgrad_small(void* stream, const adh_conv_desc* d, float* slab, int KP, int NcP);
int adh_wgrad_reduce_small(void* stream, const float* slab, int nslabs, int KP, int NcP, const adh_wlayout* L, float* dst,
                           int accumulate);

/* Forward of a 3x3 s1 p1 convolution with at most four output channels (conv_fewout.hip: one output pixel per thread, halo in
 * LDS, weights through the scalar cache -- the MFMA kernels would pad 3 channels to a 32-wide tile): the reconstruction head
 * Conv2d(C -> 3) of every branch (/root/reference models/dehazing/high_intensity.py:78, medium_intensity.py, low_intensity.py).
 * Same descriptor; epilogue scale / shift / ReLU, no residual, no statistics.  Weights: adh_pack_weights_fewout ->
 * [9][K8 / 4][4][4] floats with K8 = round_up(K, 8) = d->Cin. */
int adh_conv_fewout_supported(const adh_conv_desc* d);
int adh_conv_fewout_forward(void* stream, const adh_conv_desc* d);
int adh_pack_weights_fewout(void* stream, const float* src, const adh_wlayout* L, float* wp);
/* The mirror image: at most four INPUT channels (K <= 4 in the first channel quad of an NHWC8 tensor: d->Cin = 8) and Cout = 4 .. 64
 * (multiple of 4) output channels: the data gradient of the reconstruction head (3 -> 48; pass the flipped layout, as for the
 * Winograd data gradients) and the guidance branch's first layer in eval mode.  Weights: adh_pack_weights_fewin ->
 * [9][4][CO] floats, CO = 16 / 48 / 64 >= Cout.  Statistics (Cout <= 16 only): adh_conv_fewin_num_blocks(d) rows. */
int adh_conv_fewin_supported(const adh_conv_desc* d);
int adh_conv_fewin_num_blocks(const adh_conv_desc* d);
int adh_conv_fewin_forward(void* stream, const adh_conv_desc* d);
int adh_pack_weights_fewin(void* stream, const float* src, const adh_wlayout* L, float* wp);

/* Forward of the 7x7 s1 p3 stem (3 -> 64 / 96 channels on the NHWC8 image; conv_stem.hip): same descriptor and fused
 * epilogue as adh_conv_forward (scale / shift, ReLU, BatchNorm partial statistics; no residual), weights packed by
 * adh_pack_weights_stem -> [7][24][NcP] floats.  Statistics rows = adh_conv_stem_num_blocks(d) (0: not the stem).
 * Replaces ATen's conv2d for init_conv (/root/reference models/dehazing/high_intensity.py:24). */
int adh_conv_stem_num_blocks(const adh_conv_desc* d);
int adh_conv_stem_forward(void* stream, const adh_conv_desc* d);
int adh_pack_weights_stem(void* stream, const float* src, const adh_wlayout* L, float* wp);

/* Weight gradient of the 7x7 s1 p3 stem (3 -> 64 / 96 channels on the NHWC8 image; conv_stem.hip, MFMA 16x16x4 with the
 * (kx, c) filter entries packed along the tile rows).  adh_conv_wgrad_stem_slabs(d) = number of slabs [49][8][NcP] the
 * launch writes (0: not the stem); add them with adh_wgrad_reduce_small(slab, n, 8, NcP, L, dW).  Replaces the weight half
 * of ATen's conv2d backward for init_conv (/root/reference models/dehazing/high_intensity.py:24). */
int adh_conv_wgrad_stem_slabs(const adh_conv_desc* d);
int adh_conv_wgrad_stem(void* stream, const adh_conv_desc* d, float* slab, int NcP);

/* Weight gradient of the same layers in the F(4x4,3x3) Winograd domain (conv_wgrad43.hip): 1/4 of the direct MFMA
 * work.  Needs Cin % 32 == 0, Cout % 96 == 0, H % 4 == 0, W % 4 == 0 (adh_conv_wgrad_wino43_groups(d) > 0: its value is
 * the number of workgroups per pixel split, one per 32 input x 96 output channels; adh_conv_wgrad_wino43_strips(d) = the
 * 4 x 16-pixel strips the splits divide among themselves).  d->in = x, d->out = dL/dy as for adh_conv_wgrad.  slab: nsplit * 36 * Cin * NcP floats; adh_wgrad_reduce_wino43 sums the splits
 * (in place) and applies A'^T (.) A' with the 1/(N_a N_b) normalisation in double.  Replaces the weight half of ATen's
 * conv2d backward (/root/reference models/dehazing/base_model.py:11-13). */
int adh_conv_wgrad_wino43_groups(const adh_conv_desc* d);
int adh_conv_wgrad_wino43_strips(const adh_conv_desc* d);
int adh_conv_wgrad_wino43(void* stream, const adh_conv_desc* d, float* slab, int nsplit);
int adh_wgrad_reduce_wino43(void* stream, float* slab, int nsplit, int KP, int NcP, const adh_wlayout* L, float* dst,
                            int accumulate);
/* weight gradient of the gather form: slab[s][tap][KP][NcP] partial sums over `nsplit` pixel
 * ranges (KP = Cin rounded up to 32); d->out is the gradient wrt the conv output. */
int adh_conv_wgrad(void* stream, const adh_conv_desc* d, float* slab, int nsplit);
/* number of slabs adh_conv_wgrad writes for (d, nsplit): nsplit, or 4*nsplit on the 3x3 unit-stride path whose
 * four waves each keep their own partial sums; size `slab` and call adh_wgrad_reduce with this count. */
int adh_conv_wgrad_slabs(const adh_conv_desc* d, int nsplit);
/* 3x3 unit-stride path only (0 otherwise): workgroups per pixel split.  One workgroup is resident per CU, so
 * callers pick nsplit with groups*nsplit close to a multiple of the CU count. */
int adh_conv_wgrad_groups(const adh_conv_desc* d);
/* dst(layout L) (+)= sum_s slab[s]; deterministic order. accumulate != 0 adds to dst. */
int adh_wgrad_reduce(void* stream, const float* slab, int nsplit, int KP, int NcP,
                     const adh_wlayout* L, float* dst, int accumulate);

/* Winograd-domain weight gradient of Conv2d k3 s1 p1 (VW % 32 == 0, VH % 4 == 0, Cin % 32 == 0, Cout % 32 == 0):
 * slab[s][16][KP][NcP] holds d/d(G w G^T) per frequency at 4/9 of the direct MFMA work; adh_wgrad_reduce_wino sums
 * the splits and applies G^T(.)G into the 3x3 layout L.  adh_conv_wgrad_wino_groups: workgroups per pixel split
 * (0 = descriptor not eligible, use adh_conv_wgrad). */
int adh_conv_wgrad_wino_groups(const adh_conv_desc* d);
int adh_conv_wgrad_wino(void* stream, const adh_conv_desc* d, float* slab, int nsplit);
int adh_wgrad_reduce_wino(void* stream, float* slab /* scratch: split 0 receives the sum */, int nsplit, int KP, int NcP,
                          const adh_wlayout* L, float* dst, int accumulate);

/* Winograd F(3x3,2x2)-domain weight gradient of the 2x2-tap gather forms (conv_wgrad32.hip): every output-parity class of
 * ConvTranspose2d k4 s2 p1 (KH = KW = 2 descriptors, dstep = +-in_s) and Conv2d k4 s2 p1 (KH = KW = 4, in_s = 2: four
 * kernel-parity classes inside one call), Cin % 32 == 0, any grid size, at 4/9 of the direct MFMA work.
 * adh_conv_wgrad_wino32_groups(d): workgroups per pixel split (0 = not eligible, use adh_conv_wgrad);
 * _classes(d): 1 or 4; _tiles(d): 3x48-pixel regions per class (to choose nsplit).  slab: nsplit * classes * 16 * Cin * NcP
 * floats, d->in = x, d->out = dL/dy as for adh_conv_wgrad; adh_wgrad_reduce_wino32 sums the splits (in place), applies
 * A^T (.) A with G's deferred factors and scatters every class's 2x2 taps into layout L (KHt*KWt = d->KH*d->KW).
 * Replaces the weight half of ATen's conv2d / conv_transpose2d backward for the down / up-sampling layers
 * (/root/reference models/dehazing/high_intensity.py:100-118, medium_intensity.py:53,63). */
int adh_conv_wgrad_wino32_groups(const adh_conv_desc* d);
int adh_conv_wgrad_wino32_classes(const adh_conv_desc* d);
int adh_conv_wgrad_wino32_tiles(const adh_conv_desc* d);
/* The class descriptors of ONE transposed layer (n = 2 .. 4 single-class descriptors that differ in out_oy / out_ox and dy0 / dx0
 * only) as one grid: slab[nsplit][n][16][Cin][NcP]; the splits are summed into split 0 by this call, the caller then runs
 * adh_wgrad_reduce_wino32(slab + m * 16 * Cin * NcP, 1, &descs[m], ...) for every class m.  ADH_E_UNSUPPORTED when the shapes are
 * not conv_wgrad32v2_kernel's or the descriptors differ in more: use adh_conv_wgrad_wino32 per descriptor. */
int adh_conv_wgrad_wino32_multi(void* stream, const adh_conv_desc* descs, int n, float* slab, int nsplit);
/* kernel launches per adh_conv_wgrad_wino32 call: 1 when the classes share one grid (then classes * groups workgroup groups per
 * pixel split run together), else the class count */
int adh_conv_wgrad_wino32_launches(const adh_conv_desc* d);
int adh_conv_wgrad_wino32(void* stream, const adh_conv_desc* d, float* slab, int nsplit);
int adh_wgrad_reduce_wino32(void* stream, float* slab /* scratch: split 0 receives the sum */, int nsplit,
                            const adh_conv_desc* d, int KP, int NcP, const adh_wlayout* L, float* dst, int accumulate);

/* Packed small-Cin weight gradient (7x7 stems, Cin <= 8 stored with cstride 8): call adh_conv_wgrad with
 * KW = ceil(kw/4), dstep_x = 4, Cin = 8 -- the 32-wide MFMA row tile then spans 4 adjacent pixels x 8
 * channels -- and unpack slab[s][ky*KWg + kxg][kxl*8 + ci][NcP] into OIHW with this call. */
int adh_wgrad_reduce_packed(void* stream, const float* slab, int nsplit, int NcP, int Cin, int KH, int KW, int Cout,
                            float* dst, int accumulate);

/* ---- BatchNorm2d + activation (base_model.py:15-19,36-41; eps 1e-5, momentum 0.1) ---------- */
/* reduce `nblk` partial rows [2][NcP] -> batch mean / biased var over `count` elements per channel;
 * writes scale = gamma*invstd, shift = beta - mean*scale, save_mean, save_invstd and updates
 * running_mean/var (unbiased var, momentum) when they are non-NULL. */
int adh_bn_finalize(void* stream, const float* partials, int nblk, int NcP, int C, double count,
                    const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var,
                    float* scale, float* shift, float* save_mean, float* save_invstd,
                    int64_t* num_batches_tracked /* optional: BatchNorm2d's int64 counter, incremented by 1 */);
/* Synchronised BatchNorm under data parallelism (SURVEY 8e mode ii; reproduces the single-process reference's
 * nn.BatchNorm2d over the GLOBAL batch, /root/reference models/dehazing/base_model.py:15-16).  adh_bn_partial_sums reduces
 * partials[nblk][2][pitch] (the forward statistics of the conv epilogue, or adh_bn_bwd_reduce's output) to
 * sums[0..C) = sum, sums[C..2C) = second row, sums[2C] = count, in fp64; the host all-reduces the 2C+1 doubles over the ranks;
 * adh_bn_finalize_sums / adh_bn_bwd_finalize_sums are adh_bn_finalize / adh_bn_bwd_finalize working from such sums
 * (backward: d-gamma / d-beta from the LOCAL sums, the input-gradient coefficients from the GLOBAL ones, as
 * torch.nn.SyncBatchNorm does). */
int adh_bn_partial_sums(void* stream, const float* partials, int nblk, int pitch, int C, double count, double* sums);
int adh_bn_finalize_sums(void* stream, const double* sums, int C, const float* gamma, const float* beta, float eps,
                         float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                         float* save_mean, float* save_invstd, int64_t* num_batches_tracked);
int adh_bn_bwd_finalize_sums(void* stream, const double* local_sums, const double* global_sums, int C, const float* gamma,
                             const float* invstd, float* dgamma, float* dbeta, int accumulate, float* coef);
/* eval mode: scale/shift from running statistics */
int adh_bn_fold_eval(void* stream, int C, const float* gamma, const float* beta,
                     const float* running_mean, const float* running_var, float eps,
                     const float* conv_bias, float* scale, float* shift);
/* eval-mode BN backward: mean[c] = beta[c], invstd[c] = 1/gamma[c] (0 where gamma is 0), padded to C4 with zeros, so
 * that adh_bn_bwd_reduce / _finalize on y := the pre-activation output give d-gamma / d-beta of a frozen-statistics
 * BatchNorm (fine-tuning with module.eval(); base_model.py:15-16 under torch autograd) */
int adh_bn_eval_bwd_vectors(void* stream, int C, int C4, const float* gamma, const float* beta, float* mean,
                            float* invstd);
/* out = act(y*scale[c] + shift[c] (+ residual)) over P pixels.  mask_bits (optional, ReLU only, C/4 even): one bit per
 * element, bit (p*C + c) of the byte array = [pre-activation > 0] -- the backward passes then read this instead of `out` */
int adh_bn_apply(void* stream, const float* y, int y_cs, const float* scale, const float* shift,
                 const float* residual, int res_cs, int act, float* out, int out_cs,
                 int64_t P, int C, uint8_t* mask_bits);
/* backward of act(BN(y) (+res)): given g_out and the forward output `out` (relu mask),
 * pass 1: per-block partial sums of g and g*xhat -> partials[nblk][2][C]
 * (nblk = adh_bn_bwd_num_blocks(P)). */
int adh_bn_bwd_num_blocks(int64_t P, int C);
/* mask_ss (optional, act == RELU, no residual): {scale[C], shift[C]} of the forward pass -- the ReLU mask is then
 * recomputed as fma(y, scale, shift) > 0, exactly the forward expression, and `out` is not read (one tensor pass less). */
int adh_bn_bwd_reduce(void* stream, const float* g_out, int g_cs, const float* out, int out_cs, int act,
                      const float* y, int y_cs, const float* mean, const float* invstd,
                      float* partials, int64_t P, int C, const float* mask_ss, const uint8_t* mask_bits);
/* finalize: dgamma = sum(g*xhat), dbeta = sum(g) (accumulated into grads when accumulate!=0) and the
 * per-channel coefficients used by pass 2. coef[3][C] = {gamma*invstd, mean_g, mean_gxhat}. */
int adh_bn_bwd_finalize(void* stream, const float* partials, int nblk, int C, double count,
                        const float* gamma, const float* invstd, float* dgamma, float* dbeta,
                        int accumulate, float* coef);
/* the same from adh_conv_wino43_dgrad_bnred's rows: partials[nblk][2][pitch] = (sum g m, sum g m (y - mean)) */
int adh_bn_bwd_finalize_centered(void* stream, const float* partials, int nblk, int pitch, int C, double count,
                                 const float* gamma, const float* invstd, float* dgamma, float* dbeta,
                                 int accumulate, float* coef);
/* pass 2: g_y = gamma*invstd*(g - mean_g - xhat*mean_gxhat); optionally g_res = g (masked).
 * training==0 (eval BN): g_y = g*scale only (coef row 0). */
int adh_bn_bwd_apply(void* stream, const float* g_out, int g_cs, const float* out, int out_cs, int act,
                     const float* y, int y_cs, const float* mean, const float* invstd, const float* coef,
                     int training, float* g_y, int gy_cs, float* g_res, int gres_cs, int64_t P, int C,
                     const float* mask_ss, const uint8_t* mask_bits);

/* ---- AttentionBlock (base_model.py:43-78) ----------------------------------------------------- */
/* pooled[n][2][C] = (mean, max) over H*W; amax_idx[n][C] = first pixel index attaining the max */
int adh_cbam_pool(void* stream, const float* x, int x_cs, int N, int HW, int C,
                  float* partial, int32_t* partial_idx, int nblk, float* pooled, int32_t* amax_idx);
int adh_cbam_pool_num_blocks(int HW);
/* ca[n][C] = sigmoid(fc(avg) + fc(max)); fc = W2 * relu(W1 * v); W1 [Ch][C], W2 [C][Ch];
 * hidden[n][2][Ch] keeps the post-relu activations for backward. */
int adh_cbam_mlp(void* stream, const float* pooled, const float* w1, const float* w2, int N, int C, int Ch,
                 float* ca, float* hidden);
/* smap[n][hw][2] = (mean_c, max_c) of x*ca; cidx[n][hw] = first channel attaining the max */
int adh_cbam_spatial_stats(void* stream, const float* x, int x_cs, const float* ca, int N, int HW, int C,
                           float* smap, int32_t* cidx);
/* sa = sigmoid(conv7x7(smap)); out = x*ca*sa */
int adh_cbam_apply(void* stream, const float* x, int x_cs, const float* ca, const float* smap, const float* wsp,
                   int N, int H, int W, int C, float* sa, float* out, int out_cs);
/* backward, pass A: gsa_pre[n][hw] = (sum_c g*x*ca) * sa*(1-sa) */
int adh_cbam_bwd_a(void* stream, const float* g, int g_cs, const float* x, int x_cs, const float* ca,
                   const float* sa, int N, int HW, int C, float* gsa_pre);
/* pass B (tiny): gsmap[n][hw][2] = conv7x7^T(gsa_pre); dwsp[2*49] (+)= sum smap (x) gsa_pre */
int adh_cbam_bwd_b(void* stream, const float* gsa_pre, const float* smap, const float* wsp,
                   int N, int H, int W, float* gsmap, float* dwsp_partial, int nblk, float* dwsp, int accumulate);
int adh_cbam_bwd_b_num_blocks(int N, int H, int W);
/* pass C: gx1 = g*sa + gsmap.mean/C + gsmap.max*[c==cidx]; gca_partial[blk][n][C] = sum_hw gx1*x */
int adh_cbam_bwd_c(void* stream, const float* g, int g_cs, const float* x, int x_cs, const float* sa,
                   const float* gsmap, const int32_t* cidx, int N, int HW, int C,
                   float* gca_partial, int nblk);
/* pass D (tiny, three launches): reduce gca, back through sigmoid + MLP: gpool[n][2][C], dW1/dW2 (+)= ;
 * scratch: adh_cbam_bwd_d_scratch_floats(N, C, Ch) floats */
int adh_cbam_bwd_d_scratch_floats(int N, int C, int Ch);
int adh_cbam_bwd_d(void* stream, const float* gca_partial, int nblk, const float* ca, const float* pooled,
                   const float* hidden, const float* w1, const float* w2, int N, int C, int Ch,
                   float* gpool, float* dw1, float* dw2, int accumulate, float* scratch);
/* pass E: gx = gx1*ca + gpool.avg/HW + gpool.max*[hw==amax_idx] */
int adh_cbam_bwd_e(void* stream, const float* g, int g_cs, const float* x, int x_cs, const float* ca,
                   const float* sa, const float* gsmap, const int32_t* cidx, const float* gpool,
                   const int32_t* amax_idx, int N, int HW, int C, float* gx, int gx_cs);

/* ---- boundary layout + branch heads ------------------------------------------------------------ */
/* NCHW [N,3,H,W] image -> NHWC with cstride 8 (channels 3..7 zero) */
int adh_image_to_nhwc8(void* stream, const float* img, int N, int H, int W, float* out);
/* ImageNet-style normalisation fused with the layout change: out[..., c] = (img[c] - mean_c) * inv_std_c
 * (training/loss.py:60-66), and its adjoint g_img[c] = g[..., c] * inv_std_c */
int adh_image_normalize_to_nhwc8(void* stream, const float* img, int N, int H, int W, float m0, float m1, float m2,
                                 float is0, float is1, float is2, float* out);
int adh_image_normalize_bwd(void* stream, const float* g, int g_cs, int N, int H, int W, float is0, float is1, float is2,
                            float* g_img);
/* generic NCHW <-> NHWC (classifier features / tests) */
int adh_nchw_to_nhwc(void* stream, const float* src, int N, int C, int H, int W, float* dst, int dst_cs);
int adh_nhwc_to_nchw(void* stream, const float* src, int src_cs, int N, int C, int H, int W, float* dst);
/* final blend of each branch, output NCHW [N,3,H,W]; r = head conv output (NHWC, r_cs), pre-activation.
 *   mode 0 Lightweight  (low_intensity.py:41-45):  (1-a)*x + a*sigmoid(r)          (a = *alpha)
 *   mode 1 Medium/COrun (medium_intensity.py:117): clamp(x + tanh(r), 0, 1)
 *   mode 2 Complex      (high_intensity.py:135-138): clamp(x + tanh(r)*sigmoid(gd), 0, 1)
 *   mode 3 LowIntensity (low_intensity.py:116):    clamp(x + (sigmoid(r)-0.5)*2, 0, 1)
 *   mode 4 DualBranch   (high_intensity.py:214):   clamp(x + (1-sigmoid(gd))*tanh(r), 0, 1)
 * gd = 1-channel map (NHWC, gd_cs) pre-sigmoid. */
int adh_head_blend(void* stream, int mode, const float* x_nchw, const float* r, int r_cs,
                   const float* gd, int gd_cs, const float* alpha, int N, int H, int W, float* out_nchw);
/* backward of the blend: g_r (NHWC r_cs, channels >=3 zeroed), g_gd (NHWC gd_cs, ch >=1 zeroed),
 * galpha_partial[nblk] (mode 0). */
int adh_head_blend_bwd(void* stream, int mode, const float* g_out_nchw, const float* x_nchw,
                       const float* r, int r_cs, const float* gd, int gd_cs, const float* alpha,
                       int N, int H, int W, float* g_r, float* g_gd, float* galpha_partial, int nblk);
int adh_head_blend_bwd_num_blocks(int N, int H, int W);

/* ---- routing (models/routing.py:41-61,110-127) ----------------------------------------------------- */
/* weights[n][3] = softmax(logits[n]/T); out = sum_i weights[n][i]*branch_i  (NCHW images, per = 3*H*W) */
int adh_softmax3(void* stream, const float* logits, float temperature, int N, float* weights);
/* g_logits[n][3] from the blend's weight gradients: gw[n][i] = sum_blk gw_partial[n][blk][i];
 * g_logits = w * (gw - sum_j gw_j w_j) / T  (softmax backward, routing.py:111) */
int adh_softmax3_bwd(void* stream, const float* weights, const float* gw_partial, int nblk, float temperature, int N,
                     float* g_logits);
int adh_soft_blend(void* stream, const float* weights, const float* o0, const float* o1, const float* o2,
                   int N, int64_t per, float* out);
/* backward: g_oi = w[n][i]*g; gw_partial[n][nblk][3] = sum g*o_i */
int adh_soft_blend_bwd(void* stream, const float* weights, const float* g, const float* o0, const float* o1,
                       const float* o2, int N, int64_t per, float* g0, float* g1, float* g2,
                       float* gw_partial, int nblk);
/* idx[n] = argmax_i logits[n][i] (first index on ties), int64 */
int adh_argmax3(void* stream, const float* logits, int N, int64_t* idx);
/* device-side compaction for HardRouter: for class c, sel[0..count) = batch indices with idx==c,
 * counts[c] written (no host sync); gather/scatter of whole images by index list */
int adh_route_compact(void* stream, const int64_t* idx, int N, int32_t* sel /*[3][N]*/, int32_t* counts /*[3]*/);
int adh_gather_images(void* stream, const float* src, const int32_t* sel, int count, int64_t per, float* dst);
int adh_scatter_images(void* stream, const float* src, const int32_t* sel, int count, int64_t per, float* dst);

/* ---- losses (training/loss.py:138,81,200) ----------------------------------------------------------- */
int adh_reduce_num_blocks(int64_t n);
/* partial[blk] = sum |a-b| ; finalize with adh_sum_partials (scaled by 1/n) */
int adh_l1_partial(void* stream, const float* a, const float* b, int64_t n, float* partial);
int adh_mse_partial(void* stream, const float* a, const float* b, int64_t n, float* partial);
int adh_sum_partials(void* stream, const float* partial, int nblk, double scale, float* out);
/* g_a = gscale * sign(a-b) * (*upstream)   (upstream: device scalar or NULL = 1) */
int adh_l1_bwd(void* stream, const float* a, const float* b, int64_t n, float gscale, const float* upstream, float* g_a);
int adh_mse_bwd(void* stream, const float* a, const float* b, int64_t n, float gscale, const float* upstream, float* g_a);
/* cross entropy over 3 classes, mean reduction: loss scalar + dlogits[n][3] (scaled by 1/N) */
int adh_cross_entropy3(void* stream, const float* logits, const int64_t* labels, int N, float* loss, float* dlogits);

/* ---- LPIPS (lpips.LPIPS(net='alex'), training/loss.py:86-108) --------------------------------------- */
/* Fused input stage: v = img*a_c + b_c (the [-1,1] rescale and lpips' ScalingLayer), zero padded by 2 and
 * rearranged space-to-depth(4): out[n][r][q][(by*4+bx)*3 + c] = v[c][4r+by-2][4q+bx-2], out is
 * [N][OH+2][OW+2][48] with OH = (H+4-11)/4+1 -- AlexNet's 11x11 stride-4 conv becomes a 3x3 stride-1 conv. */
int adh_lpips_s2d(void* stream, const float* img, int N, int H, int W, const float* a3, const float* b3, int OHp, int OWp,
                  float* out);
int adh_lpips_s2d_bwd(void* stream, const float* g, int N, int H, int W, const float* a3, int OHp, int OWp, float* g_img);
/* one LPIPS tap: val[n] += (1/HW) * sum_p sum_c w[c] * (a/(|a|+eps) - b/(|b|+eps))^2 ; partial[n][nblk] */
int adh_lpips_layer(void* stream, const float* fa, const float* fb, const float* w, int N, int HW, int C, float* partial,
                    int nblk);
int adh_lpips_layer_num_blocks(int HW);
/* gradient wrt fa given the upstream per-image cotangent g_val[n] (already including any scalar factor) */
int adh_lpips_layer_bwd(void* stream, const float* fa, const float* fb, const float* w, const float* g_val, int N, int HW,
                        int C, float* g_fa);
/* out[n] = sum_b partial[n][b] * scale (+ out[n] if accumulate) */
int adh_rows_sum(void* stream, const float* partial, int N, int nblk, float scale, float* out, int accumulate);

/* ---- optimiser (training/train_joint.py:86-90: Adam, weight_decay 1e-4) ------------------------------ */
/* `repeats` consecutive Adam updates with the same gradient and shared state (the reference lists every
 * branch parameter twice, train_joint.py:81-84); step = count before the call. */
int adh_adam_step(void* stream, float* p, const float* g, float* m, float* v, int64_t n, int step,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int repeats);

/* ---- misc elementwise ---------------------------------------------------------------------------------- */
int adh_add_inplace(void* stream, float* dst, const float* src, int64_t n);            /* dst += src */
int adh_axpby_strided(void* stream, float* dst, int dst_cs, const float* src, int src_cs,
                      int64_t P, int C, float a, float b);                           /* dst = a*dst + b*src */
/* MaxPool2d(k, stride, pad) with -inf padding; idx[n][oy][ox][c] = winning input pixel (first on ties) */
int adh_maxpool(void* stream, const float* x, int x_cs, int N, int H, int W, int C, int k, int stride, int pad,
                float* out, int out_cs, int32_t* idx);
int adh_maxpool_bwd(void* stream, const float* g, int g_cs, const int32_t* idx, int N, int OH, int OW, int C, int k,
                    int stride, int pad, int H, int W, float* gx, int gx_cs);
int adh_mul(void* stream, float* dst, const float* a, const float* b, int64_t n);       /* dst = a*b */
/* AvgPool2d(k, stride k) forward (torchvision densenet transition) */
int adh_avgpool(void* stream, const float* x, int x_cs, int N, int H, int W, int C, int k, float* out, int out_cs);
/* adjoint of the global average pool: gx[n][p][c] = g[n][c] / HW (torchvision resnet avgpool) */
int adh_global_avgpool_bwd(void* stream, const float* g, int N, int HW, int C, float* gx, int gx_cs);
/* bilinear resize, align_corners 0/1 (medium_intensity.py:93-99,147,152) and its adjoint */
int adh_bilinear(void* stream, const float* x, int x_cs, int N, int H, int W, int C, int OH, int OW,
                 int align_corners, float* out, int out_cs);
int adh_bilinear_bwd(void* stream, const float* g, int g_cs, int N, int H, int W, int C, int OH, int OW,
                     int align_corners, float* gx, int gx_cs);

/* ---- training loop / evaluation harness around the hot path (csrc/train_io.hip) ------------------------------ */
/* Multi-tensor Adam: ONE launch updates every listed tensor (training/train_joint.py:86-90,153-154;
 * training/train_dehazing.py:52-57,95-96 call torch.optim.Adam.step, one ATen op group per tensor).
 * table_dev[i] describes tensor i (device pointers; `step` = updates taken when the table was uploaded; `repeats` = how
 * many times the reference lists the parameter, train_joint.py:81-84; the table may stay resident across calls:
 * `calls_since_upload` calls have advanced every tensor by `repeats` each since then, so the host uploads it only when a
 * pointer or the live set changes -- no per-step host-to-device copy); chunks_dev holds nchunks pairs (tensor index, chunk
 * index), a chunk being adh_adam_chunk_elems() consecutive floats.  g is read as g*grad_scale (1/world for summed
 * data-parallel gradients).  dup_mode 0: `repeats` consecutive full updates (torch's single-tensor loop: the CPU path
 * and every torch < 2.0); 1: torch >= 2.0's foreach form on CUDA (see train_io.hip).  max_repeats <= 4. */
typedef struct adh_adam_tensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
    int32_t step;
    int32_t repeats;
} adh_adam_tensor;
int adh_adam_chunk_elems(void);
int adh_adam_multi(void* stream, const adh_adam_tensor* table_dev, const int32_t* chunks_dev, int nchunks, float lr,
                   float beta1, float beta2, float eps, float weight_decay, float grad_scale, int dup_mode,
                   int max_repeats, int calls_since_upload);
/* ---- detector stage (csrc/detect.hip): what torchvision's Faster R-CNN -- the detector
 * /root/reference models/detection.py:23-29 instantiates, consumed by evaluation/evaluate.py:288-344 -- does between its
 * convolutions.  torchvision is a third-party dependency absent from /root/reference: the algorithms follow its published
 * source (models/detection/{rpn,roi_heads,_utils,anchor_utils}.py, ops/{poolers,roi_align,boxes}.py). --------------------- */
/* FPN top-down merge: lateral[n,y,x,:] += top[n, nearest(y), nearest(x), :] (F.interpolate(mode="nearest") + add). */
int adh_upsample_nearest_add(void* stream, const float* top, int top_cs, int th, int tw, float* lateral, int lat_cs, int N,
                             int H, int W, int C);
/* RPN level: anchors (base_anchors[A][4] shifted by the strides, position major / anchor minor), BoxCoder(1,1,1,1).decode,
 * clip to the image.  cls [N,H,W,>=A], reg [N,H,W,>=4A] -> boxes [N,H*W*A,4], logits [N,H*W*A]. */
int adh_rpn_decode(void* stream, const float* cls, int cls_cs, const float* reg, int reg_cs, int N, int H, int W, int A,
                   int stride_h, int stride_w, const float* base_anchors, float img_h, float img_w, float* boxes, float* logits);
/* Grouped NMS (torchvision.ops.batched_nms): boxes [M,4] sorted by descending score, group [M]; keep[i] = 1 if box i survives.
 * mask_workspace: M * adh_nms_words(M) uint64.  M <= 16384. */
int adh_nms_words(int M);
int adh_nms_sorted(void* stream, const float* boxes, const int32_t* group, int M, float iou_threshold, void* mask_workspace,
                   int32_t* keep);
/* MultiScaleRoIAlign (7x7, sampling_ratio 2, aligned = False) over up to four NHWC pyramid levels; rois [R,5] = (image, x1, y1,
 * x2, y2); out [R][C*49] channel-major (torch's flatten(1) of [R,C,7,7]). */
typedef struct adh_fpn_levels {
    const float* f[4];
    int32_t H[4], W[4], cs[4];
    float scale[4];
    int32_t nlevels;
} adh_fpn_levels;
int adh_roi_align_fpn(void* stream, const adh_fpn_levels* levels, const float* rois, int R, int C, float* out);
/* Box head post-processing up to the NMS: softmax, BoxCoder(10,10,5,5).decode per foreground class, clip, validity
 * (score > thresh, w and h >= min_size).  -> boxes [R,NC-1,4], scores [R,NC-1], valid [R,NC-1]. */
int adh_box_postprocess(void* stream, const float* logits, int l_cs, const float* deltas, int d_cs, const float* props,
                        const float* img_hw, const int32_t* img, int R, int NC, float score_thresh, float min_size, float* boxes,
                        float* scores, int32_t* valid);
/* Paired augmentation on NCHW float images in [0,1] (/root/reference data/dataset.py:59-64,100-116: RandomHorizontalFlip,
 * RandomVerticalFlip, ColorJitter(brightness 0.1, contrast 0.1) with one seed shared by hazy / clear / dehazed).
 * params[n] = {flip_h, flip_v, brightness_first, b, c} (5 floats per image, drawn on the host in torchvision's order);
 * brightness = clamp(b x), contrast = clamp(c x + (1 - c) mean(gray)), gray = 0.2989 R + 0.587 G + 0.114 B.
 * partial: N * adh_augment_num_blocks(H * W) doubles of workspace.  out != x. */
int adh_augment_num_blocks(int64_t HW);
int adh_paired_augment(void* stream, const float* x_nchw, const float* params, int N, int H, int W, double* partial, int nblk,
                       float* out_nchw);
/* Synthetic fog on NCHW images, per-image beta / airlight: hazy = clip(clear*t + A*(1-t), 0, 1),
 * t = exp(-beta*(0.3 + 0.7*sqrt((x-0.5)^2 + (y-0.2)^2))) on the unit grid (utils/helpers.py:241-258, transmission in
 * float64 as numpy evaluates it). */
int adh_apply_fog(void* stream, const float* clear_nchw, const float* beta, const float* airlight, int N, int H, int W,
                  float* hazy_nchw);
/* Per-image PSNR over per_image = 3*H*W elements (skimage peak_signal_noise_ratio as called at
 * evaluation/metrics.py:27 and training/train_joint.py:217); partial: double[N * adh_psnr_num_blocks(per_image)];
 * mse (optional) and psnr: float[N]. */
int adh_psnr_num_blocks(int64_t per_image);
int adh_psnr(void* stream, const float* pred, const float* target, int N, int64_t per_image, float data_range,
             double* partial, int nblk, float* mse, float* psnr);
/* Per-image SSIM of the channel-mean grayscale images with skimage's defaults (7x7 uniform window, K1 .01, K2 .03,
 * sample covariance, mean over the image cropped by 3 px) as called at evaluation/metrics.py:29-32 and
 * training/train_joint.py:221-224; partial: double[N * adh_ssim_num_blocks(H, W)]; ssim: float[N].
 * ADH_E_UNSUPPORTED when H or W < 7 (skimage raises there too). */
int adh_ssim_num_blocks(int H, int W);
int adh_ssim_gray(void* stream, const float* pred_nchw, const float* target_nchw, int N, int H, int W, float data_range,
                  double* partial, int nblk, float* ssim);

#ifdef __cplusplus
}
#endif
#endif /* ADAM_DEHAZE_HIP_H */
