/*
 * hnd_hip.h -- C ABI of libhnd_hip.so: the MI355X (gfx950) kernels behind the HND/GHND
 * head-network-distillation step.
 *
 * The reference (yoshitomo-matsubara/hnd-ghnd-object-detectors) has NO native/FFI layer: every
 * FLOP of its hot path is a torch 1.3.1 / torchvision 0.4.2 nn.Module call made from
 *   src/mimic_runner.py:38-59      distill_model (zero_grad / backward / step)
 *   src/distillation/tool.py:40-61 DistillationBox.forward
 *   src/distillation/loss.py:25-34 GeneralizedCustomLoss.forward
 *   src/models/org/rcnn.py:65-110  CustomRCNNTransform.forward, CustomRCNN.forward
 *   src/models/custom/resnet.py:26-30,95-105 ; src/models/mimic/resnet_layer.py:40-70
 * so each entry point below names the reference call (file:line) whose arithmetic it replaces.
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C: pointers are DEVICE pointers into caller-owned buffers (torch-ROCm tensors in the
 *     Python host); the library allocates nothing and keeps no pointer after return.
 *   - activations are NHWC fp32 (physical layout of a torch channels_last tensor).
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it.
 *   - return 0 on success, a negative hnd_status otherwise; hnd_last_error_string() explains.
 *     Nothing throws across the ABI and nothing calls exit().
 */
#ifndef HND_HIP_H
#define HND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HND_ABI_VERSION 12

typedef enum hnd_status {
  HND_OK = 0,
  HND_ERR_INVALID = -1,   /* bad argument / unsupported geometry */
  HND_ERR_LAUNCH = -2,    /* HIP reported an error at launch */
  HND_ERR_ASYNC = -3      /* a previously enqueued kernel failed (see hnd_sync_check) */
} hnd_status;

const char* hnd_last_error_string(void);
int hnd_abi_version(void);
/* hipStreamSynchronize + hipGetLastError; surfaces asynchronous faults. */
int hnd_sync_check(void* stream);
/* Non-zero once a B-streamed GEMM (hnd_conv2d_igemm with relay_ws) gave up waiting for a partial tile; from then on
 * hnd_conv2d_igemm launches of that kernel return HND_ERR_LAUNCH and hnd_sync_check returns HND_ERR_ASYNC until the
 * caller acknowledges with reset = 1 (the workspaces themselves need no repair: their flags carry launch epochs). */
int hnd_relay_timeouts(int reset);
/* name of the device the library sees ("gfx950..."), or "" */
const char* hnd_device_arch(void);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution, fp32 MFMA (v_mfma_f32_16x16x4_f32 in every kernel since round 2), NHWC.
 * One kernel serves forward convs and data-gradients: the launch describes, per output pixel
 * (oh, ow) of a *virtual* output grid and per tap (i, j), the source pixel
 *      ih = oh*sh + i*dh + bh,   iw = ow*sw + j*dw + bw        (zero when out of range)
 * and where the virtual pixel lands in the real output tensor
 *      yh_idx = oh*y_sh + y_oh,  yw_idx = ow*y_sw + y_ow.
 *   forward conv (stride s, pad p):           sh=s dh=+1 bh=-p,  y_s=1 y_o=0
 *   dgrad of a stride-1 conv:                 sh=1 dh=-1 bh=+p   (weights packed transposed)
 *   dgrad of a stride-2 conv: four launches, one per output parity, each with its tap subset.
 * Replaces nn.Conv2d forward / autograd conv backward(input) reached from
 * src/models/custom/resnet.py:96, src/models/mimic/resnet_layer.py:43-62, torchvision Bottleneck
 * and FeaturePyramidNetwork (src/models/org/rcnn.py:391-414), autograd at src/mimic_runner.py:53.
 *
 * Fused prologue (on the gathered input, per input channel; padding taps stay exactly 0):
 *      a = x*pro_scale[c] + pro_shift[c]; if (pro_relu) a = max(a, 0)
 *   = train-mode BatchNorm2d(+ReLU) of resnet_layer.py:44-64 applied on load, or the
 *     FrozenBatchNorm2d scale of a frozen layer during dgrad.
 * Fused epilogue (per output channel c / element):
 *      v = acc; v = v*epi_scale[c] + epi_shift[c]      FrozenBatchNorm2d (no eps) or conv bias
 *      v += res1 (+ nearest-upsampled when res1_mode=1: FPN top-down path); v += res2
 *      if (mask)  v = mask>0 ? v : 0                   ReLU backward of a frozen block
 *      if (relu)  v = max(v, 0)
 *      store; if (stats) per-channel partial (sum v, sum v^2) per 128-row tile -> stats[tile][2][cout]
 *   stats feed hnd_bn_finalize (train-mode BN batch statistics without a second pass).
 * ------------------------------------------------------------------------------------------ */
typedef struct hnd_conv_desc {
  const float* x;          /* input  [n][h][w][cin]   (cin multiple of 32, or exactly 4)          */
  const float* w;          /* packed weights [round_up(cout,64)][kdim], K contiguous (hnd_pack_weights) */
  float* y;                /* output [n][yh][yw][ldc]                                              */
  const float* pro_scale;  /* [cin] or NULL                                                        */
  const float* pro_shift;  /* [cin]; required whenever pro_scale is given                           */
  const float* epi_scale;  /* [cout] or NULL                                                       */
  const float* epi_shift;  /* [cout] or NULL                                                       */
  const float* res1;       /* same geometry as y (mode 0) or [n][res1_h][res1_w][ldc] (mode 1), or NULL */
  const float* res2;       /* same geometry as y, or NULL                                          */
  const float* mask;       /* same geometry as y, or NULL                                          */
  float* stats;            /* [ceil(M/128)][2][cout] or NULL, M = n*oh*ow                          */
  int32_t n, h, w_, cin;
  int32_t oh, ow;          /* virtual output grid of this launch                                   */
  int32_t yh, yw, cout, ldc;
  int32_t y_sh, y_oh, y_sw, y_ow;
  int32_t kh, kw;
  int32_t sh, dh, bh, sw, dw, bw;
  int32_t kdim;            /* row stride of w: kh*kw*cin rounded up to a multiple of 32            */
  int32_t pro_relu, relu;
  int32_t res1_mode, res1_h, res1_w;
  /* grouped weights (0 = off): output rows [g*w_group_rows, (g+1)*w_group_rows) use the packed matrix at
   * w + g*w_group_stride floats; w_group_rows must be a multiple of 128.  Used by the Winograd path: the 16
   * transform components are 16 GEMMs over disjoint row groups of one launch. */
  int32_t w_group_rows, w_group_stride;
  /* Work-balancing workspace of the B-streamed kernel (csrc/conv_bstream.hip), or NULL.  hnd_conv2d_igemm_workspace()
   * bytes, zero-filled ONCE by the caller before the first launch (the kernel keeps a launch counter in it and never
   * needs it cleared again), owned by this launch: two launches that may run concurrently must not share it.  With it,
   * a workgroup whose share of the launch ends inside a tile parks the tile's accumulators here and its neighbour
   * continues the same k chain (same bits as without).  The neighbour's wait is bounded: if it gives up, the launch's
   * output is invalid and a sticky process-wide error is raised -- see hnd_relay_timeouts. */
  float* relay_ws;
  /* ReLU masks as NIBBLES (ABI 8): one byte per (pixel, group of 4 channels) of a tensor of the output's geometry, bit k
   * of byte (pixel * ldc + c) / 4 = [value of channel (c & ~3) + k > 0].  The backward pass of a FROZEN layer needs only
   * the sign of its activations, and the data gradients that apply them are HBM-bound (out + residual + mask at K = 128):
   * 1/16 of the fp32 mask's bytes.  ldc % 4 == 0 required.
   *   mask_bits: consumed like `mask` (v = bit ? v : 0); give one or the other, not both.
   *   mask_out:  produced from the STORED value (after ReLU), next to y. */
  const uint8_t* mask_bits;
  uint8_t* mask_out;
  /* BatchNorm-backward sums from the epilogue of the DATA gradient that produces g (ABI 9), or bwd_x = NULL.  With
   * bwd_x (the raw input of a train-mode BatchNorm(+ReLU), same geometry and ldc as y) the launch writes into `stats`,
   * instead of sum / sum of squares of y, the partials hnd_bn_bwd_reduce would make from the stored y:
   *   stats[tile][0][ch] = sum d,  stats[tile][1][ch] = sum d * (x - mean) * rstd,
   *   d = bwd_relu ? [x * scale + shift > 0] y : y      (per 128-pixel tile, ready for hnd_bn_bwd_finalize).
   * Plain epilogues only (no residual, no mask, no ReLU); taken by the tiled kernels (the persistent ones leave any
   * launch with `stats` to them).  bwd_scale / shift / mean / rstd: ldc floats each. */
  const float* bwd_x;
  const float* bwd_scale;
  const float* bwd_shift;
  const float* bwd_mean;
  const float* bwd_rstd;
  int bwd_relu;
  /* fp32 EMULATED on the bf16 matrix pipe (csrc/conv_bx3.hip; ABI 10 opt-in, the Python host's default since ABI 12): the
   * weight operand pre-split into three bf16 planes by hnd_pack_bf16x3, or NULL = native fp32 MFMA.  Attaching the image ASKS
   * for the emulation: every launch it can take (tap-free K = 128 or 256 P <= 2048; scale / shift / residual / ReLU /
   * mask-nibble epilogue; dense output; any row count) then computes each product as six bf16 MFMAs of the operands' exact
   * 8+8+8-bit planes with fp32 accumulation -- fp32-level accuracy (rel-L2 2.4e-7 vs fp64, native 2.9e-7), a different
   * summation from the fp32 kernels' (not their bits), 1.4-1.7x their rate.  Attach it by LAYER (hnd_bf16x3_recommended),
   * never by batch, so that an image's maps do not depend on the batch it is computed in.
   * DEVIATIONS from fp32 arithmetic, both outside a healthy training step (tests/test_bx3_gpu.py):
   *   non-finite: an Inf or NaN in an A row (or in a weight row) makes EVERY output of that row (that channel) NaN -- never
   *     a finite value; native fp32 gives +-Inf where no Inf - Inf / 0 * Inf occurs.  (Inf splits into Inf + NaN + NaN.)
   *   denormal / tiny: the bf16 pipe flushes denormal plane values.  An operand element |x| < 2^-126 counts as 0; for
   *     2^-126 <= |x| < 2^-110 its mid / lo planes (|.| < 2^-8 |x|, 2^-16 |x|) may flush, i.e. x is used with relative error
   *     <= 2^-8: absolute error per product <= 2^-118 |b|.  Elements >= 2^-110 (7.7e-34) are exact.  (Upper bounds; measured
   *     on gfx950, worst |y - exact| / sum |a||b|: 9e-6 in the middle range, 3.4e-2 for denormal inputs, 3e-7 above 2^-110;
   *     native fp32 MFMA 4e-7 ... 5e-6.) */
  const uint16_t* w_bf16x3;
  /* ABI 12: the STREAM image of the same three planes (hnd_pack_bf16x3s), or NULL.  Attached, it asks for the B-streamed
   * emulation kernel (csrc/conv_bxs.hip) on launches the B-resident one above does not take: convolutions over taps (cin %
   * 64 == 0), any K % 128 == 0, strided outputs, every prologue / epilogue operand of this descriptor including `stats`
   * and `bwd_x`.  Same arithmetic and deviations as w_bf16x3; the relay workspace works as for the native B-streamed kernel. */
  const uint16_t* w_bf16x3s;
} hnd_conv_desc;

int hnd_conv2d_igemm(const hnd_conv_desc* desc, void* stream);
/* bytes of relay_ws this launch can use (0: none -- the field is ignored) */
size_t hnd_conv2d_igemm_workspace(const hnd_conv_desc* desc);
/* which block tile the launch above would use: 0 = 128x128, 1 = 128x64, 2 = 64x128, 3 = 64x64 (pixels x channels),
 * 4 = the vector-ALU kernel for cout <= 4 (one thread per pixel; bit-identical to the MFMA tiles),
 * 5 / 6 = the B-resident persistent GEMM (csrc/conv_bres.hip) with a 128- / 64-column weight slice held in LDS: 1x1
 * taps, K <= 256 / 512, no statistics; bit-identical to the tiled kernel.  HND_BRES=0 in the environment turns it off;
 * 7 / 8 = its one-wave-per-SIMD build (plain epilogues); 9 = the 7x7 stride-2 stem from an LDS-staged input patch
 * (csrc/conv_stem.hip; bit-identical to the generic 4-channel-input kernel, HND_STEM7=0 turns it off); 10 = unused
 * (a K = 1024 build of the one-wave kernel, removed in round 4); 11 / 12 = the B-streamed persistent GEMM for long K
 * (csrc/conv_bstream.hip: 128 x 128 / 256 x 64 block tile, the weight slice streams through three LDS stages, A
 * fragments straight from global memory; K % 128 == 0, K >= 1024 or taps; bit-identical to the tiled kernel;
 * HND_BSTREAM=0 turns it off). */
int hnd_conv2d_igemm_tile(const hnd_conv_desc* desc);      /* (13 = the bf16x3 emulation kernel, csrc/conv_bx3.hip; 14 / 15 =
                                                             * its B-streamed build, 128 x 128 / 256 x 64 tile, csrc/conv_bxs.hip) */
/* ABI 12: should a layer's launches carry a weight image (run on the emulation)?  Decided from the LAYER alone: GEMM rows ONE
 * image contributes (oh * ow of a 1x1 conv; components x tiles of a Winograd launch), depth and output channels -- priced
 * at 16 images per GPU, never at the launch's own batch.  1 = attach hnd_conv_desc.w_bf16x3, 0 = leave it NULL. */
int hnd_bf16x3_recommended(int64_t rows_per_image, int kdim, int cout);
/* The three-plane bf16 image of a packed weight operand for hnd_conv_desc.w_bf16x3: w_packed [groups][rows_pad][kdim]
 * fp32 (hnd_pack_weights / hnd_wino*_weights; group g at w_packed + g * group_stride floats) -> img: per (group, 64-row
 * slice) [3 planes hi / mid / lo][64 rows][kdim] bf16, 16-byte chunk c of row r stored at position c ^ (r & 15) (the
 * kernel's conflict-free LDS image, copied linearly).  x == hi + mid + lo exactly (truncation split).  kdim 128 / 256, or 512
 * as two k parts of 256 (part p at p * groups * (rows_pad / 64) slices; the launch then runs two passes).
 * hnd_pack_bf16x3_elems: uint16 elements of img. */
size_t hnd_pack_bf16x3_elems(int rows_pad, int kdim, int groups);
int hnd_pack_bf16x3(const float* w_packed, uint16_t* img, int rows_pad, int kdim, int groups, int64_t group_stride,
                    void* stream);

/* The STREAM image for hnd_conv_desc.w_bf16x3s (ABI 12): per (group, 64-row slice, 64-k stage) three planes [64 rows][64 k]
 * of bf16, 16-byte chunk c of row r at position c ^ ((r >> 1) & 7); any packed operand with kdim % 128 == 0 (taps included). */
int hnd_bf16x3s_recommended(int64_t rows_per_image, int kdim, int cout, int taps);     /* by layer, like hnd_bf16x3_recommended */
size_t hnd_pack_bf16x3s_elems(int rows_pad, int kdim, int groups);
int hnd_pack_bf16x3s(const float* w_packed, uint16_t* img, int rows_pad, int kdim, int groups, int64_t group_stride,
                     void* stream);

/* Weight gradient (autograd conv backward(weight), src/mimic_runner.py:53) of the trainable convs:
 * student stem conv1 (custom/resnet.py:26) and the eight 2x2 convs (resnet_layer.py:43-62).
 *      dW[co][ci][i][j] = sum_m dy[m][co] * a(m; i, j, ci),  a = prologue(x) as in the forward.
 * Split-K implicit GEMM (K = n*oh*ow) on fp32 MFMA into `splitk` partial slabs, then a
 * fixed-order reduce that writes the torch OIHW layout into `dw` (deterministic, no atomics). */
typedef struct hnd_wgrad_desc {
  const float* x;          /* conv input [n][h][w][cin] (cin multiple of 32, or 4 = padded 3)     */
  const float* dy;         /* grad of conv output [n][oh][ow][cout]                                */
  float* dw;               /* [cout][cin_real][kh][kw] (torch layout)                              */
  float* slabs;            /* workspace, hnd_conv2d_wgrad_workspace() bytes                        */
  const float* pro_scale;
  const float* pro_shift;
  int32_t n, h, w_, cin, cin_real;
  int32_t oh, ow, cout, ldy;    /* ldy = channel stride of dy (>= cout; 4 for the 3-channel bottleneck) */
  int32_t kh, kw, stride, pad;
  int32_t pro_relu;
  int32_t splitk;          /* 0 = choose */
  /* batched launch (0 or 1 = single): `groups` independent problems of this geometry whose operands lie
   * x_group_stride / dy_group_stride / dw_group_stride floats apart (the 25 component GEMMs of the Winograd-domain
   * weight gradient); the workspace holds `groups` consecutive slab sets. */
  int32_t groups;
  int64_t x_group_stride, dy_group_stride, dw_group_stride;
} hnd_wgrad_desc;

size_t hnd_conv2d_wgrad_workspace(const hnd_wgrad_desc* desc);
int hnd_conv2d_wgrad(const hnd_wgrad_desc* desc, void* stream);
/* which kernel the launch above would use: 0 = the LDS-staged split-K kernel (csrc/conv_wgrad.hip), 1 = the 7x7 stem from
 * an LDS patch (csrc/conv_stem.hip), 2 = the vector-ALU kernel of the two 3-channel gradients, 3 = the ring kernel
 * (csrc/conv_wgrad_ring.hip: cout % 128 == 0, cin % 64 == 0; both operands straight from global memory through a
 * counted register ring, accumulators resident; by default the 1x1 / grouped Winograd-domain form only;
 * HND_WGRAD_RING=0 turns it off, =2 also takes the tap form).  All sum their split-K slabs in a fixed
 * order: each is bitwise reproducible; different kernels differ in rounding. */
int hnd_conv2d_wgrad_variant(const hnd_wgrad_desc* desc);

/* Re-layout torch OIHW weights [cout][cin][kh][kw] into the K-contiguous GEMM operand of
 * hnd_conv2d_igemm.  transposed=0: rows=cout, K=(tap, cin_pad) (forward);  transposed=1: rows=cin,
 * K=(tap, cout_pad) (dgrad).  Taps are the sub-grid i = i0 + a*istep (a < ni), j likewise, in
 * (a, b) order, so stride-2 dgrad parity classes get their own operand.  Rows are padded to a
 * multiple of 64 and K to a multiple of 32 with zeros; dst must hold rows_pad*kdim floats. */
int hnd_pack_weights(const float* src, float* dst, int cout, int cin, int kh, int kw, int transposed,
                     int chan_pad, int i0, int istep, int ni, int j0, int jstep, int nj, void* stream);

/* Several operands in one launch (same layouts as hnd_pack_weights): the trainable head re-packs its forward,
 * transposed and tap-subset operands after every optimizer step. */
typedef struct hnd_pack_desc {
  const float* src;
  float* dst;
  int32_t cout, cin, kh, kw, transposed, chan_pad, i0, istep, ni, j0, jstep, nj;
} hnd_pack_desc;
int hnd_pack_weights_batched(const hnd_pack_desc* ops, int count, void* stream);

/* packed[r][tap*chan_pad + c] *= scale[c] for c < nscale, all rows_pad rows and ntaps taps of an operand made by
 * hnd_pack_weights: folds a per-channel scale of the GEMM's K operand into the weights.  Used for the data gradient
 * through a frozen conv + FrozenBatchNorm2d: W^T (dy * s) == (W^T diag(s)) dy, so the launch needs no prologue. */
int hnd_scale_packed_k(float* packed, int rows_pad, int kdim, int ntaps, int chan_pad, const float* scale, int nscale,
                       void* stream);

/* FrozenBatchNorm2d (torchvision 0.4.2 ops/misc.py, built at src/models/org/rcnn.py:391,394):
 *   scale = weight * rsqrt(running_var + eps),  shift = bias - running_mean*scale,
 * eps = 0 for FrozenBatchNorm2d (0.4.2 has none); eps = 1e-5 folds an eval-mode nn.BatchNorm2d.
 * scale/shift have cs >= c entries, pad entries are zeroed. */
int hnd_fbn_fold(const float* weight, const float* bias, const float* mean, const float* var,
                 float* scale, float* shift, int c, int cs, float eps, void* stream);

/* CustomRCNNTransform.forward for one image (src/models/org/rcnn.py:65-82 + torchvision
 * GeneralizedRCNNTransform.normalize / batch_images): (x-mean)/std, bilinear resize
 * (align_corners=False, scale as passed to F.interpolate) to out_h x out_w, written into image
 * `index` of the zero-padded NHWC4 batch [n][hp][wp][4] (4th channel 0). src is CHW fp32. */
int hnd_transform_image(const float* src, int h, int w, float* dst, int index, int out_h, int out_w,
                        int hp, int wp, float scale_h, float scale_w, const float mean[3], const float std[3],
                        void* stream);

/* On-device input pipeline: the same transform fed by the DECODED image, uint8, HWC
 * ([h][w][3], as PIL / numpy hand it over; hwc != 0) or CHW ([3][h][w]): fuses structure.transformer.ToTensor
 * (src/structure/transformer.py:52-55, value / 255), RandomHorizontalFlip (:32-49, when flip != 0) and the
 * normalise / resize / pad of hnd_transform_image into one pass -- a quarter of the host->device bytes and no
 * float image on the host. */
int hnd_transform_image_u8(const uint8_t* src, int h, int w, int hwc, int flip, float* dst, int index, int out_h,
                           int out_w, int hp, int wp, float scale_h, float scale_w, const float mean[3],
                           const float std[3], void* stream);

/* The same transform for a whole batch in ONE launch: image i of `imgs` -> image i of dst [count][hp][wp][4].
 * src is float [3][h][w] (is_u8 = 0) or uint8 [h][w][3] / [3][h][w] (is_u8 = 1, hwc, flip as above). */
typedef struct hnd_image_desc {
  const void* src;
  int32_t h, w, out_h, out_w;
  int32_t is_u8, hwc, flip;
  float scale_h, scale_w;      /* source step per output pixel = 1 / resize scale */
} hnd_image_desc;
int hnd_transform_images(const hnd_image_desc* imgs, int count, float* dst, int hp, int wp, const float mean[3],
                         const float std[3], void* stream);

/* Ground-truth boxes of the batch rescaled with the images (src/models/org/rcnn.py:50-53 resize_boxes: x * (new_w /
 * old_w), y * (new_h / old_h) in fp32), every image of the list in ONE launch: dst[j] = src[j] * (j odd ? scale_h :
 * scale_w) for the 4*k floats of each image.  dst may equal src. */
typedef struct hnd_boxes_desc {
  const float* src;
  float* dst;
  int32_t k;
  float scale_w, scale_h;
} hnd_boxes_desc;
int hnd_scale_boxes(const hnd_boxes_desc* items, int count, void* stream);

/* ---- Winograd F(tile x tile, 3x3), tile = 2, 4 or 6, for stride-1 pad-1 3x3 convolutions (torchvision Bottleneck.conv2
 * and the FPN output convs built at src/models/org/rcnn.py:391-414; forward and data gradient) ----------------------
 * y = out_transform( GEMM_f( in_transform(x), U_f ) ), f = 0..(tile+2)^2-1, with all GEMMs issued as ONE
 * hnd_conv2d_igemm launch (1x1 conv over ncomp*tiles_pad "pixels", w_group_rows = tiles_pad, w_group_stride =
 * rows_pad*depth).  tile 2: 16 products per 2x2 outputs (2.25x fewer multiplies than direct, transformed tensors 4x);
 * tile 4: 36 per 4x4 outputs (4x fewer, tensors 2.25x, points 0, +-1, +-2, inf: ~1e-5 relative fp32 error);
 * tile 6: 64 per 6x6 outputs (5.06x fewer, tensors 1.78x, points 0, +-1, +-2, +-1/2, inf: twice tile 4's error).
 * hnd_wino_tiles_pad: rows per component = n*ceil(h/tile)*ceil(w/tile) rounded up to 128.
 * hnd_wino_weights:   OIHW [cout][cin][3][3] -> u [ncomp][round_up(rows,64)][depth] (forward: rows = cout, depth = cin;
 *                     dgrad != 0: rows = cin, depth = cout, taps flipped -- the transposed convolution).
 * hnd_wino_input:     x [n][h][w][c] (+ optional per-channel prologue scale/shift/relu on in-bounds elements)
 *                     -> v [ncomp][tiles_pad][c].
 * hnd_wino_output:    m [ncomp][tiles_pad][cout] -> y [n][h][w][ldc], epilogue order as hnd_conv2d_igemm:
 *                     scale/shift, + res1, mask (ReLU backward), ReLU.  mask_out (ABI 12; NULL = off; tile 4 / 6, cout ==
 *                     ldc): [stored value > 0] as nibbles beside y, like hnd_conv_desc.mask_out -- the ReLU mask conv3's
 *                     data gradient applies comes out of the transform that writes a2, not a pass of its own. */
int64_t hnd_wino_tiles_pad(int n, int h, int w, int tile);
int hnd_wino_weights(const float* weight, float* u, int cout, int cin, int dgrad, int tile, void* stream);
int hnd_wino_input(const float* x, float* v, int n, int h, int w, int c, const float* pro_scale,
                   const float* pro_shift, int pro_relu, int tile, void* stream);
int hnd_wino_output(const float* m, float* y, int n, int h, int w, int cout, int ldc, const float* epi_scale,
                    const float* epi_shift, const float* res1, const float* mask, int relu, int tile, uint8_t* mask_out,
                    void* stream);

/* ---- Winograd F(tile x tile, 2x2), tile = 4 or 6, for the student head's 2x2 convolutions
 * (src/models/mimic/resnet_layer.py:43-62), forward (padding 1 in the encoder, 0 in the decoder) and data gradient (the
 * same correlation with flipped, transposed weights and padding 1 - pad): (tile+1)^2 = 25 / 49 GEMMs in one grouped
 * hnd_conv2d_igemm launch, 2.56x / 2.94x fewer multiplies than direct, transformed tensors 1.56x / 1.36x the activation
 * (points 0, +-1, 2, inf / 0, +-1, +-2, 1/2, inf).  Geometry is given by the OUTPUT extent oh x ow = (h + 2*pad - 1) x
 * (w + 2*pad - 1).  hnd_wino2_output with `stats` also emits the per-block partial (sum, sum of squares) per channel of
 * what it stored: stats[hnd_wino2_stats_blocks(...)][2][cout], the layout hnd_bn_finalize reads (ntiles = blocks). */
int64_t hnd_wino2_tiles_pad(int n, int oh, int ow, int tile);
int hnd_wino2_stats_blocks(int n, int oh, int ow, int cout, int tile);
int hnd_wino2_weights(const float* weight, float* u, int cout, int cin, int dgrad, int tile, void* stream);
int hnd_wino2_input(const float* x, float* v, int n, int h, int w, int c, int pad, const float* pro_scale,
                    const float* pro_shift, int pro_relu, int tile, void* stream);
int hnd_wino2_output(const float* m, float* y, int n, int oh, int ow, int cout, int ldc, const float* epi_scale,
                     const float* epi_shift, int relu, float* stats, int tile, void* stream);
/* Weight gradient of the same convs in the Winograd domain, F(2x2 taps, tile x tile) over the same points: the data
 * transform is the v [(tile+1)^2][tiles_pad][cin] hnd_wino2_input made in the forward pass (keep it); hnd_wino2_dy
 * makes z [(tile+1)^2][tiles_pad][cout] from dy [n][oh][ow][ldy]; the grouped reductions over the tiles
 * (hnd_conv2d_wgrad with kh = kw = 1, groups = (tile+1)^2, x = v, dy = z) give s [groups][cout][cin];
 * hnd_wino2_wgrad_output -> dW [cout][cin][2][2]. */
int hnd_wino2_dy(const float* dy, float* z, int n, int oh, int ow, int cout, int ldy, int tile, void* stream);
/* F(6x6,2x2) output transform of a head conv's DATA gradient that also makes the BatchNorm-backward statistics of the
 * gradient it writes: y = the gradient g w.r.t. the output of the train-mode BatchNorm(+ReLU) that normalises x (raw conv
 * output, geometry of y); partials [hnd_wino2_stats_blocks(n, oh, ow, cout, 6)][2][cout] = per-block sums of
 * d = g * [x*scale+shift > 0] (mask only when relu_of_bn) and d * (x - mean) * rstd -- what hnd_bn_bwd_reduce computes in a
 * pass of its own -- for hnd_bn_bwd_finalize (ntiles = that block count).  cout must divide 512. */
int hnd_wino26_output_bnbwd_stats(const float* m, float* y, int n, int oh, int ow, int cout, int ldc, const float* x,
                                  const float* scale, const float* shift, const float* mean, const float* rstd,
                                  int relu_of_bn, float* partials, void* stream);
/* BatchNorm backward "apply" fused into both consumers of its result (F(6x6,2x2) only): with d = g * [x*scale+shift > 0]
 * (mask only when relu) and dy = k1*d + k2*x + k3 (k123 from hnd_bn_bwd_finalize), writes
 *   v = the input transform of the conv's DATA gradient over dy with padding `pad` (what hnd_wino2_input(dy, pad) gives),
 *   z = the dy transform of its Winograd-domain WEIGHT gradient (what hnd_wino2_dy(dy) gives),
 * without materialising dy: g, x [n][oh][ow][c] are read once per patch instead of hnd_bn_bwd_apply (12 B per element)
 * plus one read of dy per transform.  v / z extents: hnd_wino2_tiles_pad(n, oh + 2 pad - 1, ow + 2 pad - 1, 6) resp.
 * hnd_wino2_tiles_pad(n, oh, ow, 6) tiles x 49 components x c. */
int hnd_wino26_bnbwd_transforms(const float* g, const float* x, const float* scale, const float* shift,
                                const float* k123, int relu, int n, int oh, int ow, int c, int pad, float* v, float* z,
                                void* stream);
int hnd_wino2_wgrad_output(const float* s, float* dw, int cout, int cin, int tile, void* stream);
/* ABI 10.  The same with s stored [groups][cin][cout] (s_transposed != 0): what the grouped reductions give when they
 * are run with the operands SWAPPED (hnd_conv2d_wgrad with x = z, dy = v) -- a 256 -> 64 conv (encoder.5,
 * resnet_layer.py:48) then presents 64 columns x 256 rows, the shape the ring weight-gradient kernel takes. */
int hnd_wino2_wgrad_output_t(const float* s, float* dw, int cout, int cin, int tile, int s_transposed, void* stream);

/* nn.MaxPool2d(3, 2, 1) (custom/resnet.py:30,99) NHWC; idx (uint8 tap 0..8) kept for backward. */
int hnd_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int n, int h, int w, int c, int oh, int ow,
                         void* stream);
/* backward of maxpool -> ReLU -> FrozenBN of the stem in one pass:
 *   dx = (sum of pooled grads routed here) * [act > 0] * fbn_scale[c]     (act = stem ReLU output) */
int hnd_maxpool3x3s2_bwd_relu_scale(const float* dy, const uint8_t* idx, const float* act, const float* fbn_scale,
                                    float* dx, int n, int h, int w, int c, int oh, int ow, void* stream);

/* Train-mode nn.BatchNorm2d (resnet_layer.py:44-64; eps 1e-5, momentum 0.1):
 * finalize batch statistics from the conv epilogue partials.  Outputs scale=gamma*rstd,
 * shift=beta-mean*scale (consumed as the next conv's prologue), saves mean/rstd for backward and
 * updates running_mean/var (unbiased var) and num_batches_tracked exactly like torch. */
/* `c` real channels; `cs` = channel stride of the activation tensor / partials (cs==c except the
 * 3-channel bottleneck stored as 4).  gamma/beta/running_* have c entries; scale/shift/save_* have cs
 * entries and are zeroed for the pad channels. */
int hnd_bn_finalize(const float* partials, int ntiles, int c, int cs, int64_t count, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float momentum, float eps, float* scale, float* shift, float* save_mean, float* save_rstd,
                    void* stream);
/* y = x*scale[ch] + shift[ch] (+ReLU), NHWC with channel stride cs (multiple of 4). */
/* mask_out (or NULL): the ReLU-mask nibbles of y, see hnd_conv_desc.mask_out */
int hnd_affine_relu(const float* x, const float* scale, const float* shift, float* y, int64_t npix, int cs,
                    int relu, uint8_t* mask_out, void* stream);
/* bits[e] = the nibble [x[4e] > 0, x[4e+1] > 0, x[4e+2] > 0, x[4e+3] > 0] (bit 0 = the first): the mask of an activation that
 * was stored without one, in hnd_conv_desc.mask_bits' layout (1/16 of the bytes for the data-gradient launch that applies
 * it).  n4 = elements / 4; x 16-byte aligned.  ABI 11. */
int hnd_relu_mask_nibbles(const float* x, uint8_t* bits, int64_t n4, void* stream);

/* BatchNorm backward, pass 1: with d = g * [ (x*scale+shift) > 0 ] (mask only when relu),
 * partial sums over pixels of  d  and  d * xhat  (xhat = (x-mean)*rstd) per channel. */
int hnd_bn_bwd_ntiles(int64_t npix);
int hnd_bn_bwd_reduce(const float* g, const float* x, const float* scale, const float* shift,
                      const float* mean, const float* rstd, int relu, int64_t npix, int cs,
                      float* partials /* [hnd_bn_bwd_ntiles][2][cs] */, void* stream);
/* pass 2 (finalize): dgamma, dbeta and the per-channel coefficients of
 *   dx = k1*d + k2*x + k3,  k1 = gamma*rstd, k2 = -k1*rstd*dgamma/N, k3 = k1*(mean*rstd*dgamma - dbeta)/N */
int hnd_bn_bwd_finalize(const float* partials, int ntiles, int c, int cs, int64_t count, const float* gamma,
                        const float* mean, const float* rstd, float* dgamma, float* dbeta,
                        float* k123 /* [3][cs] */, void* stream);
/* pass 3: dx = k1*d + k2*x + k3 with the same mask recomputed. */
int hnd_bn_bwd_apply(const float* g, const float* x, const float* scale, const float* shift, const float* k123,
                     int relu, float* dx, int64_t npix, int cs, void* stream);

/* GHND/HND loss (src/distillation/loss.py:27-33 with nn.MSELoss(reduction='sum')): for up to 8
 * (teacher, student) pairs  loss = sum_k factor_k * sum (t_k - s_k)^2, and in the same pass the
 * gradient w.r.t. the student tensor  grad_k = 2*factor_k*(s_k - t_k)  (masked by s_k>0 when
 * relu_mask, i.e. already pushed through the ReLU that produced s_k).
 * loss_out: device double[1 + npairs] = total, per-term.  scratch: device double, hnd_mse_scratch_elems(). */
typedef struct hnd_mse_pair {
  const float* teacher;
  const float* student;
  float* grad;            /* may be NULL */
  int64_t numel;
  float factor;
  int32_t relu_mask;
} hnd_mse_pair;
size_t hnd_mse_scratch_elems(void);
int hnd_mse_sum_fwd_bwd(const hnd_mse_pair* pairs, int npairs, double* loss_out, double* scratch, void* stream);
/* grad *= *scale_dev unless *scale_dev == 1 (autograd's grad_output of the scalar loss) */
int hnd_scale_by_device_scalar(float* x, int64_t numel, const float* scale_dev, void* stream);

/* torch.optim.Adam.step (func_util.get_optimizer at src/mimic_runner.py:67-68; defaults
 * betas (0.9,0.999), eps 1e-8, no weight decay, no amsgrad) over one flat arena; `step` is the
 * 1-based step count; grad_scale folds the 1/world_size of the DDP gradient average. */
int hnd_adam_step_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
                       float lr, float beta1, float beta2, float eps, int64_t step, float grad_scale, void* stream);

/* LastLevelMaxPool: F.max_pool2d(x, 1, 2, 0) == x[:, ::2, ::2] (torchvision FPN). */
int hnd_subsample2(const float* x, float* y, int n, int h, int w, int c, int oh, int ow, void* stream);
int hnd_fill(float* x, int64_t numel, float value, void* stream);
/* ABI 12, student-side loss terms below the layer outputs (src/distillation/tool.py:25-35 hooks ANY module):
 * hnd_upsample_nearest_bwd: autograd backward of F.interpolate(coarse, size=(H, W), mode='nearest') in the FPN's top-down path
 * (torchvision 0.4.2 ops/feature_pyramid_network.py): g_coarse[n][h][w][c] (+)= the sum of g_fine over every fine pixel
 * that read it (fixed order, no atomics); hnd_add_inplace: x += y (the gradient of a term on the bottleneck tensor joins the
 * gradient arriving from the decoder). */
int hnd_upsample_nearest_bwd(const float* g_fine, float* g_coarse, int n, int H, int W, int h, int w, int c, int accumulate,
                             void* stream);
int hnd_add_inplace(float* x, const float* y, int64_t numel, void* stream);

/* Eval-time bottleneck codec: Quantizer / Dequantizer of src/structure/transformer.py:131-153
 * over myutils tensor_util.quantize_tensor / dequantize_tensor (un-vendored; semantics as used at :139,:152):
 *   scale = (max - min) / (2^bits - 1);  zero_point = trunc(clamp(0 - min/scale, 0, 2^bits - 1));
 *   q = round_half_even(clamp(zero_point + x/scale, 0, 2^bits - 1));   x' = scale * (q - zero_point)
 * x is NHWC with channel stride cs; only the first c channels take part (pad channels stay 0).
 * qparams (device float[4]) = {min, max, scale, zero_point}; scratch: device float[hnd_minmax_scratch_elems()]. */
size_t hnd_minmax_scratch_elems(void);
int hnd_quantize_u8(const float* x, int64_t npix, int c, int cs, uint8_t* q, float* qparams, float* scratch,
                    void* stream);
int hnd_dequantize_u8(const uint8_t* q, const float* qparams, float* x, int64_t npix, int c, int cs, void* stream);
/* Quantizer(num_bits=16): z.half() then .float() (transformer.py:136-137,149-150) as one round trip */
int hnd_roundtrip_f16(float* x, int64_t numel, void* stream);

/* ---- neural filter: Ext4ResNet, src/models/ext/classifier.py:16-37 --------------------
 * Its three convolutions (+bias via epi_shift) and train-mode BatchNorms reuse hnd_conv2d_igemm / _wgrad / hnd_bn_*.
 * nn.AdaptiveAvgPool2d((oh, ow)) (classifier.py:20,30) on NHWC, c % 4 == 0; window o = [floor(o*in/out),
 * ceil((o+1)*in/out)); the backward is the gather form (deterministic). */
int hnd_adaptive_avgpool_fwd(const float* x, float* y, int n, int h, int w, int c, int oh, int ow, void* stream);
int hnd_adaptive_avgpool_bwd(const float* dy, float* dx, int n, int h, int w, int c, int oh, int ow, void* stream);
/* nn.Linear on z.flatten(1) (classifier.py:32,36): x is NHWC [n][hw][cs] holding logical [n][c][hw]; weight is
 * torch's [nout][c*hw] (NCHW flatten order), out [n][nout].  hnd_linear_bwd writes dweight [nout][c*hw], dbias [nout]
 * (either may be NULL together) and dx [n][hw][cs] (NULL to skip; pad channels written as 0). */
int hnd_linear_fwd(const float* x, const float* weight, const float* bias, float* out, int n, int hw, int c, int cs,
                   int nout, void* stream);
int hnd_linear_bwd(const float* x, const float* weight, const float* dout, float* dweight, float* dbias, float* dx,
                   int n, int hw, int c, int cs, int nout, void* stream);
/* z.softmax(dim=1) of the eval-mode classifier (classifier.py:37) */
int hnd_softmax_rows(const float* x, float* y, int rows, int cols, void* stream);
/* ABI 10.  nn.functional.cross_entropy(ext_logits, ext_targets) of the filter's training step (reference
 * src/ext_runner.py:58) and its gradient in one launch: loss[0] = mean over the counted rows of
 * logsumexp(logits[r]) - logits[r][labels[r]]; dlogits[r] = (softmax(logits[r]) - onehot(labels[r])) / counted.  Rows
 * whose label equals ignore_index (torch's default: -100) are not counted and get a zero gradient.  logits, dlogits:
 * [rows, cols] fp32 row-major; labels: [rows] int64.  Fixed summation order. */
int hnd_softmax_ce_rows_fwd_bwd(const float* logits, const int64_t* labels, int rows, int cols, int64_t ignore_index,
                                float* loss, float* dlogits, void* stream);
/* per-channel sum over pixels of x [npix][cs] -> out[c]: the bias gradient of nn.Conv2d (autograd, ext_runner.py:72).
 * Two passes in a fixed order (bit-reproducible); scratch: device float[hnd_channel_sum_scratch_elems(c)]. */
size_t hnd_channel_sum_scratch_elems(int c);
int hnd_channel_sum(const float* x, float* out, int64_t npix, int c, int cs, float* scratch, void* stream);
/* torch.optim.SGD.step (ext_runner.py:118-120; config/ext: lr 1e-3, momentum 0.9, weight_decay 1e-4) over a flat
 * arena: d = grad*grad_scale + weight_decay*p; buf = first_step ? d : momentum*buf + (1-dampening)*d;
 * p -= lr * (nesterov ? d + momentum*buf : buf).  momentum_buf may be NULL when momentum == 0. */
int hnd_sgd_step_flat(float* param, const float* grad, float* momentum_buf, int64_t numel, float lr, float momentum,
                      float dampening, float weight_decay, int nesterov, int first_step, float grad_scale,
                      void* stream);

/* ---- data-parallel gradient exchange: replaces torch DistributedDataParallel's gradient
 * averaging for the student (src/mimic_runner.py:141-143; process group of src/utils/main_util.py:43-62).
 * One process per GPU; RCCL over xGMI, resolved at run time (the RCCL PyTorch-ROCm already loaded is reused).
 *   rank 0:  hnd_comm_unique_id(id, 128) -> ship the 128 bytes to every rank (the host does this over its
 *            torch.distributed store / any side channel) -> every rank: hnd_comm_init(rank, world, id, 128, &comm)
 *            with its GPU current.  The communicator is the ONLY persistent object the library owns.
 *   per step: hnd_allreduce_avg_flat(comm, flat_grad, n, stream): in-place average of the flat fp32 gradient arena
 *            (586 566 floats for the b3ch student) enqueued on `stream`; the caller orders it after the last
 *            weight-gradient kernel and before hnd_adam_step_flat with stream events. */
int hnd_comm_unique_id(void* id_out, size_t bytes);
int hnd_comm_init(int rank, int world, const void* unique_id, size_t bytes, void** comm_out);
int hnd_allreduce_avg_flat(void* comm, float* flat, int64_t n, void* stream);
int hnd_comm_destroy(void* comm);

/* ---- validation path: box branch of the eval-mode detector, reached from
 * src/utils/main_util.py:75-113 (evaluate) -> src/models/org/rcnn.py:124-127 (rpn -> roi_heads -> postprocess); the
 * arithmetic is torchvision 0.4.2's (models/detection/{rpn,roi_heads,_utils}.py, ops/{boxes,poolers}.py and the native
 * CPU operators csrc/cpu/{nms_cpu,ROIAlign_cpu}.cpp).  Convolutions / Linear layers of the heads run on
 * hnd_conv2d_igemm (fc6 = a 7x7 "valid" conv over the pooled 7x7 map).  Kept sets are index work: on identical
 * inputs hnd_nms returns exactly the CPU operator's set (every IoU term is one correctly rounded fp32 operation).
 *
 * hnd_rpn_decode: head = RPN head output of one pyramid level [n][h][w][ldc], channel a = objectness logit of anchor
 *   a, channel A + 4a + c = box delta c of anchor a.  Writes objectness[n][anchors_per_image] and
 *   proposals[n][anchors_per_image][4] at level_offset in (y, x, a) order (concat_box_prediction_layers), where the
 *   anchor is base_anchors[a] (HOST pointer, A x 4 floats: AnchorGenerator.generate_anchors) + (x*stride_w, y*stride_h)
 *   and the box is BoxCoder(1,1,1,1).decode_single with dw, dh clamped to xform_clip = log(1000/16). */
int hnd_rpn_decode(const float* head, int n, int h, int w, int ldc, int num_anchors, const float* base_anchors,
                   float stride_h, float stride_w, int64_t level_offset, int64_t anchors_per_image, float xform_clip,
                   float* objectness, float* proposals, void* stream);
/* clip_boxes_to_image: x to [0, width], y to [0, height], in place; boxes [n][4] */
int hnd_clip_boxes(float* boxes, int64_t n, float height, float width, void* stream);
/* torchvision.ops.nms.  boxes [n][4]; order [n] = indices in descending-score order (hnd_argsort_desc_f32 makes it on
 * the device); keep [n] (bytes) = 1 for boxes that survive greedy suppression at IoU > iou_threshold, indexed like
 * boxes; workspace: hnd_nms_workspace(n) bytes.  n <= 131072. */
size_t hnd_nms_workspace(int64_t n);
int hnd_nms(const float* boxes, const int64_t* order, int64_t n, float iou_threshold, void* workspace,
            uint8_t* keep, void* stream);
/* torchvision.ops.roi_align (non-aligned form) on an NHWC feature map feat [n][h][w][c], c % 4 == 0.
 * rois [K][5] = (batch index, x1, y1, x2, y2); idx [k] selects the rows this launch pools (one pyramid level of
 * MultiScaleRoIAlign); out [K][pooled_h][pooled_w][c], row idx[i] written. */
int hnd_roi_align(const float* feat, int n, int h, int w, int c, const float* rois, const int64_t* idx, int64_t k,
                  float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio, float* out, void* stream);
/* RoIHeads.postprocess_detections, first half: BoxCoder(wx, wy, ww, wh).decode of every class's deltas
 * deltas [nroi][ld] (class c at columns 4c..4c+3) against rois [nroi][5], then clip to that roi's image
 * (image_hw [images][2] = height, width as floats); out [nroi][ncls][4]. */
int hnd_box_decode_clip(const float* deltas, int ld, const float* rois, const float* image_hw, int64_t nroi, int ncls,
                        float wx, float wy, float ww, float wh, float xform_clip, float* out, void* stream);

/* ---- mask / keypoint branches of the eval-mode detector (torchvision 0.4.2 roi_heads.py maskrcnn_inference /
 * keypointrcnn_inference / heatmaps_to_keypoints, transform.py postprocess -> paste_masks_in_image), reached from
 * src/models/org/rcnn.py:124-127 for Mask / Keypoint R-CNN; their convolutions run on hnd_conv2d_igemm (the transposed
 * ones as the data-gradient form).  csrc/detect_heads.hip. */
/* probs[k][m][m] = sigmoid(logits[k][m][m][labels[k]]); logits NHWC with channel stride ldc */
int hnd_mask_probs(const float* logits, const int64_t* labels, int64_t k, int m, int ldc, float* probs, void* stream);
/* paste_masks_in_image after the host expanded / truncated the boxes: probs [k][m][m]; boxes [k][4] int64 (x0, y0, x1,
 * y1, already scaled by (m+2)/m about the centre and truncated); out [k][im_h][im_w] = the mask, zero-padded by one
 * pixel, resized bilinearly (align_corners=False) to (y1-y0+1, x1-x0+1) and pasted at (y0, x0), zero elsewhere */
int hnd_paste_masks(const float* probs, const int64_t* boxes, int64_t k, int m, int im_h, int im_w, float* out,
                    void* stream);
/* Run boundaries of the thresholded masks for the COCO segm evaluator (reference src/utils/coco_eval_util.py:101
 * `masks > 0.5` + pycocotools mask.encode): probs [n][h][w]; out receives, in ANY order, the keys k*h*w + p of every
 * column-major position p = x*h + y >= 1 of mask k whose bit (probs > threshold) differs from the bit at p - 1;
 * *count (device) = how many exist (entries beyond `capacity` are dropped: call again with a larger buffer);
 * first[k] = bit of mask k at p = 0.  The caller sorts the keys; run lengths are their differences. */
int hnd_mask_run_boundaries(const float* probs, int64_t n, int h, int w, float threshold, int64_t* out, int64_t capacity,
                            int64_t* count, uint8_t* first, void* stream);
/* Ground-truth mask resize of the transform (reference src/models/org/rcnn.py:54-57:
 * interpolate(mask[None].float(), scale_factor=s)[0].byte(), mode 'nearest'): in [k][h][w] uint8 -> out [k][oh][ow]
 * uint8 with oh = floor(h*s), ow = floor(w*s); source index min(floor(dst * (float)(1/s)), in-1) as ATen. */
int hnd_resize_mask_nearest_u8(const unsigned char* in, int64_t k, int h, int w, int oh, int ow, double scale_factor,
                               unsigned char* out, void* stream);
/* F.interpolate(scale_factor=factor, mode='bilinear', align_corners=False) on an NHWC tensor in [k][h][w][c] */
int hnd_upsample_bilinear_nhwc(const float* in, int64_t k, int h, int w, int c, int factor, float* out, void* stream);
/* heatmaps_to_keypoints: maps [k][h][w][ldc] (keypoint j = channel j), rois [k][4] (x1, y1, x2, y2); per RoI the
 * heatmaps are resized bicubically (A=-0.75, align_corners=False) to (ceil(max(y2-y1,1)), ceil(max(x2-x1,1))), the
 * first maximum is located, xy [k][num_keypoints][3] = (x, y, 1), scores [k][num_keypoints] = the maximum.
 * h*w*4 bytes must fit 64 KiB of LDS (56x56 here). */
int hnd_heatmaps_to_keypoints(const float* maps, int64_t k, int h, int w, int ldc, int num_keypoints, const float* rois,
                              float* xy, float* scores, void* stream);

/* ---- index bookkeeping of the validation path without library kernels (csrc/select.hip): what torchvision 0.4.2's
 * rpn.py / roi_heads.py / poolers.py do with torch.nonzero, torch.topk and torch.sort.
 * hnd_nonzero_*: out[0..count) = the indices i in [0, n) where the predicate holds, ASCENDING (== torch.nonzero(p)
 * .squeeze(1)); out has room for n entries, *count (device) receives how many were written.
 *   _u8: flags[i] != 0 (the NMS keep bytes);  _gt_f32: x[i] > threshold (roi_heads.py box_score_thresh);
 *   _eq_i64: x[i] == value (poolers.py level assignment);  _min_size: boxes [n][4], width >= m and height >= m
 *   (boxes.py remove_small_boxes).
 * hnd_argsort_desc_f32: order[0..n) = indices sorting `keys` DESCENDING, ties in ascending index order (== torch.sort(
 * descending=True, stable=True)[1]; the first k entries are torch.topk's index set): the RPN's per-level pre-NMS top-k
 * and the score order of hnd_nms.  workspace: hnd_argsort_desc_workspace(n) bytes (caller-owned). */
int hnd_nonzero_u8(const uint8_t* flags, int64_t n, int64_t* out, int64_t* count, void* stream);
int hnd_nonzero_gt_f32(const float* x, int64_t n, float threshold, int64_t* out, int64_t* count, void* stream);
int hnd_nonzero_eq_i64(const int64_t* x, int64_t n, int64_t value, int64_t* out, int64_t* count, void* stream);
int hnd_nonzero_min_size(const float* boxes, int64_t n, float min_size, int64_t* out, int64_t* count, void* stream);
size_t hnd_argsort_desc_workspace(int64_t n);
int hnd_argsort_desc_f32(const float* keys, int64_t n, int64_t* order, void* workspace, void* stream);

/* ---- generic workspace query: bytes of caller-provided scratch an op needs.  `desc` is the op's
 * descriptor where it has one (hnd_wgrad_desc for HND_OP_CONV2D_WGRAD), `arg` an op-specific integer (channels for
 * HND_OP_CHANNEL_SUM), both ignored otherwise.  The op-specific helpers above return the same numbers. */
typedef enum hnd_op {
  HND_OP_CONV2D_IGEMM = 0,   /* no workspace                                   */
  HND_OP_CONV2D_WGRAD = 1,   /* split-K slabs, desc = const hnd_wgrad_desc*    */
  HND_OP_MSE = 2,            /* double[hnd_mse_scratch_elems()]                */
  HND_OP_QUANTIZE_U8 = 3,    /* float[hnd_minmax_scratch_elems()]              */
  HND_OP_CHANNEL_SUM = 4,    /* float[hnd_channel_sum_scratch_elems(arg)]      */
  HND_OP_COMM_UNIQUE_ID = 5, /* bytes of the id buffer of hnd_comm_unique_id   */
  HND_OP_NMS = 6             /* hnd_nms_workspace(arg = number of boxes)       */
} hnd_op;
size_t hnd_workspace_size(int op, const void* desc, int64_t arg);

#ifdef __cplusplus
}
#endif
#endif /* HND_HIP_H */
