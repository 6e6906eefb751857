/*
 * chexpert_hip.h -- C ABI of libchexpert_hip.so (gfx950 / MI355X).
 *
 * The reference (kamenbliznashki/chexpert) has no FFI layer: its hot path is the implicit ATen/cuDNN
 * work behind `model(x)`, `loss.backward()` and `optimizer.step()` (chexpert.py:159-164, :204).  This
 * library is what a maintainer binds INSTEAD of those ATen ops (see INTEGRATION.md for the ctypes
 * stub).  Each entry point cites the reference statement whose arithmetic it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer owned by the caller
 *     (no ownership transfer, no allocation inside the library);
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - returns 0 on success, a positive hipError_t from the launch, or a negative CX_E* validation
 *     code; nothing is launched when validation fails;
 *   - activations are NHWC bf16 with an explicit channel pitch (`ld*`, in elements) so a kernel can
 *     read / write a channel slice of a wider dense-block buffer (the reference's torch.cat,
 *     torchvision _DenseLayer.forward, is never materialised);
 *   - statistics, coefficient vectors and gradients of parameters are fp32.
 */
#ifndef CHEXPERT_HIP_H
#define CHEXPERT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CX_ABI_VERSION 10

enum { CX_EINVAL = -1, CX_EALIGN = -2, CX_ESHAPE = -3, CX_EUNSUPPORTED = -4, CX_ESTATROWS = -5 };

/* storage type of the activation tensors of a CxConv / CxWgrad: bf16 (the fast path: fp32 accumulation, fp32 statistics) or
 * fp32 (the parity mode of north_star "1e-3 fp32": same schedule, exact f32 MFMA, fp32 packed weights [tap][N][K])        */
enum { CX_DT_BF16 = 0, CX_DT_F32 = 1 };

/* A-operand prologues of the implicit GEMM (fused normalisation, never stored) */
enum {
  CX_PRO_NONE = 0,         /* a = x                                                              */
  CX_PRO_AFFINE_RELU = 1,  /* a = relu(x*pa[c] + pb[c])          BN/IN (scale,shift) + ReLU       */
  CX_PRO_AFFINE2 = 2,      /* a = x*pa[c] + x2*pb[c] + pc[c]     BN backward / deferred correction */
  CX_PRO_JOIN = 3          /* ABI 9, the residual join of the block BELOW as the prologue of a Bottleneck's conv1
                              (attn_aug_conv.py:188-211: `out = relu(bn3(conv3(.)) + identity)` followed by the next block's
                              `conv1(out)`): t = relu(x*pa[c] + (x2 [+ x3]) + pc[c]) with x = conv3's raw output, x2 / x3 = the
                              hi / lo planes of the identity operand (the residual stream, cx_join_fwd), a = bf16(t).  Side
                              outputs, written once (by the first N tile): pro_out = hi plane of t (bf16, pitch ldpo), po_lo =
                              its lo plane (one byte per element; NULL: single-plane output, the sign bits are then those of the
                              rounded tensor as cx_affine2_relu_mask writes them), po_mask = sign bits [t > 0] (one byte per 8
                              channels, bit c%8; NULL in eval mode), both in the side-plane layout of cx_join_fwd.  Bit for bit what cx_join_fwd leaves.  1x1, stride 1, bf16, K % 64 == 0. */
};
/* addressing modes */
enum {
  CX_MODE_CONV = 0,   /* kh x kw taps, stride, zero padding (applied after the prologue)          */
  CX_MODE_POOL2 = 1,  /* 1x1 on the 2x2 average of the prologue output (transition: conv o avgpool
                         commute, attn_aug_conv.py:433-434)                                      */
  CX_MODE_STEM = 2    /* 7x7 stride 2 pad 3 on a (B,H,W,4) bf16 image; 7 row-taps of 8px*4ch      */
};
/* epilogues */
enum {
  CX_EPI_STORE = 0,   /* y = bf16(acc); optional per-channel sum / sum of squares (fp32 atomics)   */
  CX_EPI_MASK = 1,    /* dz = acc * [ex*e_sc + e_sh > 0]; S1 += sum dz; S2 += sum dz*(ex-e_mu)*e_r;
                         y = (accumulate ? y : 0) + e_scale*dz            (ReLU+BN backward)      */
  CX_EPI_JOIN = 2     /* ABI 8, residual join backward folded into the input gradient that completes the join's output
                         gradient (attn_aug_conv.py:202-209: out = relu(bn3(z3) + identity)): t = bf16(y + acc) is the gradient of
                         `out` (y holds the part that arrived through the identity path: accumulate must be set),
                         dz = t * [bit of emask], S1 += sum dz, S2 += sum dz*(ex-e_mu)*e_r with ex = z3 (the join BatchNorm's
                         input), y = dz.  Bit for bit what CX_EPI_STORE + cx_relu_bwd_stats_mask leave in y.  bf16, stride 1. */
};

typedef struct CxConv {
  const void* x;        /* bf16 input (B,H,W,ldx)                                                 */
  const void* x2;       /* second input for CX_PRO_AFFINE2 (same geometry, pitch ldx2) or NULL     */
  const void* w;        /* packed bf16 weights [taps][N][K] (K contiguous), see cx_pack_weights    */
  void* y;              /* bf16 output (B,Ho,Wo,ldy), written at channel 0 of the pointer          */
  const float* pa; const float* pb; const float* pc;          /* prologue vectors [K]             */
  float* stat_sum; float* stat_sq;                             /* [N] or NULL                      */
  const void* ex;       /* CX_EPI_MASK: bf16 tensor (B,Ho,Wo,ldex) the ReLU mask / xhat come from  */
  const float* e_sc; const float* e_sh; const float* e_mu; const float* e_r; const float* e_scale; /* [N] */
  int32_t B, H, W, Ho, Wo;
  int32_t K, N;         /* channels per tap in / out; both multiples of 8                          */
  int32_t ldx, ldx2, ldy, ldex;
  int32_t kh, kw, stride, pad;
  int32_t prologue, mode, epilogue, accumulate;   /* accumulate: y += result (both epilogues)             */
  int32_t tstride;      /* 2: input gradient of a stride-2 conv (x is its output gradient,              */
                        /* stride must be 1, pad = kh-1-forward_pad, weights packed with transpose=1)    */
  int32_t stat_replicas; /* R > 1: workgroup b adds its statistics to replica b % R, replica r of channel */
  int32_t stat_rstride;  /* n lives at stat_sum[r*stat_rstride + n].  Thousands of workgroups adding to   */
                         /* the same N floats serialise at the memory side; the consumer (cx_bn_coef /   */
                         /* cx_bn_bwd_coef) sums the replicas.  0 or 1: a single copy                     */
  int32_t stat_det;      /* != 0: DETERMINISTIC statistics.  No atomics: the launch writes `rows` complete rows   */
                         /* stat_sum[r*stat_rstride + n] (r < rows, every n < N; one writer per element, partial */
                         /* sums combined in a fixed order inside the workgroup), rows <= stat_replicas (the     */
                         /* caller's capacity) or CX_ESTATROWS; cx_last_stat_rows() then returns `rows` and the   */
                         /* consumer sums exactly those rows in row order (cx_bn_coef / cx_bn_bwd_coef with      */
                         /* replicas = rows): bit-identical results from run to run.  No zero-fill is needed      */
  int32_t dtype;         /* CX_DT_BF16 (0) or CX_DT_F32: x, x2, y, ex are fp32, w is fp32 [tap][N][K]; K, N, ld* % 4 */
  /* ABI 7.  Optional side output of the prologue: the transformed input (what the convolution actually consumes, e.g. the     */
  /* dense-layer output gradient after the deferred BatchNorm correction) stored as a dense bf16 (B,H,W,ldpo) tensor, K        */
  /* channels per pixel.  The 3x3 weight gradient of the same layer then reads ONE dense 64-byte row per pixel               */
  /* (g_prologue NONE) instead of two 64-byte pieces of 512..2048-byte rows of the block buffers.  Kernels that cannot        */
  /* write it ignore the field: cx_last_pro_out() says whether the last cx_conv_gemm of this thread did.                     */
  void* pro_out;
  int32_t ldpo;
  int32_t dil;           /* ABI 9 (was a zero pad field).  0 / 1: adjacent taps.  d > 1: taps d pixels apart -- torchvision conv3x3(..., dilation) of a */
                         /* Bottleneck under replace_stride_with_dilation (attn_aug_conv.py:183, :266-271): CX_MODE_CONV, kernel extent           */
                         /* d (kh - 1) + 1 in every shape rule; runs on the generic implicit GEMM (the tiled kernels assume adjacent taps)          */
  /* ABI 8.  CX_EPI_JOIN: the forward join's sign bits as cx_affine2_relu_mask / cx_join_fwd wrote them (side-plane layout, ABI 9): bit n%8 of chunk n/8 = [out[m][n] > 0] */
  const uint8_t* emask;
  /* ABI 9.  CX_PRO_JOIN: lo plane of the identity operand (B,H,W,K) int8 (NULL: x2 is a single-plane bf16 tensor), and the lo /   */
  /* sign-bit side outputs (dense: K bytes resp. K/8 bytes per pixel)                                                            */
  const int8_t* x3;
  int8_t* po_lo;
  uint8_t* po_mask;
  /* ABI 10.  Per-call kernel selection for tests and micro-benchmarks (stateless: replaces the process-wide dbg_* selectors of      */
  /* earlier ABIs).  0: the library picks.  Low byte 1: the generic implicit-GEMM kernels (conv_gemm.hip / conv_wgrad.hip),        */
  /* 2: the tiled kernels (conv_mm.hip / wgrad_mm.hip) where the shape allows.  Second byte f + 1: tile form f (1 = 128 x 128,     */
  /* 2 = 256 x 128 (weight gradient only), 3 = 128 x 256; weight gradient f = 0: the strip kernel for 3x3; f = 5: the activation-   */
  /* stationary 1x1 kernel, conv1x1_xs.hip, or an error; f = 7 / 8: ring / producer-consumer 3x3 forms).  CX_KERNEL_HINT(on, f).     */
  int32_t kernel_hint;
  int32_t pad2_;
} CxConv;
#define CX_KERNEL_HINT(on, form) ((((on) < 0) ? 0 : (on) + 1) | ((((form) < 0) ? 0 : (form) + 1) << 8))

/* Weight gradient of the same convolution:  dW[n][c][ky][kx] += sum_m G[m][n] * A[m@tap][c]
 *   G = dY (prologue NONE or AFFINE2 with vectors ga/gb/gc over N, second tensor g2)
 *   A = the forward A operand (prologue NONE / AFFINE_RELU with pa/pb over K, any mode)          */
typedef struct CxWgrad {
  const void* g; const void* g2;          /* bf16 (B,Ho,Wo,ldg)                                    */
  const void* x;                           /* bf16 forward input (B,H,W,ldx)                        */
  float* dw;                               /* fp32 OIHW gradient, accumulated with atomics          */
  const float* ga; const float* gb; const float* gc;   /* [N]                                       */
  const float* pa; const float* pb;                      /* [K]                                       */
  int32_t B, H, W, Ho, Wo, K, N;
  int32_t ldg, ldg2, ldx;
  int32_t kh, kw, stride, pad;
  int32_t g_prologue, x_prologue, mode;
  int32_t splits;                          /* pixel-range splits (0 = library picks)                */
  int32_t dtype;                           /* CX_DT_BF16 (0) or CX_DT_F32 (g, g2, x fp32)          */
  int32_t dil;                             /* ABI 9 (fills what was alignment padding): as CxConv.dil           */
  /* Optional workspace for a reproducible sum (ABI 4).  The pixel range of a weight gradient is split over workgroups; */
  /* with scratch == NULL (or too small for this launch) the partial tiles are added to dw with fp32 atomics, whose order */
  /* changes from run to run.  With scratch_floats >= splits * |dW| every workgroup plain-stores its partial tile into   */
  /* slab `split` and a second launch on the same stream adds the slabs to dw in split order: bit-identical results.      */
  /* The library picks `splits`; 16 M floats cover every layer of the reference's networks at their benchmark batches.    */
  float* scratch;
  int64_t scratch_floats;
  int32_t kernel_hint;                     /* ABI 10: as CxConv.kernel_hint                                      */
  int32_t pad_;
} CxWgrad;

/* ABI 8.  The 3x3 weight gradients (K = 128 -> N = 32, stride 1, pad 1) of up to CX_WGRAD_BATCH_MAX dense layers of ONE dense block
 * (same B, H, W, pitches) in one launch: torchvision `_DenseLayer.conv2` as restated at attn_aug_conv.py:13, weight gradient of
 * every layer of a `_DenseBlock` (:476-483).  A dense layer's 3x3 weight gradient feeds only the flat gradient buffer, so the
 * launches of a block do not depend on each other once every layer's output-gradient slice exists as a dense tensor
 * (CxConv.pro_out) -- batched they leave the input-gradient chain and share one grid (workgroup = layer x pixel range x 32-channel
 * tile).  `geo` gives the common geometry and prologue (g_prologue NONE, x_prologue AFFINE_RELU), the slab workspace (scratch:
 * n * splits * 36864 floats are used, cx_last_slab_floats() reports them) and dtype; its g / x / dw / pa / pb are ignored.
 * items[i]: g = dense (B,H,W,ldg) gradient slice, x = saved bottleneck tensor (B,H,W,ldx), pa / pb = its norm2 scale / shift [128],
 * dw = fp32 OIHW gradient (32,128,3,3), added in split order (immediately or through cx_wgrad_defer like cx_conv_wgrad).
 * CX_EUNSUPPORTED: shape / workspace outside the kernel's range -- call cx_conv_wgrad per layer instead.                          */
#define CX_WGRAD_BATCH_MAX 24
typedef struct CxWgradBatch {
  const void* g[CX_WGRAD_BATCH_MAX];
  const void* x[CX_WGRAD_BATCH_MAX];
  const float* pa[CX_WGRAD_BATCH_MAX];
  const float* pb[CX_WGRAD_BATCH_MAX];
  float* dw[CX_WGRAD_BATCH_MAX];
  int32_t n, pad_;
} CxWgradBatch;
int cx_conv3x3_wgrad_batch(const CxWgrad* geo, const CxWgradBatch* items, void* stream);

/* ABI 8.  Channel-padded twin (the CIFAR DenseNet-BC of models/test_model.py:306: growth 12, widths 24 + 12 i are not multiples of
 * the kernels' 8-channel vectors).  The network runs on a twin whose dense layers write kp = 16 channels (12 real + 4 that stay
 * zero: zero weights, zero BatchNorm gain and shift) and whose transitions are padded likewise; this entry point moves parameters
 * / running statistics real -> padded (dir 0, store) and gradients / running statistics padded -> real (dir 1; accumulate: add)
 * between two flat fp32 buffers, one descriptor per tensor, ONE launch.  Element (o, j, t) of the real [O][Ireal][taps] tensor is
 * element (o, pos(j), t) of the padded [.][Ipad][taps] one, pos(j) = j (+ shift from `split` on) for j < c0r, else
 * c0p + ((j-c0r)/k)*kp + (j-c0r)%k.                                                                                             */
typedef struct CxChanMapDesc {
  int64_t real_off, pad_off;         /* element offsets into the two flat buffers                                             */
  int32_t O, taps, Ireal, Ipad;      /* rows copied (the padded tensor may have more), kh*kw, channels per row                   */
  int32_t c0r, c0p, k, kp;           /* channel map (k = kp = 1, c0r = Ireal: identity with a wider row)                          */
  int32_t split, shift;              /* inside the first c0r channels: j >= split sits at j + shift (split = c0r, shift = 0: none) --
                                        the output of an attention-augmented transition is [conv branch | pad | attention channels] */
} CxChanMapDesc;
int cx_chan_map_table(float* real_flat, float* padded_flat, const CxChanMapDesc* table_dev, int n_desc, int dir, int accumulate,
                      void* stream);

/* Deferred slab sums (ABI 6).  A training step issues ~120 weight-gradient launches whose partial tiles (CxWgrad.scratch) each
 * need a small ordered sum into dw; one launch per sum puts ~60 of them on the critical stream (6-7 us each on DenseNet121).
 * cx_wgrad_defer(1) switches the calling thread to deferral: every weight-gradient entry point (cx_conv_wgrad,
 * cx_conv1x1_dgrad_wgrad_ws) then only stores its partial tiles -- the caller must hand each launch a scratch region that
 * stays untouched until the sums have run (cx_last_slab_floats() tells how much of it the launch used; 0 = it fell back to atomics)
 * -- and records {dw, slab, total, splits}.  cx_wgrad_defer_take() returns the records (first_block filled in), which the caller
 * copies to the device once per distinct schedule, and cx_dw_reduce_table() adds them all in ONE launch on a stream that has
 * been joined with the producers: every dw element receives exactly the additions, in the order, of its own immediate sum,
 * so results are bit-identical to the immediate form.  cx_wgrad_defer(0) returns to immediate sums and keeps the pending records
 * (a single launch can be taken out of the deferral that way), cx_wgrad_defer(-1) also drops them.                              */
typedef struct CxReduceDesc {
  float* dw; const float* slab;      /* dw[i] += slab[0*total + i] + slab[1*total + i] + ... (fixed association)             */
  int64_t total;                     /* elements of dw                                                                        */
  int32_t splits, vec, first_block, pad_;
  int32_t cols, dw_ld;               /* ABI 9: cols != 0: the slab tile is [total / cols][cols] and goes to dw[r * dw_ld + c] (a column   */
  int64_t pad2_;                     /* range of a wider matrix: a layer's newest 32 input channels, or all but those); 0: contiguous    */
} CxReduceDesc;
int cx_wgrad_defer(int on);                                            /* returns the previous state                          */
int cx_wgrad_defer_take(CxReduceDesc* out_host, int capacity, int64_t* total_blocks);   /* n records, or -n if capacity < n   */
int cx_last_slab_floats(void);
int cx_dw_reduce_table(const CxReduceDesc* table_dev, int n, int64_t total_blocks, void* stream);

int cx_abi_version(void);
int cx_last_pro_out(void);      /* 1: the last cx_conv_gemm of the calling thread wrote CxConv.pro_out */
const char* cx_error_string(int code);
/* name of the kernel instantiation the most recent cx_conv_gemm / cx_conv1x1_dgrad_wgrad* / cx_conv_wgrad call of this thread
 * dispatched to, spelled as rocprofv3 lists it (e.g. "pw_bwd2_kernel<2, true, 0>"; a stride-2 input gradient that runs as up to
 * four parity-class launches carries the suffix " x parity classes").  For measurement tools: valid until the next call.       */
const char* cx_last_kernel(void);
/* rows written by the most recent successful stat_det launch issued from the calling thread */
int cx_last_stat_rows(void);

/* conv forward / input-gradient as one implicit GEMM family.
 * Replaces F.conv2d + F.batch_norm + F.relu (+ torch.cat, avg_pool2d) forward and their autograd
 * input-gradients: torchvision _DenseLayer (norm1-relu1-conv1-norm2-relu2-conv2), _Transition
 * (attn_aug_conv.py:431-434), features.conv0 (:461), Bottleneck convs (:194-203).               */
int cx_conv_gemm(const CxConv* p, void* stream);
/* weight gradient (autograd of the same convs)                                                   */
int cx_conv_wgrad(const CxWgrad* p, void* stream);

/* Input gradient AND weight gradient of the dense-layer bottleneck 1x1 convolution in one pass over dZ and the activation
 * slice (K = 128 gradient channels, CX_EPI_MASK): p as for cx_conv_gemm (p->ex = the activation slice x, p->e_sc / p->e_sh its
 * BatchNorm scale / shift), and dW[n][c] += sum_m dZ[m][n] * relu(x[m][c]*e_sc[c] + e_sh[c]) into the fp32 OIHW gradient
 * dw (128, N, 1, 1) -- the same result as cx_conv_gemm followed by cx_conv_wgrad with x_prologue = AFFINE_RELU(e_sc, e_sh). */
int cx_conv1x1_dgrad_wgrad(const CxConv* p, float* dw, void* stream);
/* the same with a workspace for the reproducible weight-gradient sum (see CxWgrad.scratch; NULL / 0 = atomics) */
int cx_conv1x1_dgrad_wgrad_ws(const CxConv* p, float* dw, float* scratch, int64_t scratch_floats, void* stream);
/* ABI 9.  The same with dw a column range of a wider fp32 matrix: row n of the (128, p->N) result goes to dw[n * dw_ld + c] (the 32
 * newest input channels of a dense layer's conv1 weight, torchvision `_DenseLayer.conv1` as restated at attn_aug_conv.py:13).     */
int cx_conv1x1_dgrad_wgrad_ld_ws(const CxConv* p, float* dw, int dw_ld, float* scratch, int64_t scratch_floats, void* stream);
/* ABI 9.  TWO dense layers of one dense block in one pass over the block's activation and gradient buffers: a = layer l restricted to
 * the N channels it shares with b = layer l - 1 (its 32 newest channels go through cx_conv1x1_dgrad_wgrad_ld_ws first: layer l - 1's
 * output gradient depends on them).  Both add e_scale * mask * (dZ W) to y[..., :N] (the second layer on top of the first, each sum
 * rounded to bf16 as the separate passes round it: y is bit-identical to them) and accumulate their weight gradients (dw_a: pitch
 * dw_ld_a, dw_b: pitch N) and statistic rows (a->stat_*, b->stat_*: distinct buffers, same geometry); x and the old gradient are read
 * once and the new gradient written once for the pair.  a and b share ex, y, N, B/H/W, ldex, ldy, accumulate, stat_det and
 * stat_replicas (stat_rstride is each layer's own); prologue AFFINE2, epilogue MASK, K = 128 (accumulate = 0: layer a takes the old gradient as zero, layer
 * b adds to layer a's).  scratch: 2 * splits * 128 * N floats for the ordered sums
 * (cx_last_slab_floats() reports what was used), or NULL for atomics.  No reference counterpart (autograd runs the layers one by one). */
int cx_conv1x1_dgrad_wgrad_pair_ws(const CxConv* a, const CxConv* b, float* dw_a, int dw_ld_a, float* dw_b, float* scratch,
                                   int64_t scratch_floats, void* stream);

/* OIHW fp32 -> packed bf16.  transpose=0: [tap][O][I] (forward);  transpose=1: [tap'][I][O] with
 * taps rotated by 180 degrees (input-gradient of a stride-1 conv).  stem=1: (64,3,7,7) -> [ky][O][8*4].  */
int cx_pack_weights(const float* w_oihw, void* packed, int O, int I, int kh, int kw, int transpose, int stem,
                    void* stream);

/* The same for every conv weight of a model in ONE launch: `flat` is the flat fp32 parameter buffer,
 * `table_dev` a DEVICE array of descriptors (element offsets into flat / packed).                  */
typedef struct CxPackDesc {
  int64_t src_off, dst_off;
  int32_t O, I, kh, kw, transpose, stem;
} CxPackDesc;
int cx_pack_weights_table(const float* flat, void* packed, const CxPackDesc* table_dev, int n_desc, void* stream);

/* (B,3,H,W) fp32 NCHW -> (B,H,W,4) bf16 (4th channel zero).  Replaces x.to(device) layout glue
 * ahead of features.conv0 (chexpert.py:159).                                                      */
int cx_nchw3_to_nhwc4(const float* x, void* y, int B, int H, int W, void* stream);
/* uint8 grey image (B,H,W) -> (B,H,W,4) bf16 with the three identical whitened channels ((u/255 - mean)/std) and a zero pad:
 * the reference transform chain `float().div(255)`, `Normalize(mean, std)`, `expand(3,-1,-1)` (chexpert.py:70-72) done on
 * the GPU from the decoded bytes, so the host pipeline ships 1 B per pixel instead of 12.
 * npix = B*H*W must be a multiple of 16.                                                                             */
int cx_u8_to_nhwc4(const uint8_t* x, void* y, size_t npix, float mean, float std, void* stream);

/* BatchNorm training statistics -> (scale, shift) of the consumer + running-stat update
 * (F.batch_norm, momentum semantics of nn.BatchNorm2d incl. unbiased running_var).
 * sum/sq: fp32 [C] over `count` elements.  gamma/beta may be NULL (-> 1, 0: InstanceNorm-like).
 * running_* may be NULL.  Also emits mean / rstd when non-NULL.                                   */
int cx_bn_coef(const float* sum, const float* sq, float count, const float* gamma, const float* beta, float eps,
               float momentum, float* running_mean, float* running_var, float* scale, float* shift, float* mean,
               float* rstd, int C, int replicas, int rstride, void* stream);   /* sum/sq: `replicas` copies, `rstride` floats apart */
/* BatchNorm coefficients from per-channel moments that already exist (a dense-block channel's batch mean / rstd are computed once,
 * by the kernel that produced the channel; every later norm1 over the concatenation -- torchvision _DenseLayer.norm1 -- re-uses
 * them with its own gamma / beta / running buffers).  Channels [c_lo, c_lo + c_n) are reduced first, in row order, from `rows`
 * deterministic statistic rows (CxConv.stat_det) into mean / rstd.                                                             */
int cx_bn_coef_moments(float* mean, float* rstd, float count, const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, float* scale, float* shift, int C, const float* sum, const float* sq,
                       int rows, int rstride, int c_lo, int c_n, void* stream);
/* eval mode: scale/shift from running statistics                                                 */
int cx_bn_coef_eval(const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                    float eps, float* scale, float* shift, float* mean, float* rstd, int C, void* stream);

/* Backward bookkeeping of one consumer BatchNorm over buffer channels [0,C):
 *   dgamma += S2, dbeta += S1 (fp32 parameter gradients);
 *   deferred per-channel correction  A[c] += r*gamma*S1/count,  Bc[c] += r*gamma*S2/count
 * (the -mean(dz) - xhat*mean(dz*xhat) terms of BN backward are linear in x and shared by every
 * consumer of a dense-block channel, so they are applied once, by the channel's producer).
 * If pa/pb/pc are non-NULL also emits the AFFINE2 vectors of a SINGLE-consumer BN (norm2):
 *   dY = dz*pa + y*pb + pc.                                                                       */
int cx_bn_bwd_coef(const float* S1, const float* S2, float count, const float* gamma, const float* mean,
                   const float* rstd, float* dgamma, float* dbeta, float* A, float* Bc, float* pa, float* pb,
                   float* pc, int C, int replicas, int rstride,
                   /* optional: the cx_bn_bwd_slice_coef vectors of channels [q_lo, q_lo + q_n) from the A / B this call has just
                      completed (their last consumer), qa/qb/qc of q_n floats, or NULL                                   */
                   float* qa, float* qb, float* qc, int q_lo, int q_n, void* stream);   /* S1/S2 replicated as above */
/* AFFINE2 vectors that apply the deferred correction to a gradient slice:
 *   dY_true = G*1 + x*(-r*Bc) + (mean*r*Bc - A)                                                   */
int cx_bn_bwd_slice_coef(const float* A, const float* Bc, const float* mean, const float* rstd, float* pa,
                         float* pb, float* pc, int C, void* stream);

/* stem: y = maxpool3x3s2p1(relu(x*scale+shift)) into a channel slice + stats of the pooled output
 * (features.norm0/relu0/pool0, attn_aug_conv.py:462-464).  argmax: (B,H/2,W/2,C) uint8 window
 * position (0..8) of the first maximum, kept for the backward pass.                               */
int cx_bnrelu_maxpool_fwd(const void* x, const float* scale, const float* shift, void* y, uint8_t* argmax,
                          float* stat_sum, float* stat_sq, int B, int H, int W, int C, int ldy, int stat_rows, void* stream);
/* backward of the same: routes (corrected) dY to the arg-max, applies the ReLU mask, writes dz (bf16,
 * (B,H,W,C)) and accumulates S1 = sum dz, S2 = sum dz*xhat                                       */
int cx_bnrelu_maxpool_bwd(const void* x, const float* scale, const float* shift, const float* mean,
                          const float* rstd, const uint8_t* argmax, const void* g, const void* gx, const float* ga,
                          const float* gb, const float* gc, void* dz, float* S1, float* S2, int B, int H, int W, int C,
                          int ldg, int ldgx, int stat_rows, void* stream);

/* head: pooled[b][c] = mean_hw relu(x*scale+shift); logits = pooled @ Wt + bias
 * (attn_aug_conv.py:514-516)                                                                      */
int cx_head_fwd(const void* x, const float* scale, const float* shift, const float* w, const float* bias,
                float* pooled, float* logits, int B, int HW, int C, int ldx, int n_classes, void* stream);
/* loss = BCEWithLogits(logits,target).sum(1).mean(0); dlogits = (sigmoid - target)/B * grad_scale
 * (chexpert.py:160, :530)                                                                         */
int cx_bce_fwd_bwd(const float* logits, const float* target, float* loss, float* loss_elem, float* dlogits,
                   float grad_scale, int B, int n_classes, void* stream);
/* loss = CrossEntropyLoss(logits, target) (mean over the batch of logsumexp - logit[target]);
 * dlogits = (softmax - onehot)/B * grad_scale; loss_elem (optional) = the per-sample terms
 * (models/test_model.py:118, :143, :331: the CIFAR harness criterion)                              */
int cx_softmax_ce_fwd_bwd(const float* logits, const int64_t* target, float* loss, float* loss_elem, float* dlogits,
                          float grad_scale, int B, int n_classes, void* stream);
/* head backward: dW += dlogits^T pooled, db += sum dlogits, dpooled = dlogits @ W;
 * then gradient into the block buffer through GAP + ReLU + norm5 (mask epilogue semantics):
 * dz = dpooled/HW * [x*scale+shift>0]; S1,S2 += ...; g = e_scale*dz (written, not accumulated)   */
int cx_head_bwd(const float* dlogits, const float* pooled, const float* w, float* dw, float* db,
                float* dpooled, int B, int C, int n_classes, void* stream);
int cx_gap_relu_bn_bwd(const float* dpooled, const void* x, const float* scale, const float* shift,
                       const float* mean, const float* rstd, const float* e_scale, void* g, float* S1, float* S2,
                       int B, int HW, int C, int ldx, int ldg, int stat_rows, void* stream);

/* transition backward glue: un-pool (each of the 4 inputs gets d/4), ReLU mask + BN mask epilogue */
int cx_unpool2_mask(const void* d, const void* x, const float* sc, const float* sh, const float* mean,
                    const float* rstd, const float* e_scale, void* g, float* S1, float* S2, int B, int H, int W,
                    int C, int ldd, int ldx, int ldg, int stat_rows, void* stream);

/* residual join of a Bottleneck: out = relu(a*pa + b*pb + pc) (bn3(conv3) + identity | bn_d(downsample),
 * attn_aug_conv.py:202-209); pa/pb/pc fp32 [C]                                                    */
int cx_affine2_relu(const void* a, const void* b, const float* pa, const float* pb, const float* pc, void* out, size_t rows,
                    int C, void* stream);
/* ABI 9.  Dropout on the new feature slice of a dense layer (torchvision `_DenseLayer.forward`: `F.dropout(new_features, p=drop_rate,
 * training=self.training)`; the reference's DenseNet hands drop_rate to torchvision's _DenseBlock, attn_aug_conv.py:453, :479-481), in place on the slice
 * y[m * ld + c], m < rows, c < C (C % 8 == 0): y = keep ? y / (1 - p) : 0, keep = hash(seed[0], uid, m * C + c) >= p * 2^32 (a
 * counter-based hash: nothing is stored, _bwd regenerates the decisions; seed is a DEVICE int64 so that a replayed hipGraph draws
 * new decisions every step).  S1 / S2 (optional): one row per workgroup of the sums / sums of squares of the result over the
 * slice's channels, at most stat_rows rows (cx_last_stat_rows() reports how many) -- they replace the rows the producing convolution
 * wrote.  _bwd, in place on the gradient slice g with the stored (post-dropout) activations x: g = keep ? (qa g + qb x + qc) /
 * (1 - p) : 0, i.e. the deferred BatchNorm correction of the slice with the keep decision on top; the 3x3 input-gradient and weight-
 * gradient kernels then read g with identity coefficients.                                                                          */
int cx_dropout_slice_fwd(void* y, int ld, int64_t rows, int C, float p, const int64_t* seed, uint32_t uid, float* S1, float* S2,
                         int stat_rows, void* stream);
int cx_dropout_slice_fwd_f32(void* y, int ld, int64_t rows, int C, float p, const int64_t* seed, uint32_t uid, float* S1, float* S2,
                             int stat_rows, void* stream);
int cx_dropout_slice_bwd(void* g, int ldg, const void* x, int ldx, const float* qa, const float* qb, const float* qc, int64_t rows, int C,
                         float p, const int64_t* seed, uint32_t uid, void* stream);
int cx_dropout_slice_bwd_f32(void* g, int ldg, const void* x, int ldx, const float* qa, const float* qb, const float* qc, int64_t rows,
                             int C, float p, const int64_t* seed, uint32_t uid, void* stream);
/* its backward: dz = dout * [out > 0]; S1 += sum dz; S2a += sum dz*(a-mu_a)*r_a; S2b += sum dz*(b-mu_b)*r_b
 * (b / S2b optional).  dz may alias dout.                                                         */
int cx_relu_bwd_stats(const void* dout, const void* out, const void* a, const float* mu_a, const float* r_a, const void* b,
                      const float* mu_b, const float* r_b, void* dz, float* S1, float* S2a, float* S2b, size_t rows, int C,
                      int stat_rows, void* stream);
/* ABI 9.  Residual join forward of a Bottleneck (attn_aug_conv.py:202-209: `out += identity; out = relu(out)` behind bn3) with the
 * residual stream kept as TWO planes: t = relu(a*pa[c] + (b [+ b_lo])*pb[c] + pc[c]) per element of (rows, C) bf16 tensors;
 * out = bf16(t) (the tensor every convolution reads), out_lo = the next 8 mantissa bits of t as a signed byte per element (NULL:
 * single-plane output, the cx_affine2_relu_mask form), mask = sign bits, one byte per 8-channel chunk (NULL: not written).  Side-plane
 * layout (lo planes and every sign-bit plane of this library, also cx_affine2_relu_mask / cx_relu_bwd_stats_mask / CxConv.emask):
 * chunk q (channels 8q..8q+7) of row m is chunk number ((q/8)*rows + m)*8 + q%8 where C % 64 == 0 -- blocks of 64 channels, so a
 * convolution k-step writes whole 128-byte lines -- and m*(C/8) + q otherwise.  b_lo = lo plane of b (NULL:
 * b is a single-plane tensor, e.g. the downsample convolution's raw output).  The reference keeps the stream in fp32; rounded to
 * bf16 at each of resnet152's 50 joins it alone costs 1.0e-2 of the train logits' abs-max -- with the lo plane 2^-17 per join.
 * cx_conv_gemm with CX_PRO_JOIN computes the same bits in the prologue of the next block's conv1.                                 */
int cx_join_fwd(const void* a, const void* b, const int8_t* b_lo, const float* pa, const float* pb, const float* pc, void* out,
                int8_t* out_lo, uint8_t* mask, size_t rows, int C, void* stream);
/* ABI 5: the same pair with the sign of `out` kept as one bit per element (mask: uint8 [rows * C / 8], byte i = the 8 channels
 * of chunk i, bit j = out[8 i + j] > 0), written by the forward and read by the backward INSTEAD of `out` (a quarter of the
 * backward's read bytes).  mask == NULL: the forms above.  `out` may be NULL in the backward when mask is given.            */
int cx_affine2_relu_mask(const void* a, const void* b, const float* pa, const float* pb, const float* pc, void* out, uint8_t* mask,
                         size_t rows, int C, void* stream);
int cx_relu_bwd_stats_mask(const void* dout, const void* out, const uint8_t* mask, const void* a, const float* mu_a,
                           const float* r_a, const void* b, const float* mu_b, const float* r_b, void* dz, float* S1, float* S2a,
                           float* S2b, size_t rows, int C, int stat_rows, void* stream);
/* the same two in the fp32 storage mode (a, b, out, dout, dz fp32; north_star "1e-3 fp32" for the ResNets) */
int cx_affine2_relu_mask_f32(const void* a, const void* b, const float* pa, const float* pb, const float* pc, void* out, uint8_t* mask,
                             size_t rows, int C, void* stream);
int cx_relu_bwd_stats_mask_f32(const void* dout, const void* out, const uint8_t* mask, const void* a, const float* mu_a, const float* r_a,
                               const void* b, const float* mu_b, const float* r_b, void* dz, float* S1, float* S2a, float* S2b,
                               size_t rows, int C, int stat_rows, void* stream);

/* dz (B,H,W,C) bf16 -> dY = dz*pa + x*pb + pc in place (BN0 backward ahead of the stem wgrad)      */
int cx_affine2_inplace(void* dz, const void* x, const float* pa, const float* pb, const float* pc, size_t rows,
                       int C, void* stream);

/* fused optimisers on flat fp32 buffers (torch.optim.Adam / SGD(nesterov) / RMSprop(momentum) as
 * wired at chexpert.py:470, :479, :499)                                                           */
int cx_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int step, float grad_scale, void* stream);
int cx_sgd_nesterov_step(float* p, const float* g, float* buf, size_t n, float lr, float momentum,
                         float weight_decay, int first_step, float grad_scale, void* stream);
int cx_rmsprop_step(float* p, const float* g, float* sq, float* buf, size_t n, float lr, float alpha, float eps,
                    float momentum, float weight_decay, float grad_scale, void* stream);

/* The same updates with the learning rate and the step count read from device memory, so a captured hipGraph of the training
 * step (chexpert.py:159-165) can be replayed while both change: hyper = float[8] {lr, steps_done, sched_kind (0 none,
 * 1 ExponentialLR chexpert.py:500, 2 MultiStepLR :480), gamma, lr_warmup_steps (:165), milestone0, milestone1, base_lr}.
 * cx_optim_tick = "step += 1; if step >= lr_warmup_steps: scheduler.step()" (:165).                                            */
int cx_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, const float* hyper, float beta1, float beta2,
                     float eps, float weight_decay, float grad_scale, void* stream);
int cx_sgd_nesterov_step_dev(float* p, const float* g, float* buf, size_t n, const float* hyper, float momentum,
                             float weight_decay, float grad_scale, void* stream);
int cx_rmsprop_step_dev(float* p, const float* g, float* sq, float* buf, size_t n, const float* hyper, float alpha, float eps,
                        float momentum, float weight_decay, float grad_scale, void* stream);
int cx_optim_tick(float* hyper, void* stream);

/* ---- attention-augmented convolution (AAConv2d, models/attn_aug_conv.py:19-100) --------------------
 * qkv: bf16 (B, H*W, ldq) output of in_proj_qkv (channels [q dk | k dk | v dv], head-major), dk = 20*nh.
 * o: fp32 (B, H*W, dv) attention output BEFORE out_proj; lse: fp32 (B*nh, H*W) log-sum-exp of the logits.
 * Logits include the relative terms of rel_to_abs / relative_logits_1d (:43-63) in closed form; the
 * (B,nh,HW,HW) tensors of the reference are never materialised.                                      */
int cx_aa_attention_fwd(const void* qkv, const float* key_rel_h, const float* key_rel_w, float* o, float* lse, int B, int H, int W,
                        int nh, int dk, int dv, int ldq, void* stream);
/* weights: fp32 (B, nh, H*W, H*W) = softmax(logits) rebuilt from the saved lse -- the tensor the reference leaves in
 * AAConv2d.weights after a forward (:87) and vis_attn reads (chexpert.py:383); visualisation only                   */
int cx_aa_attention_weights(const void* qkv, const float* key_rel_h, const float* key_rel_w, const float* lse, float* weights, int B,
                            int H, int W, int nh, int dk, int dv, int ldq, void* stream);
/* dqkv: fp32 (B, H*W, 2dk+dv) fully written; d_rel_h / d_rel_w (dkh, 2H-1 / 2W-1) ACCUMULATED.  With a workspace of
 * ceil(HW/128)*B*nh * dkh*(2H-1 + 2W-1) floats every query-side workgroup stores its partial tables and they are added in
 * workgroup order (bit-reproducible); scratch == NULL or too small: fp32 atomics                                          */
int cx_aa_attention_bwd(const void* qkv, const float* key_rel_h, const float* key_rel_w, const float* o, const float* d_o,
                        const float* lse, float* dqkv, float* d_rel_h, float* d_rel_w, int B, int H, int W, int nh, int dk, int dv,
                        int ldq, float* scratch, int64_t scratch_floats, void* stream);
/* out_proj (dv x dv, :92) forward into a bf16 channel slice (+stats) and its backward (dY = g*ga+gx*gb+gc).
 * stat_rows > 0: deterministic statistic rows as CxConv.stat_det (row r at stat_sum[r*stat_rstride + c], at most stat_rows rows,
 * cx_last_stat_rows() tells how many); 0: one fp32 atomic per channel and workgroup.  The backward's dW partial tiles go
 * through the CxWgrad.scratch protocol (and the deferred sums) when a workspace is given.                                  */
int cx_aa_outproj_fwd(const float* o, const float* w, void* y, int ldy, float* stat_sum, float* stat_sq, size_t npix, int dv,
                      int stat_rows, int stat_rstride, void* stream);
int cx_aa_outproj_bwd(const void* g, int ldg, const void* gx, int ldgx, const float* ga, const float* gb, const float* gc,
                      const float* o, const float* w, float* d_o, float* dw, size_t npix, int dv, float* scratch, int64_t scratch_floats,
                      void* stream);
/* dst[c] (+)= rows[0*rstride + c] + rows[1*rstride + c] + ... in row order (accumulate != 0: added to dst, else assigned)      */
int cx_rows_reduce(float* dst, const float* rows, int n_rows, int C, int rstride, int accumulate, void* stream);
/* InstanceNorm2d + ReLU ahead of the AAConv2d (:438-439): per-(b,c) sums, per-(b,c) affine+ReLU, and the backward.  The sums are
 * plain stores from one owner per (b, c) (no atomics, no zero-fill needed, bit-reproducible)                                 */
int cx_stats_bc(const void* x, float* sum, float* sq, int B, int HW, int C, int ldx, void* stream);
int cx_affine_relu_bc(const void* x, const float* sc, const float* sh, void* y, int B, int HW, int C, int ldx, void* stream);
int cx_in_relu_bwd(const void* da, const void* x, const float* sc, const float* sh, float* S1, float* S2, void* gout, int B, int HW,
                   int C, int ldx, int ldg, void* stream);
/* fp32 storage-mode twins of the AAConv2d entry points (qkv / activations fp32; the per-query VALU kernels, no MFMA row kernels) */
int cx_aa_attention_fwd_f32(const void* qkv, const float* rel_h, const float* rel_w, float* o, float* lse, int B, int H, int W, int nh,
                        int dk, int dv, int ldq, void* stream);
int cx_aa_attention_weights_f32(const void* qkv, const float* rel_h, const float* rel_w, const float* lse, float* weights, int B, int H,
                            int W, int nh, int dk, int dv, int ldq, void* stream);
int cx_aa_attention_bwd_f32(const void* qkv, const float* rel_h, const float* rel_w, const float* o, const float* d_o, const float* lse,
                        float* dqkv, float* d_rel_h, float* d_rel_w, int B, int H, int W, int nh, int dk, int dv, int ldq,
                        float* scratch, int64_t scratch_floats, void* stream);
int cx_stats_bc_f32(const void* x, float* sum, float* sq, int B, int HW, int C, int ldx, void* stream);
int cx_affine_relu_bc_f32(const void* x, const float* sc, const float* sh, void* y, int B, int HW, int C, int ldx, void* stream);
int cx_aa_outproj_fwd_f32(const float* o, const float* w, void* y, int ldy, float* stat_sum, float* stat_sq, size_t npix, int dv,
                      int stat_rows, int stat_rstride, void* stream);
int cx_aa_outproj_bwd_f32(const void* g, int ldg, const void* gx, int ldgx, const float* ga, const float* gb, const float* gc,
                      const float* o, const float* w, float* d_o, float* dw, size_t npix, int dv, float* scratch, int64_t scratch_floats,
                      void* stream);
int cx_in_relu_bwd_f32(const void* da, const void* x, const float* sc, const float* sh, float* S1, float* S2, void* gout, int B, int HW,
                   int C, int ldx, int ldg, void* stream);
int cx_f32_to_bf16(const float* x, void* y, size_t n, void* stream);

/* ---- EfficientNet blocks (models/efficientnet.py:27-131): depthwise conv, squeeze-excitation, Swish glue ----------
 * depthwise k x k conv on (B,H,W,C) bf16, weights fp32 (C,1,k,k); the preceding BatchNorm+Swish is applied on load
 * when sc/sh are given (sc == NULL: raw input).  Output / statistics as cx_conv_gemm.                          */
int cx_nchw3_to_nhwc8(const float* x, void* y, int B, int H, int W, void* stream);
/* uint8 grey image (npix pixels) -> whitened, channel-expanded (.., 8) bf16 for the EfficientNet stem (chexpert.py:70-72 on the GPU) */
int cx_u8_to_nhwc8(const uint8_t* x, void* y, size_t npix, float mean, float std, void* stream);
/* stat_rows (cx_dwconv_fwd / cx_dwconv_dgrad / cx_bn_lin_bwd_stats / cx_se_act_bwd): > 0 = DETERMINISTIC statistic rows with the
 * CxConv.stat_det convention (row r at sum[r * C + c], at most stat_rows rows, cx_last_stat_rows() tells how many; the consumer
 * cx_bn_coef / cx_bn_bwd_coef sums them in row order), 0 = fp32 atomics into [C] vectors the caller zeroed.
 * cx_dwconv_wgrad: scratch / scratch_floats = the CxWgrad.scratch slab protocol (reproducible sums, deferrable), NULL = atomics.
 * cx_gap_affine_act / cx_se_bwd_reduce: scratch of splits * B * C floats (splits <= 1024 / B) = every pixel split plain-stores its
 * partial per-(image, channel) sum and a second launch adds the rows in order; cx_se_bwd: scratch = slab workspace, one slab per
 * group of images for dW1 / db1 / dW2 / db2.  NULL: fp32 atomics.                                                             */
int cx_dwconv_fwd(const void* x, const float* w, const float* sc, const float* sh, void* y, float* stat_sum, float* stat_sq, int B, int H,
                  int W, int C, int k, int stride, int pad, int stat_rows, void* stream);
/* dY = g*ga + g2*gb + gc;  dz = (sum_t dY w) * swish'(x*sc+sh), S1 += dz, S2 += dz*(x-mean)*rstd (sc==NULL: dz = sum) */
int cx_dwconv_dgrad(const void* g, const void* g2, const float* ga, const float* gb, const float* gc, const float* w, const void* x,
                    const float* sc, const float* sh, const float* mean, const float* rstd, void* dz, float* S1, float* S2, int B, int H,
                    int W, int C, int k, int stride, int pad, int accumulate, int stat_rows, void* stream);
int cx_dwconv_wgrad(const void* g, const void* g2, const float* ga, const float* gb, const float* gc, const void* x, const float* sc,
                    const float* sh, float* dw, int B, int H, int W, int C, int k, int stride, int pad, float* scratch,
                    int64_t scratch_floats, void* stream);
/* pooled[b][c] = mean_hw act(x*sc+sh) (act 0 none / 1 relu / 2 swish): SELayer pool (:69), head pool (:162)      */
int cx_gap_affine_act(const void* x, const float* sc, const float* sh, float* pooled, int B, int HW, int C, int act, float* scratch,
                      int64_t scratch_floats, void* stream);
/* SELayer FCs (:70-73): h1 = W1 pooled + b1, s = sigmoid(W2 swish(h1) + b2); and their backward                */
int cx_se_fwd(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, float* h1, float* s, int B, int C,
              int R, void* stream);
/* ABI 10.  SELayer squeeze + excitation (:69-73) as TWO launches: cx_gap_affine_act's per-split partial means are added by the
 * excitation kernel itself (in row order: reproducible), which also leaves pooled[b][c] for the backward pass -- the reduce launch
 * between the two is gone (33 launches per EfficientNet-B4 step).  scratch as for cx_gap_affine_act; NULL / too small: the
 * three-launch form with atomics.                                                                                        */
int cx_gap_se_fwd(const void* x, const float* sc, const float* sh, float* pooled, const float* w1, const float* b1, const float* w2,
                  const float* b2, float* h1, float* s, int B, int HW, int C, int R, int act, float* scratch, int64_t scratch_floats,
                  void* stream);
int cx_gap_se_fwd_f32(const void* x, const float* sc, const float* sh, float* pooled, const float* w1, const float* b1, const float* w2,
                      const float* b2, float* h1, float* s, int B, int HW, int C, int R, int act, float* scratch, int64_t scratch_floats,
                      void* stream);
int cx_se_bwd(const float* ds, const float* s, const float* h1, const float* pooled, const float* w1, const float* w2, float* dw1,
              float* db1, float* dw2, float* db2, float* dpooled, int B, int C, int R, float* scratch, int64_t scratch_floats,
              void* stream);
/* ABI 10.  cx_se_bwd_reduce + cx_se_bwd without the reduce launch between them: the split rows of ds[b][c] = sum_hw du * swish(x*sc+sh)
 * (rows_scratch: splits * B * C floats, as for cx_se_bwd_reduce) are added, in row order, by the first FC pass.  `ds` (B x C) is written
 * only when that form cannot run (no row / slab workspace: the four-launch sequence with what workspaces there are).            */
int cx_se_bwd_fused(const void* du, const void* x, const float* sc, const float* sh, float* ds, const float* s, const float* h1,
                    const float* pooled, const float* w1, const float* w2, float* dw1, float* db1, float* dw2, float* db2, float* dpooled,
                    int B, int HW, int C, int R, float* rows_scratch, int64_t rows_floats, float* scratch, int64_t scratch_floats,
                    void* stream);
int cx_se_bwd_fused_f32(const void* du, const void* x, const float* sc, const float* sh, float* ds, const float* s, const float* h1,
                        const float* pooled, const float* w1, const float* w2, float* dw1, float* db1, float* dw2, float* db2, float* dpooled,
                        int B, int HW, int C, int R, float* rows_scratch, int64_t rows_floats, float* scratch, int64_t scratch_floats,
                        void* stream);
/* u = swish(x*sc+sh) * s[b][c] (s NULL: no SE scaling)                                                          */
int cx_scale_act_bc(const void* x, const float* sc, const float* sh, const float* s, void* u, int B, int HW, int C, void* stream);
int cx_se_bwd_reduce(const void* du, const void* x, const float* sc, const float* sh, float* ds, int B, int HW, int C, float* scratch,
                     int64_t scratch_floats, void* stream);
/* dz = (du*s[b][c] + dpooled[b][c]/HW) * swish'(x*sc+sh) + BN backward sums (du or dpooled may be NULL)          */
int cx_se_act_bwd(const void* du, const void* x, const float* sc, const float* sh, const float* mean, const float* rstd, const float* s,
                  const float* dpooled, void* dz, float* S1, float* S2, int B, int HW, int C, int stat_rows, void* stream);
int cx_bn_lin_bwd_stats(const void* g, const void* y, const float* mean, const float* rstd, float* S1, float* S2, size_t rows, int C,
                        int stat_rows, void* stream);
/* out = s*(a*pa + pc) + b*pb (b may be NULL): projection BatchNorm output, DropConnect, skip (:105-110).  sample_scale
 * (optional, one float per image, rows_per_sample rows each) is the DropConnect mask / keep probability (:44-51)     */
int cx_affine2_out(const void* a, const void* b, const float* pa, const float* pb, const float* pc, const float* sample_scale,
                   size_t rows_per_sample, void* out, size_t rows, int C, void* stream);
/* out[row][:] = sample_scale[row / rows_per_sample] * g[row][:]: gradient entering a DropConnect-ed branch            */
int cx_scale_rows(const void* g, const float* sample_scale, size_t rows_per_sample, void* out, size_t rows, int C, void* stream);
/* out[i] in {0, 1/keep_prob}: counter-based (splitmix64 of seed and index) Bernoulli mask for Dropout (:170) and DropConnect */
/* fp32 storage-mode twins of the EfficientNet entry points above (activations fp32; the generic kernels, no tiled fast paths) */
int cx_nchw3_to_nhwc8_f32(const float* x, void* y, int B, int H, int W, void* stream);
int cx_u8_to_nhwc8_f32(const uint8_t* x, void* y, size_t npix, float mean, float std, void* stream);
int cx_dwconv_fwd_f32(const void* x, const float* w, const float* sc, const float* sh, void* y, float* stat_sum, float* stat_sq, int B, int H,
                  int W, int C, int k, int stride, int pad, int stat_rows, void* stream);
int cx_dwconv_dgrad_f32(const void* g, const void* g2, const float* ga, const float* gb, const float* gc, const float* w, const void* x,
                    const float* sc, const float* sh, const float* mean, const float* rstd, void* dz, float* S1, float* S2, int B, int H,
                    int W, int C, int k, int stride, int pad, int accumulate, int stat_rows, void* stream);
int cx_dwconv_wgrad_f32(const void* g, const void* g2, const float* ga, const float* gb, const float* gc, const void* x, const float* sc,
                    const float* sh, float* dw, int B, int H, int W, int C, int k, int stride, int pad, float* scratch,
                    int64_t scratch_floats, void* stream);
int cx_gap_affine_act_f32(const void* x, const float* sc, const float* sh, float* pooled, int B, int HW, int C, int act, float* scratch,
                      int64_t scratch_floats, void* stream);
int cx_scale_act_bc_f32(const void* x, const float* sc, const float* sh, const float* s, void* u, int B, int HW, int C, void* stream);
int cx_bn_lin_bwd_stats_f32(const void* g, const void* y, const float* mean, const float* rstd, float* S1, float* S2, size_t rows, int C,
                        int stat_rows, void* stream);
int cx_se_bwd_reduce_f32(const void* du, const void* x, const float* sc, const float* sh, float* ds, int B, int HW, int C, float* scratch,
                     int64_t scratch_floats, void* stream);
int cx_se_act_bwd_f32(const void* du, const void* x, const float* sc, const float* sh, const float* mean, const float* rstd, const float* s,
                  const float* dpooled, void* dz, float* S1, float* S2, int B, int HW, int C, int stat_rows, void* stream);
int cx_affine2_out_f32(const void* a, const void* b, const float* pa, const float* pb, const float* pc, const float* sample_scale,
                   size_t rows_per_sample, void* out, size_t rows, int C, void* stream);
int cx_scale_rows_f32(const void* g, const float* sample_scale, size_t rows_per_sample, void* out, size_t rows, int C, void* stream);
int cx_dropout_mask(float* out, size_t n, float keep_prob, unsigned long long seed, void* stream);
/* ABI 7: the same mask with seed = base + step[0] * 1000003 assembled on the device, and the one-element counter bump that goes  */
/* with it -- a captured training step (hipGraph) then draws new Dropout / DropConnect masks at every replay                  */
int cx_dropout_mask_dev(float* out, size_t n, float keep_prob, unsigned long long base, const unsigned long long* step, void* stream);
int cx_counter_add(unsigned long long* counter, unsigned long long inc, void* stream);
int cx_mul_f32(const float* a, const float* b, float* out, size_t n, void* stream);
int cx_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C, int N, void* stream);

/* Grad-CAM as the reference code executes it (chexpert.py:260-303):
 * cam[b][p] = relu(sum_c w[c]*f(x*scale+shift)) with class-independent w[c] = mean_b pooled[b][c]*B/n_cls,
 * then per-image (t-min)/(max-min+1e-5) and bilinear upsampling with align_corners=True.
 * inner_relu = 1: f = relu (DenseNet: the hook on features.norm5 ends up holding the in-place ReLU'd tensor, :468);
 * inner_relu = 0: f = identity (ResNet layer4 output, already >= 0, :484; EfficientNet head[1] = BatchNorm output, :498) */
int cx_gradcam_map(const void* x, const float* scale, const float* shift, const float* w, float* cam, int B, int HW, int C,
                   int ldx, int inner_relu, void* stream);
int cx_cam_norm_upsample(const float* cam, float* out, int B, int h, int w, int H, int W, void* stream);

/* stat_rows (cx_bnrelu_maxpool_fwd / _bwd, cx_gap_relu_bn_bwd, cx_unpool2_mask, cx_relu_bwd_stats): 0 = the statistics are added to the single
 * copy S1 / S2 [C] with atomics; > 0 = deterministic rows: the launch uses at most stat_rows workgroups (cx_gap_relu_bn_bwd: one row
 * per image, B <= stat_rows) and plain-stores row r at S[r*C + c]; cx_last_stat_rows() gives the row count for the consumer.   */

/* Brightness / contrast jitter of the decoded grey images (B, HW) uint8 on the GPU: the "+ data aug" of the reference's
 * `_data_aug` README rows = ColorJitter(brightness=0.25, contrast=0.25) of explore_data.ipynb cell 6, torchvision tensor
 * semantics on uint8 (y = trunc(clamp(b*x)); y = trunc(clamp(c*x + (1-c)*mean(x)))), per-image factors and order (0 = brightness
 * first) drawn by the caller.  HW % 16 == 0, HW <= 150 KiB (the image is parked in LDS between the two passes).               */
int cx_u8_jitter(const uint8_t* x, uint8_t* y, int B, int HW, const float* brightness, const float* contrast, const int* order,
                 void* stream);

/* ---- fp32 storage mode (CX_DT_F32): the element-wise kernels of the DenseNet path with fp32 activation tensors (same arguments,
 * `const void*` tensors are fp32, pitches in elements), the fp32 weight table ([tap][O][I] fp32; descriptors with stem = 1 give
 * [49][O][4]) and the fp32 image layouts.  cx_conv_gemm / cx_conv_wgrad take CxConv.dtype / CxWgrad.dtype = CX_DT_F32.          */
int cx_pack_weights_table_f32(const float* flat, float* packed, const CxPackDesc* table_dev, int n_desc, void* stream);
int cx_nchw3_to_nhwc4_f32(const float* x, float* y, int B, int H, int W, void* stream);
int cx_u8_to_nhwc4_f32(const uint8_t* x, float* y, size_t npix, float mean, float std, void* stream);
int cx_bnrelu_maxpool_fwd_f32(const void* x, const float* scale, const float* shift, void* y, uint8_t* argmax,
                          float* stat_sum, float* stat_sq, int B, int H, int W, int C, int ldy, int stat_rows, void* stream);
int cx_bnrelu_maxpool_bwd_f32(const void* x, const float* scale, const float* shift, const float* mean,
                          const float* rstd, const uint8_t* argmax, const void* g, const void* gx, const float* ga,
                          const float* gb, const float* gc, void* dz, float* S1, float* S2, int B, int H, int W, int C,
                          int ldg, int ldgx, int stat_rows, void* stream);
int cx_head_fwd_f32(const void* x, const float* scale, const float* shift, const float* w, const float* bias,
                float* pooled, float* logits, int B, int HW, int C, int ldx, int n_classes, void* stream);
int cx_gap_relu_bn_bwd_f32(const float* dpooled, const void* x, const float* scale, const float* shift,
                       const float* mean, const float* rstd, const float* e_scale, void* g, float* S1, float* S2,
                       int B, int HW, int C, int ldx, int ldg, int stat_rows, void* stream);
int cx_unpool2_mask_f32(const void* d, const void* x, const float* sc, const float* sh, const float* mean,
                    const float* rstd, const float* e_scale, void* g, float* S1, float* S2, int B, int H, int W,
                    int C, int ldd, int ldx, int ldg, int stat_rows, void* stream);

/* utilities */
int cx_fill_f32(float* p, float v, size_t n, void* stream);
/* ABI 9: dst = src as a 16-byte-per-lane grid-stride copy (bytes % 16 == 0, both 16-byte aligned): the measured stream rate of the
 * device that bench.py reports next to the 8 TB/s specification ("a stream-copy micro-benchmark"); no
 * reference counterpart */
int cx_copy_stream(const void* src, void* dst, size_t bytes, void* stream);
/* y (B,C,H,W) fp32 = act(x*scale + shift) of a bf16 NHWC tensor (scale/shift NULL: identity; relu != 0: ReLU): the tensor a
 * forward hook on the reference's hook targets receives (features.norm5 / layer4 / head[1], chexpert.py:468, :484, :498)      */
int cx_affine_to_f32_nchw(const void* x, const float* scale, const float* shift, int relu, float* y, int B, int H, int W, int C,
                          int ldx, void* stream);
int cx_bf16_to_f32_nchw(const void* x, float* y, int B, int H, int W, int C, int ldx, void* stream);

#ifdef __cplusplus
}
#endif
#endif
