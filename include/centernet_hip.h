/* libcenternet_hip.so — C ABI of the MI355X-native CenterNet hot path.
 *
 * The reference (tteepe/CenterNet-pytorch-lightning) has no FFI of its own: its hot path is
 * ATen ops + the DCNv2 CUDA extension.  Each entry point below replaces the ATen/DCNv2 call the
 * cited reference line makes; Python host code (centernet-pytorch-lightning_amd/_hip.py) binds
 * them with ctypes.  Conventions:
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless noted
 *   - activations are NHWC in `dtype` (CN_F32 | CN_BF16); public tensors (images, head maps,
 *     targets, parameters, gradients) are NCHW fp32 exactly like the reference
 *   - nothing allocates, frees or synchronises: work is enqueued on `stream` (hipStream_t),
 *     so every call is hipGraph-capturable; scratch comes from the caller (`ws`, `ws_bytes`)
 *   - returns 0 on success, <0 for a rejected argument/shape (cn_last_error() explains),
 *     >0 = hipError_t of the failed launch
 */
#ifndef CENTERNET_HIP_H
#define CENTERNET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CN_F32 0
#define CN_BF16 1

#define CN_OK 0
#define CN_EINVAL (-1)
#define CN_EUNSUPPORTED (-2)
#define CN_EWORKSPACE (-3)

int cn_version(void);
const char* cn_last_error(void); /* thread-local, host pointer */

/* ---- layout boundary (NCHW fp32 public tensors <-> NHWC activations) -------------------- */
/* dst[n,h,w,c] = src[n,c,h,w] (c < C), zero for C <= c < Cpad */
int cn_nchw_to_nhwc(const float* src, void* dst, int N, int C, int H, int W, int Cpad, int dtype, void* stream);
/* dst[n,c,h,w] = src[n,h,w,c]; src row pitch = ld channels */
int cn_nhwc_to_nchw(const void* src, float* dst, int N, int C, int H, int W, int ld, int dtype, void* stream);
/* dst[p, dst_off + c] = src[p, src_off + c], c < nch  (torch.cat / slicing along channels: pose_dla_dcn.py:182) */
int cn_copy_channels(const void* src, int src_ld, int src_off, void* dst, int dst_ld, int dst_off,
                     int64_t npix, int nch, int dtype, void* stream);
/* out = a + b (pose_dla_dcn.py:488 `layers[i] + layers[i-1]`); accumulate: out += a */
int cn_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream);
/* p[0 .. nbytes) = 0 (any alignment).  Replaces the ATen fills (`torch.zeros`, `.zero_()`) in front of the split-K weight-gradient
 * accumulators, the DCN far-sample buffers and the flat gradient buffer (engine.FlatAdam.zero_grad): same position in the stream,
 * so the accumulator is still L2-resident for the atomics that follow (docs/NEGATIVE_RESULTS.md), but the step holds no at::native launch. */
int cn_zero(void* p, int64_t nbytes, void* stream);
/* measurement aid (tools/tail_stamps.py): *dst = the device's constant-rate wall clock (100 MHz ticks) when this launch runs */
int cn_stamp(int64_t* dst, void* stream);
/* measurement aid: while buf != NULL the k-th cn_zero launch (host order) also stores the wall clock at which it starts in buf[k] */
int cn_zero_stamps(int64_t* buf, int n);
int cn_zero_stamps_used(void);
int cn_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);

/* ---- dense convolution engine (replaces nn.Conv2d / nn.ConvTranspose2d + their backward) -- */
/* Pack a 4-D fp32 parameter W[A][B][KH][KW] (t = kh*KW+kw) into the GEMM layout the engine reads:
 *   mode 0: Wp[b][t*inner_pad + a] = W[a][b][t]   rows = B   -- Conv2d data-gradient / ConvTranspose2d forward
 *   mode 1: Wp[a][t*inner_pad + b] = W[a][b][t]   rows = A   -- Conv2d forward / ConvTranspose2d data-gradient
 *   mode 2: Wp[t*B + b][a]         = W[a][b][t]   rows = KH*KW*B, row length inner_pad -- DCN column gradient
 * rows are zero padded to rows_pad (multiple of 32), the inner (reduction) channel count to inner_pad (multiple
 * of 16).  row_scale (nullable, fp32[rows]) multiplies each row (eval-mode BN folding). */
int cn_pack_weight(const float* w, void* wp, int A, int B, int KH, int KW, int mode, int rows_pad, int inner_pad,
                   const float* row_scale, int dtype, void* stream);
/* All weight packings of a training step in ONE launch (the per-layer calls were ~160 tiny launches per step).
 * table: device array of n_entries records of 10 int64:
 *   { w (fp32 device pointer), wp (device pointer), A, B, KH*KW, mode, rows_pad, inner_pad, first_block, AA | BB << 8 | tiles_b << 16 }
 * A record is packed by tiles of AA x BB (a, b) pairs, AA | BB << 8 = cn_pack_weight_tile(taps, mode) (0: taps > 64, pack that weight
 * with cn_pack_weight): tiles_a = ceil(a_range / AA), tiles_b = ceil(b_range / BB) over the PADDED ranges (mode 1: a < rows_pad,
 * b < inner_pad; mode 0: a < inner_pad, b < rows_pad; mode 2: a < inner_pad, b < B), blocks = tiles_a * tiles_b (+ 1 for mode 2 when
 * rows_pad > KH*KW*B); first_block = the running sum of blocks over the preceding records, n_blocks its total.  block_record
 * (nullable, int32[n_blocks] on the device): the record index of every workgroup (otherwise each workgroup searches the table). */
int cn_pack_weight_tile(int taps, int mode);
int cn_pack_weight_batch(const void* table, int n_entries, int n_blocks, const int* block_record, int dtype, void* stream);
/* inverse of mode 1 for gradients: dw[a][b][t] (+)= dwp[a][t*inner_pad + b] (fp32 -> fp32); accumulate != 0 adds into dw
 * (used to deposit gradients straight into the flat gradient buffer from a side stream) */
int cn_unpack_wgrad(const float* dwp, float* dw, int A, int B, int KH, int KW, int inner_pad, int accumulate, void* stream);
/* column block of a 1x1 weight gradient: dw[a][b] (+)= dwp[a][b], b < B, with dw pointing at the block's first column and
 * dw_ld the full row length (the per-source weight gradients of cn_conv1x1_cat_fwd) */
int cn_unpack_wgrad_cols(const float* dwp, float* dw, int A, int B, int inner_pad, int dw_ld, int accumulate, void* stream);

/* ---- per-call hooks --------------------------------------------------------------------------
 * Optional extras of ONE call, handed over explicitly (the `_h` twin of an entry point takes `cn_hooks* hooks` in front of the
 * stream; NULL or a zeroed struct = the plain entry point).  Nothing is remembered between calls: the library keeps no armed
 * state (SURVEY 8b "Threading": no mutable globals except kernel handles and the thread-local error string).  The struct
 * lives in HOST memory; `bn_taken` / `bnb_taken` are written by the call.
 *   pre_ss / pre_C / pre_relu   input pre-affine (cn_conv2d_fwd_h, cn_conv2d_wgrad_h): x is the RAW output of the previous
 *        convolution and the kernel uses x' = bf16(fma(x, pre_ss[c], pre_ss[C + c])) (pre_relu: max(., 0)) — the training-mode
 *        BN (+ ReLU) of pose_dla_dcn.py:283-296 applied by the consumer, bit-identical to the tensor cn_bn_train_fwd_sink
 *        would have stored, zero padding applied to x'.  pre_ss = fp32 [2][C] scale | shift in device memory
 *        (cn_bn_finalize_sink writes it).  Only the 16-input-channel bf16 3x3 kernels have the hook; any other shape returns
 *        CN_EUNSUPPORTED (no silent fallback on the raw tensor).
 *   bn_part / bn_slots / bn_C -> bn_taken   BatchNorm statistics of the OUTPUT from the producer's epilogue (cn_conv2d_fwd_h,
 *        cn_conv1x1_cat_fwd_h, cn_dcn_fwd_h, cn_stem_conv_fwd_h; pose_dla_dcn.py:55-68, 435-454, msra_resnet.py:29-58): when the
 *        kernel the call dispatches to has the hook (bf16, y_ld == bn_C), every workgroup adds per-channel sum / sum of squares
 *        of the values it stores (after rounding to bf16) to row (workgroup % bn_slots) of bn_part[bn_slots][2][bn_C] (fp32
 *        atomics; all-zero on entry, bn_slots <= 1024: cn_bn_stats_slots()) and bn_taken = 1; otherwise bn_part is untouched and
 *        bn_taken = 0 (the caller falls back to cn_bn_train_fwd, which reads x itself).
 *   bnb_* -> bnb_taken   BN-BACKWARD statistics from the kernel that PRODUCES the gradient (cn_conv2d_fwd_h with transposed
 *        != 0): its output y is the gradient w.r.t. the output of a training-mode BN (+ ReLU: bnb_relu) with input bnb_x (NHWC,
 *        pitch bnb_C == y_ld) and saved statistics bnb_stats = fp32 [4][C] mean | invstd | scale | shift; a kernel with the hook
 *        (the 16-channel bf16 data-gradient kernels) adds per channel sum g and sum g * xhat of the values it stores to
 *        bnb_part[bnb_slots][2][bnb_C] (all-zero on entry) — what cn_bn_bwd_stats computes in a pass of its own — and
 *        bnb_taken = 1; otherwise 0 and the sink is untouched.
 *   wgrad_blocks   number of workgroups the split-K weight-gradient kernels of this call spread over (<= 0: the default, 1536
 *        = ~6 per CU).  A host that runs them on a second stream beside the data-gradient chain lowers it (~160) so they stay in
 *        the background.  The scratch size of cn_conv2d_wgrad_direct depends on it: ask cn_conv2d_wgrad_direct_bytes_h
 *        with the same value (a launch whose grid needs more scratch than it was given returns CN_EWORKSPACE). */
typedef struct cn_hooks {
    const float* pre_ss; int32_t pre_C; int32_t pre_relu;
    float* bn_part; int32_t bn_slots; int32_t bn_C;
    float* bnb_part; int32_t bnb_slots; int32_t bnb_C; const void* bnb_x; const float* bnb_stats; int32_t bnb_relu;
    int32_t bn_taken; int32_t bnb_taken;
    int32_t wgrad_blocks;
} cn_hooks;
/* sizeof(cn_hooks) as this library was built: a binding that mirrors the struct (ctypes.Structure, cgo, JNI) checks its own size
 * against it once at load time */
size_t cn_hooks_size(void);

/* Implicit-GEMM convolution, NHWC.  y[n,oh,ow,co] = act(bias[co] + res[..] + sum_{t,ci} xg * Wp[co][t*Ci+ci])
 *   transposed == 0: xg = x[n, oh*stride - pad + kh, ow*stride - pad + kw, ci]       (nn.Conv2d)
 *   transposed == 1: xg = x[n, (oh + pad - kh)/stride, (ow + pad - kw)/stride, ci]   (nn.ConvTranspose2d /
 *                    data-gradient of a strided conv), taps that do not divide are skipped
 * x pitch = x_ld channels, y pitch = y_ld, residual pitch = res_ld (same dtype as y); bias fp32 nullable.
 * out_dtype = dtype, or CN_F32 to keep a bf16-computed result in fp32 (DCN offsets / mask logits).
 * relu: 0 none, 1 ReLU, 2 = ReLU-BACKWARD mask: `residual` is not added, the result is zeroed where residual <= 0
 *   (fuses the hidden ReLU's backward of the heads, heads.py:11-17, into the 1x1 conv's data gradient).
 * Channels Co .. y_ld-1 of y (the activation's zero padding) are written as zeros.  Requires Ci % 16 == 0. */
int cn_conv2d_fwd(const void* x, const void* wp, const float* bias, const void* residual, void* y,
                  int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co, int y_ld, int res_ld,
                  int KH, int KW, int stride, int pad, int transposed, int relu, int dtype, int out_dtype, void* stream);
int cn_conv2d_fwd_h(const void* x, const void* wp, const float* bias, const void* residual, void* y,
                    int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co, int y_ld, int res_ld,
                    int KH, int KW, int stride, int pad, int transposed, int relu, int dtype, int out_dtype, cn_hooks* hooks, void* stream);
/* 1x1 conv with a tiny contraction (K <= 4 real input channels; x pitch x_ld, weights packed with 16-wide rows like
 * cn_pack_weight): the data gradient of the 1- and 2-channel heads' last conv (heads.py:9-15, backwards).  relu = 2 masks the
 * result with residual > 0 (fused ReLU backward), relu = 0 with residual adds it.  bf16, y_ld == Co. */
int cn_conv1x1_smallk(const void* x, const void* wp, const void* residual, void* y, int64_t P, int K, int x_ld, int Co,
                      int y_ld, int res_ld, int relu, int dtype, void* stream);
/* which kernel cn_conv2d_fwd dispatches to: BN*1000 + BK = `conv_igemm_kernel<T,BN,BK>`;
 * 3000000 + BN*1000 + CK = the 3x3/s1/p1 halo-tile kernel `conv3x3s1_kernel<T,BN,CK>` */
int cn_conv2d_variant(int Ci, int Co, int KH, int KW, int stride, int pad, int dtype);
/* Weight gradient: dwp[co][t*Ci+ci] += sum_{n,oh,ow} dy[n,oh,ow,co] * x[n, oh*stride-pad+kh, ow*stride-pad+kw, ci]
 * dwp is fp32 [Co_pad32][KH*KW*Ci] and must be zeroed by the caller (split-K uses atomics).  db (nullable,
 * fp32[Co], zeroed) accumulates the bias gradient. */
int cn_conv2d_wgrad(const void* x, const void* dy, float* dwp, float* db,
                    int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co, int dy_ld,
                    int KH, int KW, int stride, int pad, int dtype, void* stream);
int cn_conv2d_wgrad_h(const void* x, const void* dy, float* dwp, float* db,
                      int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co, int dy_ld,
                      int KH, int KW, int stride, int pad, int dtype, cn_hooks* hooks, void* stream);
/* The same gradient straight into the PARAMETER layout dw[Co][Ci][KH][KW] (fp32, accumulate != 0 adds — e.g. into the flat
 * gradient buffer), for the shapes whose kernel has the slab form (bf16, 3x3 / stride 1 or 2 / pad 1, Ci > 16): every workgroup
 * stores its split-K partial as a private slab in `ws` and one reduction launch sums them in a fixed order — no fp32 atomics
 * (device-scope atomics execute at the memory side: 0.25 us per workgroup flush, serialised chip-wide), no pre-zeroed packed
 * gradient, no cn_unpack_wgrad launch.  cn_conv2d_wgrad_direct_bytes = scratch size, 0 when the shape is not handled (use
 * cn_conv2d_wgrad + cn_unpack_wgrad).  db: as in cn_conv2d_wgrad (accumulated into, nullable). */
size_t cn_conv2d_wgrad_direct_bytes(int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co, int dy_ld, int KH, int KW,
                                    int stride, int pad, int dtype);
int cn_conv2d_wgrad_direct(const void* x, const void* dy, float* dw, float* db, int accumulate, void* ws, size_t ws_bytes,
                           int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co, int dy_ld,
                           int KH, int KW, int stride, int pad, int dtype, void* stream);
size_t cn_conv2d_wgrad_direct_bytes_h(int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co, int dy_ld, int KH, int KW,
                                      int stride, int pad, int dtype, int wgrad_blocks);
int cn_conv2d_wgrad_direct_h(const void* x, const void* dy, float* dw, float* db, int accumulate, void* ws, size_t ws_bytes,
                             int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co, int dy_ld,
                             int KH, int KW, int stride, int pad, int dtype, cn_hooks* hooks, void* stream);
/* bias gradient alone: db[c] += sum_p dy[p][c] (db accumulated into; dy_ld a vector multiple) */
int cn_colsum(const void* dy, float* db, int64_t P, int Co, int dy_ld, int dtype, void* stream);

/* Stem convolution for tiny Ci (3 input channels, 7x7): direct kernel on the NCHW fp32 image
 * (msra_resnet.py:110, pose_dla_dcn.py:282).  w is the raw fp32 parameter [Co,Ci,KH,KW]; scale / bias (nullable fp32[Co])
 * and relu fold an eval-mode BN + ReLU into the epilogue: y = act(fma(conv, scale, bias)). */
int cn_stem_conv_fwd(const float* x_nchw, const float* w, const float* scale, const float* bias, void* y, int N, int Ci, int H, int W, int Co,
                     int KH, int KW, int stride, int pad, int OH, int OW, int relu, int dtype, void* stream);
int cn_stem_conv_fwd_h(const float* x_nchw, const float* w, const float* scale, const float* bias, void* y, int N, int Ci, int H, int W, int Co,
                       int KH, int KW, int stride, int pad, int OH, int OW, int relu, int dtype, cn_hooks* hooks, void* stream);
/* cn_stem_conv_wgrad through the training-mode BN (+ ReLU) behind the stem: dy = gradient w.r.t. the BN output, y_raw = the stem's raw
 * output, coef from cn_bn_bwd_coef_sink; the BN input gradient is formed on load and never stored (bf16, 7x7 / pad 3, Co % 16 == 0). */
int cn_stem_conv_wgrad_bn(const float* x_nchw, const void* dy, const void* y_raw, const float* coef, float* dw, int N, int Ci, int H,
                          int W, int Co, int KH, int KW, int stride, int pad, int OH, int OW, int relu, int dtype, void* stream);
int cn_stem_conv_wgrad(const float* x_nchw, const void* dy, float* dw, int N, int Ci, int H, int W, int Co,
                       int KH, int KW, int stride, int pad, int OH, int OW, int dtype, void* stream);
int cn_stem_conv_wgrad_bn_h(const float* x_nchw, const void* dy, const void* y_raw, const float* coef, float* dw, int N, int Ci, int H,
                            int W, int Co, int KH, int KW, int stride, int pad, int OH, int OW, int relu, int dtype, cn_hooks* hooks, void* stream);
int cn_stem_conv_wgrad_h(const float* x_nchw, const void* dy, float* dw, int N, int Ci, int H, int W, int Co,
                         int KH, int KW, int stride, int pad, int OH, int OW, int dtype, cn_hooks* hooks, void* stream);

/* ---- batch norm (nn.BatchNorm2d, momentum 0.1) + ReLU + residual add ---------------------- */
size_t cn_bn_workspace_bytes(int64_t npix, int C);
/* Training-mode conv + BN (pose_dla_dcn.py:55-68, 435-454; msra_resnet.py:29-58): the kernel that PRODUCES the BN input also
 * accumulates the batch statistics — cn_hooks.bn_part of the producing call (cn_conv2d_fwd_h, cn_conv1x1_cat_fwd_h, cn_dcn_fwd_h,
 * cn_stem_conv_fwd_h).  cn_bn_train_fwd_stats = cn_bn_train_fwd without the statistics pass over x: finalize from `part` (handed
 * back all-zero) + the apply pass.  The input pre-affine of the consumer (cn_hooks.pre_ss) and the BN-backward statistics of a
 * data-gradient launch (cn_hooks.bnb_part) are described at cn_hooks. */
/* BN backward for a consumer that applies it on load (the stem's weight gradient, cn_stem_conv_wgrad_bn): cn_bn_bwd_stats = the
 * statistics pass of cn_bn_train_bwd_sink alone; cn_bn_bwd_coef_sink = totals of the sink -> dgamma / dbeta and coef fp32 [5][C]
 * (ca | cp | cq | sc | sh: g = relu ? (fma(x, sc, sh) > 0 ? dy : 0) : dy, dx = fma(ca, g, fma(cp, x, cq))). */
/* cn_bn_train_bwd_apply = the apply half of cn_bn_train_bwd_sink alone, on a sink that is already filled (cn_hooks.bnb_part of the
 * data-gradient call that produced dy, or cn_bn_bwd_stats). */
int cn_bn_train_bwd_apply(const void* dy, const void* x, const void* y, const float* gamma, const float* save_mean,
                          const float* save_invstd, const float* scale_shift, void* dx, void* dres, const void* dres_acc,
                          float* dgamma, float* dbeta, int accumulate, const float* sink, int slots, float* clear, int64_t clear_n,
                          int64_t npix, int C, int relu, int dtype, void* stream);
int cn_bn_bwd_stats(const void* dy, const void* x, const void* y, const float* save_mean, const float* save_invstd,
                    const float* scale_shift, float* sink, int slots, int64_t npix, int C, int relu, int dtype, void* stream);
int cn_bn_bwd_coef_sink(const float* sink, int slots, const float* gamma, const float* save_mean, const float* save_invstd,
                        const float* scale_shift, float* dgamma, float* dbeta, int accumulate, float* coef, float* clear,
                        int64_t clear_n, int64_t npix, int C, void* stream);
int cn_bn_stats_slots(void);
/* cn_bn_finalize_sink: the statistics half of cn_bn_train_fwd_stats alone (mean / invstd / running stats / scale | shift from `part`,
 * handed back all-zero) — no apply pass: the consumer applies the affine map on load (cn_hooks.pre_ss). */
int cn_bn_finalize_sink(float* part, int slots, const float* gamma, const float* beta, float* running_mean, float* running_var,
                        float* save_mean, float* save_invstd, float* save_scale_shift, int64_t npix, int C, float momentum,
                        float eps, void* stream);
int cn_bn_train_fwd_stats(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                          float* save_scale_shift, float* part, int slots, int64_t npix, int C, float momentum, float eps,
                          int relu, int dtype, void* ws, size_t ws_bytes, void* stream);
/* cn_bn_train_fwd_stats in ONE launch: the apply kernel reduces `part` (slots * 2 * C floats from L2 per workgroup) in its prologue
 * and its first workgroup column writes the saved / running statistics — no finalize launch in front of the pass.  `part` is NOT
 * cleared (no kernel can clear what its own workgroups are still reading); instead the launch zeroes `clear` (nullable, clear_n
 * floats): another sink, which an EARLIER launch on the same stream consumed.  The caller keeps the chain: every sink consumed by
 * one of the *_sink entry points is handed as `clear` to the next one (ops.BnStats). */
int cn_bn_train_fwd_sink(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                         float* save_scale_shift, const float* part, int slots, float* clear, int64_t clear_n,
                         int64_t npix, int C, float momentum, float eps, int relu, int dtype, void* stream);
/* training forward: batch statistics over npix rows; y = act(gamma*(x-mean)*invstd + beta [+ residual]);
 * updates running stats (unbiased var) in place; saves mean / invstd for backward.  save_scale_shift (nullable,
 * fp32 [2][C]) receives the per-channel affine the apply pass used (y = act(fma(x, scale, shift) [+ residual])). */
int cn_bn_train_fwd(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                    float* save_scale_shift, int64_t npix, int C, float momentum, float eps, int relu, int dtype,
                    void* ws, size_t ws_bytes, void* stream);
/* y = act(x*scale[c] + shift[c] [+ residual])  (eval-mode BN when it cannot be folded into a conv) */
int cn_scale_shift_act(const void* x, const void* residual, void* y, const float* scale, const float* shift,
                       int64_t npix, int C, int relu, int dtype, void* stream);
/* training backward.  ReLU mask: from y (the forward output) when given; with y == NULL and scale_shift (the forward's
 * save_scale_shift) given it is recomputed as fma(x, scale, shift) > 0 — the same decision bit for bit, one tensor less
 * to read in each of the two passes (only valid for layers without a residual input).  dres (nullable) receives the
 * gradient that flows to the residual input (= dy masked by ReLU).  accumulate != 0: dgamma / dbeta are ADDED to (the
 * caller passes the parameters' .grad buffers and skips a separate accumulation launch per parameter). */
int cn_bn_train_bwd(const void* dy, const void* x, const void* y, const float* gamma, const float* save_mean,
                    const float* save_invstd, const float* scale_shift, void* dx, void* dres, float* dgamma, float* dbeta,
                    int accumulate, int64_t npix, int C, int relu, int dtype, void* ws, size_t ws_bytes, void* stream);
/* the same with dres = masked dy + dres_acc (nullable, same layout as dres): the residual input is a tensor with several consumers
 * (pose_dla_dcn.py:60-68: a block's input feeds conv1 AND the skip path; :262 it is also a child of the Root) whose other
 * consumers' gradients are already summed in dres_acc — the sum happens in this store instead of in autograd's add pass */
int cn_bn_train_bwd_acc(const void* dy, const void* x, const void* y, const float* gamma, const float* save_mean,
                        const float* save_invstd, const float* scale_shift, void* dx, void* dres, const void* dres_acc,
                        float* dgamma, float* dbeta, int accumulate, int64_t npix, int C, int relu, int dtype, void* ws,
                        size_t ws_bytes, void* stream);
/* cn_bn_train_bwd_acc in TWO launches instead of three: the statistics pass adds its per-workgroup sums into `sink` (fp32
 * [slots][2][C], all-zero on entry; fp32 atomics) and the apply pass reduces the sink in its prologue (dgamma / dbeta written by its
 * first workgroup column).  `sink` / `clear` follow the protocol of cn_bn_train_fwd_sink. */
int cn_bn_train_bwd_sink(const void* dy, const void* x, const void* y, const float* gamma, const float* save_mean,
                         const float* save_invstd, const float* scale_shift, void* dx, void* dres, const void* dres_acc,
                         float* dgamma, float* dbeta, int accumulate, float* sink, int slots, float* clear, int64_t clear_n,
                         int64_t npix, int C, int relu, int dtype, void* stream);
/* dx = dy * (y > 0)  (ReLU backward for conv+bias+ReLU heads) */
int cn_relu_bwd(const void* dy, const void* y, void* dx, int64_t n, int dtype, void* stream);

/* ---- pooling / depthwise up-sampling -------------------------------------------------------- */
/* F.max_pool2d (msra_resnet.py:113 3x3/s2/p1, pose_dla_dcn.py:219 2x2/s2).  argmax (nullable) u8 [N,OH,OW,C]: window position
 * kh*k+kw of the FIRST maximum (ATen's tie rule); the backward routes dy through it instead of re-reading x. */
int cn_maxpool_fwd(const void* x, void* y, unsigned char* argmax, int N, int H, int W, int C, int k, int stride, int pad,
                   int OH, int OW, int dtype, void* stream);
int cn_maxpool_bwd(const unsigned char* argmax, const void* dy, void* dx, int N, int H, int W, int C, int k, int stride, int pad,
                   int OH, int OW, int dtype, void* stream);
/* dx = acc + routed dy (acc nullable, [N,H,W,C]): the pooled tensor also feeds other layers (pose_dla_dcn.py:245-262: a Tree's
 * input goes through `downsample` AND `tree1`), whose summed gradients arrive in acc */
int cn_maxpool_bwd_acc(const unsigned char* argmax, const void* dy, const void* acc, void* dx, int N, int H, int W, int C, int k,
                       int stride, int pad, int OH, int OW, int dtype, void* stream);
/* Hourglass merge (large_hourglass.py:108-125 MergeUp/make_unpool_layer, :196-204 kp_module.forward):
 * y[N,2H,2W,C] = a + nearest_up2x(low[N,H,W,C]); a == NULL gives plain nn.Upsample(scale_factor=2).  Backward of the
 * up-sampled operand: dlow[N,H,W,C] = 2x2 block sums of dy[N,2H,2W,C] (the `a` operand's gradient is dy itself). */
int cn_upsample2x_add(const void* a, const void* low, void* y, int N, int H, int W, int C, int dtype, void* stream);
int cn_sumpool2x2(const void* dy, void* dlow, int N, int H, int W, int C, int dtype, void* stream);
/* depthwise ConvTranspose2d(o,o,2f,stride=f,padding=f/2,groups=o) — pose_dla_dcn.py:466-475; w fp32 [C,1,k,k].
 * residual (nullable, [N,OH,OW,C]) is added in the store: IDAUp's `node(up(proj(x)) + layers[i-1])`, pose_dla_dcn.py:483-488. */
int cn_dwdeconv_fwd(const void* x, const float* w, const void* residual, void* y, int N, int H, int W, int C, int k, int stride,
                    int pad, int OH, int OW, int dtype, void* stream);
int cn_dwdeconv_bwd_input(const void* dy, const float* w, void* dx, int N, int H, int W, int C, int k, int stride,
                          int pad, int OH, int OW, int dtype, void* stream);
int cn_dwdeconv_bwd_weight(const void* x, const void* dy, float* dw /* zeroed fp32 [C,k,k] */, int N, int H, int W,
                           int C, int k, int stride, int pad, int OH, int OW, int dtype, void* stream);
/* the same for the bilinear x2 layers of IDAUp (pose_dla_dcn.py:424-432: k = 4, stride 2, pad 1, OH = 2 H, OW = 2 W; bf16; C in
 * {64, 128, 256} or a multiple of 512): every dy pixel is read once (a lane walks a dy row with a sliding four-pixel window) and the
 * workgroups' partial sums meet in slabs instead of same-address atomics; dw += the sum.  ws: cn_dwdeconv_wgrad_ws_bytes() of
 * scratch.  Any other shape -> CN_EUNSUPPORTED (call cn_dwdeconv_bwd_weight). */
size_t cn_dwdeconv_wgrad_ws_bytes(int N, int OH, int C);
int cn_dwdeconv_bwd_weight_rows(const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, int N, int H, int W, int C, int k,
                                int stride, int pad, int OH, int OW, int dtype, void* stream);
int cn_dwdeconv_bwd_weight_rows_h(const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, int N, int H, int W, int C, int k,
                                  int stride, int pad, int OH, int OW, int dtype, cn_hooks* hooks, void* stream);

/* A 2-channel task head (heads.py:9-15: conv3x3 + bias -> ReLU -> conv1x1 + bias; width_height / regression) in ONE launch, no-grad
 * path: out fp32 NCHW [N, 2, H, W] — ALL-ZERO at launch, the kernel adds — = conv1x1(relu(conv3x3(x) + b1)) + b2; x NHWC bf16, Ci = 64,
 * wp1 = cn_pack_weight(hidden weight, mode 1), w2 fp32 [2][Ch] (the 1x1 weight as stored), Ch a multiple of 64.  The hidden activation
 * (537 MB at bs 64, 128x128) is never written or re-read.  CN_EUNSUPPORTED when the weight-stationary kernel declines the shape.
 * NOT bit-reproducible run to run: every pixel is the fp32-atomic sum of Ch/32 per-wave partials in arrival order (differences in the
 * last bits of the map; the two-launch pair cn_conv2d_fwd + cn_conv1x1_nchw_fwd is the deterministic form — the host mirror takes it
 * under torch.use_deterministic_algorithms(True) or CN_DISABLE_HEAD2). */
int cn_head2_fwd(const void* x, const void* wp1, const float* b1, const float* w2, const float* b2, float* out, int N, int H, int W,
                 int Ci, int x_ld, int Ch, int dtype, void* stream);
/* A head's last layer (heads.py:15-17: nn.Conv2d(head_conv, out_channels, 1) on the hidden activation) straight into the public
 * layout: y fp32 NCHW [N,Co,H,W] = conv1x1(x) + bias, x [N,H,W,x_ld] in `dtype`, wp = cn_pack_weight(mode 1).  Replaces
 * cn_conv2d_fwd (NHWC bf16 out) + cn_nhwc_to_nchw.  bf16, Ci == 256, Co <= 128, H*W % 32 == 0, N*H*W >= 65536; anything else
 * -> CN_EUNSUPPORTED (run the pair). */
int cn_conv1x1_nchw_fwd(const void* x, const void* wp, const float* bias, float* y, int N, int H, int W, int Ci, int x_ld, int Co,
                        int dtype, void* stream);
/* 1x1 / stride-1 convolution over the channel CONCATENATION of nsrc (<= 6) NHWC tensors of the same N,H,W — DLA's Root:
 * `conv(torch.cat(x, 1))`, pose_dla_dcn.py:180-188 — without materialising the concatenation: the K loop of the implicit GEMM
 * walks the sources.  x_i: contiguous [N,H,W,c_i] (c_i a multiple of 16; unused slots NULL / 0); wp = cn_pack_weight mode 1 of
 * the [Co, sum c_i, 1, 1] weight; epilogue (bias, residual, relu) as cn_conv2d_fwd. */
int cn_conv1x1_cat_fwd(const void* x0, const void* x1, const void* x2, const void* x3, const void* x4, const void* x5,
                       int c0, int c1, int c2, int c3, int c4, int c5, int nsrc, const void* wp, const float* bias,
                       const void* residual, void* y, int N, int H, int W, int Co, int y_ld, int res_ld, int relu,
                       int dtype, void* stream);
int cn_conv1x1_cat_fwd_h(const void* x0, const void* x1, const void* x2, const void* x3, const void* x4, const void* x5,
                         int c0, int c1, int c2, int c3, int c4, int c5, int nsrc, const void* wp, const float* bias,
                         const void* residual, void* y, int N, int H, int W, int Co, int y_ld, int res_ld, int relu,
                         int dtype, cn_hooks* hooks, void* stream);

/* ---- DCNv2 (DCN.dcn_v2.DCN, pose_dla_dcn.py:441-449) --------------------- */
/* om = conv_offset_mask(x) as NHWC FP32 [P][om_ld] in both compute modes — sampling coordinates stay exact —
 * (channels 0..17 interleaved dy,dx per tap; 18..26 mask logits).
 * col[p][k*Ci + c] = sigmoid(om[p][18+k]) * bilinear(x[n,:,:,c], h-1+i+dy, w-1+j+dx), k = 3i+j. */
int cn_dcn_im2col(const void* x, const float* om, void* col, int N, int H, int W, int Ci, int x_ld, int om_ld,
                  int dtype, void* stream);
/* Given dcol (gradient of col): dx = dx_tile + dx_far, both fp32 [P][Ci].  dx_tile is fully overwritten by an atomic-free
 * gather (every destination pixel sums the samples within 3 pixels whose bilinear weight reaches it); dx_far (zeroed by
 * the caller) only receives samples displaced by more than 3 pixels, through global atomics.  dom fp32 [P][om_ld]:
 * channels 0..26 are overwritten. */
int cn_dcn_col2im(const void* dcol, const void* x, const float* om, float* dx_tile, float* dx_far, float* dom,
                  int N, int H, int W, int Ci, int x_ld, int om_ld, int dtype, void* stream);
/* Which kernel template a DCNv2 entry point dispatches to (bf16, standard pitches): entry 0 = cn_dcn_fwd, 1 = cn_dcn_wgrad,
 * 2 = cn_dcn_bwd_dom, 3 = cn_dcn_bwd_dx; codes are listed at the definition (csrc/conv_igemm.hip).  Measurement aid: bench.py
 * names its per-kernel roofline rows with it so that they match rocprofv3's kernel names.  No reference counterpart. */
int cn_dcn_variant(int entry, int Ci, int Co);
/* The same with the map size (H, W): the 16x16-tile forward kernel of csrc/dcn_b2.hip is chosen by size as well (code 5000000). */
int cn_dcn_variant_hw(int entry, int Ci, int Co, int H, int W);
/* Fused DCNv2 forward: bilinear sampling straight into the MFMA operand tile in LDS — no column tensor in HBM.
 * y = act(bias + sum_k W_k * sigmoid(om[18+k]) * bilinear_k(x)); wp = cn_pack_weight mode 1 ([Co_pad32][tap*Ci + ci]). */
int cn_dcn_fwd(const void* x, const float* om, const void* wp, const float* bias, void* y,
               int N, int H, int W, int Ci, int x_ld, int Co, int y_ld, int om_ld, int relu, int dtype, void* stream);
int cn_dcn_fwd_h(const void* x, const float* om, const void* wp, const float* bias, void* y,
                 int N, int H, int W, int Ci, int x_ld, int Co, int y_ld, int om_ld, int relu, int dtype, cn_hooks* hooks, void* stream);
/* Fused DCNv2 weight gradient (bf16): dwp[co][tap*Ci+ci] += sum_p dy[p][co] * sampled_x[p,tap][ci], the sampled operand
 * rebuilt per tap in LDS (no column tensor).  dwp fp32 [Co_pad32][9*Ci], zeroed by the caller.  fp32 -> CN_EUNSUPPORTED. */
int cn_dcn_wgrad(const void* x, const float* om, const void* dy, float* dwp, int N, int H, int W, int Ci, int x_ld,
                 int Co, int dy_ld, int om_ld, int dtype, void* stream);
int cn_dcn_wgrad_h(const void* x, const float* om, const void* dy, float* dwp, int N, int H, int W, int Ci, int x_ld,
                   int Co, int dy_ld, int om_ld, int dtype, cn_hooks* hooks, void* stream);
/* Fused DCNv2 backward (no column gradient in HBM), used instead of cn_dcn_col2im:
 *   cn_dcn_bwd_dom: GEMM dcol = dY x W^T (wpd2 = cn_pack_weight mode 2) whose epilogue reduces dcol against the bilinear
 *     corner differences of x -> dom fp32 [P][om_ld] (channels 0..26; zeroed by the caller when Ci > 128), and scatters
 *     samples displaced by more than 3 px into dx_far (fp32 [P][Ci], zeroed by the caller).
 *   cn_dcn_bwd_dx: dx[q] = sum_k W_k^T G_k[q], G_k = adjoint bilinear sampling of dY (hit lists in LDS, atomic-free),
 *     + dx_far in the epilogue.  wpd0 = cn_pack_weight mode 0.  dy_ld must equal the weights' inner_pad = rup16(Co).
 * Lazy dx_far protocol (far_flag != NULL, one int32 on the device, zeroed by the caller): dom sets the flag when it
 * scattered at least one far sample; dx adds dx_far only if the flag is set and then writes zeros back, so the caller
 * keeps ONE persistent zero-initialised dx_far per shape instead of clearing (and reading) 4*P*Ci bytes per layer per
 * step.  far_flag == NULL: dx_far is zeroed by the caller and always added.  When Ci == 64 (one channel block) dom also
 * writes the padding channels 27..om_ld-1 of `dom`, so the caller need not clear it.
 * Slabs: with Ci > 64 the offset gradient is summed over 64-channel blocks of x.  Global fp32 atomics for that cost 2.3x
 * the kernel, so the caller may pass dom as S = cn_dcn_bwd_dom_slabs() consecutive copies [S][P][om_ld] (dom_slabs = S):
 * every channel block then writes its own copy with plain stores (all om_ld channels, nothing to clear) and the caller
 * folds them with cn_sum_slabs.  dom_slabs = 1 keeps the single-copy / atomic behaviour.  dom_slabs = 0 (bf16, Ci == 64,
 * cn_dcn_bwd_dom_slabs() == 1): `dom` is the FINAL bf16 tensor [P][om_ld], written directly by the tile kernel — no fp32
 * copy, no cn_sum_slabs cast pass. */
int cn_dcn_bwd_dom_slabs(int Ci, int dy_ld, int dtype);
int cn_dcn_bwd_dom(const void* dy, const void* wpd2, const void* x, const float* om, float* dom, int dom_slabs, float* dx_far,
                   int* far_flag, int N, int H, int W, int Ci, int Co, int dy_ld, int x_ld, int om_ld, int dtype, void* stream);
/* dst[i] = sum_s src[s*n + i] for S fp32 slabs of n elements -> dst in `dtype` */
int cn_sum_slabs(const float* src, void* dst, int S, int64_t n, int dtype, void* stream);
int cn_dcn_bwd_dx(const void* dy, const void* wpd0, const float* om, float* dx_far, int* far_flag, void* dx,
                  int N, int H, int W, int Ci, int dy_ld, int om_ld, int dtype, void* stream);
/* out[i] = a[i] + b[i] for fp32 a, b -> out in `dtype` (combines dx_tile + dx_far into the activation dtype) */
int cn_add_f32_to(const float* a, const float* b, void* out, int64_t n, int dtype, void* stream);

/* ---- losses (utils/losses.py) ---------------------------------------------------------------- */
/* in-place sigmoid on x, y = clamp(x, lo, 1-lo)  (utils/decode.py:43-45) */
int cn_sigmoid_clamp_fwd(float* x, float* y, int64_t n, float lo, void* stream);
/* dz = dy * p*(1-p) * [lo <= p <= 1-lo] where p = x (the in-place sigmoid result) */
int cn_sigmoid_clamp_bwd(const float* dy, const float* x_sig, float* dz, int64_t n, float lo, void* stream);
size_t cn_focal_workspace_bytes(int64_t n);
/* penalty-reduced focal loss (utils/losses.py:14-39).  pred [B,C,HW], gt [Bg,Cg,HW] with broadcasting when
 * Bg/Cg == 1.  out[0] = loss, out[1] = pos_sum, out[2] = neg_sum, out[3] = num_pos (device fp32[4]). */
int cn_focal_fwd(const float* pred, const float* gt, float* out4, int B, int C, int64_t HW, int gtB, int gtC,
                 void* ws, size_t ws_bytes, void* stream);
/* dpred = gout[0] * dloss/dpred, using out4 from the forward */
int cn_focal_bwd(const float* pred, const float* gt, const float* out4, const float* gout, float* dpred,
                 int B, int C, int64_t HW, int gtB, int gtC, void* stream);
/* cn_focal_bwd followed by cn_sigmoid_clamp_bwd in one pass (the training path always chains them: centernet_detection.py:103-106):
 * x_sig = sigmoid(logits) as left in place by cn_sigmoid_clamp_fwd, out4 / gout as for cn_focal_bwd, dz = d loss / d logits. */
int cn_sigmoid_focal_bwd(const float* x_sig, const float* gt, const float* out4, const float* gout, float* dz, int B, int C,
                         int64_t HW, int gtB, int gtC, float lo, void* stream);
/* cn_sigmoid_focal_bwd that also writes dz as the head's backward wants it: dz_nhwc bf16 [B,HW,ld] (channels C..ld-1 zero), next to
 * the fp32 NCHW dz — the consumer (heads.py:15-17 backwards) then skips its cn_nchw_to_nhwc pass.  gt has the maps' shape; HW % 64
 * == 0, C <= ld <= 256, ld % 8 == 0, else CN_EUNSUPPORTED (run cn_sigmoid_focal_bwd).  dz is bit-identical to cn_sigmoid_focal_bwd's. */
int cn_sigmoid_focal_bwd_dual(const float* x_sig, const float* gt, const float* out4, const float* gout, float* dz, void* dz_nhwc,
                              int B, int C, int64_t HW, int ld, float lo, void* stream);
/* cn_sigmoid_clamp_fwd + cn_focal_fwd in one pass (centernet_detection.py:103-106: `sigmoid_clamped` then FocalLoss): x fp32 [n]
 * becomes sigmoid(x) in place, y its clamped copy, out4 as for cn_focal_fwd on (y, gt); gt has x's shape (no broadcast), n % 4
 * == 0, 16-byte aligned pointers (CN_EUNSUPPORTED otherwise: run the two entry points).  Results are bit-identical to the pair. */
int cn_sigmoid_clamp_focal_fwd(float* x, float* y, const float* gt, float* out4, int64_t n, float lo, void* ws, size_t ws_bytes,
                               void* stream);
/* masked gather-L1 (utils/losses.py:53-63, 81-91): feat NCHW fp32 [B,C,HW]; ind int64 [B,N]; mask uint8 [B,N]
 * (mask_has_c == 0) or [B,N,C]; target fp32 [B,N,C].  out[0] = loss, out[1] = sum|.|, out[2] = sum(mask). */
int cn_gather_l1_fwd(const float* feat, const int64_t* ind, const uint8_t* mask, const float* target, float* out3,
                     int B, int C, int64_t HW, int N, int mask_has_c, void* stream);
/* dfeat must be zeroed by the caller; scatter-adds gout[0] * dloss/dfeat */
int cn_gather_l1_bwd(const float* feat, const int64_t* ind, const uint8_t* mask, const float* target,
                     const float* out3, const float* gout, float* dfeat,
                     int B, int C, int64_t HW, int N, int mask_has_c, void* stream);
/* loss assembly (centernet_detection.py:108-116 `hm_weight * hm_loss + wh_weight * wh_loss + off_weight * off_loss`;
 * centernet_multi_pose.py:126-140): out[0] = sum_{i<n} w_i * t_i[0] over n <= 8 scalar terms; backward out[i] = w_i * g[0]. */
int cn_weighted_sum(const float* t0, const float* t1, const float* t2, const float* t3, const float* t4, const float* t5,
                    const float* t6, const float* t7, float w0, float w1, float w2, float w3, float w4, float w5, float w6,
                    float w7, int n, float* out, void* stream);
int cn_weighted_sum_bwd(const float* g, float w0, float w1, float w2, float w3, float w4, float w5, float w6, float w7,
                        int n, float* out, void* stream);

/* ---- backward of a head whose loss gathers its output at `ind` (heads.py:4-25 under RegL1Loss / RegWeightedL1Loss,
 * utils/losses.py:53-63, 81-91): the output gradient is zero except at ind[b, :], so autograd's dense conv1x1 <- ReLU <- conv3x3
 * backward reduces to sums over the R = B*M rows.  cn_head_sparse_gather builds the compact operands (dtype = the head's
 * activation dtype): h [B,H,W,h_ld] hidden activation (post-ReLU), x [B,H,W,x_ld] head input, ind int64 [B,M] (pixel index
 * y*W + x), dout fp32 NCHW [B,C,H,W] (the dense output gradient; read at ind only, repeated indices counted once), w2 fp32
 * [C,Ch] (the 1x1 conv's parameter).  Outputs: hg [R,Ch] = h rows, dhc [R,Ch] = (h > 0) * (w2^T g), xg [R,9*Ci] = the 3x3 input
 * patch of every row in PARAMETER order (column ci*9 + kh*3 + kw, zeros outside the image), gq [R,Cq] = g rows zero-padded to
 * Cq channels.  C <= 64. */
int cn_head_sparse_gather(const void* h, const void* x, const int64_t* ind, const float* dout, const float* w2, void* hg, void* dhc,
                          void* xg, void* gq, int B, int M, int C, int H, int W, int Ch, int h_ld, int Ci, int x_ld, int Cq, int dtype,
                          void* stream);
/* the same gather in two halves for a head whose forward was the one-launch cn_head2_fwd (no hidden activation stored): mode 1 writes
 * the 3x3 input patches xg and the output-gradient rows gq (h and dhc unused, may be NULL); the caller recomputes the hidden rows
 * relu(xg W1^T + b1) with the 1x1 entry points; mode 2 takes those rows as h [B*M, h_ld] and writes the masked hidden gradient dhc. */
int cn_head_sparse_gather_rows(const void* h, const void* x, const int64_t* ind, const float* dout, const float* w2, void* dhc, void* xg,
                               void* gq, int B, int M, int C, int H, int W, int Ch, int h_ld, int Ci, int x_ld, int Cq, int mode,
                               int dtype, void* stream);
/* dx[b, p + (kh-1, kw-1), ci] += dxc[r, ci*9 + kh*3 + kw] for p = ind[r] (r = b*M + m), taps outside the image dropped; dx
 * [B,H,W,dx_ld] in `dtype` (fp32: atomic add; bf16: compare-and-swap on the containing word), dxc fp32 [R, 9*Ci]. */
int cn_scatter3x3_add(const float* dxc, const int64_t* ind, void* dx, int B, int M, int H, int W, int Ci, int dx_ld, int dtype,
                      void* stream);

/* ---- ground-truth encoding (replaces the per-sample host loop of sample/ctdet.py:39-90) -------------- */
/* boxes fp32 [B][M][4] = COCO (x, y, w, h) in input pixels, cls int32 [B][M], nobj int32 [B] (objects beyond nobj[b] are
 * ignored).  heatmap fp32 [B][C][OH][OW] must be ZEROED by the caller (gaussians are max-splatted into it);
 * mask uint8 [B][M], indices int64 [B][M], wh / reg fp32 [B][M][2] are fully written.  gaussian_type 0: umich gaussians
 * (the reference's default; utils/gaussian.py:28-58), 1: msra (utils/gaussian.py:61-83 with the integer radius as sigma,
 * sample/ctdet.py:54); radius from gaussian_radius with min_overlap 0.7 (utils/gaussian.py:6-26) in both. */
int cn_encode_ctdet(const float* boxes, const int* cls, const int* nobj, float* heatmap, unsigned char* mask,
                    int64_t* indices, float* wh, float* reg, int B, int M, int C, int OH, int OW, int down_ratio,
                    int gaussian_type, void* stream);
/* Replaces the host loop of MultiPoseSample.__call__ (sample/multi_pose.py:35-112; draw_msra_gaussian utils/gaussian.py:61-83)
 * for a whole batch.  boxes fp32 [B,M,4] (x,y,w,h, input pixels), keypoints fp32 [B,M,J,3] (x,y,visibility), nobj int32 [B].
 * Outputs (collated layouts of :103-110): heatmap_keypoints fp32 [B,J,OH,OW] (ZEROED by the caller), keypoints fp32 [B,M,2J],
 * keypoints_mask u8 [B,M,2J], heatmap_keypoints_offset fp32 [B,M*J,2], heatmap_keypoints_indices int64 [B,M*J],
 * heatmap_keypoints_mask u8 [B,M*J].  The ctdet part of a multi_pose sample is cn_encode_ctdet with C = 1. */
int cn_encode_multi_pose(const float* boxes, const float* keypoints, const int* nobj, float* heatmap_keypoints,
                         float* kp_out, unsigned char* kp_mask, float* hp_offset, int64_t* hp_indices,
                         unsigned char* hp_mask, int B, int M, int J, int OH, int OW, int down_ratio, void* stream);

/* ---- test-time augmentation + detection post-processing (centernet_detection.py:132-225, utils/nms.py:5-107) ---------- */
/* test_step :139-158 for a batch: out[b] = pad(normalize(img[b])) with zero padding BEFORE the normalisation, and, when
 * flip, out[B+b] = hflip(out[b]).  img fp32 [B,3,H,W] (already resized for scales != 1), out fp32 [B*(1+flip),3,H+2pad_y,W+2pad_x]. */
int cn_tta_prepare(const float* img, float* out, int B, int H, int W, int pad_x, int pad_y, float mean0, float mean1,
                   float mean2, float std0, float std1, float std2, int flip, void* stream);
/* test_step :139-158 with the multi-scale resize of :139-141 (`VF.resize(img, (new_h, new_w))` on a tensor: bilinear, half-pixel
 * centres, no antialias, edge clamp = ATen upsample_bilinear2d, align_corners=False) folded into the same launch.  img fp32
 * [B,3,H,W] at the ORIGINAL size, out fp32 [B*(1+flip),3,new_h+2pad_y,new_w+2pad_x].  new_h == H && new_w == W: cn_tta_prepare. */
int cn_tta_prepare_scaled(const float* img, float* out, int B, int H, int W, int new_h, int new_w, int pad_x, int pad_y,
                          float mean0, float mean1, float mean2, float std0, float std1, float std2, int flip, void* stream);
/* test_step :167-171: out[B,C,H,W] = (x[0:B] + hflip(x[B:2B])) / 2 on NCHW fp32 head maps. */
int cn_flip_merge(const float* x, float* out, int B, int C, int H, int W, void* stream);
/* test_step_end :173-225 for a batch: dets fp32 [S,B,K,6] (ctdet_decode output per test scale), meta fp32 [S,4] =
 * (pad_x, pad_y, scale_x, scale_y).  Boxes * down_ratio - pad, / scale; grouped by class; soft_nms(Nt, method) per class
 * when S > 1 (utils/nms.py, arithmetic in double as under numba); then only scores >= the max_per_image-th largest stay.
 * rows fp32 [B,S*K,6] class-ascending (x1,y1,x2,y2,score,class), zero padded; counts int32 [B].  S*K <= 1024, C <= 256. */
int cn_ctdet_merge(const float* dets, const float* meta, float* rows, int* counts, int S, int B, int K, int C,
                   int down_ratio, int max_per_image, int nms_method, float nms_nt, float nms_sigma, float nms_threshold,
                   void* stream);
/* Pose-aware mirror merge (centernet_multi_pose.py:200-211): out[b,c] = (x[b,c] + sign[c] * hflip(x[B+b, perm[c]])) / 2;
 * perm int32 [C] and sign fp32 [C] on the device (keypoints: flip_idx per joint, -1 on x components; heat maps: flip_idx, +1). */
int cn_flip_merge_perm(const float* x, float* out, const int* perm, const float* sign, int B, int C, int H, int W,
                       void* stream);
/* CenterNetMultiPose.test_step_end (centernet_multi_pose.py:213-264) for a batch: dets fp32 [S,B,K,D] rows of
 * multi_pose_decode (D = 57), meta as cn_ctdet_merge.  Boxes (cols 0-3) and keypoints (cols 5-38) to image coordinates,
 * scales concatenated, soft_nms_39 (utils/nms.py:109-206; it moves columns 0-38 only) when S > 1, scores >= the
 * max_per_image-th largest kept.  rows fp32 [B,S*K,D] zero padded, counts int32 [B].  S*K <= 1024. */
int cn_pose_merge(const float* dets, const float* meta, float* rows, int* counts, int S, int B, int K, int D,
                  int down_ratio, int max_per_image, int nms_method, float nms_nt, float nms_sigma, float nms_threshold,
                  void* stream);

/* ---- decode (utils/decode.py, decode/ctdet.py, decode/multi_pose.py) -------------------------- */
/* keep[b,c,h,w] = heat * (maxpool3x3(heat) == heat)   (utils/decode.py:5-10) */
int cn_nms3x3(const float* heat, float* out, int B, int C, int H, int W, void* stream);
/* the same with any odd window (utils/decode.py:5 `kernel`; the reference only ever passes the default 3) */
int cn_nms(const float* heat, float* out, int B, int C, int H, int W, int kernel, void* stream);
/* per (b,c): top-K of (apply_nms ? nms3x3(heat) : heat) over H*W, descending, ties -> lower index.
 * scores fp32 [B,C,K], inds int32 [B,C,K].  K <= 256.  (utils/decode.py:16, :34) */
int cn_topk_channel(const float* heat, float* scores, int32_t* inds, int B, int C, int H, int W, int K,
                    int apply_nms, void* stream);
/* generic row top-K: x fp32 [R][L] -> vals [R][K], idx int32 [R][K]; L <= 65536 (utils/decode.py:22) */
int cn_topk_rows(const float* x, float* vals, int32_t* idx, int R, int L, int K, void* stream);
/* out[b,n,c] = feat[b,c,ind[b,n]]  (utils/decode.py:59-63); ind int64 */
int cn_gather_rows(const float* feat, const int64_t* ind, float* out, int B, int C, int64_t HW, int N, void* stream);
/* fused ctdet_decode (decode/ctdet.py:6-38): heat is post-sigmoid.  det fp32 [B,K,6]; inds int64 [B,K] and
 * clses int32 [B,K] are optional outputs (nullable).  ws >= cn_ctdet_decode_workspace_bytes. */
size_t cn_ctdet_decode_workspace_bytes(int B, int C, int K);
int cn_ctdet_decode(const float* heat, const float* wh, const float* reg /* nullable */, float* det,
                    int64_t* inds, int32_t* clses, int B, int C, int H, int W, int K,
                    void* ws, size_t ws_bytes, void* stream);
/* cn_ctdet_decode on the LOGITS of the class heat map: the top-K kernel applies clamp(sigmoid(x), lo, 1 - lo) on load (the arithmetic of
 * cn_sigmoid_clamp_fwd, bit-identical detections), the map is left untouched — no sigmoid pass over the map in inference
 * (centernet_detection.py:183-187).  128x128 maps only: CN_EUNSUPPORTED otherwise (run cn_sigmoid_clamp_fwd + cn_ctdet_decode). */
int cn_ctdet_decode_logits(const float* heat_logits, const float* wh, const float* reg, float* det, int64_t* inds, int32_t* clses,
                           int B, int C, int H, int W, int K, float lo, void* ws, size_t ws_bytes, void* stream);
/* fused multi_pose_decode (decode/multi_pose.py:7-96): det fp32 [B,K,5+2J+1+J] */
size_t cn_multi_pose_decode_workspace_bytes(int B, int J, int K);
int cn_multi_pose_decode(const float* heat, const float* wh, const float* kps, const float* reg /* nullable */,
                         const float* hm_hp, const float* hp_offset /* nullable */, float* det,
                         int B, int J, int H, int W, int K, void* ws, size_t ws_bytes, void* stream);

/* ---- optimiser (torch.optim.Adam defaults, centernet.py:94-95) -------------------------------- */
/* flat fp32 buffers; bias corrections bc1 = 1-b1^t, bc2 = 1-b2^t are computed by the host.  hyper (nullable, device
 * fp32[3] = {lr, bc1, bc2}) overrides the scalar arguments so that a captured hipGraph sees fresh values per replay. */
int cn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                 float eps, float bc1, float bc2, float grad_scale, const float* hyper, void* stream);
/* t += 1 and the bias corrections of cn_adam_step's `hyper` buffer, on the device: hyper fp32[4] = {lr, 1-b1^t, 1-b2^t,
 * t as int32 bits}.  Launched right before cn_adam_step (inside the same captured graph), so replays never depend on a
 * per-step host upload; the host only writes hyper[0] when the learning rate changes and hyper[3] when it restores t. */
int cn_adam_advance(float* hyper, float b1, float b2, void* stream);

#ifdef __cplusplus
}
#endif
#endif
