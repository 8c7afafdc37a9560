// Stem convolution (7x7, 3 input channels: msra_resnet.py:110 stride 2, pose_dla_dcn.py:282 stride 1).
// Ci = 3 is useless for MFMA K-slices, and the layer is < 2 % of the network's FLOPs, so this is a
// direct VALU kernel: one output pixel per lane, 16 output channels per workgroup pass, the NCHW fp32
// image tile and the weight slab staged in LDS (weights are read as wave-wide broadcasts).
#include "common.h"

#define ST_TH 8
#define ST_TW 32
#define ST_COB 16
#define ST_MAXK 7
#define ST_MAXCI 4

template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       T* __restrict__ y, int Ci, int H, int W, int Co, int KH, int KW,
                                                       int stride, int pad, int OH, int OW, int tiles_w,
                                                       const float* __restrict__ scale, const float* __restrict__ bias, int relu) {
    extern __shared__ float smem[];
    const int IH = (ST_TH - 1) * stride + KH, IW = (ST_TW - 1) * stride + KW;
    float* xt = smem;                       // [Ci][IH][IW]
    float* wt = smem + Ci * IH * IW;        // [Ci*KH*KW][ST_COB]
    const int tid = threadIdx.x;
    const int n = blockIdx.z, co0 = blockIdx.y * ST_COB;
    const int th0 = (blockIdx.x / tiles_w) * ST_TH, tw0 = (blockIdx.x % tiles_w) * ST_TW;
    const int ih0 = th0 * stride - pad, iw0 = tw0 * stride - pad;
    for (int i = tid; i < Ci * IH * IW; i += 256) {
        int c = i / (IH * IW), r = i - c * IH * IW;
        int hh = r / IW, ww = r - hh * IW;
        int ih = ih0 + hh, iw = iw0 + ww;
        float v = 0.f;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = x[(((int64_t)n * Ci + c) * H + ih) * W + iw];
        xt[i] = v;
    }
    const int ntap = Ci * KH * KW;
    for (int i = tid; i < ntap * ST_COB; i += 256) {
        int t = i / ST_COB, c = i - t * ST_COB;
        wt[i] = (co0 + c < Co) ? w[(int64_t)(co0 + c) * ntap + t] : 0.f;   // w[co][ci][kh][kw], t = (ci*KH+kh)*KW+kw
    }
    __syncthreads();
    const int ty = tid / ST_TW, tx = tid % ST_TW;
    float acc[ST_COB];
#pragma unroll
    for (int c = 0; c < ST_COB; ++c) acc[c] = 0.f;
    for (int ci = 0; ci < Ci; ++ci)
        for (int kh = 0; kh < KH; ++kh) {
            const float* xr = xt + (ci * IH + ty * stride + kh) * IW + tx * stride;
            const float* wr = wt + ((ci * KH + kh) * KW) * ST_COB;
            for (int kw = 0; kw < KW; ++kw) {
                const float xv = xr[kw];
                const float4* wv = reinterpret_cast<const float4*>(wr + kw * ST_COB);
#pragma unroll
                for (int q = 0; q < ST_COB / 4; ++q) {
                    float4 ww = wv[q];
                    acc[4 * q + 0] = fmaf(xv, ww.x, acc[4 * q + 0]);
                    acc[4 * q + 1] = fmaf(xv, ww.y, acc[4 * q + 1]);
                    acc[4 * q + 2] = fmaf(xv, ww.z, acc[4 * q + 2]);
                    acc[4 * q + 3] = fmaf(xv, ww.w, acc[4 * q + 3]);
                }
            }
        }
    if (scale || bias || relu) {                 // folded eval-mode BN (+ReLU): y = act(fma(conv, scale, shift))
#pragma unroll
        for (int c = 0; c < ST_COB; ++c) {
            const bool in = co0 + c < Co;
            float v = acc[c];
            v = fmaf(v, (scale && in) ? scale[co0 + c] : 1.f, (bias && in) ? bias[co0 + c] : 0.f);
            acc[c] = relu ? fmaxf(v, 0.f) : v;
        }
    }
    const int oh = th0 + ty, ow = tw0 + tx;
    if (oh < OH && ow < OW) {
        T* dst = y + (((int64_t)n * OH + oh) * OW + ow) * Co + co0;
        if (co0 + ST_COB <= Co) {
#pragma unroll
            for (int q = 0; q < ST_COB / Vec16<T>::N; ++q) Vec16<T>::store(dst + q * Vec16<T>::N, acc + q * Vec16<T>::N);
        } else {
            for (int c = 0; c < ST_COB && co0 + c < Co; ++c) Elem<T>::st(dst + c, acc[c]);
        }
    }
}

// Weight gradient for layers with <= 16 input channels (7x7 stem on the NCHW fp32 image; DLA level0/level1 3x3 convs on
// 16-channel NHWC maps).  K = pixels is huge (16.7 M at 512x512, batch 64) while M x N = Co x (Ci*taps) is tiny, and a
// 16-channel operand wastes 3/4 of a 64x64 MFMA tile plus its LDS transposes — so this is a VALU kernel that is
// FMA-issue bound: thread = one (ci, kh, kw) position x 16 output channels held in registers; per pixel it does ONE
// LDS read of the image tile and four wave-broadcast ds_read_b128 of the dy row for 16 FMAs.  Workgroups walk many
// pixel tiles and flush with one round of atomics.
#define SW_THREADS 320
template <typename T, bool X_NCHW>
__global__ __launch_bounds__(SW_THREADS) void small_wgrad_kernel(const void* __restrict__ xv, const T* __restrict__ dy,
                                                                 float* __restrict__ dw, int N, int Ci, int x_ld, int H,
                                                                 int W, int Co, int dy_ld, int KH, int KW, int stride,
                                                                 int pad, int OH, int OW, int tiles_h, int tiles_w,
                                                                 int os_co, int os_ci, int os_tap) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int IH = (ST_TH - 1) * stride + KH, IW = (ST_TW - 1) * stride + KW;
    float* dt = smem;                                   // [256 pixels][16 co]   (16-byte aligned rows)
    float* xt = smem + ST_TH * ST_TW * ST_COB;          // [Ci][IH][IW]
    const int tid = threadIdx.x;
    const int co0 = blockIdx.y * ST_COB;
    const int NT = Ci * KH * KW;
    const int u = tid % NT, half = tid / NT;            // half 0: tile rows 0..3, half 1: rows 4..7
    const bool active = tid < 2 * NT;
    const int ci = u / (KH * KW), r0 = u - ci * KH * KW;
    const int kh = r0 / KW, kw = r0 - kh * KW;
    const int xoff = (ci * IH + kh + half * (ST_TH / 2) * stride) * IW + kw;
    float acc[ST_COB];
#pragma unroll
    for (int c = 0; c < ST_COB; ++c) acc[c] = 0.f;
    const int64_t ntiles = (int64_t)N * tiles_h * tiles_w;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = (int)(tile / (tiles_h * tiles_w));
        const int r = (int)(tile - (int64_t)n * tiles_h * tiles_w);
        const int th0 = (r / tiles_w) * ST_TH, tw0 = (r % tiles_w) * ST_TW;
        const int ih0 = th0 * stride - pad, iw0 = tw0 * stride - pad;
        __syncthreads();
        for (int i = tid; i < Ci * IH * IW; i += SW_THREADS) {
            float v = 0.f;
            if (X_NCHW) {
                const int c = i / (IH * IW), rr = i - c * IH * IW;
                const int ih = ih0 + rr / IW, iw = iw0 + rr % IW;
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                    v = reinterpret_cast<const float*>(xv)[(((int64_t)n * Ci + c) * H + ih) * W + iw];
                xt[i] = v;
            } else {                                     // NHWC: consecutive threads -> consecutive channels of one pixel
                const int c = i % Ci, pp = i / Ci;
                const int ih = ih0 + pp / IW, iw = iw0 + pp % IW;
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                    v = Elem<T>::ld(reinterpret_cast<const T*>(xv) + (((int64_t)n * H + ih) * W + iw) * x_ld + c);
                xt[(c * IH + pp / IW) * IW + pp % IW] = v;
            }
        }
        for (int i = tid; i < ST_TH * ST_TW * ST_COB; i += SW_THREADS) {
            const int px = i / ST_COB, c = i - px * ST_COB;
            const int oh = th0 + px / ST_TW, ow = tw0 + px % ST_TW;
            float v = 0.f;
            if (oh < OH && ow < OW && co0 + c < Co) v = Elem<T>::ld(dy + (((int64_t)n * OH + oh) * OW + ow) * dy_ld + co0 + c);
            dt[i] = v;
        }
        __syncthreads();
        if (active) {
            for (int py = 0; py < ST_TH / 2; ++py) {
                const float* xr = xt + xoff + py * stride * IW;
                const float4* dr = reinterpret_cast<const float4*>(dt + ((half * (ST_TH / 2) + py) * ST_TW) * ST_COB);
#pragma unroll 4
                for (int px = 0; px < ST_TW; ++px) {
                    const float x = xr[px * stride];
#pragma unroll
                    for (int q = 0; q < ST_COB / 4; ++q) {
                        const float4 d = dr[px * (ST_COB / 4) + q];
                        acc[4 * q + 0] = fmaf(d.x, x, acc[4 * q + 0]);
                        acc[4 * q + 1] = fmaf(d.y, x, acc[4 * q + 1]);
                        acc[4 * q + 2] = fmaf(d.z, x, acc[4 * q + 2]);
                        acc[4 * q + 3] = fmaf(d.w, x, acc[4 * q + 3]);
                    }
                }
            }
        }
    }
    if (active) {
#pragma unroll
        for (int c = 0; c < ST_COB; ++c)
            if (co0 + c < Co) atomicAdd(dw + (int64_t)(co0 + c) * os_co + (int64_t)ci * os_ci + (int64_t)(kh * KW + kw) * os_tap, acc[c]);
    }
}

// shared launcher: returns false when the shape does not fit the small-channel kernel
template <typename T>
static bool launch_small_wgrad(const void* x, bool x_nchw, const T* dy, float* dw, int N, int Ci, int x_ld, int H, int W, int Co,
                               int dy_ld, int KH, int KW, int stride, int pad, int OH, int OW, int os_co, int os_ci, int os_tap,
                               hipStream_t st) {
    if (Ci * KH * KW * 2 > SW_THREADS || (stride != 1 && stride != 2) || KH > ST_MAXK || KW > ST_MAXK) return false;
    const int tiles_h = cdiv(OH, ST_TH), tiles_w = cdiv(OW, ST_TW);
    const int IH = (ST_TH - 1) * stride + KH, IW = (ST_TW - 1) * stride + KW;
    const size_t smem = (size_t)(ST_TH * ST_TW * ST_COB + Ci * IH * IW) * sizeof(float);
    if (smem > 64 * 1024) return false;
    const int64_t ntiles = (int64_t)N * tiles_h * tiles_w;
    const int gx = (int)(ntiles < 1024 ? ntiles : 1024);
    dim3 grid(gx, cdiv(Co, ST_COB));
    if (x_nchw)
        hipLaunchKernelGGL((small_wgrad_kernel<T, true>), grid, dim3(SW_THREADS), smem, st, x, dy, dw, N, Ci, x_ld, H, W, Co, dy_ld,
                           KH, KW, stride, pad, OH, OW, tiles_h, tiles_w, os_co, os_ci, os_tap);
    else
        hipLaunchKernelGGL((small_wgrad_kernel<T, false>), grid, dim3(SW_THREADS), smem, st, x, dy, dw, N, Ci, x_ld, H, W, Co, dy_ld,
                           KH, KW, stride, pad, OH, OW, tiles_h, tiles_w, os_co, os_ci, os_tap);
    return true;
}

// used by cn_conv2d_wgrad for Ci <= 16 (packed output layout dwp[co][tap*Ci + ci])
bool wgrad_c16_nhwc_launch(const void* x, const void* dy, float* dwp, int N, int H, int W, int Ci, int x_ld, int Co, int dy_ld,
                           int stride, int OH, int OW, hipStream_t st, const float* pre_ss, int pre_relu, int target);
bool wgrad_c16_stem_launch(const float* x, const void* dy, float* dw, int N, int Ci, int H, int W, int Co, int dy_ld, int stride,
                           int OH, int OW, hipStream_t st, const void* bn_x, const float* bn_coef, int bn_relu, int target);
bool stem7_fwd_launch(const float* x, const float* w, const float* scale, const float* bias, int relu, void* y, int N, int Ci, int H, int W, int Co,
                      int stride, int OH, int OW, float* bn_part, int bn_slots, int* bn_taken, hipStream_t st);

bool small_wgrad_packed(const void* x, const void* dy, float* dwp, int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co,
                        int dy_ld, int KH, int KW, int stride, int pad, int dtype, hipStream_t st, const float* pre_ss, int pre_relu, int target) {
    if (Ci > 16) return false;
    if (dtype == CN_BF16 && KH == 3 && KW == 3 && pad == 1 && OH == (H - 1) / stride + 1 && OW == (W - 1) / stride + 1 &&
        wgrad_c16_nhwc_launch(x, dy, dwp, N, H, W, Ci, x_ld, Co, dy_ld, stride, OH, OW, st, pre_ss, pre_relu, target))
        return true;
    if (pre_ss) return false;                    // only the MFMA kernel above has the pre-affine hook
    if (dtype == CN_F32)
        return launch_small_wgrad<float>(x, false, (const float*)dy, dwp, N, Ci, x_ld, H, W, Co, dy_ld, KH, KW, stride, pad, OH, OW,
                                         KH * KW * Ci, 1, Ci, st);
    return launch_small_wgrad<bf16_t>(x, false, (const bf16_t*)dy, dwp, N, Ci, x_ld, H, W, Co, dy_ld, KH, KW, stride, pad, OH, OW,
                                      KH * KW * Ci, 1, Ci, st);
}

static int stem_check(int Ci, int KH, int KW, int stride) {
    if (Ci > ST_MAXCI || KH > ST_MAXK || KW > ST_MAXK || (stride != 1 && stride != 2)) {
        cn_set_error("cn_stem_conv: Ci<=%d, kernel<=%d, stride 1|2 only (got Ci=%d k=%dx%d s=%d)", ST_MAXCI, ST_MAXK, Ci, KH,
                     KW, stride);
        return CN_EUNSUPPORTED;
    }
    return CN_OK;
}

extern "C" int cn_stem_conv_fwd_h(const float* x, const float* w, const float* scale, const float* bias, void* y, int N, int Ci, int H,
                                  int W, int Co, int KH, int KW, int stride, int pad, int OH, int OW, int relu, int dtype, cn_hooks* hooks,
                                  void* stream) {
    const BnSink sink = hooks_sink(hooks);       // BatchNorm statistics sink of this call (cn_hooks.bn_part)
    CN_CHECK_ARG(x && w && y && N > 0 && Co > 0, "cn_stem_conv_fwd: bad args");
    int rc = stem_check(Ci, KH, KW, stride);
    if (rc) return rc;
    const bool sink_ok = sink.part && dtype == CN_BF16 && sink.C == Co;
    if (dtype == CN_BF16 && KH == 7 && KW == 7 && pad == 3 && OH == (H + 6 - 7) / stride + 1 && OW == (W + 6 - 7) / stride + 1 &&
        stem7_fwd_launch(x, w, scale, bias, relu, y, N, Ci, H, W, Co, stride, OH, OW, sink_ok ? sink.part : nullptr, sink.slots, hooks ? &hooks->bn_taken : nullptr, (hipStream_t)stream)) {
        CN_LAUNCH_CHECK("cn_stem_conv_fwd(mfma)");
        return CN_OK;
    }
    int tiles_h = cdiv(OH, ST_TH), tiles_w = cdiv(OW, ST_TW);
    int IH = (ST_TH - 1) * stride + KH, IW = (ST_TW - 1) * stride + KW;
    size_t smem = (size_t)(Ci * IH * IW + Ci * KH * KW * ST_COB) * sizeof(float);
    dim3 grid(tiles_h * tiles_w, cdiv(Co, ST_COB), N);
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(stem_fwd_kernel<T>, grid, dim3(256), smem, (hipStream_t)stream, x, w, (T*)y,
                                                   Ci, H, W, Co, KH, KW, stride, pad, OH, OW, tiles_w, scale, bias, relu));
    CN_LAUNCH_CHECK("cn_stem_conv_fwd");
    return CN_OK;
}
extern "C" int cn_stem_conv_fwd(const float* x, const float* w, const float* scale, const float* bias, void* y, int N, int Ci, int H,
                                int W, int Co, int KH, int KW, int stride, int pad, int OH, int OW, int relu, int dtype, void* stream) {
    return cn_stem_conv_fwd_h(x, w, scale, bias, y, N, Ci, H, W, Co, KH, KW, stride, pad, OH, OW, relu, dtype, nullptr, stream);
}

extern "C" int cn_stem_conv_wgrad_h(const float* x, const void* dy, float* dw, int N, int Ci, int H, int W, int Co, int KH,
                                    int KW, int stride, int pad, int OH, int OW, int dtype, cn_hooks* hooks, void* stream) {
    CN_CHECK_ARG(x && dy && dw && N > 0 && Co > 0, "cn_stem_conv_wgrad: bad args");
    int rc = stem_check(Ci, KH, KW, stride);
    if (rc) return rc;
    bool ok;
    if (dtype == CN_BF16 && KH == 7 && KW == 7 && pad == 3 && OH == (H - 1) / stride + 1 && OW == (W - 1) / stride + 1 &&
        wgrad_c16_stem_launch(x, dy, dw, N, Ci, H, W, Co, Co, stride, OH, OW, (hipStream_t)stream, nullptr, nullptr, 0, cn_wgrad_target(hooks)))
        ok = true;
    else if (dtype == CN_F32)
        ok = launch_small_wgrad<float>(x, true, (const float*)dy, dw, N, Ci, 0, H, W, Co, Co, KH, KW, stride, pad, OH, OW,
                                       Ci * KH * KW, KH * KW, 1, (hipStream_t)stream);
    else if (dtype == CN_BF16)
        ok = launch_small_wgrad<bf16_t>(x, true, (const bf16_t*)dy, dw, N, Ci, 0, H, W, Co, Co, KH, KW, stride, pad, OH, OW,
                                        Ci * KH * KW, KH * KW, 1, (hipStream_t)stream);
    else
        CN_CHECK_ARG(false, "cn_stem_conv_wgrad: bad dtype %d", dtype);
    if (!ok) CN_UNSUPPORTED("cn_stem_conv_wgrad: shape does not fit the small-channel kernel");
    CN_LAUNCH_CHECK("cn_stem_conv_wgrad");
    return CN_OK;
}
extern "C" int cn_stem_conv_wgrad(const float* x, const void* dy, float* dw, int N, int Ci, int H, int W, int Co, int KH,
                                  int KW, int stride, int pad, int OH, int OW, int dtype, void* stream) {
    return cn_stem_conv_wgrad_h(x, dy, dw, N, Ci, H, W, Co, KH, KW, stride, pad, OH, OW, dtype, nullptr, stream);
}

// cn_stem_conv_wgrad THROUGH the training-mode BatchNorm (+ ReLU) that follows the stem (pose_dla_dcn.py:283-287 base_layer): dy is
// the gradient w.r.t. the BN OUTPUT, y_raw the stem's own (raw) output, coef = fp32 [5][Co] from cn_bn_bwd_coef_sink.  The kernel
// forms the BN input gradient on load — g = relu ? (fma(y_raw, sc, sh) > 0 ? dy : 0) : dy, dx = fma(ca, g, fma(cp, y_raw, cq)), rounded
// to bf16 like the tensor cn_bn_train_bwd_sink would have stored — so that tensor is never written or read.  bf16, 7x7 / pad 3, Ci <= 3,
// Co a multiple of 16 only (CN_EUNSUPPORTED otherwise: the caller runs cn_bn_train_bwd_sink + cn_stem_conv_wgrad).
extern "C" int cn_stem_conv_wgrad_bn_h(const float* x, const void* dy, const void* y_raw, const float* coef, float* dw, int N, int Ci,
                                       int H, int W, int Co, int KH, int KW, int stride, int pad, int OH, int OW, int relu, int dtype,
                                       cn_hooks* hooks, void* stream) {
    CN_CHECK_ARG(x && dy && y_raw && coef && dw && N > 0 && Co > 0, "cn_stem_conv_wgrad_bn: bad args");
    CN_CHECK_ARG((((uintptr_t)dy | (uintptr_t)y_raw) & 15) == 0, "cn_stem_conv_wgrad_bn: dy / y_raw must be 16-byte aligned");
    int rc = stem_check(Ci, KH, KW, stride);
    if (rc) return rc;
    if (!(dtype == CN_BF16 && KH == 7 && KW == 7 && pad == 3 && OH == (H - 1) / stride + 1 && OW == (W - 1) / stride + 1 &&
          wgrad_c16_stem_launch(x, dy, dw, N, Ci, H, W, Co, Co, stride, OH, OW, (hipStream_t)stream, y_raw, coef, relu, cn_wgrad_target(hooks))))
        CN_UNSUPPORTED("cn_stem_conv_wgrad_bn: bf16, 7x7 / pad 3, Ci <= 3, Co a multiple of 16");
    CN_LAUNCH_CHECK("cn_stem_conv_wgrad_bn");
    return CN_OK;
}
extern "C" int cn_stem_conv_wgrad_bn(const float* x, const void* dy, const void* y_raw, const float* coef, float* dw, int N, int Ci, int H,
                                     int W, int Co, int KH, int KW, int stride, int pad, int OH, int OW, int relu, int dtype, void* stream) {
    return cn_stem_conv_wgrad_bn_h(x, dy, y_raw, coef, dw, N, Ci, H, W, Co, KH, KW, stride, pad, OH, OW, relu, dtype, nullptr, stream);
}
