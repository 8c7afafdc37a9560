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
                                                       int stride, int pad, int OH, int OW, int tiles_w) {
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

// dW[co][ci][kh][kw] += sum_pix dy[pix][co] * x[ci][oh*s-p+kh][ow*s-p+kw]; every workgroup walks many pixel tiles,
// keeps its (tap, co) partial sums in registers and flushes them with one round of atomics.
#define ST_WG_ACC 10   // ceil(3*7*7*16 / 256)
template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                                                         float* __restrict__ dw, int N, int Ci, int H, int W, int Co,
                                                         int KH, int KW, int stride, int pad, int OH, int OW,
                                                         int tiles_h, int tiles_w) {
    extern __shared__ float smem[];
    const int IH = (ST_TH - 1) * stride + KH, IW = (ST_TW - 1) * stride + KW;
    float* xt = smem;                          // [Ci][IH][IW]
    float* dt = smem + Ci * IH * IW;           // [256 pixels][ST_COB]
    const int tid = threadIdx.x;
    const int co0 = blockIdx.y * ST_COB;
    const int ntap = Ci * KH * KW;
    const int nout = ntap * ST_COB;
    float acc[ST_WG_ACC];
    int o_xoff[ST_WG_ACC], o_co[ST_WG_ACC];
#pragma unroll
    for (int i = 0; i < ST_WG_ACC; ++i) {
        acc[i] = 0.f;
        int o = tid + i * 256;
        int t = o / ST_COB;
        o_co[i] = o % ST_COB;
        if (o < nout) {
            int ci = t / (KH * KW), r = t - ci * KH * KW;
            int kh = r / KW, kw = r - kh * KW;
            o_xoff[i] = (ci * IH + kh) * IW + kw;
        } else {
            o_xoff[i] = -1;
        }
    }
    const int64_t ntiles = (int64_t)N * tiles_h * tiles_w;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = (int)(tile / (tiles_h * tiles_w));
        const int r = (int)(tile - (int64_t)n * tiles_h * tiles_w);
        const int th0 = (r / tiles_w) * ST_TH, tw0 = (r % tiles_w) * ST_TW;
        const int ih0 = th0 * stride - pad, iw0 = tw0 * stride - pad;
        __syncthreads();
        for (int i = tid; i < Ci * IH * IW; i += 256) {
            int c = i / (IH * IW), rr = i - c * IH * IW;
            int hh = rr / IW, ww = rr - hh * IW;
            int ih = ih0 + hh, iw = iw0 + ww;
            float v = 0.f;
            if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = x[(((int64_t)n * Ci + c) * H + ih) * W + iw];
            xt[i] = v;
        }
        {
            const int ty = tid / ST_TW, tx = tid % ST_TW;
            const int oh = th0 + ty, ow = tw0 + tx;
            const bool ok = oh < OH && ow < OW;
            const T* src = dy + (((int64_t)n * OH + oh) * OW + ow) * Co + co0;
            for (int c = 0; c < ST_COB; ++c) dt[tid * ST_COB + c] = (ok && co0 + c < Co) ? Elem<T>::ld(src + c) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ST_WG_ACC; ++i) {
            if (o_xoff[i] < 0) continue;
            const float* xb = xt + o_xoff[i];
            const float* db = dt + o_co[i];
            float s = 0.f;
            for (int py = 0; py < ST_TH; ++py) {
                const float* xr = xb + py * stride * IW;
                const float* dr = db + py * ST_TW * ST_COB;
#pragma unroll 8
                for (int px = 0; px < ST_TW; ++px) s = fmaf(dr[px * ST_COB], xr[px * stride], s);
            }
            acc[i] += s;
        }
    }
#pragma unroll
    for (int i = 0; i < ST_WG_ACC; ++i) {
        int o = tid + i * 256;
        if (o_xoff[i] >= 0 && co0 + o_co[i] < Co) atomicAdd(dw + (int64_t)(co0 + o_co[i]) * ntap + o / ST_COB, acc[i]);
    }
}

static int stem_check(int Ci, int KH, int KW, int stride) {
    if (Ci > ST_MAXCI || KH > ST_MAXK || KW > ST_MAXK || (stride != 1 && stride != 2)) {
        cn_set_error("cn_stem_conv: Ci<=%d, kernel<=%d, stride 1|2 only (got Ci=%d k=%dx%d s=%d)", ST_MAXCI, ST_MAXK, Ci, KH,
                     KW, stride);
        return CN_EUNSUPPORTED;
    }
    return CN_OK;
}

extern "C" int cn_stem_conv_fwd(const float* x, const float* w, void* y, int N, int Ci, int H, int W, int Co, int KH,
                                int KW, int stride, int pad, int OH, int OW, int dtype, void* stream) {
    CN_CHECK_ARG(x && w && y && N > 0 && Co > 0, "cn_stem_conv_fwd: bad args");
    int rc = stem_check(Ci, KH, KW, stride);
    if (rc) return rc;
    int tiles_h = cdiv(OH, ST_TH), tiles_w = cdiv(OW, ST_TW);
    int IH = (ST_TH - 1) * stride + KH, IW = (ST_TW - 1) * stride + KW;
    size_t smem = (size_t)(Ci * IH * IW + Ci * KH * KW * ST_COB) * sizeof(float);
    dim3 grid(tiles_h * tiles_w, cdiv(Co, ST_COB), N);
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(stem_fwd_kernel<T>, grid, dim3(256), smem, (hipStream_t)stream, x, w, (T*)y,
                                                   Ci, H, W, Co, KH, KW, stride, pad, OH, OW, tiles_w));
    CN_LAUNCH_CHECK("cn_stem_conv_fwd");
    return CN_OK;
}

extern "C" int cn_stem_conv_wgrad(const float* x, const void* dy, float* dw, int N, int Ci, int H, int W, int Co, int KH,
                                  int KW, int stride, int pad, int OH, int OW, int dtype, void* stream) {
    CN_CHECK_ARG(x && dy && dw && N > 0 && Co > 0, "cn_stem_conv_wgrad: bad args");
    int rc = stem_check(Ci, KH, KW, stride);
    if (rc) return rc;
    if (Ci * KH * KW * ST_COB > ST_WG_ACC * 256) CN_UNSUPPORTED("cn_stem_conv_wgrad: too many taps");
    int tiles_h = cdiv(OH, ST_TH), tiles_w = cdiv(OW, ST_TW);
    int IH = (ST_TH - 1) * stride + KH, IW = (ST_TW - 1) * stride + KW;
    size_t smem = (size_t)(Ci * IH * IW + 256 * ST_COB) * sizeof(float);
    int64_t ntiles = (int64_t)N * tiles_h * tiles_w;
    int gx = (int)(ntiles < 1024 ? ntiles : 1024);
    dim3 grid(gx, cdiv(Co, ST_COB));
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(stem_wgrad_kernel<T>, grid, dim3(256), smem, (hipStream_t)stream, x,
                                                   (const T*)dy, dw, N, Ci, H, W, Co, KH, KW, stride, pad, OH, OW, tiles_h,
                                                   tiles_w));
    CN_LAUNCH_CHECK("cn_stem_conv_wgrad");
    return CN_OK;
}
