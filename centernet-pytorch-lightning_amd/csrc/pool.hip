// Max-pool (DLA Tree.downsample 2x2/s2: pose_dla_dcn.py:243; ResNet stem 3x3/s2/p1: msra_resnet.py:113) and the
// depthwise bilinear ConvTranspose2d of IDAUp (pose_dla_dcn.py:466-475).  NHWC, one 16-byte channel
// vector per lane: pure HBM-bound gathers, no LDS needed (neighbouring taps hit L1/L2).
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, unsigned char* __restrict__ idx,
                                                          int N, int H, int W, int CV, int k, int s, int p, int OH, int OW) {
    CN_MAIN_PRIO_SET();
    constexpr int V = Vec16<T>::N;
    const int64_t total = (int64_t)N * OH * OW * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t iu = (uint32_t)i;                       // 32-bit index arithmetic: three 64-bit divisions per vector made these kernels ALU bound
        const int cv = (int)(iu % (uint32_t)CV);
        uint32_t pix = iu / (uint32_t)CV;
        const int ow = (int)(pix % (uint32_t)OW);
        pix /= (uint32_t)OW;
        const int oh = (int)(pix % (uint32_t)OH), n = (int)(pix / (uint32_t)OH);
        float m[V];
        int am[V];                                   // window position kh*k + kw of the FIRST maximum (ATen's tie rule)
#pragma unroll
        for (int j = 0; j < V; ++j) { m[j] = -INFINITY; am[j] = -1; }
        for (int kh = 0; kh < k; ++kh) {
            const int ih = oh * s - p + kh;
            if ((unsigned)ih >= (unsigned)H) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int iw = ow * s - p + kw;
                if ((unsigned)iw >= (unsigned)W) continue;
                float v[V];
                Vec16<T>::load(x + ((((int64_t)n * H + ih) * W + iw) * CV + cv) * V, v);
#pragma unroll
                for (int j = 0; j < V; ++j)
                    if (v[j] > m[j] || am[j] < 0) { m[j] = v[j]; am[j] = kh * k + kw; }
            }
        }
        Vec16<T>::store(y + i * V, m);
        if (idx) {
            unsigned char* d = idx + i * V;          // V bytes: 8 (bf16) or 4 (fp32), naturally aligned
            if constexpr (V == 8) {
                *reinterpret_cast<uint2*>(d) = make_uint2((am[0] & 255) | (am[1] & 255) << 8 | (am[2] & 255) << 16 | (unsigned)(am[3] & 255) << 24,
                                                          (am[4] & 255) | (am[5] & 255) << 8 | (am[6] & 255) << 16 | (unsigned)(am[7] & 255) << 24);
            } else {
                *reinterpret_cast<uint32_t*>(d) = (am[0] & 255) | (am[1] & 255) << 8 | (am[2] & 255) << 16 | (unsigned)(am[3] & 255) << 24;
            }
        }
    }
}

// gather form: every input element sums dy of the windows whose recorded arg-max (`idx`, written by the forward) it is — the
// windows' x values are not re-read (the first version recomputed every arg-max from x: 2.2x the traffic of this one).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const unsigned char* __restrict__ idx, const T* __restrict__ dy,
                                                          const T* __restrict__ acc, T* __restrict__ dx, int N, int H, int W,
                                                          int CV, int k, int s, int p, int OH, int OW) {
    CN_MAIN_PRIO_SET();
    constexpr int V = Vec16<T>::N;
    const int64_t total = (int64_t)N * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t iu = (uint32_t)i;                       // 32-bit index arithmetic: three 64-bit divisions per vector made these kernels ALU bound
        const int cv = (int)(iu % (uint32_t)CV);
        uint32_t pix = iu / (uint32_t)CV;
        const int iw = (int)(pix % (uint32_t)W);
        pix /= (uint32_t)W;
        const int ih = (int)(pix % (uint32_t)H), n = (int)(pix / (uint32_t)H);
        float g[V];
        if (acc) {                                             // x is a shared tensor: start from what its other consumers sent
            Vec16<T>::load(acc + i * V, g);
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) g[j] = 0.f;
        }
        int oh_lo = ih + p - k + 1;
        oh_lo = oh_lo > 0 ? (oh_lo + s - 1) / s : 0;
        int oh_hi = (ih + p) / s;
        if (oh_hi > OH - 1) oh_hi = OH - 1;
        int ow_lo = iw + p - k + 1;
        ow_lo = ow_lo > 0 ? (ow_lo + s - 1) / s : 0;
        int ow_hi = (iw + p) / s;
        if (ow_hi > OW - 1) ow_hi = OW - 1;
        for (int oh = oh_lo; oh <= oh_hi; ++oh)
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const int mine = (ih - (oh * s - p)) * k + (iw - (ow * s - p));      // my position inside that window
                const int64_t o = ((((int64_t)n * OH + oh) * OW + ow) * CV + cv) * V;
                unsigned int a[2] = {0, 0};
                if constexpr (V == 8) { const uint2 t = *reinterpret_cast<const uint2*>(idx + o); a[0] = t.x; a[1] = t.y; }
                else a[0] = *reinterpret_cast<const uint32_t*>(idx + o);
                float d[V];
                Vec16<T>::load(dy + o, d);
#pragma unroll
                for (int j = 0; j < V; ++j)
                    if ((int)((a[j >> 2] >> (8 * (j & 3))) & 255u) == mine) g[j] += d[j];
            }
        Vec16<T>::store(dx + i * V, g);
    }
}

// y[n,oh,ow,c] = sum_{kh,kw : (oh+p-kh) % s == 0 ...} x[n,(oh+p-kh)/s,(ow+p-kw)/s,c] * w[c,kh,kw]
// The [C][k][k] fp32 weights are staged in LDS TRANSPOSED to [tap][C]: a lane then reads the 8 (or 4) weights of its
// channel vector with one or two ds_read_b128 instead of 8 strided global loads per tap (those loads, not the data
// stream, were what bounded these kernels).  Dynamic LDS = k*k*C floats.
__device__ static inline void dw_stage_weights(const float* __restrict__ w, float* wl, int C, int kk) {
    for (int i = threadIdx.x; i < C * kk; i += blockDim.x) {
        const int c = i / kk, t = i - c * kk;
        wl[t * C + c] = w[i];
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void dwdeconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const T* __restrict__ res,
                                                           T* __restrict__ y, int N, int H, int W, int CV, int k, int s,
                                                           int p, int OH, int OW) {
    CN_MAIN_PRIO_SET();
    constexpr int V = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float wl[];
    const int C = CV * V;
    dw_stage_weights(w, wl, C, k * k);
    const int64_t total = (int64_t)N * OH * OW * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t iu = (uint32_t)i;                       // 32-bit index arithmetic: three 64-bit divisions per vector made these kernels ALU bound
        const int cv = (int)(iu % (uint32_t)CV);
        uint32_t pix = iu / (uint32_t)CV;
        const int ow = (int)(pix % (uint32_t)OW);
        pix /= (uint32_t)OW;
        const int oh = (int)(pix % (uint32_t)OH), n = (int)(pix / (uint32_t)OH);
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
        if (k == 2 * s) {
            // k = 2 * stride (every up-conv of the reference: k4 s2, k8 s4): exactly 2 x 2 taps reach an output pixel.  All four
            // x loads and the residual load are issued together, branch-free (clamped address + zero weight): the tap loops
            // below compile to exec-masked branches with a full wait after each load (2.7 TB/s on 64ch @64^2 -> 128^2).
            const int kh0 = (oh + p) % s, kw0 = (ow + p) % s;
            uint4 xr[4], rr;
            bool ok[4];
            int tap[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kh = kh0 + (t >> 1) * s, kw = kw0 + (t & 1) * s;
                const int dh = oh + p - kh, dw_ = ow + p - kw;
                const int ih = dh / s, iw = dw_ / s;                   // dh, dw_ are multiples of s by construction
                ok[t] = dh >= 0 && ih < H && dw_ >= 0 && iw < W;
                tap[t] = kh * k + kw;
                const int ihc = ok[t] ? ih : 0, iwc = ok[t] ? iw : 0;
                xr[t] = ldg16(x + ((((int64_t)n * H + ihc) * W + iwc) * CV + cv) * V);
            }
            if (res) rr = ldg16(res + i * V);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v[V];
                // an out-of-image tap was loaded from the clamped address (n,0,0): zero the DATA, not the weight — 0 * Inf would
                // otherwise put NaN into every border output of that channel (ConvTranspose2d ignores such taps)
                const uint4 xv = ok[t] ? xr[t] : make_uint4(0, 0, 0, 0);
                Vec16<T>::unpack(xv, v);
                const float* wt = wl + tap[t] * C + cv * V;
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] = fmaf(v[j], wt[j], acc[j]);
            }
            if (res) {
                float r[V];
                Vec16<T>::unpack(rr, r);
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] += r[j];
            }
            Vec16<T>::store(y + i * V, acc);
            continue;
        }
        for (int kh = (oh + p) % s; kh < k; kh += s) {
            const int ih = (oh + p - kh) / s;
            if (oh + p - kh < 0 || ih >= H) continue;
            for (int kw = (ow + p) % s; kw < k; kw += s) {
                const int iw = (ow + p - kw) / s;
                if (ow + p - kw < 0 || iw >= W) continue;
                float v[V];
                Vec16<T>::load(x + ((((int64_t)n * H + ih) * W + iw) * CV + cv) * V, v);
                const float* wt = wl + (kh * k + kw) * C + cv * V;
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] = fmaf(v[j], wt[j], acc[j]);
            }
        }
        if (res) {                                  // IDAUp: node(up(proj(x)) + layers[i-1]) — the add rides in this store
            float r[V];
            Vec16<T>::load(res + i * V, r);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += r[j];
        }
        Vec16<T>::store(y + i * V, acc);
    }
}

// dx[n,ih,iw,c] = sum_{kh,kw} dy[n, ih*s-p+kh, iw*s-p+kw, c] * w[c,kh,kw]
template <typename T>
__global__ __launch_bounds__(256) void dwdeconv_bwd_input_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                                 T* __restrict__ dx, int N, int H, int W, int CV, int k,
                                                                 int s, int p, int OH, int OW) {
    CN_MAIN_PRIO_SET();
    constexpr int V = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float wl[];
    const int C = CV * V;
    dw_stage_weights(w, wl, C, k * k);
    const int64_t total = (int64_t)N * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t iu = (uint32_t)i;                       // 32-bit index arithmetic: three 64-bit divisions per vector made these kernels ALU bound
        const int cv = (int)(iu % (uint32_t)CV);
        uint32_t pix = iu / (uint32_t)CV;
        const int iw = (int)(pix % (uint32_t)W);
        pix /= (uint32_t)W;
        const int ih = (int)(pix % (uint32_t)H), n = (int)(pix / (uint32_t)H);
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
        for (int kh = 0; kh < k; ++kh) {
            const int oh = ih * s - p + kh;
            if ((unsigned)oh >= (unsigned)OH) continue;
            const T* row = dy + ((((int64_t)n * OH + oh) * OW) * CV + cv) * V;
            const float* wr = wl + kh * k * C + cv * V;
#pragma unroll 4
            for (int kw = 0; kw < k; ++kw) {
                const int ow = iw * s - p + kw;
                if ((unsigned)ow >= (unsigned)OW) continue;
                float v[V];
                Vec16<T>::load(row + (int64_t)ow * CV * V, v);
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] = fmaf(v[j], wr[kw * C + j], acc[j]);
            }
        }
        Vec16<T>::store(dx + i * V, acc);
    }
}

// dw[c,kh,kw] += sum_{n,ih,iw} x[n,ih,iw,c] * dy[n, ih*s-p+kh, iw*s-p+kw, c]; lane = (tap, channel vector).
// Four pixels per trip keep eight independent 16-byte loads in flight per lane (the one-pixel loop was latency bound).
template <typename T>
__global__ __launch_bounds__(256) void dwdeconv_bwd_weight_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                  float* __restrict__ dw, int N, int H, int W, int CV, int k,
                                                                  int s, int p, int OH, int OW, int64_t chunk) {
    constexpr int V = Vec16<T>::N;
    // a workgroup covers NP = min(256, CV*k*k - 256*blockIdx.y) (tap, channel-vector) pairs; when that leaves lanes idle (64 channels x
    // 16 taps = 128 pairs) the spare lanes take every other group of 4 pixels of the chunk, and the planes meet in LDS before the atomics
    const int npairs = CV * k * k - blockIdx.y * 256;
    const int NP = npairs < 256 ? npairs : 256;
    const int planes = 256 / NP;
    const int plane = threadIdx.x / NP;
    const int pair = blockIdx.y * 256 + threadIdx.x % NP;
    const bool live = plane < planes;
    const int cv = pair % CV, tap = pair / CV;
    const int kh = tap / k, kw = tap - kh * k;
    const int64_t P = (int64_t)N * H * W;
    const int64_t p0 = (int64_t)blockIdx.x * chunk, p1 = p0 + chunk < P ? p0 + chunk : P;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int64_t pb = p0 + 4 * plane; live && pb < p1; pb += 4 * planes) {
        uint4 ra[4], rb[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t pix = pb + u;
            const uint32_t pu = (uint32_t)pix;
            const int iw = (int)(pu % (uint32_t)W);
            const uint32_t t = pu / (uint32_t)W;
            const int ih = (int)(t % (uint32_t)H), n = (int)(t / (uint32_t)H);
            const int oh = ih * s - p + kh, ow = iw * s - p + kw;
            ok[u] = pix < p1 && (unsigned)oh < (unsigned)OH && (unsigned)ow < (unsigned)OW;
            ra[u] = ldg16_masked(x, (pix * CV + cv) * 16, ok[u]);
            rb[u] = ldg16_masked(dy, (((((int64_t)n * OH + oh) * OW + ow) * CV) + cv) * 16, ok[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float a[V], b[V];
            if constexpr (sizeof(T) == 2) {
                const uint32_t wa[4] = {ra[u].x, ra[u].y, ra[u].z, ra[u].w}, wb[4] = {rb[u].x, rb[u].y, rb[u].z, rb[u].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a[2 * q] = __uint_as_float(wa[q] << 16); a[2 * q + 1] = __uint_as_float(wa[q] & 0xffff0000u);
                    b[2 * q] = __uint_as_float(wb[q] << 16); b[2 * q + 1] = __uint_as_float(wb[q] & 0xffff0000u);
                }
            } else {
                a[0] = __uint_as_float(ra[u].x); a[1] = __uint_as_float(ra[u].y); a[2] = __uint_as_float(ra[u].z); a[3] = __uint_as_float(ra[u].w);
                b[0] = __uint_as_float(rb[u].x); b[1] = __uint_as_float(rb[u].y); b[2] = __uint_as_float(rb[u].z); b[3] = __uint_as_float(rb[u].w);
            }
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] = fmaf(a[j], b[j], acc[j]);
        }
    }
    __shared__ float red[256 * 8];
    if (planes > 1) {
#pragma unroll
        for (int j = 0; j < V; ++j) red[threadIdx.x * V + j] = acc[j];
        __syncthreads();
        if (plane == 0)
            for (int q = 1; q < planes; ++q)
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] += red[(q * NP + threadIdx.x) * V + j];
    }
    if (plane == 0) {
#pragma unroll
        for (int j = 0; j < V; ++j) atomicAdd(dw + ((int64_t)(cv * V + j) * k + kh) * k + kw, acc[j]);
    }
}

// ---- nearest-neighbour x2 up-sampling fused with the hourglass merge (large_hourglass.py:108-125, 196-204) ----------------
// y[n, oh, ow, :] = a[n, oh, ow, :] + low[n, oh>>1, ow>>1, :]   (a == nullptr: plain nn.Upsample(scale_factor=2)).
// One thread owns one LOW pixel vector and writes its 2x2 output footprint: low is read once, a/y stream as 16-byte vectors.
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_add_kernel(const T* __restrict__ a, const T* __restrict__ low, T* __restrict__ y,
                                                             int N, int H, int W, int CV) {
    constexpr int V = Vec16<T>::N;
    const int64_t total = (int64_t)N * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t iu = (uint32_t)i;                       // 32-bit index arithmetic: three 64-bit divisions per vector made these kernels ALU bound
        const int cv = (int)(iu % (uint32_t)CV);
        uint32_t pix = iu / (uint32_t)CV;
        const int w = (int)(pix % (uint32_t)W);
        pix /= (uint32_t)W;
        const int h = (int)(pix % (uint32_t)H), n = (int)(pix / (uint32_t)H);
        float lv[V];
        Vec16<T>::load(low + i * V, lv);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int64_t o = ((((int64_t)n * 2 * H + 2 * h + dy) * 2 * W + 2 * w + dx) * CV + cv) * V;
                float v[V];
                if (a) {
                    Vec16<T>::load(a + o, v);
#pragma unroll
                    for (int j = 0; j < V; ++j) v[j] += lv[j];
                    Vec16<T>::store(y + o, v);
                } else
                    Vec16<T>::store(y + o, lv);
            }
    }
}

// adjoint: dlow[n, h, w, :] = sum of the 2x2 footprint of dy (fp32 accumulation)
template <typename T>
__global__ __launch_bounds__(256) void sumpool2x2_kernel(const T* __restrict__ dy, T* __restrict__ dlow, int N, int H, int W, int CV) {
    constexpr int V = Vec16<T>::N;
    const int64_t total = (int64_t)N * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t iu = (uint32_t)i;                       // 32-bit index arithmetic: three 64-bit divisions per vector made these kernels ALU bound
        const int cv = (int)(iu % (uint32_t)CV);
        uint32_t pix = iu / (uint32_t)CV;
        const int w = (int)(pix % (uint32_t)W);
        pix /= (uint32_t)W;
        const int h = (int)(pix % (uint32_t)H), n = (int)(pix / (uint32_t)H);
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
        for (int dy_ = 0; dy_ < 2; ++dy_)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float v[V];
                Vec16<T>::load(dy + ((((int64_t)n * 2 * H + 2 * h + dy_) * 2 * W + 2 * w + dx) * CV + cv) * V, v);
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] += v[j];
            }
        Vec16<T>::store(dlow + i * V, acc);
    }
}

static int pool_grid(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (int)(g > 32768 ? 32768 : (g < 1 ? 1 : g));
}

#define POOL_ARGS_CHECK(name)                                                                                      \
    CN_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0 && k > 0 && stride > 0, name ": bad dims"); \
    const int V = dtype == CN_F32 ? 4 : 8;                                                                         \
    CN_CHECK_ARG(C % V == 0, name ": C=%d must be a multiple of %d", C, V);                                        \
    CN_CHECK_ARG((int64_t)N * (H > OH ? H : OH) * (W > OW ? W : OW) * (C / V) < (1ll << 31), name ": more than 2^31 channel vectors (32-bit index arithmetic)")

extern "C" int cn_maxpool_fwd(const void* x, void* y, unsigned char* argmax, int N, int H, int W, int C, int k, int stride, int pad, int OH,
                              int OW, int dtype, void* stream) {
    CN_CHECK_ARG(x && y, "cn_maxpool_fwd: null");
    POOL_ARGS_CHECK("cn_maxpool_fwd");
    CN_CHECK_ARG(k * k <= 255, "cn_maxpool_fwd: window %dx%d does not fit the 8-bit arg-max", k, k);
    int64_t total = (int64_t)N * OH * OW * (C / V);
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(pool_grid(total)), dim3(256), 0,
                                                   (hipStream_t)stream, (const T*)x, (T*)y, argmax, N, H, W, C / V, k, stride, pad, OH, OW));
    CN_LAUNCH_CHECK("cn_maxpool_fwd");
    return CN_OK;
}

extern "C" int cn_maxpool_bwd_acc(const unsigned char* argmax, const void* dy, const void* acc, void* dx, int N, int H, int W, int C, int k,
                                  int stride, int pad, int OH, int OW, int dtype, void* stream) {
    CN_CHECK_ARG(argmax && dy && dx, "cn_maxpool_bwd: null");
    POOL_ARGS_CHECK("cn_maxpool_bwd");
    int64_t total = (int64_t)N * H * W * (C / V);
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(pool_grid(total)), dim3(256), 0,
                                                   (hipStream_t)stream, argmax, (const T*)dy, (const T*)acc, (T*)dx, N, H, W, C / V, k,
                                                   stride, pad, OH, OW));
    CN_LAUNCH_CHECK("cn_maxpool_bwd");
    return CN_OK;
}

extern "C" int cn_maxpool_bwd(const unsigned char* argmax, const void* dy, void* dx, int N, int H, int W, int C, int k, int stride, int pad,
                              int OH, int OW, int dtype, void* stream) {
    return cn_maxpool_bwd_acc(argmax, dy, nullptr, dx, N, H, W, C, k, stride, pad, OH, OW, dtype, stream);
}

extern "C" int cn_dwdeconv_fwd(const void* x, const float* w, const void* residual, void* y, int N, int H, int W, int C, int k, int stride,
                               int pad, int OH, int OW, int dtype, void* stream) {
    CN_CHECK_ARG(x && w && y, "cn_dwdeconv_fwd: null");
    POOL_ARGS_CHECK("cn_dwdeconv_fwd");
    int64_t total = (int64_t)N * OH * OW * (C / V);
    const size_t wbytes = (size_t)C * k * k * sizeof(float);
    if (wbytes > 160 * 1024 - 1024) CN_UNSUPPORTED("dwdeconv_fwd_kernel: C*k*k = %d weights do not fit LDS", C * k * k);
    if (wbytes > 48 * 1024) {
        if (dtype == CN_F32) (void)hipFuncSetAttribute((const void*)dwdeconv_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wbytes);
        else (void)hipFuncSetAttribute((const void*)dwdeconv_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wbytes);
    }
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(dwdeconv_fwd_kernel<T>, dim3(pool_grid(total) > 2048 ? 2048 : pool_grid(total)), dim3(256),
                                                   (size_t)C * k * k * sizeof(float), (hipStream_t)stream, (const T*)x, w, (const T*)residual, (T*)y, N, H, W, C / V, k, stride, pad, OH, OW));
    CN_LAUNCH_CHECK("cn_dwdeconv_fwd");
    return CN_OK;
}

extern "C" int cn_dwdeconv_bwd_input(const void* dy, const float* w, void* dx, int N, int H, int W, int C, int k, int stride,
                                     int pad, int OH, int OW, int dtype, void* stream) {
    CN_CHECK_ARG(dy && w && dx, "cn_dwdeconv_bwd_input: null");
    POOL_ARGS_CHECK("cn_dwdeconv_bwd_input");
    int64_t total = (int64_t)N * H * W * (C / V);
    const size_t wbytes = (size_t)C * k * k * sizeof(float);
    if (wbytes > 160 * 1024 - 1024) CN_UNSUPPORTED("dwdeconv_bwd_input_kernel: C*k*k = %d weights do not fit LDS", C * k * k);
    if (wbytes > 48 * 1024) {
        if (dtype == CN_F32) (void)hipFuncSetAttribute((const void*)dwdeconv_bwd_input_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wbytes);
        else (void)hipFuncSetAttribute((const void*)dwdeconv_bwd_input_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wbytes);
    }
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(dwdeconv_bwd_input_kernel<T>, dim3(pool_grid(total) > 2048 ? 2048 : pool_grid(total)), dim3(256),
                                                   (size_t)C * k * k * sizeof(float), (hipStream_t)stream, (const T*)dy, w, (T*)dx, N, H, W, C / V, k, stride, pad, OH, OW));
    CN_LAUNCH_CHECK("cn_dwdeconv_bwd_input");
    return CN_OK;
}

extern "C" int cn_dwdeconv_bwd_weight(const void* x, const void* dy, float* dw, int N, int H, int W, int C, int k, int stride,
                                      int pad, int OH, int OW, int dtype, void* stream) {
    CN_CHECK_ARG(x && dy && dw, "cn_dwdeconv_bwd_weight: null");
    POOL_ARGS_CHECK("cn_dwdeconv_bwd_weight");
    int64_t P = (int64_t)N * H * W;
    const int nchunks = (C / V) * k * k <= 128 ? 1024 : 512;   // spare lanes split a chunk, so the atomics per pixel stay the same
    int64_t chunk = (P + nchunks - 1) / nchunks;
    if (chunk < 64) chunk = 64;
    chunk = (chunk + 3) / 4 * 4;
    dim3 grid(cdiv(P, chunk), cdiv((C / V) * k * k, 256));
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(dwdeconv_bwd_weight_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)x, (const T*)dy, dw, N, H, W, C / V, k, stride, pad, OH, OW, chunk));
    CN_LAUNCH_CHECK("cn_dwdeconv_bwd_weight");
    return CN_OK;
}

extern "C" int cn_upsample2x_add(const void* a, const void* low, void* y, int N, int H, int W, int C, int dtype, void* stream) {
    CN_CHECK_ARG(low && y && N > 0 && H > 0 && W > 0 && C > 0, "cn_upsample2x_add: bad args");
    const int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(C % V == 0, "cn_upsample2x_add: C=%d must be a multiple of %d", C, V);
    int64_t total = (int64_t)N * H * W * (C / V);
    CN_CHECK_ARG(total * 4 < (1ll << 31), "cn_upsample2x_add: more than 2^31 channel vectors");
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(upsample2x_add_kernel<T>, dim3(pool_grid(total)), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)a, (const T*)low, (T*)y, N, H, W, C / V));
    CN_LAUNCH_CHECK("cn_upsample2x_add");
    return CN_OK;
}

extern "C" int cn_sumpool2x2(const void* dy, void* dlow, int N, int H, int W, int C, int dtype, void* stream) {
    CN_CHECK_ARG(dy && dlow && N > 0 && H > 0 && W > 0 && C > 0, "cn_sumpool2x2: bad args");
    const int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(C % V == 0, "cn_sumpool2x2: C=%d must be a multiple of %d", C, V);
    int64_t total = (int64_t)N * H * W * (C / V);
    CN_CHECK_ARG(total * 4 < (1ll << 31), "cn_sumpool2x2: more than 2^31 channel vectors");
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(sumpool2x2_kernel<T>, dim3(pool_grid(total)), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)dy, (T*)dlow, N, H, W, C / V));
    CN_LAUNCH_CHECK("cn_sumpool2x2");
    return CN_OK;
}
