// DCNv2 (modulated deformable 3x3 conv, stride 1, pad 1, dilation 1, 1 deformable group) — the sampling half.
// The contraction half runs on the implicit-GEMM engine as a 1x1 conv over the sampled columns
//   col[p][k*Ci + c] = sigmoid(mask_logit[p,k]) * bilinear(x[n,:,:,c], h-1+i+dy[p,k], w-1+j+dx[p,k]),  k = 3i+j
// NHWC makes each bilinear corner one contiguous channel vector, so both kernels are plain 16-byte
// gathers (HBM/L2-bound).  Offsets are interleaved (dy,dx) per tap; mask logits follow.
#include "common.h"

struct Tap {
    float w00, w01, w10, w11;   // bilinear corner weights (0 where the corner is outside the image)
    float lh, lw;
    int h0, w0;
    bool ok00, ok01, ok10, ok11;
};

__device__ static inline Tap make_tap(float py, float px, int H, int W) {
    Tap t;
    const float fh = floorf(py), fw = floorf(px);
    t.h0 = (int)fh;
    t.w0 = (int)fw;
    t.lh = py - fh;
    t.lw = px - fw;
    const bool h0ok = t.h0 >= 0 && t.h0 <= H - 1, h1ok = t.h0 + 1 >= 0 && t.h0 + 1 <= H - 1;
    const bool w0ok = t.w0 >= 0 && t.w0 <= W - 1, w1ok = t.w0 + 1 >= 0 && t.w0 + 1 <= W - 1;
    t.ok00 = h0ok && w0ok; t.ok01 = h0ok && w1ok; t.ok10 = h1ok && w0ok; t.ok11 = h1ok && w1ok;
    t.w00 = t.ok00 ? (1.f - t.lh) * (1.f - t.lw) : 0.f;
    t.w01 = t.ok01 ? (1.f - t.lh) * t.lw : 0.f;
    t.w10 = t.ok10 ? t.lh * (1.f - t.lw) : 0.f;
    t.w11 = t.ok11 ? t.lh * t.lw : 0.f;
    return t;
}

__device__ static inline float sigmoidf_(float z) { return 1.f / (1.f + __expf(-z)); }

template <typename T>
__global__ __launch_bounds__(256) void dcn_im2col_kernel(const T* __restrict__ x, const float* __restrict__ om,
                                                         T* __restrict__ col, int N, int H, int W, int Ci, int x_ld,
                                                         int om_ld) {
    constexpr int V = Vec16<T>::N;
    const int CV = Ci / V;
    const int64_t total = (int64_t)N * H * W * 9 * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int64_t r = i / CV;
        const int k = (int)(r % 9);
        const int64_t p = r / 9;
        const int w = (int)(p % W);
        const int64_t q = p / W;
        const int h = (int)(q % H), n = (int)(q / H);
        const float* o = om + p * om_ld;
        const float py = (float)(h - 1 + k / 3) + o[2 * k];
        const float px = (float)(w - 1 + k % 3) + o[2 * k + 1];
        const float m = sigmoidf_(o[18 + k]);
        const Tap t = make_tap(py, px, H, W);
        const T* xb = x + (int64_t)n * H * W * x_ld + cv * V;
        float acc[V], v[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
        if (t.ok00) { Vec16<T>::load(xb + ((int64_t)t.h0 * W + t.w0) * x_ld, v);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += v[j] * t.w00; }
        if (t.ok01) { Vec16<T>::load(xb + ((int64_t)t.h0 * W + t.w0 + 1) * x_ld, v);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += v[j] * t.w01; }
        if (t.ok10) { Vec16<T>::load(xb + ((int64_t)(t.h0 + 1) * W + t.w0) * x_ld, v);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += v[j] * t.w10; }
        if (t.ok11) { Vec16<T>::load(xb + ((int64_t)(t.h0 + 1) * W + t.w0 + 1) * x_ld, v);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += v[j] * t.w11; }
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] *= m;
        Vec16<T>::store(col + (p * 9 + k) * Ci + cv * V, acc);
    }
}

// One lane group (GS lanes, power of two <= 64) per (pixel, tap); lanes stride over the channel vectors.
template <typename T>
__global__ __launch_bounds__(256) void dcn_col2im_kernel(const T* __restrict__ dcol, const T* __restrict__ x,
                                                         const float* __restrict__ om, float* __restrict__ dx,
                                                         float* __restrict__ dom, int N, int H, int W, int Ci, int x_ld,
                                                         int om_ld, int GS) {
    constexpr int V = Vec16<T>::N;
    const int CV = Ci / V;
    const int64_t ngroups = (int64_t)N * H * W * 9;
    const int gpb = 256 / GS;  // groups per workgroup
    const int gl = threadIdx.x / GS, lg = threadIdx.x % GS;
    for (int64_t g = (int64_t)blockIdx.x * gpb + gl; g < ngroups; g += (int64_t)gridDim.x * gpb) {
        const int k = (int)(g % 9);
        const int64_t p = g / 9;
        const int w = (int)(p % W);
        const int64_t q = p / W;
        const int h = (int)(q % H), n = (int)(q / H);
        const float* o = om + p * om_ld;
        const float py = (float)(h - 1 + k / 3) + o[2 * k];
        const float px = (float)(w - 1 + k % 3) + o[2 * k + 1];
        const float m = sigmoidf_(o[18 + k]);
        const Tap t = make_tap(py, px, H, W);
        const int64_t img = (int64_t)n * H * W;
        float s_m = 0.f, s_y = 0.f, s_x = 0.f;
        for (int cv = lg; cv < CV; cv += GS) {
            float gcol[V], x00[V], x01[V], x10[V], x11[V];
            Vec16<T>::load(dcol + (p * 9 + k) * Ci + cv * V, gcol);
#pragma unroll
            for (int j = 0; j < V; ++j) { x00[j] = 0.f; x01[j] = 0.f; x10[j] = 0.f; x11[j] = 0.f; }
            const int64_t i00 = img + (int64_t)t.h0 * W + t.w0;
            if (t.ok00) Vec16<T>::load(x + i00 * x_ld + cv * V, x00);
            if (t.ok01) Vec16<T>::load(x + (i00 + 1) * x_ld + cv * V, x01);
            if (t.ok10) Vec16<T>::load(x + (i00 + W) * x_ld + cv * V, x10);
            if (t.ok11) Vec16<T>::load(x + (i00 + W + 1) * x_ld + cv * V, x11);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float gj = gcol[j];
                const float val = x00[j] * t.w00 + x01[j] * t.w01 + x10[j] * t.w10 + x11[j] * t.w11;
                s_m = fmaf(gj, val, s_m);
                s_y = fmaf(gj, (1.f - t.lw) * (x10[j] - x00[j]) + t.lw * (x11[j] - x01[j]), s_y);
                s_x = fmaf(gj, (1.f - t.lh) * (x01[j] - x00[j]) + t.lh * (x11[j] - x10[j]), s_x);
                const float gm = gj * m;
                const int c = cv * V + j;
                if (t.w00 != 0.f) atomicAdd(dx + i00 * Ci + c, gm * t.w00);
                if (t.w01 != 0.f) atomicAdd(dx + (i00 + 1) * Ci + c, gm * t.w01);
                if (t.w10 != 0.f) atomicAdd(dx + (i00 + W) * Ci + c, gm * t.w10);
                if (t.w11 != 0.f) atomicAdd(dx + (i00 + W + 1) * Ci + c, gm * t.w11);
            }
        }
        for (int ofs = GS >> 1; ofs > 0; ofs >>= 1) {
            s_m += __shfl_xor(s_m, ofs, 64);
            s_y += __shfl_xor(s_y, ofs, 64);
            s_x += __shfl_xor(s_x, ofs, 64);
        }
        if (lg == 0) {
            float* d = dom + p * om_ld;
            d[2 * k] = s_y * m;
            d[2 * k + 1] = s_x * m;
            d[18 + k] = s_m * m * (1.f - m);
        }
    }
}

extern "C" int cn_dcn_im2col(const void* x, const float* om, void* col, int N, int H, int W, int Ci, int x_ld, int om_ld,
                             int dtype, void* stream) {
    CN_CHECK_ARG(x && om && col && N > 0 && H > 0 && W > 0 && Ci > 0, "cn_dcn_im2col: bad args");
    const int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(Ci % V == 0 && x_ld % V == 0 && om_ld >= 27, "cn_dcn_im2col: Ci/x_ld must be multiples of %d, om_ld >= 27", V);
    int64_t total = (int64_t)N * H * W * 9 * (Ci / V);
    int64_t g = (total + 255) / 256;
    int grid = (int)(g > 65536 ? 65536 : g);
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(dcn_im2col_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)x, om, (T*)col, N, H, W, Ci, x_ld, om_ld));
    CN_LAUNCH_CHECK("cn_dcn_im2col");
    return CN_OK;
}

extern "C" int cn_dcn_col2im(const void* dcol, const void* x, const float* om, float* dx_f32, float* dom, int N, int H, int W,
                             int Ci, int x_ld, int om_ld, int dtype, void* stream) {
    CN_CHECK_ARG(dcol && x && om && dx_f32 && dom && N > 0 && H > 0 && W > 0 && Ci > 0, "cn_dcn_col2im: bad args");
    const int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(Ci % V == 0 && x_ld % V == 0 && om_ld >= 27, "cn_dcn_col2im: Ci/x_ld must be multiples of %d, om_ld >= 27", V);
    int CV = Ci / V, GS = 1;
    while (GS * 2 <= CV && GS < 64) GS *= 2;
    int64_t ngroups = (int64_t)N * H * W * 9;
    int gpb = 256 / GS;
    int64_t g = (ngroups + gpb - 1) / gpb;
    int grid = (int)(g > 65536 ? 65536 : g);
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(dcn_col2im_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)dcol, (const T*)x, om, dx_f32, dom, N, H, W, Ci,
                                                   x_ld, om_ld, GS));
    CN_LAUNCH_CHECK("cn_dcn_col2im");
    return CN_OK;
}
