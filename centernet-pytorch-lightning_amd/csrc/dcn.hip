// DCNv2 (modulated deformable 3x3 conv, stride 1, pad 1, dilation 1, 1 deformable group) — the sampling half.
// The contraction half runs on the implicit-GEMM engine as a 1x1 conv over the sampled columns
//   col[p][k*Ci + c] = sigmoid(mask_logit[p,k]) * bilinear(x[n,:,:,c], h-1+i+dy[p,k], w-1+j+dx[p,k]),  k = 3i+j
// NHWC makes each bilinear corner one contiguous channel vector, so both kernels are plain 16-byte
// gathers (HBM/L2-bound).  Offsets are interleaved (dy,dx) per tap; mask logits follow.
#include "dcn_common.h"

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

// Backward of the sampling: given dcol, produce
//   dx  [P][Ci]  fp32 : dx[q,c] = sum_{p,k} mask[p,k] * hat(py[p,k]-qy) * hat(px[p,k]-qx) * dcol[p,k,c],  hat(t)=max(0,1-|t|)
//   dom [P][27]  fp32 : d/d(offset y, offset x) and d/d(mask logit) per tap
// A scatter needs up to 36 fp32 atomics per (pixel, channel) (79 % of the first train step profile; LDS atomics were no
// better).  dx is therefore computed as a GATHER, atomic-free and written once:
//   * gather kernel: one wave owns NQ consecutive destination pixels.  Test phase: its 64 lanes scan the (2R+1) x
//     (NQ+2R) x 9 samples around them (2 floats each from the fp32 offset map), and push the samples with a non-zero
//     bilinear "hat" weight into per-destination hit lists in LDS.  Channel phase: GS lanes per destination walk its hit
//     list (~36 entries) doing 16-byte dcol loads + FMAs in registers.
//   * source kernel: per (pixel, tap) lane group: offset / mask gradients, plus the rare samples displaced by more than
//     R pixels, which the gather window cannot see, via global atomics into dx_far.   dx = dx_tile + dx_far.
#define COL2IM_R 3
#define COL2IM_MAXHITS 96

template <typename T>
__global__ __launch_bounds__(256) void dcn_col2im_gather_kernel(const T* __restrict__ dcol, const float* __restrict__ om,
                                                                float* __restrict__ dx_tile, float* __restrict__ dx_far,
                                                                int N, int H, int W, int Ci, int om_ld, int GS, int ngroups_x) {
    constexpr int V = Vec16<T>::N;
    constexpr int R = COL2IM_R;
    __shared__ int hit_idx[4][8][COL2IM_MAXHITS];
    __shared__ float hit_w[4][8][COL2IM_MAXHITS];
    __shared__ int hit_cnt[4][8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int NQ = 64 / GS;                 // destination pixels per wave (<= 8)
    const int SW = NQ + 2 * R;
    const int ncand = (2 * R + 1) * SW * 9;
    const int c0 = blockIdx.y * GS * V;     // channel chunk of this workgroup
    const int ql = lane / GS, lg = lane % GS;
    const bool ch_ok = c0 + lg * V < Ci;
    const int64_t total = (int64_t)N * H * ngroups_x;
    const int64_t iters = (total + (int64_t)gridDim.x * 4 - 1) / ((int64_t)gridDim.x * 4);
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t G = ((int64_t)it * gridDim.x + blockIdx.x) * 4 + wv;
        const bool valid = G < total;
        const int xg = (int)(G % ngroups_x);
        const int64_t t2 = G / ngroups_x;
        const int qy = (int)(t2 % H), n = (int)(t2 / H);
        const int qx0 = xg * NQ;
        const int64_t img = (int64_t)n * H * W;
        if (lane < 8) hit_cnt[wv][lane] = 0;
        __syncthreads();
        if (valid) {
            for (int c = lane; c < ncand; c += 64) {
                const int k = c % 9, s = c / 9;
                const int sy = qy - R + s / SW, sx = qx0 - R + s % SW;
                if ((unsigned)sy >= (unsigned)H || (unsigned)sx >= (unsigned)W) continue;
                const int64_t p = img + (int64_t)sy * W + sx;
                const float* o = om + p * om_ld;
                const float py = (float)(sy - 1 + k / 3) + o[2 * k];
                const float wy = 1.f - fabsf(py - (float)qy);
                if (!(wy > 0.f)) continue;
                const float px = (float)(sx - 1 + k % 3) + o[2 * k + 1];
                const int x0 = (int)floorf(px);
                float m = -1.f;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int qx = x0 + e, q = qx - qx0;
                    if (q < 0 || q >= NQ || qx >= W || qx < 0) continue;
                    if (qx - sx > R || sx - qx > R) continue;          // beyond the window: the source kernel's far path owns it
                    const float wx = 1.f - fabsf(px - (float)qx);
                    if (!(wx > 0.f)) continue;
                    if (m < 0.f) m = sigmoidf_(o[18 + k]);
                    const float wm = wy * wx * m;
                    const int slot = atomicAdd(&hit_cnt[wv][q], 1);
                    if (slot < COL2IM_MAXHITS) {
                        hit_idx[wv][q][slot] = (int)((p - img) * 9 + k);
                        hit_w[wv][q][slot] = wm;
                    } else {                                             // pathological pile-up: this lane adds the whole chunk itself
                        const T* src = dcol + (p * 9 + k) * Ci;
                        float* dst = dx_far + (img + (int64_t)qy * W + qx) * Ci;
                        for (int cc = c0; cc < c0 + GS * V && cc < Ci; ++cc) atomicAdd(dst + cc, Elem<T>::ld(src + cc) * wm);
                    }
                }
            }
        }
        __syncthreads();
        if (valid && ql < NQ) {
            const int qx = qx0 + ql;
            if (qx < W && ch_ok) {
                int nh = hit_cnt[wv][ql];
                nh = nh < COL2IM_MAXHITS ? nh : COL2IM_MAXHITS;
                float acc[V];
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] = 0.f;
                const T* base = dcol + img * 9 * Ci + c0 + lg * V;
                for (int i = 0; i < nh; ++i) {
                    const float wm = hit_w[wv][ql][i];
                    float v[V];
                    Vec16<T>::load(base + (int64_t)hit_idx[wv][ql][i] * Ci, v);
#pragma unroll
                    for (int j = 0; j < V; ++j) acc[j] = fmaf(v[j], wm, acc[j]);
                }
                float* dst = dx_tile + (img + (int64_t)qy * W + qx) * Ci + c0 + lg * V;
#pragma unroll
                for (int j = 0; j < V; j += 4) *reinterpret_cast<float4*>(dst + j) = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
            }
        }
        __syncthreads();
    }
}

// One lane group (GS lanes, power of two <= 64) per (source pixel, tap); lanes stride over the channel vectors.
template <typename T>
__global__ __launch_bounds__(256) void dcn_col2im_source_kernel(const T* __restrict__ dcol, const T* __restrict__ x,
                                                                const float* __restrict__ om, float* __restrict__ dx_far,
                                                                float* __restrict__ dom, int N, int H, int W, int Ci, int x_ld,
                                                                int om_ld, int GS) {
    constexpr int V = Vec16<T>::N;
    constexpr int R = COL2IM_R;
    const int CV = Ci / V;
    const int64_t ngroups = (int64_t)N * H * W * 9;
    const int gpb = 256 / GS;
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
        const int dh0 = t.h0 - h, dw0 = t.w0 - w;
        const bool far_h0 = dh0 > R || dh0 < -R, far_h1 = dh0 + 1 > R || dh0 + 1 < -R;
        const bool far_w0 = dw0 > R || dw0 < -R, far_w1 = dw0 + 1 > R || dw0 + 1 < -R;
        const bool any_far = far_h0 || far_h1 || far_w0 || far_w1;
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
            }
            if (any_far) {   // samples the gather window cannot see
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float gm = gcol[j] * m;
                    const int c = cv * V + j;
                    if (t.w00 != 0.f && (far_h0 || far_w0)) atomicAdd(dx_far + i00 * Ci + c, gm * t.w00);
                    if (t.w01 != 0.f && (far_h0 || far_w1)) atomicAdd(dx_far + (i00 + 1) * Ci + c, gm * t.w01);
                    if (t.w10 != 0.f && (far_h1 || far_w0)) atomicAdd(dx_far + (i00 + W) * Ci + c, gm * t.w10);
                    if (t.w11 != 0.f && (far_h1 || far_w1)) atomicAdd(dx_far + (i00 + W + 1) * Ci + c, gm * t.w11);
                }
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

extern "C" int cn_dcn_col2im(const void* dcol, const void* x, const float* om, float* dx_tile, float* dx_far, float* dom,
                             int N, int H, int W, int Ci, int x_ld, int om_ld, int dtype, void* stream) {
    CN_CHECK_ARG(dcol && x && om && dx_tile && dx_far && dom && N > 0 && H > 0 && W > 0 && Ci > 0, "cn_dcn_col2im: bad args");
    const int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(Ci % V == 0 && x_ld % V == 0 && om_ld >= 27, "cn_dcn_col2im: Ci/x_ld must be multiples of %d, om_ld >= 27", V);
    hipStream_t st = (hipStream_t)stream;
    const int CV = Ci / V;
    {   // source kernel: dom + far samples
        int GS = 1;
        while (GS * 2 <= CV && GS < 64) GS *= 2;
        const int64_t ngroups = (int64_t)N * H * W * 9;
        const int gpb = 256 / GS;
        int64_t g = (ngroups + gpb - 1) / gpb;
        const int grid = (int)(g > 65536 ? 65536 : g);
        CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(dcn_col2im_source_kernel<T>, dim3(grid), dim3(256), 0, st, (const T*)dcol,
                                                       (const T*)x, om, dx_far, dom, N, H, W, Ci, x_ld, om_ld, GS));
        CN_LAUNCH_CHECK("cn_dcn_col2im(source)");
    }
    {   // gather kernel: dx_tile
        int GS = 8;                                   // lanes per destination pixel: 8..64, power of two, covers CV when possible
        while (GS < CV && GS < 64) GS *= 2;
        const int chunks = (CV + GS - 1) / GS;
        const int NQ = 64 / GS;
        const int ngx = (W + NQ - 1) / NQ;
        const int64_t total = (int64_t)N * H * ngx;
        int64_t g = (total + 3) / 4;
        const int grid = (int)(g > 16384 ? 16384 : g);
        CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(dcn_col2im_gather_kernel<T>, dim3(grid, chunks), dim3(256), 0, st,
                                                       (const T*)dcol, om, dx_tile, dx_far, N, H, W, Ci, om_ld, GS, ngx));
        CN_LAUNCH_CHECK("cn_dcn_col2im(gather)");
    }
    return CN_OK;
}
