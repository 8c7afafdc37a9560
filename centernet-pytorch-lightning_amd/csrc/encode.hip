// Ground-truth encoding on the device: COCO boxes -> CenterNet ctdet targets, one launch per batch.
// Reference: CenterNet/sample/ctdet.py:39-90 (per-sample Python loop on the host) + utils/gaussian.py:6-58.
//   per object k: box / down_ratio, clipped to the map; if h > 0 and w > 0:
//     radius = max(0, int(gaussian_radius(ceil(h), ceil(w))));  ct = box centre (fp32), ct_int = trunc(ct)
//     heatmap[cls] = max(heatmap[cls], truncated gaussian(2r+1, sigma = (2r+1)/6) centred on ct_int)     (draw_umich_gaussian)
//     wh[k] = (w, h); indices[k] = ct_int.y * OW + ct_int.x; regression[k] = ct - ct_int; mask[k] = 1
// One workgroup per (object, image); the max-splat uses integer atomicMax on the bit pattern (all values are >= 0, for
// which the integer order equals the float order), so overlapping objects need no ordering and the result is deterministic.
#include "common.h"
#include <math.h>

__device__ static inline double gaussian_radius_d(double h, double w) {       // utils/gaussian.py:6-26, min_overlap = 0.7
    const double mo = 0.7;
    const double b1 = h + w, c1 = w * h * (1 - mo) / (1 + mo);
    const double r1 = (b1 + sqrt(b1 * b1 - 4 * c1)) / 2;
    const double b2 = 2 * (h + w), c2 = (1 - mo) * w * h;
    const double r2 = (b2 + sqrt(b2 * b2 - 16 * c2)) / 2;
    const double a3 = 4 * mo, b3 = -2 * mo * (h + w), c3 = (mo - 1) * w * h;
    const double r3 = (b3 + sqrt(b3 * b3 - 4 * a3 * c3)) / 2;
    return fmin(r1, fmin(r2, r3));
}

__global__ __launch_bounds__(256) void encode_ctdet_kernel(const float* __restrict__ boxes, const int* __restrict__ cls,
                                                           const int* __restrict__ nobj, float* __restrict__ heatmap,
                                                           unsigned char* __restrict__ mask, int64_t* __restrict__ indices,
                                                           float* __restrict__ wh, float* __restrict__ reg, int M, int C, int OH,
                                                           int OW, float down) {
    const int k = blockIdx.x, b = blockIdx.y;
    __shared__ int s_geo[4];                     // cx, cy, radius, valid
    if (threadIdx.x == 0) {
        const int64_t o = (int64_t)b * M + k;
        float w_ = 0.f, h_ = 0.f, rx = 0.f, ry = 0.f;
        int64_t ind = 0;
        int valid = 0, cx = 0, cy = 0, rad = 0;
        if (k < nobj[b]) {
            const float* bb = boxes + o * 4;
            float x0 = bb[0] / down, y0 = bb[1] / down, x1 = (bb[0] + bb[2]) / down, y1 = (bb[1] + bb[3]) / down;
            x0 = fminf(fmaxf(x0, 0.f), (float)(OW - 1)); x1 = fminf(fmaxf(x1, 0.f), (float)(OW - 1));
            y0 = fminf(fmaxf(y0, 0.f), (float)(OH - 1)); y1 = fminf(fmaxf(y1, 0.f), (float)(OH - 1));
            const float h = y1 - y0, w = x1 - x0;
            const int c = cls[o];
            if (h > 0.f && w > 0.f && c >= 0 && c < C) {
                const double r = gaussian_radius_d(ceil((double)h), ceil((double)w));
                rad = r > 0.0 ? (int)r : 0;
                const float ctx = (x0 + x1) / 2.f, cty = (y0 + y1) / 2.f;
                cx = (int)ctx; cy = (int)cty;
                w_ = w; h_ = h; rx = ctx - (float)cx; ry = cty - (float)cy;
                ind = (int64_t)cy * OW + cx;
                valid = 1;
            }
        }
        wh[o * 2] = w_; wh[o * 2 + 1] = h_;
        reg[o * 2] = rx; reg[o * 2 + 1] = ry;
        indices[o] = ind;
        mask[o] = (unsigned char)valid;
        s_geo[0] = cx; s_geo[1] = cy; s_geo[2] = rad; s_geo[3] = valid;
    }
    __syncthreads();
    if (!s_geo[3]) return;
    const int cx = s_geo[0], cy = s_geo[1], rad = s_geo[2];
    const int diam = 2 * rad + 1;
    const float sigma = (float)diam / 6.f;
    const float denom = 2.f * sigma * sigma;
    int* hm = reinterpret_cast<int*>(heatmap + ((int64_t)b * C + cls[(int64_t)b * M + k]) * OH * OW);
    for (int i = threadIdx.x; i < diam * diam; i += blockDim.x) {
        const int dy = i / diam - rad, dx = i % diam - rad;
        const int y = cy + dy, x = cx + dx;
        if ((unsigned)y >= (unsigned)OH || (unsigned)x >= (unsigned)OW) continue;
        float v = expf(-(float)(dx * dx + dy * dy) / denom);
        if (v < 1.1920929e-07f) v = 0.f;                     // gaussian2D: h[h < eps * h.max()] = 0 (max = 1 at the centre)
        atomicMax(hm + (int64_t)y * OW + x, __float_as_int(v));
    }
}

extern "C" int cn_encode_ctdet(const float* boxes, const int* cls, const int* nobj, float* heatmap, unsigned char* mask,
                               int64_t* indices, float* wh, float* reg, int B, int M, int C, int OH, int OW, int down_ratio,
                               void* stream) {
    CN_CHECK_ARG(boxes && cls && nobj && heatmap && mask && indices && wh && reg, "cn_encode_ctdet: null pointer");
    CN_CHECK_ARG(B > 0 && M > 0 && C > 0 && OH > 0 && OW > 0 && down_ratio > 0 && B <= 65535, "cn_encode_ctdet: bad dims");
    hipLaunchKernelGGL(encode_ctdet_kernel, dim3(M, B), dim3(256), 0, (hipStream_t)stream, boxes, cls, nobj, heatmap, mask, indices,
                       wh, reg, M, C, OH, OW, (float)down_ratio);
    CN_LAUNCH_CHECK("cn_encode_ctdet");
    return CN_OK;
}
