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
                                                           int OW, float down, int msra) {
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
    if (msra) {
        // draw_msra_gaussian(heatmap[cls], ct_int, radius) (sample/ctdet.py:54, 70; utils/gaussian.py:61-83) with the INTEGER radius
        // as sigma: a (6 sigma + 1)^2 patch centred on ct_int, dropped entirely when it touches the border (:68, the reference's
        // swapped w/h names: columns against shape[1], rows against shape[0]), no truncation of small values.  radius 0 is the
        // reference's 0/0: the centre pixel becomes NaN (np.maximum propagates it; NaN's bit pattern wins the integer max too).
        const int t = 3 * rad, size = 2 * t + 1;
        if (cx + t + 1 >= OW || cy + t + 1 >= OH || cx - t < 0 || cy - t < 0) return;
        const float den = 2.f * (float)(rad * rad);
        int* hmm = reinterpret_cast<int*>(heatmap + ((int64_t)b * C + cls[(int64_t)b * M + k]) * OH * OW);
        for (int i = threadIdx.x; i < size * size; i += blockDim.x) {
            const int dy = i / size - t, dx = i % size - t;
            const float v = rad == 0 ? __int_as_float(0x7fc00000) : expf(-(float)(dx * dx + dy * dy) / den);
            atomicMax(hmm + (int64_t)(cy + dy) * OW + cx + dx, __float_as_int(v));
        }
        return;
    }
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
                               int gaussian_type, void* stream) {
    CN_CHECK_ARG(boxes && cls && nobj && heatmap && mask && indices && wh && reg, "cn_encode_ctdet: null pointer");
    CN_CHECK_ARG(B > 0 && M > 0 && C > 0 && OH > 0 && OW > 0 && down_ratio > 0 && B <= 65535, "cn_encode_ctdet: bad dims");
    CN_CHECK_ARG(gaussian_type == 0 || gaussian_type == 1, "cn_encode_ctdet: gaussian_type %d (0 = umich, 1 = msra)", gaussian_type);
    hipLaunchKernelGGL(encode_ctdet_kernel, dim3(M, B), dim3(256), 0, (hipStream_t)stream, boxes, cls, nobj, heatmap, mask, indices,
                       wh, reg, M, C, OH, OW, (float)down_ratio, gaussian_type);
    CN_LAUNCH_CHECK("cn_encode_ctdet");
    return CN_OK;
}

// ---- multi_pose targets (sample/multi_pose.py:35-112) ----------------------------------------------------------------------
// One workgroup per (object k, joint j, image b).  Lane 0 redoes the object's box arithmetic (fp32, like the numpy original),
// writes keypoints / masks / offsets / indices; all lanes splat the joint's gaussian (draw_msra_gaussian, utils/gaussian.py:
// 61-83) with the FLOAT sigma = gaussian_radius(ceil h, ceil w) the reference passes:  t = 3 sigma, ul = trunc(c - t),
// br = trunc(c + t + 1), dropped entirely when the box touches the border, g(i, j) = exp(-((i - x0)^2 + (j - x0)^2) / (2 sigma^2))
// with x0 = floor(t + 0.5) relative to ul — the peak is one pixel up/left of the keypoint when frac(t) < 0.5, as in the reference.
// Every slot of the six outputs is written (zeros for empty slots) except the heat map, which the caller zero-fills.
__global__ __launch_bounds__(256) void encode_multi_pose_kernel(const float* __restrict__ boxes, const float* __restrict__ kps,
                                                                const int* __restrict__ nobj, float* __restrict__ hm_hp,
                                                                float* __restrict__ kp_out, unsigned char* __restrict__ kp_mask,
                                                                float* __restrict__ hp_off, int64_t* __restrict__ hp_ind,
                                                                unsigned char* __restrict__ hp_mask, int M, int J, int OH, int OW,
                                                                float down) {
    const int j = blockIdx.x % J, k = blockIdx.x / J, b = blockIdx.y;
    __shared__ int s_i[6];          // valid, ulx, uly, nx, ny
    __shared__ float s_f[2];        // x0, 2 sigma^2
    if (threadIdx.x == 0) {
        const int64_t o = (int64_t)b * M + k;
        float kx = 0.f, ky = 0.f, ox = 0.f, oy = 0.f;
        int64_t ind = 0;
        int valid = 0, draw = 0, ulx = 0, uly = 0, nx = 0, ny = 0;
        float x0 = 0.f, den = 1.f;
        if (k < nobj[b]) {
            const float* bb = boxes + o * 4;
            float bx0 = bb[0] / down, by0 = bb[1] / down, bx1 = (bb[0] + bb[2]) / down, by1 = (bb[1] + bb[3]) / down;
            bx0 = fminf(fmaxf(bx0, 0.f), (float)(OW - 1)); bx1 = fminf(fmaxf(bx1, 0.f), (float)(OW - 1));
            by0 = fminf(fmaxf(by0, 0.f), (float)(OH - 1)); by1 = fminf(fmaxf(by1, 0.f), (float)(OH - 1));
            const float h = by1 - by0, w = bx1 - bx0;
            const float* pt = kps + (o * J + j) * 3;
            if (h > 0.f && w > 0.f && pt[2] != 0.f) {
                const int ctx = (int)((bx0 + bx1) / 2.f), cty = (int)((by0 + by1) / 2.f);
                const float px = fminf(fmaxf(pt[0] / down, 0.f), (float)(OW - 1)), py = fminf(fmaxf(pt[1] / down, 0.f), (float)(OH - 1));
                const int ix = (int)px, iy = (int)py;
                kx = px - (float)ctx; ky = py - (float)cty;
                ox = px - (float)ix; oy = py - (float)iy;
                ind = (int64_t)iy * OW + ix;
                valid = 1;
                const double sigma = gaussian_radius_d(ceil((double)h), ceil((double)w));
                const double t = sigma * 3.0;
                const int ux = (int)((double)ix - t), uy = (int)((double)iy - t);
                const int rx = (int)((double)ix + t + 1.0), ry = (int)((double)iy + t + 1.0);
                if (!(rx >= OW || ry >= OH || ux < 0 || uy < 0)) {
                    draw = 1; ulx = ux; uly = uy; nx = rx - ux; ny = ry - uy;
                    x0 = (float)floor((2.0 * t + 1.0) / 2.0);
                    den = (float)(2.0 * sigma * sigma);
                }
            }
        }
        const int64_t oj = o * J + j;
        kp_out[oj * 2] = kx; kp_out[oj * 2 + 1] = ky;
        kp_mask[oj * 2] = kp_mask[oj * 2 + 1] = (unsigned char)valid;
        hp_off[oj * 2] = ox; hp_off[oj * 2 + 1] = oy;
        hp_ind[oj] = ind;
        hp_mask[oj] = (unsigned char)valid;
        s_i[0] = draw; s_i[1] = ulx; s_i[2] = uly; s_i[3] = nx; s_i[4] = ny;
        s_f[0] = x0; s_f[1] = den;
    }
    __syncthreads();
    if (!s_i[0]) return;
    const int ulx = s_i[1], uly = s_i[2], nx = s_i[3], ny = s_i[4];
    const float x0 = s_f[0], den = s_f[1];
    int* hm = reinterpret_cast<int*>(hm_hp + ((int64_t)b * J + j) * OH * OW);
    for (int i = threadIdx.x; i < nx * ny; i += blockDim.x) {
        const int gy = i / nx, gx = i % nx;
        const float dx = (float)gx - x0, dy = (float)gy - x0;
        const float v = expf(-(dx * dx + dy * dy) / den);
        atomicMax(hm + (int64_t)(uly + gy) * OW + ulx + gx, __float_as_int(v));
    }
}

extern "C" int cn_encode_multi_pose(const float* boxes, const float* keypoints, const int* nobj, float* heatmap_keypoints,
                                    float* kp_out, unsigned char* kp_mask, float* hp_offset, int64_t* hp_indices,
                                    unsigned char* hp_mask, int B, int M, int J, int OH, int OW, int down_ratio, void* stream) {
    CN_CHECK_ARG(boxes && keypoints && nobj && heatmap_keypoints && kp_out && kp_mask && hp_offset && hp_indices && hp_mask,
                 "cn_encode_multi_pose: null pointer");
    CN_CHECK_ARG(B > 0 && M > 0 && J > 0 && OH > 0 && OW > 0 && down_ratio > 0 && B <= 65535, "cn_encode_multi_pose: bad dims");
    hipLaunchKernelGGL(encode_multi_pose_kernel, dim3(M * J, B), dim3(256), 0, (hipStream_t)stream, boxes, keypoints, nobj,
                       heatmap_keypoints, kp_out, kp_mask, hp_offset, hp_indices, hp_mask, M, J, OH, OW, (float)down_ratio);
    CN_LAUNCH_CHECK("cn_encode_multi_pose");
    return CN_OK;
}
