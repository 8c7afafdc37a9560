// Test-time augmentation and detection post-processing on the device (soft-NMS of f-4).
// Reference: CenterNetDetection.test_step / test_step_end (centernet_detection.py:132-225) and utils/nms.py:5-107.  There,
// every image goes through torchvision pad/normalize/flip on the host side of the step, `.cpu()` per image after decode, a
// Python loop over 80 classes and numba soft-NMS.  Here: one launch prepares the padded / normalised / mirrored batch, one
// launch merges the mirrored head maps, and one launch per batch maps the decoded boxes back to image coordinates, groups
// them by class, runs soft-NMS (multi-scale only) and applies the max-per-image cut — the host receives one tensor per batch.
#include "common.h"
#include <math.h>

// out[b] = pad(normalize(img[b])) ; out[B + b] = hflip(out[b]) when flip.  The reference pads with zeros BEFORE normalising
// (centernet_detection.py:146-151), so the border holds (0 - mean) / std.
// RESIZE: the image is first resized to (H, W) from (SH, SW) — bilinear, half-pixel centres, no antialias, edge clamp: what
// VF.resize does to a tensor (centernet_detection.py:141; ATen upsample_bilinear2d, align_corners=False) — inside the same launch
template <bool RESIZE>
__global__ __launch_bounds__(256) void tta_prepare_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int SH, int SW,
                                                          int H, int W, int pad_x, int pad_y, float m0, float m1, float m2, float s0,
                                                          float s1, float s2, int flip) {
    const int PH = H + 2 * pad_y, PW = W + 2 * pad_x;
    const int64_t total = (int64_t)B * 3 * PH * PW;
    const float rh = (float)SH / (float)H, rw = (float)SW / (float)W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % PW);
        int64_t r = i / PW;
        const int y = (int)(r % PH);
        r /= PH;
        const int c = (int)(r % 3), b = (int)(r / 3);
        const int iy = y - pad_y, ix = x - pad_x;
        const bool in = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const float* __restrict__ src = img + ((int64_t)b * 3 + c) * SH * SW;
        float v = 0.f;
        if (in) {
            if (RESIZE) {
                const float fy = fmaxf(rh * ((float)iy + 0.5f) - 0.5f, 0.f), fx = fmaxf(rw * ((float)ix + 0.5f) - 0.5f, 0.f);
                const int y0 = (int)fy, x0 = (int)fx;
                const int y1 = y0 + (y0 < SH - 1), x1 = x0 + (x0 < SW - 1);
                const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
                v = hy * (hx * src[(int64_t)y0 * SW + x0] + lx * src[(int64_t)y0 * SW + x1])
                  + ly * (hx * src[(int64_t)y1 * SW + x0] + lx * src[(int64_t)y1 * SW + x1]);
            } else {
                v = src[(int64_t)iy * SW + ix];
            }
        }
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        const float o = (v - mean) / sd;
        out[i] = o;
        if (flip) out[((((int64_t)(B + b)) * 3 + c) * PH + y) * PW + (PW - 1 - x)] = o;
    }
}

// out[b] = (x[b] + hflip(x[B + b])) / 2 on NCHW fp32 head maps (centernet_detection.py:167-171)
__global__ __launch_bounds__(256) void flip_merge_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t half, int W) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        out[i] = (x[i] + x[half + i - w + (W - 1 - w)]) / 2.f;
    }
}

#define PP_MAXCAND 1024
#define PP_MAXCLS 256

// utils/nms.py:5-107 verbatim on one class segment (sequential by nature: the array is reordered as it goes).  numba
// promotes `float32 + 1` to float64, so the overlap arithmetic runs in double and the decayed score is stored as float32.
__device__ static int soft_nms_segment(float* bx, int N, double sigma, double Nt, double threshold, int method) {
    const int N0 = N;
    for (int i = 0; i < N0; ++i) {
        if (i >= N) break;                                   // (the reference's remaining iterations are no-ops)
        float maxscore = bx[i * 5 + 4];
        int maxpos = i;
        for (int pos = i + 1; pos < N; ++pos)
            if (maxscore < bx[pos * 5 + 4]) { maxscore = bx[pos * 5 + 4]; maxpos = pos; }
        for (int q = 0; q < 5; ++q) { const float t = bx[i * 5 + q]; bx[i * 5 + q] = bx[maxpos * 5 + q]; bx[maxpos * 5 + q] = t; }
        const double tx1 = bx[i * 5], ty1 = bx[i * 5 + 1], tx2 = bx[i * 5 + 2], ty2 = bx[i * 5 + 3];
        int pos = i + 1;
        while (pos < N) {
            const double x1 = bx[pos * 5], y1 = bx[pos * 5 + 1], x2 = bx[pos * 5 + 2], y2 = bx[pos * 5 + 3];
            const double area = (x2 - x1 + 1) * (y2 - y1 + 1);
            const double iw = fmin(tx2, x2) - fmax(tx1, x1) + 1;
            if (iw > 0) {
                const double ih = fmin(ty2, y2) - fmax(ty1, y1) + 1;
                if (ih > 0) {
                    const double ua = (tx2 - tx1 + 1) * (ty2 - ty1 + 1) + area - iw * ih;
                    const double ov = iw * ih / ua;
                    double weight;
                    if (method == 1) weight = ov > Nt ? 1 - ov : 1;
                    else if (method == 2) weight = exp(-(ov * ov) / sigma);
                    else weight = ov > Nt ? 0 : 1;
                    bx[pos * 5 + 4] = (float)(weight * (double)bx[pos * 5 + 4]);
                    if ((double)bx[pos * 5 + 4] < threshold) {
                        for (int q = 0; q < 5; ++q) bx[pos * 5 + q] = bx[(N - 1) * 5 + q];
                        --N;
                        --pos;
                    }
                }
            }
            ++pos;
        }
    }
    return N;
}

// dets [S, B, K, 6] (x1, y1, x2, y2, score, class; output-map coordinates of scale s), meta [S, 4] = pad_x, pad_y, scale_x,
// scale_y.  One workgroup per image: transform (x down, - pad, / scale in fp32, in that order), group by class (scale-major,
// rank order inside a scale == np.concatenate of the per-scale class arrays), soft-NMS per class when S > 1, then keep the
// scores >= the max_per-th largest.  rows [B, S*K, 6] come out class-ascending; counts [B].
__global__ __launch_bounds__(256) void ctdet_merge_kernel(const float* __restrict__ dets, const float* __restrict__ meta,
                                                          float* __restrict__ rows, int* __restrict__ counts, int S, int B, int K,
                                                          int C, float down, int max_per, int nms_method, float nms_nt,
                                                          float nms_sigma, float nms_thresh) {
    __shared__ float cand[PP_MAXCAND * 6];
    __shared__ float seg[PP_MAXCAND * 5];
    __shared__ int cnt[PP_MAXCLS], off[PP_MAXCLS + 1], keepn[PP_MAXCLS];
    __shared__ int total_s;
    const int b = blockIdx.x, tid = threadIdx.x, n = S * K;
    for (int e = tid; e < n; e += blockDim.x) {
        const int s = e / K, r = e % K;
        const float* d = dets + (((int64_t)s * B + b) * K + r) * 6;
        const float px = meta[s * 4], py = meta[s * 4 + 1], sx = meta[s * 4 + 2], sy = meta[s * 4 + 3];
        cand[e * 6 + 0] = (d[0] * down - px) / sx;
        cand[e * 6 + 1] = (d[1] * down - py) / sy;
        cand[e * 6 + 2] = (d[2] * down - px) / sx;
        cand[e * 6 + 3] = (d[3] * down - py) / sy;
        cand[e * 6 + 4] = d[4];
        cand[e * 6 + 5] = d[5];
    }
    __syncthreads();
    for (int c = tid; c < C; c += blockDim.x) {
        int m = 0;
        for (int e = 0; e < n; ++e) m += (cand[e * 6 + 5] == (float)c);
        cnt[c] = m;
    }
    __syncthreads();
    if (tid == 0) {
        int a = 0;
        for (int c = 0; c < C; ++c) { off[c] = a; a += cnt[c]; }
        off[C] = a;
    }
    __syncthreads();
    for (int c = tid; c < C; c += blockDim.x) {
        float* bx = seg + off[c] * 5;
        int m = 0;
        for (int e = 0; e < n; ++e)
            if (cand[e * 6 + 5] == (float)c) {
                for (int q = 0; q < 5; ++q) bx[m * 5 + q] = cand[e * 6 + q];
                ++m;
            }
        keepn[c] = (S > 1 && m > 0) ? soft_nms_segment(bx, m, (double)nms_sigma, (double)nms_nt, (double)nms_thresh, nms_method) : m;
    }
    __syncthreads();
    if (tid == 0) {                                          // compact the kept prefix of every class segment into cand
        int a = 0;
        for (int c = 0; c < C; ++c) {
            for (int i = 0; i < keepn[c]; ++i, ++a) {
                for (int q = 0; q < 5; ++q) cand[a * 6 + q] = seg[(off[c] + i) * 5 + q];
                cand[a * 6 + 5] = (float)c;
            }
        }
        total_s = a;
    }
    __syncthreads();
    const int total = total_s;
    // keep score >= the max_per-th largest  <=>  fewer than max_per scores are strictly greater (centernet_detection.py:217-223)
    for (int e = tid; e < total; e += blockDim.x) {
        int g = 0;
        if (total > max_per) {
            const float s = cand[e * 6 + 4];
            for (int j = 0; j < total; ++j) g += (cand[j * 6 + 4] > s);
        }
        seg[e] = (g < max_per) ? 1.f : 0.f;
    }
    __syncthreads();
    if (tid == 0) {
        int a = 0;
        float* out = rows + (int64_t)b * n * 6;
        for (int e = 0; e < total; ++e)
            if (seg[e] != 0.f) {
                for (int q = 0; q < 6; ++q) out[a * 6 + q] = cand[e * 6 + q];
                ++a;
            }
        counts[b] = a;
        for (int e = a * 6; e < n * 6; ++e) out[e] = 0.f;
    }
}

// out[b, c] = (x[b, c] + sign[c] * hflip(x[B + b, perm[c]])) / 2 — the pose-aware merges of centernet_multi_pose.py:200-211:
// keypoint regression maps swap left/right joints (flip_idx) and negate the x component, keypoint heat maps only swap.
__global__ __launch_bounds__(256) void flip_merge_perm_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                              const int* __restrict__ perm, const float* __restrict__ sign, int B,
                                                              int C, int H, int W) {
    const int64_t half = (int64_t)B * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        int64_t r = i / W;
        const int h = (int)(r % H);
        r /= H;
        const int c = (int)(r % C), b = (int)(r / C);
        const float f = x[((((int64_t)(B + b)) * C + perm[c]) * H + h) * W + (W - 1 - w)] * sign[c];
        out[i] = (x[i] + f) / 2.f;
    }
}

#define PP_POSE_MAXROWS 1024
// utils/nms.py:109-206 (soft_nms_39) on LDS-resident box columns; the keypoint columns 5..38 travel with the boxes through
// the index list `pay`, columns 39.. (class, keypoint scores) are NOT moved by the reference and stay with their position.
__device__ static int soft_nms39_lds(float* bx, int* pay, int N, double sigma, double Nt, double threshold, int method) {
    const int N0 = N;
    for (int i = 0; i < N0 && i < N; ++i) {
        float maxscore = bx[i * 5 + 4];
        int maxpos = i;
        for (int pos = i + 1; pos < N; ++pos)
            if (maxscore < bx[pos * 5 + 4]) { maxscore = bx[pos * 5 + 4]; maxpos = pos; }
        for (int q = 0; q < 5; ++q) { const float t = bx[i * 5 + q]; bx[i * 5 + q] = bx[maxpos * 5 + q]; bx[maxpos * 5 + q] = t; }
        { const int t = pay[i]; pay[i] = pay[maxpos]; pay[maxpos] = t; }
        const double tx1 = bx[i * 5], ty1 = bx[i * 5 + 1], tx2 = bx[i * 5 + 2], ty2 = bx[i * 5 + 3];
        int pos = i + 1;
        while (pos < N) {
            const double x1 = bx[pos * 5], y1 = bx[pos * 5 + 1], x2 = bx[pos * 5 + 2], y2 = bx[pos * 5 + 3];
            const double area = (x2 - x1 + 1) * (y2 - y1 + 1);
            const double iw = fmin(tx2, x2) - fmax(tx1, x1) + 1;
            if (iw > 0) {
                const double ih = fmin(ty2, y2) - fmax(ty1, y1) + 1;
                if (ih > 0) {
                    const double ua = (tx2 - tx1 + 1) * (ty2 - ty1 + 1) + area - iw * ih;
                    const double ov = iw * ih / ua;
                    double weight;
                    if (method == 1) weight = ov > Nt ? 1 - ov : 1;
                    else if (method == 2) weight = exp(-(ov * ov) / sigma);
                    else weight = ov > Nt ? 0 : 1;
                    bx[pos * 5 + 4] = (float)(weight * (double)bx[pos * 5 + 4]);
                    if ((double)bx[pos * 5 + 4] < threshold) {
                        for (int q = 0; q < 5; ++q) bx[pos * 5 + q] = bx[(N - 1) * 5 + q];      // copied (:190-194) ...
                        { const int t = pay[pos]; pay[pos] = pay[N - 1]; pay[N - 1] = t; }       // ... swapped (:195-198)
                        --N;
                        --pos;
                    }
                }
            }
            ++pos;
        }
    }
    return N;
}

// dets [S, B, K, D] rows of multi_pose_decode (D = 57: box 4, score, 34 keypoint coords, class, 17 keypoint scores); meta as
// above.  One workgroup per image: boxes and keypoints to image coordinates, scales concatenated, soft_nms_39 when S > 1,
// then scores >= the max_per-th largest.  rows [B, S*K, D] zero padded, counts [B].  (centernet_multi_pose.py:213-264)
__global__ __launch_bounds__(256) void pose_merge_kernel(const float* __restrict__ dets, const float* __restrict__ meta,
                                                         float* __restrict__ rows, int* __restrict__ counts, int S, int B, int K, int D,
                                                         float down, int max_per, int nms_method, float nms_nt, float nms_sigma,
                                                         float nms_thresh) {
    __shared__ float bx[PP_POSE_MAXROWS * 5];
    __shared__ int pay[PP_POSE_MAXROWS], dst[PP_POSE_MAXROWS];
    __shared__ int n_s, kept_s;
    const int b = blockIdx.x, tid = threadIdx.x, n = S * K;
    auto src_row = [&](int e) { return dets + (((int64_t)(e / K) * B + b) * K + (e % K)) * D; };
    for (int e = tid; e < n; e += blockDim.x) {
        const int s = e / K;
        const float* d = src_row(e);
        const float px = meta[s * 4], py = meta[s * 4 + 1], sx = meta[s * 4 + 2], sy = meta[s * 4 + 3];
        bx[e * 5 + 0] = (d[0] * down - px) / sx;
        bx[e * 5 + 1] = (d[1] * down - py) / sy;
        bx[e * 5 + 2] = (d[2] * down - px) / sx;
        bx[e * 5 + 3] = (d[3] * down - py) / sy;
        bx[e * 5 + 4] = d[4];
        pay[e] = e;
    }
    __syncthreads();
    if (tid == 0) n_s = S > 1 ? soft_nms39_lds(bx, pay, n, (double)nms_sigma, (double)nms_nt, (double)nms_thresh, nms_method) : n;
    __syncthreads();
    const int total = n_s;
    for (int e = tid; e < total; e += blockDim.x) {
        int g = 0;
        if (total > max_per) {
            const float sc = bx[e * 5 + 4];
            for (int j = 0; j < total; ++j) g += (bx[j * 5 + 4] > sc);
        }
        dst[e] = g < max_per ? 1 : 0;
    }
    __syncthreads();
    if (tid == 0) {
        int a = 0;
        for (int e = 0; e < total; ++e) dst[e] = dst[e] ? a++ : -1;
        kept_s = a;
        counts[b] = a;
    }
    __syncthreads();
    const int kept = kept_s;
    float* out = rows + (int64_t)b * n * D;
    for (int i = tid; i < total * D; i += blockDim.x) {
        const int e = i / D, q = i % D;
        if (dst[e] < 0) continue;
        float v;
        if (q < 5) v = bx[e * 5 + q];
        else if (q < 39) {                                   // keypoints travelled with the box: take them from row pay[e]
            const int o = pay[e], s = o / K;
            const float pad = (q - 5) & 1 ? meta[s * 4 + 1] : meta[s * 4], sc = (q - 5) & 1 ? meta[s * 4 + 3] : meta[s * 4 + 2];
            v = (src_row(o)[q] * down - pad) / sc;
        } else v = src_row(e)[q];                            // class / keypoint scores stay at their position (reference quirk)
        out[(int64_t)dst[e] * D + q] = v;
    }
    for (int i = kept * D + tid; i < n * D; i += blockDim.x) out[i] = 0.f;
}

static int pp_grid(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

extern "C" int cn_tta_prepare_scaled(const float* img, float* out, int B, int H, int W, int new_h, int new_w, int pad_x, int pad_y,
                                     float mean0, float mean1, float mean2, float std0, float std1, float std2, int flip, void* stream) {
    CN_CHECK_ARG(img && out && B > 0 && H > 0 && W > 0 && new_h > 0 && new_w > 0 && pad_x >= 0 && pad_y >= 0, "cn_tta_prepare_scaled: bad args");
    const int64_t total = (int64_t)B * 3 * (new_h + 2 * pad_y) * (new_w + 2 * pad_x);
    if (new_h == H && new_w == W)
        hipLaunchKernelGGL(tta_prepare_kernel<false>, dim3(pp_grid(total)), dim3(256), 0, (hipStream_t)stream, img, out, B, H, W, H, W,
                           pad_x, pad_y, mean0, mean1, mean2, std0, std1, std2, flip);
    else
        hipLaunchKernelGGL(tta_prepare_kernel<true>, dim3(pp_grid(total)), dim3(256), 0, (hipStream_t)stream, img, out, B, H, W, new_h,
                           new_w, pad_x, pad_y, mean0, mean1, mean2, std0, std1, std2, flip);
    CN_LAUNCH_CHECK("cn_tta_prepare_scaled");
    return CN_OK;
}

extern "C" int cn_tta_prepare(const float* img, float* out, int B, int H, int W, int pad_x, int pad_y, float mean0, float mean1,
                              float mean2, float std0, float std1, float std2, int flip, void* stream) {
    return cn_tta_prepare_scaled(img, out, B, H, W, H, W, pad_x, pad_y, mean0, mean1, mean2, std0, std1, std2, flip, stream);
}

extern "C" int cn_flip_merge(const float* x, float* out, int B, int C, int H, int W, void* stream) {
    CN_CHECK_ARG(x && out && B > 0 && C > 0 && H > 0 && W > 0, "cn_flip_merge: bad args");
    const int64_t half = (int64_t)B * C * H * W;
    hipLaunchKernelGGL(flip_merge_kernel, dim3(pp_grid(half)), dim3(256), 0, (hipStream_t)stream, x, out, half, W);
    CN_LAUNCH_CHECK("cn_flip_merge");
    return CN_OK;
}

extern "C" int cn_ctdet_merge(const float* dets, const float* meta, float* rows, int* counts, int S, int B, int K, int C,
                              int down_ratio, int max_per_image, int nms_method, float nms_nt, float nms_sigma, float nms_threshold,
                              void* stream) {
    CN_CHECK_ARG(dets && meta && rows && counts && S > 0 && B > 0 && K > 0 && C > 0 && max_per_image > 0, "cn_ctdet_merge: bad args");
    if (S * K > PP_MAXCAND || C > PP_MAXCLS) CN_UNSUPPORTED("cn_ctdet_merge: S*K <= %d and C <= %d (got %d, %d)", PP_MAXCAND, PP_MAXCLS, S * K, C);
    hipLaunchKernelGGL(ctdet_merge_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dets, meta, rows, counts, S, B, K, C,
                       (float)down_ratio, max_per_image, nms_method, nms_nt, nms_sigma, nms_threshold);
    CN_LAUNCH_CHECK("cn_ctdet_merge");
    return CN_OK;
}

extern "C" int cn_flip_merge_perm(const float* x, float* out, const int* perm, const float* sign, int B, int C, int H, int W,
                                  void* stream) {
    CN_CHECK_ARG(x && out && perm && sign && B > 0 && C > 0 && H > 0 && W > 0, "cn_flip_merge_perm: bad args");
    hipLaunchKernelGGL(flip_merge_perm_kernel, dim3(pp_grid((int64_t)B * C * H * W)), dim3(256), 0, (hipStream_t)stream, x, out, perm,
                       sign, B, C, H, W);
    CN_LAUNCH_CHECK("cn_flip_merge_perm");
    return CN_OK;
}

extern "C" int cn_pose_merge(const float* dets, const float* meta, float* rows, int* counts, int S, int B, int K, int D,
                             int down_ratio, int max_per_image, int nms_method, float nms_nt, float nms_sigma, float nms_threshold,
                             void* stream) {
    CN_CHECK_ARG(dets && meta && rows && counts && S > 0 && B > 0 && K > 0 && D >= 39 && max_per_image > 0, "cn_pose_merge: bad args");
    if (S * K > PP_POSE_MAXROWS) CN_UNSUPPORTED("cn_pose_merge: S*K <= %d (got %d)", PP_POSE_MAXROWS, S * K);
    hipLaunchKernelGGL(pose_merge_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dets, meta, rows, counts, S, B, K, D,
                       (float)down_ratio, max_per_image, nms_method, nms_nt, nms_sigma, nms_threshold);
    CN_LAUNCH_CHECK("cn_pose_merge");
    return CN_OK;
}
