// Losses of the CenterNet heads (utils/losses.py) on the public NCHW fp32 head maps.
// Focal loss: one streaming pass (16-byte loads), per-workgroup partials -> fp64 finalize on device, so there is
// no host sync on `num_pos == 0` (utils/losses.py:35).  Gather-L1: reads only the <=N indexed rows, never
// materialises the NHWC transpose the reference makes (utils/decode.py:59-63).
#include "common.h"

#define FOCAL_MAX_BLOCKS 2048

__global__ __launch_bounds__(256) void sigmoid_clamp_fwd_kernel(float* __restrict__ x, float* __restrict__ y, int64_t n,
                                                                float lo) {
    const float hi = 1.f - lo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float s = 1.f / (1.f + expf(-x[i]));
        x[i] = s;
        y[i] = fminf(fmaxf(s, lo), hi);
    }
}

__global__ __launch_bounds__(256) void sigmoid_clamp_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                                float* __restrict__ dz, int64_t n, float lo) {
    const float hi = 1.f - lo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float p = s[i];
        dz[i] = (p >= lo && p <= hi) ? dy[i] * p * (1.f - p) : 0.f;
    }
}

// 16-byte form of the streaming loss kernels (n % 4 == 0, 16-byte aligned, gt not broadcast): same per-element arithmetic, four
// elements per lane per access — the scalar forms ran at 3.4-3.6 TB/s on the 335 MB class heat map.
__global__ __launch_bounds__(256) void sigmoid_clamp_fwd_vec_kernel(float4* __restrict__ x, float4* __restrict__ y, int64_t n4, float lo) {
    const float hi = 1.f - lo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = x[i];
        float* e = &v.x;
        float4 c;
        float* ce = &c.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sg = 1.f / (1.f + expf(-e[j]));
            e[j] = sg;
            ce[j] = fminf(fmaxf(sg, lo), hi);
        }
        x[i] = v;
        y[i] = c;
    }
}

static inline bool loss_vec_ok(const void* a, const void* b, const void* c, int64_t n) {
    return n % 4 == 0 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0);
}

extern "C" int cn_sigmoid_clamp_fwd(float* x, float* y, int64_t n, float lo, void* stream) {
    CN_CHECK_ARG(x && y && n > 0, "cn_sigmoid_clamp_fwd: bad args");
    if (loss_vec_ok(x, y, nullptr, n)) {
        int64_t g4 = (n / 4 + 255) / 256;
        hipLaunchKernelGGL(sigmoid_clamp_fwd_vec_kernel, dim3((int)(g4 > 8192 ? 8192 : g4)), dim3(256), 0, (hipStream_t)stream,
                           (float4*)x, (float4*)y, n / 4, lo);
        CN_LAUNCH_CHECK("cn_sigmoid_clamp_fwd");
        return CN_OK;
    }
    int64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(sigmoid_clamp_fwd_kernel, dim3((int)(g > 16384 ? 16384 : g)), dim3(256), 0, (hipStream_t)stream, x, y, n, lo);
    CN_LAUNCH_CHECK("cn_sigmoid_clamp_fwd");
    return CN_OK;
}

extern "C" int cn_sigmoid_clamp_bwd(const float* dy, const float* x_sig, float* dz, int64_t n, float lo, void* stream) {
    CN_CHECK_ARG(dy && x_sig && dz && n > 0, "cn_sigmoid_clamp_bwd: bad args");
    int64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(sigmoid_clamp_bwd_kernel, dim3((int)(g > 16384 ? 16384 : g)), dim3(256), 0, (hipStream_t)stream, dy, x_sig, dz, n, lo);
    CN_LAUNCH_CHECK("cn_sigmoid_clamp_bwd");
    return CN_OK;
}

// ------------------------------------------------------------------------------------------------ focal
extern "C" size_t cn_focal_workspace_bytes(int64_t n) {
    (void)n;
    return (size_t)FOCAL_MAX_BLOCKS * 3 * sizeof(float);
}

__device__ static inline int64_t gt_index(int64_t i, int64_t HW, int C, int gtB, int gtC) {
    // pred index i = (b*C + c)*HW + s  ->  gt index with size-1 broadcast on batch / channel
    const int64_t s = i % HW;
    const int64_t bc = i / HW;
    const int c = (int)(bc % C);
    const int64_t b = bc / C;
    return ((gtB == 1 ? 0 : b) * gtC + (gtC == 1 ? 0 : c)) * HW + s;
}

__global__ __launch_bounds__(256) void focal_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                        float* __restrict__ part, int64_t n, int64_t HW, int C, int gtB,
                                                        int gtC, int same) {
    float pos = 0.f, neg = 0.f, np = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float p = pred[i];
        const float g = gt[same ? i : gt_index(i, HW, C, gtB, gtC)];
        if (g == 1.f) {
            const float q = 1.f - p;
            pos += logf(p) * q * q;
            np += 1.f;
        } else if (g < 1.f) {
            const float w = 1.f - g;
            const float w2 = w * w;
            neg += logf(1.f - p) * p * p * (w2 * w2);
        }
    }
    __shared__ float red[3][4];
    pos = wave_sum(pos); neg = wave_sum(neg); np = wave_sum(np);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[0][wv] = pos; red[1][wv] = neg; red[2][wv] = np; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 3 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        part[blockIdx.x * 3 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        part[blockIdx.x * 3 + 2] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    }
}

__global__ __launch_bounds__(256) void focal_fwd_vec_kernel(const float4* __restrict__ pred, const float4* __restrict__ gt,
                                                            float* __restrict__ part, int64_t n4) {
    float pos = 0.f, neg = 0.f, np = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 pv = pred[i], gv = gt[i];
        const float* pe = &pv.x;
        const float* ge = &gv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float p = pe[j], g = ge[j];
            if (g == 1.f) {
                const float q = 1.f - p;
                pos += logf(p) * q * q;
                np += 1.f;
            } else if (g < 1.f) {
                const float w = 1.f - g;
                const float w2 = w * w;
                neg += logf(1.f - p) * p * p * (w2 * w2);
            }
        }
    }
    __shared__ float red[3][4];
    pos = wave_sum(pos); neg = wave_sum(neg); np = wave_sum(np);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[0][wv] = pos; red[1][wv] = neg; red[2][wv] = np; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 3 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        part[blockIdx.x * 3 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        part[blockIdx.x * 3 + 2] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    }
}

__global__ __launch_bounds__(64) void focal_finalize_kernel(const float* __restrict__ part, int nblk, float* __restrict__ out4) {
    double pos = 0.0, neg = 0.0, np = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) { pos += part[b * 3]; neg += part[b * 3 + 1]; np += part[b * 3 + 2]; }
    pos = wave_sum_d(pos); neg = wave_sum_d(neg); np = wave_sum_d(np);
    if (threadIdx.x == 0) {
        out4[0] = (float)(np == 0.0 ? -neg : -(pos + neg) / np);
        out4[1] = (float)pos;
        out4[2] = (float)neg;
        out4[3] = (float)np;
    }
}

__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                        const float* __restrict__ out4, const float* __restrict__ gout,
                                                        float* __restrict__ dpred, int64_t n, int64_t HW, int C, int gtB,
                                                        int gtC, int same) {
    const float np = out4[3];
    const float scale = -gout[0] / (np == 0.f ? 1.f : np);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float p = pred[i];
        const float g = gt[same ? i : gt_index(i, HW, C, gtB, gtC)];
        float d = 0.f;
        if (g == 1.f) {
            const float q = 1.f - p;
            d = q * q / p - 2.f * q * logf(p);
        } else if (g < 1.f) {
            const float w = 1.f - g;
            const float w2 = w * w;
            d = (w2 * w2) * (2.f * p * logf(1.f - p) - p * p / (1.f - p));
        }
        dpred[i] = d * scale;
    }
}

// d loss / d logits in ONE pass: focal backward on p = clamp(s) times the sigmoid/clamp backward s(1-s)[lo <= s <= 1-lo]
// (utils/losses.py:14-39 through utils/decode.py:43-45).  The two-kernel form writes and re-reads d loss / d p (2 x 335 MB at C3).
__global__ __launch_bounds__(256) void sigmoid_focal_bwd_kernel(const float* __restrict__ s, const float* __restrict__ gt,
                                                                const float* __restrict__ out4, const float* __restrict__ gout,
                                                                float* __restrict__ dz, int64_t n, int64_t HW, int C, int gtB,
                                                                int gtC, int same, float lo) {
    const float np = out4[3], hi = 1.f - lo;
    const float scale = -gout[0] / (np == 0.f ? 1.f : np);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float sv = s[i];
        const float g = gt[same ? i : gt_index(i, HW, C, gtB, gtC)];
        float d = 0.f;
        if (sv >= lo && sv <= hi) {                    // outside, the clamp blocks the gradient (and p == the clamp bound)
            const float p = sv;
            if (g == 1.f) {
                const float q = 1.f - p;
                d = q * q / p - 2.f * q * logf(p);
            } else if (g < 1.f) {
                const float w = 1.f - g;
                const float w2 = w * w;
                d = (w2 * w2) * (2.f * p * logf(1.f - p) - p * p / (1.f - p));
            }
            d = (d * scale) * p * (1.f - p);           // same association as the two kernels: (d * scale), then * p * (1 - p)
        }
        dz[i] = d;
    }
}

__global__ __launch_bounds__(256) void sigmoid_focal_bwd_vec_kernel(const float4* __restrict__ s, const float4* __restrict__ gt,
                                                                    const float* __restrict__ out4, const float* __restrict__ gout,
                                                                    float4* __restrict__ dz, int64_t n4, float lo) {
    const float np = out4[3], hi = 1.f - lo;
    const float scale = -gout[0] / (np == 0.f ? 1.f : np);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 sv4 = s[i], gv4 = gt[i];
        const float* se = &sv4.x;
        const float* ge = &gv4.x;
        float4 o;
        float* oe = &o.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sv = se[j], g = ge[j];
            float d = 0.f;
            if (sv >= lo && sv <= hi) {
                const float p = sv;
                if (g == 1.f) {
                    const float q = 1.f - p;
                    d = q * q / p - 2.f * q * logf(p);
                } else if (g < 1.f) {
                    const float w = 1.f - g;
                    const float w2 = w * w;
                    d = (w2 * w2) * (2.f * p * logf(1.f - p) - p * p / (1.f - p));
                }
                d = (d * scale) * p * (1.f - p);
            }
            oe[j] = d;
        }
        dz[i] = o;
    }
}

// sigmoid_focal_bwd_vec_kernel that ALSO leaves d loss / d logits in the layout its usual consumer wants (NHWC, compute dtype bf16,
// channels zero-padded to `ld`): a workgroup owns 64 consecutive pixels of one image x all C channels — fp32 NCHW rows out as they
// are computed (256-byte runs per channel), the bf16 transpose staged in LDS and written as whole pixel rows.  Saves the consumer's
// cn_nchw_to_nhwc pass (read 4 B + write 2 B per element) for one extra 2 B write here.  HW % 64 == 0, ld % 8 == 0, C <= ld <= 256.
__global__ __launch_bounds__(256) void sigmoid_focal_bwd_dual_kernel(const float* __restrict__ s, const float* __restrict__ gt,
                                                                     const float* __restrict__ out4, const float* __restrict__ gout,
                                                                     float* __restrict__ dz, bf16_t* __restrict__ dz_nhwc, int C,
                                                                     int64_t HW, int ld, float lo) {
    extern __shared__ __attribute__((aligned(16))) bf16_t tile[];          // [64 px][ld + 8]
    const int P = ld + 8;
    const float np = out4[3], hi = 1.f - lo;
    const float scale = -gout[0] / (np == 0.f ? 1.f : np);
    const int64_t blocks_per_img = HW / 64;
    const int64_t b = blockIdx.x / blocks_per_img, p0 = (blockIdx.x - b * blocks_per_img) * 64;
    const int tid = threadIdx.x, px4 = tid & 15, crow = tid >> 4;
    for (int i = tid; i < 64 * (ld - C); i += 256) tile[(i / (ld - C)) * P + C + i % (ld - C)] = 0;      // channel padding
    for (int c = crow; c < C; c += 16) {
        const int64_t off = (b * C + c) * HW + p0 + 4 * px4;
        const float4 sv4 = *reinterpret_cast<const float4*>(s + off), gv4 = *reinterpret_cast<const float4*>(gt + off);
        const float* se = &sv4.x;
        const float* ge = &gv4.x;
        float4 o;
        float* oe = &o.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sv = se[j], g = ge[j];
            float d = 0.f;
            if (sv >= lo && sv <= hi) {
                const float p = sv;
                if (g == 1.f) {
                    const float q = 1.f - p;
                    d = q * q / p - 2.f * q * logf(p);
                } else if (g < 1.f) {
                    const float w = 1.f - g;
                    const float w2 = w * w;
                    d = (w2 * w2) * (2.f * p * logf(1.f - p) - p * p / (1.f - p));
                }
                d = (d * scale) * p * (1.f - p);
            }
            oe[j] = d;
            tile[(4 * px4 + j) * P + c] = f2bf(d);
        }
        *reinterpret_cast<float4*>(dz + off) = o;
    }
    __syncthreads();
    const int vpp = ld / 8;                                                // 16-byte vectors per pixel
    bf16_t* const dst = dz_nhwc + (b * HW + p0) * ld;
    for (int v = tid; v < 64 * vpp; v += 256) {
        const int px = v / vpp, cv = v - px * vpp;
        st16(dst + (int64_t)px * ld + cv * 8, *reinterpret_cast<const uint4*>(tile + px * P + cv * 8));
    }
}

static int focal_grid(int64_t n) {
    int64_t g = (n + 256 * 8 - 1) / (256 * 8);
    return (int)(g > FOCAL_MAX_BLOCKS ? FOCAL_MAX_BLOCKS : (g < 1 ? 1 : g));
}

extern "C" int cn_focal_fwd(const float* pred, const float* gt, float* out4, int B, int C, int64_t HW, int gtB, int gtC,
                            void* ws, size_t ws_bytes, void* stream) {
    CN_CHECK_ARG(pred && gt && out4 && ws && B > 0 && C > 0 && HW > 0, "cn_focal_fwd: bad args");
    CN_CHECK_ARG((gtB == B || gtB == 1) && (gtC == C || gtC == 1), "cn_focal_fwd: gt [%d,%d] does not broadcast to [%d,%d]", gtB, gtC, B, C);
    if (ws_bytes < cn_focal_workspace_bytes(0)) { cn_set_error("cn_focal_fwd: workspace too small"); return CN_EWORKSPACE; }
    const int64_t n = (int64_t)B * C * HW;
    const int grid = focal_grid(n);
    if (gtB == B && gtC == C && loss_vec_ok(pred, gt, nullptr, n))
        hipLaunchKernelGGL(focal_fwd_vec_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4*)pred, (const float4*)gt,
                           (float*)ws, n / 4);
    else
        hipLaunchKernelGGL(focal_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pred, gt, (float*)ws, n, HW, C, gtB, gtC,
                           (int)(gtB == B && gtC == C));
    CN_LAUNCH_CHECK("cn_focal_fwd");
    hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)ws, grid, out4);
    CN_LAUNCH_CHECK("cn_focal_fwd(finalize)");
    return CN_OK;
}

// sigmoid_clamp_fwd_vec_kernel + focal_fwd_vec_kernel in ONE pass over the class heat map: x <- sigmoid(x), y <- clamp(x), focal
// partial sums of y against gt (same grid, same per-thread order as focal_fwd_vec_kernel: the partials are bit-identical to the
// two-kernel sequence, which re-read the 335 MB of y it had just written)
__global__ __launch_bounds__(256) void sigmoid_clamp_focal_fwd_vec_kernel(float4* __restrict__ x, float4* __restrict__ y,
                                                                          const float4* __restrict__ gt, float* __restrict__ part,
                                                                          int64_t n4, float lo) {
    const float hi = 1.f - lo;
    float pos = 0.f, neg = 0.f, np = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = x[i];
        const float4 gv = gt[i];
        float* e = &v.x;
        const float* ge = &gv.x;
        float4 c;
        float* ce = &c.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sg = 1.f / (1.f + expf(-e[j]));
            e[j] = sg;
            const float p = fminf(fmaxf(sg, lo), hi), g = ge[j];
            ce[j] = p;
            if (g == 1.f) {
                const float q = 1.f - p;
                pos += logf(p) * q * q;
                np += 1.f;
            } else if (g < 1.f) {
                const float w = 1.f - g;
                const float w2 = w * w;
                neg += logf(1.f - p) * p * p * (w2 * w2);
            }
        }
        x[i] = v;
        y[i] = c;
    }
    __shared__ float red[3][4];
    pos = wave_sum(pos); neg = wave_sum(neg); np = wave_sum(np);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[0][wv] = pos; red[1][wv] = neg; red[2][wv] = np; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 3 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        part[blockIdx.x * 3 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        part[blockIdx.x * 3 + 2] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    }
}

extern "C" int cn_sigmoid_clamp_focal_fwd(float* x, float* y, const float* gt, float* out4, int64_t n, float lo, void* ws,
                                          size_t ws_bytes, void* stream) {
    CN_CHECK_ARG(x && y && gt && out4 && ws && n > 0, "cn_sigmoid_clamp_focal_fwd: bad args");
    if (ws_bytes < cn_focal_workspace_bytes(0)) { cn_set_error("cn_sigmoid_clamp_focal_fwd: workspace too small"); return CN_EWORKSPACE; }
    if (!loss_vec_ok(x, y, gt, n)) CN_UNSUPPORTED("cn_sigmoid_clamp_focal_fwd: needs n %% 4 == 0 and 16-byte aligned maps (use the two-kernel sequence)");
    const int grid = focal_grid(n);
    hipLaunchKernelGGL(sigmoid_clamp_focal_fwd_vec_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float4*)x, (float4*)y,
                       (const float4*)gt, (float*)ws, n / 4, lo);
    CN_LAUNCH_CHECK("cn_sigmoid_clamp_focal_fwd");
    hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)ws, grid, out4);
    CN_LAUNCH_CHECK("cn_sigmoid_clamp_focal_fwd(finalize)");
    return CN_OK;
}

extern "C" int cn_focal_bwd(const float* pred, const float* gt, const float* out4, const float* gout, float* dpred, int B,
                            int C, int64_t HW, int gtB, int gtC, void* stream) {
    CN_CHECK_ARG(pred && gt && out4 && gout && dpred && B > 0 && C > 0 && HW > 0, "cn_focal_bwd: bad args");
    CN_CHECK_ARG((gtB == B || gtB == 1) && (gtC == C || gtC == 1), "cn_focal_bwd: gt does not broadcast");
    const int64_t n = (int64_t)B * C * HW;
    int64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(focal_bwd_kernel, dim3((int)(g > 16384 ? 16384 : g)), dim3(256), 0, (hipStream_t)stream, pred, gt, out4,
                       gout, dpred, n, HW, C, gtB, gtC, (int)(gtB == B && gtC == C));
    CN_LAUNCH_CHECK("cn_focal_bwd");
    return CN_OK;
}

extern "C" int cn_sigmoid_focal_bwd(const float* x_sig, const float* gt, const float* out4, const float* gout, float* dz, int B, int C,
                                    int64_t HW, int gtB, int gtC, float lo, void* stream) {
    CN_CHECK_ARG(x_sig && gt && out4 && gout && dz && B > 0 && C > 0 && HW > 0, "cn_sigmoid_focal_bwd: bad args");
    CN_CHECK_ARG((gtB == B || gtB == 1) && (gtC == C || gtC == 1), "cn_sigmoid_focal_bwd: gt does not broadcast");
    const int64_t n = (int64_t)B * C * HW;
    if (gtB == B && gtC == C && loss_vec_ok(x_sig, gt, dz, n)) {
        int64_t g4 = (n / 4 + 255) / 256;
        hipLaunchKernelGGL(sigmoid_focal_bwd_vec_kernel, dim3((int)(g4 > 8192 ? 8192 : g4)), dim3(256), 0, (hipStream_t)stream,
                           (const float4*)x_sig, (const float4*)gt, out4, gout, (float4*)dz, n / 4, lo);
        CN_LAUNCH_CHECK("cn_sigmoid_focal_bwd");
        return CN_OK;
    }
    int64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(sigmoid_focal_bwd_kernel, dim3((int)(g > 16384 ? 16384 : g)), dim3(256), 0, (hipStream_t)stream, x_sig, gt, out4,
                       gout, dz, n, HW, C, gtB, gtC, (int)(gtB == B && gtC == C), lo);
    CN_LAUNCH_CHECK("cn_sigmoid_focal_bwd");
    return CN_OK;
}

extern "C" int cn_sigmoid_focal_bwd_dual(const float* x_sig, const float* gt, const float* out4, const float* gout, float* dz,
                                         void* dz_nhwc, int B, int C, int64_t HW, int ld, float lo, void* stream) {
    CN_CHECK_ARG(x_sig && gt && out4 && gout && dz && dz_nhwc && B > 0 && C > 0 && HW > 0, "cn_sigmoid_focal_bwd_dual: bad args");
    if (HW % 64 != 0 || (ld & 7) || ld < C || ld > 256 || B * (HW / 64) > 0x7fffffffLL ||
        ((((uintptr_t)x_sig | (uintptr_t)gt | (uintptr_t)dz | (uintptr_t)dz_nhwc)) & 15))
        CN_UNSUPPORTED("cn_sigmoid_focal_bwd_dual: needs HW %% 64 == 0, C <= ld <= 256, ld %% 8 == 0, 16-byte aligned maps");
    const size_t smem = (size_t)64 * (ld + 8) * sizeof(bf16_t);
    hipLaunchKernelGGL(sigmoid_focal_bwd_dual_kernel, dim3((unsigned)(B * (HW / 64))), dim3(256), smem, (hipStream_t)stream, x_sig, gt,
                       out4, gout, dz, (bf16_t*)dz_nhwc, C, HW, ld, lo);
    CN_LAUNCH_CHECK("cn_sigmoid_focal_bwd_dual");
    return CN_OK;
}

// ------------------------------------------------------------------------------------------------ gather L1
__global__ __launch_bounds__(1024) void gather_l1_fwd_kernel(const float* __restrict__ feat, const int64_t* __restrict__ ind,
                                                             const uint8_t* __restrict__ mask, const float* __restrict__ tgt,
                                                             float* __restrict__ out3, int B, int C, int64_t HW, int N,
                                                             int mask_has_c) {
    const int64_t total = (int64_t)B * N * C;
    double s = 0.0, ms = 0.0;
    for (int64_t i = threadIdx.x; i < total; i += 1024) {
        const int c = (int)(i % C);
        const int64_t bn = i / C;
        const int b = (int)(bn / N);
        const float m = mask[mask_has_c ? i : bn] ? 1.f : 0.f;
        int64_t id = ind[bn];
        id = id < 0 ? 0 : (id >= HW ? HW - 1 : id);
        const float p = feat[((int64_t)b * C + c) * HW + id];
        s += (double)fabsf(p * m - tgt[i] * m);
        ms += (double)m;
    }
    __shared__ double red[2][16];
    s = wave_sum_d(s); ms = wave_sum_d(ms);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[0][wv] = s; red[1][wv] = ms; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b2 = 0.0;
        for (int w = 0; w < 16; ++w) { a += red[0][w]; b2 += red[1][w]; }
        out3[0] = (float)a / ((float)b2 + 1e-4f);
        out3[1] = (float)a;
        out3[2] = (float)b2;
    }
}

__global__ __launch_bounds__(256) void gather_l1_bwd_kernel(const float* __restrict__ feat, const int64_t* __restrict__ ind,
                                                            const uint8_t* __restrict__ mask, const float* __restrict__ tgt,
                                                            const float* __restrict__ out3, const float* __restrict__ gout,
                                                            float* __restrict__ dfeat, int B, int C, int64_t HW, int N,
                                                            int mask_has_c) {
    const int64_t total = (int64_t)B * N * C;
    const float scale = gout[0] / (out3[2] + 1e-4f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t bn = i / C;
        const int b = (int)(bn / N);
        if (!mask[mask_has_c ? i : bn]) continue;
        int64_t id = ind[bn];
        id = id < 0 ? 0 : (id >= HW ? HW - 1 : id);
        const int64_t a = ((int64_t)b * C + c) * HW + id;
        const float d = feat[a] - tgt[i];
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        if (sg != 0.f) atomicAdd(dfeat + a, sg * scale);
    }
}

extern "C" int cn_gather_l1_fwd(const float* feat, const int64_t* ind, const uint8_t* mask, const float* target, float* out3,
                                int B, int C, int64_t HW, int N, int mask_has_c, void* stream) {
    CN_CHECK_ARG(feat && ind && mask && target && out3 && B > 0 && C > 0 && HW > 0 && N > 0, "cn_gather_l1_fwd: bad args");
    hipLaunchKernelGGL(gather_l1_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, feat, ind, mask, target, out3, B, C, HW,
                       N, mask_has_c);
    CN_LAUNCH_CHECK("cn_gather_l1_fwd");
    return CN_OK;
}

extern "C" int cn_gather_l1_bwd(const float* feat, const int64_t* ind, const uint8_t* mask, const float* target,
                                const float* out3, const float* gout, float* dfeat, int B, int C, int64_t HW, int N,
                                int mask_has_c, void* stream) {
    CN_CHECK_ARG(feat && ind && mask && target && out3 && gout && dfeat && B > 0 && C > 0, "cn_gather_l1_bwd: bad args");
    int64_t total = (int64_t)B * N * C;
    int64_t g = (total + 255) / 256;
    hipLaunchKernelGGL(gather_l1_bwd_kernel, dim3((int)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream, feat, ind, mask,
                       target, out3, gout, dfeat, B, C, HW, N, mask_has_c);
    CN_LAUNCH_CHECK("cn_gather_l1_bwd");
    return CN_OK;
}

// ---- loss assembly (centernet_detection.py:108-116, centernet_multi_pose.py:126-140): total = sum_i w_i * term_i over <= 8 scalar
// loss terms, and its backward d term_i = w_i * g, one single-thread launch each (the reference leaves this to ~16 ATen scalar
// kernels per step, launched at the forward / backward seam where nothing overlaps them) ----
struct WsumArgs { const float* t[8]; float w[8]; };
__global__ void weighted_sum_kernel(WsumArgs a, int n, float* __restrict__ out) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += a.w[i] * a.t[i][0];
    out[0] = s;
}
__global__ void weighted_sum_bwd_kernel(WsumArgs a, int n, const float* __restrict__ g, float* __restrict__ out) {
    if ((int)threadIdx.x < n) out[threadIdx.x] = a.w[threadIdx.x] * g[0];
}

extern "C" int cn_weighted_sum(const float* t0, const float* t1, const float* t2, const float* t3, const float* t4, const float* t5,
                               const float* t6, const float* t7, float w0, float w1, float w2, float w3, float w4, float w5, float w6,
                               float w7, int n, float* out, void* stream) {
    CN_CHECK_ARG(n >= 1 && n <= 8 && out, "cn_weighted_sum: 1 <= n <= 8 terms");
    WsumArgs a = {{t0, t1, t2, t3, t4, t5, t6, t7}, {w0, w1, w2, w3, w4, w5, w6, w7}};
    for (int i = 0; i < n; ++i) CN_CHECK_ARG(a.t[i] != nullptr, "cn_weighted_sum: term %d is null", i);
    hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, a, n, out);
    CN_LAUNCH_CHECK("cn_weighted_sum");
    return CN_OK;
}

extern "C" int cn_weighted_sum_bwd(const float* g, float w0, float w1, float w2, float w3, float w4, float w5, float w6, float w7,
                                   int n, float* out, void* stream) {
    CN_CHECK_ARG(n >= 1 && n <= 8 && out && g, "cn_weighted_sum_bwd: bad args");
    WsumArgs a = {{nullptr}, {w0, w1, w2, w3, w4, w5, w6, w7}};
    hipLaunchKernelGGL(weighted_sum_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, n, g, out);
    CN_LAUNCH_CHECK("cn_weighted_sum_bwd");
    return CN_OK;
}
