// BatchNorm2d (training statistics, running-stat update, backward) + ReLU + residual add on NHWC.
// All kernels are HBM-bound: 16-byte vector accesses, fp32 accumulation, deterministic two-level
// reduction (per-workgroup partials -> one finalize workgroup in fp64), no atomics.
#include "common.h"

#define BN_MAX_BLOCKS 1024

struct BnLayout {
    int CV;    // 16-byte channel vectors per pixel
    int CVB;   // vectors handled per workgroup column (<= 256)
    int RPB;   // pixel rows per workgroup pass = 256 / CVB
    int ycols; // grid.y
    int nblk;  // grid.x
    int64_t rows_per_blk;
};

static BnLayout bn_layout(int64_t npix, int C, int vec) {
    BnLayout L;
    L.CV = C / vec;
    L.CVB = L.CV < 256 ? L.CV : 256;
    L.RPB = 256 / L.CVB;
    L.ycols = (L.CV + L.CVB - 1) / L.CVB;
    int64_t want = (npix + (int64_t)L.RPB * 8 - 1) / ((int64_t)L.RPB * 8);  // >= 8 passes per workgroup
    L.nblk = (int)(want < 1 ? 1 : (want > BN_MAX_BLOCKS ? BN_MAX_BLOCKS : want));
    L.rows_per_blk = (npix + L.nblk - 1) / L.nblk;
    return L;
}

// layout of the element-wise passes: same thread -> (row slot, channel vector) map, up to 8 workgroups per CU, >= 4*U rows each
static BnLayout ew_layout(int64_t npix, int C, int vec, int wg_cap = 2048) {
    BnLayout L = bn_layout(npix, C, vec);
    int64_t want = (npix + (int64_t)L.RPB * 16 - 1) / ((int64_t)L.RPB * 16);
    int64_t cap = wg_cap / L.ycols;
    L.nblk = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    L.rows_per_blk = (npix + L.nblk - 1) / L.nblk;
    return L;
}

// workgroups of the sink-reducing element-wise kernels: each pays the reduction prologue once (A/B: CN_BN_SINK_WGS)
static int sink_wg_cap() {
    static const int cap = [] { const char* e = getenv("CN_BN_SINK_WGS"); const int v = e ? atoi(e) : 0; return v >= 256 ? v : 2048; }();
    return cap;
}

extern "C" size_t cn_bn_workspace_bytes(int64_t npix, int C) {
    (void)npix;
    // partials [BN_MAX_BLOCKS][2][C] fp32 + coefficients [4][C] fp32
    return ((size_t)BN_MAX_BLOCKS * 2 * C + 4 * (size_t)C) * sizeof(float);
}

// MODE 0: (sum x, sum x^2)      MODE 1: (sum dy', sum dy' * xhat)
// ReLU mask of MODE 1: from the forward output y when given, else recomputed as fma(x, scale, shift) > 0 with the very
// coefficients the forward pass applied (bit-identical decision, one tensor less to read).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                         const T* __restrict__ y, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ ss,
                                                         float* __restrict__ part, int64_t npix, int C, BnLayout L, int relu,
                                                         int sink_slots) {
    CN_MAIN_PRIO_SET();
    constexpr int V = Vec16<T>::N;
    constexpr int U = 4;                                   // rows in flight per thread
    __shared__ float red[2][256][V + 1];
    const int tid = threadIdx.x;
    const int cvl = tid % L.CVB, prow = tid / L.CVB;
    const int cv = blockIdx.y * L.CVB + cvl;
    const bool active = prow < L.RPB && cv < L.CV;
    float s0[V], s1[V], mu[V], is[V], sc[V], sh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { s0[j] = 0.f; s1[j] = 0.f; mu[j] = 0.f; is[j] = 1.f; sc[j] = 0.f; sh[j] = 0.f; }
    const bool mask_x = MODE == 1 && relu && y == nullptr;
    if (active) {
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < V; ++j) { mu[j] = mean[cv * V + j]; is[j] = invstd[cv * V + j]; }
            if (mask_x) {
#pragma unroll
                for (int j = 0; j < V; ++j) { sc[j] = ss[cv * V + j]; sh[j] = ss[C + cv * V + j]; }
            }
        }
        const int64_t r0 = (int64_t)blockIdx.x * L.rows_per_blk;
        const int64_t r1 = r0 + L.rows_per_blk < npix ? r0 + L.rows_per_blk : npix;
        for (int64_t rb = r0 + prow; rb < r1; rb += (int64_t)U * L.RPB) {
            uint4 xr[U], gr[U], yr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                  // branch-free: rows past the end re-read row rb and are masked
                const int64_t r = rb + (int64_t)u * L.RPB;
                const int64_t rc = r < r1 ? r : rb;
                xr[u] = *reinterpret_cast<const uint4*>(x + rc * C + cv * V);
                if (MODE == 1) {
                    gr[u] = *reinterpret_cast<const uint4*>(dy + rc * C + cv * V);
                    if (relu && !mask_x) yr[u] = *reinterpret_cast<const uint4*>(y + rc * C + cv * V);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = rb + (int64_t)u * L.RPB < r1;
                float xv[V], gv[V], yv[V];
                Vec16<T>::unpack(xr[u], xv);
                if (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        const float xx = ok ? xv[j] : 0.f;
                        s0[j] += xx; s1[j] = fmaf(xx, xx, s1[j]);
                    }
                } else {
                    Vec16<T>::unpack(gr[u], gv);
                    if (relu) {
                        if (mask_x) {
#pragma unroll
                            for (int j = 0; j < V; ++j) gv[j] = fmaf(xv[j], sc[j], sh[j]) > 0.f ? gv[j] : 0.f;
                        } else {
                            Vec16<T>::unpack(yr[u], yv);
#pragma unroll
                            for (int j = 0; j < V; ++j) gv[j] = yv[j] > 0.f ? gv[j] : 0.f;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        const float gg = ok ? gv[j] : 0.f;
                        s0[j] += gg; s1[j] = fmaf(gg, (xv[j] - mu[j]) * is[j], s1[j]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) { red[0][tid][j] = s0[j]; red[1][tid][j] = s1[j]; }
    __syncthreads();
    int top = 1;
    while (top < L.RPB) top <<= 1;
    for (int st = top >> 1; st > 0; st >>= 1) {            // fixed-shape tree over the row slots: deterministic
        if (prow < st && prow + st < L.RPB) {
#pragma unroll
            for (int j = 0; j < V; ++j) { red[0][tid][j] += red[0][tid + st * L.CVB][j]; red[1][tid][j] += red[1][tid + st * L.CVB][j]; }
        }
        __syncthreads();
    }
    if (prow == 0 && cv < L.CV) {
        if (sink_slots > 0) {       // statistics sink (all-zero on entry): the apply kernel reduces its rows itself, no finalize launch
            float* row = part + ((int64_t)(blockIdx.x % sink_slots) * 2) * C + cv * V;
#pragma unroll
            for (int j = 0; j < V; ++j) { atomicAdd(row + j, red[0][tid][j]); atomicAdd(row + C + j, red[1][tid][j]); }
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                part[((int64_t)blockIdx.x * 2 + 0) * C + cv * V + j] = red[0][tid][j];
                part[((int64_t)blockIdx.x * 2 + 1) * C + cv * V + j] = red[1][tid][j];
            }
        }
    }
}

// ---- statistics sinks reduced by their consumer: the element-wise kernels below take part[slots][2][C] (filled by the producing
// conv's epilogue in the forward pass, by bn_partial_kernel<1> in the backward pass) and every workgroup reduces the rows of its own
// channel vectors in its prologue (slots * 2 * C * 4 bytes from L2), instead of a 6-10 us finalize launch in front of every BN
// pass (106 dependent launches per DLA-34 step).  Workgroup column 0 also writes what the finalize kernels wrote (saved
// statistics, running statistics, dgamma / dbeta).  A sink cannot be cleared by the kernel that reads it (no grid-wide order), so
// each of these kernels clears ANOTHER sink whose readers have all retired: the one the previous BN pass on the stream consumed.
template <int V>
__device__ inline void bn_sink_totals(const float* __restrict__ part, int slots, int C, const BnLayout& L, int cv, bool cv_ok,
                                      float (*red)[256][V + 1], float (&t0)[V], float (&t1)[V]) {
    const int tid = threadIdx.x, cvl = tid % L.CVB, prow = tid / L.CVB;
    float s0[V], s1[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
    if (cv_ok && prow < L.RPB) {
        for (int sl = prow; sl < slots; sl += L.RPB) {
            const float* p = part + ((int64_t)sl * 2) * C + cv * V;
#pragma unroll
            for (int v = 0; v < V / 4; ++v) {
                const float4 a = *reinterpret_cast<const float4*>(p + 4 * v);
                const float4 b = *reinterpret_cast<const float4*>(p + C + 4 * v);
                s0[4 * v] += a.x; s0[4 * v + 1] += a.y; s0[4 * v + 2] += a.z; s0[4 * v + 3] += a.w;
                s1[4 * v] += b.x; s1[4 * v + 1] += b.y; s1[4 * v + 2] += b.z; s1[4 * v + 3] += b.w;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) { red[0][tid][j] = s0[j]; red[1][tid][j] = s1[j]; }
    __syncthreads();
    int top = 1;
    while (top < L.RPB) top <<= 1;
    for (int st = top >> 1; st > 0; st >>= 1) {
        if (prow < st && prow + st < L.RPB) {
#pragma unroll
            for (int j = 0; j < V; ++j) { red[0][tid][j] += red[0][tid + st * L.CVB][j]; red[1][tid][j] += red[1][tid + st * L.CVB][j]; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < V; ++j) { t0[j] = red[0][cvl][j]; t1[j] = red[1][cvl][j]; }      // row slot 0 of the column holds the totals
}

__device__ inline void bn_clear_retired(float* __restrict__ clear, int clear_n) {
    if (!clear) return;
    const int nthr = gridDim.x * gridDim.y * 256;
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < clear_n; i += nthr) clear[i] = 0.f;
}

// scale_shift_act_kernel with the forward finalize (bn_finalize_fwd_kernel) in its prologue
template <typename T>
__global__ __launch_bounds__(256) void bn_fwd_apply_sink_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                                const float* __restrict__ part, int slots,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ rmean, float* __restrict__ rvar,
                                                                float* __restrict__ smean, float* __restrict__ sinvstd,
                                                                float* __restrict__ save_ss, float momentum, float eps,
                                                                int64_t npix, int C, BnLayout L, int relu,
                                                                float* __restrict__ clear, int clear_n) {
    CN_MAIN_PRIO_SET();
    constexpr int V = Vec16<T>::N;
    constexpr int U = 4;
    __shared__ float red[2][256][V + 1];
    const int tid = threadIdx.x;
    const int cvl = tid % L.CVB, prow = tid / L.CVB;
    const int cv = blockIdx.y * L.CVB + cvl;
    const bool cv_ok = cv < L.CV;
    bn_clear_retired(clear, clear_n);
    float t0[V], t1[V];
    bn_sink_totals<V>(part, slots, C, L, cv, cv_ok, red, t0, t1);
    if (prow >= L.RPB || !cv_ok) return;
    float sc[V], sh[V];
    const bool writer = blockIdx.x == 0 && prow == 0;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int c = cv * V + j;
        const double m = (double)t0[j] / (double)npix;
        double var = (double)t1[j] / (double)npix - m * m;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        sc[j] = gamma[c] * invstd;
        sh[j] = beta[c] - (float)m * sc[j];
        if (writer) {
            smean[c] = (float)m;
            sinvstd[c] = invstd;
            if (rmean) {
                const double unb = npix > 1 ? var * (double)npix / (double)(npix - 1) : var;
                rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
                rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
            }
            if (save_ss) { save_ss[c] = sc[j]; save_ss[C + c] = sh[j]; }
        }
    }
    const int64_t r0 = (int64_t)blockIdx.x * L.rows_per_blk;
    const int64_t r1 = r0 + L.rows_per_blk < npix ? r0 + L.rows_per_blk : npix;
    for (int64_t rb = r0 + prow; rb < r1; rb += (int64_t)U * L.RPB) {
        uint4 xr[U], rr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = rb + (int64_t)u * L.RPB;
            const int64_t rc = r < r1 ? r : rb;
            xr[u] = ldg16(x + rc * C + cv * V);
            if (res) rr[u] = ldg16(res + rc * C + cv * V);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = rb + (int64_t)u * L.RPB;
            float xv[V], rv[V];
            Vec16<T>::unpack(xr[u], xv);
            if (res) Vec16<T>::unpack(rr[u], rv);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float v = fmaf(xv[j], sc[j], sh[j]);
                if (res) v += rv[j];
                if (relu) v = fmaxf(v, 0.f);
                xv[j] = v;
            }
            if (r < r1) Vec16<T>::store(y + r * C + cv * V, xv);
        }
    }
}

// coef layout (after the partials): [0]=scale|a  [1]=shift|b  [2]=c  [3]=unused
__global__ __launch_bounds__(256) void bn_finalize_fwd_kernel(const float* __restrict__ part, int nblk, int C, int64_t npix,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ rmean, float* __restrict__ rvar,
                                                              float* __restrict__ smean, float* __restrict__ sinvstd,
                                                              float* __restrict__ coef, float* __restrict__ save_ss,
                                                              float momentum, float eps, float* __restrict__ clear) {
    // one wave per channel: lanes stride over the per-workgroup partials, fp64 butterfly
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int b = lane; b < nblk; b += 64) { s += (double)part[((int64_t)b * 2) * C + c]; q += (double)part[((int64_t)b * 2 + 1) * C + c]; }
    if (clear) {            // statistics sink of a producing kernel (cn_hooks.bn_part): hand it back all-zero
        for (int b = lane; b < nblk; b += 64) { clear[((int64_t)b * 2) * C + c] = 0.f; clear[((int64_t)b * 2 + 1) * C + c] = 0.f; }
    }
    s = wave_sum_d(s); q = wave_sum_d(q);
    if (lane != 0) return;
    const double m = s / (double)npix;
    double var = q / (double)npix - m * m;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    smean[c] = (float)m;
    sinvstd[c] = invstd;
    if (rmean) {
        const double unb = npix > 1 ? var * (double)npix / (double)(npix - 1) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
    const float sc = gamma[c] * invstd;
    const float shf = beta[c] - (float)m * sc;
    coef[c] = sc;
    coef[C + c] = shf;
    if (save_ss) { save_ss[c] = sc; save_ss[C + c] = shf; }
}

__global__ __launch_bounds__(256) void bn_finalize_bwd_kernel(const float* __restrict__ part, int nblk, int C, int64_t npix,
                                                              const float* __restrict__ gamma, const float* __restrict__ sinvstd,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coef, int accumulate) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int b = lane; b < nblk; b += 64) { s += (double)part[((int64_t)b * 2) * C + c]; q += (double)part[((int64_t)b * 2 + 1) * C + c]; }
    s = wave_sum_d(s); q = wave_sum_d(q);
    if (lane != 0) return;
    dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;      // accumulate: deposit straight into the .grad buffers
    dgamma[c] = accumulate ? dgamma[c] + (float)q : (float)q;
    coef[c] = gamma[c] * sinvstd[c];
    coef[C + c] = (float)(s / (double)npix);
    coef[2 * C + c] = (float)(q / (double)npix);
}

// Element-wise passes with the per-channel coefficients in REGISTERS: a thread owns one 16-byte channel vector (the same one
// for every pixel row it visits, layout of bn_partial_kernel) and keeps U rows in flight.  The grid-stride form these replace
// re-loaded 2-5 coefficients per channel from L1 for every vector: 16-40 four-byte loads next to 2-3 sixteen-byte ones
// (3.7 TB/s on 64ch @128^2; the loads, not HBM, were the limit).
template <typename T>
__global__ __launch_bounds__(256) void scale_shift_act_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                              T* __restrict__ y, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int64_t npix, int C, BnLayout L,
                                                              int relu) {
    CN_MAIN_PRIO_SET();
    constexpr int V = Vec16<T>::N;
    constexpr int U = 4;
    const int tid = threadIdx.x;
    const int cvl = tid % L.CVB, prow = tid / L.CVB;
    const int cv = blockIdx.y * L.CVB + cvl;
    if (prow >= L.RPB || cv >= L.CV) return;
    float sc[V], sh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { sc[j] = scale[cv * V + j]; sh[j] = shift[cv * V + j]; }
    const int64_t r0 = (int64_t)blockIdx.x * L.rows_per_blk;
    const int64_t r1 = r0 + L.rows_per_blk < npix ? r0 + L.rows_per_blk : npix;
    for (int64_t rb = r0 + prow; rb < r1; rb += (int64_t)U * L.RPB) {
        uint4 xr[U], rr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                      // branch-free: rows past the end re-read row rb, their store is skipped
            const int64_t r = rb + (int64_t)u * L.RPB;
            const int64_t rc = r < r1 ? r : rb;
            xr[u] = ldg16(x + rc * C + cv * V);
            if (res) rr[u] = ldg16(res + rc * C + cv * V);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = rb + (int64_t)u * L.RPB;
            float xv[V], rv[V];
            Vec16<T>::unpack(xr[u], xv);
            if (res) Vec16<T>::unpack(rr[u], rv);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float v = fmaf(xv[j], sc[j], sh[j]);
                if (res) v += rv[j];
                if (relu) v = fmaxf(v, 0.f);
                xv[j] = v;
            }
            if (r < r1) Vec16<T>::store(y + r * C + cv * V, xv);
        }
    }
}

// dx = a (g' - b - xhat c),  xhat = (x - mean) invstd   ==   a g' + p x + q   with  p = -a c invstd,  q = -a b - p mean
// (a = gamma invstd, b = mean(g'), c = mean(g' xhat): bn_finalize_bwd_kernel).  g' = ReLU-masked dy.
// SINK: the statistics come from part[slots][2][C] (bn_partial_kernel<1> with atomics), reduced here (bn_finalize_bwd_kernel in the
// prologue: coefficients, dgamma / dbeta by workgroup column 0), and `clear` (another, retired sink) is zeroed on the way
template <typename T, bool SINK>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const T* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ coef,
                                                           const float* __restrict__ ss, T* __restrict__ dx,
                                                           T* __restrict__ dres, const T* __restrict__ racc, int64_t npix, int C,
                                                           BnLayout L, int relu, const float* __restrict__ part, int slots,
                                                           const float* __restrict__ gamma, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int accumulate, float* __restrict__ clear,
                                                           int clear_n) {
    CN_MAIN_PRIO_SET();
    constexpr int V = Vec16<T>::N;
    constexpr int U = 4;
    __shared__ float red[2][SINK ? 256 : 1][V + 1];
    const int tid = threadIdx.x;
    const int cvl = tid % L.CVB, prow = tid / L.CVB;
    const int cv = blockIdx.y * L.CVB + cvl;
    float t0[V], t1[V];
    if constexpr (SINK) {
        bn_clear_retired(clear, clear_n);
        bn_sink_totals<V>(part, slots, C, L, cv, cv < L.CV, red, t0, t1);
    }
    if (prow >= L.RPB || cv >= L.CV) return;
    const bool mask_x = relu && y == nullptr;
    float ca[V], cp[V], cq[V], sc[V], sh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int c = cv * V + j;
        float a, b, cc;
        const float mu = mean[c], is = invstd[c];
        if constexpr (SINK) {
            a = gamma[c] * is;
            b = (float)((double)t0[j] / (double)npix);
            cc = (float)((double)t1[j] / (double)npix);
            if (blockIdx.x == 0 && prow == 0) {
                dbeta[c] = accumulate ? dbeta[c] + t0[j] : t0[j];
                dgamma[c] = accumulate ? dgamma[c] + t1[j] : t1[j];
            }
        } else {
            a = coef[c]; b = coef[C + c]; cc = coef[2 * C + c];
        }
        // same operation order as the two-step form (xh = (x - mu) * is; a * (g - b - xh * cc)) is NOT kept: the fused affine
        // differs from it by rounding only (fp32), far below the bf16 / 2e-5 test tolerances
        ca[j] = a;
        cp[j] = -a * cc * is;
        cq[j] = -a * b - cp[j] * mu;
        sc[j] = mask_x ? ss[c] : 0.f;
        sh[j] = mask_x ? ss[C + c] : 0.f;
    }
    const int64_t r0 = (int64_t)blockIdx.x * L.rows_per_blk;
    const int64_t r1 = r0 + L.rows_per_blk < npix ? r0 + L.rows_per_blk : npix;
    for (int64_t rb = r0 + prow; rb < r1; rb += (int64_t)U * L.RPB) {
        uint4 gr[U], xr[U], yr[U], ar[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = rb + (int64_t)u * L.RPB;
            const int64_t rc = r < r1 ? r : rb;
            gr[u] = ldg16(dy + rc * C + cv * V);
            xr[u] = ldg16(x + rc * C + cv * V);
            if (relu && !mask_x) yr[u] = ldg16(y + rc * C + cv * V);
            if (racc) ar[u] = ldg16(racc + rc * C + cv * V);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = rb + (int64_t)u * L.RPB;
            float gv[V], xv[V], yv[V];
            Vec16<T>::unpack(gr[u], gv);
            Vec16<T>::unpack(xr[u], xv);
            if (relu) {
                if (mask_x) {                              // same decision as the forward pass, without reading y
#pragma unroll
                    for (int j = 0; j < V; ++j) gv[j] = fmaf(xv[j], sc[j], sh[j]) > 0.f ? gv[j] : 0.f;
                } else {
                    Vec16<T>::unpack(yr[u], yv);
#pragma unroll
                    for (int j = 0; j < V; ++j) gv[j] = yv[j] > 0.f ? gv[j] : 0.f;
                }
            }
            if (dres && r < r1) {
                if (racc) {                                // the residual input is a shared tensor: add what its other consumers sent
                    float av[V], rv[V];
                    Vec16<T>::unpack(ar[u], av);
#pragma unroll
                    for (int j = 0; j < V; ++j) rv[j] = gv[j] + av[j];
                    Vec16<T>::store(dres + r * C + cv * V, rv);
                } else {
                    Vec16<T>::store(dres + r * C + cv * V, gv);
                }
            }
#pragma unroll
            for (int j = 0; j < V; ++j) xv[j] = fmaf(ca[j], gv[j], fmaf(cp[j], xv[j], cq[j]));
            if (r < r1) Vec16<T>::store(dx + r * C + cv * V, xv);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                                                       int64_t nvec) {
    constexpr int V = Vec16<T>::N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        float gv[V], yv[V];
        Vec16<T>::load(dy + i * V, gv);
        Vec16<T>::load(y + i * V, yv);
#pragma unroll
        for (int j = 0; j < V; ++j) gv[j] = yv[j] > 0.f ? gv[j] : 0.f;
        Vec16<T>::store(dx + i * V, gv);
    }
}

static int ew_grid(int64_t nvec) {
    int64_t g = (nvec + 255) / 256;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

extern "C" int cn_bn_train_fwd(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                               float* save_scale_shift, int64_t npix, int C, float momentum, float eps, int relu, int dtype,
                               void* ws, size_t ws_bytes, void* stream) {
    CN_CHECK_ARG(x && y && gamma && beta && save_mean && save_invstd && ws && npix > 0 && C > 0, "cn_bn_train_fwd: bad args");
    int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(C % V == 0, "cn_bn_train_fwd: C=%d must be a multiple of %d", C, V);
    if (ws_bytes < cn_bn_workspace_bytes(npix, C)) { cn_set_error("cn_bn_train_fwd: workspace too small"); return CN_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    BnLayout L = bn_layout(npix, C, V);
    float* part = (float*)ws;
    float* coef = part + (size_t)BN_MAX_BLOCKS * 2 * C;
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_partial_kernel<T, 0>), dim3(L.nblk, L.ycols), dim3(256), 0, st,
                                                   (const T*)x, (const T*)nullptr, (const T*)nullptr, (const float*)nullptr,
                                                   (const float*)nullptr, (const float*)nullptr, part, npix, C, L, 0, 0));
    CN_LAUNCH_CHECK("cn_bn_train_fwd(partial)");
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, part, L.nblk, C, npix, gamma, beta,
                       running_mean, running_var, save_mean, save_invstd, coef, save_scale_shift, momentum, eps, (float*)nullptr);
    CN_LAUNCH_CHECK("cn_bn_train_fwd(finalize)");
    BnLayout E = ew_layout(npix, C, V);
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(scale_shift_act_kernel<T>, dim3(E.nblk, E.ycols), dim3(256), 0, st,
                                                   (const T*)x, (const T*)residual, (T*)y, coef, coef + C, npix, C, E, relu));
    CN_LAUNCH_CHECK("cn_bn_train_fwd(apply)");
    return CN_OK;
}

// ---- statistics sink: the kernel that PRODUCES x accumulates sum x / sum x^2 in its epilogue (N1: conv + BN + ReLU in training) ----
// cn_hooks.bn_part of a forward call (cn_conv2d_fwd_h, cn_conv1x1_cat_fwd_h, cn_dcn_fwd_h, cn_stem_conv_fwd_h): if the kernel the call
// dispatches to has the hook, every workgroup adds the per-channel sums of the values it STORES (after rounding to the output dtype)
// to row (workgroup % slots) of part[slots][2][C] with fp32 atomics and the call sets cn_hooks.bn_taken; otherwise nothing is touched
// (the caller then lets cn_bn_train_fwd read x itself).  `part` must be all-zero on entry; cn_bn_train_fwd_stats hands it back all-zero.
extern "C" size_t cn_hooks_size(void) { return sizeof(cn_hooks); }
extern "C" int cn_bn_stats_slots(void) { return BN_STAT_SLOTS; }


// cn_bn_train_fwd with the statistics already in `part` (filled through cn_hooks.bn_part by the kernel that wrote x): finalize
// (+ running-stat update) and the apply pass only — x is read once instead of twice.  `part` is cleared on the way.
extern "C" int cn_bn_train_fwd_stats(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                                     float* save_scale_shift, float* part, int slots, int64_t npix, int C, float momentum, float eps,
                                     int relu, int dtype, void* ws, size_t ws_bytes, void* stream) {
    CN_CHECK_ARG(x && y && gamma && beta && save_mean && save_invstd && ws && part && npix > 0 && C > 0 && slots > 0 && slots <= BN_MAX_BLOCKS,
                 "cn_bn_train_fwd_stats: bad args");
    int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(C % V == 0, "cn_bn_train_fwd_stats: C=%d must be a multiple of %d", C, V);
    if (ws_bytes < cn_bn_workspace_bytes(npix, C)) { cn_set_error("cn_bn_train_fwd_stats: workspace too small"); return CN_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    float* coef = (float*)ws + (size_t)BN_MAX_BLOCKS * 2 * C;
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, part, slots, C, npix, gamma, beta,
                       running_mean, running_var, save_mean, save_invstd, coef, save_scale_shift, momentum, eps, part);
    CN_LAUNCH_CHECK("cn_bn_train_fwd_stats(finalize)");
    BnLayout E = ew_layout(npix, C, V);
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(scale_shift_act_kernel<T>, dim3(E.nblk, E.ycols), dim3(256), 0, st,
                                                   (const T*)x, (const T*)residual, (T*)y, coef, coef + C, npix, C, E, relu));
    CN_LAUNCH_CHECK("cn_bn_train_fwd_stats(apply)");
    return CN_OK;
}

// The statistics half of cn_bn_train_fwd_stats on its own: finalize from `part` (batch mean / invstd, running-stat update, scale | shift
// into save_scale_shift[2][C]) with NO apply pass — the consumer of x applies the affine map itself (cn_hooks.pre_ss).
// `part` is handed back all-zero.
extern "C" int cn_bn_finalize_sink(float* part, int slots, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   float* save_mean, float* save_invstd, float* save_scale_shift, int64_t npix, int C, float momentum,
                                   float eps, void* stream) {
    CN_CHECK_ARG(part && gamma && beta && save_mean && save_invstd && save_scale_shift && npix > 0 && C > 0 && slots > 0 && slots <= BN_MAX_BLOCKS,
                 "cn_bn_finalize_sink: bad args");
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, part, slots, C, npix, gamma, beta,
                       running_mean, running_var, save_mean, save_invstd, save_scale_shift, (float*)nullptr, momentum, eps, part);
    CN_LAUNCH_CHECK("cn_bn_finalize_sink");
    return CN_OK;
}

// cn_bn_train_fwd_stats as ONE launch: the apply kernel reduces `part` itself (see bn_sink_totals).  `part` is left as it is (the
// caller retires it); `clear` (nullable, clear_n floats, 16-byte aligned, NOT `part`) is a retired sink this launch zeroes.
extern "C" int cn_bn_train_fwd_sink(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                                    float* save_scale_shift, const float* part, int slots, float* clear, int64_t clear_n,
                                    int64_t npix, int C, float momentum, float eps, int relu, int dtype, void* stream) {
    CN_CHECK_ARG(x && y && gamma && beta && save_mean && save_invstd && part && npix > 0 && C > 0 && slots > 0 && slots <= BN_MAX_BLOCKS,
                 "cn_bn_train_fwd_sink: bad args");
    CN_CHECK_ARG(clear != part && clear_n >= 0 && clear_n < (1 << 30), "cn_bn_train_fwd_sink: a launch cannot clear the sink it reads");
    int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(C % V == 0 && C % 4 == 0, "cn_bn_train_fwd_sink: C=%d must be a multiple of %d", C, V);
    BnLayout E = ew_layout(npix, C, V, sink_wg_cap());
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(bn_fwd_apply_sink_kernel<T>, dim3(E.nblk, E.ycols), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)x, (const T*)residual, (T*)y, part, slots, gamma, beta, running_mean,
                                                   running_var, save_mean, save_invstd, save_scale_shift, momentum, eps, npix, C, E,
                                                   relu, clear, (int)clear_n));
    CN_LAUNCH_CHECK("cn_bn_train_fwd_sink");
    return CN_OK;
}

extern "C" int cn_scale_shift_act(const void* x, const void* residual, void* y, const float* scale, const float* shift,
                                  int64_t npix, int C, int relu, int dtype, void* stream) {
    CN_CHECK_ARG(x && y && scale && shift && npix > 0 && C > 0, "cn_scale_shift_act: bad args");
    int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(C % V == 0, "cn_scale_shift_act: C=%d must be a multiple of %d", C, V);
    BnLayout E = ew_layout(npix, C, V);
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(scale_shift_act_kernel<T>, dim3(E.nblk, E.ycols), dim3(256), 0,
                                                   (hipStream_t)stream, (const T*)x, (const T*)residual, (T*)y, scale, shift,
                                                   npix, C, E, relu));
    CN_LAUNCH_CHECK("cn_scale_shift_act");
    return CN_OK;
}

extern "C" int cn_bn_train_bwd_acc(const void* dy, const void* x, const void* y, const float* gamma, const float* save_mean,
                                   const float* save_invstd, const float* scale_shift, void* dx, void* dres, const void* dres_acc,
                                   float* dgamma, float* dbeta, int accumulate, int64_t npix, int C, int relu, int dtype, void* ws,
                                   size_t ws_bytes, void* stream) {
    CN_CHECK_ARG(!dres_acc || dres, "cn_bn_train_bwd_acc: dres_acc without dres");
    CN_CHECK_ARG(dy && x && gamma && save_mean && save_invstd && dx && dgamma && dbeta && ws && npix > 0 && C > 0,
                 "cn_bn_train_bwd: bad args");
    CN_CHECK_ARG(!relu || y || scale_shift, "cn_bn_train_bwd: relu needs the forward output or the saved scale/shift");
    int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(C % V == 0, "cn_bn_train_bwd: C=%d must be a multiple of %d", C, V);
    if (ws_bytes < cn_bn_workspace_bytes(npix, C)) { cn_set_error("cn_bn_train_bwd: workspace too small"); return CN_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    BnLayout L = bn_layout(npix, C, V);
    float* part = (float*)ws;
    float* coef = part + (size_t)BN_MAX_BLOCKS * 2 * C;
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_partial_kernel<T, 1>), dim3(L.nblk, L.ycols), dim3(256), 0, st,
                                                   (const T*)x, (const T*)dy, (const T*)y, save_mean, save_invstd, scale_shift, part,
                                                   npix, C, L, relu, 0));
    CN_LAUNCH_CHECK("cn_bn_train_bwd(partial)");
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, part, L.nblk, C, npix, gamma, save_invstd,
                       dgamma, dbeta, coef, accumulate);
    CN_LAUNCH_CHECK("cn_bn_train_bwd(finalize)");
    BnLayout E = ew_layout(npix, C, V);
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false>), dim3(E.nblk, E.ycols), dim3(256), 0, st,
                                                   (const T*)dy, (const T*)x, (const T*)y, save_mean, save_invstd, coef,
                                                   scale_shift, (T*)dx, (T*)dres, (const T*)dres_acc, npix, C, E, relu,
                                                   (const float*)nullptr, 0, (const float*)nullptr, (float*)nullptr, (float*)nullptr, 0,
                                                   (float*)nullptr, 0));
    CN_LAUNCH_CHECK("cn_bn_train_bwd(apply)");
    return CN_OK;
}

// ---- BN backward for consumers that apply it on load (cn_stem_conv_wgrad_bn) ----
// one wave per channel: totals of the sink -> dgamma / dbeta and the coefficients ca | cp | cq | sc | sh (fp32 [5][C]) of
//   g = relu ? (fma(x, sc, sh) > 0 ? dy : 0) : dy,   dx = fma(ca, g, fma(cp, x, cq))          (bn_bwd_apply_kernel's arithmetic)
__global__ __launch_bounds__(256) void bn_bwd_coef_kernel(const float* __restrict__ part, int slots, int C, int64_t npix,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ ss,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                          float* __restrict__ coef, float* __restrict__ clear, int clear_n) {
    if (clear) {
        const int nthr = gridDim.x * 256;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < clear_n; i += nthr) clear[i] = 0.f;
    }
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int b = lane; b < slots; b += 64) { s += (double)part[((int64_t)b * 2) * C + c]; q += (double)part[((int64_t)b * 2 + 1) * C + c]; }
    s = wave_sum_d(s); q = wave_sum_d(q);
    if (lane != 0) return;
    dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;
    dgamma[c] = accumulate ? dgamma[c] + (float)q : (float)q;
    const float is = invstd[c], mu = mean[c];
    const float a = gamma[c] * is, b = (float)(s / (double)npix), cc = (float)(q / (double)npix);
    const float cp = -a * cc * is;
    coef[c] = a;
    coef[C + c] = cp;
    coef[2 * C + c] = -a * b - cp * mu;
    coef[3 * C + c] = ss ? ss[c] : 0.f;
    coef[4 * C + c] = ss ? ss[C + c] : 0.f;
}

// the statistics pass of cn_bn_train_bwd_sink alone: (sum dy', sum dy' * xhat) per channel added to `sink` (fp32 [slots][2][C], all-zero on entry)
extern "C" int cn_bn_bwd_stats(const void* dy, const void* x, const void* y, const float* save_mean, const float* save_invstd,
                               const float* scale_shift, float* sink, int slots, int64_t npix, int C, int relu, int dtype, void* stream) {
    CN_CHECK_ARG(dy && x && save_mean && save_invstd && sink && npix > 0 && C > 0 && slots > 0 && slots <= BN_MAX_BLOCKS, "cn_bn_bwd_stats: bad args");
    CN_CHECK_ARG(!relu || y || scale_shift, "cn_bn_bwd_stats: relu needs the forward output or the saved scale/shift");
    int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(C % V == 0 && C % 4 == 0, "cn_bn_bwd_stats: C=%d must be a multiple of %d", C, V);
    BnLayout L = bn_layout(npix, C, V);
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_partial_kernel<T, 1>), dim3(L.nblk, L.ycols), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)x, (const T*)dy, (const T*)y, save_mean, save_invstd, scale_shift, sink,
                                                   npix, C, L, relu, slots));
    CN_LAUNCH_CHECK("cn_bn_bwd_stats");
    return CN_OK;
}

// sink -> dgamma / dbeta (accumulate != 0: added) + coef fp32 [5][C] for a consumer that applies the BN backward on load; `sink` is left
// as it is (the caller retires it), `clear` (another, retired sink) is zeroed on the way
extern "C" int cn_bn_bwd_coef_sink(const float* sink, int slots, const float* gamma, const float* save_mean, const float* save_invstd,
                                   const float* scale_shift, float* dgamma, float* dbeta, int accumulate, float* coef, float* clear,
                                   int64_t clear_n, int64_t npix, int C, void* stream) {
    CN_CHECK_ARG(sink && gamma && save_mean && save_invstd && dgamma && dbeta && coef && npix > 0 && C > 0 && slots > 0 && slots <= BN_MAX_BLOCKS,
                 "cn_bn_bwd_coef_sink: bad args");
    CN_CHECK_ARG(clear != sink && clear_n >= 0 && clear_n < (1 << 30), "cn_bn_bwd_coef_sink: a launch cannot clear the sink it reads");
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, sink, slots, C, npix, gamma, save_mean,
                       save_invstd, scale_shift, dgamma, dbeta, accumulate, coef, clear, (int)clear_n);
    CN_LAUNCH_CHECK("cn_bn_bwd_coef_sink");
    return CN_OK;
}

// cn_bn_train_bwd_acc as TWO launches: the statistics pass adds its per-workgroup sums to `sink` (fp32 [slots][2][C], all-zero on
// entry, fp32 atomics: the summation order of the batch sums then varies from run to run like the forward sinks') and the apply
// pass reduces it itself (no finalize launch).  `sink` is left as it is (the caller retires it); `clear` as in cn_bn_train_fwd_sink.
extern "C" int cn_bn_train_bwd_sink(const void* dy, const void* x, const void* y, const float* gamma, const float* save_mean,
                                    const float* save_invstd, const float* scale_shift, void* dx, void* dres, const void* dres_acc,
                                    float* dgamma, float* dbeta, int accumulate, float* sink, int slots, float* clear, int64_t clear_n,
                                    int64_t npix, int C, int relu, int dtype, void* stream) {
    CN_CHECK_ARG(!dres_acc || dres, "cn_bn_train_bwd_sink: dres_acc without dres");
    CN_CHECK_ARG(dy && x && gamma && save_mean && save_invstd && dx && dgamma && dbeta && sink && npix > 0 && C > 0 && slots > 0 &&
                     slots <= BN_MAX_BLOCKS, "cn_bn_train_bwd_sink: bad args");
    CN_CHECK_ARG(!relu || y || scale_shift, "cn_bn_train_bwd_sink: relu needs the forward output or the saved scale/shift");
    CN_CHECK_ARG(clear != sink && clear_n >= 0 && clear_n < (1 << 30), "cn_bn_train_bwd_sink: a launch cannot clear the sink it reads");
    int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(C % V == 0 && C % 4 == 0, "cn_bn_train_bwd_sink: C=%d must be a multiple of %d", C, V);
    hipStream_t st = (hipStream_t)stream;
    BnLayout L = bn_layout(npix, C, V);
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_partial_kernel<T, 1>), dim3(L.nblk, L.ycols), dim3(256), 0, st,
                                                   (const T*)x, (const T*)dy, (const T*)y, save_mean, save_invstd, scale_shift, sink,
                                                   npix, C, L, relu, slots));
    CN_LAUNCH_CHECK("cn_bn_train_bwd_sink(partial)");
    BnLayout E = ew_layout(npix, C, V, sink_wg_cap());
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true>), dim3(E.nblk, E.ycols), dim3(256), 0, st,
                                                   (const T*)dy, (const T*)x, (const T*)y, save_mean, save_invstd, (const float*)nullptr,
                                                   scale_shift, (T*)dx, (T*)dres, (const T*)dres_acc, npix, C, E, relu,
                                                   (const float*)sink, slots, gamma, dgamma, dbeta, accumulate, clear, (int)clear_n));
    CN_LAUNCH_CHECK("cn_bn_train_bwd_sink(apply)");
    return CN_OK;
}

// the apply half of cn_bn_train_bwd_sink alone: `sink` already holds the statistics (cn_bn_bwd_stats, or the epilogue of the kernel that
// produced dy: cn_hooks.bnb_part)
extern "C" int cn_bn_train_bwd_apply(const void* dy, const void* x, const void* y, const float* gamma, const float* save_mean,
                                     const float* save_invstd, const float* scale_shift, void* dx, void* dres, const void* dres_acc,
                                     float* dgamma, float* dbeta, int accumulate, const float* sink, int slots, float* clear, int64_t clear_n,
                                     int64_t npix, int C, int relu, int dtype, void* stream) {
    CN_CHECK_ARG(!dres_acc || dres, "cn_bn_train_bwd_apply: dres_acc without dres");
    CN_CHECK_ARG(dy && x && gamma && save_mean && save_invstd && dx && dgamma && dbeta && sink && npix > 0 && C > 0 && slots > 0 &&
                     slots <= BN_MAX_BLOCKS, "cn_bn_train_bwd_apply: bad args");
    CN_CHECK_ARG(!relu || y || scale_shift, "cn_bn_train_bwd_apply: relu needs the forward output or the saved scale/shift");
    CN_CHECK_ARG(clear != sink && clear_n >= 0 && clear_n < (1 << 30), "cn_bn_train_bwd_apply: a launch cannot clear the sink it reads");
    int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(C % V == 0 && C % 4 == 0, "cn_bn_train_bwd_apply: C=%d must be a multiple of %d", C, V);
    BnLayout E = ew_layout(npix, C, V, sink_wg_cap());
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true>), dim3(E.nblk, E.ycols), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)dy, (const T*)x, (const T*)y, save_mean, save_invstd, (const float*)nullptr,
                                                   scale_shift, (T*)dx, (T*)dres, (const T*)dres_acc, npix, C, E, relu,
                                                   sink, slots, gamma, dgamma, dbeta, accumulate, clear, (int)clear_n));
    CN_LAUNCH_CHECK("cn_bn_train_bwd_apply");
    return CN_OK;
}

extern "C" int cn_bn_train_bwd(const void* dy, const void* x, const void* y, const float* gamma, const float* save_mean,
                               const float* save_invstd, const float* scale_shift, void* dx, void* dres, float* dgamma,
                               float* dbeta, int accumulate, int64_t npix, int C, int relu, int dtype, void* ws, size_t ws_bytes,
                               void* stream) {
    return cn_bn_train_bwd_acc(dy, x, y, gamma, save_mean, save_invstd, scale_shift, dx, dres, nullptr, dgamma, dbeta, accumulate,
                               npix, C, relu, dtype, ws, ws_bytes, stream);
}

extern "C" int cn_relu_bwd(const void* dy, const void* y, void* dx, int64_t n, int dtype, void* stream) {
    CN_CHECK_ARG(dy && y && dx && n > 0, "cn_relu_bwd: bad args");
    int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(n % V == 0, "cn_relu_bwd: n must be a multiple of %d", V);
    int64_t nvec = n / V;
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(relu_bwd_kernel<T>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream,
                                                   (const T*)dy, (const T*)y, (T*)dx, nvec));
    CN_LAUNCH_CHECK("cn_relu_bwd");
    return CN_OK;
}
