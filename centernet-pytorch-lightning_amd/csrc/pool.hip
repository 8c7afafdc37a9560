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

// ---- depthwise up-conv weight gradient, bilinear x2 layers (k = 4, stride 2, pad 1, bf16) — round 4 ---------------------------------
// The kernel above gives every (tap, channel vector) pair its own lane: per input pixel and channel vector 16 loads of x (one line, 16
// lanes) and 16 of dy, each dy pixel fetched by the 4 taps that touch it — 1.4 TB/s of algorithmic traffic, the largest single excess
// over its roofline left in the step (profiles/r04_ops_by_shape.txt).  Here a lane owns a channel vector and walks along ONE dy row oh:
// that row only meets the two kernel rows kh = kh0, kh0 + 2 (kh0 = (oh + 1) & 1) through the x rows ih = (oh + 1 - kh0) / 2 and ih - 1,
// and its four kernel columns through a sliding window of four dy pixels (ow = 2 iw - 1 .. 2 iw + 2: two new pixels per x pixel).
// Every dy pixel is loaded exactly once, every x pixel twice; 64 fp32 accumulators per lane (2 kernel rows x 4 columns x 8 channels).
// Lanes of a wave = CVW channel vectors x 64 / CVW rows; a lane keeps the PARITY of its rows (even task stride), so its accumulators
// always mean the same taps; the workgroup folds them through LDS atomics and adds 16 C values to dw.
template <int CVW>
__global__ __launch_bounds__(256) void dwdeconv_wgrad_rows_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ ws,
                                                                  int N, int H, int W, int C, int OH, int OW, int ntasks) {
    constexpr int TPW = 64 / CVW;                 // dy rows per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cvl = lane % CVW, tslot = lane / CVW;
    const int cv = blockIdx.y * CVW + cvl;        // channel vector (8 channels)
    const int slot = (blockIdx.x * 4 + wave) * TPW + tslot, nslots = gridDim.x * 4 * TPW;      // nslots is even: a lane's rows keep their parity
    float acc[2][4][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[a][b][e] = 0.f;
    auto unpack = [](const uint4& v, float (&f)[8]) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { f[2 * q] = __uint_as_float(w[q] << 16); f[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
    };
    for (int task = slot; task < ntasks; task += nslots) {
        const int n = task / OH, oh = task - n * OH;
        const int kh0 = (oh + 1) & 1, iha = (oh + 1 - kh0) >> 1, ihb = iha - 1;
        const bool va = iha < H, vb = ihb >= 0;
        const bf16_t* dr = dy + ((int64_t)(n * OH + oh) * OW) * C + cv * 8;
        const bf16_t* xa = x + ((int64_t)(n * H + (va ? iha : 0)) * W) * C + cv * 8;
        const bf16_t* xb = x + ((int64_t)(n * H + (vb ? ihb : 0)) * W) * C + cv * 8;
        // dy window ow = 2 iw - 1 .. 2 iw + 2 as raw bf16 vectors; the loads of step iw + 4 are issued in step iw (a ring of four
        // steps x {x row a, x row b, two new dy pixels} = 16 vectors in flight per lane: with one step of lead the walk was latency
        // bound — 142 us on the 64-channel 64^2 layer, slower than the tap-per-lane kernel it replaces)
        uint4 w0 = make_uint4(0u, 0u, 0u, 0u), w1 = *reinterpret_cast<const uint4*>(dr), w2 = *reinterpret_cast<const uint4*>(dr + C),
              w3 = ldg16_masked(dr, (int64_t)2 * C * 2, 2 < OW);
        uint4 rxa[4], rxb[4], rd0[4], rd1[4];
        auto issue = [&](int iw, uint4& qa, uint4& qb, uint4& q0, uint4& q1) {
            const bool in = iw < W;
            qa = ldg16_masked(xa, (int64_t)iw * C * 2, va && in);
            qb = ldg16_masked(xb, (int64_t)iw * C * 2, vb && in);
            q0 = ldg16_masked(dr, (int64_t)(2 * iw + 3) * C * 2, 2 * iw + 3 < OW);
            q1 = ldg16_masked(dr, (int64_t)(2 * iw + 4) * C * 2, 2 * iw + 4 < OW);
        };
#pragma unroll
        for (int d = 0; d < 4; ++d) issue(d, rxa[d], rxb[d], rd0[d], rd1[d]);
        for (int iw0 = 0; iw0 < W; iw0 += 4) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint4 cxa = rxa[d], cxb = rxb[d], cd0 = rd0[d], cd1 = rd1[d];
                issue(iw0 + d + 4, rxa[d], rxb[d], rd0[d], rd1[d]);
                float fa[8], fb[8], fw[8];
                unpack(cxa, fa);                  // (x pixels past the row end were loaded as zeros: steps iw >= W add nothing)
                unpack(cxb, fb);
                const uint4 wv[4] = {w0, w1, w2, w3};
#pragma unroll
                for (int kw = 0; kw < 4; ++kw) {
                    unpack(wv[kw], fw);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        acc[0][kw][e] = fmaf(fa[e], fw[e], acc[0][kw][e]);
                        acc[1][kw][e] = fmaf(fb[e], fw[e], acc[1][kw][e]);
                    }
                }
                w0 = w2; w1 = w3; w2 = cd0; w3 = cd1;     // slide: ow = 2 (iw + 1) - 1 .. 2 (iw + 1) + 2
            }
        }
    }
    // fold the workgroup's lanes: red[parity of the lane's rows][kernel-row slot][kw][channel of this block]
    __shared__ float red[2 * 2 * 4 * CVW * 8];
    for (int i = threadIdx.x; i < 2 * 2 * 4 * CVW * 8; i += 256) red[i] = 0.f;
    __syncthreads();
    const int par = slot & 1;                     // parity of oh (OH and nslots are even)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int kw = 0; kw < 4; ++kw)
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(&red[((par * 2 + a) * 4 + kw) * CVW * 8 + cvl * 8 + e], acc[a][kw][e]);
    __syncthreads();
    // the workgroup's 16 x (CVW * 8) sums leave as ONE private slab (plain 4-byte stores, coalesced), folded by dwdeconv_wgrad_reduce_kernel:
    // same-address fp32 atomics execute at the memory side one after the other (~0.3 us each: 160 workgroups = a 50 us chain per address,
    // the whole floor of the first version of this kernel)
    float* slab = ws + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * (2 * 2 * 4 * CVW * 8);
    for (int i = threadIdx.x; i < 2 * 2 * 4 * CVW * 8; i += 256) slab[i] = red[i];
}

// dw[c][kh][kw] += sum over the nx row-block slabs; slab layout [x block][channel block][row parity][kernel-row slot][kw][channel].
// 64 outputs per workgroup, four threads per output each folding every fourth slab with eight loads in flight (one thread per
// output walking 160 slabs one load after the other was 30 us of pure latency: the floor of every small layer)
__global__ __launch_bounds__(256) void dwdeconv_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int C, int cvw8, int nx) {
    __shared__ float part[4][64];
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;                       // (c, kh, kw)
    float s = 0.f;
    if (i < C * 16) {
        const int kw = i & 3, kh = (i >> 2) & 3, c = i >> 4;
        const int yb = c / cvw8, ch = c - yb * cvw8, ny = C / cvw8;
        const int pr = (kh + 1) & 1, a = kh >> 1;            // rows of parity pr meet kernel rows kh0 = (pr + 1) & 1 and kh0 + 2
        const float* p = ws + (int64_t)yb * (16 * cvw8) + ((pr * 2 + a) * 4 + kw) * cvw8 + ch;
        const int64_t pitch = (int64_t)ny * 16 * cvw8;
        int b = q;
        for (; b + 28 < nx; b += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(b + 4 * u) * pitch];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nx; b += 4) s += p[b * pitch];
    }
    part[q][o] = s;
    __syncthreads();
    if (q == 0 && i < C * 16) dw[i] += (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);
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

// the x2 layers (k = 4, stride 2, pad 1, OH = 2 H, OW = 2 W, bf16, C in {64, 128, 256} or a multiple of 512) through the row-walking
// kernel + slab reduction; ws: cn_dwdeconv_wgrad_ws_bytes(N, OH, C) bytes of scratch.  CN_EUNSUPPORTED for any other shape.
static int dw_rows_grid(int N, int OH, int C, int* cvw_out, int target) {
    const int CV = C / 8, cvw = CV >= 64 ? 64 : CV, tpw = 64 / cvw;
    int gx = target / (CV / cvw);
    const int need = (N * OH + 4 * tpw - 1) / (4 * tpw);
    if (gx > need) gx = need;
    if (gx > 512) gx = 512;
    if (gx < 1) gx = 1;
    *cvw_out = cvw;
    return gx;
}
static bool dw_rows_ok(int C, int k, int stride, int pad, int H, int W, int OH, int OW, int dtype) {
    static const bool no_rows = getenv("CN_DISABLE_DWDECONV_ROWS") != nullptr;
    return !no_rows && dtype == CN_BF16 && k == 4 && stride == 2 && pad == 1 && OH == 2 * H && OW == 2 * W && OW >= 3 &&
           (C == 64 || C == 128 || C == 256 || (C > 0 && (C & 511) == 0));
}
extern "C" size_t cn_dwdeconv_wgrad_ws_bytes(int N, int OH, int C) {
    if (C <= 0 || (C & 63)) return 0;
    return (size_t)512 * 16 * C * sizeof(float);            // the grid is capped at 512 row blocks
}
extern "C" int cn_dwdeconv_bwd_weight_rows_h(const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, int N, int H, int W, int C, int k,
                                             int stride, int pad, int OH, int OW, int dtype, cn_hooks* hooks, void* stream) {
    CN_CHECK_ARG(x && dy && dw && ws && N > 0 && H > 0 && W > 0, "cn_dwdeconv_bwd_weight_rows: bad args");
    if (!dw_rows_ok(C, k, stride, pad, H, W, OH, OW, dtype) || (int64_t)N * OH >= (1ll << 30))
        CN_UNSUPPORTED("cn_dwdeconv_bwd_weight_rows: bf16 k=4 s=2 p=1 layers with 64/128/256/512n channels only");
    int cvw;
    const int gx = dw_rows_grid(N, OH, C, &cvw, cn_wgrad_target(hooks));
    const int CV = C / 8, ntasks = N * OH;
    if (ws_bytes < (size_t)gx * 16 * C * sizeof(float)) { cn_set_error("cn_dwdeconv_bwd_weight_rows: workspace too small"); return CN_EWORKSPACE; }
    dim3 grid(gx, CV / cvw);
    const bf16_t* xp = (const bf16_t*)x; const bf16_t* dp = (const bf16_t*)dy;
    hipStream_t st = (hipStream_t)stream;
    float* wsf = (float*)ws;
    if (cvw == 8) hipLaunchKernelGGL(dwdeconv_wgrad_rows_kernel<8>, grid, dim3(256), 0, st, xp, dp, wsf, N, H, W, C, OH, OW, ntasks);
    else if (cvw == 16) hipLaunchKernelGGL(dwdeconv_wgrad_rows_kernel<16>, grid, dim3(256), 0, st, xp, dp, wsf, N, H, W, C, OH, OW, ntasks);
    else if (cvw == 32) hipLaunchKernelGGL(dwdeconv_wgrad_rows_kernel<32>, grid, dim3(256), 0, st, xp, dp, wsf, N, H, W, C, OH, OW, ntasks);
    else hipLaunchKernelGGL(dwdeconv_wgrad_rows_kernel<64>, grid, dim3(256), 0, st, xp, dp, wsf, N, H, W, C, OH, OW, ntasks);
    hipLaunchKernelGGL(dwdeconv_wgrad_reduce_kernel, dim3((C * 16 + 63) / 64), dim3(256), 0, st, wsf, dw, C, cvw * 8, gx);
    CN_LAUNCH_CHECK("cn_dwdeconv_bwd_weight_rows");
    return CN_OK;
}
extern "C" int cn_dwdeconv_bwd_weight_rows(const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, int N, int H, int W, int C, int k,
                                           int stride, int pad, int OH, int OW, int dtype, void* stream) {
    return cn_dwdeconv_bwd_weight_rows_h(x, dy, dw, ws, ws_bytes, N, H, W, C, k, stride, pad, OH, OW, dtype, nullptr, stream);
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
