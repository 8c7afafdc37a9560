// DCNv2 data gradient without materialising the 9x-wide column gradient:
//   dx[q][ci] = sum_k sum_co W[co][ci][k] * G_k[q][co],    G_k[q][co] = sum_p mask[p,k] * hat(py-qy) * hat(px-qx) * dY[p][co]
// (the adjoint of the bilinear sampling applied to dY, then an ordinary 9-tap contraction — linearity lets the
// contraction move outside the scatter).  Same skeleton as the 3x3 halo-tile conv: a workgroup owns an 8x16 tile of
// destination pixels and BN output channels; per tap
//   1. every source pixel within R=3 of the tile computes its sample position (2 fp32 offsets) and appends
//      (source, weight*mask) to the hit list of each of the <=4 destination pixels it touches inside the tile (LDS);
//   2. 8..16 lanes per destination pixel walk its list (typically ~4 entries): 16-byte dY loads + FMAs in registers,
//      and write the bf16 G tile [128][CK] to LDS;
//   3. MFMA against the tap's weight slice (register-prefetched one step ahead, staged through LDS).
// Samples displaced by more than R pixels are not seen here: cn_dcn_bwd_dom scatters those into dx_far, which this
// kernel adds in its epilogue (fp32 residual).  No atomics on HBM, dcol is never written.
#include "conv_common.h"

#define DX_TH 8
#define DX_TW 16
#define DX_R 3
#define DX_MAXH 8
#define DX_OVF 256

template <typename T, int BN, int CK>
__global__ __launch_bounds__(256) void dcn_bwd_dx_kernel(const ConvGeom g) {
    constexpr int BM = DX_TH * DX_TW;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int PITCH = CK + Mma<T>::PAD;
    constexpr int VPR = CK / VEC;
    constexpr int B_VECS = BN * VPR;
    constexpr int B_PASS = (B_VECS + 255) / 256;
    constexpr int WGN = (BN >= 64) ? 2 : 1;
    constexpr int WGM = 4 / WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MI = WM / 32, NJ = WN / 32;
    constexpr int KSTEPS = CK / Mma<T>::KSTEP;
    constexpr int SH = DX_TH + 2 * DX_R, SW = DX_TW + 2 * DX_R;   // source window
    constexpr int ITEMS = BM * VPR;                                // (destination pixel, channel vector) work items
    constexpr int I_PASS = (ITEMS + 255) / 256;

    __shared__ __attribute__((aligned(16))) T lds[(BM + BN) * PITCH];   // G tile | weight slice (rewritten behind the trailing barrier)
    __shared__ int hit_p[BM][DX_MAXH];
    __shared__ float hit_w[BM][DX_MAXH];
    __shared__ int hit_n[BM];
    __shared__ int ovf_q[DX_OVF], ovf_p[DX_OVF];
    __shared__ float ovf_w[DX_OVF];
    __shared__ int ovf_n;
    T* const Gs = lds;
    T* const Bs = lds + BM * PITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_w = (g.W + DX_TW - 1) / DX_TW;
    const int th0 = (blockIdx.x / tiles_w) * DX_TH, tw0 = (blockIdx.x % tiles_w) * DX_TW;
    const int n0 = blockIdx.y * BN;
    const int n = blockIdx.z;
    const int wm = (wave / WGN) * WM, wn = (wave % WGN) * WN;
    const int64_t img = (int64_t)n * g.H * g.W;
    const T* __restrict__ DY = reinterpret_cast<const T*>(g.x) + img * g.x_ld;
    const T* __restrict__ Wp = reinterpret_cast<const T*>(g.w);
    const float* __restrict__ OM = g.dcn_om + img * g.dcn_omld;

    f32x16_t acc[NJ][MI];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    uint4 rb[B_PASS];
    auto bload = [&](int tap, int c0) {
        const int wofs = tap * g.Ci + c0;
#pragma unroll
        for (int p = 0; p < B_PASS; ++p) {
            const int v = tid + p * 256;
            const int row = v / VPR, col = (v % VPR) * VEC;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (v < B_VECS && n0 + row < g.co_pad) val = *reinterpret_cast<const uint4*>(Wp + (int64_t)(n0 + row) * g.ktot + wofs + col);
            rb[p] = val;
        }
    };
    auto bstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < B_PASS; ++p) {
            const int v = tid + p * 256;
            if (v < B_VECS) lds_store_vec<T, PITCH>(Bs + buf * BN * PITCH, v / VPR, (v % VPR) * VEC, rb[p]);
        }
    };

    const int nchunks = g.Ci / CK;          // g.Ci = padded Co of the forward layer = contraction length per tap
    bload(0, 0);
    int step = 0;                            // (tap, chunk) steps; weight slice `step` lives in buffer step & 1
    for (int tap = 0; tap < 9; ++tap) {
        // ---- 1. hit lists of this tap ----
        for (int i = tid; i < BM; i += 256) hit_n[i] = 0;
        if (tid == 0) ovf_n = 0;
        __syncthreads();
        for (int s = tid; s < SH * SW; s += 256) {
            const int sy = th0 - DX_R + s / SW, sx = tw0 - DX_R + s % SW;
            if ((unsigned)sy >= (unsigned)g.H || (unsigned)sx >= (unsigned)g.W) continue;
            const int sp = sy * g.W + sx;
            const float* o = OM + (int64_t)sp * g.dcn_omld;
            const float py = (float)(sy - 1 + tap / 3) + o[2 * tap];
            const float px = (float)(sx - 1 + tap % 3) + o[2 * tap + 1];
            const int y0 = (int)floorf(py), x0 = (int)floorf(px);
            const float ly = py - (float)y0, lx = px - (float)x0;
            float m = -1.f;
#pragma unroll
            for (int cnr = 0; cnr < 4; ++cnr) {
                const int qy = y0 + (cnr >> 1), qx = x0 + (cnr & 1);
                const int ty = qy - th0, tx = qx - tw0;
                if ((unsigned)ty >= (unsigned)DX_TH || (unsigned)tx >= (unsigned)DX_TW || qy >= g.H || qx >= g.W) continue;
                if (qy - sy > DX_R || sy - qy > DX_R || qx - sx > DX_R || sx - qx > DX_R) continue;   // far: handled by dx_far
                const float wgt = ((cnr >> 1) ? ly : 1.f - ly) * ((cnr & 1) ? lx : 1.f - lx);
                if (!(wgt > 0.f)) continue;
                if (m < 0.f) m = sigmoidf_(o[18 + tap]);
                const int ql = ty * DX_TW + tx;
                const int slot = atomicAdd(&hit_n[ql], 1);
                if (slot < DX_MAXH) { hit_p[ql][slot] = sp; hit_w[ql][slot] = wgt * m; }
                else {
                    const int e = atomicAdd(&ovf_n, 1);
                    if (e < DX_OVF) { ovf_q[e] = ql; ovf_p[e] = sp; ovf_w[e] = wgt * m; }
                }
            }
        }
        __syncthreads();
        for (int ch = 0; ch < nchunks; ++ch, ++step) {
            const int c0 = ch * CK;
            // ---- 2. G tile of this (tap, channel slice) ----
#pragma unroll
            for (int ip = 0; ip < I_PASS; ++ip) {
                const int it = tid + ip * 256;
                if (it < ITEMS) {
                    const int ql = it / VPR, col = (it % VPR) * VEC;
                    float a[VEC];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) a[j] = 0.f;
                    int nh = hit_n[ql];
                    nh = nh < DX_MAXH ? nh : DX_MAXH;
                    const T* src = DY + c0 + col;
                    for (int h0 = 0; h0 < nh; h0 += 4) {     // 4 independent 16-byte gathers in flight per item
                        float v[4][VEC], wg[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool on = h0 + u < nh;
                            wg[u] = on ? hit_w[ql][h0 + u] : 0.f;
                            Vec16<T>::load(src + (int64_t)hit_p[ql][on ? h0 + u : 0] * g.x_ld, v[u]);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int j = 0; j < VEC; ++j) a[j] = fmaf(v[u][j], wg[u], a[j]);
                    }
                    const int no = ovf_n < DX_OVF ? ovf_n : DX_OVF;
                    for (int e = 0; e < no; ++e) {
                        if (ovf_q[e] != ql) continue;
                        float v[VEC];
                        Vec16<T>::load(DY + (int64_t)ovf_p[e] * g.x_ld + c0 + col, v);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) a[j] = fmaf(v[j], ovf_w[e], a[j]);
                    }
                    if constexpr (sizeof(T) == 2) {
                        uint4 o;
                        o.x = (uint32_t)f2bf(a[0]) | ((uint32_t)f2bf(a[1]) << 16); o.y = (uint32_t)f2bf(a[2]) | ((uint32_t)f2bf(a[3]) << 16);
                        o.z = (uint32_t)f2bf(a[4]) | ((uint32_t)f2bf(a[5]) << 16); o.w = (uint32_t)f2bf(a[6]) | ((uint32_t)f2bf(a[7]) << 16);
                        lds_store_vec<T, PITCH>(Gs, ql, col, o);
                    } else {
                        lds_store_vec<T, PITCH>(Gs, ql, col, make_uint4(__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3])));
                    }
                }
            }
            bstore(0);
            __syncthreads();
            // prefetch the next weight slice while this one is multiplied
            {
                const int nt = (ch + 1 < nchunks) ? tap : tap + 1, nc = (ch + 1 < nchunks) ? ch + 1 : 0;
                if (nt < 9) bload(nt, nc * CK);
            }
            const T* bt = Bs;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                typename Mma<T>::Frag fa[MI], fb[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i] = Mma<T>::load(Gs, PITCH, wm + i * 32, kk, lane);
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = Mma<T>::load(bt, PITCH, wn + j * 32, kk, lane);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i) acc[j][i] = Mma<T>::mma(fb[j], fa[i], acc[j][i]);
            }
            __syncthreads();             // G tile and hit lists may be overwritten
        }
    }

    int64_t pix[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = wm + i * 32 + (lane & 31);
        const int oh = th0 + m / DX_TW, ow = tw0 + m % DX_TW;
        pix[i] = (oh < g.H && ow < g.W) ? img + (int64_t)oh * g.W + ow : -1;
    }
    conv_epilogue<T, MI, NJ>(g, acc, pix, n0 + wn, lane);
}

template <typename T, int BN, int CK>
static void launch_dx(const ConvGeom& g, hipStream_t st) {
    dim3 grid(((g.H + DX_TH - 1) / DX_TH) * ((g.W + DX_TW - 1) / DX_TW), (g.Co + BN - 1) / BN, g.N);
    hipLaunchKernelGGL((dcn_bwd_dx_kernel<T, BN, CK>), grid, dim3(256), 0, st, g);
}

void dcn_bwd_dx_launch(const ConvGeom& g, int dtype, hipStream_t st) {
    const int bn = g.Co % 128 == 0 ? 128 : (g.Co % 64 == 0 ? 64 : 32);
    if (dtype == CN_BF16) {
        if (g.Ci % 64 == 0) { if (bn == 128) launch_dx<bf16_t, 128, 64>(g, st); else if (bn == 64) launch_dx<bf16_t, 64, 64>(g, st); else launch_dx<bf16_t, 32, 64>(g, st); }
        else if (g.Ci % 32 == 0) { if (bn == 128) launch_dx<bf16_t, 128, 32>(g, st); else if (bn == 64) launch_dx<bf16_t, 64, 32>(g, st); else launch_dx<bf16_t, 32, 32>(g, st); }
        else { if (bn == 128) launch_dx<bf16_t, 128, 16>(g, st); else if (bn == 64) launch_dx<bf16_t, 64, 16>(g, st); else launch_dx<bf16_t, 32, 16>(g, st); }
    } else {
        if (bn == 128) launch_dx<float, 128, 16>(g, st); else if (bn == 64) launch_dx<float, 64, 16>(g, st); else launch_dx<float, 32, 16>(g, st);
    }
}


// ================================================================================================ forward
// y[p][co] = bias[co] + sum_k sum_ci W[co][ci][k] * mask[p,k] * bilinear(x[:, :, ci], pos(p,k))
// Fused: the sampled operand is built tile by tile in LDS and fed straight to the MFMAs — the 9x-wide column tensor of
// the im2col formulation (1.2 GB per 64-channel 128x128 layer at batch 64) is never written or read.  Per tap:
//   1. 128 lanes compute the bilinear geometry of their pixel (fp32 offsets) -> clamped corner indices + weights*mask in LDS;
//   2. every (pixel, 16-byte channel vector) item blends four unconditional 16-byte corner loads and stores the A tile;
//   3. MFMA against the tap's weight slice (register-prefetched one step ahead).
template <typename T, int BN, int CK>
__global__ __launch_bounds__(256) void dcn_fwd_kernel(const ConvGeom g) {
    constexpr int BM = DX_TH * DX_TW;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int PITCH = CK + Mma<T>::PAD;
    constexpr int VPR = CK / VEC;
    constexpr int B_VECS = BN * VPR;
    constexpr int B_PASS = (B_VECS + 255) / 256;
    constexpr int WGN = (BN >= 64) ? 2 : 1;
    constexpr int WGM = 4 / WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MI = WM / 32, NJ = WN / 32;
    constexpr int KSTEPS = CK / Mma<T>::KSTEP;
    constexpr int ITEMS = BM * VPR;
    constexpr int I_PASS = (ITEMS + 255) / 256;

    __shared__ __attribute__((aligned(16))) T lds[(BM + BN) * PITCH];
    __shared__ int s_idx[4][BM];
    __shared__ float s_w[4][BM];
    T* const As = lds;
    T* const Bs = lds + BM * PITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_w = (g.W + DX_TW - 1) / DX_TW;
    const int th0 = (blockIdx.x / tiles_w) * DX_TH, tw0 = (blockIdx.x % tiles_w) * DX_TW;
    const int n0 = blockIdx.y * BN;
    const int n = blockIdx.z;
    const int wm = (wave / WGN) * WM, wn = (wave % WGN) * WN;
    const int64_t img = (int64_t)n * g.H * g.W;
    const T* __restrict__ X = reinterpret_cast<const T*>(g.x) + img * g.x_ld;
    const T* __restrict__ Wp = reinterpret_cast<const T*>(g.w);
    const float* __restrict__ OM = g.dcn_om + img * g.dcn_omld;

    f32x16_t acc[NJ][MI];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    uint4 rb[B_PASS];
    auto bload = [&](int tap, int c0) {
        const int wofs = tap * g.Ci + c0;
#pragma unroll
        for (int p = 0; p < B_PASS; ++p) {
            const int v = tid + p * 256;
            const int row = v / VPR, col = (v % VPR) * VEC;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (v < B_VECS && n0 + row < g.co_pad) val = *reinterpret_cast<const uint4*>(Wp + (int64_t)(n0 + row) * g.ktot + wofs + col);
            rb[p] = val;
        }
    };
    auto bstore = [&]() {
#pragma unroll
        for (int p = 0; p < B_PASS; ++p) {
            const int v = tid + p * 256;
            if (v < B_VECS) lds_store_vec<T, PITCH>(Bs, v / VPR, (v % VPR) * VEC, rb[p]);
        }
    };

    const int nchunks = g.Ci / CK;
    bload(0, 0);
    for (int tap = 0; tap < 9; ++tap) {
        if (tid < BM) {
            const int h = th0 + tid / DX_TW, w = tw0 + tid % DX_TW;
            int i0 = 0, i1 = 0, i2 = 0, i3 = 0;
            float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
            if (h < g.H && w < g.W) {
                const float* o = OM + ((int64_t)h * g.W + w) * g.dcn_omld;
                const float py = (float)(h - 1 + tap / 3) + o[2 * tap];
                const float px = (float)(w - 1 + tap % 3) + o[2 * tap + 1];
                const float m = sigmoidf_(o[18 + tap]);
                const Tap t = make_tap(py, px, g.H, g.W);
                const int hc0 = min(max(t.h0, 0), g.H - 1), hc1 = min(max(t.h0 + 1, 0), g.H - 1);
                const int wc0 = min(max(t.w0, 0), g.W - 1), wc1 = min(max(t.w0 + 1, 0), g.W - 1);
                i0 = hc0 * g.W + wc0; i1 = hc0 * g.W + wc1; i2 = hc1 * g.W + wc0; i3 = hc1 * g.W + wc1;
                w0 = t.w00 * m; w1 = t.w01 * m; w2 = t.w10 * m; w3 = t.w11 * m;     // zero for out-of-image corners
            }
            s_idx[0][tid] = i0; s_idx[1][tid] = i1; s_idx[2][tid] = i2; s_idx[3][tid] = i3;
            s_w[0][tid] = w0; s_w[1][tid] = w1; s_w[2][tid] = w2; s_w[3][tid] = w3;
        }
        __syncthreads();
        for (int ch = 0; ch < nchunks; ++ch) {
            const int c0 = ch * CK;
#pragma unroll
            for (int ip = 0; ip < I_PASS; ++ip) {
                const int it = tid + ip * 256;
                if (it < ITEMS) {
                    const int pl = it / VPR, col = (it % VPR) * VEC;
                    const T* src = X + c0 + col;
                    float v0[VEC], v1[VEC], v2[VEC], v3[VEC], a[VEC];
                    Vec16<T>::load(src + (int64_t)s_idx[0][pl] * g.x_ld, v0);
                    Vec16<T>::load(src + (int64_t)s_idx[1][pl] * g.x_ld, v1);
                    Vec16<T>::load(src + (int64_t)s_idx[2][pl] * g.x_ld, v2);
                    Vec16<T>::load(src + (int64_t)s_idx[3][pl] * g.x_ld, v3);
                    const float w0 = s_w[0][pl], w1 = s_w[1][pl], w2 = s_w[2][pl], w3 = s_w[3][pl];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) a[j] = v0[j] * w0 + v1[j] * w1 + v2[j] * w2 + v3[j] * w3;
                    if constexpr (sizeof(T) == 2) {
                        uint4 o;
                        o.x = (uint32_t)f2bf(a[0]) | ((uint32_t)f2bf(a[1]) << 16); o.y = (uint32_t)f2bf(a[2]) | ((uint32_t)f2bf(a[3]) << 16);
                        o.z = (uint32_t)f2bf(a[4]) | ((uint32_t)f2bf(a[5]) << 16); o.w = (uint32_t)f2bf(a[6]) | ((uint32_t)f2bf(a[7]) << 16);
                        lds_store_vec<T, PITCH>(As, pl, col, o);
                    } else {
                        lds_store_vec<T, PITCH>(As, pl, col, make_uint4(__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3])));
                    }
                }
            }
            bstore();
            __syncthreads();
            {
                const int nt = (ch + 1 < nchunks) ? tap : tap + 1, nc = (ch + 1 < nchunks) ? ch + 1 : 0;
                if (nt < 9) bload(nt, nc * CK);
            }
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                typename Mma<T>::Frag fa[MI], fb[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i] = Mma<T>::load(As, PITCH, wm + i * 32, kk, lane);
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = Mma<T>::load(Bs, PITCH, wn + j * 32, kk, lane);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i) acc[j][i] = Mma<T>::mma(fb[j], fa[i], acc[j][i]);
            }
            __syncthreads();
        }
    }

    int64_t pix[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = wm + i * 32 + (lane & 31);
        const int oh = th0 + m / DX_TW, ow = tw0 + m % DX_TW;
        pix[i] = (oh < g.H && ow < g.W) ? img + (int64_t)oh * g.W + ow : -1;
    }
    conv_epilogue<T, MI, NJ>(g, acc, pix, n0 + wn, lane);
}

template <typename T, int BN, int CK>
static void launch_fwd(const ConvGeom& g, hipStream_t st) {
    dim3 grid(((g.H + DX_TH - 1) / DX_TH) * ((g.W + DX_TW - 1) / DX_TW), (g.Co + BN - 1) / BN, g.N);
    hipLaunchKernelGGL((dcn_fwd_kernel<T, BN, CK>), grid, dim3(256), 0, st, g);
}

void dcn_fwd_launch(const ConvGeom& g, int dtype, hipStream_t st) {
    const int co32 = (g.Co + 31) / 32 * 32;
    int bn = 32, bw = co32;
    for (int c : {64, 128}) {
        int w = (co32 + c - 1) / c * c;
        if (w <= bw) { bn = c; bw = w; }
    }
    if (dtype == CN_BF16) {
        if (g.Ci % 64 == 0) { if (bn == 128) launch_fwd<bf16_t, 128, 64>(g, st); else if (bn == 64) launch_fwd<bf16_t, 64, 64>(g, st); else launch_fwd<bf16_t, 32, 64>(g, st); }
        else if (g.Ci % 32 == 0) { if (bn == 128) launch_fwd<bf16_t, 128, 32>(g, st); else if (bn == 64) launch_fwd<bf16_t, 64, 32>(g, st); else launch_fwd<bf16_t, 32, 32>(g, st); }
        else { if (bn == 128) launch_fwd<bf16_t, 128, 16>(g, st); else if (bn == 64) launch_fwd<bf16_t, 64, 16>(g, st); else launch_fwd<bf16_t, 32, 16>(g, st); }
    } else {
        if (bn == 128) launch_fwd<float, 128, 16>(g, st); else if (bn == 64) launch_fwd<float, 64, 16>(g, st); else launch_fwd<float, 32, 16>(g, st);
    }
}
