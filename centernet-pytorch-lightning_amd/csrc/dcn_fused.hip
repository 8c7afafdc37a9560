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
#include <stdlib.h>

#define DX_TH 8
#define DX_TW 16
#define DX_R 3
#define DX_MAXH 8
#define DX_OVF 256

#ifdef DX_PROBE   // development build only (tools/dx_probe.py): cycle stamps of wave 0 at the phase boundaries of every tap
__device__ unsigned long long dx_ts[1024 * 9 * 4];
#define DX_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.z == 0 && blockIdx.y == 0 && blockIdx.x < 1024) dx_ts[(blockIdx.x * 9 + tap) * 4 + (k)] = clock64(); } while (0)
extern "C" int dx_probe_dump(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(dx_ts), sizeof(dx_ts)); }
#else
#define DX_STAMP(k) do { } while (0)
#endif

template <typename T, int BN, int CK>
__global__ __launch_bounds__(256) void dcn_bwd_dx_kernel(const ConvGeom g) {
    CN_MAIN_PRIO_SET();
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
    // the offsets of the next tap's sources are fetched one tap ahead (no exposed latency in the list-building phase)
    constexpr int NSRC = (SH * SW + 255) / 256;
    float so[NSRC][3];
    auto oload = [&](int tap) {
#pragma unroll
        for (int si = 0; si < NSRC; ++si) {
            const int s = tid + si * 256;
            const int sy = th0 - DX_R + s / SW, sx = tw0 - DX_R + s % SW;
            so[si][0] = so[si][1] = so[si][2] = 0.f;
            if (s < SH * SW && (unsigned)sy < (unsigned)g.H && (unsigned)sx < (unsigned)g.W) {
                const float* o = OM + ((int64_t)sy * g.W + sx) * g.dcn_omld;
                so[si][0] = o[2 * tap]; so[si][1] = o[2 * tap + 1]; so[si][2] = o[18 + tap];
            }
        }
    };
    oload(0);
    int step = 0;                            // (tap, chunk) steps; weight slice `step` lives in buffer step & 1
    for (int tap = 0; tap < 9; ++tap) {
        // ---- 1. hit lists of this tap ----
        DX_STAMP(0);
        for (int i = tid; i < BM; i += 256) hit_n[i] = 0;
        if (tid == 0) ovf_n = 0;
        __syncthreads();
#pragma unroll
        for (int si = 0; si < NSRC; ++si) {
            const int s = tid + si * 256;
            if (s >= SH * SW) continue;
            const int sy = th0 - DX_R + s / SW, sx = tw0 - DX_R + s % SW;
            if ((unsigned)sy >= (unsigned)g.H || (unsigned)sx >= (unsigned)g.W) continue;
            const int sp = sy * g.W + sx;
            const float py = (float)(sy - 1 + tap / 3) + so[si][0];
            const float px = (float)(sx - 1 + tap % 3) + so[si][1];
            const float mlogit = so[si][2];
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
                if (m < 0.f) m = sigmoidf_(mlogit);
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
        DX_STAMP(1);
        if (tap < 8) oload(tap + 1);
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
                        o.x = pk_bf16(a[0], a[1]); o.y = pk_bf16(a[2], a[3]);
                        o.z = pk_bf16(a[4], a[5]); o.w = pk_bf16(a[6], a[7]);
                        lds_store_vec<T, PITCH>(Gs, ql, col, o);
                    } else {
                        lds_store_vec<T, PITCH>(Gs, ql, col, make_uint4(__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3])));
                    }
                }
            }
            bstore(0);
            __syncthreads();
            DX_STAMP(2);
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
            DX_STAMP(3);
        }
    }

    int64_t pix[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = wm + i * 32 + (lane & 31);
        const int oh = th0 + m / DX_TW, ow = tw0 + m % DX_TW;
        pix[i] = (oh < g.H && ow < g.W) ? img + (int64_t)oh * g.W + ow : -1;
    }
    // lazy dx_far: only touched (added, then restored to zero) when a dom kernel flagged far samples for this layer
    const bool use_far = g.res32 != nullptr && (g.far_flag == nullptr || *g.far_flag != 0);
    conv_epilogue<T, MI, NJ>(g, acc, pix, n0 + wn, lane, use_far);
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
    CN_MAIN_PRIO_SET();
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
    // offsets / mask logit of the NEXT tap are fetched while the current tap is sampled and multiplied
    const int gh_ = th0 + (tid & (BM - 1)) / DX_TW, gw_ = tw0 + (tid & (BM - 1)) % DX_TW;
    const bool glive = tid < BM && gh_ < g.H && gw_ < g.W;
    const float* const orow = OM + ((int64_t)(glive ? gh_ : 0) * g.W + (glive ? gw_ : 0)) * g.dcn_omld;
    float ro[3] = {0.f, 0.f, 0.f};
    if (glive) { ro[0] = orow[0]; ro[1] = orow[1]; ro[2] = orow[18]; }
    for (int tap = 0; tap < 9; ++tap) {
        if (tid < BM) {
            const int h = gh_, w = gw_;
            int i0 = 0, i1 = 0, i2 = 0, i3 = 0;
            float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
            if (glive) {
                const float py = (float)(h - 1 + tap / 3) + ro[0];
                const float px = (float)(w - 1 + tap % 3) + ro[1];
                const float m = sigmoidf_(ro[2]);
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
        if (glive && tap < 8) { ro[0] = orow[2 * (tap + 1)]; ro[1] = orow[2 * (tap + 1) + 1]; ro[2] = orow[18 + tap + 1]; }
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
                        o.x = pk_bf16(a[0], a[1]); o.y = pk_bf16(a[2], a[3]);
                        o.z = pk_bf16(a[4], a[5]); o.w = pk_bf16(a[6], a[7]);
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
    if constexpr (sizeof(T) == 2 && (size_t)WGM * 32 * (BN + 4) * sizeof(float) <= (size_t)(BM + BN) * PITCH * sizeof(T)) {
        if (g.epi_tile) {                              // the main loop ended on a barrier: the operand tiles are dead.  LDS-staged epilogue:
            // 16-byte stores and the BatchNorm statistics hook (sum / sum of squares of the stored values)
            conv_epilogue_tile<MI, NJ, WGM, WGN>(g, acc, reinterpret_cast<float*>(lds), n0, tid, [&](int m) -> int64_t {
                const int oh = th0 + m / DX_TW, ow = tw0 + m % DX_TW;
                return (oh < g.H && ow < g.W) ? img + (int64_t)oh * g.W + ow : -1;
            });
            return;
        }
    }
    conv_epilogue<T, MI, NJ>(g, acc, pix, n0 + wn, lane);
}

template <typename T, int BN, int CK>
static void launch_fwd(const ConvGeom& g, hipStream_t st) {
    dim3 grid(((g.H + DX_TH - 1) / DX_TH) * ((g.W + DX_TW - 1) / DX_TW), (g.Co + BN - 1) / BN, g.N);
    hipLaunchKernelGGL((dcn_fwd_kernel<T, BN, CK>), grid, dim3(256), 0, st, g);
}

void dcn_fwd_launch(const ConvGeom& g_, int dtype, hipStream_t st) {
    ConvGeom g = g_;
    const int co32 = (g.Co + 31) / 32 * 32;
    int bn = 32, bw = co32;
    for (int c : {64, 128}) {
        int w = (co32 + c - 1) / c * c;
        if (w <= bw) { bn = c; bw = w; }
    }
    // LDS-staged epilogue (16-byte stores, BatchNorm statistics hook) where the output rows are vectors and the slab fits the operand
    // tiles' LDS (64-channel slices: every tile width; 32-channel slices: the 32-wide tile only)
    static const bool no_tile = getenv("CN_DISABLE_EPI_TILE") != nullptr || getenv("CN_DISABLE_DCN_GATHER_EPI_TILE") != nullptr;
    const bool fits = g.Ci % 64 == 0 || (g.Ci % 32 == 0 && bn == 32);
    g.epi_tile = (dtype == CN_BF16 && !no_tile && fits && conv_epi_tile_ok(g, dtype)) ? 1 : 0;
    if (g.bn_part) {
        if (g.epi_tile) mark_taken(g.bn_taken); else g.bn_part = nullptr;
    }
    if (dtype == CN_BF16) {
        if (g.Ci % 64 == 0) { if (bn == 128) launch_fwd<bf16_t, 128, 64>(g, st); else if (bn == 64) launch_fwd<bf16_t, 64, 64>(g, st); else launch_fwd<bf16_t, 32, 64>(g, st); }
        else if (g.Ci % 32 == 0) { if (bn == 128) launch_fwd<bf16_t, 128, 32>(g, st); else if (bn == 64) launch_fwd<bf16_t, 64, 32>(g, st); else launch_fwd<bf16_t, 32, 32>(g, st); }
        else { if (bn == 128) launch_fwd<bf16_t, 128, 16>(g, st); else if (bn == 64) launch_fwd<bf16_t, 64, 16>(g, st); else launch_fwd<bf16_t, 32, 16>(g, st); }
    } else {
        if (bn == 128) launch_fwd<float, 128, 16>(g, st); else if (bn == 64) launch_fwd<float, 64, 16>(g, st); else launch_fwd<float, 32, 16>(g, st);
    }
}


// ================================================================================================ offset / mask gradient
// dom[p][2k], dom[p][2k+1], dom[p][18+k]  (see cn_dcn_bwd_dom) with every operand tile-resident:
//   * the dY tile [128][Co] (A operand) and the x HALO tile [(8+7)x(16+7)][64] are loaded once per workgroup;
//   * per tap: weight slice [64 ci][Co] -> LDS, MFMA (K = Co) gives dcol_k [128][64] in registers -> LDS (bf16);
//     8 lanes per pixel then dot dcol against the bilinear corner combinations read from the halo tile in LDS
//     (16-byte ds reads instead of 36 L2 gathers per pixel), shuffle-reduce, one store per (pixel, tap).
// Corners that fall outside the halo (offsets beyond about +-2 px) are fetched from global memory; samples displaced by more
// than DCN_FAR_R pixels are scattered into dx_far.  512 threads = 8 waves, one 32x32 MFMA block each.  bf16 only.
#define DM_HR 3                       // halo reaches from -3 to +4 around the tile
#define DM_HH (DX_TH + 7)
#define DM_HW (DX_TW + 7)
#define DM_HP (DM_HH * DM_HW)

struct DomGeom {
    const bf16_t* dy; const bf16_t* wd2; const bf16_t* x; const float* om; float* dom; float* far; int* far_flag;
    int N, H, W, Ci, Co, dy_ld, x_ld, om_ld;
    int64_t slab;      // elements between the per-channel-block copies of dom (0: one copy, channel blocks meet with atomics)
    bf16_t* dom16;     // non-null (Ci == 64 only): dom is the FINAL bf16 tensor [P][om_ld], written directly (no fp32 copy + cast pass)
};

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ static inline float dot8_bf16(const uint4& a, const uint4& b, float acc) {   // v_dot2c_f32_bf16 x 4
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.x), __builtin_bit_cast(bf16x2_t, b.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.y), __builtin_bit_cast(bf16x2_t, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.z), __builtin_bit_cast(bf16x2_t, b.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.w), __builtin_bit_cast(bf16x2_t, b.w), acc, false);
    return acc;
}

__device__ static inline float quad_sum(float v) {        // sum over the 4 lanes of a quad, DPP only (no LDS traffic)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // lane ^ 1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // lane ^ 2
    return v;
}

// Persistent workgroups (one per CU, 512 threads): each owns a contiguous run of 8x16-pixel tiles.  Per tile the dY
// tile, the x halo tile and the 27 offset/mask values of every pixel are staged in LDS once; the NEXT tile's copies are
// fetched into registers while the current one is processed, and the per-tap weight slices run through a 3-deep register
// ring, so no global-memory latency is exposed inside the tap loop.  Per tap: dcol_k = dY x W_k^T on the matrix cores
// (bf16 tile back through LDS), then per pixel the four corner dot products D_ab = sum_c dcol[c] * x_ab[c] with
// v_dot2c_f32_bf16; the three gradients are linear in D_ab:
//   d mask   = a00 D00 + a01 D01 + a10 D10 + a11 D11
//   d off_y  = m ((1-lw)(D10-D00) + lw (D11-D01)),   d off_x = m ((1-lh)(D01-D00) + lh (D11-D10)).
template <int COP>   // padded Co (contraction length): 64 or 128
__global__ __launch_bounds__(512) void dcn_bwd_dom_kernel(const DomGeom g) {
    CN_MAIN_PRIO_SET();
    constexpr int BM = DX_TH * DX_TW, BN = 64;
    constexpr int AP = COP + 8;                 // dY tile / weight slice pitch
    constexpr int DP = BN + 8;                  // dcol tile / halo pitch
    constexpr int OMP = 27;                     // odd pitch: conflict-free per-pixel reads
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    bf16_t* const As = lds;                     // [128][AP]
    bf16_t* const Bs = As + BM * AP;            // 2 x [64][AP]
    bf16_t* const Ds = Bs + 2 * BN * AP;        // [128][DP]
    bf16_t* const Xh = Ds + BM * DP;            // [DM_HP][DP]
    float* const Om = reinterpret_cast<float*>(Xh + DM_HP * DP);   // [128][27]
    float* const Geo = Om + BM * OMP;           // 2 x [5][128]: h0, w0 (as int bits), lh, lw, m

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_w = (g.W + DX_TW - 1) / DX_TW, tiles_h = (g.H + DX_TH - 1) / DX_TH;
    const int tiles_img = tiles_w * tiles_h, ntiles = tiles_img * g.N;
    const int ci0 = blockIdx.y * BN;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const bool whole = g.Ci == BN || g.slab != 0;     // this workgroup's results are the only ones written to its dom rows
    // contiguous tile run per workgroup; consecutive runs stay on one XCD (workgroups are dealt round-robin to the 8 XCDs)
    const int G = gridDim.x;
    const int lb = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
    const int tpb = (ntiles + G - 1) / G;
    const int t_begin = lb * tpb, t_end = min(ntiles, t_begin + tpb);
    if (t_begin >= t_end) return;

    constexpr int DYV = BM * (COP / 8) / 512;
    constexpr int XV = (DM_HP * (BN / 8) + 511) / 512;
    constexpr int OV = (BM * 27 + 511) / 512;
    constexpr int BV = BN * (COP / 8) / 512;
    uint4 rdy[DYV], rx[XV], rb[3][BV];
    float rom[OV];

    auto tile_load = [&](int t) {               // branch-free: out-of-image pieces are masked, not skipped
        const int n = t / tiles_img, r = t % tiles_img;
        const int th0 = (r / tiles_w) * DX_TH, tw0 = (r % tiles_w) * DX_TW;
        const int64_t img = (int64_t)n * g.H * g.W;
#pragma unroll
        for (int i = 0; i < DYV; ++i) {
            const int v = tid + i * 512;
            const int pl = v / (COP / 8), col = (v % (COP / 8)) * 8;
            const int h = th0 + pl / DX_TW, w = tw0 + pl % DX_TW;
            rdy[i] = ldg16_masked(g.dy, ((img + (int64_t)h * g.W + w) * g.dy_ld + col) * 2, h < g.H && w < g.W && col < g.dy_ld);
        }
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = tid + i * 512;
            const int hp = v / (BN / 8), col = (v % (BN / 8)) * 8;
            const int h = th0 - DM_HR + hp / DM_HW, w = tw0 - DM_HR + hp % DM_HW;
            rx[i] = ldg16_masked(g.x, ((img + (int64_t)h * g.W + w) * g.x_ld + ci0 + col) * 2,
                                 hp < DM_HP && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W);
        }
#pragma unroll
        for (int i = 0; i < OV; ++i) {
            const int e = tid + i * 512;
            const int pl = e / 27, k = e % 27;
            const int h = th0 + pl / DX_TW, w = tw0 + pl % DX_TW;
            const bool ok = pl < BM && h < g.H && w < g.W;
            const float v = g.om[ok ? (img + (int64_t)h * g.W + w) * g.om_ld + k : 0];
            rom[i] = ok ? v : 0.f;
        }
    };
    auto tile_store = [&]() {
#pragma unroll
        for (int i = 0; i < DYV; ++i) {
            const int v = tid + i * 512;
            st16(As + (v / (COP / 8)) * AP + (v % (COP / 8)) * 8, rdy[i]);
        }
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int v = tid + i * 512;
            if (v < DM_HP * (BN / 8)) st16(Xh + (v / (BN / 8)) * DP + (v % (BN / 8)) * 8, rx[i]);
        }
#pragma unroll
        for (int i = 0; i < OV; ++i) {
            const int e = tid + i * 512;
            if (e < BM * 27) Om[e] = rom[i];    // pitch 27: the flat element index is the LDS index
        }
    };
    auto bload = [&](uint4 (&r)[BV], int tap) { // rows tap*Ci + ci0 .. +63 of the mode-2 packed matrix [9*Ci][COP]
#pragma unroll
        for (int i = 0; i < BV; ++i) {
            const int v = tid + i * 512;
            r[i] = ldg16(g.wd2 + ((int64_t)tap * g.Ci + ci0 + v / (COP / 8)) * COP + (v % (COP / 8)) * 8);
        }
    };
    auto stage_tap = [&](const uint4 (&r)[BV], int tap, int buf, int th0, int tw0) {   // weight slice + sample geometry -> LDS
#pragma unroll
        for (int i = 0; i < BV; ++i) {
            const int v = tid + i * 512;
            st16(Bs + buf * BN * AP + (v / (COP / 8)) * AP + (v % (COP / 8)) * 8, r[i]);
        }
        if (tid < BM) {
            const int gh = th0 + tid / DX_TW, gw = tw0 + tid % DX_TW;
            const float* o = Om + tid * OMP;
            const float py = (float)(gh - 1 + tap / 3) + o[2 * tap], px = (float)(gw - 1 + tap % 3) + o[2 * tap + 1];
            const float fh = floorf(py), fw = floorf(px);
            float* gq = Geo + buf * 5 * BM + tid;
            gq[0] = __int_as_float((int)fh); gq[BM] = __int_as_float((int)fw);
            gq[2 * BM] = py - fh; gq[3 * BM] = px - fw;
            gq[4 * BM] = (gh < g.H && gw < g.W) ? sigmoidf_(o[18 + tap]) : 0.f;
        }
    };

    tile_load(t_begin);
#pragma unroll
    for (int d = 0; d < 3; ++d) bload(rb[d], d);
    tile_store();
    __syncthreads();
    {
        const int r0 = t_begin % tiles_img;
        stage_tap(rb[0], 0, 0, (r0 / tiles_w) * DX_TH, (r0 % tiles_w) * DX_TW);
    }
    bload(rb[0], 3);
    __syncthreads();

#pragma unroll 1
    for (int t = t_begin; t < t_end; ++t) {
        const int n = t / tiles_img, r = t % tiles_img;
        const int th0 = (r / tiles_w) * DX_TH, tw0 = (r % tiles_w) * DX_TW;
        const int64_t img = (int64_t)n * g.H * g.W;
        const int tn = (t + 1 < t_end) ? t + 1 : t;            // the last tile prefetches itself again: no branches around loads
        const int rn = tn % tiles_img;
        const int nth0 = (rn / tiles_w) * DX_TH, ntw0 = (rn % tiles_w) * DX_TW;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // invariant: Bs[tap&1], Geo[tap&1] hold this tap; ring slot (tap+1)%3 holds tap+1, (tap+2)%3 holds tap+2,
            // slot tap%3 is loading tap+3
            const int buf = tap & 1;
            if (tap == 1) tile_load(tn);
            // ---- dcol_k tile = dY tile x W_k^T : one 32x32 block per wave ----
            f32x16_t acc;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) acc[rr] = 0.f;
#pragma unroll
            for (int kk = 0; kk < COP / 16; ++kk) {
                const bf16x8_t fa = Mma<bf16_t>::load(As, AP, wm, kk, lane);                       // pixels
                const bf16x8_t fb = Mma<bf16_t>::load(Bs + buf * BN * AP, AP, wn, kk, lane);       // channels (ci)
                acc = Mma<bf16_t>::mma(fb, fa, acc);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint2 o;
                o.x = pk_bf16(acc[q * 4 + 0], acc[q * 4 + 1]);
                o.y = pk_bf16(acc[q * 4 + 2], acc[q * 4 + 3]);
                *reinterpret_cast<uint2*>(Ds + (wm + (lane & 31)) * DP + wn + 8 * q + 4 * (lane >> 5)) = o;
            }
            // stage the next tap while this one is reduced (tap 8 stages tap 0 of the next tile after the tile swap below)
            if (tap < 8) {
                stage_tap(rb[(tap + 1) % 3], tap + 1, buf ^ 1, th0, tw0);
                bload(rb[(tap + 1) % 3], (tap + 4) % 9);
            }
            __syncthreads();
            // ---- per (pixel, 16-channel slice): corner dot products from the halo tile; 4 lanes share a pixel ----
            const float* gq = Geo + buf * 5 * BM;
            {
                const int pl = tid >> 2, lq = tid & 3;
                const int h = th0 + pl / DX_TW, w = tw0 + pl % DX_TW;
                const bool live = h < g.H && w < g.W;
                const int h0 = __float_as_int(gq[pl]), w0 = __float_as_int(gq[BM + pl]);
                const float lh = gq[2 * BM + pl], lw = gq[3 * BM + pl], mk = gq[4 * BM + pl];
                const int hy = h0 - (th0 - DM_HR), hx = w0 - (tw0 - DM_HR);       // halo coordinates of corner 00
                uint4 x00[2], x01[2], x10[2], x11[2];
                if (hy >= 0 && hy + 1 < DM_HH && hx >= 0 && hx + 1 < DM_HW) {     // all four corners inside the LDS halo (zeros outside the image)
                    const bf16_t* b = Xh + (hy * DM_HW + hx) * DP + lq * 16;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        x00[u] = *reinterpret_cast<const uint4*>(b + u * 8);
                        x01[u] = *reinterpret_cast<const uint4*>(b + DP + u * 8);
                        x10[u] = *reinterpret_cast<const uint4*>(b + DM_HW * DP + u * 8);
                        x11[u] = *reinterpret_cast<const uint4*>(b + DM_HW * DP + DP + u * 8);
                    }
                } else {                                                          // rare: fetch from global, zero outside the image
                    auto gl = [&](int hh, int ww, int u) {
                        return ldg16_masked(g.x, ((img + (int64_t)hh * g.W + ww) * g.x_ld + ci0 + lq * 16 + u * 8) * 2,
                                            (unsigned)hh < (unsigned)g.H && (unsigned)ww < (unsigned)g.W);
                    };
#pragma unroll
                    for (int u = 0; u < 2; ++u) { x00[u] = gl(h0, w0, u); x01[u] = gl(h0, w0 + 1, u); x10[u] = gl(h0 + 1, w0, u); x11[u] = gl(h0 + 1, w0 + 1, u); }
                }
                const uint4 gc0 = *reinterpret_cast<const uint4*>(Ds + pl * DP + lq * 16), gc1 = *reinterpret_cast<const uint4*>(Ds + pl * DP + lq * 16 + 8);
                const float d00 = dot8_bf16(gc1, x00[1], dot8_bf16(gc0, x00[0], 0.f)), d01 = dot8_bf16(gc1, x01[1], dot8_bf16(gc0, x01[0], 0.f));
                const float d10 = dot8_bf16(gc1, x10[1], dot8_bf16(gc0, x10[0], 0.f)), d11 = dot8_bf16(gc1, x11[1], dot8_bf16(gc0, x11[0], 0.f));
                const float a00 = (1.f - lh) * (1.f - lw), a01 = (1.f - lh) * lw, a10 = lh * (1.f - lw), a11 = lh * lw;
                float sm = a00 * d00 + a01 * d01 + a10 * d10 + a11 * d11;
                float sy = (1.f - lw) * (d10 - d00) + lw * (d11 - d01);
                float sx = (1.f - lh) * (d01 - d00) + lh * (d11 - d10);
                // far samples (the adjoint-gather window of cn_dcn_bwd_dx cannot see them)
                const int dh0 = h0 - h, dw0 = w0 - w;
                const bool far_h0 = dh0 > DCN_FAR_R || dh0 < -DCN_FAR_R, far_h1 = dh0 + 1 > DCN_FAR_R || dh0 + 1 < -DCN_FAR_R;
                const bool far_w0 = dw0 > DCN_FAR_R || dw0 < -DCN_FAR_R, far_w1 = dw0 + 1 > DCN_FAR_R || dw0 + 1 < -DCN_FAR_R;
                if (live && (far_h0 || far_h1 || far_w0 || far_w1)) {
                    const bool in_h0 = (unsigned)h0 < (unsigned)g.H, in_h1 = (unsigned)(h0 + 1) < (unsigned)g.H;
                    const bool in_w0 = (unsigned)w0 < (unsigned)g.W, in_w1 = (unsigned)(w0 + 1) < (unsigned)g.W;
                    if (g.far_flag) *g.far_flag = 1;
                    float* far = g.far + (img + (int64_t)h0 * g.W + w0) * g.Ci + ci0 + lq * 16;
                    float gc[16];
                    Vec16<bf16_t>::load(Ds + pl * DP + lq * 16, gc);
                    Vec16<bf16_t>::load(Ds + pl * DP + lq * 16 + 8, gc + 8);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float gm = gc[e] * mk;
                        if (in_h0 && in_w0 && a00 != 0.f && (far_h0 || far_w0)) atomicAdd(far + e, gm * a00);
                        if (in_h0 && in_w1 && a01 != 0.f && (far_h0 || far_w1)) atomicAdd(far + g.Ci + e, gm * a01);
                        if (in_h1 && in_w0 && a10 != 0.f && (far_h1 || far_w0)) atomicAdd(far + (int64_t)g.W * g.Ci + e, gm * a10);
                        if (in_h1 && in_w1 && a11 != 0.f && (far_h1 || far_w1)) atomicAdd(far + (int64_t)(g.W + 1) * g.Ci + e, gm * a11);
                    }
                }
                sm = quad_sum(sm); sy = quad_sum(sy); sx = quad_sum(sx);
                if (lq == 0 && live) {
                    float* d = g.dom + (int64_t)blockIdx.y * g.slab + (img + (int64_t)h * g.W + w) * g.om_ld;
                    const float vy = sy * mk, vx = sx * mk, vm = sm * mk * (1.f - mk);
                    if (g.dom16) {
                        bf16_t* d16 = g.dom16 + (img + (int64_t)h * g.W + w) * g.om_ld;
                        d16[2 * tap] = f2bf(vy); d16[2 * tap + 1] = f2bf(vx); d16[18 + tap] = f2bf(vm);
                        if (tap == 0) for (int c = 27; c < g.om_ld; ++c) d16[c] = 0;
                    } else if (whole) {
                        d[2 * tap] = vy; d[2 * tap + 1] = vx; d[18 + tap] = vm;
                        if (tap == 0) for (int c = 27; c < g.om_ld; ++c) d[c] = 0.f;     // channel padding
                    }
                    else { atomicAdd(d + 2 * tap, vy); atomicAdd(d + 2 * tap + 1, vx); atomicAdd(d + 18 + tap, vm); }
                }
            }
            __syncthreads();      // Ds and this tap's Bs / geometry may be rewritten
        }
        // ---- tile swap: everyone is done with As / Xh / Om ----
        tile_store();
        __syncthreads();
        stage_tap(rb[0], 0, 0, nth0, ntw0);     // ring slot 0 holds tap 0 again (loaded at tap 5)
        bload(rb[0], 3);
        __syncthreads();
    }
}

// returns false when the shape is not handled by the tile-resident kernel
bool dcn_bwd_dom_tile_launch(const void* dy, const void* wd2, const void* x, const float* om, float* dom, int dom_slabs, float* far, int* far_flag,
                             int N, int H, int W, int Ci, int Co, int dy_ld, int x_ld, int om_ld, hipStream_t st) {
    static const bool disabled = getenv("CN_DISABLE_DOM_TILE") != nullptr;
    if (disabled || Ci % 64 != 0 || (dy_ld != 64 && dy_ld != 128)) return false;
    (void)Co;
    DomGeom g;
    g.dy = (const bf16_t*)dy; g.wd2 = (const bf16_t*)wd2; g.x = (const bf16_t*)x; g.om = om; g.dom = dom; g.far = far; g.far_flag = far_flag;
    g.N = N; g.H = H; g.W = W; g.Ci = Ci; g.Co = Co; g.dy_ld = dy_ld; g.x_ld = x_ld; g.om_ld = om_ld;
    g.slab = (dom_slabs > 1 && dom_slabs == Ci / 64) ? (int64_t)N * H * W * om_ld : 0;
    g.dom16 = nullptr;
    if (dom_slabs == 0) {                       // direct bf16 result: only when ONE workgroup column owns every dom row
        if (Ci != 64) return false;
        g.dom16 = (bf16_t*)dom;
    }
    const int ntiles = ((H + DX_TH - 1) / DX_TH) * ((W + DX_TW - 1) / DX_TW) * N;
    int gx = 256 / (Ci / 64);                   // one persistent workgroup per CU
    if (gx < 8) gx = 8;
    if (gx > ntiles) gx = ntiles;
    dim3 grid(gx, Ci / 64, 1);
    const int cop = dy_ld;
    const size_t smem = ((size_t)(128 + 2 * 64) * (cop + 8) + (size_t)(128 + DM_HP) * 72) * sizeof(bf16_t) + (size_t)(128 * 27 + 2 * 5 * 128) * sizeof(float);
    if (cop == 64) {
        (void)hipFuncSetAttribute((const void*)dcn_bwd_dom_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(dcn_bwd_dom_kernel<64>, grid, dim3(512), smem, st, g);
    } else {
        (void)hipFuncSetAttribute((const void*)dcn_bwd_dom_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(dcn_bwd_dom_kernel<128>, grid, dim3(512), smem, st, g);
    }
    return true;
}
