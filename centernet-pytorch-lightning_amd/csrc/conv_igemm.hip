// Implicit-GEMM convolution for gfx950 (NHWC activations, packed weights [Co_pad][taps*Ci]).
//
//   D[channel][pixel] = sum_k Wp[channel][k] * Xg[pixel][k]        (k = tap*Ci + ci)
//
// One workgroup = 4 waves (256 threads) computes a BM=128 pixel x BN channel tile.  K is walked in
// BK-element slices that never straddle a tap (Ci % BK == 0), so an A slice is BK contiguous
// channels of one (possibly out-of-image -> zero) input pixel: 16-byte coalesced NHWC loads.
// Slices are register-prefetched one step ahead and staged through double-buffered LDS (one
// barrier per step); rows are padded by 16 B so the ds_read_b128 fragment reads are conflict-free.
// bf16 uses v_mfma_f32_32x32x16_bf16, fp32 (parity mode) v_mfma_f32_32x32x2_f32 — both accumulate
// in fp32.  The weight tile is the FIRST MFMA operand so each lane ends up with 4 consecutive
// output channels of one pixel per accumulator quad -> 8/16-byte NHWC stores.
//
// "transposed" mode (ConvTranspose2d forward and the data gradient of a strided conv) is
// decomposed by output parity class (blockIdx.z): only the taps whose offset divides the stride
// are visited, so no MACs are wasted.  Geometry per class comes as small tap tables in the
// kernel arguments (built by the host below).
#include "conv_common.h"
#include <stdlib.h>

template <typename T, int BN, int BK, bool DOM = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvGeom g) {
    CN_MAIN_PRIO_SET();
    constexpr int BM = 128;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int VPR = BK / VEC;         // 16-byte vectors per tile row
    constexpr int RPP = 256 / VPR;        // rows loaded per pass
    constexpr int APASS = BM / RPP;
    constexpr int BPASS = (BN + RPP - 1) / RPP;
    constexpr int PITCH = BK + Mma<T>::PAD;
    constexpr int WGN = (BN >= 64) ? 2 : 1;   // waves along channels
    constexpr int WGM = 4 / WGN;              // waves along pixels
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MI = WM / 32, NJ = WN / 32;
    constexpr int KSTEPS = BK / Mma<T>::KSTEP;

    constexpr int BUF = (BM + BN) * PITCH;  // elements per pipeline stage: [A tile | B tile]
    constexpr int EPI_ELEMS = (sizeof(T) == 2 && !DOM) ? WGM * 32 * (BN + 4) * 2 : 0;   // fp32 slab of the LDS-staged epilogue
    __shared__ __attribute__((aligned(16))) T lds[2 * BUF > EPI_ELEMS ? 2 * BUF : EPI_ELEMS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cls = blockIdx.z;
    const int ph = cls / g.so, pw = cls % g.so;
    const int OHc = (g.OH - ph + g.so - 1) / g.so, OWc = (g.OW - pw + g.so - 1) / g.so;
    const int Mc = g.N * OHc * OWc;
    const int m0 = blockIdx.x * BM;
    if (m0 >= Mc) return;
    const int n0 = blockIdx.y * BN;
    const int ntaps = g.ntaps[cls];
    const int cpt = g.Ci / BK;  // K slices per tap
    const int nsteps = ntaps * cpt;

    const T* __restrict__ X = reinterpret_cast<const T*>(g.x);
    const T* __restrict__ Wp = reinterpret_cast<const T*>(g.w);

    // ---- loader coordinates (fixed per thread) ----
    const int vcol = (tid % VPR) * VEC;
    const int lrow = tid / VPR;
    int a_ihb[APASS], a_iwb[APASS];
    int64_t a_img[APASS];
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
        int m = m0 + lrow + p * RPP;
        if (m < Mc) {
            int n = m / (OHc * OWc);
            int r = m - n * (OHc * OWc);
            int oh = r / OWc, ow = r - oh * OWc;
            a_ihb[p] = oh * g.sm;
            a_iwb[p] = ow * g.sm;
            a_img[p] = (int64_t)n * g.H * g.W;
        } else {
            a_ihb[p] = -100000;  // never in bounds
            a_iwb[p] = 0;
            a_img[p] = 0;
        }
    }

    // Two register sets: slices are fetched TWO steps ahead (one L2 round trip is longer than one slice of MFMA work), with
    // branch-free guarded loads so the compiler can keep counting the loads in flight (vmcnt(N), not vmcnt(0)).
    uint4 ra[2][APASS], rb[2][BPASS];
    auto gload = [&](uint4 (&qa)[APASS], uint4 (&qb)[BPASS], int step_raw) {
        const int step = step_raw < nsteps ? step_raw : nsteps - 1;      // past the end: re-read the last slice (never used)
        const int t = step / cpt;
        const int c0 = (step - t * cpt) * BK;
        const int dh = g.dh[cls][t], dw = g.dw[cls][t];
        const int wofs = (int)g.wt[cls][t] * g.Ci + c0 + vcol;
        // channel concatenation of several sources (1x1 only: one tap, c0 = K index): pick the source this slice lies in —
        // wave-uniform scalar work, the slice never straddles two sources
        const T* Xs = X;
        int ld = g.x_ld, cin = c0;
        if (g.nsrc > 0) {
            int sidx = 0;
#pragma unroll
            for (int q = 1; q < CN_MAX_SRC; ++q) sidx += (q < g.nsrc && c0 >= g.xs_k0[q]) ? 1 : 0;
            Xs = reinterpret_cast<const T*>(g.xs[sidx]);
            ld = g.xs_c[sidx];
            cin = c0 - g.xs_k0[sidx];
        }
#pragma unroll
        for (int p = 0; p < APASS; ++p) {
            const int ih = a_ihb[p] + dh, iw = a_iwb[p] + dw;
            qa[p] = ldg16_masked(Xs, ((a_img[p] + (int64_t)ih * g.W + iw) * ld + cin + vcol) * (int64_t)sizeof(T),
                                 (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W);
        }
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
            int row = n0 + lrow + p * RPP;                               // rows past the packed matrix re-read its last row
            row = row < g.co_pad ? row : g.co_pad - 1;
            qb[p] = *reinterpret_cast<const uint4*>(Wp + (int64_t)row * g.ktot + wofs);
        }
    };
    auto lstore = [&](const uint4 (&qa)[APASS], const uint4 (&qb)[BPASS], int buf) {
#pragma unroll
        for (int p = 0; p < APASS; ++p) lds_store_vec<T, PITCH>((lds + buf * BUF), lrow + p * RPP, vcol, qa[p]);
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
            const int row = lrow + p * RPP;
            if (BN % RPP == 0 || row < BN) lds_store_vec<T, PITCH>((lds + buf * BUF + BM * PITCH), row, vcol, qb[p]);
        }
    };

    f32x16_t acc[NJ][MI];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    const int wm = (wave / WGN) * WM, wn = (wave % WGN) * WN;
    auto compute = [&](int buf) {
        const T* at = (lds + buf * BUF);
        const T* bt = (lds + buf * BUF + BM * PITCH);
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            typename Mma<T>::Frag fa[MI], fb[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = Mma<T>::load(at, PITCH, wm + i * 32, kk, lane);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = Mma<T>::load(bt, PITCH, wn + j * 32, kk, lane);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[j][i] = Mma<T>::mma(fb[j], fa[i], acc[j][i]);
        }
    };

    if (nsteps > 0) {
        gload(ra[0], rb[0], 0);
        gload(ra[1], rb[1], 1);
        lstore(ra[0], rb[0], 0);
    }
    __syncthreads();
#pragma unroll 1
    for (int step = 0; step < nsteps; step += 2) {
        // slice `step` is in LDS buffer 0, slice step+1 in register set 1 (in flight)
        gload(ra[0], rb[0], step + 2);
        compute(0);
        lstore(ra[1], rb[1], 1);
        __syncthreads();
        gload(ra[1], rb[1], step + 3);
        if (step + 1 < nsteps) compute(1);
        lstore(ra[0], rb[0], 0);
        __syncthreads();
    }

    // ---- epilogue ----
    int64_t pix[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm + i * 32 + (lane & 31);
        if (m >= Mc) { pix[i] = -1; continue; }
        if (g.so == 1) {
            pix[i] = m;
        } else {
            int n = m / (OHc * OWc);
            int r = m - n * (OHc * OWc);
            int oh = r / OWc, ow = r - oh * OWc;
            pix[i] = ((int64_t)n * g.OH + oh * g.so + ph) * g.OW + ow * g.so + pw;
        }
    }
    if constexpr (!DOM) {
        if constexpr (sizeof(T) == 2) {
            if (g.epi_tile) {                          // main loop ended on a barrier: the LDS is free
                conv_epilogue_tile<MI, NJ, WGM, WGN>(g, acc, reinterpret_cast<float*>(lds), n0, tid, [&](int ml) -> int64_t {
                    const int m = m0 + ml;
                    if (m >= Mc) return -1;
                    if (g.so == 1) return m;
                    const int n = m / (OHc * OWc);
                    const int r = m - n * (OHc * OWc);
                    const int oh = r / OWc, ow = r - oh * OWc;
                    return ((int64_t)n * g.OH + oh * g.so + ph) * g.OW + ow * g.so + pw;
                });
                return;
            }
        }
        conv_epilogue<T, MI, NJ>(g, acc, pix, n0 + wn, lane);
    } else {
        // fused DCNv2 offset / mask gradient (the GEMM result dcol is never stored to HBM)
        const int Ci = g.dcn_Ci;
        if constexpr (sizeof(T) == 2) {
            // bf16: park the dcol tile [128 pixels][BN channels] in LDS, then GS = BN/8 lanes per (pixel, tap) read 16-byte
            // slices of it and of the four bilinear corners of x (coalesced 128/256-byte rows), reduce with shuffles.
            constexpr int DP = BN + 8;
            constexpr int GS = BN / 8;
            static_assert(BM * DP <= 2 * BUF, "dcol tile must fit in the pipeline buffers");
            bf16_t* dt = reinterpret_cast<bf16_t*>(lds);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint2 o;
                        o.x = pk_bf16(acc[j][i][q * 4 + 0], acc[j][i][q * 4 + 1]);
                        o.y = pk_bf16(acc[j][i][q * 4 + 2], acc[j][i][q * 4 + 3]);
                        *reinterpret_cast<uint2*>(dt + (wm + i * 32 + (lane & 31)) * DP + wn + j * 32 + 8 * q + 4 * (lane >> 5)) = o;
                    }
            __syncthreads();
            const int k = n0 / Ci, cb = n0 - k * Ci;              // BN <= Ci here: one tap, channel offset cb
            const bool whole = BN >= Ci;
            const int lg = threadIdx.x % GS, gl = threadIdx.x / GS;
            const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(g.dcn_x);
            const int H = g.dcn_H, W = g.dcn_W;
            // 128 pixels / (256/GS) pixels per pass = NP passes; all passes' dependent loads (offsets -> corners) are issued
            // together so a workgroup pays two memory round trips instead of 2*NP.
            constexpr int NPT = BM / (256 / GS);     // passes in total
            constexpr int NP = NPT < 4 ? NPT : 4;     // passes batched together (register budget)
#pragma unroll 1
            for (int ub = 0; ub < NPT; ub += NP) {
            const int glb = gl + ub * (256 / GS);
            float py[NP], px[NP], mk[NP];
            bool live[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int m = m0 + glb + u * (256 / GS);
                live[u] = m < Mc;
                const float* o = g.dcn_om + (int64_t)(live[u] ? m : m0) * g.dcn_omld;
                py[u] = o[2 * k]; px[u] = o[2 * k + 1]; mk[u] = o[18 + k];
            }
            uint4 r00[NP], r01[NP], r10[NP], r11[NP];
            Tap tp[NP];
            int hh[NP], ww[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int m = live[u] ? m0 + glb + u * (256 / GS) : m0;
                ww[u] = m % W; hh[u] = (m / W) % H;
                const int64_t img = (int64_t)m - ((int64_t)hh[u] * W + ww[u]);
                py[u] += (float)(hh[u] - 1 + k / 3);
                px[u] += (float)(ww[u] - 1 + k % 3);
                mk[u] = sigmoidf_(mk[u]);
                tp[u] = make_tap(py[u], px[u], H, W);
                // unconditional loads from clamped addresses; out-of-image corners are zeroed below
                const int hc0 = min(max(tp[u].h0, 0), H - 1), hc1 = min(max(tp[u].h0 + 1, 0), H - 1);
                const int wc0 = min(max(tp[u].w0, 0), W - 1), wc1 = min(max(tp[u].w0 + 1, 0), W - 1);
                const bf16_t* xb = X + img * g.dcn_xld + cb + lg * 8;
                r00[u] = *reinterpret_cast<const uint4*>(xb + ((int64_t)hc0 * W + wc0) * g.dcn_xld);
                r01[u] = *reinterpret_cast<const uint4*>(xb + ((int64_t)hc0 * W + wc1) * g.dcn_xld);
                r10[u] = *reinterpret_cast<const uint4*>(xb + ((int64_t)hc1 * W + wc0) * g.dcn_xld);
                r11[u] = *reinterpret_cast<const uint4*>(xb + ((int64_t)hc1 * W + wc1) * g.dcn_xld);
            }
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int ml = glb + u * (256 / GS);
                const int m = m0 + ml;
                const Tap& t = tp[u];
                const int h = hh[u], w = ww[u];
                const int64_t img = (int64_t)m - ((int64_t)h * W + w);
                float gc[8];
                Vec16<bf16_t>::load(dt + ml * DP + lg * 8, gc);
                const float k00 = t.ok00 ? 1.f : 0.f, k01 = t.ok01 ? 1.f : 0.f, k10 = t.ok10 ? 1.f : 0.f, k11 = t.ok11 ? 1.f : 0.f;
                const uint32_t w00_[4] = {r00[u].x, r00[u].y, r00[u].z, r00[u].w}, w01_[4] = {r01[u].x, r01[u].y, r01[u].z, r01[u].w};
                const uint32_t w10_[4] = {r10[u].x, r10[u].y, r10[u].z, r10[u].w}, w11_[4] = {r11[u].x, r11[u].y, r11[u].z, r11[u].w};
                float s_m = 0.f, s_y = 0.f, s_x = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int sh = (e & 1) ? 0 : 16;
                    const float a = __uint_as_float((e & 1) ? (w00_[e >> 1] & 0xffff0000u) : (w00_[e >> 1] << sh)) * k00;
                    const float b = __uint_as_float((e & 1) ? (w01_[e >> 1] & 0xffff0000u) : (w01_[e >> 1] << sh)) * k01;
                    const float c = __uint_as_float((e & 1) ? (w10_[e >> 1] & 0xffff0000u) : (w10_[e >> 1] << sh)) * k10;
                    const float d = __uint_as_float((e & 1) ? (w11_[e >> 1] & 0xffff0000u) : (w11_[e >> 1] << sh)) * k11;
                    s_m = fmaf(gc[e], a * t.w00 + b * t.w01 + c * t.w10 + d * t.w11, s_m);
                    s_y = fmaf(gc[e], (1.f - t.lw) * (c - a) + t.lw * (d - b), s_y);
                    s_x = fmaf(gc[e], (1.f - t.lh) * (b - a) + t.lh * (d - c), s_x);
                }
                const int dh0 = t.h0 - h, dw0 = t.w0 - w;
                const bool far_h0 = dh0 > DCN_FAR_R || dh0 < -DCN_FAR_R, far_h1 = dh0 + 1 > DCN_FAR_R || dh0 + 1 < -DCN_FAR_R;
                const bool far_w0 = dw0 > DCN_FAR_R || dw0 < -DCN_FAR_R, far_w1 = dw0 + 1 > DCN_FAR_R || dw0 + 1 < -DCN_FAR_R;
                if (live[u] && (far_h0 || far_h1 || far_w0 || far_w1)) {       // samples the adjoint-gather window cannot see
                    const int64_t i00 = img + (int64_t)t.h0 * W + t.w0;
                    if (g.far_flag) *g.far_flag = 1;
                    float* far = g.dcn_far + cb + lg * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float gm = gc[e] * mk[u];
                        if (t.w00 != 0.f && (far_h0 || far_w0)) atomicAdd(far + i00 * Ci + e, gm * t.w00);
                        if (t.w01 != 0.f && (far_h0 || far_w1)) atomicAdd(far + (i00 + 1) * Ci + e, gm * t.w01);
                        if (t.w10 != 0.f && (far_h1 || far_w0)) atomicAdd(far + (i00 + W) * Ci + e, gm * t.w10);
                        if (t.w11 != 0.f && (far_h1 || far_w1)) atomicAdd(far + (i00 + W + 1) * Ci + e, gm * t.w11);
                    }
                }
#pragma unroll
                for (int ofs = GS >> 1; ofs > 0; ofs >>= 1) {
                    s_m += __shfl_xor(s_m, ofs, 64);
                    s_y += __shfl_xor(s_y, ofs, 64);
                    s_x += __shfl_xor(s_x, ofs, 64);
                }
                if (lg == 0 && live[u]) {
                    float* d = g.dcn_dom + (int64_t)m * g.dcn_omld;
                    const float vy = s_y * mk[u], vx = s_x * mk[u], vm = s_m * mk[u] * (1.f - mk[u]);
                    if (whole) { d[2 * k] = vy; d[2 * k + 1] = vx; d[18 + k] = vm; }
                    else { atomicAdd(d + 2 * k, vy); atomicAdd(d + 2 * k + 1, vx); atomicAdd(d + 18 + k, vm); }
                }
            }
            }
        } else {
            // fp32 (parity mode): per-lane accumulation straight from the MFMA registers, LDS fold across the channel waves
            const int tpb = BN >= Ci ? BN / Ci : 1;                 // taps covered by this workgroup's BN channels
            float* red = reinterpret_cast<float*>(lds);              // [128][tpb][3]   (the tiles are dead after the last barrier)
            for (int i = threadIdx.x; i < BM * tpb * 3; i += 256) red[i] = 0.f;
            __syncthreads();
            int mloc[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) mloc[i] = wm + i * 32 + (lane & 31);
            dcn_dom_accumulate<T, MI, NJ>(g, acc, pix, mloc, n0, n0 + wn, lane, red, tpb);
            __syncthreads();
            const bool whole = BN >= Ci;                             // this workgroup saw every channel of its taps
            for (int i = threadIdx.x; i < BM * tpb; i += 256) {
                const int ml = i / tpb, tl = i - ml * tpb;
                const int m = m0 + ml;
                const int k = n0 / Ci + tl;
                if (m >= Mc || k >= 9) continue;
                const float mk = sigmoidf_(g.dcn_om[(int64_t)m * g.dcn_omld + 18 + k]);
                const float vy = red[i * 3 + 0] * mk, vx = red[i * 3 + 1] * mk, vm = red[i * 3 + 2] * mk * (1.f - mk);
                float* d = g.dcn_dom + (int64_t)m * g.dcn_omld;
                if (whole) { d[2 * k] = vy; d[2 * k + 1] = vx; d[18 + k] = vm; }
                else { atomicAdd(d + 2 * k, vy); atomicAdd(d + 2 * k + 1, vx); atomicAdd(d + 18 + k, vm); }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host
static int build_geom(ConvGeom& g, int KH, int KW, int stride, int pad, int transposed) {
    if (!transposed) {
        g.sm = stride;
        g.so = 1;
        int nt = KH * KW;
        if (nt > CN_MAX_TAPS) return -1;
        g.ntaps[0] = nt;
        for (int kh = 0; kh < KH; ++kh)
            for (int kw = 0; kw < KW; ++kw) {
                int t = kh * KW + kw;
                g.dh[0][t] = (signed char)(kh - pad);
                g.dw[0][t] = (signed char)(kw - pad);
                g.wt[0][t] = (unsigned char)t;
            }
        return 1;
    }
    if (stride * stride > CN_MAX_CLS) return -1;
    g.sm = 1;
    g.so = stride;
    for (int ph = 0; ph < stride; ++ph)
        for (int pw = 0; pw < stride; ++pw) {
            int c = ph * stride + pw, nt = 0;
            for (int kh = 0; kh < KH; ++kh) {
                if (((ph + pad - kh) % stride + stride) % stride != 0) continue;
                for (int kw = 0; kw < KW; ++kw) {
                    if (((pw + pad - kw) % stride + stride) % stride != 0) continue;
                    if (nt >= CN_MAX_TAPS) return -1;
                    // ih = (ohc*stride + ph + pad - kh)/stride = ohc + (ph + pad - kh)/stride  (exact)
                    int qh = ph + pad - kh, qw = pw + pad - kw;
                    g.dh[c][nt] = (signed char)(qh >= 0 ? qh / stride : -((-qh) / stride));
                    g.dw[c][nt] = (signed char)(qw >= 0 ? qw / stride : -((-qw) / stride));
                    g.wt[c][nt] = (unsigned char)(kh * KW + kw);
                    ++nt;
                }
            }
            g.ntaps[c] = nt;
        }
    return stride * stride;
}

template <typename T, int BN, int BK>
static void launch_igemm(const ConvGeom& g, int ncls, hipStream_t st) {
    int so = g.so;
    int OHc = (g.OH + so - 1) / so, OWc = (g.OW + so - 1) / so;
    int64_t Mc = (int64_t)g.N * OHc * OWc;
    dim3 grid(cdiv(Mc, 128), cdiv(g.Co, BN), ncls);
    static const bool no_tile = getenv("CN_DISABLE_EPI_TILE") != nullptr;
    ConvGeom gg = g;
    gg.epi_tile = (!no_tile && conv_epi_tile_ok(g, sizeof(T) == 2 ? CN_BF16 : CN_F32)) ? 1 : 0;
    if (gg.bn_part) {                                  // BN statistics sink: the LDS-staged epilogue has the hook
        if (gg.epi_tile) mark_taken(gg.bn_taken); else gg.bn_part = nullptr;
    }
    hipLaunchKernelGGL((conv_igemm_kernel<T, BN, BK>), grid, dim3(256), 0, st, gg);
}

// tile choice: BN over {32,64,128}; BK by divisibility of Ci
static void pick_tile(int Ci, int Co, int dtype, int* bn_out, int* bk_out) {
    // fewest channel blocks first (every channel block re-reads the whole pixel operand from HBM/L2, while padded MFMA
    // columns are nearly free), then the least padding
    int best = 32, bestn = (Co + 31) / 32;
    for (int bn : {64, 128}) {
        int nb = (Co + bn - 1) / bn;
        if (nb < bestn) { best = bn; bestn = nb; }
    }
    int bk = 16;
    if (dtype == CN_BF16) {
        bk = (Ci % 64 == 0) ? 64 : (Ci % 32 == 0 ? 32 : 16);
        if (best == 128 && bk == 64) bk = 32;  // keep LDS <= 40 KB/block for 128x128 tiles
    }
    *bn_out = best;
    *bk_out = bk;
}

static bool use_conv3x3(int KH, int KW, int stride, int pad, int H, int W, int OH, int OW) {
    static const bool disabled = getenv("CN_DISABLE_CONV3X3") != nullptr;   // A/B switch for profiling
    return !disabled && KH == 3 && KW == 3 && stride == 1 && pad == 1 && OH == H && OW == W;
}

extern "C" int cn_conv2d_variant(int Ci, int Co, int KH, int KW, int stride, int pad, int dtype) {
    int bn, bk;
    pick_tile(Ci, Co, dtype, &bn, &bk);
    if (use_conv3x3(KH, KW, stride, pad, 1, 1, 1, 1)) {
        int ck = dtype == CN_BF16 ? (Ci % 64 == 0 ? 64 : (Ci % 32 == 0 ? 32 : 16)) : 16;
        return 3000000 + bn * 1000 + ck;        // conv3x3s1_kernel<T, BN, CK>
    }
    if (KH == 3 && KW == 3 && stride == 2 && pad == 1 && dtype == CN_BF16 && Ci % 32 == 0 && Ci >= 32 && getenv("CN_DISABLE_CONV3X3_S2") == nullptr) {
        int b3 = 32, nb3 = (Co + 31) / 32;      // conv3x3s2_launch's tile rule
        for (int c : {64, 128}) { int nb = (Co + c - 1) / c; if (nb < nb3) { b3 = c; nb3 = nb; } }
        return 4000000 + b3 * 1000 + 32;        // conv3x3s1_kernel<bf16, BN, 32, NW, S = 2> (forward; the data gradient stays on the implicit GEMM)
    }
    return bn * 1000 + bk;                      // conv_igemm_kernel<T, BN, BK>
}

template <typename T>
static int dispatch_igemm(const ConvGeom& g, int ncls, hipStream_t st) {
    int best, bk;
    pick_tile(g.Ci, g.Co, sizeof(T) == 2 ? CN_BF16 : CN_F32, &best, &bk);
#define CN_IG(BN_, BK_) launch_igemm<T, BN_, BK_>(g, ncls, st)
    if constexpr (sizeof(T) == 2) {
        if (best == 128) { if (bk == 32) CN_IG(128, 32); else CN_IG(128, 16); }
        else if (best == 64) { if (bk == 64) CN_IG(64, 64); else if (bk == 32) CN_IG(64, 32); else CN_IG(64, 16); }
        else { if (bk == 64) CN_IG(32, 64); else if (bk == 32) CN_IG(32, 32); else CN_IG(32, 16); }
    } else {
        if (best == 128) CN_IG(128, 16); else if (best == 64) CN_IG(64, 16); else CN_IG(32, 16);
    }
#undef CN_IG
    return 0;
}

// ---- pre-affine of the input: the consumer applies the previous layer's training-mode BatchNorm (+ ReLU) ----
// cn_hooks (include/centernet_hip.h) carries the optional extras of ONE call: the input pre-affine (only the 16-input-channel kernels
// have the hook, conv_c16.hip / wgrad_c16.hip: any other shape fails with CN_EUNSUPPORTED — there is no fallback that would silently
// convolve the raw tensor), the BatchNorm statistics sink of the output and the BN-backward statistics sink of a data gradient.  The
// entry point unpacks the struct it is handed (common.h: hooks_sink / hooks_pre / hooks_bnb); nothing is remembered between calls.
extern "C" int cn_conv2d_fwd_h(const void* x, const void* wp, const float* bias, const void* residual, void* y,
                               int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co, int y_ld, int res_ld,
                               int KH, int KW, int stride, int pad, int transposed, int relu, int dtype, int out_dtype,
                               cn_hooks* hooks, void* stream) {
    const BnSink sink = hooks_sink(hooks);
    const PreAffine pre = hooks_pre(hooks);
    const BnbArm bnb = hooks_bnb(hooks);
    CN_CHECK_ARG(x && wp && y, "cn_conv2d_fwd: null pointer");
    CN_CHECK_ARG(N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && Co > 0, "cn_conv2d_fwd: bad dims");
    if (Ci % 16 != 0 || Ci <= 0) CN_UNSUPPORTED("cn_conv2d_fwd: Ci=%d must be a positive multiple of 16", Ci);
    int V = dtype == CN_F32 ? 4 : 8;
    CN_CHECK_ARG(x_ld % V == 0 && x_ld >= Ci && y_ld >= Co, "cn_conv2d_fwd: bad pitches x_ld=%d y_ld=%d", x_ld, y_ld);
    CN_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)wp & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
                     ((uintptr_t)residual & 15) == 0,
                 "cn_conv2d_fwd: pointers must be 16-byte aligned");
    if (stride != 1 && stride != 2) CN_UNSUPPORTED("cn_conv2d_fwd: stride %d", stride);
    ConvGeom g;
    memset(&g, 0, sizeof(g));
    g.x = x; g.w = wp; g.bias = bias; g.res = residual; g.y = y;
    g.N = N; g.H = H; g.W = W; g.Ci = Ci; g.x_ld = x_ld; g.OH = OH; g.OW = OW; g.Co = Co; g.y_ld = y_ld;
    g.res_ld = res_ld; g.ktot = KH * KW * Ci; g.co_pad = (Co + 31) / 32 * 32; g.relu = relu;
    g.y_f32 = (out_dtype == CN_F32);
    CN_CHECK_ARG(out_dtype == dtype || out_dtype == CN_F32, "cn_conv2d_fwd: out_dtype must be the compute dtype or fp32");
    CN_CHECK_ARG(!(residual && out_dtype != dtype), "cn_conv2d_fwd: residual needs out_dtype == dtype");
    int ncls = build_geom(g, KH, KW, stride, pad, transposed);
    if (ncls < 0) CN_UNSUPPORTED("cn_conv2d_fwd: kernel %dx%d stride %d not supported", KH, KW, stride);
    // the sink is honoured by the kernels that have the hook
    if (sink.part && dtype == CN_BF16 && out_dtype == dtype && sink.C == y_ld) { g.bn_part = sink.part; g.bn_slots = sink.slots; g.bn_taken = &hooks->bn_taken; }
    if (dtype != CN_F32 && dtype != CN_BF16) CN_CHECK_ARG(false, "cn_conv2d_fwd: bad dtype %d", dtype);
    if (bnb.part && transposed && dtype == CN_BF16 && out_dtype == dtype && bnb.C == y_ld && !sink.part) {   // honoured by the kernels that have the hook
        g.bnb_part = bnb.part; g.bnb_slots = bnb.slots; g.bnb_x = bnb.x; g.bnb_stats = bnb.stats; g.bnb_relu = bnb.relu;
        g.bnb_taken = &hooks->bnb_taken;
    }
    if (pre.ss) {                         // input pre-affine: only the 16-input-channel row-walking kernel has the hook
        CN_CHECK_ARG(pre.C == Ci, "cn_conv2d_fwd: pre-affine armed for %d channels, conv has %d", pre.C, Ci);
        g.pre_ss = pre.ss; g.pre_relu = pre.relu;
        if (transposed || KH != 3 || KW != 3 || pad != 1 || !conv_c16r_launch(g, dtype, stride, (hipStream_t)stream))
            CN_UNSUPPORTED("cn_conv2d_fwd: an input pre-affine is armed but this shape has no kernel with the hook (bf16, 3x3 / pad 1, 16 input channels)");
        CN_LAUNCH_CHECK("cn_conv2d_fwd(16 ch, pre-affine)");
        return CN_OK;
    }
    if (!transposed && dtype == CN_BF16 && KH == 3 && KW == 3 && stride == 2 && pad == 1 && Ci == 16 && OH == (H - 1) / 2 + 1 && OW == (W - 1) / 2 + 1 &&
        conv_c16r_launch(g, dtype, 2, (hipStream_t)stream)) {
        CN_LAUNCH_CHECK("cn_conv2d_fwd(16 ch, stride 2)");
        return CN_OK;
    }
    if (transposed && dtype == CN_BF16 && out_dtype == dtype && KH == 3 && KW == 3 && stride == 2 && pad == 1 && ((Ci == 32 && Co == 16) || (Ci == 64 && Co == 32 && getenv("CN_DISABLE_DGRAD3X3_S2"))) &&
        OH == 2 * H && OW == 2 * W && !bias && !residual && !relu && dgrad_s2_c32to16_launch(g, (hipStream_t)stream)) {
        CN_LAUNCH_CHECK("cn_conv2d_fwd(dgrad s2 32->16)");
        return CN_OK;
    }
    if (transposed && ncls == 4 && KH == 3 && KW == 3 && stride == 2 && pad == 1 && out_dtype == dtype && dgrad3x3s2_launch(g, dtype, (hipStream_t)stream)) {
        CN_LAUNCH_CHECK("cn_conv2d_fwd(dgrad 3x3 stride 2)");
        return CN_OK;
    }
    if (use_conv3x3(KH, KW, stride, pad, H, W, OH, OW) && conv3x3s1_launch(g, dtype, (hipStream_t)stream)) {
        CN_LAUNCH_CHECK("cn_conv2d_fwd(3x3)");
        return CN_OK;
    }
    if (KH == 1 && KW == 1 && stride == 1 && pad == 0 && OH == H && OW == W && conv1x1_stream_launch(g, dtype, (hipStream_t)stream)) {
        CN_LAUNCH_CHECK("cn_conv2d_fwd(1x1 stream)");
        return CN_OK;
    }
    if (!transposed && KH == 3 && KW == 3 && stride == 2 && pad == 1 && OH == (H - 1) / 2 + 1 && OW == (W - 1) / 2 + 1 &&
        conv3x3s2_launch(g, dtype, (hipStream_t)stream)) {
        CN_LAUNCH_CHECK("cn_conv2d_fwd(3x3 stride 2)");
        return CN_OK;
    }
    if (dtype == CN_F32) dispatch_igemm<float>(g, ncls, (hipStream_t)stream);
    else if (dtype == CN_BF16) dispatch_igemm<bf16_t>(g, ncls, (hipStream_t)stream);
    else CN_CHECK_ARG(false, "cn_conv2d_fwd: bad dtype %d", dtype);
    CN_LAUNCH_CHECK("cn_conv2d_fwd");
    return CN_OK;
}

extern "C" int cn_conv2d_fwd(const void* x, const void* wp, const float* bias, const void* residual, void* y,
                             int N, int H, int W, int Ci, int x_ld, int OH, int OW, int Co, int y_ld, int res_ld,
                             int KH, int KW, int stride, int pad, int transposed, int relu, int dtype, int out_dtype,
                             void* stream) {
    return cn_conv2d_fwd_h(x, wp, bias, residual, y, N, H, W, Ci, x_ld, OH, OW, Co, y_ld, res_ld, KH, KW, stride, pad, transposed, relu,
                           dtype, out_dtype, nullptr, stream);
}

// A 2-channel task head in one launch (heads.py:9-15 `conv3x3 -> ReLU -> conv1x1`; width_height / regression): out fp32 NCHW
// [N, 2, H, W] (ALL-ZERO at launch: the kernel adds) = conv1x1(relu(conv3x3(x) + b1)) + b2, x NHWC bf16 with 64 channels, wp1 = the
// hidden conv's weights packed with mode 1 ([Ch_pad32][9 * 64]), w2 fp32 [2][Ch], Ch a multiple of 64.  The hidden activation is never
// written.  Shapes the weight-stationary kernel does not take (H, W multiples of 16, enough tiles) -> CN_EUNSUPPORTED.
extern "C" int cn_head2_fwd(const void* x, const void* wp1, const float* b1, const float* w2, const float* b2, float* out, int N, int H,
                            int W, int Ci, int x_ld, int Ch, int dtype, void* stream) {
    CN_CHECK_ARG(x && wp1 && w2 && b2 && out && N > 0 && H > 0 && W > 0, "cn_head2_fwd: bad args");
    if (dtype != CN_BF16 || Ci != 64 || (Ch & 63) || Ch <= 0) CN_UNSUPPORTED("cn_head2_fwd: bf16, 64 input channels, hidden width a multiple of 64");
    ConvGeom g;
    memset(&g, 0, sizeof(g));
    g.x = x; g.w = wp1; g.bias = b1; g.y = out;
    g.N = N; g.H = H; g.W = W; g.Ci = Ci; g.x_ld = x_ld; g.OH = H; g.OW = W; g.Co = Ch; g.y_ld = Ch;
    g.ktot = 9 * Ci; g.co_pad = (Ch + 31) / 32 * 32; g.relu = 1;
    g.head_w = w2; g.head_b = b2; g.head_nc = 2;
    if (build_geom(g, 3, 3, 1, 1, 0) < 0 || !conv3x3_ws_launch(g, dtype, (hipStream_t)stream))
        CN_UNSUPPORTED("cn_head2_fwd: shape not handled by the weight-stationary kernel (H, W multiples of 16, >= 4 tiles per workgroup)");
    CN_LAUNCH_CHECK("cn_head2_fwd");
    return CN_OK;
}

bool conv1x1_stream_nchw_launch(const ConvGeom& g, int dtype, hipStream_t st);

extern "C" int cn_conv1x1_nchw_fwd(const void* x, const void* wp, const float* bias, float* y, int N, int H, int W, int Ci, int x_ld,
                                   int Co, int dtype, void* stream) {
    CN_CHECK_ARG(x && wp && y && N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0 && x_ld >= Ci, "cn_conv1x1_nchw_fwd: bad args");
    ConvGeom g;
    memset(&g, 0, sizeof(g));
    g.x = x; g.w = wp; g.bias = bias; g.y = y;
    g.N = N; g.H = H; g.W = W; g.Ci = Ci; g.x_ld = x_ld; g.OH = H; g.OW = W; g.Co = Co; g.y_ld = Co;
    g.ktot = Ci; g.co_pad = (Co + 31) / 32 * 32; g.y_f32 = 1;
    if (!conv1x1_stream_nchw_launch(g, dtype, (hipStream_t)stream))
        CN_UNSUPPORTED("cn_conv1x1_nchw_fwd: shape not handled (bf16, Ci == 256, Co <= 128, H*W %% 32 == 0, >= 64 Ki pixels): run cn_conv2d_fwd + cn_nhwc_to_nchw");
    CN_LAUNCH_CHECK("cn_conv1x1_nchw_fwd");
    return CN_OK;
}

// 1x1 / stride 1 conv over the channel concatenation of up to CN_MAX_SRC NHWC tensors (each contiguous: pitch = its channel count)
extern "C" int cn_conv1x1_cat_fwd_h(const void* x0, const void* x1, const void* x2, const void* x3, const void* x4, const void* x5,
                                    int c0, int c1, int c2, int c3, int c4, int c5, int nsrc, const void* wp, const float* bias,
                                    const void* residual, void* y, int N, int H, int W, int Co, int y_ld, int res_ld, int relu,
                                    int dtype, cn_hooks* hooks, void* stream) {
    const BnSink sink = hooks_sink(hooks);
    const void* xs[CN_MAX_SRC] = {x0, x1, x2, x3, x4, x5};
    const int cs[CN_MAX_SRC] = {c0, c1, c2, c3, c4, c5};
    CN_CHECK_ARG(nsrc >= 1 && nsrc <= CN_MAX_SRC && wp && y && N > 0 && H > 0 && W > 0 && Co > 0 && y_ld >= Co,
                 "cn_conv1x1_cat_fwd: bad args");
    CN_CHECK_ARG(dtype == CN_F32 || dtype == CN_BF16, "cn_conv1x1_cat_fwd: bad dtype %d", dtype);
    ConvGeom g;
    memset(&g, 0, sizeof(g));
    int k = 0, div = 64;
    for (int i = 0; i < nsrc; ++i) {
        CN_CHECK_ARG(xs[i] && (((uintptr_t)xs[i]) & 15) == 0 && cs[i] > 0 && cs[i] % 16 == 0,
                     "cn_conv1x1_cat_fwd: source %d must be a 16-byte aligned tensor with a multiple of 16 channels", i);
        g.xs[i] = xs[i]; g.xs_c[i] = cs[i]; g.xs_k0[i] = k;
        k += cs[i];
        while (cs[i] % div) div >>= 1;          // the K slice has to divide every source
    }
    CN_CHECK_ARG((((uintptr_t)wp | (uintptr_t)y | (uintptr_t)residual) & 15) == 0, "cn_conv1x1_cat_fwd: pointers must be 16-byte aligned");
    g.nsrc = nsrc;
    g.x = xs[0]; g.w = wp; g.bias = bias; g.res = residual; g.y = y;
    g.N = N; g.H = H; g.W = W; g.Ci = k; g.x_ld = cs[0]; g.OH = H; g.OW = W; g.Co = Co; g.y_ld = y_ld; g.res_ld = res_ld;
    g.ktot = k; g.co_pad = (Co + 31) / 32 * 32; g.relu = relu;
    const int ncls = build_geom(g, 1, 1, 1, 0, 0);
    if (sink.part && dtype == CN_BF16 && sink.C == y_ld) { g.bn_part = sink.part; g.bn_slots = sink.slots; g.bn_taken = &hooks->bn_taken; }
    // tile choice of dispatch_igemm with the K slice bounded by the smallest source granule
    ConvGeom gp = g;
    gp.Ci = div;                                // pick_tile() only looks at divisibility
    int bn, bk;
    pick_tile(div, Co, dtype, &bn, &bk);
    hipStream_t st = (hipStream_t)stream;
#define CN_IGC(T_, BN_, BK_) launch_igemm<T_, BN_, BK_>(g, ncls, st)
    if (dtype == CN_BF16) {
        if (bn == 128) { if (bk >= 32) CN_IGC(bf16_t, 128, 32); else CN_IGC(bf16_t, 128, 16); }
        else if (bn == 64) { if (bk == 64) CN_IGC(bf16_t, 64, 64); else if (bk == 32) CN_IGC(bf16_t, 64, 32); else CN_IGC(bf16_t, 64, 16); }
        else { if (bk == 64) CN_IGC(bf16_t, 32, 64); else if (bk == 32) CN_IGC(bf16_t, 32, 32); else CN_IGC(bf16_t, 32, 16); }
    } else {
        if (bn == 128) CN_IGC(float, 128, 16); else if (bn == 64) CN_IGC(float, 64, 16); else CN_IGC(float, 32, 16);
    }
#undef CN_IGC
    CN_LAUNCH_CHECK("cn_conv1x1_cat_fwd");
    return CN_OK;
}
extern "C" int cn_conv1x1_cat_fwd(const void* x0, const void* x1, const void* x2, const void* x3, const void* x4, const void* x5,
                                  int c0, int c1, int c2, int c3, int c4, int c5, int nsrc, const void* wp, const float* bias,
                                  const void* residual, void* y, int N, int H, int W, int Co, int y_ld, int res_ld, int relu,
                                  int dtype, void* stream) {
    return cn_conv1x1_cat_fwd_h(x0, x1, x2, x3, x4, x5, c0, c1, c2, c3, c4, c5, nsrc, wp, bias, residual, y, N, H, W, Co, y_ld, res_ld, relu,
                                dtype, nullptr, stream);
}

// dom / dx_far of the DCNv2 backward, fused into the GEMM dcol = dY x W^T (wpd2 = cn_pack_weight mode 2: [9*Ci][Co_pad16])
extern "C" int cn_dcn_bwd_dom_slabs(int Ci, int dy_ld, int dtype) {
    static const bool disabled = getenv("CN_DISABLE_DOM_TILE") != nullptr;
    return (!disabled && dtype == CN_BF16 && Ci % 64 == 0 && (dy_ld == 64 || dy_ld == 128)) ? Ci / 64 : 1;
}

extern "C" int cn_dcn_bwd_dom(const void* dy, const void* wpd2, const void* x, const float* om, float* dom, int dom_slabs,
                              float* dx_far, int* far_flag, int N, int H, int W, int Ci, int Co, int dy_ld, int x_ld, int om_ld,
                              int dtype, void* stream) {
    CN_CHECK_ARG(dy && wpd2 && x && om && dom && dx_far && N > 0 && H > 0 && W > 0, "cn_dcn_bwd_dom: bad args");
    if (Ci % 32 != 0 || dy_ld % 16 != 0) CN_UNSUPPORTED("cn_dcn_bwd_dom: Ci=%d must be a multiple of 32, dy_ld=%d of 16", Ci, dy_ld);
    CN_CHECK_ARG(om_ld >= 27 && x_ld >= Ci, "cn_dcn_bwd_dom: bad pitches");
    CN_CHECK_ARG(dom_slabs == 1 || dom_slabs == cn_dcn_bwd_dom_slabs(Ci, dy_ld, dtype) ||
                     (dom_slabs == 0 && Ci == 64 && cn_dcn_bwd_dom_slabs(Ci, dy_ld, dtype) == 1 && dtype == CN_BF16 && (dy_ld == 64 || dy_ld == 128)),
                 "cn_dcn_bwd_dom: dom_slabs=%d (ask cn_dcn_bwd_dom_slabs; 0 = direct bf16 result, Ci == 64 only)", dom_slabs);
    if (dtype == CN_BF16 && dcn_dom_bm_launch(dy, wpd2, x, om, dom, dom_slabs, dx_far, far_flag, N, H, W, Ci, dy_ld, x_ld, om_ld, (hipStream_t)stream)) {
        CN_LAUNCH_CHECK("cn_dcn_bwd_dom(bm)");
        return CN_OK;
    }
    if (dtype == CN_BF16 && dcn_bwd_dom_tile_launch(dy, wpd2, x, om, dom, dom_slabs, dx_far, far_flag, N, H, W, Ci, Co, dy_ld, x_ld, om_ld, (hipStream_t)stream)) {
        CN_LAUNCH_CHECK("cn_dcn_bwd_dom(tile)");
        return CN_OK;
    }
    if (dom_slabs == 0) CN_UNSUPPORTED("cn_dcn_bwd_dom: the direct bf16 result needs the tile kernel (CN_DISABLE_DOM_TILE is set?)");
    ConvGeom g;
    memset(&g, 0, sizeof(g));
    g.x = dy; g.w = wpd2; g.y = dom;
    g.N = N; g.H = H; g.W = W; g.Ci = dy_ld; g.x_ld = dy_ld; g.OH = H; g.OW = W; g.Co = 9 * Ci; g.y_ld = 9 * Ci;
    g.ktot = dy_ld; g.co_pad = (9 * Ci + 31) / 32 * 32;
    (void)Co;
    build_geom(g, 1, 1, 1, 0, 0);
    g.dcn_x = x; g.dcn_om = om; g.dcn_dom = dom; g.dcn_far = dx_far; g.far_flag = far_flag;
    g.dcn_Ci = Ci; g.dcn_H = H; g.dcn_W = W; g.dcn_xld = x_ld; g.dcn_omld = om_ld;
    hipStream_t st = (hipStream_t)stream;
    const int64_t M = (int64_t)N * H * W;
    int bn = Ci % 128 == 0 ? 128 : (Ci % 64 == 0 ? 64 : 32);
    if (bn == 128 && dtype == CN_BF16 && dy_ld % 32 != 0) bn = 64;   // the 128-wide tile needs the BK=32 pipeline buffers for its dcol tile
    dim3 grid(cdiv(M, 128), cdiv(9 * Ci, bn), 1);
#define CN_DOM(T, BN_, BK_) hipLaunchKernelGGL((conv_igemm_kernel<T, BN_, BK_, true>), grid, dim3(256), 0, st, g)
    if (dtype == CN_BF16) {
        const int bk = dy_ld % 32 == 0 ? 32 : 16;     // BK=32 keeps the LDS footprint at ~31-41 KB -> 4-5 workgroups per CU
        if (bn == 128) { CN_DOM(bf16_t, 128, 32); }
        else if (bn == 64) { if (bk == 64) CN_DOM(bf16_t, 64, 64); else if (bk == 32) CN_DOM(bf16_t, 64, 32); else CN_DOM(bf16_t, 64, 16); }
        else { if (bk == 64) CN_DOM(bf16_t, 32, 64); else if (bk == 32) CN_DOM(bf16_t, 32, 32); else CN_DOM(bf16_t, 32, 16); }
    } else if (dtype == CN_F32) {
        if (bn == 128) CN_DOM(float, 128, 16); else if (bn == 64) CN_DOM(float, 64, 16); else CN_DOM(float, 32, 16);
    } else {
        CN_CHECK_ARG(false, "cn_dcn_bwd_dom: bad dtype %d", dtype);
    }
#undef CN_DOM
    CN_LAUNCH_CHECK("cn_dcn_bwd_dom");
    return CN_OK;
}

// Which kernel template a DCNv2 entry point dispatches to for bf16 activations with the standard pitches (x_ld = Ci, y_ld = dy_ld = Co,
// om_ld = 32); bench.py names its per-kernel roofline rows with it so that they agree with rocprofv3's kernel names.
//   entry 0 = cn_dcn_fwd:     1000000 + NCB = dcn_fwd_bm_kernel<NCB>; 2000000 + BN = dcn_fwd_tile_kernel<BN>; 3000000 + BN*1000 + CK = dcn_fwd_kernel<bf16,BN,CK>
//   entry 1 = cn_dcn_wgrad:   1 = dcn_wgrad_bm_kernel; BMW*1000000 + BNW*1000 + TAPS = dcn_wgrad_kernel<BMW,BNW,TAPS>
//   entry 2 = cn_dcn_bwd_dom: 1000000 + COP = dcn_dom_bm_kernel<COP>; COP = dcn_bwd_dom_kernel<COP> (0: the generic GEMM-epilogue kernel)
//   entry 3 = cn_dcn_bwd_dx:  1000000 + NCB = dcn_dx_bm_kernel<NCB>; 3000000 + BN*1000 + CK = dcn_bwd_dx_kernel<bf16,BN,CK>
//   entry 0 with a map size (cn_dcn_variant_hw): 5000000 = dcn_fwd_b2_kernel (dcn_b2.hip)
extern "C" int cn_dcn_variant_hw(int entry, int Ci, int Co, int H, int W) {
    const int co32 = (Co + 31) / 32 * 32;
    if (entry == 0) {
        if (dcn_fwd_b2_shape_ok(Ci, Ci, Co, Co, 32, H, W)) return 5000000;     // dcn_fwd_b2_kernel
        if (dcn_fwd_bm_shape_ok(Ci, Ci, Co, Co, 32)) return 1000000 + Co / 32;
        if (dcn_fwd_tile_shape_ok(Ci, Ci, Co, Co)) return 2000000 + (Co % 128 == 0 ? 128 : 64);
        int bn = 32, bw = co32;
        for (int c : {64, 128}) {
            const int w = (co32 + c - 1) / c * c;
            if (w <= bw) { bn = c; bw = w; }
        }
        return 3000000 + bn * 1000 + (Ci % 64 == 0 ? 64 : (Ci % 32 == 0 ? 32 : 16));
    }
    if (entry == 1) {
        static const bool taps3 = getenv("CN_DCN_WGRAD_TAPS3") != nullptr;
        if (dcn_wgrad_bm_shape_ok(Ci, Ci, Co, Co, 32)) return 1;            // dcn_wgrad_bm_kernel
        return Co > 64 ? 128064003 : (taps3 ? 64064003 : 64064009);
    }
    if (entry == 2 && dcn_dom_bm_shape_ok(Ci, Co, Ci, 32)) return 1000000 + Co;                        // dcn_dom_bm_kernel<COP>
    if (entry == 2) return (Ci % 64 == 0 && cn_dcn_bwd_dom_slabs(128, Co, CN_BF16) == 2) ? Co : 0;   // slabs(128, .) == 2 <=> the tile kernel takes this dy_ld
    if (entry == 3) {
        if (dcn_dx_bm_shape_ok(Ci, Co, 32)) return 1000000 + (Ci > 64 ? 2 : Ci / 32);      // wider dx: 64-channel blocks on grid z, still <2>
        const int bn = Ci % 128 == 0 ? 128 : (Ci % 64 == 0 ? 64 : 32);
        return 3000000 + bn * 1000 + (Co % 64 == 0 ? 64 : (Co % 32 == 0 ? 32 : 16));
    }
    return -1;
}
extern "C" int cn_dcn_variant(int entry, int Ci, int Co) { return cn_dcn_variant_hw(entry, Ci, Co, 0, 0); }

// Fused DCNv2 forward (sampling -> LDS -> MFMA, dcn_fused.hip): y = act(bias + sum_k W_k * mask_k * bilinear_k(x) [+ residual]).
// wp = cn_pack_weight mode 1 of the layer weight ([Co_pad32][tap*Ci + ci]); om fp32 [P][om_ld]; bias fp32[Co] nullable.
extern "C" int cn_dcn_fwd_h(const void* x, const float* om, const void* wp, const float* bias, void* y,
                            int N, int H, int W, int Ci, int x_ld, int Co, int y_ld, int om_ld, int relu, int dtype, cn_hooks* hooks, void* stream) {
    const BnSink sink = hooks_sink(hooks);       // BatchNorm statistics sink of this call (the kernels with an LDS-staged epilogue have the hook)
    int* const taken = hooks ? &hooks->bn_taken : nullptr;
    CN_CHECK_ARG(x && om && wp && y && N > 0 && H > 0 && W > 0 && Ci > 0 && Co > 0, "cn_dcn_fwd: bad args");
    if (Ci % 16 != 0) CN_UNSUPPORTED("cn_dcn_fwd: Ci=%d must be a multiple of 16", Ci);
    if (N > 65535) CN_UNSUPPORTED("cn_dcn_fwd: batch %d", N);
    if (dtype != CN_F32 && dtype != CN_BF16) CN_CHECK_ARG(false, "cn_dcn_fwd: bad dtype %d", dtype);
    CN_CHECK_ARG(om_ld >= 27 && x_ld >= Ci && y_ld >= Co, "cn_dcn_fwd: bad pitches");
    ConvGeom g;
    memset(&g, 0, sizeof(g));
    g.x = x; g.w = wp; g.bias = bias; g.y = y;
    g.N = N; g.H = H; g.W = W; g.Ci = Ci; g.x_ld = x_ld; g.OH = H; g.OW = W; g.Co = Co; g.y_ld = y_ld;
    g.ktot = 9 * Ci; g.co_pad = (Co + 31) / 32 * 32; g.relu = relu; g.so = 1; g.sm = 1;
    g.dcn_om = om; g.dcn_omld = om_ld;
    const bool sink_ok = sink.part && dtype == CN_BF16 && sink.C == y_ld;
    if (dtype == CN_BF16 && dcn_fwd_b2_launch(x, om, wp, bias, y, N, H, W, Ci, x_ld, Co, y_ld, om_ld, g.ktot, relu, sink_ok ? sink.part : nullptr,
                                              sink.slots, taken, (hipStream_t)stream)) {
        CN_LAUNCH_CHECK("cn_dcn_fwd(b2)");
        return CN_OK;
    }
    if (dtype == CN_BF16 && dcn_fwd_bm_launch(x, om, wp, bias, y, N, H, W, Ci, x_ld, Co, y_ld, om_ld, g.ktot, relu, sink_ok ? sink.part : nullptr,
                                              sink.slots, taken, (hipStream_t)stream)) {
        CN_LAUNCH_CHECK("cn_dcn_fwd(bm)");
        return CN_OK;
    }
    if (!(dtype == CN_BF16 && dcn_fwd_tile_launch(x, om, wp, bias, y, N, H, W, Ci, x_ld, Co, y_ld, om_ld, g.ktot, relu, sink_ok ? sink.part : nullptr, sink.slots,
                                                  taken, (hipStream_t)stream))) {
        if (sink_ok) { g.bn_part = sink.part; g.bn_slots = sink.slots; g.bn_taken = taken; }      // honoured where the gather kernel runs its LDS-staged epilogue
        dcn_fwd_launch(g, dtype, (hipStream_t)stream);
    }
    CN_LAUNCH_CHECK("cn_dcn_fwd");
    return CN_OK;
}
extern "C" int cn_dcn_fwd(const void* x, const float* om, const void* wp, const float* bias, void* y,
                          int N, int H, int W, int Ci, int x_ld, int Co, int y_ld, int om_ld, int relu, int dtype, void* stream) {
    return cn_dcn_fwd_h(x, om, wp, bias, y, N, H, W, Ci, x_ld, Co, y_ld, om_ld, relu, dtype, nullptr, stream);
}

// dx of the DCNv2 backward (adjoint-gather + contraction, dcn_fused.hip).  wpd0 = cn_pack_weight mode 0 of the layer weight
// ([Ci rows][tap*Co_pad16 + co]); dx_far = fp32 [P][Ci] from cn_dcn_bwd_dom (added in the epilogue); dx in `dtype`.
extern "C" int cn_dcn_bwd_dx(const void* dy, const void* wpd0, const float* om, float* dx_far, int* far_flag, void* dx,
                             int N, int H, int W, int Ci, int dy_ld, int om_ld, int dtype, void* stream) {
    CN_CHECK_ARG(dy && wpd0 && om && dx_far && dx && N > 0 && H > 0 && W > 0, "cn_dcn_bwd_dx: bad args");
    if (Ci % 32 != 0 || dy_ld % 16 != 0) CN_UNSUPPORTED("cn_dcn_bwd_dx: Ci=%d must be a multiple of 32, dy_ld=%d of 16", Ci, dy_ld);
    if (N > 65535) CN_UNSUPPORTED("cn_dcn_bwd_dx: batch %d", N);
    if (dtype != CN_F32 && dtype != CN_BF16) CN_CHECK_ARG(false, "cn_dcn_bwd_dx: bad dtype %d", dtype);
    if (dtype == CN_BF16 && dcn_dx_bm_launch(dy, wpd0, om, dx_far, far_flag, dx, N, H, W, Ci, dy_ld, om_ld, (hipStream_t)stream)) {
        CN_LAUNCH_CHECK("cn_dcn_bwd_dx(bm)");
        return CN_OK;
    }
    ConvGeom g;
    memset(&g, 0, sizeof(g));
    g.x = dy; g.w = wpd0; g.y = dx;
    g.N = N; g.H = H; g.W = W; g.Ci = dy_ld; g.x_ld = dy_ld; g.OH = H; g.OW = W; g.Co = Ci; g.y_ld = Ci;
    g.ktot = 9 * dy_ld; g.co_pad = (Ci + 31) / 32 * 32; g.so = 1; g.sm = 1;
    g.dcn_om = om; g.dcn_omld = om_ld; g.res32 = dx_far; g.res32_ld = Ci; g.far_flag = far_flag;
    dcn_bwd_dx_launch(g, dtype, (hipStream_t)stream);
    CN_LAUNCH_CHECK("cn_dcn_bwd_dx");
    return CN_OK;
}

// ------------------------------------------------------------------------------------------------ packing
// mode 0: rows = B, k = t*inner_pad + a   (Wp[b][t*ip + a] = W[a][b][t])
// mode 1: rows = A, k = t*inner_pad + b   (Wp[a][t*ip + b] = W[a][b][t])
// mode 2: rows = (t, b) = t*B + b, k = a  (Wp[t*B + b][a]  = W[a][b][t])   inner_pad pads k
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ wp, int A, int B,
                                                          int taps, int mode, int rows_pad, int inner_pad,
                                                          const float* __restrict__ row_scale) {
    const int64_t ktot = mode == 2 ? inner_pad : (int64_t)taps * inner_pad;
    const int64_t total = (int64_t)rows_pad * ktot;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / ktot);
        const int k = (int)(i - (int64_t)r * ktot);
        int a = -1, b = -1, t = 0;
        if (mode == 2) {
            t = r / B; b = r - t * B; a = k;
            if (t >= taps) b = -1;
        } else {
            t = k / inner_pad;
            const int c = k - t * inner_pad;
            if (mode == 1) { a = r; b = c; } else { a = c; b = r; }
        }
        float v = 0.f;
        if (a >= 0 && a < A && b >= 0 && b < B) {
            v = w[((int64_t)a * B + b) * taps + t];
            if (row_scale) v *= row_scale[r];
        }
        Elem<T>::st(wp + i, v);
    }
}

extern "C" int cn_pack_weight(const float* w, void* wp, int A, int B, int KH, int KW, int mode, int rows_pad, int inner_pad,
                              const float* row_scale, int dtype, void* stream) {
    CN_CHECK_ARG(w && wp && A > 0 && B > 0 && KH > 0 && KW > 0 && mode >= 0 && mode <= 2, "cn_pack_weight: bad args");
    const int taps = KH * KW;
    const int rows = mode == 0 ? B : (mode == 1 ? A : taps * B);
    const int inner = mode == 0 ? A : (mode == 1 ? B : A);
    CN_CHECK_ARG(rows_pad >= rows && inner_pad >= inner, "cn_pack_weight: rows_pad %d < %d or inner_pad %d < %d", rows_pad, rows,
                 inner_pad, inner);
    int64_t total = (int64_t)rows_pad * (mode == 2 ? inner_pad : (int64_t)taps * inner_pad);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(pack_weight_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w,
                                                   (T*)wp, A, B, taps, mode, rows_pad, inner_pad, row_scale));
    CN_LAUNCH_CHECK("cn_pack_weight");
    return CN_OK;
}

// One launch packs every weight operand of a step.  Table record (10 int64): w, wp, A, B, taps, mode, rows_pad, inner_pad, first_block,
// AA | BB << 8 | tiles_b << 16.  A workgroup owns a tile of AA x BB (a, b) pairs with all their taps: the fp32 source
// w[a][b0 .. b0+BB-1][0 .. taps-1] is one contiguous run per a (coalesced loads into LDS), the packed rows leave as 16-byte vectors of 8
// consecutive INNER elements (mode 1: b, tiles 4 x 64; modes 0 / 2: a, tiles 64 x 4, 64 x 16 for 1x1 weights) read back from LDS with the
// (a, b, t) -> (row, tap, inner) permutation.  The tiles cover the padded ranges, so every element of wp (padding = 0) is written by
// every launch; mode 2's padding rows behind tap*B + b have one extra workgroup per record.  (History: output-element driven with a
// stride of `taps` floats per lane, 9x over-fetch: 266 us; a thread per (row, 8 inner) pair walking its taps, 64 cache lines per
// wave load: 207 us for DLA-34's 2 x 20 M elements — the first kernel of every step.)
#define PACK_LDS_FLOATS 4096
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_batch_kernel(const int64_t* __restrict__ tab, int n, const int* __restrict__ block_record) {
    __shared__ float L[PACK_LDS_FLOATS];
    int lo = 0, hi = n - 1;                                 // last record whose first_block <= blockIdx.x
    if (block_record) lo = block_record[blockIdx.x];        // (eight dependent L2 round trips per workgroup otherwise: 3 of its ~4 us)
    else
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tab[mid * 10 + 8] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
        }
    const int64_t* e = tab + lo * 10;
    const float* __restrict__ w = reinterpret_cast<const float*>(e[0]);
    T* __restrict__ wp = reinterpret_cast<T*>(e[1]);
    const int A = (int)e[2], B = (int)e[3], taps = (int)e[4], mode = (int)e[5], rows_pad = (int)e[6], ip = (int)e[7];
    const int AA = (int)(e[9] & 255), BB = (int)((e[9] >> 8) & 255), tiles_b = (int)(e[9] >> 16);
    const int tile = (int)((int64_t)blockIdx.x - e[8]);
    const int tid = threadIdx.x;
    auto put8 = [&](T* dst, const float (&v)[8]) {
        if constexpr (sizeof(T) == 2) st16(dst, make_uint4(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])));
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[j] = v[j];
        }
    };
    const int a_range = mode == 1 ? rows_pad : ip, b_range = mode == 1 ? ip : (mode == 0 ? rows_pad : B);
    const int tiles_a = (a_range + AA - 1) / AA;
    if (tile >= tiles_a * tiles_b) {                        // mode 2: the zero rows behind (tap, b)
        const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int ipv = ip / 8, nv = (rows_pad - taps * B) * ipv;
        for (int i = tid; i < nv; i += 256) put8(wp + (int64_t)taps * B * ip + (int64_t)i * 8, z);
        return;
    }
    const int ta = tile / tiles_b, tb = tile - ta * tiles_b;
    const int a0 = ta * AA, b0 = tb * BB;
    const int run = BB * taps;                              // LDS pitch of one a
    const int bval = B - b0 < BB ? (B - b0 > 0 ? B - b0 : 0) : BB;
    // exact quotients by the two small run-time divisors without integer division: q = umulhi(e, ceil(2^32 / d)) for e < 2^12, d <= 2^12
    const unsigned m_run = (unsigned)((0x100000000ull + (unsigned)run - 1) / (unsigned)run);
    const unsigned m_taps = taps > 1 ? (unsigned)((0x100000000ull + (unsigned)taps - 1) / (unsigned)taps) : 0u;
    auto div_taps = [&](int x) { return taps > 1 ? (int)__umulhi((unsigned)x, m_taps) : x; };
    const int64_t src0 = ((int64_t)a0 * B + (bval > 0 ? b0 : 0)) * taps;
    const int lim = bval * taps;
    const int total = AA * run;
#pragma unroll 1
    for (int base = tid; base < total; base += 256 * 8) {   // eight branch-free loads in flight per thread, then the LDS stores
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int el = base + u * 256;
            const int al = (int)__umulhi((unsigned)el, m_run), off = el - al * run;
            const bool ok = el < total && a0 + al < A && off < lim;
            const float x = w[ok ? src0 + (int64_t)al * B * taps + off : 0];
            v[u] = ok ? x : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (base + u * 256 < total) L[base + u * 256] = v[u];
    }
    __syncthreads();
    if (mode == 1) {                                        // wp[a][t * ip + b]: vectors of 8 b
        const int bv = BB / 8, nvec = AA * taps * bv;
        for (int i = tid; i < nvec; i += 256) {
            const int v = i % bv, at = i / bv, al = div_taps(at), t = at - al * taps;      // bv is a power of two
            const int a = a0 + al, b = b0 + 8 * v;
            if (a >= rows_pad || b >= ip) continue;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = L[al * run + (8 * v + j) * taps + t];
            put8(wp + (int64_t)a * taps * ip + (int64_t)t * ip + b, o);
        }
    } else {                                                // mode 0: wp[b][t * ip + a]; mode 2: wp[t * B + b][a]: vectors of 8 a
        const int av = AA / 8, nvec = BB * taps * av;
        for (int i = tid; i < nvec; i += 256) {
            const int v = i % av, bt = i / av, bl = div_taps(bt), t = bt - bl * taps;      // av = 8
            const int a = a0 + 8 * v, b = b0 + bl;
            if (a >= ip || b >= b_range) continue;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = L[(8 * v + j) * run + bl * taps + t];
            put8(mode == 0 ? wp + (int64_t)b * taps * ip + (int64_t)t * ip + a : wp + ((int64_t)t * B + b) * ip + a, o);
        }
    }
}

// tile shape of a record (the caller builds the table with it): -> AA | BB << 8, 0 when the record does not fit (taps > 64)
extern "C" int cn_pack_weight_tile(int taps, int mode) {
    if (taps < 1 || taps > PACK_LDS_FLOATS / 64 || mode < 0 || mode > 2) return 0;
    int AA, BB;
    if (mode == 1) { BB = 64; AA = PACK_LDS_FLOATS / (64 * taps); AA = AA > 4 ? 4 : AA; }
    else { AA = 64; BB = PACK_LDS_FLOATS / (64 * taps); const int cap = taps == 1 ? 16 : 4; BB = BB > cap ? cap : BB; }
    return AA | (BB << 8);
}

extern "C" int cn_pack_weight_batch(const void* table, int n_entries, int n_blocks, const int* block_record, int dtype, void* stream) {
    CN_CHECK_ARG(table && n_entries > 0 && n_blocks > 0, "cn_pack_weight_batch: bad args");
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(pack_weight_batch_kernel<T>, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream,
                                                   (const int64_t*)table, n_entries, block_record));
    CN_LAUNCH_CHECK("cn_pack_weight_batch");
    return CN_OK;
}

// dw[a][b][t] = dwp[a][t*inner_pad + b]   (inverse of mode 1)
__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int A,
                                                           int B, int taps, int inner_pad, int accumulate) {
    const int64_t total = (int64_t)A * B * taps;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int t = (int)(i % taps);
        int64_t ab = i / taps;
        int b = (int)(ab % B), a = (int)(ab / B);
        const float v = dwp[(int64_t)a * taps * inner_pad + (int64_t)t * inner_pad + b];
        dw[i] = accumulate ? dw[i] + v : v;
    }
}

// 1x1 weights, column block: dw[a][col0 + b] (+)= dwp[a][b] for b < B, dw rows of length dw_ld
__global__ __launch_bounds__(256) void unpack_wgrad_cols_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int A, int B,
                                                                int inner_pad, int dw_ld, int accumulate) {
    const int64_t total = (int64_t)A * B;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i % B), a = (int)(i / B);
        const float v = dwp[(int64_t)a * inner_pad + b];
        float* d = dw + (int64_t)a * dw_ld + b;
        *d = accumulate ? *d + v : v;
    }
}

extern "C" int cn_unpack_wgrad_cols(const float* dwp, float* dw, int A, int B, int inner_pad, int dw_ld, int accumulate, void* stream) {
    CN_CHECK_ARG(dwp && dw && A > 0 && B > 0 && inner_pad >= B && dw_ld >= B, "cn_unpack_wgrad_cols: bad args");
    int64_t total = (int64_t)A * B;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(unpack_wgrad_cols_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dwp, dw, A, B, inner_pad, dw_ld, accumulate);
    CN_LAUNCH_CHECK("cn_unpack_wgrad_cols");
    return CN_OK;
}

extern "C" int cn_unpack_wgrad(const float* dwp, float* dw, int A, int B, int KH, int KW, int inner_pad, int accumulate,
                               void* stream) {
    CN_CHECK_ARG(dwp && dw && A > 0 && B > 0 && inner_pad >= B, "cn_unpack_wgrad: bad args");
    int64_t total = (int64_t)A * B * KH * KW;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dwp, dw, A, B, KH * KW, inner_pad, accumulate);
    CN_LAUNCH_CHECK("cn_unpack_wgrad");
    return CN_OK;
}
