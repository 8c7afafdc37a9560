// Data gradient of the 3x3 / stride-2 / pad-1 convolutions that open DLA levels 3-5 (64 -> 128 @128^2 -> 64^2, 128 -> 256, 256 -> 512;
// pose_dla_dcn.py:303-312): dx[2r + ph][2c + pw] = sum over the taps of parity class (ph, pw) of W[kh][kw]^T dy[r + dh][c + dw], dh, dw in
// {0, 1} — one tap for class (0,0), two for (0,1) and (1,0), four for (1,1).  The implicit GEMM ran one launch plane per class with
// K = taps x Co as short as 128 and re-read dy for every tap: 5-12 % of the MFMA peak (221 / 127 / 113 us at batch 64).  Here the
// halo-tile skeleton of conv3x3s1_kernel: a workgroup owns an 8 x 16 tile of dy POSITIONS, the (8+1) x (16+1) dy halo tile of a
// channel slice is staged once in LDS, the nine taps run back to back (their weight slices through the same register ring / LDS
// double buffer) and accumulate into the accumulator set of their class; four LDS-staged epilogues scatter the classes to their
// output pixels (residual / shared-gradient add and channel padding as in conv_epilogue_tile).
#include "conv_common.h"
#include <stdlib.h>

#define D2_TH 8
#define D2_TW 16

template <int BN, int CK, int NW>
__global__ __launch_bounds__(NW * 64) void dgrad3x3s2_kernel(const ConvGeom g) {
    CN_MAIN_PRIO_SET();
    typedef bf16_t T;
    constexpr int NT = NW * 64;
    constexpr int BM = D2_TH * D2_TW;
    constexpr int HW_ = D2_TW + 1, HH_ = D2_TH + 1, HP = HH_ * HW_;
    constexpr int VEC = 8, PITCH = CK + Mma<T>::PAD, VPR = CK / VEC;
    constexpr int A_VECS = HP * VPR, A_PASS = (A_VECS + NT - 1) / NT;
    constexpr int B_VECS = BN * VPR, B_PASS = (B_VECS + NT - 1) / NT;
    constexpr int WGN = (BN >= 64) ? 2 : 1, WGM = NW / WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 32, NJ = WN / 32;
    constexpr int KSTEPS = CK / Mma<T>::KSTEP;
    constexpr int RPAD = ((16 - (HW_ * PITCH / 8) % 16) % 16) * 8;          // halo row pitch = 0 mod 16 slots (see conv3x3s1_kernel)
    constexpr int RP = HW_ * PITCH + RPAD;
    constexpr int MAIN_ELEMS = HH_ * RP + 2 * BN * PITCH;
    constexpr int EPI_ELEMS = WGM * 32 * (BN + 4) * 2;
    __shared__ __attribute__((aligned(16))) T lds[MAIN_ELEMS > EPI_ELEMS ? MAIN_ELEMS : EPI_ELEMS];
    T* const As = lds;
    T* const Bs = lds + HH_ * RP;
    // stage s of a channel slice = tap ST[s] of class SC[s] (classes in build_geom's order: (ph, pw) = (0,0), (0,1), (1,0), (1,1))
    constexpr int SC[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3}, ST[9] = {0, 0, 1, 0, 1, 0, 1, 2, 3};

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_w = (g.W + D2_TW - 1) / D2_TW;
    const int th0 = (blockIdx.x / tiles_w) * D2_TH, tw0 = (blockIdx.x % tiles_w) * D2_TW;
    const int n0 = blockIdx.y * BN;
    const int n = blockIdx.z;
    const int wm = (wave / WGN) * WM, wn = (wave % WGN) * WN;
    const T* __restrict__ X = reinterpret_cast<const T*>(g.x) + (int64_t)n * g.H * g.W * g.x_ld;
    const T* __restrict__ Wp = reinterpret_cast<const T*>(g.w);

    int hbase[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = wm + i * 32 + (lane & 31);
        hbase[i] = (m / D2_TW) * RP + (m % D2_TW) * PITCH;
    }
    f32x16_t acc[4][NJ][MI];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][j][i][r] = 0.f;

    constexpr int PF = 3;
    uint4 rb[PF][B_PASS];
    uint4 ra[A_PASS];
    auto bload = [&](uint4 (&r)[B_PASS], int stage, int c0) {
        const int wofs = (int)g.wt[SC[stage]][ST[stage]] * g.Ci + c0;
#pragma unroll
        for (int p = 0; p < B_PASS; ++p) {
            const int v = (B_VECS % NT == 0) ? tid + p * NT : min(tid + p * NT, B_VECS - 1);
            const int row = min(n0 + v / VPR, g.co_pad - 1), col = (v % VPR) * VEC;
            r[p] = *reinterpret_cast<const uint4*>(Wp + (int64_t)row * g.ktot + wofs + col);
        }
    };
    auto bstore = [&](const uint4 (&r)[B_PASS], int buf) {
#pragma unroll
        for (int p = 0; p < B_PASS; ++p) {
            const int v = tid + p * NT;
            if (B_VECS % NT == 0 || v < B_VECS) lds_store_vec<T, PITCH>(Bs + buf * BN * PITCH, v / VPR, (v % VPR) * VEC, r[p]);
        }
    };
    auto aload = [&](int c0) {
#pragma unroll
        for (int p = 0; p < A_PASS; ++p) {
            const int v = tid + p * NT;
            const int hp = v / VPR, col = (v % VPR) * VEC;
            const int ih = th0 + hp / HW_, iw = tw0 + hp % HW_;
            const bool ok = v < A_VECS && ih < g.H && iw < g.W;
            ra[p] = ldg16_masked(X, (((int64_t)ih * g.W + iw) * g.x_ld + c0 + col) * (int64_t)sizeof(T), ok);
        }
    };
    auto astore = [&]() {
#pragma unroll
        for (int p = 0; p < A_PASS; ++p) {
            const int v = tid + p * NT;
            if (v < A_VECS) lds_store_vec<T, PITCH>(As + ((v / VPR) / HW_) * RP, (v / VPR) % HW_, (v % VPR) * VEC, ra[p]);
        }
    };

    const int nchunks = g.Ci / CK;
    aload(0);
#pragma unroll
    for (int d = 0; d < PF; ++d) bload(rb[d], d, 0);
    astore();
    bstore(rb[0], 0);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
        const int c0 = c * CK;
        const int c1 = c + 1 < nchunks ? c0 + CK : c0;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            if (s + PF < 9) bload(rb[s % PF], s + PF, c0);
            else bload(rb[s % PF], s + PF - 9, c1);
            if (s == 4) aload(c1);
            const int shift = (int)g.dh[SC[s]][ST[s]] * RP + (int)g.dw[SC[s]][ST[s]] * PITCH;
            const T* bt = Bs + ((c + s) & 1) * BN * PITCH;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                bf16x8_t fa[MI], fb[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    fa[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(As + hbase[i] + shift + kk * 16 + (lane >> 5) * 8));
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = Mma<T>::load(bt, PITCH, wn + j * 32, kk, lane);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i) acc[SC[s]][j][i] = Mma<T>::mma(fb[j], fa[i], acc[SC[s]][j][i]);
            }
            bstore(rb[(s + 1) % PF], (c + s + 1) & 1);
            __syncthreads();
            if (s == 8) {
                astore();
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
        if (cls > 0) __syncthreads();                  // the previous class's slab has been read
        const int ph = cls >> 1, pw = cls & 1;
        conv_epilogue_tile<MI, NJ, WGM, WGN, NT>(g, acc[cls], reinterpret_cast<float*>(lds), n0, tid, [&](int m) -> int64_t {
            const int oh = 2 * (th0 + m / D2_TW) + ph, ow = 2 * (tw0 + m % D2_TW) + pw;
            return (th0 + m / D2_TW < g.H && tw0 + m % D2_TW < g.W && oh < g.OH && ow < g.OW) ? ((int64_t)n * g.OH + oh) * g.OW + ow : -1;
        });
    }
}

// caller: transposed 3x3 / stride 2 / pad 1 geometry in g (four parity classes), bf16; false when the shape is not handled
bool dgrad3x3s2_launch(const ConvGeom& g, int dtype, hipStream_t st) {
    static const bool disabled = getenv("CN_DISABLE_DGRAD3X3_S2") != nullptr;
    if (disabled || dtype != CN_BF16 || g.N > 65535 || (g.Ci & 31) || g.Ci < 32 || g.Co < 32 || g.nsrc != 0 || g.dcn_x || g.y_f32 || g.res32 ||
        g.head_nc || g.pre_ss || g.bn_part || g.relu > 2)
        return false;
    if (g.sm != 1 || g.so != 2 || g.ntaps[0] != 1 || g.ntaps[1] != 2 || g.ntaps[2] != 2 || g.ntaps[3] != 4) return false;
    for (int c = 0; c < 4; ++c)
        for (int t = 0; t < g.ntaps[c]; ++t)
            if ((unsigned)g.dh[c][t] > 1u || (unsigned)g.dw[c][t] > 1u) return false;
    if (g.OH > 2 * g.H || g.OW > 2 * g.W || !conv_epi_tile_ok(g, dtype)) return false;
    const_cast<ConvGeom&>(g).epi_tile = 1;
    const int bn = g.Co <= 32 ? 32 : 64;
    dim3 grid(((g.H + D2_TH - 1) / D2_TH) * ((g.W + D2_TW - 1) / D2_TW), (g.Co + bn - 1) / bn, g.N);
    if (g.Ci % 64 == 0) {
        if (bn == 64) hipLaunchKernelGGL((dgrad3x3s2_kernel<64, 64, 8>), grid, dim3(512), 0, st, g);
        else hipLaunchKernelGGL((dgrad3x3s2_kernel<32, 64, 4>), grid, dim3(256), 0, st, g);
    } else {
        if (bn == 64) hipLaunchKernelGGL((dgrad3x3s2_kernel<64, 32, 8>), grid, dim3(512), 0, st, g);
        else hipLaunchKernelGGL((dgrad3x3s2_kernel<32, 32, 4>), grid, dim3(256), 0, st, g);
    }
    return true;
}
