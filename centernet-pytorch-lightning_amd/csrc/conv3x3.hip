// 3x3 / stride 1 / pad 1 convolution (and its data gradient = the same conv with mirrored taps) — the shape that
// carries most of DLA-34's FLOPs.  The generic implicit-GEMM kernel re-reads every input pixel once per tap and pays
// one global->LDS round trip + barrier per tap slice, which left it latency-bound (7-15 % of the MFMA peak, ~13 % of the
// HBM roofline on the 64-channel 128x128 layers).  Here a workgroup owns an 8x16 output-pixel tile:
//   * the (8+2)x(16+2) input HALO tile of a 64-channel slice is loaded into LDS ONCE (1.4x over-read instead of 9x)
//     and all 9 taps read their A fragments from it with ds_read_b128 at a per-lane shifted pixel address;
//   * the per-tap weight slice [BN][64] is double-buffered through LDS with a register prefetch one tap ahead, so the
//     only exposed global latency per workgroup is the halo load; 2-3 workgroups per CU overlap it with MFMA work.
// bf16: v_mfma_f32_32x32x16_bf16; fp32 (parity mode): v_mfma_f32_32x32x2_f32.  Epilogue shared with the generic kernel.
#include "conv_common.h"

#define T3_TH 8
#define T3_TW 16

// S = 2 (round 4): the stride-2 / pad-1 forward convs that open every DLA / ResNet stage (32 -> 64 @256^2 ... 256 -> 512 @32^2) on the same
// skeleton — a (2*8+1) x (2*16+1) halo tile, every other halo pixel per output pixel; they ran on the generic implicit GEMM, which
// re-reads the input once per tap (1.5-2.0 TB/s of algorithmic traffic, 0.1-0.25 of the MFMA peak).
template <typename T, int BN, int CK, int NW = 4, int S = 1>   // NW waves per workgroup: 4 (64x64 wave tiles) or 8 (32x64: twice the waves per SIMD)
__global__ __launch_bounds__(NW * 64) void conv3x3s1_kernel(const ConvGeom g) {
    CN_MAIN_PRIO_SET();
    constexpr int NT = NW * 64;
    constexpr int BM = T3_TH * T3_TW;                  // 128 output pixels
    constexpr int HW_ = T3_TW * S + 3 - S, HH_ = T3_TH * S + 3 - S;    // halo tile: 10 x 18 (stride 1), 17 x 33 (stride 2)
    constexpr int HP = HH_ * HW_;                      // 180 / 561 halo pixels
    constexpr int VEC = 16 / sizeof(T);
    constexpr int PITCH = CK + Mma<T>::PAD;
    constexpr int VPR = CK / VEC;                      // 16-byte vectors per pixel / weight row
    constexpr int A_VECS = HP * VPR;
    constexpr int A_PASS = (A_VECS + NT - 1) / NT;
    constexpr int B_VECS = BN * VPR;
    constexpr int B_PASS = (B_VECS + NT - 1) / NT;
    constexpr int WGN = (BN >= 64) ? 2 : 1;
    constexpr int WGM = NW / WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MI = WM / 32, NJ = WN / 32;
    constexpr int KSTEPS = CK / Mma<T>::KSTEP;

    // ds_read_b128 is serviced in four fixed 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32): an A-fragment read of a
    // group touches columns {0-3,12-15} of one tile row and {4-11} of the next.  With the pixel pitch of 9 slots (144 B) the two
    // sets tile all 16 slots exactly when the halo ROW pitch is 0 mod 16 slots; the natural 18 x 9 = 162 = 2 (mod 16) made two
    // slots collide in every group (every A read cost 2 LDS cycles instead of 1).  Rows are padded to 176 slots.
    constexpr int RPAD = (sizeof(T) == 2) ? ((16 - (HW_ * PITCH / 8) % 16) % 16) * 8 : 0;
    constexpr int RP = HW_ * PITCH + RPAD;             // halo row pitch in elements
    constexpr int MAIN_ELEMS = HH_ * RP + 2 * BN * PITCH;
    constexpr int EPI_ELEMS = (sizeof(T) == 2) ? WGM * 32 * (BN + 4) * 2 : 0;   // fp32 slab of the LDS-staged epilogue
    __shared__ __attribute__((aligned(16))) T lds[MAIN_ELEMS > EPI_ELEMS ? MAIN_ELEMS : EPI_ELEMS];
    T* const As = lds;
    T* const Bs = lds + HH_ * RP;                      // two buffers of BN rows

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_w = (g.OW + T3_TW - 1) / T3_TW;
    const int th0 = (blockIdx.x / tiles_w) * T3_TH, tw0 = (blockIdx.x % tiles_w) * T3_TW;
    const int n0 = blockIdx.y * BN;
    const int n = blockIdx.z;
    const int wm = (wave / WGN) * WM, wn = (wave % WGN) * WN;

    const T* __restrict__ X = reinterpret_cast<const T*>(g.x) + (int64_t)n * g.H * g.W * g.x_ld;
    const T* __restrict__ Wp = reinterpret_cast<const T*>(g.w);

    // lane's pixels in the halo tile (centre position, i.e. tap shift (0,0))
    int hbase[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = wm + i * 32 + (lane & 31);
        hbase[i] = ((m / T3_TW) * S + 1) * RP + ((m % T3_TW) * S + 1) * PITCH;   // element offset of the centre pixel
    }

    f32x16_t acc[NJ][MI];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    // Weight slices are prefetched PF taps ahead through a ring of register sets: one L2 round trip is ~1 us under
    // load while a tap's MFMA work is 0.1-0.25 us, so a one-tap prefetch left every tap waiting on its weights.
    constexpr int PF = 3;                              // 9 % PF == 0 keeps the ring index static per unrolled tap
    uint4 rb[PF][B_PASS];
    uint4 ra[A_PASS];
    auto bload = [&](uint4 (&r)[B_PASS], int tap, int c0) {
        const int wofs = (int)g.wt[0][tap] * g.Ci + c0;
#pragma unroll
        for (int p = 0; p < B_PASS; ++p) {
            // branch-free: rows past the packed matrix re-read its last row (those output channels are never stored)
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
    auto aload = [&](int c0) {                         // halo tile of one channel slice: one pass over HBM/L2
#pragma unroll
        for (int p = 0; p < A_PASS; ++p) {
            const int v = tid + p * NT;
            const int hp = v / VPR, col = (v % VPR) * VEC;
            const int ih = th0 * S - 1 + hp / HW_, iw = tw0 * S - 1 + hp % HW_;
            const bool ok = v < A_VECS && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
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
        const bool more = c + 1 < nchunks;
        const int c1 = more ? c0 + CK : c0;            // the last slice prefetches itself again: no branches around loads
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // stage s = 9c + tap; its weights sit in LDS buffer (s & 1); stage s+PF goes into the ring slot just freed
            if (tap + PF < 9) bload(rb[tap % PF], tap + PF, c0);
            else bload(rb[tap % PF], tap + PF - 9, c1);
            if (tap == 4) aload(c1);
            const int shift = (int)g.dh[0][tap] * RP + (int)g.dw[0][tap] * PITCH;
            const T* bt = Bs + ((c + tap) & 1) * BN * PITCH;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                typename Mma<T>::Frag fa[MI], fb[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    if constexpr (sizeof(T) == 2) {
                        const uint4 v = *reinterpret_cast<const uint4*>(As + hbase[i] + shift + kk * 16 + (lane >> 5) * 8);
                        fa[i] = __builtin_bit_cast(bf16x8_t, v);
                    } else {
                        fa[i] = As[hbase[i] + shift + kk * 2 + (lane >> 5)];
                    }
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = Mma<T>::load(bt, PITCH, wn + j * 32, kk, lane);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i) acc[j][i] = Mma<T>::mma(fb[j], fa[i], acc[j][i]);
            }
            bstore(rb[(tap + 1) % PF], (c + tap + 1) & 1);
            __syncthreads();
            if (tap == 8) {                            // everybody is done with this slice's halo tile
                astore();
                __syncthreads();
            }
        }
    }

    int64_t pix[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = wm + i * 32 + (lane & 31);
        const int oh = th0 + m / T3_TW, ow = tw0 + m % T3_TW;
        pix[i] = (oh < g.OH && ow < g.OW) ? ((int64_t)n * g.OH + oh) * g.OW + ow : -1;
    }
    if constexpr (sizeof(T) == 2) {
        if (g.epi_tile) {                              // main loop ended on a barrier: the LDS is free
            conv_epilogue_tile<MI, NJ, WGM, WGN, NT>(g, acc, reinterpret_cast<float*>(lds), n0, tid, [&](int m) -> int64_t {
                const int oh = th0 + m / T3_TW, ow = tw0 + m % T3_TW;
                return (oh < g.OH && ow < g.OW) ? ((int64_t)n * g.OH + oh) * g.OW + ow : -1;
            });
            return;
        }
    }
    conv_epilogue<T, MI, NJ>(g, acc, pix, n0 + wn, lane);
}

// ---- 16 -> 16 channel special case (DLA level0 and its data gradient at 512x512) -----------------------------------
// These layers are pure HBM streams (1 GB in+out, 2.5 GFLOP/MB); the tile kernel above wastes 3/4 of every MFMA and
// all of its LDS staging on them.  Here a wave owns one 64-pixel row strip and feeds v_mfma_f32_16x16x32_bf16 straight
// from global memory: M = 16 output channels, N = 16 pixels, K = 32 = two taps x 16 input channels, so five MFMAs cover
// the 9 taps (the tenth half is masked).  Operand lanes: weights A[co = lane&15][k = 8*(lane>>4)..+7] (held in registers
// for the whole launch), pixels B[k][px = lane&15] = 16 contiguous bytes of pixel (w+px+dw, h+dh) — a 16-pixel group is
// one 512-byte run, so every load is fully coalesced and the 9x tap re-reads are L1/L2 hits.  D[co = 4*(lane>>4)+r][px]:
// each lane stores 4 consecutive channels of one pixel (8 bytes), a group again being one contiguous 512-byte run.
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#define C16_GROUPS 4
__global__ __launch_bounds__(256, 4) void conv3x3_c16_kernel(const ConvGeom g) {
    CN_MAIN_PRIO_SET();     // <= 128 registers: four waves per SIMD (an HBM stream)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, kc = lane >> 4;
    const int half = kc & 1, tsel = kc >> 1;
    const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(g.x);
    const bf16_t* __restrict__ Wp = reinterpret_cast<const bf16_t*>(g.w);
    bf16_t* __restrict__ Y = reinterpret_cast<bf16_t*>(g.y);

    bf16x8_t wa[5];
    int dh[5], dw[5];
    bool tv[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const int t1 = 2 * m + 1 < 9 ? 2 * m + 1 : 8;
        const int wt = tsel ? (int)g.wt[0][t1] : (int)g.wt[0][2 * m];
        dh[m] = tsel ? (int)g.dh[0][t1] : (int)g.dh[0][2 * m];
        dw[m] = tsel ? (int)g.dw[0][t1] : (int)g.dw[0][2 * m];
        tv[m] = 2 * m + tsel < 9;
        wa[m] = __builtin_bit_cast(bf16x8_t, ldg16_masked(Wp, ((int64_t)px * g.ktot + wt * 16 + half * 8) * 2, tv[m]));
    }
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * kc + r < g.Co) bias4[r] = g.bias[4 * kc + r];
    }

    // BatchNorm statistics of the stored values (sink protocol of bn.hip): this lane's four channels over every pixel it stores
    const bool stats = g.bn_part != nullptr;
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    const int segs = (g.W + 16 * C16_GROUPS - 1) / (16 * C16_GROUPS);
    const int64_t strips = (int64_t)g.N * g.H * segs;
#pragma unroll 1
    for (int64_t s = (int64_t)blockIdx.x * 4 + wave; s < strips; s += (int64_t)gridDim.x * 4) {
        const int seg = (int)(s % segs);
        const int64_t row = s / segs;                   // n * H + h
        const int h = (int)(row % g.H);
        const int64_t img_row0 = row - h;               // n * H
        uint4 xb[C16_GROUPS][5];
#pragma unroll
        for (int gq = 0; gq < C16_GROUPS; ++gq) {
            const int wq = seg * 16 * C16_GROUPS + gq * 16 + px;
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                const int ih = h + dh[m], iw = wq + dw[m];
                const bool ok = tv[m] && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
                xb[gq][m] = ldg16_masked(X, (((img_row0 + ih) * g.W + iw) * g.x_ld + half * 8) * 2, ok);
            }
        }
#pragma unroll
        for (int gq = 0; gq < C16_GROUPS; ++gq) {
            f32x4_t acc = {bias4[0], bias4[1], bias4[2], bias4[3]};
#pragma unroll
            for (int m = 0; m < 5; ++m)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[m], __builtin_bit_cast(bf16x8_t, xb[gq][m]), acc, 0, 0, 0);
            const int wq = seg * 16 * C16_GROUPS + gq * 16 + px;
            if (wq < g.W && 4 * kc < g.y_ld) {           // y_ld is a multiple of 4; padding channels are written as zeros
                float v[4] = {acc[0], acc[1], acc[2], acc[3]};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (g.relu == 1) v[r] = fmaxf(v[r], 0.f);
                    if (4 * kc + r >= g.Co) v[r] = 0.f;
                }
                uint2 o;
                o.x = pk_bf16(v[0], v[1]); o.y = pk_bf16(v[2], v[3]);
                *reinterpret_cast<uint2*>(Y + ((row * g.W) + wq) * g.y_ld + 4 * kc) = o;
                if (stats) {
                    const float a0 = __uint_as_float(o.x << 16), a1 = __uint_as_float(o.x & 0xffff0000u);
                    const float a2 = __uint_as_float(o.y << 16), a3 = __uint_as_float(o.y & 0xffff0000u);
                    s0[0] += a0; s0[1] += a1; s0[2] += a2; s0[3] += a3;
                    s1[0] = fmaf(a0, a0, s1[0]); s1[1] = fmaf(a1, a1, s1[1]); s1[2] = fmaf(a2, a2, s1[2]); s1[3] = fmaf(a3, a3, s1[3]);
                }
            }
        }
    }
    if (stats) {
        __shared__ float red[4 * 32];
        bn_stats_flush_c16<256>(s0, s1, red, g.bn_part, g.bn_slots, g.y_ld, g.Co, blockIdx.x, threadIdx.x);
    }
}

// ---- data gradient of the 16 -> 32 channel 3x3 / stride-2 conv (DLA level1: dx 16ch @512x512 from dy 32ch @256x256) --------
// Another pure HBM stream (268 MB in, 537 MB out) that the parity-class implicit GEMM ran at 30 TFLOP/s.  K = 32 = the 32
// channels of dy, so ONE v_mfma_f32_16x16x32_bf16 is one tap for 16 output pixels.  A wave owns output rows (2r, 2r+1) x 32
// columns: it loads the four dy fragments (rows r, r+1) x (column shift 0, +1) straight from global memory — 16 pixels x 64 B
// = 1 KB contiguous each — and issues the 1 + 2 + 2 + 4 MFMAs of the four output parity classes (taps whose offset divides
// the stride; the zero-stuffed positions are never touched).  D[ci = 4*(lane>>4)+r][px]: 8-byte stores, the even and the
// odd column of a pair land next to each other.
template <int KC, int NB>      // dy channels = 32*KC, dx channels = 16*NB  (1,1: DLA level1; 2,2: level2's 32 <- 64)
__global__ __launch_bounds__(256) void dgrad_s2_c32to16_kernel(const ConvGeom g) {
    CN_MAIN_PRIO_SET();
    constexpr int CO = 32 * KC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, kc = lane >> 4;
    const bf16_t* __restrict__ DY = reinterpret_cast<const bf16_t*>(g.x);
    const bf16_t* __restrict__ Wp = reinterpret_cast<const bf16_t*>(g.w);
    bf16_t* __restrict__ DX = reinterpret_cast<bf16_t*>(g.y);
    bf16x8_t wa[NB][KC][9];                                // [ci = 16*nb + px][co = 32*k + 8*kc..+7] of weight tap t = kh*3 + kw
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int k = 0; k < KC; ++k)
#pragma unroll
            for (int t = 0; t < 9; ++t)
                wa[nb][k][t] = __builtin_bit_cast(bf16x8_t, ldg16(Wp + (int64_t)(16 * nb + px) * g.ktot + t * CO + 32 * k + 8 * kc));
    // BatchNorm backward statistics of the stored gradient (cn_hooks.bnb_part; 16 output channels only)
    const bool bnb = NB == 1 && g.bnb_part != nullptr;
    BnbLane bl;
    bnb_lane_init(bl, bnb ? g.bnb_stats : nullptr, g.y_ld, 4 * kc);
    float bs0[4] = {0.f, 0.f, 0.f, 0.f}, bs1[4] = {0.f, 0.f, 0.f, 0.f};
    const int segs = (g.W + 15) / 16;                      // 16 dy columns -> 32 dx columns
    const int64_t strips = (int64_t)g.N * g.H * segs;      // one strip = one dy row r of one image -> dx rows 2r, 2r+1
#pragma unroll 1
    for (int64_t sidx = (int64_t)blockIdx.x * 4 + wave; sidx < strips; sidx += (int64_t)gridDim.x * 4) {
        const int seg = (int)(sidx % segs);
        const int64_t row = sidx / segs;                   // n * H + r
        const int r = (int)(row % g.H);
        const int c0 = seg * 16 + px;                      // dy column of this lane at shift 0
        bf16x8_t b[2][2][KC];
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int dw = 0; dw < 2; ++dw) {
                const bool ok = r + dh < g.H && c0 + dw < g.W;
#pragma unroll
                for (int k = 0; k < KC; ++k)
                    b[dh][dw][k] = __builtin_bit_cast(bf16x8_t, ldg16_masked(DY, (((row + dh) * g.W + c0 + dw) * g.x_ld + 32 * k + 8 * kc) * 2, ok));
            }
        const int64_t n = (row - r) / g.H;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4_t ee = {0.f, 0.f, 0.f, 0.f}, eo = ee, oe = ee, oo = ee;
            // dx[2r + ph][2c + pw] = sum over taps (kh, kw) with ph + 1 - kh, pw + 1 - kw even of W[kh][kw]^T dy[r + (ph+1-kh)/2][c + (pw+1-kw)/2]
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                ee = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nb][k][4], b[0][0][k], ee, 0, 0, 0);                 // (1,1)
                eo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nb][k][3], b[0][1][k], eo, 0, 0, 0);                 // (1,0): dw = 1
                eo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nb][k][5], b[0][0][k], eo, 0, 0, 0);                 // (1,2): dw = 0
                oe = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nb][k][1], b[1][0][k], oe, 0, 0, 0);                 // (0,1): dh = 1
                oe = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nb][k][7], b[0][0][k], oe, 0, 0, 0);                 // (2,1): dh = 0
                oo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nb][k][0], b[1][1][k], oo, 0, 0, 0);                 // (0,0)
                oo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nb][k][2], b[1][0][k], oo, 0, 0, 0);                 // (0,2)
                oo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nb][k][6], b[0][1][k], oo, 0, 0, 0);                 // (2,0)
                oo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[nb][k][8], b[0][0][k], oo, 0, 0, 0);                 // (2,2)
            }
            auto put = [&](const f32x4_t& v, int ph, int pw) {
                const int oh = 2 * r + ph, ow = 2 * c0 + pw;
                if (oh < g.OH && ow < g.OW) {
                    const uint2 o = make_uint2(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]));
                    const int64_t at = ((n * g.OH + oh) * g.OW + ow) * g.y_ld + 16 * nb + 4 * kc;
                    *reinterpret_cast<uint2*>(DX + at) = o;
                    if constexpr (NB == 1) {
                        if (bnb) bnb_lane_add(bl, o, *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(g.bnb_x) + at), g.bnb_relu, bs0, bs1);
                    }
                }
            };
            put(ee, 0, 0); put(eo, 0, 1); put(oe, 1, 0); put(oo, 1, 1);
        }
    }
    if constexpr (NB == 1) {
        if (bnb) {                                         // uniform over the workgroup
            __shared__ float red[4 * 32];
            bn_stats_flush_c16<256>(bs0, bs1, red, g.bnb_part, g.bnb_slots, g.y_ld, g.Co, blockIdx.x, threadIdx.x);
        }
    }
}

// caller guarantees: bf16, transposed 3x3 / stride 2 / pad 1, (32 -> 16) or (64 -> 32) channels, OH = 2H, OW = 2W, no bias / residual / ReLU
bool dgrad_s2_c32to16_launch(const ConvGeom& g, hipStream_t st) {
    static const bool disabled = getenv("CN_DISABLE_CONV_C16") != nullptr;
    if (disabled || (g.x_ld & 7) || g.y_ld != g.Co) return false;
    const int64_t strips = (int64_t)g.N * g.H * ((g.W + 15) / 16);
    int64_t blocks = (strips + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    if (g.bnb_part) {
        if (g.Ci == 32 && g.Co == 16 && g.bnb_slots > 0) mark_taken(g.bnb_taken); else const_cast<ConvGeom&>(g).bnb_part = nullptr;
    }
    if (g.Ci == 32 && g.Co == 16) hipLaunchKernelGGL((dgrad_s2_c32to16_kernel<1, 1>), dim3((unsigned)blocks), dim3(256), 0, st, g);
    else if (g.Ci == 64 && g.Co == 32) hipLaunchKernelGGL((dgrad_s2_c32to16_kernel<2, 2>), dim3((unsigned)blocks), dim3(256), 0, st, g);
    else return false;
    return true;
}

template <typename T, int BN, int CK>
static void launch3(const ConvGeom& g, hipStream_t st) {
    dim3 grid(((g.OH + T3_TH - 1) / T3_TH) * ((g.OW + T3_TW - 1) / T3_TW), (g.Co + BN - 1) / BN, g.N);
    // 8 waves (32x64 wave tiles, 4 waves/SIMD with two workgroups per CU) hide the per-tap LDS/barrier latency better on the
    // 128-channel tile: 901 vs 850 TFLOP/s; on the 64-channel tile the extra fragment reads cost more than they hide (840 vs 900)
    static const int nw_env = getenv("CN_CONV3X3_WAVES") ? atoi(getenv("CN_CONV3X3_WAVES")) : 0;
    if constexpr (sizeof(T) == 2 && CK == 64 && BN >= 64) {
        const int nw = nw_env ? nw_env : (BN == 128 ? 8 : 4);
        if (nw == 8) { hipLaunchKernelGGL((conv3x3s1_kernel<T, BN, CK, 8>), grid, dim3(512), 0, st, g); return; }
    }
    hipLaunchKernelGGL((conv3x3s1_kernel<T, BN, CK>), grid, dim3(256), 0, st, g);
}

// 3x3 / stride 2 / pad 1 forward conv (class 0 of g holds the nine taps with dh, dw in {-1, 0, 1} relative to input pixel (2 oh, 2 ow)):
// bf16, Ci a multiple of 32; returns false when the shape is not handled (the caller runs the implicit GEMM)
bool conv3x3s2_launch(const ConvGeom& g, int dtype, hipStream_t st) {
    static const bool disabled = getenv("CN_DISABLE_CONV3X3_S2") != nullptr;
    if (disabled || dtype != CN_BF16 || g.N > 65535 || (g.Ci & 31) || g.Ci < 32 || g.nsrc != 0 || g.dcn_x != nullptr || g.y_f32 || g.res32) return false;
    if (g.ntaps[0] != 9 || g.sm != 2 || g.so != 1) return false;
    for (int t = 0; t < 9; ++t)
        if (g.dh[0][t] < -1 || g.dh[0][t] > 1 || g.dw[0][t] < -1 || g.dw[0][t] > 1) return false;
    static const bool no_tile = getenv("CN_DISABLE_EPI_TILE") != nullptr;
    const_cast<ConvGeom&>(g).epi_tile = (conv_epi_tile_ok(g, dtype) && !no_tile) ? 1 : 0;
    if (g.bn_part) {                                   // BN statistics sink: the LDS-staged epilogue has the hook
        if (g.epi_tile) mark_taken(g.bn_taken); else const_cast<ConvGeom&>(g).bn_part = nullptr;
    }
    int bn = 32, bnb = (g.Co + 31) / 32;
    for (int c : {64, 128}) {
        int nb = (g.Co + c - 1) / c;
        if (nb < bnb) { bn = c; bnb = nb; }
    }
    dim3 grid(((g.OH + T3_TH - 1) / T3_TH) * ((g.OW + T3_TW - 1) / T3_TW), (g.Co + bn - 1) / bn, g.N);
    if (bn == 128) hipLaunchKernelGGL((conv3x3s1_kernel<bf16_t, 128, 32, 8, 2>), grid, dim3(512), 0, st, g);
    else if (bn == 64) hipLaunchKernelGGL((conv3x3s1_kernel<bf16_t, 64, 32, 4, 2>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((conv3x3s1_kernel<bf16_t, 32, 32, 4, 2>), grid, dim3(256), 0, st, g);
    return true;
}

bool conv3x3s1_launch(const ConvGeom& g, int dtype, hipStream_t st) {
    // caller guarantees: 3x3, stride 1, pad 1 geometry in class 0 of g (normal or mirrored taps), OH == H, OW == W
    if (g.N > 65535) return false;
    static const bool no_tile = getenv("CN_DISABLE_EPI_TILE") != nullptr, no_c16 = getenv("CN_DISABLE_CONV_C16") != nullptr;
    const_cast<ConvGeom&>(g).epi_tile = (conv_epi_tile_ok(g, dtype) && !no_tile) ? 1 : 0;
    if (g.Ci == 16 && conv_c16r_launch(g, dtype, 1, st)) return true;     // row-walking 16-channel kernel (conv_c16.hip)
    if (dtype == CN_BF16 && g.Ci == 16 && g.Co <= 16 && g.co_pad >= 16 && !g.res && !g.res32 && !g.y_f32 &&
        (g.y_ld & 3) == 0 && g.y_ld <= 16 && (g.x_ld & 7) == 0 && !no_c16) {
        const int64_t strips = (int64_t)g.N * g.H * ((g.W + 16 * C16_GROUPS - 1) / (16 * C16_GROUPS));
        int64_t blocks = (strips + 3) / 4;
        if (blocks > 4096) blocks = 4096;
        if (g.bn_part) mark_taken(g.bn_taken);            // the direct 16-channel kernel has the statistics hook
        hipLaunchKernelGGL(conv3x3_c16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g);
        return true;
    }
    if (conv3x3_ws_launch(g, dtype, st)) return true;  // 64 input channels, enough tiles: weight-stationary persistent kernel
    if (g.bn_part) {                                   // BN statistics sink: the LDS-staged epilogue has the hook
        if (g.epi_tile) mark_taken(g.bn_taken); else const_cast<ConvGeom&>(g).bn_part = nullptr;
    }
    int bn = 32, bnb = (g.Co + 31) / 32;               // same rule as pick_tile(): fewest channel blocks
    for (int c : {64, 128}) {
        int nb = (g.Co + c - 1) / c;
        if (nb < bnb) { bn = c; bnb = nb; }
    }
    if (dtype == CN_BF16) {
        if (g.Ci % 64 == 0) {
            if (bn == 128) launch3<bf16_t, 128, 64>(g, st); else if (bn == 64) launch3<bf16_t, 64, 64>(g, st); else launch3<bf16_t, 32, 64>(g, st);
        } else if (g.Ci % 32 == 0) {
            if (bn == 128) launch3<bf16_t, 128, 32>(g, st); else if (bn == 64) launch3<bf16_t, 64, 32>(g, st); else launch3<bf16_t, 32, 32>(g, st);
        } else {
            if (bn == 128) launch3<bf16_t, 128, 16>(g, st); else if (bn == 64) launch3<bf16_t, 64, 16>(g, st); else launch3<bf16_t, 32, 16>(g, st);
        }
    } else {
        if (bn == 128) launch3<float, 128, 16>(g, st); else if (bn == 64) launch3<float, 64, 16>(g, st); else launch3<float, 32, 16>(g, st);
    }
    return true;
}
