// DCNv2 offset / mask gradient with the corner dot products ON THE MATRIX CORES (bf16; companion of dcn_bm.hip).
//
//   dcol_k[p][ci] = sum_co W_k[co][ci] dY[p][co]                     (column gradient of tap k, never stored)
//   D_k[p][q]     = sum_ci dcol_k[p][ci] x[q][ci]                    (q = a source pixel of p's sampling window)
//   d mask[p][k]  = sum_ab wy_a wx_b D_k[p][corner ab]               wy = (1-ly, ly), wx = (1-lx, lx)
//   d off_y[p][k] = m sum_b wx_b (D[1b] - D[0b]),   d off_x[p][k] = m sum_a wy_a (D[a1] - D[a0])   (the formulas
//   of dcn_bwd_dom_kernel, dcn_fused.hip)
//
// The tile kernel (dcn_bwd_dom_kernel) sends dcol through LDS as a bf16 tile and lets four lanes per pixel gather the four corner
// rows and dot them on the VALU (v_dot2c): 160 KB of LDS traffic and two barriers per tap and 8x16 tile, 4 100 cycles per tap
// with one 8-wave workgroup per CU — latency bound.  Here a WAVE owns a 4x8 group of pixels for all nine taps and nothing goes
// through LDS but the x window itself:
//   * dcol^T[ci][p] = W_k^T dY^T leaves the matrix pipe with lane = pixel, registers = channels.  The ROWS of the W operand are
//     loaded in the order that makes registers 8a .. 8a+7 of a lane half hold 8 CONSECUTIVE channels (row index with bits 2 and 3
//     swapped), so those registers, packed to bf16, are the B operand of the second product with no data movement;
//   * D^T[q][p] = X[q][ci] dcol^T[ci][p] for one PAIR of window rows (2 rows x 16 columns = the M of a 32x32x16 MFMA); the A
//     operand is a plain 16-byte read of the halo image (128 B per pixel, 16-byte chunks XOR-swizzled with the pixel number:
//     conflict-free);
//   * every lane then holds D for its own pixel against 16 of the 32 window pixels of the pair and reduces them with the
//     separable weights above (8 column weights per tap, two row weights per pair: ~40 FMAs) — no selection, no transposition.
//     Only the row pairs somebody in the wave samples are visited (three with zero offsets, four with N(0, 0.5 px)); the next
//     pair's fragments are requested between this pair's MFMAs and its reduction.
//   * dY fragments (loaded once, straight from global memory) live in registers for the whole tile.  The tap's W fragments: dY
//     with 64 channels — shared through LDS, two buffers, each wave loads two of the eight fragments one tap ahead, one barrier
//     per tap; 128 channels — one register set per wave, re-loaded from L2 right after its last use.  The nine results of a pixel
//     overwrite the geometry entries they were computed from in the wave's LDS table and leave as one coalesced row.
// Samples whose corners leave the window (|offset| > 3 px) take a per-lane VALU path from global memory; the dx_far scatter
// (corners more than DCN_FAR_R pixels from their pixel: the ones dcn_dx_bm_kernel / dcn_bwd_dx_kernel do not see) is the tile
// kernel's, per lane from the fp32 dcol registers.
#include "conv_common.h"
#include <stdlib.h>

#define DB_TH 8
#define DB_TW 16
#define DB_MG 4
#define DB_WR (DB_TH + 2 * DB_MG)        // 16 halo rows
#define DB_WC (DB_TW + 2 * DB_MG)        // 24 halo columns
#define DB_PIX 128                       // bytes per halo pixel (64 bf16); the eight 16-byte chunks of pixel number n are XOR-swizzled
#define DB_ROW (DB_WC * DB_PIX)          // with (n >> 1) & 7: the 16 consecutive pixels of an operand read then tile all 64 banks
// LDS byte offset of 16-byte chunk c (0..7) of halo pixel n
__device__ static inline int db_chunk(int n, int c) { return n * DB_PIX + ((c ^ ((n >> 1) & 7)) << 4); }

#ifdef DOMB_PROBE   // development build only (tools/dom_probe.py): cycle stamps of wave 0 of the first workgroups
__device__ unsigned long long domb_ts[1024 * 48];
#define DB_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 1024) domb_ts[blockIdx.x * 48 + (k)] = clock64(); } while (0)
extern "C" int domb_probe_dump(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(domb_ts), sizeof(domb_ts)); }
#else
#define DB_STAMP(k) do { } while (0)
#endif

struct DomBmGeom {
    const bf16_t* dy; const bf16_t* wd2; const bf16_t* x; const float* om; float* dom; bf16_t* dom16; float* far; int* far_flag;
    int N, H, W, Ci, x_ld;
    int64_t slab;
};

typedef uint32_t u32x4w __attribute__((ext_vector_type(4)));
typedef float f32x2w __attribute__((ext_vector_type(2)));

// acc + sum of the 8 bf16 products of two 16-byte vectors (4 x v_dot2c_f32_bf16; the elements are named one by one: with the
// vectors subscripted by an unrolled loop variable hipcc 7.2 used element 0 four times)
typedef __bf16 bf16x2w __attribute__((ext_vector_type(2)));
__device__ static inline float dom_dot8(const u32x4w& a, const u32x4w& b, float acc) {
    const uint32_t a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w, b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2w, a0), __builtin_bit_cast(bf16x2w, b0), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2w, a1), __builtin_bit_cast(bf16x2w, b1), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2w, a2), __builtin_bit_cast(bf16x2w, b2), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2w, a3), __builtin_bit_cast(bf16x2w, b3), acc, false);
    return acc;
}

__device__ static inline float half_sum(float v) {      // v(lane) + v(lane ^ 32) in every lane
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}

// GA (round 5): the corner dot products by a direct GATHER instead of the second MFMA product — every lane reads the 16-byte chunks of
// its own four corner pixels that hold the channels of its dcol registers (4 chunks x 4 corners) from the halo image and runs
// v_dot2c_f32_bf16 over them: ~150 VALU instructions and 8 MFMAs per tap instead of ~290 and ~22 (the row-pair products and their
// separable reductions; profiles/r04_pmc_sq.txt: 20 VALU instructions per MFMA), no data-dependent loop, all 16 reads in flight at
// once, and the in-window test is the TILE's halo (16 x 24) instead of the group's 12 x 16 window.
template <int COP, bool GA>   // channels of dY (contraction length of the first product): 64 or 128
__global__ __launch_bounds__(256, 2) void dcn_dom_bm_kernel(const DomBmGeom g) {
    CN_MAIN_PRIO_SET();
    constexpr int KS = COP / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Xw = smem;                                              // [16][24] pixels x 128 B (chunk-swizzled)
    float* const OmAll = reinterpret_cast<float*>(smem + DB_WR * DB_ROW);        // 4 waves x [32 px][29]
    // COP == 64: the tap's W fragments are shared through LDS (two buffers of 8 fragments x 1 KB): loaded from global memory by ONE
    // wave each (two fragments per wave) instead of by all four — the fragment loads (32 rows x 32 bytes per instruction) kept the
    // CU's vector-memory pipe busy for a seventh of the kernel (350 vs 405 us with tap 0's fragments reused for every tap)
    constexpr bool WLDS = COP == 64;
    unsigned char* const Ws = smem + DB_WR * DB_ROW + 4 * 32 * 29 * 4;           // WLDS: 2 x [2 cb][KS][64 lanes] x 16 B

    DB_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_w = (g.W + DB_TW - 1) / DB_TW, tiles_h = (g.H + DB_TH - 1) / DB_TH;
    const int tiles_img = tiles_w * tiles_h;
    // consecutive tiles (they share halo columns / rows) stay on one XCD: workgroups are dealt round-robin to the 8 XCDs
    const int G = gridDim.x;
    const int lb = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
    const int n = lb / tiles_img, rt = lb - n * tiles_img;
    const int ty0 = (rt / tiles_w) * DB_TH, tx0 = (rt % tiles_w) * DB_TW;
    const int ci0 = blockIdx.y * 64;
    const int64_t img = (int64_t)n * g.H * g.W;
    const bf16_t* __restrict__ X = g.x + img * g.x_ld + ci0;
    const float* __restrict__ OM = g.om + img * 32;

    const int grow = (wave >> 1) * 4, gcol = (wave & 1) * 8;
    const int nl = lane & 31, hh = lane >> 5;
    const int gy = ty0 + grow + (nl >> 3), gx = tx0 + gcol + (nl & 7);
    const bool live = gy < g.H && gx < g.W;

    // ---- loads in the order they are needed: offsets / mask logits, halo image, dY fragments, first W fragments ----
    float4 omr[4];
    const int part = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = (lane >> 3) + 8 * i;
        const int py_ = ty0 + grow + (p >> 3), px_ = tx0 + gcol + (p & 7);
        const bool ok = py_ < g.H && px_ < g.W;
        omr[i] = *reinterpret_cast<const float4*>(OM + ((int64_t)(ok ? py_ : 0) * g.W + (ok ? px_ : 0)) * 32 + part * 4);
    }
    constexpr int NV = DB_WR * DB_WC * 8 / 256;     // 12 sixteen-byte vectors per thread
    uint4 hv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * 256;
        const int pix = v >> 3, q = v & 7;
        const int hy = ty0 - DB_MG + pix / DB_WC, hx = tx0 - DB_MG + pix % DB_WC;
        const bool ok = (unsigned)hy < (unsigned)g.H && (unsigned)hx < (unsigned)g.W;
        hv[i] = ldg16_masked(X, (((int64_t)hy * g.W + hx) * g.x_ld + q * 8) * 2, ok);
    }
    // dY^T fragments (B operand: lane = pixel, 8 consecutive co per k-step and lane half), zeros for pixels outside the image
    u32x4w dyf[KS];
    {
        const int64_t pofs = (img + (int64_t)(live ? gy : 0) * g.W + (live ? gx : 0)) * COP;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const uint4 v = ldg16_masked(g.dy, (pofs + 16 * s + 8 * hh) * 2, live);
            dyf[s] = u32x4w{v.x, v.y, v.z, v.w};
        }
    }
    // W_k fragments (A operand: lane = ci row, 8 consecutive co): mode-2 pack [9*Ci][COP]; row r of block cb holds channel
    // 32 cb + swap_bits_2_3(r), so that accumulator registers 8a .. 8a+7 of lane half hh are channels 32 cb + 16 a + 8 hh .. +7
    const int wrow = (nl & ~12) | ((nl & 4) << 1) | ((nl & 8) >> 1);
    const bf16_t* const wbase = g.wd2 + ((int64_t)ci0 + wrow) * COP + 8 * hh;
    // COP == 128: one register set, re-loaded right after its last use (two would not fit)
    u32x4w wfa[2][KS];
    auto wload = [&](u32x4w (&wf)[2][KS], int tap) {
#ifdef DOMB_NOW      // timing experiment only: every tap uses tap 0's fragments (no per-tap global loads)
        if (tap > 1) return;
#endif
        const bf16_t* p = wbase + (int64_t)tap * g.Ci * COP;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int s = 0; s < KS; ++s) wf[cb][s] = *reinterpret_cast<const u32x4w*>(p + (int64_t)32 * cb * COP + 16 * s);
    };
    // WLDS: this wave's two fragments (index f = 2 wave + i = cb * KS + s) of a tap, global -> registers -> LDS buffer
    u32x4w wq[2];
    auto wq_load = [&](int tap) {
        const bf16_t* p = wbase + (int64_t)tap * g.Ci * COP;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = 2 * wave + i;
            wq[i] = *reinterpret_cast<const u32x4w*>(p + (int64_t)32 * (f / KS) * COP + 16 * (f % KS));
        }
    };
    auto wq_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4w*>(Ws + buf * (2 * KS * 1024) + ((2 * wave + i) * 64 + lane) * 16) = wq[i];
    };
    if (WLDS) wq_load(0); else wload(wfa, 0);

    // ---- geometry table of the wave's 32 pixels: entry 2k / 2k+1 = sampling position of tap k in IMAGE coordinates, formed like the
    //      reference forms it (one fp32 add of the integer position h - 1 + ky and the offset: floor / fraction are then bit-identical
    //      to the reference's, which matters where a sample sits on a pixel row or column exactly — d/d offset is one-sided there);
    //      entry 18+k = sigmoid(mask logit), 0 outside the image ----
    float* const Om = OmAll + wave * (32 * 29);
    {
        float tv[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = (lane >> 3) + 8 * i;
            const float v4[4] = {omr[i].x, omr[i].y, omr[i].z, omr[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = part * 4 + j, k = e >> 1, k3 = (k * 11) >> 5;          // k / 3 for k < 9
                tv[i][j] = v4[j] + (float)((e & 1) ? tx0 + gcol + (p & 7) - 1 + (k - 3 * k3) : ty0 + grow + (p >> 3) - 1 + k3);
            }
        }
        if (part >= 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = (lane >> 3) + 8 * i;
                const bool ok = ty0 + grow + (p >> 3) < g.H && tx0 + gcol + (p & 7) < g.W;
                const float v4[4] = {omr[i].x, omr[i].y, omr[i].z, omr[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (part * 4 + j >= 18) tv[i][j] = ok ? __builtin_amdgcn_rcpf(1.f + __expf(-v4[j])) : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float* d = Om + ((lane >> 3) + 8 * i) * 29 + part * 4;
            if (part < 7) { d[0] = tv[i][0]; d[1] = tv[i][1]; d[2] = tv[i][2]; if (part < 6) d[3] = tv[i][3]; }
        }
    }
    DB_STAMP(40);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * 256;
        st16(Xw + db_chunk(v >> 3, v & 7), hv[i]);
    }
    DB_STAMP(41);
    if (WLDS) wq_store(0);
    __syncthreads();        // COP == 128: the only barrier — from here on a wave reads the halo image and its own table
    DB_STAMP(1);
    if (WLDS) wq_load(1);

    float* const orow = Om + nl * 29;
    float ro[3] = {orow[0], orow[1], orow[18]};
    // A-operand address of the second product: lane (q, hh) reads window pixel (row q >> 4 of the pair, column q & 15), 8 channels
    // (pixel number n = row * 24 + column: (n >> 1) & 7 does not depend on the row PAIR, so one address per k-step serves every pair)
    const unsigned char* xop[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xop[s] = Xw + db_chunk((grow + (nl >> 4)) * DB_WC + gcol + (nl & 15), 2 * s + hh);
    const int prow = nl >> 3, pcol = nl & 7;                // this lane's pixel inside the group
    const int wy_org = ty0 + grow - DB_MG, wx_org = tx0 + gcol - DB_MG;      // image coordinates of the group's window origin
    const bool direct_far = g.far != nullptr;

    auto tap_body = [&](const int tap, u32x4w (&wf)[2][KS]) {
        // ---- geometry of (own pixel, tap) ----
        DB_STAMP(2 + 4 * tap);
        const float pyr = ro[0], pxr = ro[1], m = ro[2];
        {
            const int nt = tap < 8 ? tap + 1 : 8;
            ro[0] = orow[2 * nt]; ro[1] = orow[2 * nt + 1]; ro[2] = orow[18 + nt];
        }
        const float fy = floorf(pyr), fx = floorf(pxr);
        const int h0 = (int)fy, w0 = (int)fx;                                           // image coordinates of corner 00
        const int wr = h0 - wy_org, wc = w0 - wx_org;                                   // ... and its window coordinates
        const float ly = pyr - fy, lx = pxr - fx;
        const bool active = m != 0.f;                                                   // inside the image (the table holds 0 outside)
        const int tr = wr + grow, tc = wc + gcol;                                       // ... and its coordinates in the tile's halo image
        const bool inwin = GA ? ((unsigned)tr <= (unsigned)(DB_WR - 2) && (unsigned)tc <= (unsigned)(DB_WC - 2)) : ((unsigned)wr <= 10u && (unsigned)wc <= 14u);
        const bool winmiss = active && !inwin;
        // row pairs of the window anybody in the wave samples (rows wr and wr + 1 of every active in-window lane)
        uint32_t pairs = (!GA && active && inwin) ? ((1u << (wr >> 1)) | (1u << ((wr + 1) >> 1))) : 0u;
        uint32_t pm = 0u;
        if (!GA) {
            pairs |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pairs, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
            pairs |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pairs, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
            pairs |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pairs, 0x141, 0xF, 0xF, true);    // row_half_mirror
            pairs |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pairs, 0x140, 0xF, 0xF, true);    // row_mirror
            pm = (uint32_t)__builtin_amdgcn_readlane((int)pairs, 0) | (uint32_t)__builtin_amdgcn_readlane((int)pairs, 16) |
                 (uint32_t)__builtin_amdgcn_readlane((int)pairs, 32) | (uint32_t)__builtin_amdgcn_readlane((int)pairs, 48);
        }
        // the first pair's window fragments are requested before the first product (their latency hides behind its 8-16 MFMAs)
        int j = pm ? __builtin_ctz(pm) : 0;
        const bool any = pm != 0;
        pm &= pm - 1;
        u32x4w xa[4];
        u32x4w wloc[2][KS];
        if (WLDS) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int s = 0; s < KS; ++s) wloc[cb][s] = *reinterpret_cast<const u32x4w*>(Ws + (tap & 1) * (2 * KS * 1024) + ((cb * KS + s) * 64 + lane) * 16);
        }
        u32x4w (&W)[2][KS] = WLDS ? wloc : wf;
        // GA: position codes of the four corner pixels (byte offset of the pixel | swizzle key << 4: chunk c sits at code ^ (c << 4)) and
        // the first half of the gather (channel block 0) go out before the first product
        uint32_t cc[4];
        u32x4w ga[4][2];
        if (GA) {
            const int n00 = inwin ? tr * DB_WC + tc : 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int nk = n00 + (k >> 1) * DB_WC + (k & 1);
                cc[k] = (((uint32_t)nk << 7) | ((((uint32_t)nk >> 1) & 7u) << 4)) ^ (uint32_t)(hh << 4);
            }
        }
        if (WLDS && !GA) {
#pragma unroll
            for (int s = 0; s < 4; ++s) xa[s] = *reinterpret_cast<const u32x4w*>(xop[s] + j * (2 * DB_ROW));
        }

        // ---- dcol^T[ci][p] = W_k^T dY^T ----
        f32x16_t dc[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dc[cb][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                dc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, W[cb][s]), __builtin_bit_cast(bf16x8_t, dyf[s]), dc[cb], 0, 0, 0);
        if (!WLDS) {                                         // single register set: re-loaded right after its last use
            wload(wf, tap < 8 ? tap + 1 : 8);
            if (!GA) {
#pragma unroll
                for (int s = 0; s < 4; ++s) xa[s] = *reinterpret_cast<const u32x4w*>(xop[s] + j * (2 * DB_ROW));
            }
        }
        if (GA) {        // channel block 0 of the four corners: chunks 0 (+ hh, folded into the codes) and 2
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ga[k][0] = *reinterpret_cast<const u32x4w*>(Xw + cc[k]);
                ga[k][1] = *reinterpret_cast<const u32x4w*>(Xw + (cc[k] ^ 32u));
            }
        }
        // ---- column weights of this lane's 8 window columns (register i of a pair row <-> column 8 (i >> 2) + 4 hh + (i & 3)); VALU
        //      work placed behind the MFMAs it does not depend on ----
        f32x2w cg[8];                                        // {bilinear weight, derivative sign} of column i for this pixel
        if (!GA) {
            const int wcl = inwin ? wc - 4 * hh : -100;      // corner column relative to this lane half's first column
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = 8 * (i >> 2) + (i & 3);
                const bool e0 = wcl == c, e1 = wcl + 1 == c;
                cg[i][0] = e0 ? 1.f - lx : (e1 ? lx : 0.f);
                cg[i][1] = e0 ? -1.f : (e1 ? 1.f : 0.f);
            }
        }
        // B operands of the second product: k-step s = channels 16 s .. 16 s + 15 = registers 8 (s & 1) .. +7 of block s >> 1
        u32x4w dcb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int d = 0; d < 4; ++d) dcb[s][d] = pk_bf16(dc[s >> 1][8 * (s & 1) + 2 * d], dc[s >> 1][8 * (s & 1) + 2 * d + 1]);

        float sm = 0.f, sy = 0.f, sx = 0.f;
        DB_STAMP(3 + 4 * tap);
        // ---- D^T[q][p] per touched row pair, reduced with the separable weights; the next pair's fragments are requested between the
        //      MFMAs of this pair and its reduction (sched_barrier: the compiler otherwise sinks the reads to the next iteration's top
        //      and every pair pays a full LDS latency) ----
        if (GA) {
            // ---- the four corner dots D_ab = sum_ci dcol[ci] x_ab[ci] over this lane's 32 channels: dcb[s] = channels 16 s + 8 hh .. + 7
            //      = chunk 2 s + hh of a pixel ----
            float d4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int s = 0; s < 2; ++s) d4[k] = dom_dot8(ga[k][s], dcb[s], d4[k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {                    // channel block 1: chunks 4 and 6
                ga[k][0] = *reinterpret_cast<const u32x4w*>(Xw + (cc[k] ^ 64u));
                ga[k][1] = *reinterpret_cast<const u32x4w*>(Xw + (cc[k] ^ 96u));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int s = 0; s < 2; ++s) d4[k] = dom_dot8(ga[k][s], dcb[2 + s], d4[k]);
            if (active && inwin) {
                sm = (1.f - ly) * ((1.f - lx) * d4[0] + lx * d4[1]) + ly * ((1.f - lx) * d4[2] + lx * d4[3]);
                sy = (1.f - lx) * (d4[2] - d4[0]) + lx * (d4[3] - d4[1]);
                sx = (1.f - ly) * (d4[1] - d4[0]) + ly * (d4[3] - d4[2]);
            }
        } else         if (any) {
            while (true) {
                f32x16_t D;
#pragma unroll
                for (int r = 0; r < 16; ++r) D[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, xa[s]), __builtin_bit_cast(bf16x8_t, dcb[s]), D, 0, 0, 0);
                const int jc = j;
                const bool more = pm != 0;
                if (more) { j = __builtin_ctz(pm); pm &= pm - 1; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s) xa[s] = *reinterpret_cast<const u32x4w*>(xop[s] + j * (2 * DB_ROW));
                __builtin_amdgcn_sched_barrier(0);
                const int d0 = 2 * jc - wr;                  // row 2 jc is corner row d0 (0 = top, 1 = bottom) of this pixel, if either
                const float wy0 = d0 == 0 ? 1.f - ly : (d0 == 1 ? ly : 0.f), wy1 = d0 == -1 ? 1.f - ly : (d0 == 0 ? ly : 0.f);
                const float gy0 = d0 == 0 ? -1.f : (d0 == 1 ? 1.f : 0.f), gy1 = d0 == -1 ? -1.f : (d0 == 0 ? 1.f : 0.f);
                f32x2w r0 = {0.f, 0.f}, r1 = {0.f, 0.f};     // {sum wx D, sum gx D} of the pair's two rows
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    r0 += cg[i] * f32x2w{D[i], D[i]};
                    r1 += cg[i] * f32x2w{D[8 + i], D[8 + i]};
                }
                sm += wy0 * r0[0] + wy1 * r1[0];
                sx += wy0 * r0[1] + wy1 * r1[1];
                sy += gy0 * r0[0] + gy1 * r1[0];
                if (!more) break;
            }
        }
        DB_STAMP(4 + 4 * tap);
        // ---- rare paths: samples outside the window (exact VALU corner dots from global memory) and the dx_far scatter ----
        const int dh0 = wr - DB_MG - prow, dw0 = wc - DB_MG - pcol;                     // corner 00 relative to the pixel itself
        const bool far_h0 = dh0 > DCN_FAR_R || dh0 < -DCN_FAR_R, far_h1 = dh0 + 1 > DCN_FAR_R || dh0 + 1 < -DCN_FAR_R;
        const bool far_w0 = dw0 > DCN_FAR_R || dw0 < -DCN_FAR_R, far_w1 = dw0 + 1 > DCN_FAR_R || dw0 + 1 < -DCN_FAR_R;
        // any of the four: d0 outside [-R, R - 1] on either axis — one unsigned key per axis, one max, one compare (the eight signed
        // compares and their scalar ORs sat in every tap of every lane)
        const bool scatter = active && direct_far &&
                             max((unsigned)(dh0 + DCN_FAR_R), (unsigned)(dw0 + DCN_FAR_R)) > (unsigned)(2 * DCN_FAR_R - 1);
        if (__builtin_amdgcn_ballot_w64(winmiss || scatter) != 0) {
            if (winmiss || scatter) {
                const bool in_h0 = (unsigned)h0 < (unsigned)g.H, in_h1 = (unsigned)(h0 + 1) < (unsigned)g.H;
                const bool in_w0 = (unsigned)w0 < (unsigned)g.W, in_w1 = (unsigned)(w0 + 1) < (unsigned)g.W;
                const float a00 = (1.f - ly) * (1.f - lx), a01 = (1.f - ly) * lx, a10 = ly * (1.f - lx), a11 = ly * lx;
                const int hc0 = min(max(h0, 0), g.H - 1), hc1 = min(max(h0 + 1, 0), g.H - 1);
                const int wc0 = min(max(w0, 0), g.W - 1), wc1 = min(max(w0 + 1, 0), g.W - 1);
                if (winmiss) {
                    float d00 = 0.f, d01 = 0.f, d10 = 0.f, d11 = 0.f;
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            const int ch = 32 * cb + 16 * a + 8 * hh;
                            float c00[8], c01[8], c10[8], c11[8];
                            Vec16<bf16_t>::unpack(ldg16_masked(X, (((int64_t)hc0 * g.W + wc0) * g.x_ld + ch) * 2, in_h0 && in_w0), c00);
                            Vec16<bf16_t>::unpack(ldg16_masked(X, (((int64_t)hc0 * g.W + wc1) * g.x_ld + ch) * 2, in_h0 && in_w1), c01);
                            Vec16<bf16_t>::unpack(ldg16_masked(X, (((int64_t)hc1 * g.W + wc0) * g.x_ld + ch) * 2, in_h1 && in_w0), c10);
                            Vec16<bf16_t>::unpack(ldg16_masked(X, (((int64_t)hc1 * g.W + wc1) * g.x_ld + ch) * 2, in_h1 && in_w1), c11);
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float dv = dc[cb][8 * a + e];
                                d00 += dv * c00[e]; d01 += dv * c01[e]; d10 += dv * c10[e]; d11 += dv * c11[e];
                            }
                        }
                    sm += a00 * d00 + a01 * d01 + a10 * d10 + a11 * d11;
                    sy += (1.f - lx) * (d10 - d00) + lx * (d11 - d01);
                    sx += (1.f - ly) * (d01 - d00) + ly * (d11 - d10);
                }
                if (scatter) {
                    if (g.far_flag) *g.far_flag = 1;
                    float* far = g.far + (img + (int64_t)h0 * g.W + w0) * g.Ci + ci0;
                    const bool s00 = in_h0 && in_w0 && a00 != 0.f && (far_h0 || far_w0), s01 = in_h0 && in_w1 && a01 != 0.f && (far_h0 || far_w1);
                    const bool s10 = in_h1 && in_w0 && a10 != 0.f && (far_h1 || far_w0), s11 = in_h1 && in_w1 && a11 != 0.f && (far_h1 || far_w1);
#pragma unroll 1
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll 1
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const int ch = 32 * cb + 16 * a + 8 * hh + e;
                                const float gm = dc[cb][8 * a + e] * m;
                                if (s00) atomicAdd(far + ch, gm * a00);
                                if (s01) atomicAdd(far + g.Ci + ch, gm * a01);
                                if (s10) atomicAdd(far + (int64_t)g.W * g.Ci + ch, gm * a10);
                                if (s11) atomicAdd(far + (int64_t)(g.W + 1) * g.Ci + ch, gm * a11);
                            }
                }
            }
        }
        // ---- the two lane halves hold the sums over their columns / channels: combine, and park the results in the table entries
        //      they came from (this tap's entries were last read a tap ago) ----
        DB_STAMP(5 + 4 * tap);
        sm = half_sum(sm); sy = half_sum(sy); sx = half_sum(sx);
        if (hh == 0) {
            orow[2 * tap] = sy * m; orow[2 * tap + 1] = sx * m; orow[18 + tap] = sm * m * (1.f - m);
        }
        if (WLDS) {           // next tap's fragments into the other buffer (last read a tap ago), the tap after that on its way
            wq_store((tap + 1) & 1);
            __syncthreads();
            wq_load(tap < 7 ? tap + 2 : 8);
        }
    };
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) tap_body(tap, wfa);

    // ---- one coalesced row of 32 values per pixel (27 results + zero padding) ----
    DB_STAMP(38);
    __builtin_amdgcn_wave_barrier();
    if (g.dom16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = lane + 64 * i, p = idx >> 4, c = (idx & 15) * 2;
            const int oy = ty0 + grow + (p >> 3), ox = tx0 + gcol + (p & 7);
            const float v0 = c < 27 ? Om[p * 29 + c] : 0.f, v1 = c + 1 < 27 ? Om[p * 29 + c + 1] : 0.f;
            if (oy < g.H && ox < g.W) *reinterpret_cast<uint32_t*>(g.dom16 + (img + (int64_t)oy * g.W + ox) * 32 + c) = pk_bf16(v0, v1);
        }
    } else {
        float* const d = g.dom + (int64_t)blockIdx.y * g.slab;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = lane + 64 * i, p = idx >> 5, c = idx & 31;
            const int oy = ty0 + grow + (p >> 3), ox = tx0 + gcol + (p & 7);
            const float v = c < 27 ? Om[p * 29 + c] : 0.f;
            if (oy < g.H && ox < g.W) d[(img + (int64_t)oy * g.W + ox) * 32 + c] = v;
        }
    }
    DB_STAMP(39);
}

bool dcn_dom_bm_shape_ok(int Ci, int dy_ld, int x_ld, int om_ld) {
    static const bool disabled = getenv("CN_DISABLE_DCN_BM") != nullptr || getenv("CN_DISABLE_DCN_DOM_BM") != nullptr;
    return !disabled && Ci % 64 == 0 && (dy_ld == 64 || dy_ld == 128) && om_ld == 32 && x_ld % 8 == 0;
}

// returns false when the shape / result protocol is not handled here (caller falls back to dcn_bwd_dom_kernel)
bool dcn_dom_bm_launch(const void* dy, const void* wd2, const void* x, const float* om, float* dom, int dom_slabs, float* far, int* far_flag,
                       int N, int H, int W, int Ci, int dy_ld, int x_ld, int om_ld, hipStream_t st) {
    if (!dcn_dom_bm_shape_ok(Ci, dy_ld, x_ld, om_ld)) return false;
    if (((uintptr_t)dy | (uintptr_t)wd2 | (uintptr_t)x | (uintptr_t)om) & 15) return false;
    const int blocks = Ci / 64;
    DomBmGeom g;
    g.dy = (const bf16_t*)dy; g.wd2 = (const bf16_t*)wd2; g.x = (const bf16_t*)x; g.om = om; g.far = far; g.far_flag = far_flag;
    g.N = N; g.H = H; g.W = W; g.Ci = Ci; g.x_ld = x_ld;
    g.dom = dom; g.dom16 = nullptr; g.slab = 0;
    if (dom_slabs == 0) {                       // direct bf16 result: one channel block owns every row
        if (blocks != 1) return false;
        g.dom16 = (bf16_t*)dom;
    } else if (dom_slabs == blocks) {           // one fp32 copy per 64-channel block of x, plain stores
        g.slab = blocks > 1 ? (int64_t)N * H * W * om_ld : 0;
    } else {
        return false;                           // blocks meeting with atomics: the tile kernel's protocol
    }
    const int64_t tiles = (int64_t)((H + DB_TH - 1) / DB_TH) * ((W + DB_TW - 1) / DB_TW) * N;
    if (tiles > 0x7fffffff) return false;
    const dim3 grid((unsigned)tiles, blocks);
    const size_t smem = (size_t)DB_WR * DB_ROW + 4 * 32 * 29 * 4 + (dy_ld == 64 ? 2 * 8 * 1024 : 0);
    static const bool gather = getenv("CN_DISABLE_DOM_GATHER") == nullptr;
#define CN_DOMB(COP_, GA_) do { (void)hipFuncSetAttribute((const void*)dcn_dom_bm_kernel<COP_, GA_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
                                hipLaunchKernelGGL((dcn_dom_bm_kernel<COP_, GA_>), grid, dim3(256), smem, st, g); } while (0)
    if (dy_ld == 64) { if (gather) CN_DOMB(64, true); else CN_DOMB(64, false); }
    else { if (gather) CN_DOMB(128, true); else CN_DOMB(128, false); }
#undef CN_DOMB
    return true;
}
