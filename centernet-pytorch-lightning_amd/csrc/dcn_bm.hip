// DCNv2 with the bilinear blend ON THE MATRIX CORES ("blend-matrix" formulation; bf16, 64-channel blocks of x).
//
//   S_k[p][ci] = sum_q Bm_k[p][q] * x[q][ci]        Bm_k: 4 non-zeros per row = bilinear weights * mask of (pixel p, tap k)
//   y[p][co]   = bias[co] + sum_k sum_ci W_k[co][ci] * S_k[p][ci]                                  
//
// The gather kernels (dcn_fused.hip) blend on the VALU: per (pixel, tap, 8 channels) 4 L2 gathers, 32 bf16 unpacks, 32 FMAs, 4
// converts — ~580 VALU instructions per thread and tap against ~20 MFMAs; PMC and instruction counts say they are bound by that
// stream (docs/NEGATIVE_RESULTS.md).  Here the blend is a second MFMA: one WAVE owns a 4x8 group of output pixels and keeps the x WINDOW that
// group can sample (12 rows x 16 columns: +-4 pixels, i.e. offsets up to |d| <= 3 px for every tap) as TRANSPOSED MFMA
// fragments in registers for all nine taps (loaded once per group with ds_read_b64_tr_b16 from the workgroup's halo image in
// LDS).  One window row = 16 source pixels = the K of one v_mfma_f32_32x32x16_bf16, so per tap and touched window row
//   S^T[ci][p] += X^T[ci][row r, 16 cols] * Bm^T[row r, 16 cols][p]
// where a lane builds the 8 bf16 of ITS OWN pixel's blend-matrix row for that window row with a handful of selects (two packed
// weight pairs, shifted to the corner column once per tap).  S^T comes out of the matrix pipe with lane = pixel, register =
// channel — exactly the B operand of the contraction y^T[co][p] += W_k^T[co][ci] S^T[ci][p] when W is staged with the matching
// channel permutation — so the sampled operand never touches LDS, the VALU never unpacks bf16, and fp32 conversion is the
// matrix pipe's.  The bilinear weights enter the MFMA as bf16 (rel. 2^-9, the same size as the bf16 rounding of S itself);
// fp32 parity mode keeps the gather kernels.  Samples whose corners leave the window (|offset| > 3 px) are blended by a
// per-lane VALU fallback into the same registers (exact fp32 weights), so any offset field is handled.
#include "conv_common.h"
#include <stdlib.h>

#define BM_TH 8
#define BM_TW 16
#define BM_MG 4                          // window margin around a pixel group
#define BM_WR (BM_TH + 2 * BM_MG)        // 16 window rows per workgroup tile
#define BM_WC (BM_TW + 2 * BM_MG)        // 24 window columns
#define BM_GR 12                         // window rows of one 4x8 group
#define BM_PIXB 128                      // bytes per window pixel in LDS (64 bf16)

#ifdef BM_PROBE   // development build only (tools/bm_probe.py): cycle stamps of wave 0 of the first workgroups
__device__ unsigned long long bm_ts[1024 * 32];
#define BM_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 1024) bm_ts[blockIdx.x * 32 + (k)] = clock64(); } while (0)
extern "C" int bm_probe_dump(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(bm_ts), sizeof(bm_ts)); }
#else
#define BM_STAMP(k) do { } while (0)
#endif

struct BmGeom {
    const bf16_t* x; const float* om; const bf16_t* wp; const float* bias; bf16_t* y;
    int N, H, W, x_ld, y_ld, om_ld, ktot, Co, relu, Ci;
    float* bn_part; int bn_slots;        // BatchNorm statistics sink (cn_hooks.bn_part), nullable
};

typedef short s16x4_t_ __attribute__((ext_vector_type(4)));
typedef short s16x8_t_ __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

// LDS byte address of (window row, window col, channel c) in the halo image: 128 B per pixel, the two 64-byte halves swapped on
// every other column PAIR so that the 4 consecutive columns a transposing read touches hit 4 disjoint 16-bank segments
__device__ static inline int bm_lds_ofs(int wr, int wc, int c) {
    return (wr * BM_WC + wc) * BM_PIXB + ((((c >> 5) ^ (wc >> 1)) & 1) << 6) + (c & 31) * 2;
}

template <int NCB, bool MB = false>   // 32-channel output blocks (Co = 32 * NCB); MB: more than one 64-channel block of x
__global__ __launch_bounds__(256, 2) void dcn_fwd_bm_kernel(const BmGeom g) {
    CN_MAIN_PRIO_SET();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Xw = smem;                                   // [16][24] pixels x 128 B
    unsigned char* const Ws = smem + BM_WR * BM_WC * BM_PIXB;         // 2 x [4 k-steps][2 halves][32*NCB co][8] bf16
    constexpr int WSB = 4 * 2 * 32 * NCB * 16;                        // bytes per weight buffer
    // byte-permute selectors that place a packed weight pair at 16-bit element c of an 8-element (4-dword) vector, c = -8 .. 14:
    // dword d = pair (s == 0), pair << 16 (s == 1), pair >> 16 (s == -1) or 0, s = c - 2d   (v_perm_b32; selector 0x0c = 0x00)
    u32x4v* const Lut = reinterpret_cast<u32x4v*>(Ws + 2 * WSB);      // [23]
    if (threadIdx.x < 23) {
        const int c = (int)threadIdx.x - 8;
        u32x4v sel;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int sft = c - 2 * d;
            sel[d] = sft == 0 ? 0x03020100u : (sft == 1 ? 0x01000c0cu : (sft == -1 ? 0x0c0c0302u : 0x0c0c0c0cu));
        }
        Lut[threadIdx.x] = sel;
    }

    BM_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_w = (g.W + BM_TW - 1) / BM_TW;
    const int ty0 = (blockIdx.x / tiles_w) * BM_TH, tx0 = (blockIdx.x % tiles_w) * BM_TW;
    const int n = blockIdx.y;
    const int64_t img = (int64_t)n * g.H * g.W;
    const bf16_t* __restrict__ X = g.x + img * g.x_ld;
    const float* __restrict__ OM = g.om + img * g.om_ld;

    // ---- weight slice of a tap: global (mode-1 pack [co][tap*64 + ci]) -> registers -> LDS in fragment order ----
    // slot = (s, h, co): 8 bf16 = W[co][ci(s,h,e)], ci = 32*(s>>1) + 16*(s&1) + 8*(e>>2) + 4*h + (e&3)   (the order S^T's registers have)
    constexpr int WSLOTS = 4 * 2 * 32 * NCB;
    constexpr int WPT = (WSLOTS + 255) / 256;
    uint2 wr_[WPT][2];
    // Ci = 64 * nblk (round 5: 128 / 256 input channels with Co <= 64 — the 128 -> 64 and 256 -> 64 projections of IDAUp — used to run on
    // the gather kernel at 0.08 of the MFMA peak): the 64-channel blocks of x one after the other, each with its own halo image, window
    // fragments and nine taps, all adding into the same accumulators; the geometry table and the selector table are built once.
    // (a separate instantiation: the one-block kernel sits at 256 registers and the loop around it costs it 50 spilled dwords)
    const int nblk = MB ? g.Ci >> 6 : 1;
    int blk = 0;
    auto wload = [&](int tap) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int slot = tid + i * 256;
            const int co = slot % (32 * NCB), sh = slot / (32 * NCB), h = sh & 1, s = sh >> 1;
            const int ci0 = 32 * (s >> 1) + 16 * (s & 1) + 4 * h;
            const bf16_t* p = g.wp + (int64_t)co * g.ktot + tap * g.Ci + blk * 64 + ci0;
            wr_[i][0] = *reinterpret_cast<const uint2*>(p);
            wr_[i][1] = *reinterpret_cast<const uint2*>(p + 8);
        }
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int slot = tid + i * 256;
            *reinterpret_cast<u32x4v*>(Ws + buf * WSB + slot * 16) = u32x4v{wr_[i][0].x, wr_[i][0].y, wr_[i][1].x, wr_[i][1].y};
        }
    };
    // ---- this wave's pixel group ----
    const int grow = (wave >> 1) * 4, gcol = (wave & 1) * 8;       // group origin inside the tile == its window origin in the halo image
    const int nl = lane & 31, hh = lane >> 5;
    const int gy = ty0 + grow + (nl >> 3), gx = tx0 + gcol + (nl & 7);
    const bool live = gy < g.H && gx < g.W;
    // Issue order = the order the loads are needed in (they return in order): offsets / mask logits, the halo image, the first weight
    // slice, the bias.
    // offsets / mask logits of the group's 32 pixels (om_ld == 32: one 128-byte row per pixel): four coalesced 16-byte loads per
    // lane, parked in an LDS table of the wave — the tap loop then has no global load on its critical path (a per-tap global
    // prefetch made the compiler wait for the weight prefetch as well: vmcnt(0) at every loop top)
    float4 omr[4];
    const int part = lane & 7;                      // which 4 of a pixel's 32 table entries this lane converts (the same for all four loads)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = (lane >> 3) + 8 * i;
        const int py_ = ty0 + grow + (p >> 3), px_ = tx0 + gcol + (p & 7);
        const bool ok = py_ < g.H && px_ < g.W;
        omr[i] = *reinterpret_cast<const float4*>(OM + ((int64_t)(ok ? py_ : 0) * g.W + (ok ? px_ : 0)) * 32 + part * 4);
    }
    // halo image of the tile: rows ty0-4 .. ty0+11, columns tx0-4 .. tx0+19, zeros outside the image; all twelve loads in flight
    constexpr int NV = BM_WR * BM_WC * 8 / 256;     // 16-byte vectors per thread
    uint4 hv[NV];
    auto hload = [&]() {                            // halo image of channel block `blk`: global -> registers
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * 256;
            const int pix = v >> 3, q = v & 7;
            const int wr = pix / BM_WC, wc = pix % BM_WC;
            const int hy = ty0 - BM_MG + wr, hx = tx0 - BM_MG + wc;
            const bool ok = (unsigned)hy < (unsigned)g.H && (unsigned)hx < (unsigned)g.W;
            hv[i] = ldg16_masked(X, (((int64_t)hy * g.W + hx) * g.x_ld + blk * 64 + q * 8) * 2, ok);
        }
    };
    auto hstore = [&]() {                           // ... -> LDS
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * 256;
            const int pix = v >> 3, q = v & 7;
            st16(Xw + bm_lds_ofs(pix / BM_WC, pix % BM_WC, q * 8), hv[i]);
        }
    };
    hload();
    wload(0);
    // the bias is the accumulators' initial value (lane = pixel, register v of block cb = channel 32 cb + 8 (v >> 2) + 4 hh + (v & 3))
    f32x16_t acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bv = *reinterpret_cast<const float4*>(g.bias + 32 * cb + 8 * q + 4 * hh);
            acc[cb][4 * q] = bv.x; acc[cb][4 * q + 1] = bv.y; acc[cb][4 * q + 2] = bv.z; acc[cb][4 * q + 3] = bv.w;
        }

    // the table holds what a tap needs, finished: entry 2k / 2k+1 = sampling position of tap k RELATIVE TO THE GROUP'S WINDOW ORIGIN
    // (pixel row + 3 + ky + dy: corner 00 of a zero offset sits in window row 3..8), entry 18+k = sigmoid(mask logit), 0 outside the image.
    // Lanes with part >= 4 hold (mostly) mask logits, the others positions: one divergent branch for the transcendental work.
    float* const Om = reinterpret_cast<float*>(Ws + 2 * WSB + 512) + wave * (32 * 29);      // [32 px][29]: odd pitch, conflict-free per-pixel reads
    {
        float tv[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = (lane >> 3) + 8 * i;
            const float v4[4] = {omr[i].x, omr[i].y, omr[i].z, omr[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = part * 4 + j, k = e >> 1, k3 = (k * 11) >> 5;          // k / 3 for k < 9
                tv[i][j] = v4[j] + (float)((e & 1) ? (p & 7) + 3 + (k - 3 * k3) : (p >> 3) + 3 + k3);
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
    hstore();
    wstore(0);
    __syncthreads();
    BM_STAMP(1);
    wload(1);

    const float* const orow = Om + nl * 29;
#pragma unroll 1
    for (;;) {          // channel blocks of x
    // ---- the group's window fragments ----
    bf16x8_t xf[BM_GR][2];
    {
        const int r16 = lane & 15, g16 = lane >> 4;
        typedef __attribute__((address_space(3))) s16x4_t_* lds_ptr;
        // two base addresses (the swizzle makes the channel-block step +-64 B per lane); rows and the +4-column half are immediates
        const int wc0 = gcol + 8 * (g16 >> 1) + (r16 >> 2);
        const unsigned char* const b0 = Xw + bm_lds_ofs(grow, wc0, 16 * (g16 & 1) + 4 * (r16 & 3));
        const unsigned char* const b1 = Xw + bm_lds_ofs(grow, wc0, 32 + 16 * (g16 & 1) + 4 * (r16 & 3));
#pragma unroll
        for (int r = 0; r < BM_GR; ++r)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const unsigned char* p = (mb ? b1 : b0) + r * (BM_WC * BM_PIXB);
                const s16x4_t_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
                const s16x4_t_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * BM_PIXB));
                const s16x8_t_ v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                xf[r][mb] = __builtin_bit_cast(bf16x8_t, v);
            }
    }
    float ro[3] = {orow[0], orow[1], orow[18]};
    BM_STAMP(2);
    __syncthreads();        // every wave holds its fragments: the halo image is dead (re-used as far-sample scratch, 8 KB per wave)

#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        BM_STAMP(4 + 3 * tap);
        // ---- geometry of (own pixel, tap): window position of corner 00 and the two packed weight pairs ----
        // (no image-border tests here: the halo image holds zeros outside the image; only the far path below needs them)
        const float pyr = ro[0], pxr = ro[1], m = ro[2];
        {   // next tap's entries of the LDS table
            const int nt = tap < 8 ? tap + 1 : 8;
            ro[0] = orow[2 * nt]; ro[1] = orow[2 * nt + 1]; ro[2] = orow[18 + nt];
        }
        const float fy = floorf(pyr), fx = floorf(pxr);
        const int wr = (int)fy, wc = (int)fx;                                           // window coordinates of corner 00
        const float ly = pyr - fy, lx = pxr - fx;
        const float wa = (1.f - ly) * m, wbt = ly * m;
        const bool inwin = (unsigned)wr <= (unsigned)(BM_GR - 2) && (unsigned)wc <= 14u;
        const bool far = !inwin && m != 0.f;
        const uint32_t P0 = inwin ? pk_bf16(wa * (1.f - lx), wa * lx) : 0u, P1 = inwin ? pk_bf16(wbt * (1.f - lx), wbt * lx) : 0u;
        const int wr_top = (P0 != 0u) ? wr : -1, wr_bot = (P1 != 0u) ? wr + 1 : -1;
        // 8-element (4-dword) images of the two weight pairs for this lane's column half: pair element a sits at column wc
        const u32x4v sel = Lut[min(max(wc - 8 * hh + 8, 0), 22)];
        uint32_t V0[4], V1[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            V0[d] = __builtin_amdgcn_perm(P0, P0, sel[d]);
            V1[d] = __builtin_amdgcn_perm(P1, P1, sel[d]);
        }
        // window rows anybody in the wave samples (wave-wide OR of the lanes' row bits: 4 DPP steps per 16-lane row + 4 readlanes)
        uint32_t rows = ((P0 != 0u) ? (1u << (wr & 15)) : 0u) | ((P1 != 0u) ? (2u << (wr & 15)) : 0u);
        rows |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rows, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
        rows |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rows, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
        rows |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rows, 0x141, 0xF, 0xF, true);    // row_half_mirror
        rows |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rows, 0x140, 0xF, 0xF, true);    // row_mirror
        const uint32_t rowmask = (uint32_t)__builtin_amdgcn_readlane((int)rows, 0) | (uint32_t)__builtin_amdgcn_readlane((int)rows, 16) |
                                 (uint32_t)__builtin_amdgcn_readlane((int)rows, 32) | (uint32_t)__builtin_amdgcn_readlane((int)rows, 48);

        BM_STAMP(5 + 3 * tap);
        // ---- S^T[ci][p] = sum over the touched window rows (K = the row's 16 source columns) ----
        f32x16_t st[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[mb][r] = 0.f;
        // the zeros must be real registers: folded into the first MFMA (C = 0) the compiler merges the skip branches below with 32
        // accumulator copies per row
        asm volatile("" : "+v"(st[0]), "+v"(st[1]));
#pragma unroll
        for (int r = 0; r < BM_GR; ++r) {
            if (!(rowmask & (1u << r))) continue;                              // wave-uniform: nobody samples this row
            const bool t0 = wr_top == r, t1 = wr_bot == r;
            u32x4v b;
#pragma unroll
            for (int d = 0; d < 4; ++d) b[d] = t0 ? V0[d] : (t1 ? V1[d] : 0u);
            const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, b);
            st[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[r][0], bf, st[0], 0, 0, 0);
            st[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[r][1], bf, st[1], 0, 0, 0);
        }
        // ---- samples that leave the window (rare): exact fp32 VALU blend from global memory.  The lane pair of the pixel writes the
        //      blended 64-channel row into the wave's slice of the (by now dead) halo image, then adds it to its S^T registers ----
        if (__builtin_amdgcn_ballot_w64(far) != 0) {
            float* const F = reinterpret_cast<float*>(Xw + wave * (32 * 64 * 4)) + nl * 64;
            if (far) {
                const int h0 = wr + (ty0 + grow - BM_MG), w0 = wc + (tx0 + gcol - BM_MG);
                const bool h0ok = (unsigned)h0 < (unsigned)g.H, h1ok = (unsigned)(h0 + 1) < (unsigned)g.H;
                const bool w0ok = (unsigned)w0 < (unsigned)g.W, w1ok = (unsigned)(w0 + 1) < (unsigned)g.W;
                const float w00 = (h0ok && w0ok) ? wa * (1.f - lx) : 0.f, w01 = (h0ok && w1ok) ? wa * lx : 0.f;
                const float w10 = (h1ok && w0ok) ? wbt * (1.f - lx) : 0.f, w11 = (h1ok && w1ok) ? wbt * lx : 0.f;
                const int hc0 = min(max(h0, 0), g.H - 1), hc1 = min(max(h0 + 1, 0), g.H - 1);
                const int wc0 = min(max(w0, 0), g.W - 1), wc1 = min(max(w0 + 1, 0), g.W - 1);
                const bf16_t* p00 = X + ((int64_t)hc0 * g.W + wc0) * g.x_ld + blk * 64 + 32 * hh;
                const bf16_t* p01 = X + ((int64_t)hc0 * g.W + wc1) * g.x_ld + blk * 64 + 32 * hh;
                const bf16_t* p10 = X + ((int64_t)hc1 * g.W + wc0) * g.x_ld + blk * 64 + 32 * hh;
                const bf16_t* p11 = X + ((int64_t)hc1 * g.W + wc1) * g.x_ld + blk * 64 + 32 * hh;
#pragma unroll 1
                for (int q = 0; q < 8; ++q) {          // this lane's half of the row: channels 32*hh + 4q .. +3
                    const uint2 a = *reinterpret_cast<const uint2*>(p00 + 4 * q), b = *reinterpret_cast<const uint2*>(p01 + 4 * q);
                    const uint2 cc = *reinterpret_cast<const uint2*>(p10 + 4 * q), d = *reinterpret_cast<const uint2*>(p11 + 4 * q);
                    float4 o;
                    o.x = __uint_as_float(a.x << 16) * w00 + __uint_as_float(b.x << 16) * w01 + __uint_as_float(cc.x << 16) * w10 + __uint_as_float(d.x << 16) * w11;
                    o.y = __uint_as_float(a.x & 0xffff0000u) * w00 + __uint_as_float(b.x & 0xffff0000u) * w01 + __uint_as_float(cc.x & 0xffff0000u) * w10 + __uint_as_float(d.x & 0xffff0000u) * w11;
                    o.z = __uint_as_float(a.y << 16) * w00 + __uint_as_float(b.y << 16) * w01 + __uint_as_float(cc.y << 16) * w10 + __uint_as_float(d.y << 16) * w11;
                    o.w = __uint_as_float(a.y & 0xffff0000u) * w00 + __uint_as_float(b.y & 0xffff0000u) * w01 + __uint_as_float(cc.y & 0xffff0000u) * w10 + __uint_as_float(d.y & 0xffff0000u) * w11;
                    *reinterpret_cast<float4*>(F + 32 * hh + 4 * q) = o;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (far) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4*>(F + 32 * mb + 8 * q + 4 * hh);
                        st[mb][4 * q] += v.x; st[mb][4 * q + 1] += v.y; st[mb][4 * q + 2] += v.z; st[mb][4 * q + 3] += v.w;
                    }
            }
            __builtin_amdgcn_wave_barrier();
        }

        BM_STAMP(6 + 3 * tap);
        // ---- y^T[co][p] += W_k^T[co][ci] S^T[ci][p]: S^T's registers ARE the B operand (k-step s = registers 8*(s&1)..+7 of block s>>1) ----
        const unsigned char* wb = Ws + (tap & 1) * WSB;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4v sb;
#pragma unroll
            for (int d = 0; d < 4; ++d) sb[d] = pk_bf16(st[s >> 1][8 * (s & 1) + 2 * d], st[s >> 1][8 * (s & 1) + 2 * d + 1]);
            const bf16x8_t sf = __builtin_bit_cast(bf16x8_t, sb);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const u32x4v wv = *reinterpret_cast<const u32x4v*>(wb + (((s * 2 + hh) * 32 * NCB) + cb * 32 + nl) * 16);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wv), sf, acc[cb], 0, 0, 0);
            }
        }
        // next tap's weights: the other buffer was last read in tap-1, a barrier ago
        wstore((tap + 1) & 1);                // (after tap 8 this writes a buffer nobody reads: branch-free on purpose)
        __syncthreads();
        wload(tap < 7 ? tap + 2 : 8);
    }
    if (!MB || ++blk == nblk) break;
    // next 64-channel block: everybody is behind tap 8's barrier — the halo image (far-sample scratch included) and weight buffer 0 are free
    hload();
    wload(0);
    hstore();
    wstore(0);
    __syncthreads();
    wload(1);
    }   // channel blocks

    BM_STAMP(3);
    // ---- epilogue: lane = pixel, registers = 4 consecutive channels per (block, quad): ReLU, 8-byte stores (the bias was the initial value) ----
    // ... through the wave's slice of the dead halo image ([32 px][72] bf16), so that a pixel's 128 bytes leave as eight 16-byte lanes
    {
        unsigned char* const Y = Xw + wave * 8192;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4] = {acc[cb][4 * q], acc[cb][4 * q + 1], acc[cb][4 * q + 2], acc[cb][4 * q + 3]};
                if (g.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                uint2 o;
                o.x = pk_bf16(v[0], v[1]); o.y = pk_bf16(v[2], v[3]);
                *reinterpret_cast<uint2*>(Y + nl * 144 + (32 * cb + 8 * q + 4 * hh) * 2) = o;
            }
        __builtin_amdgcn_wave_barrier();
        constexpr int CPP = 4 * NCB;                 // 16-byte chunks per pixel
        // BN statistics of the stored values (sink protocol of bn.hip): a lane visits the same 8-channel chunk in every pass
        const bool stats = g.bn_part != nullptr;
        float s0[8], s1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < 32 * CPP / 64; ++i) {
            const int idx = lane + 64 * i, p = idx / CPP, ch = idx % CPP;
            const int oy = ty0 + grow + (p >> 3), ox = tx0 + gcol + (p & 7);
            const u32x4v o = *reinterpret_cast<const u32x4v*>(Y + p * 144 + ch * 16);
            if (oy < g.H && ox < g.W) {
                *reinterpret_cast<u32x4v*>(g.y + (img + (int64_t)oy * g.W + ox) * g.y_ld + ch * 8) = o;
                if (stats) { const uint32_t w[4] = {o[0], o[1], o[2], o[3]}; bn_stat_add(s0, s1, w); }
            }
        }
        if (stats)      // (the flush barriers first: every wave is done with its slice of the halo image, which becomes the scratch)
            bn_stats_flush<CPP, 256>(s0, s1, reinterpret_cast<float*>(Xw), g.bn_part, g.bn_slots, g.y_ld, 0, g.Co,
                                     blockIdx.x + blockIdx.y * gridDim.x, tid);
    }
    BM_STAMP(31);
}

bool dcn_fwd_bm_shape_ok(int Ci, int x_ld, int Co, int y_ld, int om_ld) {
    static const bool disabled = getenv("CN_DISABLE_DCN_BM") != nullptr || getenv("CN_DISABLE_DCN_FWD_BM") != nullptr;
    static const bool no_blocks = getenv("CN_DCN_FWD_BM_ONE_BLOCK") != nullptr;      // A/B: Ci = 128 / 256 on the gather kernel as before round 5
    return !disabled && (Ci == 64 || (!no_blocks && (Ci == 128 || Ci == 256))) && x_ld == Ci && om_ld == 32 && (Co == 64 || Co == 32) && y_ld == Co;
}

// returns false when the shape is not handled here (caller falls back to the gather / LDS-tile kernels)
bool dcn_fwd_bm_launch(const void* x, const float* om, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci, int x_ld,
                       int Co, int y_ld, int om_ld, int ktot, int relu, float* bn_part, int bn_slots, int* bn_taken, hipStream_t st) {
    if (!dcn_fwd_bm_shape_ok(Ci, x_ld, Co, y_ld, om_ld) || bias == nullptr || ((uintptr_t)om & 15) || ktot != 9 * Ci || N > 65535) return false;
    if (((uintptr_t)x | (uintptr_t)wp | (uintptr_t)y) & 15) return false;
    if ((uintptr_t)bias & 15) return false;
    BmGeom g;
    g.x = (const bf16_t*)x; g.om = om; g.wp = (const bf16_t*)wp; g.bias = bias; g.y = (bf16_t*)y;
    g.N = N; g.H = H; g.W = W; g.x_ld = x_ld; g.y_ld = y_ld; g.om_ld = om_ld; g.ktot = ktot; g.Co = Co; g.relu = relu; g.Ci = Ci;
    g.bn_part = bn_part; g.bn_slots = bn_slots;
    if (bn_part) mark_taken(bn_taken);
    const dim3 grid(((H + BM_TH - 1) / BM_TH) * ((W + BM_TW - 1) / BM_TW), N);
#define CN_FWDBM(NCB_, MB_, SMEM_) do { (void)hipFuncSetAttribute((const void*)dcn_fwd_bm_kernel<NCB_, MB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SMEM_)); \
                                        hipLaunchKernelGGL((dcn_fwd_bm_kernel<NCB_, MB_>), grid, dim3(256), (SMEM_), st, g); } while (0)
    if (Co == 64) {
        const size_t smem = (size_t)BM_WR * BM_WC * BM_PIXB + 2 * (4 * 2 * 64 * 16) + 512 + 4 * 32 * 29 * 4;
        if (Ci > 64) CN_FWDBM(2, true, smem); else CN_FWDBM(2, false, smem);
    } else {
        const size_t smem = (size_t)BM_WR * BM_WC * BM_PIXB + 2 * (4 * 2 * 32 * 16) + 512 + 4 * 32 * 29 * 4;
        if (Ci > 64) CN_FWDBM(1, true, smem); else CN_FWDBM(1, false, smem);
    }
#undef CN_FWDBM
    return true;
}


// ================================================================================================ data gradient
// dx[q][ci] = sum_k sum_co W_k[co][ci] * G_k[q][co],   G_k[q][co] = sum_p Bm_k[p][q] * dY[p][co]        (adjoint of the sampling)
// Same skeleton with the roles turned round: a wave owns a 4x8 group of DESTINATION pixels q and keeps the dY window of every
// source p that may reach them (|q - p| <= 3 in both directions, the rule cn_dcn_bwd_dom uses for its dx_far scatter) as
// transposed fragments in registers.  The blend matrix of a tap is wanted with lane = destination, elements = sources, but is
// KNOWN per source (the offsets belong to p): per tap the 12x16 window is walked in three passes of 64 sources (one per lane);
// a lane computes its source's four corners and drops the bf16 weights of those that land in the group into a small LDS tile
// T[32 q][64 p] (zeroed per pass), which is read back row-wise — no transposition, no hit lists, no list walk, no atomics — as
// the B operand of G^T[co][q] += dY^T[co][p-row] * T^T.  G^T's registers are again the B operand of the contraction
// dx^T[ci][q] += W_k[.][ci]^T G^T.  The cost no longer depends on the offset field: the adjoint-gather kernel (dcn_fused.hip)
// takes 541 us with zero offsets and 955 us with N(0, 0.5 px) offsets on 64->64 @128^2 (one hit per destination and tap vs four).
// The source geometry of all nine taps (tile-relative sampling position, sigmoid(mask)) is computed once per tile into an LDS
// table that re-uses the halo image once the fragments are loaded.
#ifdef DXB_PROBE   // development build only (tools/dxbm_probe.py)
__device__ unsigned long long dxb_ts[1024 * 40];
#define DXB_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 1024) dxb_ts[blockIdx.x * 40 + (k)] = clock64(); } while (0)
extern "C" int dxb_probe_dump(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(dxb_ts), sizeof(dxb_ts)); }
#else
#define DXB_STAMP(k) do { } while (0)
#endif

struct DxBmGeom {
    const bf16_t* dy; const float* om; const bf16_t* wp; float* far; int* far_flag; bf16_t* dx;
    int N, H, W, dx_ld, ktot;
};

#define DXB_TR 14                         // table rows / cols: sources within 3 px of the tile
#define DXB_TC 22
#define DXB_TBYTES 33280                  // 14 * 22 * 27 floats = 33264, rounded up to 16
#define DXB_TP 144                        // byte pitch of a T row (64 sources x bf16 + 16: conflict-free 16-byte row reads)

template <int NCB>   // 32-channel blocks of dx (Ci = 32 * NCB); dY has 64 channels
__global__ __launch_bounds__(256, 2) void dcn_dx_bm_kernel(const DxBmGeom g) {
    CN_MAIN_PRIO_SET();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Yw = smem;                                   // halo image of dY [16][24] x 128 B; later: geometry table | T tiles
    float* const Tab = reinterpret_cast<float*>(smem);                // [14*22 sources][27]
    unsigned char* const Tt = smem + DXB_TBYTES;                      // 4 waves x [32 q][DXB_TP]
    unsigned char* const Ws = smem + DXB_TBYTES + 4 * 32 * DXB_TP;    // 2 weight buffers (behind the halo image: 51712 > 49152)
    constexpr int WSB = 4 * 2 * 32 * NCB * 16;

    DXB_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_w = (g.W + BM_TW - 1) / BM_TW;
    const int ty0 = (blockIdx.x / tiles_w) * BM_TH, tx0 = (blockIdx.x % tiles_w) * BM_TW;
    const int n = blockIdx.y;
    const int64_t img = (int64_t)n * g.H * g.W;
    const bf16_t* __restrict__ DY = g.dy + img * 64;
    const float* __restrict__ OM = g.om + img * 32;

    constexpr int WSLOTS = 4 * 2 * 32 * NCB;
    constexpr int WPT = (WSLOTS + 255) / 256;
    uint2 wr_[WPT][2];
    auto wload = [&](int tap) {          // wp = mode-0 pack [ci][tap*64 + co]: rows = this kernel's output channels
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int slot = tid + i * 256;
            const int ci = slot % (32 * NCB), sh = slot / (32 * NCB), h = sh & 1, s = sh >> 1;
            const int c0 = 32 * (s >> 1) + 16 * (s & 1) + 4 * h;
            const bf16_t* p = g.wp + (int64_t)(32 * NCB * blockIdx.z + ci) * g.ktot + tap * 64 + c0;     // blockIdx.z: 64-channel block of dx
            wr_[i][0] = *reinterpret_cast<const uint2*>(p);
            wr_[i][1] = *reinterpret_cast<const uint2*>(p + 8);
        }
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int slot = tid + i * 256;
            *reinterpret_cast<u32x4v*>(Ws + buf * WSB + slot * 16) = u32x4v{wr_[i][0].x, wr_[i][0].y, wr_[i][1].x, wr_[i][1].y};
        }
    };

    const int grow = (wave >> 1) * 4, gcol = (wave & 1) * 8;
    const int nl = lane & 31, hh = lane >> 5;
    const int gy = ty0 + grow + (nl >> 3), gx = tx0 + gcol + (nl & 7);
    const bool live = gy < g.H && gx < g.W;

    // ---- loads, in the order they are needed: halo image of dY, first weight slice, offsets / masks of the 14x22 sources ----
    constexpr int NV = BM_WR * BM_WC * 8 / 256;
    uint4 hv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * 256;
        const int pix = v >> 3, q = v & 7;
        const int hy = ty0 - BM_MG + pix / BM_WC, hx = tx0 - BM_MG + pix % BM_WC;
        const bool ok = (unsigned)hy < (unsigned)g.H && (unsigned)hx < (unsigned)g.W;
        hv[i] = ldg16_masked(DY, (((int64_t)hy * g.W + hx) * 64 + q * 8) * 2, ok);
    }
    wload(0);
    constexpr int NSRC = DXB_TR * DXB_TC, NOM = (NSRC + 31) / 32;     // 308 sources, 32 per pass: thread = (source, 16-byte part)
    const int part = tid & 7;
    float4 omr[NOM];
#pragma unroll
    for (int i = 0; i < NOM; ++i) {
        const int src = (tid >> 3) + 32 * i;
        const int sy = ty0 - 3 + src / DXB_TC, sx = tx0 - 3 + src % DXB_TC;
        const bool ok = src < NSRC && (unsigned)sy < (unsigned)g.H && (unsigned)sx < (unsigned)g.W && part < 7;
        omr[i] = *reinterpret_cast<const float4*>(OM + ((int64_t)(ok ? sy : 0) * g.W + (ok ? sx : 0)) * 32 + (part < 7 ? part : 0) * 4);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * 256;
        const int pix = v >> 3, q = v & 7;
        st16(Yw + bm_lds_ofs(pix / BM_WC, pix % BM_WC, q * 8), hv[i]);
    }
    __syncthreads();
    DXB_STAMP(32);

    // ---- the group's dY window as transposed fragments (rows = co) ----
    bf16x8_t yf[BM_GR][2];
    {
        const int r16 = lane & 15, g16 = lane >> 4;
        typedef __attribute__((address_space(3))) s16x4_t_* lds_ptr;
        const int wc0 = gcol + 8 * (g16 >> 1) + (r16 >> 2);
        const unsigned char* const b0 = Yw + bm_lds_ofs(grow, wc0, 16 * (g16 & 1) + 4 * (r16 & 3));
        const unsigned char* const b1 = Yw + bm_lds_ofs(grow, wc0, 32 + 16 * (g16 & 1) + 4 * (r16 & 3));
#pragma unroll
        for (int r = 0; r < BM_GR; ++r)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const unsigned char* p = (mb ? b1 : b0) + r * (BM_WC * BM_PIXB);
                const s16x4_t_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
                const s16x4_t_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * BM_PIXB));
                const s16x8_t_ v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                yf[r][mb] = __builtin_bit_cast(bf16x8_t, v);
            }
    }
    __syncthreads();        // the halo image is dead: geometry table + T tiles + first weight slice go in
    DXB_STAMP(33);

    // table entry 2k / 2k+1 of a source: sampling position of tap k relative to the TILE origin (row - ty0 - 1 + ky + dy),
    // entry 18+k: sigmoid(mask logit), 0 for sources outside the image
#pragma unroll
    for (int i = 0; i < NOM; ++i) {
        const int src = (tid >> 3) + 32 * i;
        const int sr = src / DXB_TC - 3, sc = src % DXB_TC - 3;               // tile-relative source coordinates
        const bool inimg = (unsigned)(ty0 + sr) < (unsigned)g.H && (unsigned)(tx0 + sc) < (unsigned)g.W;
        const float v4[4] = {omr[i].x, omr[i].y, omr[i].z, omr[i].w};
        float* d = Tab + src * 27 + part * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = part * 4 + j, k = e >> 1, k3 = (k * 11) >> 5;
            float val = v4[j] + (float)((e & 1) ? sc - 1 + (k - 3 * k3) : sr - 1 + k3);
            if (part >= 4 && e >= 18) val = inimg ? __builtin_amdgcn_rcpf(1.f + __expf(-v4[j])) : 0.f;
            if (src < NSRC && e < 27) d[j] = val;
        }
    }
    wstore(0);
    wload(1);
    __syncthreads();
    DXB_STAMP(1);

    f32x16_t acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;

    unsigned char* const T = Tt + wave * (32 * DXB_TP);
    // this lane's source in pass j: group-window row 4j + (lane >> 4), column lane & 15 (window origin = group origin - 4)
    const int s_col = lane & 15, s_row0 = lane >> 4;
    const bool col_ok = s_col >= 1 && s_col <= 14;                           // |q - p| <= 3 needs window columns 1..14
    const int tcol = gcol + s_col - 1;                                       // table column (tile-window column - 1)
    bool src_ok[3];                                                          // pass j: this lane's window position is a source (rows 1..10)
#pragma unroll
    for (int j = 0; j < 3; ++j) src_ok[j] = col_ok && 4 * j + s_row0 >= 1 && 4 * j + s_row0 <= 10;

#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        DXB_STAMP(2 + 3 * tap);
        f32x16_t st[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[mb][r] = 0.f;
        asm volatile("" : "+v"(st[0]), "+v"(st[1]));
        // the three passes' table entries in one batch of LDS reads (clamped addresses for the lanes without a source): a pass then
        // has ONE dependent LDS round trip (its T tile) instead of two
        float tpy[3], tpx[3], tpm[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int sr = min(max(4 * j + s_row0, 1), 10), tc = min(max(tcol, 0), DXB_TC - 1);
            const float* te = Tab + ((grow + sr - 1) * DXB_TC + tc) * 27;
            tpy[j] = te[2 * tap]; tpx[j] = te[2 * tap + 1]; tpm[j] = te[18 + tap];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            // ---- zero the T tile (4608 B), then every source lane drops its corners' weights ----
#pragma unroll
            for (int z = 0; z < 5; ++z)
                if (z < 4 || lane < 32) *reinterpret_cast<u32x4v*>(T + (lane + 64 * z) * 16) = u32x4v{0u, 0u, 0u, 0u};
            __builtin_amdgcn_wave_barrier();
            const int s_row = 4 * j + s_row0;                                // group-window row of the source
            bool any = false;
            // Row test first: a source whose two sampling rows (floor(py), and floor(py) + 1 when py has a fraction) miss the group's
            // four rows has nothing to scatter.  When that holds for every lane of the pass — with offsets near zero a tap's sources
            // sit in 4-6 of the 12 window rows, i.e. in one or two of the three passes — the wave skips the rest of the geometry.
            const float pyt = tpy[j];
            const float fy = floorf(pyt), ly = pyt - fy;
            const int y0 = (int)fy - grow;                                    // destination row of corner 00, group-local
            if (src_ok[j] && ((unsigned)y0 <= 3u || (y0 == -1 && ly > 0.f))) {
                const float pxt = tpx[j], m = tpm[j];
                const float fx = floorf(pxt);
                const int x0 = (int)fx - gcol;
                const float lx = pxt - fx;
                const float wy[2] = {(1.f - ly) * m, ly * m}, wx[2] = {1.f - lx, lx};
                const int sy_l = s_row - 4, sx_l = s_col - 4;                 // the source itself, group-local
                // A corner (qy, qx) lands in the group and within reach iff qy in [0, 4), qx in [0, 8), |qy - sy_l| <= 3, |qx - sx_l| <= 3.
                // One unsigned key per axis and corner row / column — (d + 3) | (out-of-range bits of q moved above 6) — and one
                // v_max + compare per corner replace seven compares and six scalar ANDs (the hit test was half of a pass's instructions).
                unsigned ky[2], kx[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int qy = y0 + i, qx = x0 + i;
                    ky[i] = (unsigned)(qy - sy_l + 3) | ((unsigned)(qy & ~3) << 1);
                    kx[i] = (unsigned)(qx - sx_l + 3) | (unsigned)(qx & ~7);
                }
                const int tbase = (y0 * 8 + x0) * DXB_TP + lane * 2;
#pragma unroll
                for (int cnr = 0; cnr < 4; ++cnr) {
                    const float w = wy[cnr >> 1] * wx[cnr & 1];
                    const bool hit = max(ky[cnr >> 1], kx[cnr & 1]) <= 6u && w > 0.f;
                    if (hit) {
                        *reinterpret_cast<bf16_t*>(T + tbase + ((cnr >> 1) * 8 + (cnr & 1)) * DXB_TP) = f2bf(w);
                        any = true;
                    }
                }
            }
            const uint64_t hits = __builtin_amdgcn_ballot_w64(any);
            __builtin_amdgcn_wave_barrier();
            // ---- G^T[co][q] += dY^T[co][window row] * T^T   (one MFMA pair per window row that has a hit) ----
            // (the four row slices of T are read in ONE batch — rows without a hit are zeros and are skipped by the MFMAs only: a
            // read per hit row cost one LDS round trip each, up to four per pass)
            u32x4v tb[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) tb[rr] = *reinterpret_cast<const u32x4v*>(T + nl * DXB_TP + (rr * 16 + hh * 8) * 2);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                if (!(hits & (0xFFFFull << (16 * rr)))) continue;
                const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, tb[rr]);
                st[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[4 * j + rr][0], bf, st[0], 0, 0, 0);
                st[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[4 * j + rr][1], bf, st[1], 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- dx^T[ci][q] += W_k[.][ci]^T G^T[.][q] ----
        DXB_STAMP(3 + 3 * tap);
        const unsigned char* wb = Ws + (tap & 1) * WSB;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4v sb;
#pragma unroll
            for (int d = 0; d < 4; ++d) sb[d] = pk_bf16(st[s >> 1][8 * (s & 1) + 2 * d], st[s >> 1][8 * (s & 1) + 2 * d + 1]);
            const bf16x8_t sf = __builtin_bit_cast(bf16x8_t, sb);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const u32x4v wv = *reinterpret_cast<const u32x4v*>(wb + (((s * 2 + hh) * 32 * NCB) + cb * 32 + nl) * 16);
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wv), sf, acc[cb], 0, 0, 0);
            }
        }
        DXB_STAMP(4 + 3 * tap);
        wstore((tap + 1) & 1);
        __syncthreads();
        wload(tap < 7 ? tap + 2 : 8);
    }
    DXB_STAMP(29);

    // ---- epilogue: + dx_far (lazy protocol of cn_dcn_bwd_dx: added and restored to zero only when a dom kernel flagged far
    //      samples), bf16 through the wave's T tile, 16-byte coalesced stores ----
    const bool use_far = g.far != nullptr && (g.far_flag == nullptr || *g.far_flag != 0);
    if (use_far && live) {
        float* fp = g.far + (img + (int64_t)gy * g.W + gx) * g.dx_ld + 32 * NCB * blockIdx.z;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4* f4 = reinterpret_cast<float4*>(fp + 32 * cb + 8 * q + 4 * hh);
                const float4 v = *f4;
                acc[cb][4 * q] += v.x; acc[cb][4 * q + 1] += v.y; acc[cb][4 * q + 2] += v.z; acc[cb][4 * q + 3] += v.w;
                if (g.far_flag) *f4 = make_float4(0.f, 0.f, 0.f, 0.f);
            }
    }
    {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint2 o;
                o.x = pk_bf16(acc[cb][4 * q], acc[cb][4 * q + 1]); o.y = pk_bf16(acc[cb][4 * q + 2], acc[cb][4 * q + 3]);
                *reinterpret_cast<uint2*>(T + nl * DXB_TP + (32 * cb + 8 * q + 4 * hh) * 2) = o;
            }
        __builtin_amdgcn_wave_barrier();
        constexpr int CPP = 4 * NCB;
#pragma unroll
        for (int i = 0; i < 32 * CPP / 64; ++i) {
            const int idx = lane + 64 * i, p = idx / CPP, ch = idx % CPP;
            const int oy = ty0 + grow + (p >> 3), ox = tx0 + gcol + (p & 7);
            const u32x4v o = *reinterpret_cast<const u32x4v*>(T + p * DXB_TP + ch * 16);
            if (oy < g.H && ox < g.W) *reinterpret_cast<u32x4v*>(g.dx + (img + (int64_t)oy * g.W + ox) * g.dx_ld + 32 * NCB * blockIdx.z + ch * 8) = o;
        }
    }
    DXB_STAMP(31);
}

bool dcn_dx_bm_shape_ok(int Ci, int dy_ld, int om_ld) {
    static const bool disabled = getenv("CN_DISABLE_DCN_BM") != nullptr || getenv("CN_DISABLE_DCN_DX_BM") != nullptr;
    // Ci = 128 / 256 (dY still 64 channels: the 128->64 and 256->64 layers): one workgroup column per 64-channel block of dx, each
    // rebuilding the adjoint blend.  Against the hit-list kernel that is level with zero offsets (128->64 @64^2: 184 us there) and
    // 1.7x ahead with N(0, 0.5 px) offsets (378 us there) — the regime a network is in once conv_offset_mask has moved.
    static const bool no_wide = getenv("CN_DCN_DX_BM_NARROW") != nullptr;
    return !disabled && dy_ld == 64 && (Ci == 64 || Ci == 32 || (!no_wide && (Ci == 128 || Ci == 256))) && om_ld == 32;
}

// returns false when the shape is not handled here (caller falls back to the adjoint-gather kernel)
bool dcn_dx_bm_launch(const void* dy, const void* wpd0, const float* om, float* far, int* far_flag, void* dx, int N, int H, int W, int Ci,
                      int dy_ld, int om_ld, hipStream_t st) {
    if (!dcn_dx_bm_shape_ok(Ci, dy_ld, om_ld) || N > 65535) return false;
    if (((uintptr_t)dy | (uintptr_t)wpd0 | (uintptr_t)dx | (uintptr_t)om | (uintptr_t)far) & 15) return false;
    DxBmGeom g;
    g.dy = (const bf16_t*)dy; g.om = om; g.wp = (const bf16_t*)wpd0; g.far = far; g.far_flag = far_flag; g.dx = (bf16_t*)dx;
    g.N = N; g.H = H; g.W = W; g.dx_ld = Ci; g.ktot = 9 * 64;
    const dim3 grid(((H + BM_TH - 1) / BM_TH) * ((W + BM_TW - 1) / BM_TW), N, Ci > 64 ? Ci / 64 : 1);
    if (Ci >= 64) {
        const size_t smem = DXB_TBYTES + 4 * 32 * DXB_TP + 2 * (4 * 2 * 64 * 16);
        (void)hipFuncSetAttribute((const void*)dcn_dx_bm_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(dcn_dx_bm_kernel<2>, grid, dim3(256), smem, st, g);
    } else {
        const size_t smem = DXB_TBYTES + 4 * 32 * DXB_TP + 2 * (4 * 2 * 32 * 16);
        (void)hipFuncSetAttribute((const void*)dcn_dx_bm_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(dcn_dx_bm_kernel<1>, grid, dim3(256), smem, st, g);
    }
    return true;
}


// ================================================================================================ weight gradient
// dW_k[co][ci] = sum_p dY[p][co] * S_k[p][ci],  S_k = the sampled operand of the forward pass (never stored: 1.2 GB per layer).
// The sampler is the forward kernel's (blend matrix x window on the matrix cores) with the operands swapped, S[p][ci] = Bm[p][.] X[.][ci],
// so that S leaves the matrix pipe with lane = channel, registers = pixels — the B operand of dW[co][ci] += dY^T[co][p] S[p][ci]
// (A = transposed fragments of the dY tile, pixel order permuted to the accumulator layout).  What makes a fused weight gradient
// hard is the accumulator: 9 taps x 64 x 64 fp32 = 144 registers per lane in the gather kernel (dcn_wgrad_kernel<64,64,9>).  Here
// a workgroup is NINE waves and wave t owns tap t: 64 accumulator registers, every wave walks the tile's four pixel groups with
// its own tap's geometry.  The window fragments are read from the halo image per touched row (a wave touches 4-8 of the 12 rows),
// which keeps a wave under 170 registers (three waves per SIMD).  Persistent workgroups (one per CU) accumulate over their run of
// tiles and flush once with fp32 atomics.
struct WgBmGeom {
    const bf16_t* x; const bf16_t* dy; const float* om; float* dwp;
    int N, H, W, Ci, Co, ktot, tiles_h, tiles_w, tiles_per_block;      // blockIdx.y = 64-channel block of x, blockIdx.z = of dY
};

#define WGB_NT 576
#define WGB_XB (BM_WR * BM_WC * BM_PIXB)       // 49 152 B: x halo image
#define WGB_YB (BM_TH * BM_TW * BM_PIXB)       // 16 384 B: dY tile
#define WGB_RB (BM_TH * BM_TW * 128)           // 16 384 B: raw offsets / mask logits of the tile (32 floats per pixel)
#define WGB_OT (4 * 32 * 29 * 4)               // 14 848 B: geometry table
__device__ uint4 wgb_zero_page[8];             // 128 zero bytes: DMA source of everything outside the image

// PIPE: the images of tile t+1 (x halo, dY tile, raw offsets: 80 KB) travel global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KB
// per wave instruction, no staging registers) into a second set of buffers WHILE tile t is computed.  The workgroup is alone on its
// CU (nine waves, 168 registers) and used to sit through every tile's loads: stage (issue 10 loads per thread, wait, store to LDS,
// barrier) and compute alternated, nothing overlapped.  The half swizzle of the images is applied to the DMA SOURCE address (lane l
// of an instruction fills physical 16-byte piece l & 7 of pixel l >> 3 and fetches the logical piece that lives there); pixels
// outside the image fetch a zero page.  The offsets arrive raw and are turned into the geometry table LDS -> LDS at the top of
// their tile.  LDS: 2 x (48 + 16) KB images + 16 KB raw offsets + 14.5 KB table = 158.9 KB of the CU's 160.
// NW = 9: wave t owns tap t.  NW = 8: the same for taps 0..7 and tap 8 is dealt out by pixel group to waves 0..3 (a second accumulator
// set; two waves per SIMD allow 256 registers).  Nine waves put three on one SIMD and two on the others, and the tile waits for the
// SIMD with three tap-works per group; eight waves leave 2 x 4 + 1 = 9 group-works on every SIMD instead of 12 on the busiest.
template <bool PIPE, int NW>
__global__ __launch_bounds__(NW * 64) void dcn_wgrad_bm_kernel(const WgBmGeom g) {
    constexpr int NT = NW * 64;
    static_assert(NW == 9 || (NW == 8 && PIPE), "eight waves: pipelined variant only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XOFF1 = PIPE ? WGB_XB : 0, YOFF0 = PIPE ? 2 * WGB_XB : WGB_XB, YOFF1 = PIPE ? YOFF0 + WGB_YB : YOFF0;
    constexpr int ROFF = YOFF1 + WGB_YB, OOFF = PIPE ? ROFF + WGB_RB : ROFF;
    unsigned char* Xw = smem;                                         // x halo image [16][24] x 128 B (PIPE: of the tile being computed)
    unsigned char* Yt = smem + YOFF0;                                 // dY tile [8][16] x 128 B (same half swizzle)
    float* const OmT = reinterpret_cast<float*>(smem + OOFF);         // [4 groups][32 px][29]
    u32x4v* const Lut = reinterpret_cast<u32x4v*>(smem + OOFF + WGB_OT);   // [23]
    const int tid = threadIdx.x, lane = tid & 63, tap = tid >> 6;
    if (tid < 23) {
        const int c = tid - 8;
        u32x4v sel;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int sft = c - 2 * d;
            sel[d] = sft == 0 ? 0x03020100u : (sft == 1 ? 0x01000c0cu : (sft == -1 ? 0x0c0c0302u : 0x0c0c0c0cu));
        }
        Lut[tid] = sel;
    }
    const int nl = lane & 31, hh = lane >> 5;
    const int r16 = lane & 15, g16 = lane >> 4;
    typedef __attribute__((address_space(3))) s16x4_t_* lds_ptr;

    f32x16_t acc[2][2], acc8[2][2];                                   // [co block][ci block] of dW_tap (acc8: this wave's share of tap 8, NW = 8)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[a][b][r] = 0.f; acc8[a][b][r] = 0.f; }

    const int64_t ntiles = (int64_t)g.N * g.tiles_h * g.tiles_w;
    const int64_t t_beg = (int64_t)blockIdx.x * g.tiles_per_block;
    const int64_t t_end = t_beg + g.tiles_per_block < ntiles ? t_beg + g.tiles_per_block : ntiles;

    // ---- PIPE: one tile's images by LDS-DMA: 48 + 16 + 16 one-KB pieces, piece J = 9 p + wave ----
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int wave_s = __builtin_amdgcn_readfirstlane(tap);
    auto issue_tile = [&](int64_t tile_, int buf_) {
        const int n_ = (int)(tile_ / (g.tiles_h * g.tiles_w));
        const int rt_ = (int)(tile_ - (int64_t)n_ * g.tiles_h * g.tiles_w);
        const int ty_ = (rt_ / g.tiles_w) * BM_TH, tx_ = (rt_ % g.tiles_w) * BM_TW;
        const int64_t img_ = (int64_t)n_ * g.H * g.W;
        const char* const Xb = reinterpret_cast<const char*>(g.x + img_ * g.Ci + 64 * blockIdx.y);
        const char* const Yb = reinterpret_cast<const char*>(g.dy + img_ * g.Co + 64 * blockIdx.z);
        const char* const Ob = reinterpret_cast<const char*>(g.om + img_ * 32);
        int ln = lane;
        asm volatile("" : "+v"(ln));                // per-lane addressing recomputed per tile: hoisted out of the tile loop it pins registers
        const char* const zp = reinterpret_cast<const char*>(wgb_zero_page) + (ln & 7) * 16;
        const int pl = ln >> 3, q = ln & 7;
#pragma unroll 1
        for (int p = 0; p < (80 + NW - 1) / NW; ++p) {
            const int J = p * NW + wave_s;
            if (J >= 80) continue;
            const char* src;
            unsigned dst;
            if (J < 48) {                                   // x halo: pixel J*8 + pl of the [16][24] window
                const int pix = J * 8 + pl, r = pix / BM_WC, c = pix - r * BM_WC;
                const int hy = ty_ - BM_MG + r, hx = tx_ - BM_MG + c;
                const bool ok = (unsigned)hy < (unsigned)g.H && (unsigned)hx < (unsigned)g.W;
                const int ql = q ^ (((c >> 1) & 1) << 2);   // the logical piece stored at physical piece q (bm_lds_ofs)
                src = ok ? Xb + (((int64_t)hy * g.W + hx) * g.Ci + ql * 8) * 2 : zp;
                dst = lds_base + (unsigned)((buf_ ? XOFF1 : 0) + J * 1024);
            } else {
                const int I = J < 64 ? J - 48 : J - 64, pix = I * 8 + pl, pr = pix >> 4, pc = pix & 15;
                const int py_ = ty_ + pr, px_ = tx_ + pc;
                const bool ok = py_ < g.H && px_ < g.W;
                if (J < 64) {                               // dY tile
                    const int ql = q ^ (((pc >> 1) & 1) << 2);
                    src = ok ? Yb + (((int64_t)py_ * g.W + px_) * g.Co + ql * 8) * 2 : zp;
                    dst = lds_base + (unsigned)((buf_ ? YOFF1 : YOFF0) + I * 1024);
                } else {                                    // raw offsets / mask logits (32 floats per pixel, unswizzled)
                    src = ok ? Ob + (((int64_t)py_ * g.W + px_) * 32 + q * 4) * 4 : zp;
                    dst = lds_base + (unsigned)(ROFF + I * 1024);
                }
            }
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };
    int buf = 0;
    if constexpr (PIPE) {
        if (t_beg < t_end) issue_tile(t_beg, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                           // also publishes Lut
    }

#pragma unroll 1
    for (int64_t tile = t_beg; tile < t_end; ++tile) {
        const int n = (int)(tile / (g.tiles_h * g.tiles_w));
        const int rt = (int)(tile - (int64_t)n * g.tiles_h * g.tiles_w);
        const int ty0 = (rt / g.tiles_w) * BM_TH, tx0 = (rt % g.tiles_w) * BM_TW;
        const int64_t img = (int64_t)n * g.H * g.W;
        const bf16_t* __restrict__ X = g.x + img * g.Ci + 64 * blockIdx.y;      // this workgroup's 64-channel block of x ...
        const bf16_t* __restrict__ DY = g.dy + img * g.Co + 64 * blockIdx.z;    // ... and of dY: dW[co block][ci block]
        const float* __restrict__ OM = g.om + img * 32;
        if constexpr (PIPE) {
            // the images of this tile are in buffer `buf` (DMA retired behind the previous tile's MFMAs); raw offsets -> table
            Xw = smem + (buf ? XOFF1 : 0);
            Yt = smem + (buf ? YOFF1 : YOFF0);
#pragma unroll
            for (int i = 0; i < (1024 + NT - 1) / NT; ++i) {
                const int v = tid + i * NT;
                const int pix = v >> 3, q = v & 7;
                if (v >= 1024 || q == 7) continue;
                const int pr = pix >> 4, pc = pix & 15;
                const float4 o4 = *reinterpret_cast<const float4*>(smem + ROFF + pix * 128 + q * 16);
                const bool inimg = ty0 + pr < g.H && tx0 + pc < g.W;
                float* d = OmT + ((((pr >> 2) * 2 + (pc >> 3)) * 32 + (pr & 3) * 8 + (pc & 7)) * 29) + q * 4;
                const float v4[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = q * 4 + j, k = e >> 1, k3 = (k * 11) >> 5;
                    float val = v4[j] + (float)((e & 1) ? (pc & 7) + 3 + (k - 3 * k3) : (pr & 3) + 3 + k3);
                    if (q >= 4 && e >= 18) val = inimg ? __builtin_amdgcn_rcpf(1.f + __expf(-v4[j])) : 0.f;
                    if (e < 27) d[j] = val;
                }
            }
            __syncthreads();                       // table complete; the raw offsets may be overwritten
            if (tile + 1 < t_end) issue_tile(tile + 1, buf ^ 1);
        } else if constexpr (NW == 9) {
        __syncthreads();                           // everybody is done with the previous tile's images
        // ---- staging: x halo (3072 vectors), dY tile (1024), offsets / masks of the 128 pixels (896) ----
        {
            constexpr int NH = (BM_WR * BM_WC * 8 + WGB_NT - 1) / WGB_NT;      // 6
            uint4 hv[NH];
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                const int v = tid + i * WGB_NT;
                const int pix = v >> 3, q = v & 7;
                const int hy = ty0 - BM_MG + pix / BM_WC, hx = tx0 - BM_MG + pix % BM_WC;
                const bool ok = v < BM_WR * BM_WC * 8 && (unsigned)hy < (unsigned)g.H && (unsigned)hx < (unsigned)g.W;
                hv[i] = ldg16_masked(X, (((int64_t)hy * g.W + hx) * g.Ci + q * 8) * 2, ok);
            }
            uint4 yv[2];
            float4 ov[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int v = tid + i * WGB_NT;
                const int pix = v >> 3, q = v & 7;
                const int py_ = ty0 + (pix >> 4), px_ = tx0 + (pix & 15);
                const bool ok = v < 1024 && py_ < g.H && px_ < g.W;
                yv[i] = ldg16_masked(DY, (((int64_t)py_ * g.W + px_) * g.Co + q * 8) * 2, ok);
                ov[i] = *reinterpret_cast<const float4*>(OM + ((int64_t)(ok ? py_ : 0) * g.W + (ok ? px_ : 0)) * 32 + (q < 7 ? q : 0) * 4);
            }
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                const int v = tid + i * WGB_NT;
                const int pix = v >> 3, q = v & 7;
                if (v < BM_WR * BM_WC * 8) st16(Xw + bm_lds_ofs(pix / BM_WC, pix % BM_WC, q * 8), hv[i]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int v = tid + i * WGB_NT;
                const int pix = v >> 3, q = v & 7;
                if (v >= 1024) continue;
                const int pr = pix >> 4, pc = pix & 15;
                st16(Yt + pix * BM_PIXB + ((((q >> 2) ^ (pc >> 1)) & 1) << 6) + (q & 3) * 16, yv[i]);
                // table of the pixel's group: positions relative to the group's window origin, sigmoid(mask) (see the forward kernel)
                const bool inimg = ty0 + pr < g.H && tx0 + pc < g.W;
                float* d = OmT + ((((pr >> 2) * 2 + (pc >> 3)) * 32 + (pr & 3) * 8 + (pc & 7)) * 29) + q * 4;
                const float v4[4] = {ov[i].x, ov[i].y, ov[i].z, ov[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = q * 4 + j, k = e >> 1, k3 = (k * 11) >> 5;
                    float val = v4[j] + (float)((e & 1) ? (pc & 7) + 3 + (k - 3 * k3) : (pr & 3) + 3 + k3);
                    if (q >= 4 && e >= 18) val = inimg ? __builtin_amdgcn_rcpf(1.f + __expf(-v4[j])) : 0.f;
                    if (e < 27) d[j] = val;
                }
            }
        }
        __syncthreads();
        }   // !PIPE

        auto do_group = [&](const int grp, const int tapx, f32x16_t (&accx)[2][2]) {
            const int grow = (grp >> 1) * 4, gcol = (grp & 1) * 8;
            // ---- geometry of (own pixel, this wave's tap) ----
            const float* orow = OmT + (grp * 32 + nl) * 29;
            const float pyr = orow[2 * tapx], pxr = orow[2 * tapx + 1], m = orow[18 + tapx];
            const float fy = floorf(pyr), fx = floorf(pxr);
            const int wr = (int)fy, wc = (int)fx;
            const float ly = pyr - fy, lx = pxr - fx;
            const float wa = (1.f - ly) * m, wbt = ly * m;
            const bool inwin = (unsigned)wr <= (unsigned)(BM_GR - 2) && (unsigned)wc <= 14u;
            const bool far = !inwin && m != 0.f;
            const uint32_t P0 = inwin ? pk_bf16(wa * (1.f - lx), wa * lx) : 0u, P1 = inwin ? pk_bf16(wbt * (1.f - lx), wbt * lx) : 0u;
            const int wr_top = (P0 != 0u) ? wr : -1, wr_bot = (P1 != 0u) ? wr + 1 : -1;
            const u32x4v sel = Lut[min(max(wc - 8 * hh + 8, 0), 22)];
            uint32_t V0[4], V1[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                V0[d] = __builtin_amdgcn_perm(P0, P0, sel[d]);
                V1[d] = __builtin_amdgcn_perm(P1, P1, sel[d]);
            }
            uint32_t rows = ((P0 != 0u) ? (1u << (wr & 15)) : 0u) | ((P1 != 0u) ? (2u << (wr & 15)) : 0u);
            rows |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rows, 0xB1, 0xF, 0xF, true);
            rows |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rows, 0x4E, 0xF, 0xF, true);
            rows |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rows, 0x141, 0xF, 0xF, true);
            rows |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rows, 0x140, 0xF, 0xF, true);
            uint32_t rowmask = (uint32_t)__builtin_amdgcn_readlane((int)rows, 0) | (uint32_t)__builtin_amdgcn_readlane((int)rows, 16) |
                               (uint32_t)__builtin_amdgcn_readlane((int)rows, 32) | (uint32_t)__builtin_amdgcn_readlane((int)rows, 48);

            // ---- S[p][ci] = Bm[p][window row] X[window row][ci] over the touched rows (fragments straight from the halo image) ----
            // (round 6: the first touched row STARTS the accumulators — C = 0 is an inline constant of the MFMA — instead of 32 v_mov per
            // unit, a seventh of the unit's VALU instructions; a unit nobody samples in skips its MFMAs altogether)
            f32x16_t st[2];
            const bool any_row = rowmask != 0u;
            const int wc0 = gcol + 8 * (g16 >> 1) + (r16 >> 2);
            const unsigned char* const b0 = Xw + bm_lds_ofs(grow, wc0, 16 * (g16 & 1) + 4 * (r16 & 3));
            const unsigned char* const b1 = Xw + bm_lds_ofs(grow, wc0, 32 + 16 * (g16 & 1) + 4 * (r16 & 3));
            auto row_mfma = [&](const int r, const f32x16_t& c0, const f32x16_t& c1) {
                const bool t0 = wr_top == r, t1 = wr_bot == r;
                u32x4v b;
#pragma unroll
                for (int d = 0; d < 4; ++d) b[d] = t0 ? V0[d] : (t1 ? V1[d] : 0u);
                const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, b);
                const int ro_ = r * (BM_WC * BM_PIXB);
                const s16x4_t_ l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(b0 + ro_));
                const s16x4_t_ h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(b0 + ro_ + 4 * BM_PIXB));
                const s16x4_t_ l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(b1 + ro_));
                const s16x4_t_ h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(b1 + ro_ + 4 * BM_PIXB));
                const s16x8_t_ x0 = {l0[0], l0[1], l0[2], l0[3], h0[0], h0[1], h0[2], h0[3]};
                const s16x8_t_ x1 = {l1[0], l1[1], l1[2], l1[3], h1[0], h1[1], h1[2], h1[3]};
                st[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, __builtin_bit_cast(bf16x8_t, x0), c0, 0, 0, 0);
                st[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, __builtin_bit_cast(bf16x8_t, x1), c1, 0, 0, 0);
            };
            if (any_row) {
                const int r = __builtin_ctz(rowmask);
                rowmask &= rowmask - 1;
                f32x16_t z;
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = 0.f;
                row_mfma(r, z, z);
            }
#pragma unroll 1
            while (rowmask) {
                const int r = __builtin_ctz(rowmask);
                rowmask &= rowmask - 1;
                row_mfma(r, st[0], st[1]);
            }
            // ---- samples that leave the window (rare): pixel by pixel, every lane blends ITS channel of S from global memory (exact fp32
            //      weights) and applies the rank-1 update dW[.][ci] += dY[p][.] * S[p][ci] to its accumulator columns directly ----
            uint32_t farmask = (uint32_t)__builtin_amdgcn_ballot_w64(far);      // lanes 0..31 = the group's pixels (32..63 repeat them)
#pragma unroll 1
            while (farmask) {
                const int p = __builtin_ctz(farmask);
                farmask &= farmask - 1;
                const float f_ly = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ly), p));
                const float f_lx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lx), p));
                const float f_m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), p));
                const int h0_ = __builtin_amdgcn_readlane(wr, p) + (ty0 + grow - BM_MG), w0_ = __builtin_amdgcn_readlane(wc, p) + (tx0 + gcol - BM_MG);
                float sv0 = 0.f, sv1 = 0.f;
#pragma unroll
                for (int cnr = 0; cnr < 4; ++cnr) {
                    const int cy = h0_ + (cnr >> 1), cx = w0_ + (cnr & 1);
                    const bool ok = (unsigned)cy < (unsigned)g.H && (unsigned)cx < (unsigned)g.W;          // wave-uniform
                    const float w = ok ? ((cnr >> 1) ? f_ly : 1.f - f_ly) * ((cnr & 1) ? f_lx : 1.f - f_lx) * f_m : 0.f;
                    const bf16_t* xp = X + ((int64_t)(ok ? cy : 0) * g.W + (ok ? cx : 0)) * g.Ci + nl;
                    sv0 += w * bf2f(xp[0]);
                    sv1 += w * bf2f(xp[32]);
                }
                const int pc = gcol + (p & 7);
                const unsigned char* yp = Yt + ((grow + (p >> 3)) * 16 + pc) * BM_PIXB;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int co = 32 * cb + 8 * (v >> 2) + 4 * hh + (v & 3);
                        const float dyv = bf2f(*reinterpret_cast<const bf16_t*>(yp + ((((co >> 5) ^ (pc >> 1)) & 1) << 6) + (co & 31) * 2));
                        accx[cb][0][v] += dyv * sv0;
                        accx[cb][1][v] += dyv * sv1;
                    }
            }
            // ---- dW[co][ci] += dY^T[co][p] S[p][ci]   (K = the group's 32 pixels in the accumulator's row order) ----
            if (any_row)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4v sb[2];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int d = 0; d < 4; ++d) sb[mb][d] = pk_bf16(st[mb][8 * ks + 2 * d], st[mb][8 * ks + 2 * d + 1]);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    // A fragment: lane = co (16 (g16 & 1) + r16 of block cb), K element e <-> group pixel 16 ks + 8 (e >> 2) + 4 (g16 >> 1) + (e & 3)
                    s16x4_t_ part[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int pgrp = 16 * ks + 8 * u + 4 * (g16 >> 1) + (r16 >> 2);          // the pixel whose 8-byte chunk this lane addresses
                        const int pr = grow + (pgrp >> 3), pc = gcol + (pgrp & 7);
                        const int ch = 32 * cb + 16 * (g16 & 1) + 4 * (r16 & 3);
                        part[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (lds_ptr)(Yt + (pr * 16 + pc) * BM_PIXB + ((((ch >> 5) ^ (pc >> 1)) & 1) << 6) + (ch & 31) * 2));
                    }
                    const s16x8_t_ ya = {part[0][0], part[0][1], part[0][2], part[0][3], part[1][0], part[1][1], part[1][2], part[1][3]};
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
                        accx[cb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ya), __builtin_bit_cast(bf16x8_t, sb[mb]),
                                                                               accx[cb][mb], 0, 0, 0);
                }
            }
        };
#pragma unroll 1
        for (int grp = 0; grp < 4; ++grp) do_group(grp, tap, acc);
        if constexpr (NW == 8) {
            if (tap < 4) do_group(tap, 8, acc8);          // tap 8: pixel group w of the tile belongs to wave w
        }
        if constexpr (PIPE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of the next tile have landed ...
            __syncthreads();                                        // ... everybody's have, and everybody is done with this tile's images
            buf ^= 1;
        }
    }
    // ---- flush: acc[cb][mb] register v of lane = dW_tap[co = 32 cb + 8 (v >> 2) + 4 hh + (v & 3)][ci = 32 mb + nl] ----
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int co = 64 * blockIdx.z + 32 * cb + 8 * (v >> 2) + 4 * hh + (v & 3), ci = 64 * blockIdx.y + 32 * mb + nl;
                atomicAdd(g.dwp + (int64_t)co * g.ktot + tap * g.Ci + ci, acc[cb][mb][v]);
                if (NW == 8 && tap < 4) atomicAdd(g.dwp + (int64_t)co * g.ktot + 8 * g.Ci + ci, acc8[cb][mb][v]);
            }
}

bool dcn_wgrad_bm_shape_ok(int Ci, int x_ld, int Co, int dy_ld, int om_ld) {
    static const bool disabled = getenv("CN_DISABLE_DCN_BM") != nullptr || getenv("CN_DISABLE_DCN_WGRAD_BM") != nullptr;
    // every (x block, dY block) pair re-samples: measured ahead of dcn_wgrad_kernel up to 8 pairs (128->64@64^2 319 -> 183 us, 128->128
    // 395 -> 296, 256->128@32^2 229 -> 175), level with it beyond (256->256 310 vs 287 zero offsets, 313 vs 343 N(0,0.5))
    static const int max_pairs = [] { const char* e = getenv("CN_DCN_WGRAD_BM_PAIRS"); return e ? atoi(e) : 8; }();
    return !disabled && Ci % 64 == 0 && x_ld == Ci && Co % 64 == 0 && dy_ld == Co && om_ld == 32 && (Ci / 64) * (Co / 64) <= max_pairs;
}

// returns false when the shape is not handled here (caller falls back to dcn_wgrad_kernel)
bool dcn_wgrad_bm_launch(const void* x, const float* om, const void* dy, float* dwp, int N, int H, int W, int Ci, int x_ld, int Co, int dy_ld,
                         int om_ld, int target_blocks, hipStream_t st) {
    if (!dcn_wgrad_bm_shape_ok(Ci, x_ld, Co, dy_ld, om_ld)) return false;
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)om) & 15) return false;
    WgBmGeom g;
    g.x = (const bf16_t*)x; g.dy = (const bf16_t*)dy; g.om = om; g.dwp = dwp;
    g.N = N; g.H = H; g.W = W; g.Ci = Ci; g.Co = Co; g.ktot = 9 * Ci;
    g.tiles_h = (H + BM_TH - 1) / BM_TH; g.tiles_w = (W + BM_TW - 1) / BM_TW;
    const int64_t ntiles = (int64_t)N * g.tiles_h * g.tiles_w;
    const int par = (Ci / 64) * (Co / 64);                             // channel-block pairs: independent workgroups
    int64_t want = (target_blocks < 256 ? target_blocks : 256) / par;  // one persistent workgroup per CU at most
    if (want < 8) want = 8;
    if (want > ntiles) want = ntiles;
    if (want < 1) want = 1;
    g.tiles_per_block = (int)((ntiles + want - 1) / want);
    const int gx = (int)((ntiles + g.tiles_per_block - 1) / g.tiles_per_block);
    static const bool no_pipe = getenv("CN_DCN_WGRAD_NO_PIPE") != nullptr;      // A/B: stage through registers, nothing overlapped
    // A/B: eight waves, tap 8 dealt out by pixel group (9 group-works per SIMD instead of 12 on the busiest): 314 vs 308 us on
    // 64->64 @128^2, step 42.00 vs 42.01 ms — the kernel is bound by the latency of a wave's own chain, not by the busiest SIMD
    static const bool nine = getenv("CN_DCN_WGRAD_WAVES8") == nullptr;
    if (no_pipe) {
        const size_t smem = (size_t)WGB_XB + WGB_YB + WGB_OT + 512;
        (void)hipFuncSetAttribute((const void*)dcn_wgrad_bm_kernel<false, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((dcn_wgrad_bm_kernel<false, 9>), dim3(gx, Ci / 64, Co / 64), dim3(576), smem, st, g);
    } else if (nine) {
        const size_t smem = (size_t)2 * WGB_XB + 2 * WGB_YB + WGB_RB + WGB_OT + 512;
        (void)hipFuncSetAttribute((const void*)dcn_wgrad_bm_kernel<true, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((dcn_wgrad_bm_kernel<true, 9>), dim3(gx, Ci / 64, Co / 64), dim3(576), smem, st, g);
    } else {
        const size_t smem = (size_t)2 * WGB_XB + 2 * WGB_YB + WGB_RB + WGB_OT + 512;
        (void)hipFuncSetAttribute((const void*)dcn_wgrad_bm_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((dcn_wgrad_bm_kernel<true, 8>), dim3(gx, Ci / 64, Co / 64), dim3(512), smem, st, g);
    }
    return true;
}
