// DCNv2 for 64-channel layers, "gather-sample" formulation (round 5; bf16).  SURVEY App. A; pose_dla_dcn.py:441-449.
//
//   S_k[p][ci] = sum_{corners c} w_c(p,k) * m(p,k) * x[corner_c(p,k)][ci]           (bilinear sample x sigmoid(mask))
//   y[p][co]   = bias[co] + sum_k sum_ci W_k[co][ci] * S_k[p][ci]
//
// The blend-matrix kernels (dcn_bm.hip) run the blend as a second MFMA; they sit at 0.15-0.25 MFMA-pipe busy with 14-22 VALU
// instructions per MFMA (profiles/r04_pmc_sq.txt): selects that build blend-matrix rows, zero fills, a barrier per tap for the
// weight slices.  Here the blend is TWO v_dot2_f32_bf16 per channel:
//   * the workgroup's x halo sits in LDS as a PAIR IMAGE: dword (r, c, ci) = {x[r][c][ci], x[r][c+1][ci]} — the two horizontal
//     neighbours of a bilinear footprint in one register, so   S = dot2(pair[top], {w00 m, w01 m}) + dot2(pair[bot], {w10 m, w11 m})
//     (the weights enter as bf16, like the blend-matrix kernels' do; products and sums are fp32);
//   * lane = (pixel, channel half): the dot2 results of a lane, packed to bf16, ARE the B operand of the contraction MFMA
//     (lane = pixel column, 8 consecutive channels per k-step) — nothing is transposed, nothing goes back through LDS;
//   * the geometry of a (pixel, tap) — two packed weight pairs, the LDS addresses of its two footprint rows — is a 16-byte RECORD
//     built once per tile from the fp32 offsets / mask logits (floor, fractions, sigmoid, image-border zeroing happen there, once,
//     instead of once per lane half and tap), so a tap costs a lane one 16-byte LDS read of geometry;
//   * the weights are STATIONARY IN REGISTERS: the two waves that share a 4x8 pixel group split the 18 (tap, channel half) units
//     9 : 9, each holding its 9 x 16 registers of W fragments for the whole launch (persistent workgroups), and meet once per
//     tile through LDS.  No weight traffic in the loop, no barrier per tap: a tile is four barriers.
// LDS per workgroup (256 threads, tile 8x8 pixels): pair image 15 rows x 16 slots x 272 B + records 9 KB + offset staging 6.75 KB
// + 0.5 KB = 80 KB -> two workgroups per CU that run out of phase (one loads its tile while the other computes).
// Samples whose footprint leaves the halo (|offset| >= 2 px): the two footprint pixels are copied from global memory into spare
// pixel slots of the workgroup's LDS (the offset staging area is dead by then, and every halo row has one unused slot) and the
// record points there — the hot loop knows nothing about them.  32 slots = 16 such samples per pass over the tile; a tile with
// more runs the unit loop again for the next 16 with every other record pointed at an all-zero pixel (any offset field is
// handled; a field with more than ~3 % far samples pays for it).
#include "conv_common.h"
#include <stdlib.h>

#define GS_T 8                                   // tile edge (pixels)
#define GS_MG 3                                  // halo margin above / left of the tile (4 below / right)
#define GS_ROWS (GS_T + 7)                       // 15 halo rows
#define GS_SLOTS 16                              // pixel slots per halo row (15 pair columns; slot = column ^ 8 on odd rows)
#define GS_PIX 272                               // bytes per pair-image pixel: 64 dwords + 16 (odd multiple of 16: 16 consecutive
                                                 // pixels tile the 64 banks for a ds_read_b128)
#define GS_IMG (GS_ROWS * GS_SLOTS * GS_PIX)     // 65 280
#define GS_NREC (GS_T * GS_T * 9)                // 576 records x 16 B
#define GS_REC_OFS GS_IMG
#define GS_OMS_OFS (GS_REC_OFS + GS_NREC * 16)   // staged offsets / mask logits, 27 floats per pixel (6 912 B); later: 24 far slots
#define GS_BMP_OFS (GS_OMS_OFS + GS_T * GS_T * 27 * 4)       // 18 dwords: bitmap of the records whose footprint left the halo; dword 18: their count
#define GS_ZERO_OFS ((8 * GS_SLOTS + 15) * GS_PIX)           // an all-zero pixel and the bias (64 floats) live in the pixel slots no pair
#define GS_BIAS_OFS ((9 * GS_SLOTS + 7) * GS_PIX)            // column maps to in halo rows 8 and 9 (rows 0 .. 7: far slots 24 .. 31)
#define GS_SMEM 81920
#define GS_FAR_PER_PASS 16

#ifdef GS_PROBE   // development build only (tools/gs_probe.py): cycle stamps of waves 0 / 1 of the first workgroups, first 4 tiles
__device__ unsigned long long gs_ts[64 * 2 * 4 * 16];
#define GS_STAMP(k) do { if (lane == 0 && wave < 2 && blockIdx.x < 64 && j < 4) gs_ts[((blockIdx.x * 2 + wave) * 4 + j) * 16 + (k)] = clock64(); } while (0)
extern "C" int gs_probe_dump(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gs_ts), sizeof(gs_ts)); }
#else
#define GS_STAMP(k) do { } while (0)
#endif

typedef uint32_t u32x4g __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2g __attribute__((ext_vector_type(2)));

struct GsFwdGeom {
    const bf16_t* x; const float* om; const bf16_t* wp; const float* bias; bf16_t* y;
    int N, H, W, relu;
    float* bn_part; int bn_slots;
    int tiles_w, tiles_img, ntiles;
};

__device__ static inline float gs_dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2g, a), __builtin_bit_cast(bf16x2g, b), c, false);
}
// a.lo * b.lo + a.hi * b.hi with the VOP3P form and an inline 0 addend (the builtin selects v_dot2c + a v_mov 0 per result)
__device__ static inline float gs_dot2z(uint32_t a, uint32_t b) {
    float d;
    asm("v_dot2_f32_bf16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// LDS byte offset of halo pixel (row r, pair column c)
__device__ static inline int gs_pix(int r, int c) { return (r * GS_SLOTS + (c ^ ((r & 1) << 3))) * GS_PIX; }
// LDS byte offset of spare pixel slot id (0 .. 31): the dead offset staging area, then the slot no pair column maps to in rows 0 .. 7
__device__ static inline int gs_far_slot(int id) {
    return id < 24 ? GS_OMS_OFS + id * GS_PIX : ((id - 24) * GS_SLOTS + (((id - 24) & 1) ? 7 : 15)) * GS_PIX;
}
#define GS_ZERO_PAIR (((uint32_t)(GS_ZERO_OFS >> 4)) | ((uint32_t)(GS_ZERO_OFS >> 4) << 16))

// tile index of step j of workgroup b: each XCD (workgroups are dealt round-robin to the 8 XCDs) owns a contiguous chunk of tiles and
// walks it with all its workgroups side by side, so that the halos neighbouring tiles share are read through one L2 at one time
__device__ static inline int gs_tile_of(int b, int G, int j, int ntiles) {
    if (G % 8 != 0) { const int t = b + j * G; return t < ntiles ? t : -1; }
    const int xcd = b & 7, slot = b >> 3, gx = G >> 3;
    const int cx = (ntiles + 7) >> 3;
    const int t = xcd * cx + slot + j * gx;
    const int lim = min((xcd + 1) * cx, ntiles);
    return t < lim ? t : -1;
}

// ---- geometry records of a tile: built from the staged offsets by all 256 threads ----
// record (tap, pixel) = { P0 = bf16 pair (w00 m, w01 m), P1 = (w10 m, w11 m), LDS offsets / 16 of the top and bottom footprint pixel
// (16 bits each), corner 00 in image coordinates (int16 pair) }.  Corners outside the image have weight 0.  A footprint outside the
// halo is noted in the bitmap and pointed at the zero pixel until gs_place_far gives it a slot.
__device__ static inline void gs_build_records(unsigned char* smem, int tid, int ty0, int tx0, int H, int W) {
    const float* const oms = reinterpret_cast<const float*>(smem + GS_OMS_OFS);
    u32x4g* const rec = reinterpret_cast<u32x4g*>(smem + GS_REC_OFS);
    uint32_t* const bmp = reinterpret_cast<uint32_t*>(smem + GS_BMP_OFS);
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {
        const int idx = tid + 256 * i;
        if (idx < GS_NREC) {
            const int tap = idx >> 6, p = idx & 63;
            const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;           // tap / 3, tap % 3 for tap < 9
            const int y = ty0 + (p >> 3), x = tx0 + (p & 7);
            const bool live = y < H && x < W;
            const float* o = oms + p * 27;
            // the sampling position is formed like the reference forms it: ONE fp32 add of the integer position and the offset
            const float py = (float)(y - 1 + ky) + o[2 * tap], px = (float)(x - 1 + kx) + o[2 * tap + 1];
            const float m = live ? __builtin_amdgcn_rcpf(1.f + __expf(-o[18 + tap])) : 0.f;
            const float fy = floorf(py), fx = floorf(px);
            const int h0 = (int)fminf(fmaxf(fy, -2.f), 32766.f), w0 = (int)fminf(fmaxf(fx, -2.f), 32766.f);
            const float ly = py - fy, lx = px - fx;
            const float wa = (1.f - ly) * m, wb = ly * m;
            const bool h0ok = (unsigned)h0 < (unsigned)H, h1ok = (unsigned)(h0 + 1) < (unsigned)H;
            const bool w0ok = (unsigned)w0 < (unsigned)W, w1ok = (unsigned)(w0 + 1) < (unsigned)W;
            const uint32_t P0 = pk_bf16((h0ok && w0ok) ? wa * (1.f - lx) : 0.f, (h0ok && w1ok) ? wa * lx : 0.f);
            const uint32_t P1 = pk_bf16((h1ok && w0ok) ? wb * (1.f - lx) : 0.f, (h1ok && w1ok) ? wb * lx : 0.f);
            const int wr = h0 - (ty0 - GS_MG), wc = w0 - (tx0 - GS_MG);
            const bool nz = ((P0 | P1) & 0x7fff7fffu) != 0u;
            const bool inwin = (unsigned)wr <= (unsigned)(GS_ROWS - 2) && (unsigned)wc <= 14u;
            uint32_t ad = GS_ZERO_PAIR;
            if (nz && inwin) ad = (uint32_t)(gs_pix(wr, wc) >> 4) | ((uint32_t)(gs_pix(wr + 1, wc) >> 4) << 16);
            if (nz && !inwin) { atomicOr(bmp + (idx >> 5), 1u << (idx & 31)); atomicAdd(bmp + 18, 1u); }
            u32x4g r;
            r[0] = nz ? P0 : 0u; r[1] = nz ? P1 : 0u; r[2] = ad; r[3] = ((uint32_t)h0 & 0xffffu) | ((uint32_t)w0 << 16);
            rec[idx] = r;
        }
    }
}

// ---- pass k over a tile with far footprints (rare): records of far sample number 16 k .. 16 k + 15 get two spare pixel slots filled
//      from global memory, every other far record — and, from the second pass on, every near record (they were accumulated in pass
//      0) — points at the zero pixel.  Each thread serves its own records. ----
__device__ static __attribute__((noinline)) void gs_place_far(unsigned char* smem, const bf16_t* X, int tid, int k, int H, int W) {
    u32x4g* const rec = reinterpret_cast<u32x4g*>(smem + GS_REC_OFS);
    const uint32_t* const bmp = reinterpret_cast<const uint32_t*>(smem + GS_BMP_OFS);
    const unsigned char* const xb = reinterpret_cast<const unsigned char*>(X);
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {
        const int idx = tid + 256 * i;
        if (idx >= GS_NREC) break;
        const uint32_t word = bmp[idx >> 5];
        uint32_t* const adp = reinterpret_cast<uint32_t*>(rec + idx) + 2;
        if (!((word >> (idx & 31)) & 1u)) { if (k > 0) *adp = GS_ZERO_PAIR; continue; }
        int ord = __builtin_popcount(word & ((1u << (idx & 31)) - 1u));
        for (int w = 0; w < (idx >> 5); ++w) ord += __builtin_popcount(bmp[w]);
        if (ord / GS_FAR_PER_PASS != k) { *adp = GS_ZERO_PAIR; continue; }
        const int s0 = gs_far_slot(2 * (ord % GS_FAR_PER_PASS)), s1 = gs_far_slot(2 * (ord % GS_FAR_PER_PASS) + 1);
        const uint32_t hw = reinterpret_cast<const uint32_t*>(rec + idx)[3];
        const int h0 = (int)(short)(hw & 0xffffu), w0 = (int)hw >> 16;
        const int wc0 = min(max(w0, 0), W - 1), wc1 = min(max(w0 + 1, 0), W - 1);
#pragma unroll 1
        for (int row = 0; row < 2; ++row) {
            const int hc = min(max(h0 + row, 0), H - 1);
            unsigned char* const d = smem + (row ? s1 : s0);
#pragma unroll 1
            for (int q = 0; q < 8; ++q) {
                const u32x4g l = *reinterpret_cast<const u32x4g*>(xb + (uint32_t)((hc * W + wc0) * 128 + q * 16));
                const u32x4g r = *reinterpret_cast<const u32x4g*>(xb + (uint32_t)((hc * W + wc1) * 128 + q * 16));
                u32x4g o0, o1;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    o0[2 * e] = __builtin_amdgcn_perm(r[e], l[e], 0x05040100u); o0[2 * e + 1] = __builtin_amdgcn_perm(r[e], l[e], 0x07060302u);
                    o1[2 * e] = __builtin_amdgcn_perm(r[2 + e], l[2 + e], 0x05040100u); o1[2 * e + 1] = __builtin_amdgcn_perm(r[2 + e], l[2 + e], 0x07060302u);
                }
                *reinterpret_cast<u32x4g*>(d + q * 32) = o0;
                *reinterpret_cast<u32x4g*>(d + q * 32 + 16) = o1;
            }
        }
        *adp = (uint32_t)(s0 >> 4) | ((uint32_t)(s1 >> 4) << 16);
    }
}

// ---- pair image of a tile's halo: rows ty0-3 .. ty0+11, pair columns tx0-3 .. tx0+11 (right elements up to tx0+12), zeros outside.
//      Item = (row, pair column, 8-channel chunk): 1 800 items, 7-8 per thread, two 16-byte loads each (a pixel and its right
//      neighbour; 32-bit byte offsets from the image's base).  Split in two so that the loads are in flight while the offsets are
//      staged and the records built: gs_image_load issues a batch, gs_image_store interleaves it into pairs and writes LDS. ----
#define GS_IMG_ITEMS (GS_ROWS * 15 * 8)
template <int B0, int NB>
__device__ static inline void gs_image_load(u32x4g (&L)[NB], u32x4g (&R)[NB], const bf16_t* __restrict__ X, int tid, int ty0, int tx0, int H, int W) {
    const unsigned char* const xb = reinterpret_cast<const unsigned char*>(X);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int idc = min(tid + 256 * (B0 + i), GS_IMG_ITEMS - 1);
        const int q = idc & 7, pc = idc >> 3, r = pc / 15, c = pc - 15 * r;
        const int hy = ty0 - GS_MG + r, hx = tx0 - GS_MG + c;
        const bool rok = (unsigned)hy < (unsigned)H;
        const bool lok = rok && (unsigned)hx < (unsigned)W, rgt = rok && (unsigned)(hx + 1) < (unsigned)W;
        const uint32_t ofs = (uint32_t)((hy * W + hx) * 128 + q * 16);
        L[i] = *reinterpret_cast<const u32x4g*>(xb + (lok ? ofs : 0u));
        R[i] = *reinterpret_cast<const u32x4g*>(xb + (rgt ? ofs + 128u : 0u));
    }
}
template <int B0, int NB>
__device__ static inline void gs_image_store(unsigned char* pi, const u32x4g (&L)[NB], const u32x4g (&R)[NB], int tid, int ty0, int tx0, int H, int W) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int idx = tid + 256 * (B0 + i);
        const int idc = min(idx, GS_IMG_ITEMS - 1);
        const int q = idc & 7, pc = idc >> 3, r = pc / 15, c = pc - 15 * r;
        const int hy = ty0 - GS_MG + r, hx = tx0 - GS_MG + c;
        const bool rok = (unsigned)hy < (unsigned)H;
        const uint32_t ml = (rok && (unsigned)hx < (unsigned)W) ? 0xffffffffu : 0u, mr = (rok && (unsigned)(hx + 1) < (unsigned)W) ? 0xffffffffu : 0u;
        const u32x4g l = L[i] & ml, rr = R[i] & mr;
        u32x4g o0, o1;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            o0[2 * d] = __builtin_amdgcn_perm(rr[d], l[d], 0x05040100u);          // {l.lo, r.lo}
            o0[2 * d + 1] = __builtin_amdgcn_perm(rr[d], l[d], 0x07060302u);      // {l.hi, r.hi}
            o1[2 * d] = __builtin_amdgcn_perm(rr[2 + d], l[2 + d], 0x05040100u);
            o1[2 * d + 1] = __builtin_amdgcn_perm(rr[2 + d], l[2 + d], 0x07060302u);
        }
        if (idx < GS_IMG_ITEMS) {
            unsigned char* const d = pi + gs_pix(r, c) + q * 32;
            *reinterpret_cast<u32x4g*>(d) = o0;
            *reinterpret_cast<u32x4g*>(d + 16) = o1;
        }
    }
}

template <bool STATS>
__global__ __launch_bounds__(256, 2) void dcn_fwd_gs_kernel(const GsFwdGeom g) {
    CN_MAIN_PRIO_SET();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const PI = smem;
    const u32x4g* const REC = reinterpret_cast<const u32x4g*>(smem + GS_REC_OFS);
    float* const OMS = reinterpret_cast<float*>(smem + GS_OMS_OFS);
    uint32_t* const BMP = reinterpret_cast<uint32_t*>(smem + GS_BMP_OFS);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave & 1, grp = wave >> 1;            // the two waves of a pixel group split the 18 (tap, channel half) units
    const int nl = lane & 31, hh = lane >> 5;

    if (tid < GS_PIX / 4) reinterpret_cast<uint32_t*>(smem + GS_ZERO_OFS)[tid] = 0u;
    if (tid < 64) reinterpret_cast<float*>(smem + GS_BIAS_OFS)[tid] = g.bias[tid];

    // ---- this wave's nine W fragments sets, for the whole launch: unit u <-> (tap, half) = ((9 role + u) >> 1, (9 role + u) & 1);
    //      fragment (s, cb): A operand, lane = output channel 32 cb + nl, 8 input channels 32 half + 16 s + 8 hh .. +7 of the tap
    //      (wp = mode-1 pack [Co][tap * 64 + ci]) ----
    u32x4g wf[9][2][2];
#pragma unroll
    for (int u = 0; u < 9; ++u) {
        const int gu = 9 * role + u, tap = gu >> 1, hf = gu & 1;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                wf[u][s][cb] = *reinterpret_cast<const u32x4g*>(g.wp + (int64_t)(32 * cb + nl) * 576 + tap * 64 + 32 * hf + 16 * s + 8 * hh);
    }

    const int lane_off = hh * 32;                         // bytes: this lane half's 8 dwords of a 16-channel k-step
    const int recpix = grp * 32 + nl;                     // tile pixel of this lane
    float s0[8], s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; }

#pragma unroll 1
    for (int j = 0;; ++j) {
        const int t = gs_tile_of(blockIdx.x, gridDim.x, j, g.ntiles);
        if (t < 0) break;
        const int n = t / g.tiles_img, rt = t - n * g.tiles_img;
        const int ty0 = (rt / g.tiles_w) * GS_T, tx0 = (rt % g.tiles_w) * GS_T;
        const int64_t img = (int64_t)n * g.H * g.W;
        const bf16_t* __restrict__ X = g.x + img * 64;

        // (the thread index is laundered once per tile: the compiler otherwise hoists the per-item index arithmetic of the staging code
        //  out of the tile loop, finds no registers for it next to the W sets, and reloads it from scratch in front of every load)
        int tidv = tid;
        asm volatile("" : "+v"(tidv));
        GS_STAMP(0);
        // ---- everything the tile reads from global memory is requested up front: offsets / mask logits (thread = (pixel, 8-float
        //      part), coalesced 32-byte pieces -> LDS [pixel][27]), its eight halo items (64 registers: the accumulators and the unit loop's pipeline registers are dead here) ----
        float4 oma, omb;
        {
            const int p = tidv >> 2, part = tidv & 3;
            const int y = ty0 + (p >> 3), x = tx0 + (p & 7);
            const bool ok = y < g.H && x < g.W;
            const float* src = g.om + (img + (int64_t)(ok ? y : 0) * g.W + (ok ? x : 0)) * 32 + part * 8;
            oma = *reinterpret_cast<const float4*>(src); omb = *reinterpret_cast<const float4*>(src + 4);
        }
        u32x4g La[8], Ra[8];
        gs_image_load<0, 8>(La, Ra, X, tidv, ty0, tx0, g.H, g.W);
        {
            const int p = tidv >> 2, part = tidv & 3;
            float* d = OMS + p * 27 + part * 8;
            d[0] = oma.x; d[1] = oma.y; d[2] = oma.z;
            if (part < 3) { d[3] = oma.w; d[4] = omb.x; d[5] = omb.y; d[6] = omb.z; d[7] = omb.w; }     // part 3 holds entries 24 .. 26 only
            if (tidv < 19) BMP[tidv] = 0u;
        }
        __syncthreads();              // [S1] staged offsets visible; everybody is done with the previous tile's LDS
        GS_STAMP(1);
        gs_build_records(smem, tidv, ty0, tx0, g.H, g.W);
        GS_STAMP(2);
        gs_image_store<0, 8>(PI, La, Ra, tidv, ty0, tx0, g.H, g.W);
        GS_STAMP(3);
        __syncthreads();              // [S2]
        GS_STAMP(4);
        const int nfar = __builtin_amdgcn_readfirstlane((int)BMP[18]);
        const int npass = nfar ? (nfar + GS_FAR_PER_PASS - 1) / GS_FAR_PER_PASS : 1;

        f32x16_t acc[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[cb][v] = 0.f;

#pragma unroll 1
        for (int pass = 0; pass < npass; ++pass) {
            if (nfar) {
                if (pass) __syncthreads();
                gs_place_far(smem, X, tid, pass, g.H, g.W);
                __syncthreads();
            }
            // Software pipeline over the 18 (unit, k-step) steps: the four 16-byte footprint reads of step i+1 are in flight while step
            // i's dot products and MFMAs issue; sched_barriers keep the compiler from hoisting more (the W sets leave ~60 registers)
            uint32_t P0, P1;
            int at, ab;
            {
                const u32x4g rec = REC[((9 * role) >> 1) * 64 + recpix];
                P0 = rec[0]; P1 = rec[1];
                at = (int)((rec[2] & 0xffffu) << 4) + lane_off + ((9 * role) & 1) * 128;
                ab = (int)((rec[2] >> 16) << 4) + lane_off + ((9 * role) & 1) * 128;
            }
            u32x4g q0 = *reinterpret_cast<const u32x4g*>(PI + at), q1 = *reinterpret_cast<const u32x4g*>(PI + at + 16);
            u32x4g q2 = *reinterpret_cast<const u32x4g*>(PI + ab), q3 = *reinterpret_cast<const u32x4g*>(PI + ab + 16);
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                // ---- k-step 0; k-step 1's reads go out first ----
                const u32x4g r0 = *reinterpret_cast<const u32x4g*>(PI + at + 64), r1 = *reinterpret_cast<const u32x4g*>(PI + at + 80);
                const u32x4g r2 = *reinterpret_cast<const u32x4g*>(PI + ab + 64), r3 = *reinterpret_cast<const u32x4g*>(PI + ab + 80);
                u32x4g nrec = {0u, 0u, 0u, 0u};
                if (u < 8) nrec = REC[((9 * role + u + 1) >> 1) * 64 + recpix];
                __builtin_amdgcn_sched_barrier(0);
                {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = gs_dot2(q2[e], P1, gs_dot2z(q0[e], P0));
                        v[4 + e] = gs_dot2(q3[e], P1, gs_dot2z(q1[e], P0));
                    }
                    u32x4g sb;
#pragma unroll
                    for (int d = 0; d < 4; ++d) sb[d] = pk_bf16(v[2 * d], v[2 * d + 1]);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[u][0][cb]), __builtin_bit_cast(bf16x8_t, sb), acc[cb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- k-step 1; the next unit's record is decoded and its k-step 0 reads go out first ----
                const uint32_t P0c = P0, P1c = P1;
                if (u < 8) {
                    const int hfn = (9 * role + u + 1) & 1;
                    P0 = nrec[0]; P1 = nrec[1];
                    at = (int)((nrec[2] & 0xffffu) << 4) + lane_off + hfn * 128;
                    ab = (int)((nrec[2] >> 16) << 4) + lane_off + hfn * 128;
                    q0 = *reinterpret_cast<const u32x4g*>(PI + at); q1 = *reinterpret_cast<const u32x4g*>(PI + at + 16);
                    q2 = *reinterpret_cast<const u32x4g*>(PI + ab); q3 = *reinterpret_cast<const u32x4g*>(PI + ab + 16);
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = gs_dot2(r2[e], P1c, gs_dot2z(r0[e], P0c));
                        v[4 + e] = gs_dot2(r3[e], P1c, gs_dot2z(r1[e], P0c));
                    }
                    u32x4g sb;
#pragma unroll
                    for (int d = 0; d < 4; ++d) sb[d] = pk_bf16(v[2 * d], v[2 * d + 1]);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[u][1][cb]), __builtin_bit_cast(bf16x8_t, sb), acc[cb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        GS_STAMP(5);
        __syncthreads();              // [S3] every wave is done with the pair image: it becomes the exchange / staging area
        GS_STAMP(6);
        // ---- the two halves of a group meet: role 1 parks its partial sums (fp32 [32 px][68]) ----
        // (lane indices laundered like the thread index above: the addresses below would otherwise be hoisted and spilled)
        int nle = nl, hhe = hh;
        asm volatile("" : "+v"(nle), "+v"(hhe));
        unsigned char* const XCb = PI + grp * (32 * 68 * 4) + nle * (68 * 4) + hhe * 16;
        if (role) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(XCb + (32 * cb + 8 * q) * 4) =
                        make_float4(acc[cb][4 * q], acc[cb][4 * q + 1], acc[cb][4 * q + 2], acc[cb][4 * q + 3]);
        }
        __syncthreads();              // [S4]
        GS_STAMP(7);
        if (!role) {
            unsigned char* const Y = PI + 2 * (32 * 68 * 4) + grp * (32 * 144);       // [32 px][72] bf16
            unsigned char* const Yl = Y + nle * 144 + hhe * 8;
            const unsigned char* const Bl = smem + GS_BIAS_OFS + hhe * 16;
            const float floor_ = g.relu ? 0.f : -__builtin_inff();
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // (lane = pixel, register v of block cb = channel 32 cb + 8 (v >> 2) + 4 hh + (v & 3))
                    const float4 o = *reinterpret_cast<const float4*>(XCb + (32 * cb + 8 * q) * 4);
                    const float4 bv = *reinterpret_cast<const float4*>(Bl + (32 * cb + 8 * q) * 4);
                    float v[4] = {acc[cb][4 * q] + o.x + bv.x, acc[cb][4 * q + 1] + o.y + bv.y, acc[cb][4 * q + 2] + o.z + bv.z,
                                  acc[cb][4 * q + 3] + o.w + bv.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] < floor_ ? floor_ : v[e];          // ReLU that keeps a NaN a NaN
                    uint2 w2;
                    w2.x = pk_bf16(v[0], v[1]); w2.y = pk_bf16(v[2], v[3]);
                    *reinterpret_cast<uint2*>(Yl + (32 * cb + 8 * q) * 2) = w2;
                }
            __builtin_amdgcn_wave_barrier();
            // a pixel's 128 bytes leave as eight 16-byte lanes; a lane visits the same 8-channel chunk in every pass (BN statistics)
            int lne = lane;
            asm volatile("" : "+v"(lne));
            const int p0 = lne >> 3, ch = lne & 7;
            const int ox = tx0 + p0;                       // pass i: pixel p0 + 8 i = (row i, column p0) of the group
            const unsigned char* const Yr = Y + p0 * 144 + ch * 16;
            bf16_t* const yo = g.y + (img + (int64_t)(ty0 + 4 * grp) * g.W + ox) * 64 + ch * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4g o = *reinterpret_cast<const u32x4g*>(Yr + i * (8 * 144));
                if (ty0 + 4 * grp + i < g.H && ox < g.W) {
                    *reinterpret_cast<u32x4g*>(yo + (int64_t)i * g.W * 64) = o;
                    if (STATS) { const uint32_t w4[4] = {o[0], o[1], o[2], o[3]}; bn_stat_add(s0, s1, w4); }
                }
            }
        }
        GS_STAMP(8);
    }
    // BN statistics of everything this workgroup stored (sink protocol of bn.hip); role-1 waves contribute zeros
    if (STATS) {
        __syncthreads();
        bn_stats_flush<8, 256>(s0, s1, reinterpret_cast<float*>(PI), g.bn_part, g.bn_slots, 64, 0, 64, blockIdx.x, tid);
    }
}

bool dcn_fwd_gs_shape_ok(int Ci, int x_ld, int Co, int y_ld, int om_ld) {
    static const bool disabled = getenv("CN_DISABLE_DCN_GS") != nullptr || getenv("CN_DISABLE_DCN_FWD_GS") != nullptr;
    return !disabled && Ci == 64 && x_ld == 64 && om_ld == 32 && Co == 64 && y_ld == 64;
}

static int gs_grid(int ntiles) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    int G = 2 * cus;                                     // two resident workgroups per CU
    if (ntiles < G) G = ntiles >= 8 ? (ntiles / 8) * 8 : ntiles;
    return G;
}

// returns false when the shape is not handled here (caller falls back to the blend-matrix / gather kernels)
bool dcn_fwd_gs_launch(const void* x, const float* om, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci, int x_ld,
                       int Co, int y_ld, int om_ld, int ktot, int relu, float* bn_part, int bn_slots, hipStream_t st) {
    if (!dcn_fwd_gs_shape_ok(Ci, x_ld, Co, y_ld, om_ld) || bias == nullptr || ktot != 9 * 64) return false;
    if (((uintptr_t)x | (uintptr_t)wp | (uintptr_t)y | (uintptr_t)om | (uintptr_t)bias) & 15) return false;
    if (H > 32767 || W > 32767 || (int64_t)H * W * 128 > 0x7fffffff) return false;
    GsFwdGeom g;
    g.x = (const bf16_t*)x; g.om = om; g.wp = (const bf16_t*)wp; g.bias = bias; g.y = (bf16_t*)y;
    g.N = N; g.H = H; g.W = W; g.relu = relu; g.bn_part = bn_part; g.bn_slots = bn_slots;
    g.tiles_w = (W + GS_T - 1) / GS_T;
    g.tiles_img = g.tiles_w * ((H + GS_T - 1) / GS_T);
    const int64_t nt = (int64_t)g.tiles_img * N;
    if (nt > 0x7fffffff) return false;
    g.ntiles = (int)nt;
    if (bn_part) {
        bn_sink_mark_taken();
        (void)hipFuncSetAttribute((const void*)dcn_fwd_gs_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GS_SMEM);
        hipLaunchKernelGGL(dcn_fwd_gs_kernel<true>, dim3(gs_grid(g.ntiles)), dim3(256), GS_SMEM, st, g);
    } else {
        (void)hipFuncSetAttribute((const void*)dcn_fwd_gs_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GS_SMEM);
        hipLaunchKernelGGL(dcn_fwd_gs_kernel<false>, dim3(gs_grid(g.ntiles)), dim3(256), GS_SMEM, st, g);
    }
    return true;
}
