// DCNv2 for 64-channel layers, "gather-sample" formulation (round 5; bf16).  SURVEY App. A; pose_dla_dcn.py:441-449.
//
//   S_k[p][ci] = sum_{corners c} w_c(p,k) * m(p,k) * x[corner_c(p,k)][ci]           (bilinear sample x sigmoid(mask))
//   y[p][co]   = bias[co] + sum_k sum_ci W_k[co][ci] * S_k[p][ci]
//
// The blend-matrix kernels (dcn_bm.hip) run the blend as a second MFMA; they sit at 0.15-0.25 MFMA-pipe busy with 14-22 VALU
// instructions per MFMA (profiles/r04_pmc_sq.txt): selects that build blend-matrix rows, zero fills, a barrier per tap for the
// weight slices, every tile's loads exposed.  Here:
//   * the blend is FOUR packed-fp16 FMAs per channel PAIR: the halo image sits in LDS as fp16 (converted from the bf16 activations
//     once per halo pixel: exact for |x| < 65 504, +-inf beyond — fp16 has 3 more mantissa bits than bf16 and the layer inputs are
//     BatchNorm outputs), a lane reads the four corner pixels of its footprint (16 bytes = 8 channels each) and
//         S = x00 * w00 m + x01 * w01 m + x10 * w10 m + x11 * w11 m         (v_pk_mul_f16 + 3 v_pk_fma_f16 per 2 channels,
//     the weights broadcast from packed fp16 pairs with op_sel) — no unpacking, no pairing, no conversion: the fp16 result pairs
//     ARE the operand of v_mfma_f32_32x32x16_f16 (the weights are converted bf16 -> fp16 once per workgroup, exactly).  The
//     instruction count is what bounds these kernels: every VALU instruction costs a SIMD 4 cycles, two waves per SIMD — measured
//     on the two bf16 forms of this kernel (v_perm pairing + v_dot2_f32_bf16: 44 instructions per (unit, k-step), 361 cycles; a
//     pre-paired image: 30, 233 cycles); this form has 22.  The blend weights and S round to fp16 (2^-11; the bf16 kernels: 2^-8);
//   * lane = (pixel, channel half): the dot2 results of a lane, packed to bf16, ARE the B operand of the contraction MFMA
//     (lane = pixel column, 8 consecutive channels per k-step) — nothing is transposed, nothing goes back through LDS;
//   * the geometry of a (pixel, tap) — two packed weight pairs and the LDS positions of its four corner pixels — is a 16-byte
//     RECORD built once per tile from the fp32 offsets / mask logits (floor, fractions, sigmoid, image-border zeroing happen there,
//     once, instead of once per lane half and tap): a unit costs a lane one 16-byte LDS read of geometry;
//   * the weights are STATIONARY IN REGISTERS: the two waves that share a 4x8 pixel group split the 18 (tap, channel half) units
//     9 : 9, each holding its 9 x 16 registers of W fragments for the whole launch (persistent workgroups, one per CU), and meet
//     once per tile through LDS.  No weight traffic in the loop, no barrier per tap: a tile is four barriers;
//   * the bf16 halo and the raw offsets of tile t+1 travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging
//     registers, no VALU but the address) into a staging buffer WHILE tile t is computed — one 1 KB piece per wave behind each of
//     the first eight units; the fp16 image is made from the staging buffer at the top of the tile.
// Halo image: 15 rows x 24 pixel slots x 128 B per 8x16-pixel tile (offsets up to |d| < 2 px stay inside), the eight 16-byte
// chunks of pixel slot n XOR-swizzled with (n >> 1) & 7, 24 slots per row: the 16 lanes of a ds_read_b128 group (4x8 pixel groups)
// then touch 16 different slot numbers mod 16 and tile all 64 banks.
// Samples whose footprint leaves the halo: their four corner pixels are copied from global memory into spare pixel slots behind
// the image (32 slots = 8 such samples per pass) and the record points there — the hot loop knows nothing about them; a tile with
// more runs the unit loop again for the next 8 with every other record pointed at an all-zero pixel (any offset field is handled;
// a field with many far samples pays for it).
#include "conv_common.h"
#include <stdlib.h>

#define GQ_TH 8                                  // tile: 8 rows x 16 columns = four 4x8 pixel groups
#define GQ_TW 16
#define GQ_MG 3                                  // halo margin above / left of the tile (4 below / right)
#define GQ_ROWS (GQ_TH + 7)                      // 15 halo rows
#define GQ_P 24                                  // pixel slots per halo row (23 columns used)
#define GQ_NPIX (GQ_TH * GQ_TW)                  // 128
#define GQ_NREC (GQ_NPIX * 9)                    // 1152 records x 16 B
#define GQ_IMG_SLOTS ((GQ_ROWS + 2) * GQ_P)      // 408: the image (360 slots) + 48 spare slots (zero pixel, far corners)
#define GQ_IMG_BYTES (GQ_IMG_SLOTS * 128)        // 52 224: fp16 image (offset 0)
#define GQ_ZERO_SLOT (GQ_ROWS * GQ_P)            // slot 360: all-zero pixel
#define GQ_FAR_SLOT0 (GQ_ROWS * GQ_P + 8)        // slots 368 .. 399: far corners (4 per sample)
#define GQ_FAR_PER_PASS 8
#define GQ_STG_OFS GQ_IMG_BYTES                  // bf16 staging buffer of the NEXT tile's halo: 360 slots x 128 B, linear (DMA target)
#define GQ_REC_OFS (GQ_STG_OFS + GQ_ROWS * GQ_P * 128)       // 98 304
#define GQ_OMS_OFS (GQ_REC_OFS + GQ_NREC * 16)   // raw offsets / mask logits [pixel][28 floats] (DMA: 7 granules per pixel)
#define GQ_HW_OFS (GQ_OMS_OFS + GQ_NPIX * 112)   // corner 00 of every record in image coordinates (int16 pair; read by the far path)
#define GQ_BIAS_OFS (GQ_HW_OFS + GQ_NREC * 4)    // 64 floats
#define GQ_BMP_OFS (GQ_BIAS_OFS + 256)           // 36 dwords: bitmap of the records whose footprint left the halo; dword 36: their count
#define GQ_SMEM (GQ_BMP_OFS + 160)               // 136 096
#define GQ_NPIECE (45 + 14)                      // DMA pieces per tile: 45 image (row r = J / 3, slots 8 (J % 3) .. + 7), 14 offsets

#ifdef GS_PROBE   // development build only (tools/gs_probe.py): cycle stamps of waves 0 / 1 of the first workgroups, first 4 tiles
__device__ unsigned long long gs_ts[64 * 2 * 4 * 16];
#define GS_STAMP(k) do { if (lane == 0 && wave < 2 && blockIdx.x < 64 && j < 4) gs_ts[((blockIdx.x * 2 + wave) * 4 + j) * 16 + (k)] = clock64(); } while (0)
extern "C" int gs_probe_dump(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gs_ts), sizeof(gs_ts)); }
#else
#define GS_STAMP(k) do { } while (0)
#endif

typedef uint32_t u32x4g __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2g __attribute__((ext_vector_type(2)));

__device__ uint4 gq_zero_page[8];                // 128 zero bytes: DMA source of halo pixels outside the image

struct GsFwdGeom {
    const bf16_t* x; const float* om; const bf16_t* wp; const float* bias; bf16_t* y;
    int N, H, W, relu;
    float* bn_part; int bn_slots;
    int tiles_w, tiles_img, ntiles;
};

typedef _Float16 f16x2g __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8g __attribute__((ext_vector_type(8)));
typedef float f32x2g __attribute__((ext_vector_type(2)));
__device__ static inline uint32_t gq_pk_f16(float a, float b) {
    const f32x2g v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2g));
}
// a packed bf16 pair as a packed fp16 pair (exact below 65 504 in magnitude)
__device__ static inline uint32_t gq_bf2h(uint32_t d) { return gq_pk_f16(__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u)); }
__device__ static inline u32x4g gq_bf2h4(u32x4g v) { return u32x4g{gq_bf2h(v[0]), gq_bf2h(v[1]), gq_bf2h(v[2]), gq_bf2h(v[3])}; }
// packed fp16: x * w.lo, x * w.lo + c, x * w.hi + c  (the weight half broadcast to both lanes with op_sel)
__device__ static inline uint32_t gq_mul_lo(uint32_t x, uint32_t w) {
    uint32_t d;
    asm("v_pk_mul_f16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(x), "v"(w));
    return d;
}
__device__ static inline uint32_t gq_fma_lo(uint32_t x, uint32_t w, uint32_t c) {
    uint32_t d;
    asm("v_pk_fma_f16 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(x), "v"(w), "v"(c));
    return d;
}
__device__ static inline uint32_t gq_fma_hi(uint32_t x, uint32_t w, uint32_t c) {
    uint32_t d;
    asm("v_pk_fma_f16 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d) : "v"(x), "v"(w), "v"(c));
    return d;
}
// the blend of 8 channels: four corner vectors, two weight pairs
__device__ static inline u32x4g gq_blend(const u32x4g& a, const u32x4g& b, const u32x4g& c, const u32x4g& d, uint32_t w01, uint32_t w23) {
    u32x4g o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = gq_fma_hi(d[i], w23, gq_fma_lo(c[i], w23, gq_fma_hi(b[i], w01, gq_mul_lo(a[i], w01))));
    return o;
}

// position code of pixel slot n inside an image buffer: byte offset of its chunk 0 | swizzle key << 4; logical 16-byte chunk q of
// the pixel lives at byte offset code ^ (q << 4)
__device__ static inline uint32_t gq_code(int n) { return ((uint32_t)n << 7) | ((((uint32_t)n >> 1) & 7u) << 4); }

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

// ---- one 1 KB piece J of a tile's bf16 halo (J < 45: halo row J / 3, pixel slots 8 (J % 3) .. + 7, lane l = chunk l & 7 of slot
//      l >> 3; pixels outside the image fetch the zero page) or of its raw offsets (granule g = 64 (J - 45) + l = (pixel g / 7,
//      16-byte part g % 7) -> [pixel][28 floats]) by LDS-DMA into the staging areas ----
__device__ static inline void gq_issue_piece(const GsFwdGeom& g, int J, const char* Xb, const char* Ob, int ty0, int tx0, unsigned lds_base, int ln) {
    const char* const zp = reinterpret_cast<const char*>(gq_zero_page) + (ln & 7) * 16;
    const char* src;
    unsigned dst;
    if (J < 45) {
        const int r = J / 3, seg = J - 3 * r;
        const int hy = ty0 - GQ_MG + r, hx = tx0 - GQ_MG + 8 * seg + (ln >> 3);
        const bool ok = (unsigned)hy < (unsigned)g.H && (unsigned)hx < (unsigned)g.W;
        src = ok ? Xb + (uint32_t)((hy * g.W + hx) * 128 + (ln & 7) * 16) : zp;
        dst = lds_base + (unsigned)(GQ_STG_OFS + J * 1024);
    } else {
        const int gi = (J - 45) * 64 + ln, p = gi / 7, part = gi - 7 * p;
        const int y = ty0 + (p >> 4), x = tx0 + (p & 15);
        const bool ok = y < g.H && x < g.W;
        src = ok ? Ob + (uint32_t)((y * g.W + x) * 128 + part * 16) : zp;
        dst = lds_base + (unsigned)(GQ_OMS_OFS + (J - 45) * 1024);
    }
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

// ---- the fp16 image of a tile from the bf16 staging buffer: item = (pixel slot, 16-byte chunk), 2 880 items, 5-6 per thread ----
__device__ static inline void gq_convert_image(unsigned char* smem, int tid) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int i = tid + 512 * k;
        if (i < GQ_ROWS * GQ_P * 8) {
            const u32x4g v = *reinterpret_cast<const u32x4g*>(smem + GQ_STG_OFS + i * 16);
            *reinterpret_cast<u32x4g*>(smem + (gq_code(i >> 3) ^ (uint32_t)((i & 7) << 4))) = gq_bf2h4(v);
        }
    }
}

// ---- geometry records of a tile: built from the staged offsets by all 512 threads ----
// record (tap, pixel) = { P0 = fp16 pair (w00 m, w01 m), P1 = (w10 m, w11 m), position codes of the corner pixels 00 | 01 << 16 and
// 10 | 11 << 16 (relative to the image buffer) }.  Corners outside the image have weight 0.  A footprint outside the halo is noted
// in the bitmap and pointed at the zero pixel until gq_place_far gives it slots.
__device__ static inline void gq_build_records(unsigned char* smem, int tid, int ty0, int tx0, int H, int W) {
    const float* const oms = reinterpret_cast<const float*>(smem + GQ_OMS_OFS);
    u32x4g* const rec = reinterpret_cast<u32x4g*>(smem + GQ_REC_OFS);
    uint32_t* const hwp = reinterpret_cast<uint32_t*>(smem + GQ_HW_OFS);
    uint32_t* const bmp = reinterpret_cast<uint32_t*>(smem + GQ_BMP_OFS);
    constexpr uint32_t ZC = ((uint32_t)GQ_ZERO_SLOT << 7) | ((((uint32_t)GQ_ZERO_SLOT >> 1) & 7u) << 4);
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {
        const int idx = tid + 512 * i;
        if (idx < GQ_NREC) {
            const int tap = idx >> 7, p = idx & 127;
            const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;           // tap / 3, tap % 3 for tap < 9
            // tile pixel p = group (p >> 5), pixel (p & 31) of the 4x8 group: groups 0 / 1 = columns 0-7 / 8-15 of rows 0-3, 2 / 3 of rows 4-7
            const int gq = p >> 5, py_t = 4 * (gq >> 1) + ((p & 31) >> 3), px_t = 8 * (gq & 1) + (p & 7);
            const int y = ty0 + py_t, x = tx0 + px_t;
            const bool live = y < H && x < W;
            const float* o = oms + (py_t * GQ_TW + px_t) * 28;
            // the sampling position is formed like the reference forms it: ONE fp32 add of the integer position and the offset
            const float py = (float)(y - 1 + ky) + o[2 * tap], px = (float)(x - 1 + kx) + o[2 * tap + 1];
            const float m = live ? __builtin_amdgcn_rcpf(1.f + __expf(-o[18 + tap])) : 0.f;
            const float fy = floorf(py), fx = floorf(px);
            const int h0 = (int)fminf(fmaxf(fy, -2.f), 32766.f), w0 = (int)fminf(fmaxf(fx, -2.f), 32766.f);
            const float ly = py - fy, lx = px - fx;
            const float wa = (1.f - ly) * m, wb = ly * m;
            const bool h0ok = (unsigned)h0 < (unsigned)H, h1ok = (unsigned)(h0 + 1) < (unsigned)H;
            const bool w0ok = (unsigned)w0 < (unsigned)W, w1ok = (unsigned)(w0 + 1) < (unsigned)W;
            const uint32_t P0 = gq_pk_f16((h0ok && w0ok) ? wa * (1.f - lx) : 0.f, (h0ok && w1ok) ? wa * lx : 0.f);
            const uint32_t P1 = gq_pk_f16((h1ok && w0ok) ? wb * (1.f - lx) : 0.f, (h1ok && w1ok) ? wb * lx : 0.f);
            const int wr = h0 - (ty0 - GQ_MG), wc = w0 - (tx0 - GQ_MG);
            const bool nz = ((P0 | P1) & 0x7fff7fffu) != 0u;
            const bool inwin = (unsigned)wr <= (unsigned)(GQ_ROWS - 2) && (unsigned)wc <= (unsigned)(GQ_P - 3);
            uint32_t c01 = ZC | (ZC << 16), c23 = c01;
            if (nz && inwin) {
                const int s00 = wr * GQ_P + wc;
                c01 = gq_code(s00) | (gq_code(s00 + 1) << 16);
                c23 = gq_code(s00 + GQ_P) | (gq_code(s00 + GQ_P + 1) << 16);
            }
            if (nz && !inwin) { atomicOr(bmp + (idx >> 5), 1u << (idx & 31)); atomicAdd(bmp + 36, 1u); }
            u32x4g r;
            r[0] = nz ? P0 : 0u; r[1] = nz ? P1 : 0u; r[2] = c01; r[3] = c23;
            rec[idx] = r;
            hwp[idx] = ((uint32_t)h0 & 0xffffu) | ((uint32_t)w0 << 16);
        }
    }
}

// ---- pass k over a tile with far footprints (rare): records of far sample number 8 k .. 8 k + 7 get four spare pixel slots filled
//      from global memory, every other far record — and, from the second pass on, every near record (they were accumulated in pass
//      0) — points at the zero pixel.  Each thread serves its own records. ----
__device__ static __attribute__((noinline)) void gq_place_far(unsigned char* smem, unsigned char* imgb, const bf16_t* X, int tid, int k, int H, int W) {
    u32x4g* const rec = reinterpret_cast<u32x4g*>(smem + GQ_REC_OFS);
    const uint32_t* const hwp = reinterpret_cast<const uint32_t*>(smem + GQ_HW_OFS);
    const uint32_t* const bmp = reinterpret_cast<const uint32_t*>(smem + GQ_BMP_OFS);
    const unsigned char* const xb = reinterpret_cast<const unsigned char*>(X);
    constexpr uint32_t ZC = ((uint32_t)GQ_ZERO_SLOT << 7) | ((((uint32_t)GQ_ZERO_SLOT >> 1) & 7u) << 4);
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {
        const int idx = tid + 512 * i;
        if (idx >= GQ_NREC) break;
        const uint32_t word = bmp[idx >> 5];
        uint32_t* const cp = reinterpret_cast<uint32_t*>(rec + idx) + 2;
        if (!((word >> (idx & 31)) & 1u)) { if (k > 0) { cp[0] = ZC | (ZC << 16); cp[1] = ZC | (ZC << 16); } continue; }
        int ord = __builtin_popcount(word & ((1u << (idx & 31)) - 1u));
        for (int w = 0; w < (idx >> 5); ++w) ord += __builtin_popcount(bmp[w]);
        if (ord / GQ_FAR_PER_PASS != k) { cp[0] = ZC | (ZC << 16); cp[1] = ZC | (ZC << 16); continue; }
        const int s0 = GQ_FAR_SLOT0 + 4 * (ord % GQ_FAR_PER_PASS);
        const uint32_t hw = hwp[idx];
        const int h0 = (int)(short)(hw & 0xffffu), w0 = (int)hw >> 16;
#pragma unroll 1
        for (int cnr = 0; cnr < 4; ++cnr) {
            const int hc = min(max(h0 + (cnr >> 1), 0), H - 1), wcc = min(max(w0 + (cnr & 1), 0), W - 1);
            const uint32_t code = gq_code(s0 + cnr);
#pragma unroll 1
            for (int q = 0; q < 8; ++q)
                *reinterpret_cast<u32x4g*>(imgb + (code ^ (uint32_t)(q << 4))) = gq_bf2h4(*reinterpret_cast<const u32x4g*>(xb + (uint32_t)((hc * W + wcc) * 128 + q * 16)));
        }
        cp[0] = gq_code(s0) | (gq_code(s0 + 1) << 16);
        cp[1] = gq_code(s0 + 2) | (gq_code(s0 + 3) << 16);
    }
}

template <bool STATS>
__global__ __launch_bounds__(512) void dcn_fwd_gs_kernel(const GsFwdGeom g) {
    CN_MAIN_PRIO_SET();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32x4g* const REC = reinterpret_cast<const u32x4g*>(smem + GQ_REC_OFS);
    uint32_t* const BMP = reinterpret_cast<uint32_t*>(smem + GQ_BMP_OFS);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned char* const IMG = smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave & 1, grp = wave >> 1;            // the two waves of a pixel group split the 18 (tap, channel half) units
    const int nl = lane & 31, hh = lane >> 5;

    // zero pixel + far slots, bias
    for (int i = tid; i < 48 * 32; i += 512) reinterpret_cast<uint32_t*>(smem + GQ_ZERO_SLOT * 128)[i] = 0u;
    if (tid < 64) reinterpret_cast<float*>(smem + GQ_BIAS_OFS)[tid] = g.bias[tid];

    // the first tile's pieces
    {
        const int t0 = gs_tile_of(blockIdx.x, gridDim.x, 0, g.ntiles);
        if (t0 >= 0) {
            const int n = t0 / g.tiles_img, rt = t0 - n * g.tiles_img;
            const int64_t img = (int64_t)n * g.H * g.W;
#pragma unroll 1
            for (int J = wave; J < GQ_NPIECE; J += 8)
                gq_issue_piece(g, J, reinterpret_cast<const char*>(g.x + img * 64), reinterpret_cast<const char*>(g.om + img * 32),
                               (rt / g.tiles_w) * GQ_TH, (rt % g.tiles_w) * GQ_TW, lds_base, lane);
        }
    }

    // ---- this wave's nine W fragments sets, for the whole launch: unit u <-> (tap, half) = ((9 role + u) >> 1, (9 role + u) & 1);
    //      fragment (s, cb): A operand, lane = output channel 32 cb + nl, 8 input channels 32 half + 16 s + 8 hh .. +7 of the tap
    //      (wp = mode-1 pack [Co][tap * 64 + ci], bf16 -> fp16: exact, the weights are far inside the fp16 range) ----
    u32x4g wf[9][2][2];
#pragma unroll
    for (int u = 0; u < 9; ++u) {
        const int gu = 9 * role + u, tap = gu >> 1, hf = gu & 1;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                wf[u][s][cb] = gq_bf2h4(*reinterpret_cast<const u32x4g*>(g.wp + (int64_t)(32 * cb + nl) * 576 + tap * 64 + 32 * hf + 16 * s + 8 * hh));
    }

    const int recpix = grp * 32 + nl;                     // record index of this lane's pixel
    float s0[8], s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; }

#pragma unroll 1
    for (int j = 0;; ++j) {
        const int t = gs_tile_of(blockIdx.x, gridDim.x, j, g.ntiles);
        if (t < 0) break;
        const int n = t / g.tiles_img, rt = t - n * g.tiles_img;
        const int ty0 = (rt / g.tiles_w) * GQ_TH, tx0 = (rt % g.tiles_w) * GQ_TW;
        const int64_t img = (int64_t)n * g.H * g.W;
        const bf16_t* __restrict__ X = g.x + img * 64;
        // the next tile: its DMA pieces are issued from inside the unit loop
        const int tn = gs_tile_of(blockIdx.x, gridDim.x, j + 1, g.ntiles);
        const int nn = tn >= 0 ? tn / g.tiles_img : 0, rtn = tn >= 0 ? tn - nn * g.tiles_img : 0;
        const int tyn = (rtn / g.tiles_w) * GQ_TH, txn = (rtn % g.tiles_w) * GQ_TW;
        const char* const Xn = reinterpret_cast<const char*>(g.x + (int64_t)nn * g.H * g.W * 64);
        const char* const On = reinterpret_cast<const char*>(g.om + (int64_t)nn * g.H * g.W * 32);

        // (the thread index is laundered once per tile: the compiler otherwise hoists per-thread index arithmetic out of the tile loop,
        //  finds no registers for it next to the W sets, and reloads it from scratch)
        int tidv = tid, lnv = lane;
        asm volatile("" : "+v"(tidv), "+v"(lnv));
        GS_STAMP(0);
        if (tidv < 37) BMP[tidv] = 0u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this tile's DMA (and the previous tile's stores) retired
        __syncthreads();              // [S1] staging + raw offsets visible; everybody is done with the previous tile's LDS
        GS_STAMP(1);
        gq_convert_image(smem, tidv);
        GS_STAMP(2);
        gq_build_records(smem, tidv, ty0, tx0, g.H, g.W);
        GS_STAMP(3);
        __syncthreads();              // [S2]
        GS_STAMP(4);
        const int nfar = __builtin_amdgcn_readfirstlane((int)BMP[36]);
        const int npass = nfar ? (nfar + GQ_FAR_PER_PASS - 1) / GQ_FAR_PER_PASS : 1;

        f32x16_t acc[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[cb][v] = 0.f;

#pragma unroll 1
        for (int pass = 0; pass < npass; ++pass) {
            if (nfar) {
                if (pass) __syncthreads();
                gq_place_far(smem, IMG, X, tid, pass, g.H, g.W);
                __syncthreads();
            }
            const bool dma = pass == 0 && tn >= 0;
            // Software pipeline over the 18 (unit, k-step) steps: the four 16-byte corner reads of step i+1 are in flight while step i's
            // blend and MFMAs issue; sched_barriers keep the compiler from hoisting more (the W sets leave ~60 registers)
            uint32_t P0, P1, c0, c1, c2, c3;
            {
                const u32x4g rec = REC[((9 * role) >> 1) * GQ_NPIX + recpix];
                const uint32_t xq = (uint32_t)((((9 * role) & 1) * 4 + hh) << 4);
                P0 = rec[0]; P1 = rec[1];
                c0 = (rec[2] & 0xffffu) ^ xq; c1 = (rec[2] >> 16) ^ xq; c2 = (rec[3] & 0xffffu) ^ xq; c3 = (rec[3] >> 16) ^ xq;
            }
            u32x4g q0 = *reinterpret_cast<const u32x4g*>(IMG + c0), q1 = *reinterpret_cast<const u32x4g*>(IMG + c1);
            u32x4g q2 = *reinterpret_cast<const u32x4g*>(IMG + c2), q3 = *reinterpret_cast<const u32x4g*>(IMG + c3);
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                // ---- k-step 0; k-step 1's reads (logical chunk + 2: position ^ 32) go out first ----
                const u32x4g r0 = *reinterpret_cast<const u32x4g*>(IMG + (c0 ^ 32u)), r1 = *reinterpret_cast<const u32x4g*>(IMG + (c1 ^ 32u));
                const u32x4g r2 = *reinterpret_cast<const u32x4g*>(IMG + (c2 ^ 32u)), r3 = *reinterpret_cast<const u32x4g*>(IMG + (c3 ^ 32u));
                u32x4g nrec = {0u, 0u, 0u, 0u};
                if (u < 8) nrec = REC[((9 * role + u + 1) >> 1) * GQ_NPIX + recpix];
                __builtin_amdgcn_sched_barrier(0);
                {
                    const u32x4g sb = gq_blend(q0, q1, q2, q3, P0, P1);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8g, wf[u][0][cb]), __builtin_bit_cast(f16x8g, sb), acc[cb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- k-step 1; the next unit's record is decoded and its k-step 0 reads go out first ----
                const uint32_t P0c = P0, P1c = P1;
                if (u < 8) {
                    const uint32_t xq = (uint32_t)((((9 * role + u + 1) & 1) * 4 + hh) << 4);
                    P0 = nrec[0]; P1 = nrec[1];
                    c0 = (nrec[2] & 0xffffu) ^ xq; c1 = (nrec[2] >> 16) ^ xq; c2 = (nrec[3] & 0xffffu) ^ xq; c3 = (nrec[3] >> 16) ^ xq;
                    q0 = *reinterpret_cast<const u32x4g*>(IMG + c0); q1 = *reinterpret_cast<const u32x4g*>(IMG + c1);
                    q2 = *reinterpret_cast<const u32x4g*>(IMG + c2); q3 = *reinterpret_cast<const u32x4g*>(IMG + c3);
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    const u32x4g sb = gq_blend(r0, r1, r2, r3, P0c, P1c);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8g, wf[u][1][cb]), __builtin_bit_cast(f16x8g, sb), acc[cb], 0, 0, 0);
                }
                // one DMA piece of the next tile behind each of the first eight units (its issue hides behind the MFMAs)
#ifndef GQ_DMA_AFTER
                if (u < 8 && dma && wave + 8 * u < GQ_NPIECE) gq_issue_piece(g, wave + 8 * u, Xn, On, tyn, txn, lds_base, lnv);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }

#ifdef GQ_DMA_AFTER      // timing experiment: all pieces behind the unit loop
        if (tn >= 0) {
#pragma unroll 1
            for (int J = wave; J < GQ_NPIECE; J += 8) gq_issue_piece(g, J, Xn, On, tyn, txn, lds_base, lnv);
        }
#endif
        GS_STAMP(5);
        __syncthreads();              // [S3] every wave is done with the image and the records: they become the exchange / staging area
        GS_STAMP(6);
        // ---- the two halves of a group meet: role 1 parks its partial sums + the bias (fp32 [32 px][68]) in the image buffer ----
        // (lane indices laundered like the thread index above: the addresses below would otherwise be hoisted and spilled)
        int nle = nl, hhe = hh;
        asm volatile("" : "+v"(nle), "+v"(hhe));
        unsigned char* const XCb = IMG + grp * (32 * 68 * 4) + nle * (68 * 4) + hhe * 16;
        if (role) {
            const unsigned char* const Bl = smem + GQ_BIAS_OFS + hhe * 16;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // (lane = pixel, register v of block cb = channel 32 cb + 8 (v >> 2) + 4 hh + (v & 3))
                    const float4 bv = *reinterpret_cast<const float4*>(Bl + (32 * cb + 8 * q) * 4);
                    *reinterpret_cast<float4*>(XCb + (32 * cb + 8 * q) * 4) =
                        make_float4(acc[cb][4 * q] + bv.x, acc[cb][4 * q + 1] + bv.y, acc[cb][4 * q + 2] + bv.z, acc[cb][4 * q + 3] + bv.w);
                }
        }
        __syncthreads();              // [S4]
        GS_STAMP(7);
        if (!role) {
            unsigned char* const Y = smem + GQ_REC_OFS + grp * (32 * 144);            // [32 px][72] bf16 (the records are dead)
            unsigned char* const Yl = Y + nle * 144 + hhe * 8;
            const float floor_ = g.relu ? 0.f : -__builtin_inff();
            float4 o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = *reinterpret_cast<const float4*>(XCb + i * 32);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 ov = o[4 * cb + q];
                    float v[4] = {acc[cb][4 * q] + ov.x, acc[cb][4 * q + 1] + ov.y, acc[cb][4 * q + 2] + ov.z, acc[cb][4 * q + 3] + ov.w};
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
            const int oy0 = ty0 + 4 * (grp >> 1), ox = tx0 + 8 * (grp & 1) + p0;          // pass i: pixel (row i, column p0) of the group
            const unsigned char* const Yr = Y + p0 * 144 + ch * 16;
            bf16_t* const yo = g.y + (img + (int64_t)oy0 * g.W + ox) * 64 + ch * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4g ov = *reinterpret_cast<const u32x4g*>(Yr + i * (8 * 144));
                if (oy0 + i < g.H && ox < g.W) {
                    *reinterpret_cast<u32x4g*>(yo + (int64_t)i * g.W * 64) = ov;
                    if (STATS) { const uint32_t w4[4] = {ov[0], ov[1], ov[2], ov[3]}; bn_stat_add(s0, s1, w4); }
                }
            }
        }
        GS_STAMP(8);
    }
    // BN statistics of everything this workgroup stored (sink protocol of bn.hip); role-1 waves contribute zeros
    if (STATS) {
        __syncthreads();
        bn_stats_flush<8, 512>(s0, s1, reinterpret_cast<float*>(smem), g.bn_part, g.bn_slots, 64, 0, 64, blockIdx.x, tid);
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
    int G = cus;                                         // one resident workgroup per CU
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
    g.tiles_w = (W + GQ_TW - 1) / GQ_TW;
    g.tiles_img = g.tiles_w * ((H + GQ_TH - 1) / GQ_TH);
    const int64_t nt = (int64_t)g.tiles_img * N;
    if (nt > 0x7fffffff) return false;
    g.ntiles = (int)nt;
    if (bn_part) {
        bn_sink_mark_taken();
        (void)hipFuncSetAttribute((const void*)dcn_fwd_gs_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GQ_SMEM);
        hipLaunchKernelGGL(dcn_fwd_gs_kernel<true>, dim3(gs_grid(g.ntiles)), dim3(512), GQ_SMEM, st, g);
    } else {
        (void)hipFuncSetAttribute((const void*)dcn_fwd_gs_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GQ_SMEM);
        hipLaunchKernelGGL(dcn_fwd_gs_kernel<false>, dim3(gs_grid(g.ntiles)), dim3(512), GQ_SMEM, st, g);
    }
    return true;
}
