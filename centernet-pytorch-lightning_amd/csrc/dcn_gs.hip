// DCNv2 for 64-channel layers, "gather-sample" formulation (round 5; bf16 in / out).  SURVEY App. A; pose_dla_dcn.py:441-449.
//
//   S_k[p][ci] = sum_{corners c} w_c(p,k) * m(p,k) * x[corner_c(p,k)][ci]           (bilinear sample x sigmoid(mask))
//   y[p][co]   = bias[co] + sum_k sum_ci W_k[co][ci] * S_k[p][ci]
//
// The blend-matrix kernels (dcn_bm.hip) run the blend as a second MFMA; they sit at 0.15-0.25 MFMA-pipe busy with 14-22 VALU
// instructions per MFMA (profiles/r04_pmc_sq.txt): selects that build blend-matrix rows, zero fills, a barrier per tap for the
// weight slices, every tile's loads exposed, two waves per SIMD.  What bounds such a kernel (tools/probe/valu_rate.hip): a WAVE
// issues one VALU instruction per ~4.9 cycles whatever the opcode (fma, perm, dot2, packed fp16 alike), while a SIMD's throughput
// scales with its waves up to four — so: few instructions per sample, many waves, and no phase in which all of them wait.  Here:
//   * the blend is FOUR packed-fp16 FMAs per channel PAIR: the halo image sits in LDS as fp16 (converted from the bf16 activations
//     once per halo pixel: exact for |x| < 65 504, +-inf beyond — fp16 has 3 more mantissa bits than bf16 and the layer inputs are
//     BatchNorm outputs), a lane reads the four corner pixels of its footprint (16 bytes = 8 channels each) and
//         S = x00 * w00 m + x01 * w01 m + x10 * w10 m + x11 * w11 m         (v_pk_mul_f16 + 3 v_pk_fma_f16 per 2 channels,
//     the weights broadcast from packed fp16 pairs with op_sel) — no unpacking, no pairing, no conversion: the fp16 result pairs
//     ARE the operand of v_mfma_f32_32x32x16_f16 (the weights are converted bf16 -> fp16 once per workgroup, exactly).  16 blend
//     instructions per (unit, k-step) against 30 (pre-paired bf16 image + v_dot2_f32_bf16) and 44 (v_perm pairing + dot2), both
//     measured in earlier forms of this kernel.  The blend weights and S round to fp16 (2^-11; the bf16 kernels: 2^-8);
//   * lane = (pixel, channel half): the blended fp16 pairs of a lane ARE the B operand of the contraction MFMA (lane = pixel
//     column, 8 consecutive channels per k-step) — nothing is transposed, nothing goes back through LDS;
//   * the geometry of a (pixel, tap) — two packed weight pairs and the LDS positions of its four corner pixels — is a 16-byte
//     RECORD built once per tile from the fp32 offsets / mask logits (floor, fractions, sigmoid, image-border zeroing happen there,
//     once, instead of once per lane half and tap): a unit costs a lane one 16-byte LDS read of geometry;
//   * the weights are STATIONARY IN REGISTERS: a workgroup is THREE waves that share one 4x8 pixel group and split the 18 (tap,
//     channel half) units 6 : 6 : 6, each holding its 6 x 16 registers of W fragments for the whole launch (persistent
//     workgroups), and meet once per tile through LDS.  No weight traffic in the loop, no barrier per tap: a tile is four
//     barriers of three waves;
//   * <= 170 registers -> three waves per SIMD, FOUR independent workgroups per CU (36 KB of LDS each): while one loads and
//     converts its halo, builds records or stores its tile, the others blend.
// Halo image: 11 rows x 15 columns around the 4x8 tile (offsets up to |d| < 2 px stay inside), 16 pixel slots per row (slot =
// column ^ 8 on odd rows), 128 B per pixel with its eight 16-byte chunks XOR-swizzled by (slot number >> 1) & 7: the 16 lanes of a
// ds_read_b128 group touch 16 different slot numbers mod 16 and tile all 64 banks.
// Samples whose footprint leaves the halo: their four corner pixels are copied from global memory into spare pixel slots behind
// the image (7 such samples per pass) and the record points there — the hot loop knows nothing about them; a tile with more
// runs the unit loop again for the next 7 with every other record pointed at an all-zero pixel (any offset field is handled; a
// field with many far samples pays for it).
#include "conv_common.h"
#include <stdlib.h>

#define GQ_TH 4                                  // tile = one 4x8 pixel group
#define GQ_TW 8
#define GQ_MG 3                                  // halo margin above / left of the tile (4 below / right)
#define GQ_ROWS (GQ_TH + 7)                      // 11 halo rows
#define GQ_COLS (GQ_TW + 7)                      // 15 halo columns
#define GQ_P 16                                  // pixel slots per halo row
#define GQ_NPIX 32
#define GQ_NREC (GQ_NPIX * 9)                    // 288 records x 16 B
#define GQ_NT 192                                // threads: three waves
#define GQ_IMG_BYTES ((GQ_ROWS + 2) * GQ_P * 128)    // 26 624: the image (176 slots) + 32 spare slots (zero pixel, far corners)
#define GQ_ZERO_SLOT (GQ_ROWS * GQ_P)            // slot 176: all-zero pixel
#define GQ_FAR_SLOT0 (GQ_ROWS * GQ_P + 4)        // slots 180 .. 207: far corners (4 per sample)
#define GQ_FAR_PER_PASS 7
#define GQ_REC_OFS GQ_IMG_BYTES
#define GQ_OMS_OFS (GQ_REC_OFS + GQ_NREC * 16)   // raw offsets / mask logits [pixel][28 floats]
#define GQ_HW_OFS (GQ_OMS_OFS + GQ_NPIX * 112)   // corner 00 of every record in image coordinates (int16 pair; read by the far path)
#define GQ_BIAS_OFS (GQ_HW_OFS + GQ_NREC * 4)    // 64 floats
#define GQ_BMP_OFS (GQ_BIAS_OFS + 256)           // 9 dwords: bitmap of the records whose footprint left the halo; dword 9: their count
#define GQ_SMEM (GQ_BMP_OFS + 48)                // 36 272

#ifdef GS_PROBE   // development build only (tools/gs_probe.py): cycle stamps of the three waves of the first workgroups, tiles 0 .. 3
__device__ unsigned long long gs_ts[64 * 3 * 4 * 16];
#define GS_STAMP(k) do { if (lane == 0 && blockIdx.x < 64 && j < 4) gs_ts[((blockIdx.x * 3 + role) * 4 + j) * 16 + (k)] = clock64(); } while (0)
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

// pixel slot of halo pixel (row r, column c)
__device__ static inline int gq_slot(int r, int c) { return r * GQ_P + (c ^ ((r & 1) << 3)); }

// ---- geometry records of a tile: built from the staged offsets by all 192 threads ----
// record (tap, pixel) = { P0 = fp16 pair (w00 m, w01 m), P1 = (w10 m, w11 m), position codes of the corner pixels 00 | 01 << 16 and
// 10 | 11 << 16 }.  Corners outside the image have weight 0.  A footprint outside the halo is noted in the bitmap and pointed at the
// zero pixel until gq_place_far gives it slots.
__device__ static inline void gq_build_records(unsigned char* smem, int tid, int ty0, int tx0, int H, int W) {
    const float* const oms = reinterpret_cast<const float*>(smem + GQ_OMS_OFS);
    u32x4g* const rec = reinterpret_cast<u32x4g*>(smem + GQ_REC_OFS);
    uint32_t* const hwp = reinterpret_cast<uint32_t*>(smem + GQ_HW_OFS);
    uint32_t* const bmp = reinterpret_cast<uint32_t*>(smem + GQ_BMP_OFS);
    constexpr uint32_t ZC = ((uint32_t)GQ_ZERO_SLOT << 7) | ((((uint32_t)GQ_ZERO_SLOT >> 1) & 7u) << 4);
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + GQ_NT * i;
        if (idx < GQ_NREC) {
            const int tap = idx >> 5, p = idx & 31;
            const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;           // tap / 3, tap % 3 for tap < 9
            const int y = ty0 + (p >> 3), x = tx0 + (p & 7);
            const bool live = y < H && x < W;
            const float* o = oms + p * 28;
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
            const bool inwin = (unsigned)wr <= (unsigned)(GQ_ROWS - 2) && (unsigned)wc <= (unsigned)(GQ_COLS - 2);
            uint32_t c01 = ZC | (ZC << 16), c23 = c01;
            if (nz && inwin) {
                c01 = gq_code(gq_slot(wr, wc)) | (gq_code(gq_slot(wr, wc + 1)) << 16);
                c23 = gq_code(gq_slot(wr + 1, wc)) | (gq_code(gq_slot(wr + 1, wc + 1)) << 16);
            }
            if (nz && !inwin) { atomicOr(bmp + (idx >> 5), 1u << (idx & 31)); atomicAdd(bmp + 9, 1u); }
            u32x4g r;
            r[0] = nz ? P0 : 0u; r[1] = nz ? P1 : 0u; r[2] = c01; r[3] = c23;
            rec[idx] = r;
            hwp[idx] = ((uint32_t)h0 & 0xffffu) | ((uint32_t)w0 << 16);
        }
    }
}

// ---- pass k over a tile with far footprints (rare): records of far sample number 7 k .. 7 k + 6 get four spare pixel slots filled
//      from global memory, every other far record — and, from the second pass on, every near record (they were accumulated in pass
//      0) — points at the zero pixel.  Each thread serves its own records. ----
__device__ static inline void gq_place_far(unsigned char* smem, const bf16_t* X, int tid, int k, int H, int W) {
    u32x4g* const rec = reinterpret_cast<u32x4g*>(smem + GQ_REC_OFS);
    const uint32_t* const hwp = reinterpret_cast<const uint32_t*>(smem + GQ_HW_OFS);
    const uint32_t* const bmp = reinterpret_cast<const uint32_t*>(smem + GQ_BMP_OFS);
    const unsigned char* const xb = reinterpret_cast<const unsigned char*>(X);
    constexpr uint32_t ZC = ((uint32_t)GQ_ZERO_SLOT << 7) | ((((uint32_t)GQ_ZERO_SLOT >> 1) & 7u) << 4);
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + GQ_NT * i;
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
                *reinterpret_cast<u32x4g*>(smem + (code ^ (uint32_t)(q << 4))) = gq_bf2h4(*reinterpret_cast<const u32x4g*>(xb + (uint32_t)((hc * W + wcc) * 128 + q * 16)));
        }
        cp[0] = gq_code(s0) | (gq_code(s0 + 1) << 16);
        cp[1] = gq_code(s0 + 2) | (gq_code(s0 + 3) << 16);
    }
}

template <bool STATS>
__global__ __launch_bounds__(GQ_NT, 3) void dcn_fwd_gs_kernel(const GsFwdGeom g) {
    CN_MAIN_PRIO_SET();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32x4g* const REC = reinterpret_cast<const u32x4g*>(smem + GQ_REC_OFS);
    uint32_t* const BMP = reinterpret_cast<uint32_t*>(smem + GQ_BMP_OFS);
    unsigned char* const IMG = smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int role = __builtin_amdgcn_readfirstlane(tid >> 6);       // the three waves split the 18 (tap, channel half) units 6 : 6 : 6
    const int nl = lane & 31, hh = lane >> 5;

    // zero pixel + far slots, bias
    for (int i = tid; i < 32 * 32; i += GQ_NT) reinterpret_cast<uint32_t*>(smem + GQ_ZERO_SLOT * 128)[i] = 0u;
    if (tid < 64) reinterpret_cast<float*>(smem + GQ_BIAS_OFS)[tid] = g.bias[tid];

    // ---- this wave's six W fragment sets, for the whole launch: unit u <-> (tap, half) = ((6 role + u) >> 1, (6 role + u) & 1);
    //      fragment (s, cb): A operand, lane = output channel 32 cb + nl, 8 input channels 32 half + 16 s + 8 hh .. +7 of the tap
    //      (wp = mode-1 pack [Co][tap * 64 + ci], bf16 -> fp16: exact, the weights are far inside the fp16 range) ----
    u32x4g wf[6][2][2];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int gu = 6 * role + u, tap = gu >> 1, hf = gu & 1;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                wf[u][s][cb] = gq_bf2h4(*reinterpret_cast<const u32x4g*>(g.wp + (int64_t)(32 * cb + nl) * 576 + tap * 64 + 32 * hf + 16 * s + 8 * hh));
    }

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

        // (the thread index is laundered once per tile: the compiler otherwise hoists per-thread index arithmetic out of the tile loop,
        //  finds no registers for it next to the W sets, and reloads it from scratch)
        int tidv = tid;
        asm volatile("" : "+v"(tidv));
        GS_STAMP(0);
        // ---- the tile's global reads: 32 pixels x 7 granules of offsets / mask logits (2 per thread) and 11 x 15 halo pixels x 8
        //      chunks = 1 320 items (7 per thread, in two batches: the W sets leave room for four in flight) ----
        const unsigned char* const xb = reinterpret_cast<const unsigned char*>(X);
        auto halo_load = [&](int k) -> u32x4g {
            const int i = min(tidv + GQ_NT * k, GQ_ROWS * GQ_COLS * 8 - 1);
            const int pc = i >> 3, r = pc / GQ_COLS, c = pc - GQ_COLS * r;
            const int hy = ty0 - GQ_MG + r, hx = tx0 - GQ_MG + c;
            const bool ok = (unsigned)hy < (unsigned)g.H && (unsigned)hx < (unsigned)g.W;
            return *reinterpret_cast<const u32x4g*>(xb + (ok ? (uint32_t)((hy * g.W + hx) * 128 + (i & 7) * 16) : 0u));
        };
        auto halo_store = [&](int k, const u32x4g& v) {
            const int i = tidv + GQ_NT * k;
            const int ic = min(i, GQ_ROWS * GQ_COLS * 8 - 1);
            const int pc = ic >> 3, r = pc / GQ_COLS, c = pc - GQ_COLS * r;
            const int hy = ty0 - GQ_MG + r, hx = tx0 - GQ_MG + c;
            const bool ok = (unsigned)hy < (unsigned)g.H && (unsigned)hx < (unsigned)g.W;
            const u32x4g z = {0u, 0u, 0u, 0u};
            if (i < GQ_ROWS * GQ_COLS * 8) *reinterpret_cast<u32x4g*>(smem + (gq_code(gq_slot(r, c)) ^ (uint32_t)((ic & 7) << 4))) = ok ? gq_bf2h4(v) : z;
        };
        float4 ov[2];
        {
            const float* const ob = g.om + img * 32;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = min(tidv + GQ_NT * k, GQ_NPIX * 7 - 1);
                const int p = i / 7, part = i - 7 * p;
                const int y = ty0 + (p >> 3), x = tx0 + (p & 7);
                const bool ok = y < g.H && x < g.W;
                ov[k] = *reinterpret_cast<const float4*>(ob + (ok ? (y * g.W + x) * 32 + part * 4 : 0));
            }
        }
        u32x4g ha[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ha[k] = halo_load(k);
        if (tidv < 10) BMP[tidv] = 0u;
        __syncthreads();              // [S1] everybody is done with the previous tile's LDS
        GS_STAMP(1);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tidv + GQ_NT * k;
            if (i < GQ_NPIX * 7) *reinterpret_cast<float4*>(smem + GQ_OMS_OFS + i * 16) = ov[k];
        }
        {
            u32x4g hb[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) hb[k] = halo_load(4 + k);
#pragma unroll
            for (int k = 0; k < 4; ++k) halo_store(k, ha[k]);
#pragma unroll
            for (int k = 0; k < 3; ++k) halo_store(4 + k, hb[k]);
        }
        GS_STAMP(2);
        __syncthreads();              // [S1b] staged offsets visible
        gq_build_records(smem, tidv, ty0, tx0, g.H, g.W);
        GS_STAMP(3);
        __syncthreads();              // [S2]
        GS_STAMP(4);
        const int nfar = __builtin_amdgcn_readfirstlane((int)BMP[9]);
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
                gq_place_far(smem, X, tid, pass, g.H, g.W);
                __syncthreads();
            }
            // 12 (unit, k-step) steps: the four 16-byte corner reads of step i+1 go out as soon as step i's blend has consumed its own
            // (their latency overlaps this step's MFMAs and the other waves of the SIMD); sched_barriers keep the order
            uint32_t P0, P1, c0, c1, c2, c3;
            {
                const u32x4g rec = REC[((6 * role) >> 1) * GQ_NPIX + nl];
                const uint32_t xq = (uint32_t)(hh << 4);            // (6 role) & 1 == 0
                P0 = rec[0]; P1 = rec[1];
                c0 = (rec[2] & 0xffffu) ^ xq; c1 = (rec[2] >> 16) ^ xq; c2 = (rec[3] & 0xffffu) ^ xq; c3 = (rec[3] >> 16) ^ xq;
            }
            u32x4g q0 = *reinterpret_cast<const u32x4g*>(IMG + c0), q1 = *reinterpret_cast<const u32x4g*>(IMG + c1);
            u32x4g q2 = *reinterpret_cast<const u32x4g*>(IMG + c2), q3 = *reinterpret_cast<const u32x4g*>(IMG + c3);
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                u32x4g nrec = {0u, 0u, 0u, 0u};
                if (u < 5) nrec = REC[((6 * role + u + 1) >> 1) * GQ_NPIX + nl];
                // ---- k-step 0 ----
                {
                    const u32x4g sb = gq_blend(q0, q1, q2, q3, P0, P1);
                    __builtin_amdgcn_sched_barrier(0);
                    // k-step 1's reads (logical chunk + 2: position ^ 32)
                    q0 = *reinterpret_cast<const u32x4g*>(IMG + (c0 ^ 32u)); q1 = *reinterpret_cast<const u32x4g*>(IMG + (c1 ^ 32u));
                    q2 = *reinterpret_cast<const u32x4g*>(IMG + (c2 ^ 32u)); q3 = *reinterpret_cast<const u32x4g*>(IMG + (c3 ^ 32u));
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8g, wf[u][0][cb]), __builtin_bit_cast(f16x8g, sb), acc[cb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- k-step 1 ----
                {
                    const u32x4g sb = gq_blend(q0, q1, q2, q3, P0, P1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (u < 5) {        // the next unit's record is decoded and its k-step 0 reads go out
                        const uint32_t xq = (uint32_t)((((u + 1) & 1) * 4 + hh) << 4);
                        P0 = nrec[0]; P1 = nrec[1];
                        c0 = (nrec[2] & 0xffffu) ^ xq; c1 = (nrec[2] >> 16) ^ xq; c2 = (nrec[3] & 0xffffu) ^ xq; c3 = (nrec[3] >> 16) ^ xq;
                        q0 = *reinterpret_cast<const u32x4g*>(IMG + c0); q1 = *reinterpret_cast<const u32x4g*>(IMG + c1);
                        q2 = *reinterpret_cast<const u32x4g*>(IMG + c2); q3 = *reinterpret_cast<const u32x4g*>(IMG + c3);
                    }
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8g, wf[u][1][cb]), __builtin_bit_cast(f16x8g, sb), acc[cb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        GS_STAMP(5);
        __syncthreads();              // [S3] every wave is done with the image and the records: they become the exchange / staging area
        GS_STAMP(6);
        // ---- the three waves meet: roles 1 and 2 park their partial sums (role 2: + the bias) as fp32 [32 px][68] in the image ----
        // (lane indices laundered like the thread index above: the addresses below would otherwise be hoisted and spilled)
        int nle = nl, hhe = hh;
        asm volatile("" : "+v"(nle), "+v"(hhe));
        unsigned char* const XC1 = IMG + nle * (68 * 4) + hhe * 16;
        if (role) {
            unsigned char* const XCb = XC1 + (role - 1) * (32 * 68 * 4);
            const unsigned char* const Bl = smem + GQ_BIAS_OFS + hhe * 16;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // (lane = pixel, register v of block cb = channel 32 cb + 8 (v >> 2) + 4 hh + (v & 3))
                    float4 bv = *reinterpret_cast<const float4*>(Bl + (32 * cb + 8 * q) * 4);
                    if (role == 1) bv = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(XCb + (32 * cb + 8 * q) * 4) =
                        make_float4(acc[cb][4 * q] + bv.x, acc[cb][4 * q + 1] + bv.y, acc[cb][4 * q + 2] + bv.z, acc[cb][4 * q + 3] + bv.w);
                }
        }
        __syncthreads();              // [S4]
        GS_STAMP(7);
        if (!role) {
            unsigned char* const Y = smem + GQ_REC_OFS;                               // [32 px][72] bf16 (the records are dead)
            unsigned char* const Yl = Y + nle * 144 + hhe * 8;
            const float floor_ = g.relu ? 0.f : -__builtin_inff();
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 o1 = *reinterpret_cast<const float4*>(XC1 + (32 * cb + 8 * q) * 4);
                    const float4 o2 = *reinterpret_cast<const float4*>(XC1 + (32 * 68 * 4) + (32 * cb + 8 * q) * 4);
                    float v[4] = {acc[cb][4 * q] + o1.x + o2.x, acc[cb][4 * q + 1] + o1.y + o2.y, acc[cb][4 * q + 2] + o1.z + o2.z,
                                  acc[cb][4 * q + 3] + o1.w + o2.w};
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
            const int ox = tx0 + p0;                                                      // pass i: pixel (row i, column p0) of the group
            const unsigned char* const Yr = Y + p0 * 144 + ch * 16;
            bf16_t* const yo = g.y + (img + (int64_t)ty0 * g.W + ox) * 64 + ch * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4g o = *reinterpret_cast<const u32x4g*>(Yr + i * (8 * 144));
                if (ty0 + i < g.H && ox < g.W) {
                    *reinterpret_cast<u32x4g*>(yo + (int64_t)i * g.W * 64) = o;
                    if (STATS) { const uint32_t w4[4] = {o[0], o[1], o[2], o[3]}; bn_stat_add(s0, s1, w4); }
                }
            }
        }
        GS_STAMP(8);
    }
    // BN statistics of everything this workgroup stored (sink protocol of bn.hip); the other two waves contribute zeros
    if (STATS) {
        __syncthreads();
        bn_stats_flush<8, GQ_NT>(s0, s1, reinterpret_cast<float*>(smem), g.bn_part, g.bn_slots, 64, 0, 64, blockIdx.x, tid);
    }
}

bool dcn_fwd_gs_shape_ok(int Ci, int x_ld, int Co, int y_ld, int om_ld) {
    // opt-in (round 5): correct, but it does not beat the blend-matrix kernel (DESIGN / docs/NEGATIVE_RESULTS.md)
    static const bool enabled = getenv("CN_ENABLE_DCN_FWD_GS") != nullptr;
    return enabled && Ci == 64 && x_ld == 64 && om_ld == 32 && Co == 64 && y_ld == 64;
}

static int gs_grid(int ntiles) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    int G = 4 * cus;                                     // four resident workgroups per CU
    if (ntiles < G) G = ntiles >= 8 ? (ntiles / 8) * 8 : ntiles;
    return G;
}

// returns false when the shape is not handled here (caller falls back to the blend-matrix / gather kernels)
bool dcn_fwd_gs_launch(const void* x, const float* om, const void* wp, const float* bias, void* y, int N, int H, int W, int Ci, int x_ld,
                       int Co, int y_ld, int om_ld, int ktot, int relu, float* bn_part, int bn_slots, int* bn_taken, hipStream_t st) {
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
        mark_taken(bn_taken);
        hipLaunchKernelGGL(dcn_fwd_gs_kernel<true>, dim3(gs_grid(g.ntiles)), dim3(GQ_NT), GQ_SMEM, st, g);
    } else {
        hipLaunchKernelGGL(dcn_fwd_gs_kernel<false>, dim3(gs_grid(g.ntiles)), dim3(GQ_NT), GQ_SMEM, st, g);
    }
    return true;
}
