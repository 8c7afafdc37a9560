// Backward of a task head whose loss reads its output only at a few pixels (RegL1Loss / RegWeightedL1Loss, utils/losses.py:53-63 /
// 81-91: `_transpose_and_gather_feat(output, ind)`): the gradient of the head's output map is zero except at the <= max_objects
// object-centre pixels of each image (128 of 16 384 at 512x512: 0.8 %).  The reference — and the dense path here — still pushes
// that map through the full backward of conv1x1 <- ReLU <- conv3x3 (heads.py:4-25): 537 MB of hidden-layer gradient written and
// read twice per head and step.  Everything downstream of a zero output gradient is zero, so the same sums can be taken over the
// R = B * M rows that can be non-zero:
//   g_r      = dOut[b, :, ind[b, m]]                                   (rows with a repeated index: first occurrence only —
//                                                                        the dense map already holds their SUM at that pixel)
//   dH_r     = (H[b, p_r] > 0) * (W2^T g_r)                            hidden-layer gradient at the row's pixel
//   dW2, db2 = sum_r g_r (x) H_r,  sum_r g_r                           |  as two small GEMMs over the compact matrices
//   dW1, db1 = sum_r dH_r (x) patch3x3(x, p_r),  sum_r dH_r            |  (existing kernels, R "pixels" of a 1x1 convolution)
//   dx[p_r + d] += W1[:, :, d]^T dH_r  for the nine taps d             -> cn_scatter3x3_add
// cn_head_sparse_gather builds the compact operands; the GEMMs are the existing 1x1 entry points; nothing of size B*H*W*256 is
// touched in this backward.
#include "common.h"

template <typename T> __device__ static inline float to_f(T v);
template <> __device__ inline float to_f<float>(float v) { return v; }
template <> __device__ inline float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ static inline T from_f(float v);
template <> __device__ inline float from_f<float>(float v) { return v; }
template <> __device__ inline bf16_t from_f<bf16_t>(float v) { return f2bf(v); }

#define HS_MAXC 64

// one workgroup per row r = b * M + m
template <typename T>
__global__ __launch_bounds__(256) void head_sparse_gather_kernel(const T* __restrict__ h, const T* __restrict__ x,
                                                                 const int64_t* __restrict__ ind, const float* __restrict__ dout,
                                                                 const float* __restrict__ w2, T* __restrict__ hg, T* __restrict__ dhc,
                                                                 T* __restrict__ xg, T* __restrict__ gq, int M, int C, int H, int W,
                                                                 int Ch, int h_ld, int Ci, int x_ld, int Cq, int mode) {
    // mode 0: h is the dense hidden activation [B, H, W, h_ld]; everything is written.  mode 1: no hidden activation exists (the head's
    // forward was the one-launch cn_head2_fwd): only the input patches xg and the output-gradient rows gq.  mode 2: h holds the hidden ROWS
    // [R, h_ld] recomputed from xg; only the masked hidden gradient dhc is written.
    __shared__ float gs[HS_MAXC];
    __shared__ int dup;
    const int r = blockIdx.x, tid = threadIdx.x;
    const int b = r / M, m = r - b * M;
    const int64_t HW = (int64_t)H * W;
    int64_t pos = ind[(int64_t)b * M + m];
    pos = pos < 0 ? 0 : (pos >= HW ? HW - 1 : pos);
    if (tid == 0) dup = 0;
    __syncthreads();
    for (int j = tid; j < m; j += 256) {
        int64_t pj = ind[(int64_t)b * M + j];
        pj = pj < 0 ? 0 : (pj >= HW ? HW - 1 : pj);
        if (pj == pos) dup = 1;                 // benign race: every writer stores 1
    }
    __syncthreads();
    const bool active = dup == 0;
    if (tid < C) gs[tid] = active ? dout[((int64_t)b * C + tid) * HW + pos] : 0.f;
    __syncthreads();
    if (mode != 2)
        for (int c = tid; c < Cq; c += 256) gq[(int64_t)r * Cq + c] = from_f<T>(c < C ? gs[c] : 0.f);
    if (mode != 1) {
        const T* hrow = mode == 2 ? h + (int64_t)r * h_ld : h + ((int64_t)b * HW + pos) * h_ld;
        for (int t = tid; t < Ch; t += 256) {
            const T hv = hrow[t];
            float s = 0.f;
            for (int c = 0; c < C; ++c) s = fmaf(gs[c], w2[(int64_t)c * Ch + t], s);
            if (mode == 0) hg[(int64_t)r * Ch + t] = hv;
            dhc[(int64_t)r * Ch + t] = from_f<T>(to_f<T>(hv) > 0.f ? s : 0.f);
        }
    }
    if (mode == 2) return;
    const int py = (int)(pos / W), px = (int)(pos - (int64_t)py * W);
    const int K = 9 * Ci;
    for (int k = tid; k < K; k += 256) {
        const int ci = k / 9, tap = k - ci * 9;
        const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
        const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        xg[(int64_t)r * K + k] = in ? x[(((int64_t)b * H + yy) * W + xx) * x_ld + ci] : from_f<T>(0.f);
    }
}

__device__ static inline void atomic_add_elem(float* p, float v) { atomicAdd(p, v); }
// bf16 destination: compare-and-swap on the aligned 32-bit word that holds the element (collisions are rare: two objects whose
// 3x3 windows overlap); the sum is taken in fp32 and rounded once per contribution
__device__ static inline void atomic_add_elem(bf16_t* p, float v) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    unsigned int* w = reinterpret_cast<unsigned int*>(a & ~(uintptr_t)3);
    const bool hi = (a & 2) != 0;
    unsigned int old = *w, assumed;
    do {
        assumed = old;
        const bf16_t cur = (bf16_t)(hi ? (assumed >> 16) : (assumed & 0xffffu));
        const unsigned int nv = (unsigned int)f2bf(bf2f(cur) + v);
        const unsigned int repl = hi ? ((assumed & 0x0000ffffu) | (nv << 16)) : ((assumed & 0xffff0000u) | nv);
        old = atomicCAS(w, assumed, repl);
    } while (old != assumed);
}

// dx[b, p_r + d, ci] += dxc[r, ci * 9 + d]; one workgroup per row
template <typename T>
__global__ __launch_bounds__(256) void scatter3x3_add_kernel(const float* __restrict__ dxc, const int64_t* __restrict__ ind,
                                                             T* __restrict__ dx, int M, int H, int W, int Ci, int dx_ld) {
    const int r = blockIdx.x, tid = threadIdx.x;
    const int b = r / M;
    const int64_t HW = (int64_t)H * W;
    int64_t pos = ind[r];
    pos = pos < 0 ? 0 : (pos >= HW ? HW - 1 : pos);
    const int py = (int)(pos / W), px = (int)(pos - (int64_t)py * W);
    const int K = 9 * Ci;
    for (int k = tid; k < K; k += 256) {
        const float v = dxc[(int64_t)r * K + k];
        if (v == 0.f) continue;
        const int ci = k / 9, tap = k - ci * 9;
        const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
            atomic_add_elem(dx + (((int64_t)b * H + yy) * W + xx) * dx_ld + ci, v);
    }
}

static int head_sparse_gather(const void* h, const void* x, const int64_t* ind, const float* dout, const float* w2, void* hg, void* dhc,
                              void* xg, void* gq, int B, int M, int C, int H, int W, int Ch, int h_ld, int Ci, int x_ld, int Cq, int mode,
                              int dtype, void* stream) {
    CN_CHECK_ARG(x && ind && dout && w2 && B > 0 && M > 0 && C > 0 && H > 0 && W > 0 && Ch > 0 && Ci > 0 && h_ld >= Ch && x_ld >= Ci &&
                     Cq >= C && (unsigned)mode <= 2u, "cn_head_sparse_gather: bad args");
    CN_CHECK_ARG((mode == 1 || (h && dhc)) && (mode == 2 || (xg && gq)) && (mode != 0 || hg), "cn_head_sparse_gather: null operand for mode %d", mode);
    if (C > HS_MAXC) CN_UNSUPPORTED("cn_head_sparse_gather: C <= %d (got %d)", HS_MAXC, C);
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(head_sparse_gather_kernel<T>, dim3(B * M), dim3(256), 0, (hipStream_t)stream,
                                                  (const T*)h, (const T*)x, ind, dout, w2, (T*)hg, (T*)dhc, (T*)xg, (T*)gq, M, C, H, W, Ch,
                                                  h_ld, Ci, x_ld, Cq, mode));
    CN_LAUNCH_CHECK("cn_head_sparse_gather");
    return CN_OK;
}
extern "C" int cn_head_sparse_gather(const void* h, const void* x, const int64_t* ind, const float* dout, const float* w2, void* hg,
                                     void* dhc, void* xg, void* gq, int B, int M, int C, int H, int W, int Ch, int h_ld, int Ci,
                                     int x_ld, int Cq, int dtype, void* stream) {
    return head_sparse_gather(h, x, ind, dout, w2, hg, dhc, xg, gq, B, M, C, H, W, Ch, h_ld, Ci, x_ld, Cq, 0, dtype, stream);
}
// the two halves of the same gather for a head whose forward never stored its hidden activation (cn_head2_fwd): mode 1 = input patches
// xg + output-gradient rows gq (h, hg, dhc unused); mode 2 = h holds the hidden ROWS [B*M, h_ld] recomputed from xg, dhc is written
extern "C" int cn_head_sparse_gather_rows(const void* h, const void* x, const int64_t* ind, const float* dout, const float* w2, void* dhc,
                                          void* xg, void* gq, int B, int M, int C, int H, int W, int Ch, int h_ld, int Ci, int x_ld, int Cq,
                                          int mode, int dtype, void* stream) {
    CN_CHECK_ARG(mode == 1 || mode == 2, "cn_head_sparse_gather_rows: mode 1 or 2");
    return head_sparse_gather(h, x, ind, dout, w2, nullptr, dhc, xg, gq, B, M, C, H, W, Ch, h_ld, Ci, x_ld, Cq, mode, dtype, stream);
}

extern "C" int cn_scatter3x3_add(const float* dxc, const int64_t* ind, void* dx, int B, int M, int H, int W, int Ci, int dx_ld,
                                 int dtype, void* stream) {
    CN_CHECK_ARG(dxc && ind && dx && B > 0 && M > 0 && H > 0 && W > 0 && Ci > 0 && dx_ld >= Ci, "cn_scatter3x3_add: bad args");
    CN_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL(scatter3x3_add_kernel<T>, dim3(B * M), dim3(256), 0, (hipStream_t)stream, dxc, ind,
                                                  (T*)dx, M, H, W, Ci, dx_ld));
    CN_LAUNCH_CHECK("cn_scatter3x3_add");
    return CN_OK;
}
