// torch.optim.Adam (defaults; centernet.py:94-95) as one fused pass over flat fp32 buffers: 16 B/param read
// (p, g, m, v) + 12 B written — HBM-bound, float4 accesses.
#include "common.h"

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2, float gscale, const float* __restrict__ hyper) {
    if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2 = hyper[2]; }   // graph replays read the per-step values from HBM
    const float step = lr / bc1;
    const float rsb2 = 1.f / sqrtf(bc2);
    const int64_t nv = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = G[j] * gscale;
            M[j] = b1 * M[j] + (1.f - b1) * gr;
            V[j] = b2 * V[j] + (1.f - b2) * gr * gr;
            P[j] -= step * M[j] / (sqrtf(V[j]) * rsb2 + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < n - nv * 4) {
        const int64_t i = nv * 4 + threadIdx.x;
        const float gr = g[i] * gscale;
        m[i] = b1 * m[i] + (1.f - b1) * gr;
        v[i] = b2 * v[i] + (1.f - b2) * gr * gr;
        p[i] -= step * m[i] / (sqrtf(v[i]) * rsb2 + eps);
    }
}

extern "C" int cn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                            float bc1, float bc2, float grad_scale, const float* hyper, void* stream) {
    CN_CHECK_ARG(p && g && m && v && n > 0, "cn_adam_step: bad args");
    CN_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "cn_adam_step: buffers must be 16-byte aligned");
    int64_t gr = (n / 4 + 255) / 256;
    int grid = (int)(gr > 8192 ? 8192 : (gr < 1 ? 1 : gr));
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, b1, b2, eps, bc1, bc2, grad_scale, hyper);
    CN_LAUNCH_CHECK("cn_adam_step");
    return CN_OK;
}

// Step counter and bias corrections live ON THE DEVICE: hyper = {lr, 1-b1^t, 1-b2^t, t (int bits)}.  A captured graph
// advances them itself, so a host that runs many replays ahead cannot race an asynchronous upload of per-step values.
__global__ void adam_advance_kernel(float* __restrict__ hyper, float b1, float b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int t = __float_as_int(hyper[3]) + 1;
    hyper[3] = __int_as_float(t);
    hyper[1] = (float)(1.0 - pow((double)b1, (double)t));
    hyper[2] = (float)(1.0 - pow((double)b2, (double)t));
}

extern "C" int cn_adam_advance(float* hyper, float b1, float b2, void* stream) {
    CN_CHECK_ARG(hyper, "cn_adam_advance: null");
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, hyper, b1, b2);
    CN_LAUNCH_CHECK("cn_adam_advance");
    return CN_OK;
}
