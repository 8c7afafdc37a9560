// micro-probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS u16[i] = i; every lane passes its own byte address.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const int* addr, uint16_t* out) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    uint32_t a = (uint32_t)(uintptr_t)lds + (uint32_t)addr[threadIdx.x];
    uint32_t lo, hi;
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    lo = (uint32_t)v; hi = (uint32_t)(v >> 32);
    out[threadIdx.x * 4 + 0] = lo & 0xffff; out[threadIdx.x * 4 + 1] = lo >> 16;
    out[threadIdx.x * 4 + 2] = hi & 0xffff; out[threadIdx.x * 4 + 3] = hi >> 16;
}
int main() {
    int h_addr[64]; uint16_t h_out[256];
    int* d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int pat = 0; pat < 3; ++pat) {
        for (int l = 0; l < 64; ++l) {
            if (pat == 0) h_addr[l] = l * 8;                           // lane l -> elements 4l..4l+3
            if (pat == 1) h_addr[l] = (l & 15) * 256 + (l >> 4) * 8;   // row (l&15) of a [16][128] u16 tile, 4 elems at col 4*(l>>4)
            if (pat == 2) h_addr[l] = (l & 3) * 8 + ((l >> 2) & 3) * 256 + (l >> 4) * 1024;
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) printf("lane %2d addr_elem %4d -> %4d %4d %4d %4d\n", l, h_addr[l] / 2, h_out[4*l], h_out[4*l+1], h_out[4*l+2], h_out[4*l+3]);
    }
    return 0;
}
