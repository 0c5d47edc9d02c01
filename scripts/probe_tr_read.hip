// Probe of ds_read_b64_tr_b16 (gfx950): every lane passes the address of "its own" 4 contiguous 16-bit elements; prints what
// each lane receives.  LDS holds element index == value.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t L[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) L[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    uint32_t addr;
    if (mode == 0) addr = (uint32_t)(((l & 15) * 4 + (l >> 4) * 64) * 2);                  // contiguous 4 x 16 blocks, one per 16 lanes
    else if (mode == 1) addr = (uint32_t)((((l & 15) >> 2) * 256 + (l & 3) * 4 + (l >> 4) * 16) * 2);   // rows 256 elements apart: lane i -> row i/4, cols 4(i%4)
    else addr = 0;                                                                            // uniform
    uint32_t base = (uint32_t)(uintptr_t)L;   // LDS address of L (generic->local low bits)
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr + base) : "memory");
    out[l * 4 + 0] = (uint16_t)(v.x & 0xFFFF);
    out[l * 4 + 1] = (uint16_t)(v.x >> 16);
    out[l * 4 + 2] = (uint16_t)(v.y & 0xFFFF);
    out[l * 4 + 3] = (uint16_t)(v.y >> 16);
}

int main() {
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * 2);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        std::vector<uint16_t> h(256);
        hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
