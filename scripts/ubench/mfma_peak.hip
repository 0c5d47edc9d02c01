// Sustained rate of v_mfma_f32_32x32x16_bf16 on this GPU, nothing else in the loop: what the "dense bf16 peak" of the convolution rooflines
// is worth in practice.  hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_peak.hip -o scripts/ubench/mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int ACC>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {
    f32x16 d[ACC];
    for (int i = 0; i < ACC; ++i)
        for (int r = 0; r < 16; ++r) d[i][r] = 0.f;
    typedef short v8s __attribute__((ext_vector_type(8)));
    v8s av = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, bv = {7, 6, 5, 4, 3, 2, 1, (short)blockIdx.x};
    bf16x8 a = __builtin_bit_cast(bf16x8, av), b = __builtin_bit_cast(bf16x8, bv);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) d[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < ACC; ++i)
        for (int r = 0; r < 16; ++r) s += d[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ACC>
void run(const char* name, int wgs_per_cu, int iters) {
    float* out;
    const int wgs = 256 * wgs_per_cu;
    hipMalloc(&out, (size_t)wgs * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma<ACC>, dim3(wgs), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    float best = 1e9f, ms_long = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma<ACC>, dim3(wgs), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        ms_long = ms;
    }
    const double flops = (double)wgs * 4 /*waves*/ * iters * ACC * 2.0 * 32 * 32 * 16;
    printf("%-44s %d wg/CU x 4 waves, %d acc tiles, %d iters: %.3f ms (last %.3f) -> %.0f TFLOP/s = %.2f of 2500\n", name, wgs_per_cu, ACC, iters, best,
           ms_long, flops / best / 1e9, flops / best / 1e9 / 2500.0);
    hipFree(out);
}

int main() {
    // short kernels (~0.3 ms: the length of a convolution launch) and long ones (~30 ms: sustained), one and two waves per SIMD
    run<8>("short, 1 wave/SIMD", 1, 2000);
    run<8>("short, 2 waves/SIMD", 2, 1000);
    run<16>("short, 1 wave/SIMD, 16 independent tiles", 1, 1000);
    run<8>("long, 1 wave/SIMD", 1, 200000);
    run<8>("long, 2 waves/SIMD", 2, 100000);
    run<4>("long, 2 waves/SIMD, 4 tiles (dependent every 4)", 2, 200000);
    return 0;
}
