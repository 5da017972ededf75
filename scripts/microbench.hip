// scripts/microbench.hip -- instruction-rate micro-benchmarks on gfx950 for the integer ops the
// prover kernels are made of (v_mad_u64_u32, v_alignbit_b32, v_add3_u32, v_xor_b32, 64-bit add).
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/microbench scripts/microbench.hip ; run on the GPU box.
// Each kernel runs ITER iterations of 8 independent chains per lane; 256 CUs x 8 blocks x 256 threads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define CHAINS 8
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
    uint32_t a[CHAINS], b[CHAINS]; uint64_t q[CHAINS];
    for (int c = 0; c < CHAINS; c++) { a[c] = seed + threadIdx.x * 7 + c; b[c] = seed * 3 + c + blockIdx.x; q[c] = ((uint64_t)a[c] << 32) | b[c]; }
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
            if (OP == 1) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(a[c]));
            if (OP == 2) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b[c]));
            if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[c]) : "v"(a[c]), "v"(b[c]) : "vcc");
            if (OP == 4) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[c]) : "v"(q[(c + 1) % CHAINS]));
            if (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
            if (OP == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
            if (OP == 7) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
        }
    }
    uint32_t r = 0;
    for (int c = 0; c < CHAINS; c++) r ^= a[c] ^ (uint32_t)q[c] ^ (uint32_t)(q[c] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> void run(const char *name, uint32_t *d) {
    const int blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, (uint32_t)r);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double laneops = (double)blocks * 256 * ITER * CHAINS;
    printf("%-16s %8.3f ms  %8.2f T lane-ops/s  (%.2f cycles per wave-instr per SIMD at 2.4 GHz)\n", name, ms, laneops / ms / 1e9,
           2.4e9 * (ms * 1e-3) / ((double)blocks * 4 /*waves*/ * ITER * CHAINS / (256.0 * 4)));
}
int main() {
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_xor_b32", d); run<5>("v_add_u32", d); run<1>("v_alignbit_b32", d); run<2>("v_add3_u32", d);
    run<4>("v_lshl_add_u64", d); run<3>("v_mad_u64_u32", d); run<6>("v_mul_lo_u32", d); run<7>("v_mul_hi_u32", d);
    return 0;
}
