// scripts/microbench_ld.hip -- does the lane stride of 16-byte loads matter for a streaming read?  Reads 2^25 16-byte elements (512 MiB)
// with 1024 x 256 threads and sums them (so nothing is optimised away):
//   P = 1: lane l reads element 64*k + l (one instruction covers 1 KiB contiguous: fully coalesced);
//   P = 4: lane l reads elements 4*g .. 4*g+3 back to back (64-byte lane stride: the four contiguous elements per lane of k_sc2_double's
//          folding pass: one instruction touches 64 different 64-byte segments, four instructions cover 4 KiB).
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/microbench_ld scripts/microbench_ld.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct alignas(16) E { uint64_t a, b; };
template <int P>
__global__ void __launch_bounds__(256) k(const E *__restrict__ s, size_t n, uint64_t *out) {
    uint64_t acc = 0;
    if (P == 1) {
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const E e = s[i]; acc += e.a ^ e.b; }
    } else {
        for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < n / 4; g += (size_t)gridDim.x * blockDim.x) {
            const E e0 = s[4 * g], e1 = s[4 * g + 1], e2 = s[4 * g + 2], e3 = s[4 * g + 3];
            acc += (e0.a ^ e0.b) + (e1.a ^ e1.b) + (e2.a ^ e2.b) + (e3.a ^ e3.b);
        }
    }
    out[blockIdx.x * (size_t)blockDim.x + threadIdx.x] = acc;
}
template <int P> void run(const char *name, const E *d, size_t n, uint64_t *o, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(256), 0, 0, d, n, o); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(256), 0, 0, d, n, o);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-28s blocks %5d  %.3f ms  %.0f GB/s\n", name, blocks, ms, n * 16.0 / ms / 1e6);
}
int main() {
    const size_t n = (size_t)1 << 25;
    E *d; uint64_t *o; hipMalloc(&d, n * 16); hipMalloc(&o, 8192 * 256 * 8); hipMemset(d, 1, n * 16);
    for (int blocks : {1024, 2048, 4096}) { run<1>("coalesced (1 element/lane)", d, n, o, blocks); run<4>("4 contiguous elements/lane", d, n, o, blocks); }
    return 0;
}
