// Do v_fma_f64 (VALU) and v_mfma_f64_16x16x4_f64 (matrix) share a pipe on gfx950?  Every wave runs NACC back-to-back MFMAs
// per iteration plus NF independent scalar-per-lane FMAs; if the wall time does not grow with NF the vector FMAs ride along.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_coissue.hip -o tools/mfma_valu_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, int NF>
__global__ void __launch_bounds__(256, 2) k_mix(double* out, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-4;
    double c[NF > 0 ? NF : 1];
    for (int j = 0; j < (NF > 0 ? NF : 1); ++j) c[j] = j * 0.25 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NF / NACC; ++j) {
                const int q = i * (NF / NACC) + j;
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c[q]) : "v"(a), "v"(b));
            }
        }
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int j = 0; j < (NF > 0 ? NF : 1); ++j) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int NF>
void run(int iters) {
    const int nb = 256 * 2;
    double* out;
    hipMalloc(&out, sizeof(double) * nb * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_mix<NACC, NF>), dim3(nb), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mix<NACC, NF>), dim3(nb), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * NACC * 2048.0 * 4 * nb, vf = (double)iters * NF * 128.0 * 4 * nb;
    printf("MFMA/iter %d, v_fma_f64/iter %d: %.3f ms  matrix %.1f TFLOP/s + vector %.1f TFLOP/s\n", NACC, NF, ms, mf / ms / 1e9, vf / ms / 1e9);
    hipFree(out);
}
template <int NACC>
__global__ void __launch_bounds__(256, 2) k_valu_only(double* out, int iters) {
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-4, c[16];
    for (int j = 0; j < 16; ++j) c[j] = j + threadIdx.x;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b));
    double s = 0;
    for (int j = 0; j < 16; ++j) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    run<8, 0>(20000);
    run<8, 8>(20000);
    run<8, 16>(20000);
    run<8, 32>(20000);
    run<8, 64>(20000);
    {   // vector FMAs alone
        const int nb = 512, iters = 200000;
        double* out; hipMalloc(&out, sizeof(double) * nb * 256);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k_valu_only<1>), dim3(nb), dim3(256), 0, 0, out, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_valu_only<1>), dim3(nb), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("v_fma_f64 alone: %.3f ms  %.1f TFLOP/s\n", ms, (double)iters * 16 * 128.0 * 4 * nb / ms / 1e9);
    }
    return 0;
}
