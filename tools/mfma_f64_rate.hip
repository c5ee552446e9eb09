// Measures the issue rate of v_mfma_f64_16x16x4_f64 on the device: every SIMD runs W waves of
// back-to-back MFMAs on NACC independent accumulators.  Prints cycles per MFMA per SIMD (s_memtime),
// the in-kernel clock (s_memtime / s_memrealtime) and the chip-wide TFLOP/s.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_rate.hip -o tools/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, bool RANDOM>
__global__ void __launch_bounds__(256, 2) k_rate(double* out, unsigned long long* cyc, unsigned long long* rt, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-4;
    double ar[8], br[8];
    if (RANDOM) {
        unsigned long long h = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1 + 977 * blockIdx.x);
        for (int i = 0; i < 8; ++i) {
            h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
            ar[i] = (double)(long long)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
            h *= 0x94D049BB133111EBull; h ^= h >> 31;
            br[i] = (double)(long long)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
        }
    }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = RANDOM ? __builtin_amdgcn_mfma_f64_16x16x4f64(ar[i & 7], br[(i >> 1) & 7], acc[i], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <int NACC, bool RANDOM = false>
void run(int blocks_per_cu, int iters) {
    int ncu = 256;
    int nb = ncu * blocks_per_cu;
    double* out; unsigned long long *cyc, *rt;
    hipMalloc(&out, sizeof(double) * nb * 256);
    hipMalloc(&cyc, 8 * nb); hipMalloc(&rt, 8 * nb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_rate<NACC, RANDOM>), dim3(nb), dim3(256), 0, 0, out, cyc, rt, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<NACC, RANDOM>), dim3(nb), dim3(256), 0, 0, out, cyc, rt, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hc(nb), hr(nb);
    hipMemcpy(hc.data(), cyc, 8 * nb, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), rt, 8 * nb, hipMemcpyDeviceToHost);
    double mfma_per_wave = (double)iters * NACC;
    double flops = mfma_per_wave * 2048.0 * 4 * nb;
    // waves per SIMD = blocks_per_cu (each block has one wave per SIMD)
    double cyc_per_mfma_simd = (double)hc[nb / 2] / (mfma_per_wave * blocks_per_cu);
    double clk_ghz = (double)hc[nb / 2] / (double)hr[nb / 2] * 0.1;
    printf("%s NACC=%d waves/SIMD=%d: %.1f cycles/MFMA/SIMD (if all co-resident), clock %.2f GHz, wall %.3f ms -> %.1f TFLOP/s\n", RANDOM ? "random  " : "constant", NACC, blocks_per_cu,
           cyc_per_mfma_simd, clk_ghz, ms, flops / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc); hipFree(rt);
}

int main() {
    run<1>(1, 20000);
    run<4>(1, 5000);
    run<16>(1, 2000);
    run<16>(2, 2000);
    run<8>(4, 2000);
    // sustained: ~100 ms of back-to-back MFMAs per launch, to see the clock the chip holds under load
    for (int rep = 0; rep < 2; ++rep) run<16>(2, 100000);
    for (int rep = 0; rep < 3; ++rep) run<16, true>(2, 100000);
    return 0;
}
