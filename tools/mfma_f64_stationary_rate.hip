// Issue rate of v_mfma_f64_16x16x4_f64 when ONE operand is a large register-resident set (the generator-stationary sweep's shape):
// every wave holds NA fragments of 16 bytes (2 NA doubles: up to 320 registers) loaded once from memory and streams MFMAs over
// them against a single B register, NACC independent accumulators, one wave per SIMD (launch_bounds(256, 1): 512 registers).
// Prints cycles per MFMA (s_memtime) and where the compiler put the operands is read off the ISA (-save-temps).
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_stationary_rate.hip -o tools/mfma_f64_stationary_rate [-mllvm -amdgpu-mfma-vgpr-form=1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

template <int NA, int NACC>
__global__ void __launch_bounds__(256, 1) k_rate(const double* src, double* out, unsigned long long* cyc, int iters) {
    d2 af[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) af[i] = *reinterpret_cast<const d2*>(src + (size_t)(i * 256 + threadIdx.x) * 2);
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double b = 0.5 - threadIdx.x * 1e-4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            acc[(2 * i) % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, af[i].x, acc[(2 * i) % NACC], 0, 0, 0);
            acc[(2 * i + 1) % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, af[i].y, acc[(2 * i + 1) % NACC], 0, 0, 0);
        }
        b += 1e-9;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NA, int NACC>
void run(int iters) {
    const int nb = 256;
    double *src, *out; unsigned long long* cyc;
    hipMalloc(&src, sizeof(double) * 2 * 256 * NA); hipMemset(src, 0, sizeof(double) * 2 * 256 * NA);
    hipMalloc(&out, sizeof(double) * nb * 256); hipMalloc(&cyc, 8 * nb);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_rate<NA, NACC>), dim3(nb), dim3(256), 0, 0, src, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> hc(nb);
    hipMemcpy(hc.data(), cyc, 8 * nb, hipMemcpyDeviceToHost);
    printf("NA=%3d (%3d registers stationary) NACC=%d: %.1f cycles/MFMA\n", NA, 4 * NA, NACC, (double)hc[nb / 2] / ((double)iters * 2 * NA));
    hipFree(src); hipFree(out); hipFree(cyc);
}

int main() {
    run<16, 4>(400);
    run<32, 4>(200);
    run<48, 4>(200);
    run<64, 4>(100);
    run<80, 4>(100);
    run<80, 2>(100);
    run<80, 8>(100);
    return 0;
}
