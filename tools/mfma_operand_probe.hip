// What does feeding the FP64 MFMA from LDS cost?  Every wave runs MT x NT v_mfma_f64_16x16x4_f64 per k-step on fragments it
// reads from LDS (conflict-free images, random data, the access pattern of dto_gemm.hip.h), nothing else: no global traffic,
// no barrier.  Prints TFLOP/s and the in-kernel clock per (MT, NT, waves per SIMD): loads per MFMA = (MT + NT) / (MT NT).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_operand_probe.hip -o tools/mfma_operand_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MT, int NT, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_probe(const double* __restrict__ src, double* out, unsigned long long* cyc,
                                                      unsigned long long* rt, int iters) {
    // per wave: A image [16 k][16 MT rows + 16] and B image [16 NT cols][18]
    constexpr int LDA = 16 * MT + 16, LDB = 18;
    constexpr int PER_WAVE = 16 * LDA + 16 * NT * LDB;
    __shared__ double lds[PER_WAVE * WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lq = lane >> 4;
    double* As = lds + wave * PER_WAVE;
    double* Bs = As + 16 * LDA;
    for (int i = lane; i < PER_WAVE; i += 64) As[i] = src[(blockIdx.x * 131 + wave * 17 + i) & 0xFFFF];
    __syncthreads();
    d4 acc[MT][NT];
    for (int i = 0; i < MT; ++i)
        for (int j = 0; j < NT; ++j) acc[i][j] = d4{0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            double af[MT], bf[NT];
#pragma unroll
            for (int ti = 0; ti < MT; ++ti) af[ti] = As[(kk + lq) * LDA + 16 * ti + lr];
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) bf[tj] = Bs[(16 * tj + lr) * LDB + kk + lq];
#pragma unroll
            for (int ti = 0; ti < MT; ++ti)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[tj], af[ti], acc[ti][tj], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < MT; ++i)
        for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <int MT, int NT, int WAVES>
void run(const double* src, int blocks_per_cu, int iters) {
    const int nb = 256 * blocks_per_cu;
    double* out; unsigned long long *cyc, *rt;
    hipMalloc(&out, sizeof(double) * nb * 64 * WAVES);
    hipMalloc(&cyc, 8 * nb); hipMalloc(&rt, 8 * nb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_probe<MT, NT, WAVES>), dim3(nb), dim3(64 * WAVES), 0, 0, src, out, cyc, rt, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe<MT, NT, WAVES>), dim3(nb), dim3(64 * WAVES), 0, 0, src, out, cyc, rt, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hc(nb), hr(nb);
    hipMemcpy(hc.data(), cyc, 8 * nb, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), rt, 8 * nb, hipMemcpyDeviceToHost);
    const double flops = (double)iters * 4 * MT * NT * 2048.0 * WAVES * nb;
    printf("MT=%d NT=%d waves/WG=%d WG/CU=%d  loads/MFMA=%.3f : %.1f TFLOP/s, clock %.2f GHz, %.2f ms\n", MT, NT, WAVES, blocks_per_cu,
           (double)(MT + NT) / (MT * NT), flops / (ms * 1e-3) / 1e12, (double)hc[nb / 2] / (double)hr[nb / 2] * 0.1, ms);
    hipFree(out); hipFree(cyc); hipFree(rt);
}

int main() {
    std::vector<double> h(65536);
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    for (auto& v : h) v = nd(rng);
    double* src;
    hipMalloc(&src, h.size() * 8);
    hipMemcpy(src, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    const int it = 40000;
    run<4, 4, 4>(src, 1, it);
    run<4, 4, 4>(src, 2, it / 2);
    run<4, 3, 4>(src, 1, it);
    run<4, 2, 8>(src, 1, it);
    run<4, 2, 8>(src, 2, it / 2);
    run<2, 2, 4>(src, 1, it * 2);
    run<8, 4, 4>(src, 1, it / 2);
    run<1, 1, 4>(src, 1, it * 4);
    return 0;
}
