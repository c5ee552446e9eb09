// Stand-alone probe of the FP64 MFMA GEMM core (csrc/dto_gemm.hip.h): batched npad^3 products with
// in-kernel s_memtime stamps per phase, for tuning.  Diagnostic build only (stamps cost cycles).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I directtrajopt.jl_amd/csrc tools/bgemm_probe.hip -o tools/bgemm_probe
#include "dto_gemm.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace dto;

template <int T>
__global__ void __launch_bounds__(256, 2) k_probe(const double* A, const double* B, double* C, int npad, int nbatch,
                                                  unsigned long long* stamps) {
    using Cfg = GemmCfg<T, T>;
    __shared__ __attribute__((aligned(16))) double smem[Cfg::SMEM_DOUBLES];
    const int t1 = npad / T, tpm = t1 * t1;
    const int total = batch_tile_count(nbatch, tpm);
    const int64_t nn = (int64_t)npad * npad;
    GemmCoord<T, T> co;
    int iter = 0;
    for (int v = blockIdx.x; v < total; v += gridDim.x, ++iter) {
        int b, tile;
        if (!decode_batch_tile(v, nbatch, tpm, b, tile)) continue;
        unsigned long long s0 = __builtin_amdgcn_s_memtime();
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
        const int tr = tile % t1, tc = tile / t1;
        GemmAcc<T, T> acc;
        acc.zero();
        gemm_accumulate<T, T>(acc, A + b * nn + (int64_t)tr * T, npad, B + b * nn + (int64_t)tc * T * npad, npad, npad, nullptr, smem);
        unsigned long long s1 = __builtin_amdgcn_s_memtime();
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
        const int row0 = tr * T + co.row_base, col0 = tc * T + co.col_base;
        double* Cb = C + b * nn;
#pragma unroll
        for (int tj = 0; tj < Cfg::NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ti = 0; ti < Cfg::MT; ++ti)
                    Cb[(int64_t)(col0 + 16 * tj + 4 * r) * npad + row0 + 16 * ti] = acc.v[ti][tj][r];
        unsigned long long s2 = __builtin_amdgcn_s_memtime();
        if (stamps && threadIdx.x == 0 && iter < 32) {
            unsigned long long* p = stamps + ((size_t)blockIdx.x * 32 + iter) * 4;
            p[0] = s0; p[1] = s1; p[2] = s2; p[3] = r1 - r0;
        }
    }
}

int main(int argc, char** argv) {
    const int npad = argc > 1 ? atoi(argv[1]) : 256;
    const int nb = argc > 2 ? atoi(argv[2]) : 2000;
    const int wgs_per_cu = argc > 3 ? atoi(argv[3]) : 2;
    const size_t nn = (size_t)npad * npad;
    double *A, *B, *C; unsigned long long* st;
    hipMalloc(&A, nn * nb * 8); hipMalloc(&B, nn * nb * 8); hipMalloc(&C, nn * nb * 8);
    std::vector<double> h(nn * 8);
    for (auto& v : h) v = (double)rand() / RAND_MAX - 0.5;
    for (int i = 0; i < nb; ++i) {
        hipMemcpy(A + nn * i, h.data() + nn * (i % 7), nn * 8, hipMemcpyHostToDevice);
        hipMemcpy(B + nn * i, h.data() + nn * ((i + 3) % 7), nn * 8, hipMemcpyHostToDevice);
    }
    const int t1 = npad / 128, total = batch_tile_count(nb, t1 * t1);
    int grid = wgs_per_cu > 0 ? std::min(total, wgs_per_cu * 256) : total;
    hipMalloc(&st, (size_t)grid * 32 * 4 * 8);
    hipMemset(st, 0, (size_t)grid * 32 * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_probe<128>), dim3(grid), dim3(256), 0, 0, A, B, C, npad, nb, st);
    hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_probe<128>), dim3(grid), dim3(256), 0, 0, A, B, C, npad, nb, (unsigned long long*)nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("npad %d batch %d grid %d: %.3f ms  %.1f TFLOP/s\n", npad, nb, grid, ms, 2.0 * nn * npad * nb / (ms * 1e-3) / 1e12);
    std::vector<unsigned long long> hs((size_t)grid * 32 * 4);
    hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    double main_sum = 0, epi_sum = 0, gap_sum = 0, rt_sum = 0; long cnt = 0, gcnt = 0;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int w = 0; w < grid; ++w)
        for (int i = 0; i < 32; ++i) {
            unsigned long long* p = &hs[((size_t)w * 32 + i) * 4];
            if (!p[0]) continue;
            main_sum += (double)(p[1] - p[0]); epi_sum += (double)(p[2] - p[1]); ++cnt;
            tmin = std::min(tmin, p[0]); tmax = std::max(tmax, p[2]);
            rt_sum += (double)p[3];
            if (i + 1 < 32 && p[4]) { gap_sum += (double)(p[4] - p[2]); ++gcnt; }
        }
    printf("tiles stamped %ld: main loop %.0f cyc, epilogue %.0f cyc, inter-tile gap %.0f cyc; kernel span %.0f cyc (ideal main loop alone: %d cyc); clock in main loop %.2f GHz\n",
           cnt, main_sum / cnt, epi_sum / cnt, gcnt ? gap_sum / gcnt : 0.0, (double)(tmax - tmin), (npad / 4) * 16 * 64, main_sum / rt_sum * 0.1);
    return 0;
}
