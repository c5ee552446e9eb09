// How fast can the polynomial-product epilogue's memory pattern go?  Streams 4 matrices in and 2 out (the two-output product's
// epilogue traffic) for 2000 matrices of 256x256 doubles with
//   pattern 0: the MFMA accumulator layout -- 16 lanes x 8 B contiguous (128-B segments), 4 column segments per wave-instruction
//   pattern 1: 16 B per lane, a wave-instruction covers 1 KB of one column
// build: hipcc --offload-arch=gfx950 -O3 tools/epilogue_stream_probe.hip -o tools/epilogue_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int NP = 256;

template <int PATTERN>
__global__ void __launch_bounds__(512, 2) k_stream(const double* __restrict__ M1, const double* __restrict__ M2, const double* __restrict__ M3,
                                                   const double* __restrict__ M4, double* __restrict__ O1, double* __restrict__ O2, int nb) {
    // one workgroup per 128x128 tile, 8 waves of 64x32 as in the engine
    const int tiles = 4;
    const int b = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int tr = tile % 2, tc = tile / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / 4, wn = wave % 4;
    const size_t base = (size_t)b * NP * NP;
    if (PATTERN == 0) {
        const int row0 = tr * 128 + wm * 64 + (lane & 15), col0 = tc * 128 + wn * 32 + (lane >> 4);
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = col0 + 16 * tj + 4 * r;
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) {
                    const size_t off = base + (size_t)col * NP + row0 + 16 * ti;
                    const double m1 = __builtin_nontemporal_load(&M1[off]), m2 = __builtin_nontemporal_load(&M2[off]),
                                 m3 = __builtin_nontemporal_load(&M3[off]), m4 = __builtin_nontemporal_load(&M4[off]);
                    __builtin_nontemporal_store(m1 + 2 * m2 + 3 * m3 + 4 * m4, &O1[off]);
                    __builtin_nontemporal_store(m1 - 2 * m2 + 3 * m3 - 4 * m4, &O2[off]);
                }
            }
    } else {
        // the same 64x32 wave tile, but a lane owns two adjacent rows: 32 lanes cover 64 rows of a column, 2 columns per instruction
        const int row0 = tr * 128 + wm * 64 + 2 * (lane & 31), col0 = tc * 128 + wn * 32 + (lane >> 5);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int col = col0 + 2 * c;
            const size_t off = base + (size_t)col * NP + row0;
            const d2 m1 = __builtin_nontemporal_load(reinterpret_cast<const d2*>(&M1[off])), m2 = __builtin_nontemporal_load(reinterpret_cast<const d2*>(&M2[off])),
                     m3 = __builtin_nontemporal_load(reinterpret_cast<const d2*>(&M3[off])), m4 = __builtin_nontemporal_load(reinterpret_cast<const d2*>(&M4[off]));
            __builtin_nontemporal_store(m1 + 2 * m2 + 3 * m3 + 4 * m4, reinterpret_cast<d2*>(&O1[off]));
            __builtin_nontemporal_store(m1 - 2 * m2 + 3 * m3 - 4 * m4, reinterpret_cast<d2*>(&O2[off]));
        }
    }
}
int main() {
    const int nb = 2000;
    const size_t n = (size_t)nb * NP * NP;
    double *M[4], *O[2];
    for (auto& p : M) { hipMalloc(&p, n * 8); hipMemset(p, 0, n * 8); }
    for (auto& p : O) hipMalloc(&p, n * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 2; ++pat)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (pat == 0) hipLaunchKernelGGL(k_stream<0>, dim3(nb * 4), dim3(512), 0, 0, M[0], M[1], M[2], M[3], O[0], O[1], nb);
            else hipLaunchKernelGGL(k_stream<1>, dim3(nb * 4), dim3(512), 0, 0, M[0], M[1], M[2], M[3], O[0], O[1], nb);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("pattern %d: %.3f ms  %.2f TB/s (4 reads + 2 writes of %.2f GB each)\n", pat, ms, 6.0 * n * 8 / ms / 1e9, n * 8 / 1e9);
        }
    return 0;
}
