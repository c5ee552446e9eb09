// Tuning probe #2: self-contained variants of the FP64 MFMA batched GEMM (tile, wave grid, K panel,
// debug modes) to locate the in-loop loss.  Not part of the product.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bgemm_probe2.hip -o tools/bgemm_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// TM x TN tile, WR x WC waves, KB panel depth; DBG: 0 normal, 1 no global loads after first panel,
// 2 no LDS fragment reads, 3 no MFMA (loads + LDS only)
template <int TM, int TN, int WR, int WC, int KB, int DBG, int MINW>
__global__ void __launch_bounds__(WR * WC * 64, MINW) k_gemm(const double* A, const double* B, double* C, int npad, int nbatch,
                                                             unsigned long long* stamps) {
    constexpr int NT_ = WR * WC * 64;
    constexpr int LDA_S = TM + 16, LDB_S = KB + 2;
    constexpr int AS = KB * LDA_S, BS = TN * LDB_S;
    constexpr int WTM = TM / WR, WTN = TN / WC, MT = WTM / 16, NT = WTN / 16;
    constexpr int A_LD = (TM * KB / 2) / NT_, B_LD = (TN * KB / 2) / NT_;
    __shared__ __attribute__((aligned(16))) double smem[2 * (AS + BS)];
    double* As = smem;
    double* Bs = smem + 2 * AS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WC, wn = wave % WC, lr = lane & 15, lq = lane >> 4;
    const int tr_n = npad / TM, tc_n = npad / TN, tpm = tr_n * tc_n;
    const int total = ((nbatch + 7) / 8) * 8 * tpm;
    const int64_t nn = (int64_t)npad * npad;
    int iter = 0;
    for (int v = blockIdx.x; v < total; v += gridDim.x, ++iter) {
        const int xcd = v & 7, idx = v >> 3;
        const int b = (idx / tpm) * 8 + xcd, tile = idx % tpm;
        if (b >= nbatch) continue;
        unsigned long long s0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        const int tr = tile % tr_n, tc = tile / tr_n;
        const double* Ab = A + b * nn + (int64_t)tr * TM;
        const double* Bb = B + b * nn + (int64_t)tc * TN * npad;
        d4 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = d4{0, 0, 0, 0};
        d2 ra[A_LD], rb[B_LD];
        auto load_panel = [&](int kb) {
            const int k0 = kb * KB;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const int id = tid + NT_ * i;
                ra[i] = *reinterpret_cast<const d2*>(Ab + (size_t)(k0 + id / (TM / 2)) * npad + 2 * (id % (TM / 2)));
            }
#pragma unroll
            for (int i = 0; i < B_LD; ++i) {
                const int id = tid + NT_ * i;
                rb[i] = *reinterpret_cast<const d2*>(Bb + (size_t)(id / (KB / 2)) * npad + k0 + 2 * (id % (KB / 2)));
            }
        };
        auto store_panel = [&](int buf) {
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const int id = tid + NT_ * i;
                *reinterpret_cast<d2*>(As + buf * AS + (id / (TM / 2)) * LDA_S + 2 * (id % (TM / 2))) = ra[i];
            }
#pragma unroll
            for (int i = 0; i < B_LD; ++i) {
                const int id = tid + NT_ * i;
                *reinterpret_cast<d2*>(Bs + buf * BS + (id / (KB / 2)) * LDB_S + 2 * (id % (KB / 2))) = rb[i];
            }
        };
        const int nkb = npad / KB;
        load_panel(0);
        store_panel(0);
        __syncthreads();
        double af[MT], bf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = 1.0 + lane;
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = 0.5 - lane;
        for (int kb = 0; kb < nkb; ++kb) {
            const int buf = kb & 1;
            if (kb + 1 < nkb && DBG != 1) load_panel(kb + 1);
            const double* as = As + buf * AS + wm * WTM + lr;
            const double* bs = Bs + buf * BS + (wn * WTN + lr) * LDB_S;
#pragma unroll
            for (int kk = 0; kk < KB; kk += 4) {
                if (DBG != 2) {
#pragma unroll
                    for (int ti = 0; ti < MT; ++ti) af[ti] = as[(kk + lq) * LDA_S + 16 * ti];
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) bf[tj] = bs[16 * tj * LDB_S + kk + lq];
                }
                if (DBG != 3) {
#pragma unroll
                    for (int ti = 0; ti < MT; ++ti)
#pragma unroll
                        for (int tj = 0; tj < NT; ++tj)
                            acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[tj], af[ti], acc[ti][tj], 0, 0, 0);
                } else {
#pragma unroll
                    for (int ti = 0; ti < MT; ++ti) acc[ti][0][0] += af[ti];
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) acc[0][tj][1] += bf[tj];
                }
            }
            if (kb + 1 < nkb && DBG != 1) store_panel(buf ^ 1);
            __syncthreads();
        }
        unsigned long long s1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        const int row0 = tr * TM + wm * WTM + lr, col0 = tc * TN + wn * WTN + lq;
        double* Cb = C + b * nn;
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ti = 0; ti < MT; ++ti) Cb[(int64_t)(col0 + 16 * tj + 4 * r) * npad + row0 + 16 * ti] = acc[ti][tj][r];
        unsigned long long s2 = __builtin_amdgcn_s_memtime();
        if (stamps && tid == 0 && iter < 32) {
            unsigned long long* p = stamps + ((size_t)blockIdx.x * 32 + iter) * 4;
            p[0] = s1 - s0; p[1] = s2 - s1; p[2] = r1 - r0; p[3] = 1;
        }
    }
}

template <int TM, int TN, int WR, int WC, int KB, int DBG, int MINW>
void run(const char* name, const double* A, const double* B, double* C, int npad, int nb, int wgs_per_cu, int reps = 5) {
    const int tpm = (npad / TM) * (npad / TN), total = ((nb + 7) / 8) * 8 * tpm;
    const int grid = std::min(total, wgs_per_cu * 256);
    unsigned long long* st;
    hipMalloc(&st, (size_t)grid * 32 * 4 * 8);
    hipMemset(st, 0, (size_t)grid * 32 * 4 * 8);
    auto k = k_gemm<TM, TN, WR, WC, KB, DBG, MINW>;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(WR * WC * 64), 0, 0, A, B, C, npad, nb, st);
    hipEventRecord(e0);
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(WR * WC * 64), 0, 0, A, B, C, npad, nb, (unsigned long long*)nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<unsigned long long> hs((size_t)grid * 32 * 4);
    hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    double m = 0, e = 0, rt = 0; long cnt = 0;
    for (size_t i = 0; i < hs.size(); i += 4) if (hs[i + 3]) { m += hs[i]; e += hs[i + 1]; rt += hs[i + 2]; ++cnt; }
    const double ideal = (double)(TM / WR / 16) * (TN / WC / 16) * (npad / 4) * 64.0;
    printf("%-34s grid %4d: %.3f ms %5.1f TF/s | main %7.0f cyc (MFMA-only %6.0f, x%d waves/SIMD) epi %6.0f cyc clock %.2f GHz\n", name, grid, ms,
           2.0 * npad * (double)npad * npad * nb / (ms * 1e-3) / 1e12, m / cnt, ideal, (WR * WC / 4) * wgs_per_cu, e / cnt, m / rt * 0.1);
    hipFree(st);
}

int main(int argc, char** argv) {
    // usage: bgemm_probe2 [nbatch reps]: with arguments only the production shapes run, on nbatch matrices -- e.g. 128 (operands and
    // output 192 MB: resident in the 256 MB Infinity Cache across the repetitions) against 2000 (3 GB: streamed from HBM)
    const int npad = 256, nb = argc > 1 ? atoi(argv[1]) : 2000;
    const int reps_arg = argc > 2 ? atoi(argv[2]) : 5;
    const size_t nn = (size_t)npad * npad;
    double *A, *B, *C;
    hipMalloc(&A, nn * nb * 8); hipMalloc(&B, nn * nb * 8); hipMalloc(&C, nn * nb * 8);
    std::vector<double> h(nn * 8);
    for (auto& v : h) v = (double)rand() / RAND_MAX - 0.5;
    for (int i = 0; i < nb; ++i) {
        hipMemcpy(A + nn * i, h.data() + nn * (i % 7), nn * 8, hipMemcpyHostToDevice);
        hipMemcpy(B + nn * i, h.data() + nn * ((i + 3) % 7), nn * 8, hipMemcpyHostToDevice);
    }
    if (argc > 1) {
        run<128, 128, 2, 2, 16, 0, 2>("128x128 4w KB16 (production)", A, B, C, npad, nb, 2, reps_arg);
        run<128, 128, 2, 4, 16, 0, 2>("128x128 8w(2x4) KB16 2wg/cu", A, B, C, npad, nb, 2, reps_arg);
        run<128, 128, 2, 2, 16, 1, 2>("  dbg1: no global loads (4w)", A, B, C, npad, nb, 2, reps_arg);
        return 0;
    }
    run<128, 128, 2, 2, 16, 0, 2>("128x128 4w KB16 (production)", A, B, C, npad, nb, 2);
    run<128, 128, 2, 2, 16, 1, 2>("  dbg1: no global loads", A, B, C, npad, nb, 2);
    run<128, 128, 2, 2, 16, 2, 2>("  dbg2: no LDS fragment reads", A, B, C, npad, nb, 2);
    run<128, 128, 2, 2, 16, 3, 2>("  dbg3: no MFMA", A, B, C, npad, nb, 2);
    run<128, 128, 2, 2, 8, 0, 2>("128x128 4w KB8", A, B, C, npad, nb, 2);
    run<128, 128, 2, 4, 16, 0, 2>("128x128 8w(2x4) KB16 2wg/cu", A, B, C, npad, nb, 2);
    run<128, 128, 4, 2, 16, 0, 2>("128x128 8w(4x2) KB16 2wg/cu", A, B, C, npad, nb, 2);
    run<256, 128, 4, 2, 16, 0, 2>("256x128 8w(4x2) KB16 1wg/cu", A, B, C, npad, nb, 1);
    run<128, 128, 2, 2, 8, 0, 3>("128x128 4w KB8 3wg/cu (<=168 VGPR)", A, B, C, npad, nb, 3);
    run<128, 128, 2, 2, 4, 0, 3>("128x128 4w KB4 3wg/cu (<=168 VGPR)", A, B, C, npad, nb, 3);
    run<128, 128, 2, 2, 8, 0, 2>("128x128 4w KB8 2wg/cu", A, B, C, npad, nb, 2);
    return 0;
}
