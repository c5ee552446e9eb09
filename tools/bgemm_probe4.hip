// Tuning probe #4 (round 3): the batched FP64 MFMA GEMM core with 16-byte LDS fragment reads and 16-byte epilogue accesses.
// Not part of the product.  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bgemm_probe4.hip -o tools/bgemm_probe4
//
// V0: the production register-staged core (8-byte fragment reads, rows 16 ti + lr per accumulator tile, 8-byte epilogue stores)
// V1: "paired rows": accumulator tile ti of a wave covers rows 32 (ti/2) + 2 lr + (ti & 1), so a lane's two tiles 2p, 2p+1 hold
//     two CONSECUTIVE rows -> one ds_read_b128 feeds both A fragments, and the epilogue moves 16 bytes per lane (256 contiguous
//     bytes per 16 lanes); the B fragments of two k-steps come from one ds_read_b128 by running the 8 k's of a double step in the
//     order {0,2,4,6},{1,3,5,7} (lane group lq supplies k = 2 lq + s in step s, for both operands).
//     LDS images: As[k][TM] unpadded (pitch 1 KB: conflict-free for the b128 lane groups), Bs[c][KB+4] (pitch 160 B: brute-forced).
// usage: bgemm_probe4 [nbatch=2000] [reps=10] [square=0|1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
#define GLBP(p) ((const __attribute__((address_space(1))) void*)(p))

// "heavy" epilogue of probe #5: the two-output polynomial product's traffic (four matrices streamed in, two out), set by the host
struct HeavyEpi { const double* M[4]; double* C2; int on; };
__device__ HeavyEpi g_heavy;
__device__ __forceinline__ void heavy_epilogue(d2 v, double* Cb_off, int64_t off_in_batch, int64_t boff) {
    const HeavyEpi h = g_heavy;
    const int64_t off = boff + off_in_batch;
    const d2 m1 = __builtin_nontemporal_load(reinterpret_cast<const d2*>(h.M[0] + off)), m2 = __builtin_nontemporal_load(reinterpret_cast<const d2*>(h.M[1] + off)),
             m3 = __builtin_nontemporal_load(reinterpret_cast<const d2*>(h.M[2] + off)), m4 = __builtin_nontemporal_load(reinterpret_cast<const d2*>(h.M[3] + off));
    __builtin_nontemporal_store(v + 0.5 * m1 + 0.25 * m2 + 0.125 * m3 + 0.0625 * m4, reinterpret_cast<d2*>(Cb_off));
    __builtin_nontemporal_store(v - 0.5 * m1 + 0.25 * m2 - 0.125 * m3 + 0.0625 * m4, reinterpret_cast<d2*>(h.C2 + off));
}

// ---------------------------------------------------------------------------------------------------- V0 (production copy)
template <int MINW>
__global__ void __launch_bounds__(256, MINW) k_v0(const double* A, const double* B, double* C, int npad, int nbatch) {
    constexpr int TM = 128, TN = 128, WR = 2, WC = 2, KB = 16, NT_ = 256;
    constexpr int LDA_S = TM + 16, LDB_S = KB + 2, AS = KB * LDA_S, BS = TN * LDB_S;
    constexpr int WTM = 64, WTN = 64, MT = 4, NT = 4, A_LD = 4, B_LD = 4;
    __shared__ __attribute__((aligned(16))) double smem[2 * (AS + BS)];
    double* As = smem;
    double* Bs = smem + 2 * AS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WC, wn = wave % WC, lr = lane & 15, lq = lane >> 4;
    const int tr_n = npad / TM, tpm = tr_n * (npad / TN);
    const int total = ((nbatch + 7) / 8) * 8 * tpm;
    const int64_t nn = (int64_t)npad * npad;
    for (int v = blockIdx.x; v < total; v += gridDim.x) {
        const int xcd = v & 7, idx = v >> 3;
        const int b = (idx / tpm) * 8 + xcd, tile = idx % tpm;
        if (b >= nbatch) continue;
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
        for (int kb = 0; kb < nkb; ++kb) {
            const int buf = kb & 1;
            if (kb + 1 < nkb) load_panel(kb + 1);
            const double* as = As + buf * AS + wm * WTM + lr;
            const double* bs = Bs + buf * BS + (wn * WTN + lr) * LDB_S;
#pragma unroll
            for (int kk = 0; kk < KB; kk += 4) {
                __builtin_amdgcn_iglp_opt(0);
                double af[MT], bf[NT];
#pragma unroll
                for (int ti = 0; ti < MT; ++ti) af[ti] = as[(kk + lq) * LDA_S + 16 * ti];
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) bf[tj] = bs[16 * tj * LDB_S + kk + lq];
#pragma unroll
                for (int ti = 0; ti < MT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[tj], af[ti], acc[ti][tj], 0, 0, 0);
            }
            if (kb + 1 < nkb) store_panel(buf ^ 1);
            __syncthreads();
        }
        const int row0 = tr * TM + wm * WTM + lr, col0 = tc * TN + wn * WTN + lq;
        double* Cb = C + b * nn;
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ti = 0; ti < MT; ++ti)
                    __builtin_nontemporal_store(acc[ti][tj][r], &Cb[(int64_t)(col0 + 16 * tj + 4 * r) * npad + row0 + 16 * ti]);
    }
}

// ---------------------------------------------------------------------------------------------------- V1 (paired rows, b128)
// STAGE: 0 = register staging for both operands, 1 = A by LDS-DMA (B by registers)
// IGLP: call iglp_opt(0) in the k loop
template <int MINW, int STAGE, int IGLP, int WC_ = 2>
__global__ void __launch_bounds__(128 * WC_, MINW * WC_ / 2) k_v1(const double* A, const double* B, double* C, int npad, int nbatch, int pitch_off) {
    constexpr int TM = 128, TN = 128, WR = 2, WC = WC_, KB = 16, NT_ = 64 * WR * WC;
    constexpr int LDB_S = KB + 4, AS = KB * TM, BS = TN * LDB_S;
    constexpr int WTM = TM / WR, WTN = TN / WC, MT = WTM / 16, NT = WTN / 16, A_LD = (TM * KB / 2) / NT_, B_LD = (TN * KB / 2) / NT_;
    __shared__ __attribute__((aligned(1024))) double smem[2 * (AS + BS)];
    double* As = smem;
    double* Bs = smem + 2 * AS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WC, wn = wave % WC, lr = lane & 15, lq = lane >> 4;
    const int tr_n = npad / TM, tpm = tr_n * (npad / TN);
    const int total = ((nbatch + 7) / 8) * 8 * tpm;
    const int64_t nn = (int64_t)npad * npad;
    for (int v = blockIdx.x; v < total; v += gridDim.x) {
        const int xcd = v & 7, idx = v >> 3;
        const int b = (idx / tpm) * 8 + xcd, tile = idx % tpm;
        if (b >= nbatch) continue;
        const int tr = tile % tr_n, tc = tile / tr_n;
        const double* Ab = A + b * nn + (int64_t)tr * TM;
        const double* Bb = B + b * nn + (int64_t)tc * TN * npad;
        d4 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = d4{0, 0, 0, 0};
        d2 ra[STAGE == 1 ? 1 : A_LD], rb[B_LD];
        auto load_panel = [&](int kb, int buf) {
            const int k0 = kb * KB;
            if constexpr (STAGE == 1) {
                // one 1 KB piece per k row: As[k][0..127] <- A[k0 + k][tile rows], wave-contiguous
#pragma unroll
                for (int q = 0; q < KB / (NT_ / 64); ++q) {
                    const int k = wave + (NT_ / 64) * q;
                    __builtin_amdgcn_global_load_lds(GLBP(Ab + (size_t)(k0 + k) * npad + 2 * lane), LDSP(As + buf * AS + k * TM), 16, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < A_LD; ++i) {
                    const int id = tid + NT_ * i;
                    ra[i] = *reinterpret_cast<const d2*>(Ab + (size_t)(k0 + id / (TM / 2)) * npad + 2 * (id % (TM / 2)));
                }
            }
#pragma unroll
            for (int i = 0; i < B_LD; ++i) {
                const int id = tid + NT_ * i;
                rb[i] = *reinterpret_cast<const d2*>(Bb + (size_t)(id / (KB / 2)) * npad + k0 + 2 * (id % (KB / 2)));
            }
        };
        auto store_panel = [&](int buf) {
            if constexpr (STAGE != 1) {
#pragma unroll
                for (int i = 0; i < A_LD; ++i) {
                    const int id = tid + NT_ * i;
                    *reinterpret_cast<d2*>(As + buf * AS + (id / (TM / 2)) * TM + 2 * (id % (TM / 2))) = ra[i];
                }
            }
#pragma unroll
            for (int i = 0; i < B_LD; ++i) {
                const int id = tid + NT_ * i;
                *reinterpret_cast<d2*>(Bs + buf * BS + (id / (KB / 2)) * LDB_S + 2 * (id % (KB / 2))) = rb[i];
            }
        };
        const int nkb = npad / KB;
        load_panel(0, 0);
        store_panel(0);
        __syncthreads();
        for (int kb = 0; kb < nkb; ++kb) {
            const int buf = kb & 1;
            if (kb + 1 < nkb) load_panel(kb + 1, buf ^ 1);
            const double* as = As + buf * AS + wm * WTM + 2 * lr;
            const double* bs = Bs + buf * BS + (wn * WTN + lr) * LDB_S + 2 * lq;
#pragma unroll
            for (int k8 = 0; k8 < KB; k8 += 8) {
                if (IGLP) __builtin_amdgcn_iglp_opt(0);
                d2 b2[NT];
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) b2[tj] = *reinterpret_cast<const d2*>(bs + 16 * tj * LDB_S + k8);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    d2 a2[MT / 2];
#pragma unroll
                    for (int p = 0; p < MT / 2; ++p) a2[p] = *reinterpret_cast<const d2*>(as + (k8 + 2 * lq + s) * TM + 32 * p);
#pragma unroll
                    for (int ti = 0; ti < MT; ++ti)
#pragma unroll
                        for (int tj = 0; tj < NT; ++tj)
                            acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(s ? b2[tj].y : b2[tj].x, (ti & 1) ? a2[ti / 2].y : a2[ti / 2].x,
                                                                              acc[ti][tj], 0, 0, 0);
                }
            }
            if (kb + 1 < nkb) store_panel(buf ^ 1);
            __syncthreads();
        }
        // lane: rows 32 p + 2 lr, +1 of the wave tile (tiles 2p, 2p+1), columns lq + 4 r of column tile tj
        const int row0 = tr * TM + wm * WTM + 2 * lr, col0 = tc * TN + wn * WTN + lq;
        double* Cb = C + b * nn + pitch_off;  // pitch_off (odd): every 16-byte store is only 8-byte aligned, like the Jacobian slab's columns
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int p = 0; p < MT / 2; ++p) {
                    d2u vv = {acc[2 * p][tj][r], acc[2 * p + 1][tj][r]};
                    if (pitch_off < 0) {   // heavy epilogue (probe #5)
                        const int64_t o = (int64_t)(col0 + 16 * tj + 4 * r) * npad + row0 + 32 * p;
                        heavy_epilogue(d2{vv.x, vv.y}, C + b * nn + o, o, b * nn);
                        continue;
                    }
                    __builtin_nontemporal_store(vv, reinterpret_cast<d2u*>(&Cb[(int64_t)(col0 + 16 * tj + 4 * r) * npad + row0 + 32 * p]));
                }
    }
}

template <class K>
double run(const char* name, K k, int threads, const double* A, const double* B, double* C, int npad, int nb, int wgs_per_cu, int reps, int extra = -999) {
    const int tpm = (npad / 128) * (npad / 128), total = ((nb + 7) / 8) * 8 * tpm;
    const int grid = wgs_per_cu > 0 ? std::min(total, wgs_per_cu * 256) : total;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto launch = [&] {
        if (extra != -999) hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 0, 0, A, B, C, npad, nb, extra);
    };
    for (int w = 0; w < 3; ++w) launch();
    hipEventRecord(e0);
    for (int w = 0; w < reps; ++w) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-58s grid %5d: %.3f ms %5.1f TF/s\n", name, grid, ms, 2.0 * npad * (double)npad * npad * nb / (ms * 1e-3) / 1e12);
    fflush(stdout);
    return ms;
}
template <class K>
double run0(const char* name, K k, const double* A, const double* B, double* C, int npad, int nb, int wgs_per_cu, int reps) {
    const int tpm = (npad / 128) * (npad / 128), total = ((nb + 7) / 8) * 8 * tpm;
    const int grid = wgs_per_cu > 0 ? std::min(total, wgs_per_cu * 256) : total;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, A, B, C, npad, nb);
    hipEventRecord(e0);
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, A, B, C, npad, nb);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-58s grid %5d: %.3f ms %5.1f TF/s\n", name, grid, ms, 2.0 * npad * (double)npad * npad * nb / (ms * 1e-3) / 1e12);
    fflush(stdout);
    return ms;
}

int main(int argc, char** argv) {
    const int npad = 256, nb = argc > 1 ? atoi(argv[1]) : 2000;
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const int square = argc > 3 ? atoi(argv[3]) : 0;
    const size_t nn = (size_t)npad * npad;
    double *A, *B, *C, *C2;
    hipMalloc(&A, nn * nb * 8);
    hipMalloc(&B, nn * nb * 8);
    hipMalloc(&C, (nn * nb + 16) * 8);
    hipMalloc(&C2, (nn * nb + 16) * 8);
    std::vector<double> h(nn * 8);
    for (auto& v : h) v = (double)rand() / RAND_MAX - 0.5;
    for (int i = 0; i < nb; ++i) {
        hipMemcpy(A + nn * i, h.data() + nn * (i % 7), nn * 8, hipMemcpyHostToDevice);
        hipMemcpy(B + nn * i, h.data() + nn * ((i + 3) % 7), nn * 8, hipMemcpyHostToDevice);
    }
    const double* Bm = square ? A : B;
    printf("nbatch %d, %s\n", nb, square ? "C = A A (squaring)" : "C = A B");
    // correctness of V1 against V0 on the first matrices
    hipLaunchKernelGGL((k_v0<2>), dim3(512), dim3(256), 0, 0, A, Bm, C, npad, std::min(nb, 16));
    hipLaunchKernelGGL((k_v1<2, 0, 0>), dim3(512), dim3(256), 0, 0, A, Bm, C2, npad, std::min(nb, 16), 0);
    hipDeviceSynchronize();
    {
        std::vector<double> c0(nn * 16), c1(nn * 16);
        hipMemcpy(c0.data(), C, c0.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(c1.data(), C2, c1.size() * 8, hipMemcpyDeviceToHost);
        double e = 0, m = 0;
        for (size_t i = 0; i < nn * std::min(nb, 16); ++i) { e = std::max(e, fabs(c0[i] - c1[i])); m = std::max(m, fabs(c0[i])); }
        printf("V1 vs V0: max abs diff %.3e (max |C| %.3e)\n", e, m);
        hipLaunchKernelGGL((k_v1<2, 1, 0>), dim3(512), dim3(256), 0, 0, A, Bm, C2, npad, std::min(nb, 16), 0);
        hipLaunchKernelGGL((k_v1<2, 0, 1, 4>), dim3(512), dim3(512), 0, 0, A, Bm, C, npad, std::min(nb, 16), 0);
        hipDeviceSynchronize();
        hipMemcpy(c1.data(), C2, c1.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> c2(nn * 16);
        hipMemcpy(c2.data(), C, c2.size() * 8, hipMemcpyDeviceToHost);
        double e1 = 0, e2 = 0;
        for (size_t i = 0; i < nn * std::min(nb, 16); ++i) { e1 = std::max(e1, fabs(c0[i] - c1[i])); e2 = std::max(e2, fabs(c0[i] - c2[i])); }
        printf("V1/DMA-A vs V0: %.3e   V1/8 waves vs V0: %.3e\n", e1, e2);
    }
    const double sustain = argc > 4 ? atof(argv[4]) : 0.0;
    if (sustain > 0) {
        // the chip lowers its clock under a sustained FP64 matrix load: rate per ~0.1 s window over `sustain` seconds
        auto sustained = [&](const char* name, auto launch) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            printf("%s, TFLOP/s per 0.1 s window:", name);
            double t = 0;
            while (t < sustain) {
                hipEventRecord(e0);
                for (int i = 0; i < 90; ++i) launch();
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                t += ms * 1e-3;
                printf(" %.1f", 90 * 2.0 * npad * (double)npad * npad * nb / (ms * 1e-3) / 1e12);
            }
            printf("\n");
            fflush(stdout);
        };
        sustained("V0 8-byte core, persistent", [&] { hipLaunchKernelGGL((k_v0<2>), dim3(512), dim3(256), 0, 0, A, Bm, C, npad, nb); });
        sustained("V1 paired rows, DMA-A, persistent", [&] { hipLaunchKernelGGL((k_v1<2, 1, 0>), dim3(512), dim3(256), 0, 0, A, Bm, C, npad, nb, 0); });
        sustained("V0 8-byte core, persistent (again)", [&] { hipLaunchKernelGGL((k_v0<2>), dim3(512), dim3(256), 0, 0, A, Bm, C, npad, nb); });
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        run0("V0 production core, persistent 2 wg/cu", k_v0<2>, A, Bm, C, npad, nb, 2, reps);
        run0("V0 production core, one wg per tile", k_v0<2>, A, Bm, C, npad, nb, 0, reps);
        run("V1 paired rows b128, regs, persistent", k_v1<2, 0, 0>, 256, A, Bm, C, npad, nb, 2, reps, 0);
        run("V1 paired rows b128, regs, iglp, persistent", k_v1<2, 0, 1>, 256, A, Bm, C, npad, nb, 2, reps, 0);
        run("V1 paired rows b128, regs, one wg per tile", k_v1<2, 0, 0>, 256, A, Bm, C, npad, nb, 0, reps, 0);
        run("V1 paired rows b128, DMA-A, persistent", k_v1<2, 1, 0>, 256, A, Bm, C, npad, nb, 2, reps, 0);
        run("V1 paired rows b128, DMA-A, iglp, persistent", k_v1<2, 1, 1>, 256, A, Bm, C, npad, nb, 2, reps, 0);
        run("V1 paired rows b128, DMA-A, one wg per tile", k_v1<2, 1, 0>, 256, A, Bm, C, npad, nb, 0, reps, 0);
        run("V1 8 waves (2x4), regs, iglp, persistent", k_v1<2, 0, 1, 4>, 512, A, Bm, C, npad, nb, 2, reps, 0);
        run("V1 8 waves (2x4), regs, one wg per tile", k_v1<2, 0, 1, 4>, 512, A, Bm, C, npad, nb, 0, reps, 0);
        run("V1 regs persistent, stores 8-byte aligned only (off 1)", k_v1<2, 0, 0>, 256, A, Bm, C, npad, nb, 2, reps, 1);
    }
    return 0;
}
