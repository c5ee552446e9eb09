// Tuning probe #3: direct global->LDS DMA staging (global_load_lds_dwordx4) for the FP64 MFMA batched GEMM.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bgemm_probe3.hip -o tools/bgemm_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 128x128 tile, 4 waves (2x2), KB = 16.  LDS images (no padding, DMA needs wave-contiguous 1 KB pieces):
//   As[k][m'] with m' = (m + 16*(k&1)) % 128   (rotation makes the two k's of a 32-lane read group hit different banks)
//   Bs[kp][c][2]  (pairs of k adjacent: a 32-lane group reads 256 contiguous bytes)
template <int MODE>
__global__ void __launch_bounds__(256, 2) k_gemm(const double* A, const double* B, double* C, int npad, int nbatch) {
    constexpr int TM = 128, TN = 128, KB = 16;
    constexpr int AS = KB * TM, BS = KB * TN;
    __shared__ __attribute__((aligned(1024))) double smem[2 * (AS + BS)];
    double* As = smem;
    double* Bs = smem + 2 * AS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lq = lane >> 4;
    const int t1 = npad / TM, tpm = t1 * t1;
    const int total = ((nbatch + 7) / 8) * 8 * tpm;
    const int64_t nn = (int64_t)npad * npad;
    for (int v = blockIdx.x; v < total; v += gridDim.x) {
        const int xcd = v & 7, idx = v >> 3;
        const int b = (idx / tpm) * 8 + xcd, tile = idx % tpm;
        if (b >= nbatch) continue;
        const int tr = tile % t1, tc = tile / t1;
        const double* Ab = A + b * nn + (int64_t)tr * TM;
        const double* Bb = B + b * nn + (int64_t)tc * TN * npad;
        d4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = d4{0, 0, 0, 0};
        auto dma_panel = [&](int kb, int buf) {
            const int k0 = kb * KB;
            // A: 16 k-rows of 1 KB; wave w issues rows w, w+4, w+8, w+12
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = wave + 4 * q;
                const int mp = 2 * lane;                      // destination position (doubles) inside the row
                const int m = (mp - 16 * (k & 1)) & (TM - 1);  // source row index
                __builtin_amdgcn_global_load_lds(GLB_PTR(Ab + (size_t)(k0 + k) * npad + m), LDS_PTR(As + buf * AS + k * TM), 16, 0, 0);
            }
            // B: 8 k-pairs x 128 columns of 16 B = 16 pieces of 1 KB; wave w issues pieces w, w+4, ...
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int piece = wave + 4 * q;
                const int g = piece * 64 + lane;
                const int kp = g / TN, c = g % TN;
                __builtin_amdgcn_global_load_lds(GLB_PTR(Bb + (size_t)c * npad + k0 + 2 * kp), LDS_PTR(Bs + buf * BS + piece * 128), 16, 0, 0);
            }
        };
        const int nkb = npad / KB;
        dma_panel(0, 0);
        __syncthreads();
        for (int kb = 0; kb < nkb; ++kb) {
            const int buf = kb & 1;
            if (kb + 1 < nkb) dma_panel(kb + 1, buf ^ 1);
            const double* as = As + buf * AS;
            const double* bs = Bs + buf * BS;
#pragma unroll
            for (int kk = 0; kk < KB; kk += 4) {
                const int k = kk + lq;
                double af[4], bf[4];
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) af[ti] = as[k * TM + ((wm * 64 + 16 * ti + lr + 16 * (k & 1)) & (TM - 1))];
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) bf[tj] = bs[(k >> 1) * (TN * 2) + (wn * 64 + 16 * tj + lr) * 2 + (k & 1)];
#pragma unroll
                for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[tj], af[ti], acc[ti][tj], 0, 0, 0);
            }
            __syncthreads();
        }
        const int row0 = tr * TM + wm * 64 + lr, col0 = tc * TN + wn * 64 + lq;
        double* Cb = C + b * nn;
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) Cb[(int64_t)(col0 + 16 * tj + 4 * r) * npad + row0 + 16 * ti] = acc[ti][tj][r];
    }
}

int main() {
    const int npad = 256, nb = 2000;
    const size_t nn = (size_t)npad * npad;
    double *A, *B, *C;
    hipMalloc(&A, nn * nb * 8); hipMalloc(&B, nn * nb * 8); hipMalloc(&C, nn * nb * 8);
    std::vector<double> h(nn * 8);
    for (auto& v : h) v = (double)rand() / RAND_MAX - 0.5;
    for (int i = 0; i < nb; ++i) {
        hipMemcpy(A + nn * i, h.data() + nn * (i % 7), nn * 8, hipMemcpyHostToDevice);
        hipMemcpy(B + nn * i, h.data() + nn * ((i + 3) % 7), nn * 8, hipMemcpyHostToDevice);
    }
    // correctness of the first matrix against a host product
    hipLaunchKernelGGL((k_gemm<0>), dim3(512), dim3(256), 0, 0, A, B, C, npad, nb);
    std::vector<double> c0(nn), ref(nn, 0.0);
    hipMemcpy(c0.data(), C + nn * 5, nn * 8, hipMemcpyDeviceToHost);
    const double* a = h.data() + nn * (5 % 7); const double* bb = h.data() + nn * ((5 + 3) % 7);
    for (int j = 0; j < npad; ++j) for (int k = 0; k < npad; ++k) { double bv = bb[k + (size_t)j * npad]; for (int i = 0; i < npad; ++i) ref[i + (size_t)j * npad] += a[i + (size_t)k * npad] * bv; }
    double err = 0; for (size_t i = 0; i < nn; ++i) err = std::max(err, std::abs(c0[i] - ref[i]));
    printf("max |C - ref| = %.3e\n", err);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {512, 8000}) {
        hipEventRecord(e0);
        const int reps = 5;
        for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_gemm<0>), dim3(grid), dim3(256), 0, 0, A, B, C, npad, nb);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        printf("DMA staging grid %d: %.3f ms  %.1f TFLOP/s\n", grid, ms, 2.0 * nn * npad * nb / (ms * 1e-3) / 1e12);
    }
    return 0;
}
