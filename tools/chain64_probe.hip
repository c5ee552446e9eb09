// Stand-alone timing of the one-launch propagator chain (dto_chain64.hip) with s_memtime stamps at its phase boundaries:
// builds the kernel's arguments directly (no engine; a synthetic column-pointer array with room for 2 n entries per column).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DC64_STAMP -I directtrajopt.jl_amd/csrc tools/chain64_probe.hip -o tools/chain64_probe
// usage: chain64_probe [n m N scale force_form]
#include "dto_chain64.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

using namespace dto;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 64, m = argc > 2 ? atoi(argv[2]) : 4, N = argc > 3 ? atoi(argv[3]) : 1000;
    const double scale = argc > 4 ? atof(argv[4]) : 0.05;
    const int force = argc > 5 ? atoi(argv[5]) : 0;
    const int z = n + 2 * m + 1, K = N - 1;
    std::mt19937_64 rng(42);
    std::normal_distribution<double> nd;
    std::vector<double> G((size_t)(m + 1) * 4096, 0.0), Z((size_t)N * z);
    for (int j = 0; j <= m; ++j)
        for (int c = 0; c < n; ++c)
            for (int r = 0; r < n; ++r) G[(size_t)j * 4096 + c * 64 + r] = scale * nd(rng);
    for (int k = 0; k < N; ++k) {
        for (int r = 0; r < n; ++r) Z[(size_t)k * z + r] = nd(rng);
        for (int j = 0; j < 2 * m; ++j) Z[(size_t)k * z + n + j] = 0.3 * nd(rng);
        Z[(size_t)k * z + z - 1] = 0.1;
    }
    std::vector<int64_t> colptr((size_t)N * z + 1);
    for (size_t i = 0; i < colptr.size(); ++i) colptr[i] = (int64_t)i * 2 * n;
    double *dG, *dZ, *vals, *norms;
    int64_t* dcol;
    int32_t *smax, *sk;
    unsigned long long* stamp;
    CK(hipMalloc(&dG, G.size() * 8)); CK(hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&dZ, Z.size() * 8)); CK(hipMemcpy(dZ, Z.data(), Z.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&dcol, colptr.size() * 8)); CK(hipMemcpy(dcol, colptr.data(), colptr.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&vals, colptr.back() * 8)); CK(hipMalloc(&norms, (size_t)K * 4 * 8));
    CK(hipMalloc(&smax, 64)); CK(hipMemset(smax, 0, 64)); CK(hipMalloc(&sk, (size_t)K * 4));
    CK(hipMalloc(&stamp, 16 * 8)); CK(hipMemset(stamp, 0, 16 * 8));
    c64_stamp_buffer = stamp;
    KProb P{};
    P.N = N; P.K = K; P.z = z; P.dt_idx = z - 1; P.D = n; P.kn_lo = 0; P.n_knots = N; P.n_int = K; P.colptr = dcol;
    KBil B{};
    B.n = n; B.m = m; B.npad = 64; B.x_off = 0; B.u_off = n; B.G = dG; B.GT = dG; B.pre = 0;
    CK(chain64_prepare());
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) CK(launch_chain64(0, P, B, dZ, vals, norms, smax, reinterpret_cast<unsigned long long*>(smax + 2), sk, 60, force, prop.multiProcessorCount));
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int it = 0; it < reps; ++it) CK(launch_chain64(0, P, B, dZ, vals, norms, smax, reinterpret_cast<unsigned long long*>(smax + 2), sk, 60, force, prop.multiProcessorCount));
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int32_t hs[4]; CK(hipMemcpy(hs, smax, 16, hipMemcpyDeviceToHost));
    std::vector<double> v(64); CK(hipMemcpy(v.data(), vals, 64 * 8, hipMemcpyDeviceToHost));
    printf("n=%d m=%d N=%d: %.1f us per launch, max squarings %d, vals[0..2] = %g %g %g\n", n, m, N, ms / reps * 1e3, hs[0], v[0], v[1], v[2]);
    unsigned long long st[16]; CK(hipMemcpy(st, stamp, sizeof(st), hipMemcpyDeviceToHost));
    const char* names[10] = {"A build", "A^2 (+norm, store)", "A^3, A^4 (+norms, stores)", "scaling, coefficients", "polynomials", "K store",
                             "Y = A^4 K, Ya, Yb stores", "Ya Yb (+ form 3: L R)", "squarings", "-E store"};
    // s_memtime ticks at 100 MHz on this part
    for (int i = 0; i < 10; ++i) printf("  %-28s %6.2f us\n", names[i], (double)(st[i + 1] - st[i]) / 100.0);
    printf("  %-28s %6.2f us\n", "interval", (double)(st[10] - st[0]) / 100.0);
    return 0;
}
