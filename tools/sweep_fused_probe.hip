// Stand-alone timing of the fused generator sweep (dto_sweep_fused.hip) on the headline shape: builds the kernel's
// arguments directly (no engine), runs the Jacobian's forward sweep (T = 1 + m column types) and prints time, the
// statistics [non-converged workgroups, max terms, sum of terms] and the implied MFMA rate.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I directtrajopt.jl_amd/csrc tools/sweep_fused_probe.hip -o tools/sweep_fused_probe [-DPROBE_...]
// usage: sweep_fused_probe [n m N d_ub tc ipw_override shared_chip]   (shared_chip = 1: the two-column-group form of 256 states)
#include "dto_sweep_fused.hip"

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

using namespace dto;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256, m = argc > 2 ? atoi(argv[2]) : 4, N = argc > 3 ? atoi(argv[3]) : 2000;
    const int d_ub = argc > 4 ? atoi(argv[4]) : 30;
    const int tc_arg = argc > 5 ? atoi(argv[5]) : -1;
    const int ipw_over = argc > 6 ? atoi(argv[6]) : 0;
    const bool shared = argc > 7 && atoi(argv[7]) != 0;
    const int npad = (n + 63) / 64 * 64, z = n + 2 * m + 1, K = N - 1;
    std::mt19937_64 rng(42);
    std::normal_distribution<double> nd;
    const size_t nn = (size_t)npad * npad;
    std::vector<double> G((m + 1) * nn + 16 * (size_t)npad, 0.0), Z((size_t)N * z);
    for (int j = 0; j <= m; ++j)
        for (int c = 0; c < n; ++c)
            for (int r = 0; r < n; ++r) G[j * nn + (size_t)c * npad + r] = nd(rng);
    for (int k = 0; k < N; ++k) {
        for (int r = 0; r < n; ++r) Z[(size_t)k * z + r] = nd(rng);
        for (int j = 0; j < m; ++j) Z[(size_t)k * z + n + j] = 0.1 * nd(rng);
        for (int j = 0; j < m; ++j) Z[(size_t)k * z + n + m + j] = nd(rng);
        Z[(size_t)k * z + z - 1] = 0.1;
    }
    double *dG, *dZ;
    CK(hipMalloc(&dG, G.size() * 8)); CK(hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&dZ, Z.size() * 8)); CK(hipMemcpy(dZ, Z.data(), Z.size() * 8, hipMemcpyHostToDevice));
    KProb P{};
    P.N = N; P.K = K; P.z = z; P.dt_idx = z - 1; P.D = n + m; P.kn_lo = 0; P.n_knots = N; P.n_int = K;
    KBil B{};
    B.n = n; B.m = m; B.npad = npad; B.x_off = 0; B.u_off = n; B.G = dG; B.GT = dG;
    SweepTypes ty{};
    ty.T = 1 + m;
    for (int j = 0; j < m; ++j) { ty.t[1 + j].n_extra = 1; ty.t[1 + j].gen[0] = 1 + j; ty.t[1 + j].src[0] = 0; ty.t[1 + j].mult[0] = 1.0; }
    SweepBuf w{};
    w.npad = npad; w.TN = 128; w.Kpad = (K + 127) / 128 * 128; w.T_alloc = ty.T;
    const size_t typesz = (size_t)w.Kpad * npad;
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&w.Z[i], typesz * ty.T * 8)); CK(hipMemset(w.Z[i], 0, typesz * ty.T * 8)); }
    CK(hipMalloc(&w.S, typesz * ty.T * 8));
    CK(hipMalloc(&w.scaleA, (m + 1) * (size_t)w.Kpad * 8)); CK(hipMalloc(&w.scaleU, (m + 1) * (size_t)w.Kpad * 8));
    CK(hipMalloc(&w.scaleE, 2 * (size_t)w.Kpad * 8));
    CK(hipMalloc(&w.stats, 16));
    CK(sweep_fused_prepare());
    FusedSweepPlan pl{};
    if (!sweep_fused_plan(npad, m, ty, K, 256, pl, shared)) { printf("no plan\n"); return 1; }
    if (ipw_over > 0) {
        pl.ipw = ipw_over; pl.NT = (ty.T * ipw_over + 15) / 16 / pl.WC; pl.nblocks = (K + ipw_over - 1) / ipw_over;
        pl.lds_bytes = (size_t)FusedLds(npad, ty.T, m, ipw_over, pl.nslot, pl.MT).total * 8;
    }
    const int tc = tc_arg >= 0 ? tc_arg : d_ub / 2 - 1;
    printf("n=%d m=%d N=%d npad=%d  plan: WC=%d MT=%d NT=%d ipw=%d blocks=%d lds=%zu  d_ub=%d tc=%d\n", n, m, N, npad, pl.WC, pl.MT, pl.NT, pl.ipw,
           pl.nblocks, pl.lds_bytes, d_ub, tc);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemset(w.stats, 0, 16));
        CK(hipEventRecord(e0));
        CK(launch_sweep_fused(nullptr, P, B, w, ty, pl, dZ, nullptr, 0, 0, 1, d_ub, tc, false, 1.1e-16));
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        int st[4];
        CK(hipMemcpy(st, w.stats, 16, hipMemcpyDeviceToHost));
        const double mean_terms = (double)st[2] / pl.nblocks;
        const double flops_useful = 2.0 * npad * npad * (m + 1) * (double)ty.T * K * (mean_terms - 1);
        const double flops_issued = 2.0 * npad * npad * (m + 1) * 16.0 * pl.NT * pl.WC * pl.nblocks * (mean_terms - 1);
        if (rep >= 3)
            printf("  %.3f ms  nonconv=%d max_terms=%d mean_terms=%.2f  useful %.1f TF/s, issued %.1f TF/s\n", ms, st[0], st[1],
                   mean_terms, flops_useful / ms * 1e-9, flops_issued / ms * 1e-9);
    }
    return 0;
}
