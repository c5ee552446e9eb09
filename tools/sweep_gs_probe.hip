// Stand-alone check + timing of the generator-stationary sweep (dto_sweep_gs.hip) against the single-workgroup fused form
// (dto_sweep_fused.hip) on the same inputs: builds the kernels' arguments directly (no engine), runs both, compares the sums S
// (and, in store mode, every stored term) word by word, prints times and statistics.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I directtrajopt.jl_amd/csrc tools/sweep_gs_probe.hip -o tools/sweep_gs_probe
// usage: sweep_gs_probe [n m N T(1 = p column only, else 1+m) d_ub store nt_override]
#include "dto_sweep_fused.hip"
#include "dto_sweep_gs.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

using namespace dto;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256, m = argc > 2 ? atoi(argv[2]) : 4, N = argc > 3 ? atoi(argv[3]) : 2000;
    const int Targ = argc > 4 ? atoi(argv[4]) : 1;
    const int d_ub = argc > 5 ? atoi(argv[5]) : 30;
    const bool store = argc > 6 && atoi(argv[6]) != 0;
    const int nt_over = argc > 7 ? atoi(argv[7]) : 0;
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
    ty.T = Targ == 1 ? 1 : 1 + m;
    for (int j = 0; j + 1 < ty.T; ++j) { ty.t[1 + j].n_extra = 1; ty.t[1 + j].gen[0] = 1 + j; ty.t[1 + j].src[0] = 0; ty.t[1 + j].mult[0] = 1.0; }
    const int dcap = d_ub + 2;
    auto make_buf = [&](SweepBuf& w) {
        w = SweepBuf{};
        w.npad = npad; w.TN = 128; w.Kpad = (K + 127) / 128 * 128; w.T_alloc = ty.T; w.dcap = dcap;
        const size_t typesz = (size_t)w.Kpad * npad;
        for (int i = 0; i < 2; ++i) { CK(hipMalloc(&w.Z[i], typesz * ty.T * 8)); CK(hipMemset(w.Z[i], 0, typesz * ty.T * 8)); }
        CK(hipMalloc(&w.S, typesz * ty.T * 8)); CK(hipMemset(w.S, 0, typesz * ty.T * 8));
        if (store) { CK(hipMalloc(&w.Zt, typesz * ty.T * 8 * dcap)); CK(hipMemset(w.Zt, 0, typesz * ty.T * 8 * dcap)); }
        CK(hipMalloc(&w.scaleA, (m + 1) * (size_t)w.Kpad * 8)); CK(hipMalloc(&w.scaleU, (m + 1) * (size_t)w.Kpad * 8));
        CK(hipMalloc(&w.scaleE, 2 * (size_t)w.Kpad * 8));
        CK(hipMalloc(&w.stats, 16));
        CK(hipMalloc(&w.nterms, 4 * (size_t)w.Kpad)); CK(hipMemset(w.nterms, 0, 4 * (size_t)w.Kpad));
    };
    SweepBuf wf, wg;
    make_buf(wf); make_buf(wg);
    const size_t typesz = (size_t)wf.Kpad * npad;
    CK(sweep_fused_prepare());
    CK(sweep_gs_prepare());
    const int tc = d_ub / 2 - 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops_term = 2.0 * npad * npad * (m + 1) * (double)ty.T * K;

    FusedSweepPlan fp{};
    bool have_fused = sweep_fused_plan(npad, m, ty, K, 256, fp, false);
    if (!have_fused && npad == 256) {
        // too few workgroups for the planner's taste (single-column sweeps, short shards): still a valid reference
        const int ipw = std::max(1, 16 / ty.T);
        fp = FusedSweepPlan{};
        fp.MT = 4; fp.NT = 1; fp.WC = 1; fp.WK = 1; fp.ipw = ipw; fp.nslot = 0; fp.nblocks = (K + ipw - 1) / ipw;
        fp.lds_bytes = (size_t)FusedLds(npad, ty.T, m, ipw, 0, 4).total * 8;
        have_fused = true;
    }
    if (have_fused) {
        printf("fused plan: MT=%d NT=%d WC=%d WK=%d ipw=%d blocks=%d\n", fp.MT, fp.NT, fp.WC, fp.WK, fp.ipw, fp.nblocks);
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(wf.stats, 0, 16));
            CK(hipEventRecord(e0));
            CK(launch_sweep_fused(nullptr, P, B, wf, ty, fp, dZ, nullptr, 0, 0, 1, d_ub, tc, store, 1.1e-16));
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int st[4]; CK(hipMemcpy(st, wf.stats, 16, hipMemcpyDeviceToHost));
            if (rep >= 2) printf("  fused  %.3f ms  nonconv=%d max_terms=%d  (%.1f TF/s at max_terms)\n", ms, st[0], st[1], flops_term * (st[1] - 1) / ms * 1e-9);
        }
    }
    GsSweepPlan gp{};
    if (!sweep_gs_plan(npad, m, ty, K, 256, gp)) { printf("no gs plan\n"); return 1; }
    if (nt_over > 0) {
        gp.NT = nt_over; gp.ipw = 16 * nt_over / ty.T; gp.n_groups = (K + gp.ipw - 1) / gp.ipw;
        gp.n_clusters = std::min(32, (gp.n_groups + 7) / 8 * 8); gp.nblocks = gp.n_clusters * gp.KU;
        gp.cap = (gp.n_groups + gp.n_clusters - 1) / gp.n_clusters * gp.ipw;
        gp.lds_bytes = gs_lds_bytes(gp.KU, gp.MP, gp.NT, gp.cap);
    }
    printf("gs plan: KU=%d MP=%d NT=%d ipw=%d has_src=%d groups=%d clusters=%d blocks=%d lds=%zu  model %.1f us/term\n", gp.KU, gp.MP, gp.NT, gp.ipw,
           gp.has_src, gp.n_groups, gp.n_clusters, gp.nblocks, gp.lds_bytes, gp.term_us);
    double* Xn; unsigned* arrive;
    CK(hipMalloc(&Xn, sweep_gs_norm_doubles(gp) * 8));
    CK(hipMalloc(&arrive, 4 * (size_t)((gp.n_groups + 3) / 4 * 4)));
#ifdef GS_STAMP
    unsigned long long* d_stamp;
    CK(hipMalloc(&d_stamp, 64 * 8 * 8)); CK(hipMemset(d_stamp, 0, 64 * 8 * 8));
    gs_stamp_buffer = d_stamp;
#endif
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(wg.stats, 0, 16));
        CK(hipEventRecord(e0));
        CK(launch_sweep_gs(nullptr, P, B, wg, ty, gp, Xn, arrive, dZ, nullptr, 0, 0, d_ub, tc, store, 1.1e-16));
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int st[4]; CK(hipMemcpy(st, wg.stats, 16, hipMemcpyDeviceToHost));
        if (rep >= 2) printf("  gs     %.3f ms  nonconv=%d max_terms=%d  (%.1f TF/s at max_terms, %.2f us/term)\n", ms, st[0], st[1],
                             flops_term * (st[1] - 1) / ms * 1e-9, ms * 1e3 / (st[1] - 1));
    }
#ifdef GS_STAMP
    {
        std::vector<unsigned long long> st(64 * 8);
        CK(hipMemcpy(st.data(), d_stamp, st.size() * 8, hipMemcpyDeviceToHost));
        printf("phase stamps of block 0 (cycles): item: B1 issue mfma land+B2 partials+B3 reduce+publish drain+signal | total\n");
        for (int i = 4; i < 24; ++i) {
            if (!st[i * 8 + 7]) break;
            printf("  item %2d:", i);
            for (int k = 1; k < 8; ++k) printf(" %6llu", st[i * 8 + k] - st[i * 8 + k - 1]);
            printf(" | %6llu   gap to next %6llu\n", st[i * 8 + 7] - st[i * 8], st[(i + 1) * 8] ? st[(i + 1) * 8] - st[i * 8 + 7] : 0ull);
        }
    }
#endif
    if (have_fused) {
        // compare the sums (and stored terms): both forms stop per group on the same test but with different groupings, so terms
        // below 1e-16 of the sum may differ -- compare relative to the column's scale
        std::vector<double> a(typesz * ty.T), b(typesz * ty.T);
        CK(hipMemcpy(a.data(), wf.S, a.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), wg.S, b.size() * 8, hipMemcpyDeviceToHost));
        double worst = 0.0, scale = 0.0; size_t nbad = 0;
        for (int t = 0; t < ty.T; ++t)
            for (int k = 0; k < K; ++k) {
                double cs = 0.0;
                for (int r = 0; r < n; ++r) cs = std::max(cs, std::fabs(a[((size_t)t * wf.Kpad + k) * npad + r]));
                for (int r = 0; r < npad; ++r) {
                    const size_t i = ((size_t)t * wf.Kpad + k) * npad + r;
                    const double d = std::fabs(a[i] - b[i]) / std::max(cs, 1e-300);
                    if (!(d <= 1e-12)) ++nbad;
                    if (d > worst || d != d) worst = d;
                }
                scale = std::max(scale, cs);
            }
        printf("S: worst column-relative difference gs vs fused %.3e (entries beyond 1e-12: %zu), max |S| %.3e\n", worst, nbad, scale);
        if (store) {
            int stf[4], stg[4];
            CK(hipMemcpy(stf, wf.stats, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(stg, wg.stats, 16, hipMemcpyDeviceToHost));
            const int nt = std::min(stf[1], stg[1]) - 2;
            std::vector<double> ta(typesz * ty.T), tb(typesz * ty.T);
            double tw = 0.0;
            for (int t = 0; t < nt; t += 3) {
                CK(hipMemcpy(ta.data(), wf.Zt + (size_t)t * ty.T * typesz, ta.size() * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(tb.data(), wg.Zt + (size_t)t * ty.T * typesz, tb.size() * 8, hipMemcpyDeviceToHost));
                double mx = 0.0, df = 0.0;
                for (size_t i = 0; i < ta.size(); ++i) { mx = std::max(mx, std::fabs(ta[i])); df = std::max(df, std::fabs(ta[i] - tb[i])); }
                tw = std::max(tw, df / std::max(mx, 1e-300));
            }
            printf("stored terms (every third, 0..%d): worst difference / max |term| %.3e\n", nt, tw);
        }
    }
    return 0;
}
