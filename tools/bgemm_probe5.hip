// Tuning probe #5 (round 3): the batched FP64 GEMM core as a RING of K panels filled by LDS-DMA for both operands.
// Not part of the product.  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bgemm_probe5.hip -o tools/bgemm_probe5
//
// What tools/stamp_analyze.py showed for the paired-rows core (V1): a workgroup alone on a CU keeps the matrix pipe 75 % busy --
// every K panel ends in  wait for the staged loads -> ds_write -> barrier -> fragment reads -> first MFMA,  a chain that all waves
// of the workgroup walk in step, so nothing covers it.  V2 removes the chain:
//   * panels of 8 k live in a ring of S slots (A: [k][128] rows, B: [n][8] columns with the 16-byte units XOR-swizzled so that
//     the b128 fragment reads are conflict-free); every slot is filled by global_load_lds (no staging registers, no ds_write);
//   * the panel that the barrier at the end of iteration q publishes is q+2, not q+1: the fragments of panel q+1 are read
//     DURING iteration q, into a second register set, between its MFMAs;
//   * the DMA pieces of panel q+S-1 are issued between the MFMAs as well; the only thing a wave does outside the MFMA stream is
//     s_waitcnt vmcnt(P) + s_barrier;
//   * persistent grid, and the ring runs on across tiles: the first panels of the next tile are in flight during the epilogue.
// usage: bgemm_probe5 [nbatch=2000] [reps=10] [square=0|1] [unused] [npad=256]
#define main probe4_main
#include "bgemm_probe4.hip"
#undef main

#include <type_traits>
// s_waitcnt immediate that waits for vmcnt <= n only (gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14)
constexpr int vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0f70; }

__device__ __forceinline__ void dma16(const double* g, double* l) {
    __builtin_amdgcn_global_load_lds(GLBP(g), LDSP(l), 16, 0, 0);
}

template <int WAVES, int S, int MINW>
__global__ void __launch_bounds__(64 * WAVES, MINW) k_v2(const double* A, const double* B, double* C, int npad, int nbatch, int dummy) {
    constexpr int TM = 128, TN = 128, KB = 8, WC = WAVES / 2;
    constexpr int WTM = 64, WTN = TN / WC, MT = 4, NT = WTN / 16;
    constexpr int SLOT = KB * TM;             // doubles per operand and slot (8 KB)
    constexpr int P = 16 / WAVES;             // DMA pieces (1 KB) per wave and panel: 8 rows of A + 8 column groups of B
    constexpr int AHEAD = S - 1;              // panels in flight ahead of the one being multiplied
    static_assert(S >= 3, "the panel after next must be in flight");
    __shared__ __attribute__((aligned(1024))) double smem[2 * S * SLOT];
    double* As = smem;
    double* Bs = smem + S * SLOT;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WC, wn = wave % WC, lr = lane & 15, lq = lane >> 4;
    const int tr_n = npad / TM, tpm = tr_n * (npad / TN);
    const int total = ((nbatch + 7) / 8) * 8 * tpm;
    const int64_t nn = (int64_t)npad * npad;
    const int nk = npad / KB;
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    // XOR swizzle of the 16-byte units of a B column (8 k = 4 units): unit u of column n sits at u ^ G[(n >> 2) & 3]
    const int gsw_r = (0x1230 >> (4 * ((lr >> 2) & 3))) & 3;        // G = {0, 3, 2, 1}
    const int a_off = wm * WTM + 2 * lr;                            // + k * TM + 32 p
    const int b_off = (wn * WTN + lr) * KB + 2 * (lq ^ gsw_r);      // + 16 tj * KB
    // DMA role of this wave: P of the 16 one-KB pieces of a panel -- waves of the first half fetch rows of A, the others column
    // groups of B (lane l of a B piece fetches the k pair that belongs in LDS unit l of the piece: the swizzle is applied at the source)
    const bool dma_a = wave < WAVES / 2;
    const int piece0 = (dma_a ? wave : wave - WAVES / 2) * P;       // first piece of this wave within its operand (0..7)
    const int dn = lane >> 2;                                       // column within a B piece
    const int gsw_d = (0x1230 >> (4 * ((dn >> 2) & 3))) & 3;
    const int dkp = (lane & 3) ^ gsw_d;
    const int64_t d_lane = dma_a ? (int64_t)piece0 * npad + 2 * lane : (int64_t)(piece0 * 16 + dn) * npad + 2 * dkp;   // + i * d_piece + kq * d_k
    const int64_t d_piece = dma_a ? npad : 16 * (int64_t)npad;
    const int64_t d_k = dma_a ? (int64_t)KB * npad : KB;
    double* const d_lds = (dma_a ? As : Bs) + piece0 * (dma_a ? TM : 16 * KB);   // + slot * SLOT + i * 128 doubles

    // tile list of this workgroup (persistent)
    auto decode = [&](int v, int& b, int& tr, int& tc) {
        const int xcd = v & 7, idx = v >> 3;
        b = (idx / tpm) * 8 + xcd;
        const int tile = idx % tpm;
        tr = tile % tr_n; tc = tile / tr_n;
        return b < nbatch;
    };
    auto operand = [&](int b, int tr, int tc) { return dma_a ? A + b * nn + (int64_t)tr * TM : B + b * nn + (int64_t)tc * TN * npad; };
    int v = blockIdx.x;
    int b = 0, tr = 0, tc = 0;
    while (v < total && !decode(v, b, tr, tc)) v += gridDim.x;
    if (v >= total) return;
    const double* Db = operand(b, tr, tc);        // the operand this wave fetches, current tile
    // the tile after this one: its first panels are requested while this one finishes (none left: the current one again, unused)
    int vn = v + gridDim.x, bn = 0, trn = 0, tcn = 0;
    while (vn < total && !decode(vn, bn, trn, tcn)) vn += gridDim.x;
    const double* Dbn = vn < total ? operand(bn, trn, tcn) : Db;

    auto issue_piece = [&](const double* Dp, int kq, int slot, int i) {   // piece i (0..P-1) of panel kq
        dma16(Dp + d_lane + i * d_piece + kq * d_k, d_lds + slot * SLOT + i * 128);
    };
    int slot = 0;   // slot of the panel being multiplied
    // prologue of the first tile: panels 0 .. AHEAD-1 requested
#pragma unroll
    for (int j = 0; j < AHEAD; ++j)
#pragma unroll
        for (int i = 0; i < P; ++i) issue_piece(Db, j, j, i);
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(P * (AHEAD - 2)));   // panels 0 and 1 are in LDS
    __builtin_amdgcn_s_barrier();

    d2 fb[2][NT], fa0[2][MT / 2], fa1[2][MT / 2];
    auto read_b = [&](int sl, int set) {
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) fb[set][tj] = *reinterpret_cast<const d2*>(Bs + sl * SLOT + b_off + 16 * tj * KB);
    };
    auto read_a = [&](int sl, int s, d2* dst) {
#pragma unroll
        for (int p = 0; p < MT / 2; ++p) dst[p] = *reinterpret_cast<const d2*>(As + sl * SLOT + (2 * lq + s) * TM + a_off + 32 * p);
    };
    read_b(0, 0);
    read_a(0, 0, fa0[0]);
    read_a(0, 1, fa1[0]);

    for (;;) {   // tiles
        d4 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = d4{0, 0, 0, 0};
        auto iteration = [&](int q, auto parity) {
            constexpr int cur = decltype(parity)::value, nxt = cur ^ 1;
            const int s_next = slot + 1 == S ? 0 : slot + 1;
            int s_fill = slot + AHEAD; if (s_fill >= S) s_fill -= S;       // slot of panel q + AHEAD (the one read in iteration q - 1)
            const bool in_tile = q + AHEAD < nk;
            const double* Dp = in_tile ? Db : Dbn;
            const int kq = in_tile ? q + AHEAD : q + AHEAD - nk;
            // group 0: k = 2 lq of the panel (all its fragments were read during the previous iteration); the pieces of panel
            // q + AHEAD are requested between its MFMAs
#pragma unroll
            for (int ti = 0; ti < MT; ++ti)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cur][tj].x, (ti & 1) ? fa0[cur][ti / 2].y : fa0[cur][ti / 2].x, acc[ti][tj], 0, 0, 0);
                    constexpr int GAP = (MT * NT) / P;
                    if ((ti * NT + tj) % GAP == GAP - 1) {   // (pinned: a DMA piece waits for the LDS reads issued before it)
                        __builtin_amdgcn_sched_barrier(0);
                        if (!(dummy & 2)) issue_piece(Dp, kq, s_fill, (ti * NT + tj) / GAP);   // (ablation 2: no DMA in the loop -- wrong results)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
            // group 1: k = 2 lq + 1; the first fragments of the next panel arrive meanwhile
            read_b(s_next, nxt);
            read_a(s_next, 0, fa0[nxt]);
            read_a(s_next, 1, fa1[nxt]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ti = 0; ti < MT; ++ti)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cur][tj].y, (ti & 1) ? fa1[cur][ti / 2].y : fa1[cur][ti / 2].x, acc[ti][tj], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // panel q + 2 has landed (everything but the last AHEAD-2 panels' pieces of this wave) and becomes visible to all
            if (!(dummy & 4)) {   // (ablation 4: no wait, no barrier -- wrong results)
                __builtin_amdgcn_s_waitcnt(vmcnt_imm(P * (AHEAD - 2)));
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            slot = s_next;
        };
        for (int q = 0; q < nk; q += 2) {
            iteration(q, std::integral_constant<int, 0>{});
            iteration(q + 1, std::integral_constant<int, 1>{});
        }
        // epilogue (plain stores).  The fragments of the next tile's first panel are already in fb[0] / fa0[0].
        const int row0 = tr * TM + wm * WTM + 2 * lr, col0 = tc * TN + wn * WTN + lq;
        double* Cb = C + b * nn;
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int p = 0; p < MT / 2; ++p) {
                    d2 vv = {acc[2 * p][tj][r], acc[2 * p + 1][tj][r]};
                    if (dummy < 0) {   // heavy epilogue
                        const int64_t o = (int64_t)(col0 + 16 * tj + 4 * r) * npad + row0 + 32 * p;
                        heavy_epilogue(vv, Cb + o, o, b * nn);
                        continue;
                    }
                    __builtin_nontemporal_store(vv, reinterpret_cast<d2*>(&Cb[(int64_t)(col0 + 16 * tj + 4 * r) * npad + row0 + 32 * p]));
                }
        if (vn >= total) break;
        v = vn; b = bn; tr = trn; tc = tcn; Db = Dbn;
        vn = v + gridDim.x;
        while (vn < total && !decode(vn, bn, trn, tcn)) vn += gridDim.x;
        Dbn = vn < total ? operand(bn, trn, tcn) : Db;
    }
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));   // no DMA piece may land after the workgroup has given its LDS back
    if (dummy == 1 && tid == 0) {   // clock of this workgroup's CU over the launch: s_memtime ticks per 10 ns of s_memrealtime
        unsigned long long* st = reinterpret_cast<unsigned long long*>(C + nn * nbatch) ;
        atomicAdd(st, __builtin_amdgcn_s_memtime() - clk0);
        atomicAdd(st + 1, __builtin_amdgcn_s_memrealtime() - rt0);
    }
}

template <class K>
void check(const char* name, K k, int threads, const double* A, const double* Bm, double* C, double* C2, int npad, int nb) {
    const size_t nn = (size_t)npad * npad;
    const int n = std::min(nb, 24);
    hipMemset(C2, 0, nn * n * 8);
    hipLaunchKernelGGL((k_v0<2>), dim3(512), dim3(256), 0, 0, A, Bm, C, npad, n);
    hipLaunchKernelGGL(k, dim3(40), dim3(threads), 0, 0, A, Bm, C2, npad, n, 0);   // 40 workgroups: several tiles each
    hipDeviceSynchronize();
    std::vector<double> c0(nn * n), c1(nn * n);
    hipMemcpy(c0.data(), C, c0.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c1.data(), C2, c1.size() * 8, hipMemcpyDeviceToHost);
    double e = 0, m = 0;
    for (size_t i = 0; i < c0.size(); ++i) { e = std::max(e, fabs(c0[i] - c1[i])); m = std::max(m, fabs(c0[i])); }
    printf("%-40s vs V0: max abs diff %.3e (max |C| %.3e)\n", name, e, m);
}

int main(int argc, char** argv) {
    const int npad = argc > 5 ? atoi(argv[5]) : 256, nb = argc > 1 ? atoi(argv[1]) : 2000;
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
    check("V2 4 waves, 4 slots", k_v2<4, 4, 2>, 256, A, Bm, C, C2, npad, nb);
    check("V2 8 waves, 4 slots", k_v2<8, 4, 4>, 512, A, Bm, C, C2, npad, nb);
    check("V2 4 waves, 3 slots", k_v2<4, 3, 2>, 256, A, Bm, C, C2, npad, nb);
    check("V2 4 waves, 8 slots", k_v2<4, 8, 1>, 256, A, Bm, C, C2, npad, nb);
    for (int rep = 0; rep < 2; ++rep) {
        run("V1 paired rows, DMA-A, 4 waves, 2 WG/CU (production)", k_v1<2, 1, 0>, 256, A, Bm, C, npad, nb, 2, reps, 0);
        run("V2 ring, 4 waves, 4 slots, 2 WG/CU", k_v2<4, 4, 2>, 256, A, Bm, C, npad, nb, 2, reps, 0);
        run("V2 ring, 4 waves, 3 slots, 2 WG/CU", k_v2<4, 3, 2>, 256, A, Bm, C, npad, nb, 2, reps, 0);
        run("V2 ring, 8 waves, 4 slots, 2 WG/CU", k_v2<8, 4, 4>, 512, A, Bm, C, npad, nb, 2, reps, 0);
        run("V2 ring, 4 waves, 4 slots, 1 WG/CU", k_v2<4, 4, 2>, 256, A, Bm, C, npad, nb, 1, reps, 0);
        {
            hipMemset(C + nn * nb, 0, 16);
            run("V2 ring, 4 waves, 3 slots, 2 WG/CU, clock stamps", k_v2<4, 3, 2>, 256, A, Bm, C, npad, nb, 2, reps, 1);
            unsigned long long st[2];
            hipMemcpy(st, C + nn * nb, 16, hipMemcpyDeviceToHost);
            printf("    in-kernel clock %.3f GHz\n", (double)st[0] / (double)st[1] * 0.1);
        }
        run("V2 ring, 4 waves, 8 slots, 1 WG/CU", k_v2<4, 8, 1>, 256, A, Bm, C, npad, nb, 1, reps, 0);
        run("V2 ring, 8 waves, 8 slots, 1 WG/CU", k_v2<8, 8, 2>, 512, A, Bm, C, npad, nb, 1, reps, 0);
        run("V1 paired rows, DMA-A, 4 waves, 1 WG/CU", k_v1<2, 1, 0>, 256, A, Bm, C, npad, nb, 1, reps, 0);
    }
    for (int abl : {0, 2, 4, 6}) {
        char name[96];
        snprintf(name, sizeof name, "V2 4 waves, 4 slots, 1 WG/CU, ablation %d (2: no DMA, 4: no barrier)", abl);
        run(name, k_v2<4, 4, 2>, 256, A, Bm, C, npad, nb, 1, reps, abl);
        snprintf(name, sizeof name, "V2 8 waves, 4 slots, 2 WG/CU, ablation %d", abl);
        run(name, k_v2<8, 4, 4>, 512, A, Bm, C, npad, nb, 2, reps, abl);
    }
    // the two-output polynomial product's epilogue on both cores: four more matrices streamed in, a second one out
    {
        HeavyEpi h{};
        for (int i = 0; i < 4; ++i) { double* m; hipMalloc(&m, nn * nb * 8); hipMemset(m, 0, nn * nb * 8); h.M[i] = m; }
        hipMalloc(&h.C2, nn * nb * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_heavy), &h, sizeof(h));
        printf("heavy epilogue (4 matrices in, 2 out):\n");
        for (int rep = 0; rep < 2; ++rep) {
            run("V1 register staging, 8 waves, one WG per tile (production)", k_v1<2, 0, 1, 4>, 512, A, Bm, C, npad, nb, 0, reps, -1);
            run("V1 register staging, 8 waves, 2 WG/CU persistent", k_v1<2, 0, 1, 4>, 512, A, Bm, C, npad, nb, 2, reps, -1);
            run("V1 DMA-A, 4 waves, 2 WG/CU persistent", k_v1<2, 1, 0>, 256, A, Bm, C, npad, nb, 2, reps, -1);
            run("V2 ring, 4 waves, 4 slots, 2 WG/CU", k_v2<4, 4, 2>, 256, A, Bm, C, npad, nb, 2, reps, -1);
            run("V2 ring, 4 waves, 3 slots, 2 WG/CU", k_v2<4, 3, 2>, 256, A, Bm, C, npad, nb, 2, reps, -1);
            run("V2 ring, 8 waves, 4 slots, 2 WG/CU", k_v2<8, 4, 4>, 512, A, Bm, C, npad, nb, 2, reps, -1);
            run("V2 ring, 8 waves, 3 slots, 2 WG/CU", k_v2<8, 3, 4>, 512, A, Bm, C, npad, nb, 2, reps, -1);
            run("V2 ring, 4 waves, 3 slots, 3 WG/CU", k_v2<4, 3, 2>, 256, A, Bm, C, npad, nb, 3, reps, -1);
        }
    }
    return 0;
}
