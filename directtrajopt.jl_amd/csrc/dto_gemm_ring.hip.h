// dto_gemm_ring.hip.h -- FP64 MFMA GEMM core over a STREAM of 128 x 128 tiles, K panels in an LDS ring (gfx950).
//
// What the phase stamps of the double-buffered core showed: every K panel ends in the chain
//   wait for the staged loads -> ds_write -> barrier -> fragment reads -> first MFMA
// which all waves of a workgroup walk in step, so a workgroup alone keeps the matrix pipe 75 % busy, and every tile starts with an
// exposed round trip for its first panel.  This core removes both:
//   * a persistent workgroup walks its tile list; the K panels of ALL its tiles form one stream of 8-deep panels;
//   * panel g of the stream lives in slot g mod S of an LDS ring (A: [k][128 rows], B: [128 columns][8 k] with the 16-byte units of a
//     column XOR-swizzled so that the b128 fragment reads are conflict-free); both operands arrive by global_load_lds -- no staging
//     registers, no ds_write -- with the swizzle applied at the source address;
//   * the pieces of panel g + S - 1 are requested between the MFMAs of panel g, so the first panels of the NEXT tile are in flight
//     while a tile finishes and during its epilogue;
//   * the barrier at the end of iteration g publishes panel g + 2 (s_waitcnt vmcnt(P (S-3)) -- all but the newest pieces -- then
//     s_barrier): the fragments of panel g + 1 are read during iteration g into a second register set, between its MFMAs
//     (a tile's first fragments after the epilogue of the tile before).
// Accumulator layout = the paired-rows core's (GemmCoordP / GemmAccS<GemmShapeP<128, 128, 2, WAVES / 2>>); same k order within a tile.
// Measured (tools/bgemm_probe5.hip, 8 waves, 4 slots, 2 workgroups per CU): 256 x 256 x 256 squaring 64-65.5 TFLOP/s against 57-62 for
// the double-buffered core; 1024^3 70.8 against 66-67, and with the two-output polynomial epilogue attached 69.9 against 63.5.
#pragma once
#include "dto_gemm.hip.h"

#include <type_traits>

namespace dto {

// s_waitcnt immediate that waits for vmcnt <= n only (gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14)
constexpr int ring_vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0f70; }

struct RingTile {
    const double* A;   // tile's first row, k = 0: element (row, k) at A[k * lda + row], 128 rows contiguous
    int64_t lda;
    const double* B;   // tile's first column, k = 0: element (k, col) at B[col * ldb + k]
    int64_t ldb;
    int nk;            // K / 8, even
};

template <int WAVES, int S>
struct RingCore {
    static constexpr int TM = 128, TN = 128, KB = 8, WC = WAVES / 2;
    static constexpr int SLOT = KB * TM;         // doubles per operand and slot (8 KB)
    static constexpr int P = 16 / WAVES;         // DMA pieces (1 KB) per wave and panel: 8 rows of A + 8 column groups of B
    static constexpr int AHEAD = S - 1;          // panels requested ahead of the one being multiplied
    static constexpr int SMEM_DOUBLES = 2 * S * SLOT;
    static constexpr int THREADS = 64 * WAVES;
    using Cfg = GemmShapeP<128, 128, 2, WC>;     // accumulator / epilogue layout
    static_assert(S >= 3 && (WAVES == 4 || WAVES == 8), "ring of at least three panels; 4 or 8 waves");

    // The workgroup's tiles are the valid ones among first, first + stride, ...:
    //   next(pos, tile) -> bool : moves pos forward (by stride) to the first valid tile at or after it and describes it (false: none);
    //   epi(pos, acc)           : consumes the accumulators of the tile at pos.
    template <class Next, class Epi>
    static __device__ __forceinline__ void run(double* smem, int first, int stride, Next next, Epi epi) {
        constexpr int MT = Cfg::MT, NT = Cfg::NT;
        double* As = smem;
        double* Bs = smem + S * SLOT;
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wm = wave / WC, wn = wave % WC, lr = lane & 15, lq = lane >> 4;
        // XOR swizzle of the 16-byte units of a B column (8 k = 4 units): unit u of column n sits at u ^ G[(n >> 2) & 3], G = {0,3,2,1}
        // (brute-forced against the ds_read_b128 lane groups of MI355X_MICROARCH.md, LDS)
        const int gsw_r = (0x1230 >> (4 * ((lr >> 2) & 3))) & 3;
        const int a_off = wm * Cfg::WTM + 2 * lr;                          // + k * TM + 32 p
        const int b_off = (wn * Cfg::WTN + lr) * KB + 2 * (lq ^ gsw_r);    // + 16 tj * KB
        // DMA role: the first half of the waves fetch rows of A, the second half column groups of B (lane l of a B piece fetches the
        // k pair that belongs in LDS unit l of the piece)
        const bool dma_a = wave < WAVES / 2;
        const int piece0 = (dma_a ? wave : wave - WAVES / 2) * P;
        const int dn = lane >> 2;
        const int dkp = (lane & 3) ^ ((0x1230 >> (4 * ((dn >> 2) & 3))) & 3);
        double* const d_lds = (dma_a ? As : Bs) + piece0 * (dma_a ? TM : 16 * KB);   // + slot * SLOT + i * 128

        // fetch cursor: (tile at fpos, panel fq), AHEAD panels ahead of the multiplication
        RingTile ft;
        int fpos = first;
        if (!next(fpos, ft)) return;
        int fq = 0;
        const double* f_base;     // + i * f_piece + fq * f_k
        int64_t f_piece, f_k;
        auto aim = [&]() {
            if (dma_a) { f_base = ft.A + (int64_t)piece0 * ft.lda + 2 * lane; f_piece = ft.lda; f_k = (int64_t)KB * ft.lda; }
            else { f_base = ft.B + (int64_t)(piece0 * 16 + dn) * ft.ldb + 2 * dkp; f_piece = 16 * ft.ldb; f_k = KB; }
        };
        aim();
        auto issue_piece = [&](int slot, int i) {
            __builtin_amdgcn_global_load_lds(DTO_GLB_PTR(f_base + i * f_piece + fq * f_k), DTO_LDS_PTR(d_lds + slot * SLOT + i * 128), 16, 0, 0);
        };
        auto advance_fetch = [&]() {   // past the end of the list the last panel is requested again (into a slot nobody reads any more)
            if (fq + 1 < ft.nk) { ++fq; return; }
            RingTile nt;
            int np = fpos + stride;
            if (next(np, nt)) { fpos = np; ft = nt; fq = 0; aim(); }
        };
        int slot = 0;                 // slot of the panel being multiplied
        int s_fill = 0;               // slot the fetch cursor fills next
#pragma unroll
        for (int j = 0; j < AHEAD; ++j) {
#pragma unroll
            for (int i = 0; i < P; ++i) issue_piece(s_fill, i);
            advance_fetch();
            s_fill = s_fill + 1 == S ? 0 : s_fill + 1;
        }
        __builtin_amdgcn_s_waitcnt(ring_vmcnt_imm(P * (AHEAD - 2)));   // the first two panels are in LDS
        __builtin_amdgcn_s_barrier();

        d2 fb[2][NT], fa0[2][MT / 2], fa1[2][MT / 2];
        auto read_frags = [&](int sl, int set) {
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) fb[set][tj] = *reinterpret_cast<const d2*>(Bs + sl * SLOT + b_off + 16 * tj * KB);
#pragma unroll
            for (int p = 0; p < MT / 2; ++p) {
                fa0[set][p] = *reinterpret_cast<const d2*>(As + sl * SLOT + (2 * lq) * TM + a_off + 32 * p);
                fa1[set][p] = *reinterpret_cast<const d2*>(As + sl * SLOT + (2 * lq + 1) * TM + a_off + 32 * p);
            }
        };
        read_frags(0, 0);

        RingTile ct;
        for (int cpos = first; next(cpos, ct); cpos += stride) {
            GemmAccS<Cfg> acc;
            acc.zero();
            auto iteration = [&](auto parity, bool last) {
                constexpr int cur = decltype(parity)::value, nxt = cur ^ 1;
                const int s_next = slot + 1 == S ? 0 : slot + 1;
                // group 0: k = 2 lq of the panel (its fragments were read during the previous iteration); the pieces of the panel
                // AHEAD of this one are requested between the MFMAs (pinned: a DMA piece waits for the LDS reads issued before it)
#pragma unroll
                for (int ti = 0; ti < MT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) {
                        acc.v[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cur][tj].x, (ti & 1) ? fa0[cur][ti / 2].y : fa0[cur][ti / 2].x,
                                                                            acc.v[ti][tj], 0, 0, 0);
                        constexpr int GAP = (MT * NT) / P;
                        if ((ti * NT + tj) % GAP == GAP - 1) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue_piece(s_fill, (ti * NT + tj) / GAP);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                advance_fetch();
                s_fill = s_fill + 1 == S ? 0 : s_fill + 1;
                __builtin_amdgcn_sched_barrier(0);
                // group 1: k = 2 lq + 1; the fragments of the tile's next panel arrive meanwhile (not across the epilogue: a tile's
                // first fragments are read after the epilogue of the one before, which keeps their registers free for it)
                if (!last) read_frags(s_next, nxt);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ti = 0; ti < MT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj)
                        acc.v[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cur][tj].y, (ti & 1) ? fa1[cur][ti / 2].y : fa1[cur][ti / 2].x,
                                                                            acc.v[ti][tj], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // the panel after next has landed (everything but this wave's newest AHEAD-2 panels' pieces) and becomes visible
                __builtin_amdgcn_s_waitcnt(ring_vmcnt_imm(P * (AHEAD - 2)));
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                slot = s_next;
            };
            for (int q = 0; q < ct.nk; q += 2) {
                iteration(std::integral_constant<int, 0>{}, false);
                iteration(std::integral_constant<int, 1>{}, q + 2 >= ct.nk);
            }
            epi(cpos, acc);
            read_frags(slot, 0);   // first panel of the next tile (landed and published two barriers ago)
        }
        __builtin_amdgcn_s_waitcnt(ring_vmcnt_imm(0));   // no DMA piece may land after the workgroup has given its LDS back
    }
};

}  // namespace dto
