// dto_gemm.hip.h -- workgroup-level FP64 MFMA GEMM core for gfx950 (CDNA4).
//
// One 256-thread workgroup (4 wavefronts of 64, one per SIMD) computes a TM x TN tile of
// C = A * B for COLUMN-MAJOR operands.  The wave grid is 2 x 2, each wave owns a
// (TM/2) x (TN/2) sub-tile made of 16x16 `v_mfma_f64_16x16x4_f64` accumulators.
//
// The product is issued TRANSPOSED (MFMA a-operand <- B fragment, b-operand <- A fragment) so
// that the accumulator's lane index runs along C's rows: lane l, register r of accumulator
// (ti,tj) holds C[row0 + 16*ti + (l&15)][col0 + 16*tj + (l>>4) + 4*r].  Sixteen consecutive lanes
// therefore touch 128 contiguous bytes of a column-major C, which is what the epilogues store.
//
// LDS staging (double buffered, one barrier per 16-deep K panel):
//   As[k][m]  : KB rows of TM doubles, row pitch TM+16  (2*(TM+16) mod 64 == 32 -> the two k's of a
//               32-lane ds_read_b64 group land on disjoint bank halves: conflict-free)
//   Bs[n][k]  : TN rows of KB doubles, row pitch KB+2   (36*c mod 64 distinct for c<16: conflict-free)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dto {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// Shape of one workgroup-level GEMM: TM x TN tile, WR x WC wavefronts (64 lanes each), KB-deep K panel.
template <int TM_, int TN_, int WR_ = 2, int WC_ = 2, int KB_ = 16>
struct GemmShape {
    static constexpr int TM = TM_, TN = TN_, WR = WR_, WC = WC_, KB = KB_;
    static constexpr bool PAIRED = false;
    static constexpr int THREADS = WR * WC * 64;
    static constexpr int LDA_S = TM + 16;  // (TM+16) mod 32 == 16 for TM multiple of 32
    static constexpr int LDB_S = KB + 2;
    static constexpr int AS_ELEMS = KB * LDA_S;
    static constexpr int BS_ELEMS = TN * LDB_S;
    static constexpr int SMEM_DOUBLES = 2 * (AS_ELEMS + BS_ELEMS);
    static constexpr int WTM = TM / WR, WTN = TN / WC;  // per-wave tile
    static constexpr int MT = WTM / 16;                 // accumulator tiles per wave along rows
    static constexpr int NT = WTN / 16;                 // along columns
    static constexpr int A_LD = (TM * KB / 2) / THREADS;  // double2 loads per thread per panel
    static constexpr int B_LD = (TN * KB / 2) / THREADS;
    static_assert(TM % (16 * WR) == 0 && TN % (16 * WC) == 0, "wave tile must be a multiple of 16");
    static_assert((TM * KB / 2) % THREADS == 0 && (TN * KB / 2) % THREADS == 0, "panel loads must divide evenly");
    static_assert(KB % 4 == 0, "K panel is a multiple of the MFMA depth");
};
template <int TM, int TN>
using GemmCfg = GemmShape<TM, TN, 2, 2, 16>;

template <class S>
struct GemmAccS {
    d4 v[S::MT][S::NT];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < S::MT; ++i)
#pragma unroll
            for (int j = 0; j < S::NT; ++j) v[i][j] = d4{0.0, 0.0, 0.0, 0.0};
    }
};
template <int TM, int TN>
using GemmAcc = GemmAccS<GemmCfg<TM, TN>>;

// acc += A[0:TM, 0:Klen] * diag-scaled B[0:Klen, 0:TN]
//   A      : pointer to the tile's first row, column 0 of the K range (column-major, lda)
//   B      : pointer to row 0 of the K range, the tile's first column (column-major, ldb)
//   colscale : nullptr, or TN per-column factors applied to B (b[k][n] *= colscale[n])
//   Klen   : multiple of KB
//   TWO    : the staged operand is colscale .* B + colscale2 .* B2 (same shape and ldb): two products with the same left
//            operand for the price of one (the sweep's G_j (c_j d^j) + G_j (dt p))
// All threads of the workgroup must call it; it ends with a barrier so LDS may be reused immediately.
template <class C, bool TWO = false>
__device__ __forceinline__ void gemm_accumulate_s(GemmAccS<C>& acc, const double* __restrict__ A, int lda,
                                                  const double* __restrict__ B, int ldb, int Klen,
                                                  const double* __restrict__ colscale, double* smem,
                                                  const double* __restrict__ B2 = nullptr,
                                                  const double* __restrict__ colscale2 = nullptr) {
    constexpr int TM = C::TM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / C::WC, wn = wave % C::WC;
    const int lr = lane & 15, lq = lane >> 4;

    double* As = smem;
    double* Bs = smem + 2 * C::AS_ELEMS;

    d2 ra[C::A_LD], rb[C::B_LD], rb2[TWO ? C::B_LD : 1];
    int a_k[C::A_LD], a_m[C::A_LD], b_n[C::B_LD], b_k[C::B_LD];
    double bsc[C::B_LD], bsc2[TWO ? C::B_LD : 1];
#pragma unroll
    for (int i = 0; i < C::A_LD; ++i) {
        const int idx = tid + C::THREADS * i;
        a_k[i] = idx / (TM / 2);
        a_m[i] = 2 * (idx % (TM / 2));
    }
#pragma unroll
    for (int i = 0; i < C::B_LD; ++i) {
        const int idx = tid + C::THREADS * i;
        b_n[i] = idx / (C::KB / 2);
        b_k[i] = 2 * (idx % (C::KB / 2));
        bsc[i] = colscale ? colscale[b_n[i]] : 1.0;
        if constexpr (TWO) bsc2[i] = colscale2[b_n[i]];
    }

    const int nkb = Klen / C::KB;

    auto load_panel = [&](int kb) {
        const int k0 = kb * C::KB;
#pragma unroll
        for (int i = 0; i < C::A_LD; ++i)
            ra[i] = *reinterpret_cast<const d2*>(A + (size_t)(k0 + a_k[i]) * lda + a_m[i]);
#pragma unroll
        for (int i = 0; i < C::B_LD; ++i) {
            rb[i] = *reinterpret_cast<const d2*>(B + (size_t)b_n[i] * ldb + k0 + b_k[i]);
            if constexpr (TWO) rb2[i] = *reinterpret_cast<const d2*>(B2 + (size_t)b_n[i] * ldb + k0 + b_k[i]);
        }
    };
    auto store_panel = [&](int buf) {
        double* as = As + buf * C::AS_ELEMS;
        double* bs = Bs + buf * C::BS_ELEMS;
#pragma unroll
        for (int i = 0; i < C::A_LD; ++i)
            *reinterpret_cast<d2*>(as + a_k[i] * C::LDA_S + a_m[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < C::B_LD; ++i) {
            d2 v = rb[i];
            v.x *= bsc[i];
            v.y *= bsc[i];
            if constexpr (TWO) {
                v.x += bsc2[i] * rb2[i].x;
                v.y += bsc2[i] * rb2[i].y;
            }
            *reinterpret_cast<d2*>(bs + b_n[i] * C::LDB_S + b_k[i]) = v;
        }
    };

    load_panel(0);
    store_panel(0);
    __syncthreads();

    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) load_panel(kb + 1);
        const double* as = As + buf * C::AS_ELEMS + wm * C::WTM + lr;
        const double* bs = Bs + buf * C::BS_ELEMS + (wn * C::WTN + lr) * C::LDB_S;
#pragma unroll
        for (int kk = 0; kk < C::KB; kk += 4) {
            // LLVM's MFMA/DS interleaving strategy for small GEMM loops: measured -2.5 % on the fused-polynomial GEMM and
            // -2..5 % on the sweep step; the DMA-staged loop below is faster without it (+8 % with it)
            __builtin_amdgcn_iglp_opt(0);
            double af[C::MT], bf[C::NT];
#pragma unroll
            for (int ti = 0; ti < C::MT; ++ti) af[ti] = as[(kk + lq) * C::LDA_S + 16 * ti];
#pragma unroll
            for (int tj = 0; tj < C::NT; ++tj) bf[tj] = bs[16 * tj * C::LDB_S + kk + lq];
#ifdef DTO_GEMM_SETPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int ti = 0; ti < C::MT; ++ti)
#pragma unroll
                for (int tj = 0; tj < C::NT; ++tj)
                    acc.v[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[tj], af[ti], acc.v[ti][tj], 0, 0, 0);
#ifdef DTO_GEMM_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        }
        if (kb + 1 < nkb) store_panel(buf ^ 1);
        __syncthreads();
    }
}
// DMA-staged form (operands unscaled): the K panels go global -> LDS directly with `global_load_lds_dwordx4`
// (1 KB per wave-instruction, no staging registers, no ds_write).  The LDS images are unpadded because a DMA
// piece must be wave-contiguous; conflicts are avoided by the SOURCE permutation instead:
//   As[k][m'] , m' = (m + 16 (k & 1)) mod TM   -- the two k's of a 32-lane ds_read_b64 group land 128 B apart
//   Bs[kp][c][2]                                -- the two k's of a k-pair are adjacent: a group reads 256 contiguous B
// Requires TM = TN = 128, KB = 16 (A: 16 pieces/panel, B: 16 pieces/panel, spread over the workgroup's waves).
#define DTO_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define DTO_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
template <class C>
__device__ __forceinline__ void gemm_accumulate_dma(GemmAccS<C>& acc, const double* __restrict__ A, int lda,
                                                    const double* __restrict__ B, int ldb, int Klen, double* smem) {
    static_assert(C::TM == 128 && C::TN == 128 && C::KB == 16, "DMA staging is laid out for 128x128x16 panels");
    constexpr int TM = 128, TN = 128, KB = 16, AS = KB * TM, BS = KB * TN;
    constexpr int NW = C::THREADS / 64, PPW = 16 / NW;  // pieces per wave and operand
    static_assert(2 * (AS + BS) <= C::SMEM_DOUBLES, "LDS budget");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WC, wn = wave % C::WC;
    const int lr = lane & 15, lq = lane >> 4;
    double* As = smem;
    double* Bs = smem + 2 * AS;
    auto dma_panel = [&](int kb, int buf) {
        const int k0 = kb * KB;
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int k = wave + NW * q;
            const int m = (2 * lane - 16 * (k & 1)) & (TM - 1);
            __builtin_amdgcn_global_load_lds(DTO_GLB_PTR(A + (size_t)(k0 + k) * lda + m), DTO_LDS_PTR(As + buf * AS + k * TM), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int piece = wave + NW * q;
            const int g = piece * 64 + lane;
            const int kp = g / TN, c = g % TN;
            __builtin_amdgcn_global_load_lds(DTO_GLB_PTR(B + (size_t)c * ldb + k0 + 2 * kp), DTO_LDS_PTR(Bs + buf * BS + piece * 128), 16, 0, 0);
        }
    };
    const int nkb = Klen / KB;
    dma_panel(0, 0);
    __syncthreads();  // its fence waits for the DMA (vmcnt(0)) before the barrier
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) dma_panel(kb + 1, buf ^ 1);
        const double* as = As + buf * AS;
        const double* bs = Bs + buf * BS;
#pragma unroll
        for (int kk = 0; kk < KB; kk += 4) {
            const int k = kk + lq;
            double af[C::MT], bf[C::NT];
#pragma unroll
            for (int ti = 0; ti < C::MT; ++ti) af[ti] = as[k * TM + ((wm * C::WTM + 16 * ti + lr + 16 * (k & 1)) & (TM - 1))];
#pragma unroll
            for (int tj = 0; tj < C::NT; ++tj) bf[tj] = bs[(k >> 1) * (TN * 2) + (wn * C::WTN + 16 * tj + lr) * 2 + (k & 1)];
#pragma unroll
            for (int ti = 0; ti < C::MT; ++ti)
#pragma unroll
                for (int tj = 0; tj < C::NT; ++tj)
                    acc.v[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[tj], af[ti], acc.v[ti][tj], 0, 0, 0);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// "Paired rows" core (round 3): 16-byte LDS fragment reads and 16-byte epilogue accesses.
//
// Accumulator tile ti of a wave covers the rows 32 (ti/2) + 2 lr + (ti & 1) of the wave tile instead of 16 ti + lr, so the
// two tiles 2p, 2p+1 of a lane hold two CONSECUTIVE rows: one ds_read_b128 feeds both A fragments, and an epilogue moves 16
// bytes per lane -- 256 contiguous bytes per 16 lanes of a column-major C (half the vector-memory instructions of the 8-byte
// form; cdna_hip_programming.md T21).  The B fragments of TWO k-steps come from one ds_read_b128 as well: the 8 k's of a
// double step run in the order {0,2,4,6}, {1,3,5,7} -- lane group lq supplies k = 2 lq + s in step s, for both operands, and
// a sum over k does not care.  LDS images (double buffered, one barrier per 16-deep panel):
//   As[k][TM]      unpadded, pitch TM*8 = a multiple of 256 B: the b128 lane groups {0-3,12-15,20-27}, ... of two adjacent k
//                  rows cover all 64 banks once; a K row is one wave-contiguous KB piece, so A can come by LDS-DMA
//                  (`global_load_lds_dwordx4`, no staging registers) exactly as it lies in memory
//   Bs[n][KB + 4]  pitch 160 B: conflict-free for the b128 reads (16-byte slot = 10 n + lq mod 16 is a bijection on every
//                  lane group) and for the staging stores (8 consecutive lanes = 8 consecutive slots)
// Measured (tools/bgemm_probe4, 2000 x 256^3, one box, steady state): 8-byte core 59.4-60.0 TFLOP/s, this core 60.5-61.1
// with register staging and 62.2-62.5 with A by DMA; rocBLAS (torch.bmm) 52 / 61 (product / squaring) on another box.
template <int TM_, int TN_, int WR_ = 2, int WC_ = 2>
struct GemmShapeP {
    static constexpr int TM = TM_, TN = TN_, WR = WR_, WC = WC_, KB = 16;
    static constexpr bool PAIRED = true;
    static constexpr int THREADS = WR * WC * 64;
    static constexpr int LDB_S = KB + 4;
    static constexpr int AS_ELEMS = KB * TM;
    static constexpr int BS_ELEMS = TN * LDB_S;
    static constexpr int SMEM_DOUBLES = 2 * (AS_ELEMS + BS_ELEMS);
    static constexpr int WTM = TM / WR, WTN = TN / WC;
    static constexpr int MT = WTM / 16, NT = WTN / 16;
    static constexpr int A_LD = (TM * KB / 2) / THREADS, B_LD = (TN * KB / 2) / THREADS;
    static constexpr int A_PIECES = KB / (THREADS / 64);  // DMA pieces (K rows) per wave and panel
    static_assert(TM % 32 == 0 && WTM % 32 == 0 && WTN % 16 == 0, "wave tile: pairs of 16-row tiles");
    static_assert((TM * KB / 2) % THREADS == 0 && (TN * KB / 2) % THREADS == 0 && KB % (THREADS / 64) == 0, "panel loads must divide evenly");
    static_assert(TM == 128, "a DMA piece is one K row of 128 doubles");
};

template <class C, bool DMA_A>
__device__ __forceinline__ void gemm_accumulate_p(GemmAccS<C>& acc, const double* __restrict__ A, int lda,
                                                  const double* __restrict__ B, int ldb, int Klen, double* smem,
                                                  unsigned long long* prof = nullptr) {
    constexpr int TM = C::TM, KB = C::KB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WC, wn = wave % C::WC;
    const int lr = lane & 15, lq = lane >> 4;
    double* As = smem;
    double* Bs = smem + 2 * C::AS_ELEMS;
    d2 ra[DMA_A ? 1 : C::A_LD], rb[C::B_LD];
    const int nkb = Klen / KB;
    auto load_panel = [&](int kb, int buf) {
        const int k0 = kb * KB;
        if constexpr (DMA_A) {
#pragma unroll
            for (int q = 0; q < C::A_PIECES; ++q) {
                const int k = wave + (C::THREADS / 64) * q;
                __builtin_amdgcn_global_load_lds(DTO_GLB_PTR(A + (size_t)(k0 + k) * lda + 2 * lane), DTO_LDS_PTR(As + buf * C::AS_ELEMS + k * TM), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < C::A_LD; ++i) {
                const int idx = tid + C::THREADS * i;
                ra[i] = *reinterpret_cast<const d2*>(A + (size_t)(k0 + idx / (TM / 2)) * lda + 2 * (idx % (TM / 2)));
            }
        }
#pragma unroll
        for (int i = 0; i < C::B_LD; ++i) {
            const int idx = tid + C::THREADS * i;
            rb[i] = *reinterpret_cast<const d2*>(B + (size_t)(idx / (KB / 2)) * ldb + k0 + 2 * (idx % (KB / 2)));
        }
    };
    auto store_panel = [&](int buf) {
        if constexpr (!DMA_A) {
#pragma unroll
            for (int i = 0; i < C::A_LD; ++i) {
                const int idx = tid + C::THREADS * i;
                *reinterpret_cast<d2*>(As + buf * C::AS_ELEMS + (idx / (TM / 2)) * TM + 2 * (idx % (TM / 2))) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < C::B_LD; ++i) {
            const int idx = tid + C::THREADS * i;
            *reinterpret_cast<d2*>(Bs + buf * C::BS_ELEMS + (idx / (KB / 2)) * C::LDB_S + 2 * (idx % (KB / 2))) = rb[i];
        }
    };
    load_panel(0, 0);
    store_panel(0);
    __syncthreads();  // (its fence also waits for the DMA pieces: vmcnt(0) before the barrier)
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) load_panel(kb + 1, buf ^ 1);
        const double* as = As + buf * C::AS_ELEMS + wm * C::WTM + 2 * lr;
        const double* bs = Bs + buf * C::BS_ELEMS + (wn * C::WTN + lr) * C::LDB_S + 2 * lq;
#pragma unroll
        for (int k8 = 0; k8 < KB; k8 += 8) {
            d2 b2[C::NT];
#pragma unroll
            for (int tj = 0; tj < C::NT; ++tj) b2[tj] = *reinterpret_cast<const d2*>(bs + 16 * tj * C::LDB_S + k8);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                d2 a2[C::MT / 2];
#pragma unroll
                for (int p = 0; p < C::MT / 2; ++p) a2[p] = *reinterpret_cast<const d2*>(as + (k8 + 2 * lq + s) * TM + 32 * p);
#pragma unroll
                for (int ti = 0; ti < C::MT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < C::NT; ++tj)
                        acc.v[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(s ? b2[tj].y : b2[tj].x, (ti & 1) ? a2[ti / 2].y : a2[ti / 2].x,
                                                                             acc.v[ti][tj], 0, 0, 0);
            }
        }
#ifdef DTO_TUNING
        if (prof) {  // phase stamps (tools/stamp_analyze.py): cycles this wave waits for the next panel's loads, and at the barrier
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            if (kb + 1 < nkb) store_panel(buf ^ 1);
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
            __syncthreads();
            const unsigned long long t3 = __builtin_amdgcn_s_memtime();
            prof[0] += t1 - t0;
            prof[1] += t3 - t2;
            continue;
        }
#endif
        if (kb + 1 < nkb) store_panel(buf ^ 1);
        __syncthreads();
    }
}

// Coordinates of the calling lane's accumulator elements inside the TM x TN tile, paired-rows core: acc.v[2p][tj][r] and
// acc.v[2p+1][tj][r] are rows row_base + 32 p and row_base + 32 p + 1 of column col_base + 16 tj + 4 r.
template <class S>
struct GemmCoordP {
    int row_base, col_base;
    __device__ __forceinline__ GemmCoordP() {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        row_base = (wave / S::WC) * S::WTM + 2 * (lane & 15);
        col_base = (wave % S::WC) * S::WTN + (lane >> 4);
    }
};
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));  // a 16-byte access at an 8-byte-aligned address (slab columns)

template <int TM, int TN>
__device__ __forceinline__ void gemm_accumulate(GemmAcc<TM, TN>& acc, const double* __restrict__ A, int lda,
                                                const double* __restrict__ B, int ldb, int Klen,
                                                const double* __restrict__ colscale, double* smem) {
    gemm_accumulate_s<GemmCfg<TM, TN>>(acc, A, lda, B, ldb, Klen, colscale, smem);
}
template <int TM, int TN>
__device__ __forceinline__ void gemm_accumulate2(GemmAcc<TM, TN>& acc, const double* __restrict__ A, int lda,
                                                 const double* __restrict__ B, int ldb, int Klen,
                                                 const double* __restrict__ colscale, const double* __restrict__ B2,
                                                 const double* __restrict__ colscale2, double* smem) {
    gemm_accumulate_s<GemmCfg<TM, TN>, true>(acc, A, lda, B, ldb, Klen, colscale, smem, B2, colscale2);
}

// Coordinates of the calling lane's accumulator elements inside the TM x TN tile.
template <class S>
struct GemmCoordS {
    int row_base;  // + 16*ti
    int col_base;  // + 16*tj + 4*r
    __device__ __forceinline__ GemmCoordS() {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        row_base = (wave / S::WC) * S::WTM + (lane & 15);
        col_base = (wave % S::WC) * S::WTN + (lane >> 4);
    }
};
template <int TM, int TN>
using GemmCoord = GemmCoordS<GemmCfg<TM, TN>>;

// XCD-aware decode of a 1-D grid into (batch, tile).  Workgroups b and b+8 share an XCD under the
// observed round-robin dispatch, so the tiles of one matrix are given ids that differ by multiples
// of 8 and sit next to each other in dispatch order: they then re-use each other's A/B panels from
// the same XCD's L2.  Placement only affects speed, never correctness.
__device__ __forceinline__ bool decode_batch_tile(int wg, int nbatch, int tiles_per_mat, int& batch, int& tile) {
    const int xcd = wg & 7;
    const int idx = wg >> 3;
    batch = (idx / tiles_per_mat) * 8 + xcd;
    tile = idx % tiles_per_mat;
    return batch < nbatch;
}
__host__ __device__ inline int batch_tile_count(int nbatch, int tiles_per_mat) { return ((nbatch + 7) / 8) * 8 * tiles_per_mat; }

}  // namespace dto
