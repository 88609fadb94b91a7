// flat_mfma_kernel.h -- query batches: MFMA GEMM tile fused with per-query top-k.
//
// The one place the search path is a real dense contraction (BASELINE.json north_star):
// S[row, q] = sum_k X[row, k] * Q[q, k] for a batch of queries, followed by top-k over rows.
// fp32 rows: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-for-bit a k-ordered fmaf chain,
// no TF32/xf32 on gfx950), peak 157 TFLOP/s = the bound for nq >~ 40 (ridge ~20 flop/B).
// bfloat16 rows: v_mfma_f32_32x32x16_bf16 (fp32 accumulate), peak ~2.5 PFLOP/s.
// Two epilogues: list mode (described below) and filter mode (see the kernel's comment).
//
// Tiling (one workgroup = WM x WN waves, always 4 or 8 so every SIMD of the CU gets the same
// number of waves):
//   block tile   BM = 32*MT*WM rows  x  BN = 32*WN queries,  K step BK = 16 or 32 words
//   wave (wm,wn) rows [32*MT*wm, +32*MT) x queries [32*wn, +32): MT accumulator tiles of 32x32.
//                A wave owns its (row slice, query column) pairs outright: its top-k lists and
//                thresholds are wave-private, no cross-wave synchronisation in the epilogue.
//   operands     A = X tile (row = lane&31, k = lane>>5), B = Q^T tile (query = lane&31);
//                C layout: column (lane&31) = query, 16 registers = rows
//                (reg&3) + 8*(reg>>2) + 4*(lane>>5) -> a lane compares its 16 rows against
//                ITS query's threshold with plain v_cmp, no shuffles.
//   LDS          NS-stage ring of X and Q tiles, 64-byte rows, 16-byte chunks XOR-swizzled by
//                (row>>2)&3 (conflict-free ds_read_b128 over any 16 rows distinct mod 16);
//                per-(wave, query) sorted k-lists; 32-float scratch per wave for the column
//                transposition on a threshold hit.
//   pipeline     direct-to-LDS loads (global_load_lds_dwordx4, no staging registers): stage
//                e+NS-1 is issued while stage e is multiplied; LDS stays linear per wave
//                instruction, the swizzle is applied to the per-lane SOURCE address and to the
//                fragment reads; counted s_waitcnt vmcnt(N) + one raw s_barrier per K step;
//                the K loop runs on across row tiles so there is no bubble between tiles.
//   grid         row_splits x query_tiles persistent blocks; block b works on row tiles
//                s, s+S, ...; blocks that share a row split (different query tiles) get
//                consecutive ids with equal b%8 so they share an XCD's L2 (speed only).
// Epilogue per row tile: ballot the lanes with a score above their query's current k-th
// key (rare once warm); for each such query dump its column to scratch, offer the 32 rows
// to the LDS-resident list through the wave-cooperative WaveTopK insert, refresh the
// threshold.  The score matrix is never written to HBM.  Every (block, wm) pair writes one
// k-list per query; flat_merge_kernel merges the S*WM lists.
#pragma once
#include "common.h"
#include "flat_mfma.h"
#include "topk.h"

namespace rag {
namespace mfma {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Operands are addressed in 4-byte words whatever the element type (the loads move 16-byte
// chunks): BF = false: a word is one fp32; BF = true: a word is two bfloat16, `ld` and BK count
// words, and a 16-byte chunk is exactly the 8-element K slice one lane feeds to
// v_mfma_f32_32x32x16_bf16 (row/query = lane&31, k = 8*(lane>>5)..+7), so the LDS addressing of
// the fragments is the same for both types.
// K step: BK words per row per stage (16 or 32).  LDS rows are BK words (64 / 128 bytes) made
// of CH = BK/4 16-byte chunks; the chunk position is XOR-swizzled with f(row) = (row>>2)&3 (CH=4)
// or (row>>1)&7 (CH=8), which makes ds_read_b128 over 16 rows distinct mod 16 conflict-free.
// BK=32 fetches whole 128-byte lines per row and halves the barriers per flop.
template <int BK>
__device__ __forceinline__ int swz(int row) {
    return BK == 16 ? (row >> 2) & 3 : (row >> 1) & 7;
}

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    // 64 lanes x 16 B -> LDS [lds_wave_base + 16*lane]; asynchronous, counted by vmcnt
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt [3:0]+[15:14], expcnt [6:4], lgkmcnt [11:8]).
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 0xF) | ((N >> 4) << 14));
}

// benchmarks/mfma_timing.hip builds this header with RAG_MFMA_TIMING: per-segment s_memtime
// totals of wave 0 of every block are added into g_mfma_timing (never in the product build).
#ifdef RAG_MFMA_TIMING
__device__ unsigned long long g_mfma_timing[8];
#define RAG_TMARK(i)                                            \
    {                                                           \
        const unsigned long long t_ = __builtin_readcyclecounter(); \
        tacc[i] += t_ - tprev;                                  \
        tprev = t_;                                             \
    }
#else
#define RAG_TMARK(i)
#endif

// FILT = false ("list mode"): every (wave, query) keeps an exact sorted k-list in LDS; output
//   cand[q][s*WM + wm][k].
// FILT = true ("filter mode", larger k): no lists.  Each query has a fixed threshold thr_q[q]
//   (the j-th best score of a 1/tile_step sample of the row tiles, found by a list-mode pass);
//   every row scoring >= it is appended to the query's buffer cand[q][0..cap) through a global
//   counter.  If k <= count <= cap the top-k of the buffer is the exact global top-k (all rows
//   >= threshold are in it); otherwise the query is redone in list mode (flat.hip).
// tile_step: only row tiles t*tile_step are visited (1 = all; the sample pass uses > 1).
// run_flag: if non-null and *run_flag == 0 the kernel returns at once (the list-mode redo is
//   enqueued unconditionally and costs a launch when nothing failed).
template <int WM, int WN, int MT, int NT, int BK, int NS, int KS, int METRIC, bool BF, bool FILT>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN > 8 ? 4 : 2)) void flat_mfma_kernel(
    const float* __restrict__ X, long long n, int ld, const float* __restrict__ Qp, int nq, int k, int S,
    int col_tiles, float* __restrict__ cand_key, int* __restrict__ cand_id, const float* __restrict__ xnorm,
    const float* __restrict__ qnorm, const float* __restrict__ zero16, long long tile_step,
    const int* __restrict__ run_flag, const float* __restrict__ thr_q, unsigned int* __restrict__ counts, int cap) {
    if (run_flag != nullptr && *run_flag == 0) return;  // uniform, before any barrier
    constexpr int NWV = WM * WN;
    constexpr int T = 64 * NWV;
    constexpr int BMW = 32 * MT;          // rows per wave
    constexpr int BM = BMW * WM;
    constexpr int BNW = 32 * NT;          // queries per wave
    constexpr int BN = BNW * WN;
    constexpr int CH = BK / 4;            // 16-byte chunks per tile row
    constexpr int LDS_LD = BK;            // tile row stride in floats
    constexpr int XI = BM * CH / 64;      // wave-instructions (1 KiB each) per X tile
    constexpr int QI = BN * CH / 64;      // ... per Q tile
    constexpr int XQ = XI / NWV;          // per wave
    constexpr int QQ = (QI + NWV - 1) / NWV;
    constexpr int LPS = XQ + QQ;          // LDS-DMA instructions per wave per stage
    static_assert(XI % NWV == 0, "X tile must split evenly over the waves");
    static_assert(NS >= 2 && (NS - 2) * LPS < 64, "stage ring too deep for vmcnt");

    extern __shared__ __align__(16) unsigned char smem_raw[];
    float* Xs = reinterpret_cast<float*>(smem_raw);                 // [NS][BM][LDS_LD]
    float* Qs = Xs + NS * BM * LDS_LD;                              // [NS][BN][LDS_LD]
    float* dummy = Qs + NS * BN * LDS_LD;                           // [256] sink of filler loads
    const int kl = FILT ? 0 : k;                                    // list entries held in LDS
    float* lkey = dummy + 256;                                      // [NWV][BNW][kl]
    int* lid = reinterpret_cast<int*>(lkey + NWV * BNW * kl);       // [NWV][BNW][kl]
    float* scratch = reinterpret_cast<float*>(lid + NWV * BNW * kl);// [NWV][32]
    float* xn_s = scratch + NWV * 32;                               // [BM] row norms (L2 only)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wn = wid % WN, wm = wid / WN;

    int s, c;
    {
        const int b = blockIdx.x;
        if ((S & 7) == 0) {
            const int xcd = b & 7, t_ = b >> 3;
            c = t_ % col_tiles;
            s = (t_ / col_tiles) * 8 + xcd;
        } else {
            c = b % col_tiles;
            s = b / col_tiles;
        }
    }
    const int q0 = c * BN;
    const long long ntiles = ((n + BM - 1) / BM + tile_step - 1) / tile_step;  // tiles this launch visits
    const long long my_tiles = s < ntiles ? (ntiles - s + S - 1) / S : 0;
    const long long tile_rows = (long long)BM * tile_step;                     // row distance of visited tiles
    const int nk = (ld + BK - 1) / BK;
    const long long total = my_tiles * nk;

    // lists start as pads
    for (int i = tid; i < NWV * BNW * kl; i += T) {
        lkey[i] = Lowest<float>::v();
        lid[i] = IdPad<int>::v();
    }
    // this lane's NT queries.  thr: list mode: current k-th key of the query (within this wave's
    // rows); filter mode: the query's fixed threshold (padding queries never pass)
    int my_q[NT];
    float thr[NT], qn[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        my_q[u] = q0 + BNW * wn + 32 * u + li;
        thr[u] = Lowest<float>::v();
        if (FILT) thr[u] = my_q[u] < nq ? thr_q[my_q[u]] : FLT_MAX;
        qn[u] = 0.f;
        if (METRIC == RAG_METRIC_L2) qn[u] = my_q[u] < nq ? qnorm[my_q[u]] : 0.f;
    }

    // Direct-to-LDS tile loads.  LDS slot s (16 B) of a tile holds row s/CH, physical chunk s%CH =
    // logical chunk (s%CH) ^ swz(row).  A wave instruction fills 64 consecutive slots;
    // out-of-range K columns / queries read zeros.  Steps past the end and waves without a Q
    // share still issue (zero-source / sink) loads so that every wave has the same number of
    // outstanding operations in every iteration, which is what the counted waits rely on.
    //
    // Everything that does not change from one K step to the next is hoisted: per lane and
    // per instruction the element offset of its (row, chunk) and the chunk's column, refreshed
    // once per row tile; per K step an address is one 64-bit add.
    const bool ktail = (ld % BK) != 0;        // only then can a chunk fall beyond the row
    long long xoff[XQ];                       // element offset of this lane's chunk at k0 = 0 (current prefetch tile)
    int xcol[XQ];                             // 4 * logical chunk
    long long qoff[QQ];                       // same for the query tile; < 0: no query behind this slot
    int qcol[QQ];
#pragma unroll
    for (int it = 0; it < XQ; ++it) {
        const int slot = (it * NWV + wid) * 64 + lane;
        xcol[it] = 4 * ((slot % CH) ^ swz<BK>(slot / CH));
    }
#pragma unroll
    for (int it = 0; it < QQ; ++it) {
        const int slot = (it * NWV + wid) * 64 + lane;
        const int r = slot / CH;
        qcol[it] = 4 * ((slot % CH) ^ swz<BK>(r));
        qoff[it] = (it * NWV + wid < QI && q0 + r < nq) ? (long long)(q0 + r) * ld + qcol[it] : -1;
    }
    long long pf_tiles_left = my_tiles;       // prefetch cursor: tiles not yet fully issued
    long long pf_row0 = (long long)s * tile_rows;
    int pf_ks = 0;
    auto set_tile_offsets = [&]() {
#pragma unroll
        for (int it = 0; it < XQ; ++it) {
            const int slot = (it * NWV + wid) * 64 + lane;
            long long row = pf_row0 + slot / CH;
            row = row < n ? row : n - 1;
            xoff[it] = row * ld + xcol[it];
        }
    };
    set_tile_offsets();
    // One LDS-DMA instruction of the stage being prefetched: part < XQ is an X chunk row, the
    // rest are Q (or sink) loads.  The prefetch cursor advances with the last part.
    auto issue_part = [&](int buf, int part) {
        float* xs = Xs + buf * BM * LDS_LD;
        float* qs = Qs + buf * BN * LDS_LD;
        if (pf_tiles_left > 0) {
            const int k0 = pf_ks * BK;
            if (part < XQ) {
                const int it = part;
                const float* gp = X + xoff[it] + k0;
                if (ktail && k0 + xcol[it] >= ld) gp = zero16;
                glds16(gp, xs + (it * NWV + wid) * 256);
            } else {
                const int it = part - XQ;
                if (it * NWV + wid < QI) {  // wave-uniform
                    const float* gp = qoff[it] >= 0 ? Qp + qoff[it] + k0 : zero16;
                    if (ktail && k0 + qcol[it] >= ld) gp = zero16;
                    glds16(gp, qs + (it * NWV + wid) * 256);
                } else {
                    glds16(zero16, dummy);
                }
            }
            if (part == LPS - 1 && ++pf_ks == nk) {
                pf_ks = 0;
                pf_row0 += (long long)S * tile_rows;
                if (--pf_tiles_left > 0) set_tile_offsets();
            }
        } else {
            glds16(zero16, dummy);
        }
    };
    auto issue_loads = [&](int buf) {
#pragma unroll
        for (int part = 0; part < LPS; ++part) issue_part(buf, part);
    };

    // accumulators are cleared after each row tile's epilogue (inside its uniform branch), not
    // by a per-K-step select
    f32x16 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    using TK = WaveTopK<float, int, KS>;

    // NS-stage ring: stages e .. e+NS-2 are in flight at the top of iteration e.
    __syncthreads();  // list initialisation visible before any epilogue
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) issue_loads(st);

    int cur = 0;
    int ks = 0;                                // K step within the current row tile
    long long row0 = (long long)s * tile_rows; // first row of the current row tile
#ifdef RAG_MFMA_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
#endif
    for (long long e = 0; e < total; ++e) {
        // this wave's share of stage e has landed; after the barrier everyone's has, and
        // everyone is done reading the buffer of stage e-1, which the next issue overwrites
        wait_vmcnt<(NS - 2) * LPS>();
        RAG_TMARK(0)
        __builtin_amdgcn_s_barrier();
        RAG_TMARK(1)
        {
            const float* xs = Xs + cur * BM * LDS_LD + wm * BMW * LDS_LD;
            const float* qs = Qs + cur * BN * LDS_LD;
            const int sw = swz<BK>(li);  // = swz(row): every tile starts at a multiple of 32 rows
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                // The next stage's loads go out behind the first MFMA batches: the waves of a CU
                // share one address path (64 B/clk), and a wave that queues there with nothing in
                // the matrix pipe leaves its SIMD idle.  Measured slower and dropped: spreading
                // the loads one by one between MFMAs with double-buffered fragments (+32 VGPRs
                // spill the 8-wave shapes); letting the two waves of a SIMD queue at different
                // points of the K step.
                if (kk == BK / 16) {
                    __builtin_amdgcn_sched_barrier(0);
                    RAG_TMARK(2)
                    int nxt = cur + NS - 1;
                    nxt = nxt >= NS ? nxt - NS : nxt;
#if defined(RAG_MFMA_TIMING) && defined(RAG_EXP_NO_GLDS)
                    if (e < 4) issue_loads(nxt);
                    else {
#pragma unroll
                        for (int part = 0; part < LPS; ++part) glds16(zero16, dummy);
                    }
#else
                    issue_loads(nxt);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int pc = ((kk * 2 + lh) ^ sw) * 4;
                f32x4 bq[NT], a[MT];
#if defined(RAG_MFMA_TIMING) && defined(RAG_EXP_NO_LDSREAD)
#pragma unroll
                for (int u = 0; u < NT; ++u) bq[u] = f32x4{1.f * pc, 2.f, 3.f, 4.f};
#pragma unroll
                for (int t = 0; t < MT; ++t) a[t] = f32x4{1.f * t, 2.f * pc, 3.f, 4.f};
#else
#pragma unroll
                for (int u = 0; u < NT; ++u)
                    bq[u] = *reinterpret_cast<const f32x4*>(qs + (BNW * wn + 32 * u + li) * LDS_LD + pc);
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    a[t] = *reinterpret_cast<const f32x4*>(xs + (32 * t + li) * LDS_LD + pc);
#endif
#if defined(RAG_MFMA_TIMING) && defined(RAG_EXP_NO_MFMA)
#pragma unroll
                for (int u = 0; u < NT; ++u)
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[t][u][0] += a[t][0] * bq[u][1];
#else
                if (BF) {
#pragma unroll
                    for (int u = 0; u < NT; ++u)
#pragma unroll
                        for (int t = 0; t < MT; ++t)
                            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, a[t]), __builtin_bit_cast(bf16x8, bq[u]), acc[t][u], 0, 0, 0);
                } else {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                        for (int u = 0; u < NT; ++u)
#pragma unroll
                            for (int t = 0; t < MT; ++t)
                                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][cc], bq[u][cc], acc[t][u], 0, 0, 0);
                }
#endif
            }
        }
        RAG_TMARK(3)
        if (ks == nk - 1) {
            // ---- epilogue of one row tile: threshold filter + list update
            const bool partial = row0 + BM > n;
            if (METRIC == RAG_METRIC_L2) {
                __syncthreads();
                for (int i = tid; i < BM; i += T) {
                    const long long row = row0 + i;
                    xn_s[i] = row < n ? xnorm[row] : 0.f;
                }
                __syncthreads();
            }
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int tile_off = wm * BMW + 32 * t;
                const long long rowbase = row0 + tile_off;
                float key[16];
                bool any = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float v = acc[t][u][r];
                    if (METRIC == RAG_METRIC_L2) v = (2.f * v - xn_s[tile_off + ro]) - qn[u];
                    key[r] = v;
                    bool ok = FILT ? v >= thr[u] : v > thr[u];
                    if (partial) ok = ok && (rowbase + ro < n);
                    any = any || ok;
                }
                const unsigned long long m = __ballot(any);
                if (FILT) {
                    if (m != 0ULL && my_q[u] < nq) {
                        // rare: append every passing row of this lane's query to its buffer
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ro = (r & 3) + 8 * (r >> 2) + 4 * lh;
                            if (key[r] >= thr[u] && rowbase + ro < n) {
                                const unsigned int slot = atomicAdd(&counts[my_q[u]], 1u);
                                if (slot < (unsigned int)cap) {
                                    const long long o = (long long)my_q[u] * cap + slot;
                                    cand_key[o] = key[r];
                                    cand_id[o] = (int)(rowbase + ro);
                                }
                            }
                        }
                    }
                } else {
                    unsigned qm = (unsigned)(m | (m >> 32));
                    while (qm) {
                        const int j = __ffs(qm) - 1;
                        qm &= qm - 1;
                        float* sc = scratch + wid * 32;
                        if (li == j) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) sc[(r & 3) + 8 * (r >> 2) + 4 * lh] = key[r];
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        const float cs = sc[li];
                        const long long crow = rowbase + li;
                        TK tk;
                        const int lo = ((wid * NT + u) * 32 + j) * k;
#pragma unroll
                        for (int sl = 0; sl < KS; ++sl) {
                            const int p = sl * 64 + lane;
                            tk.key[sl] = p < k ? lkey[lo + p] : Lowest<float>::v();
                            tk.id[sl] = p < k ? lid[lo + p] : IdPad<int>::v();
                        }
                        float thr_k;
                        int thr_i;
                        tk.kth(k, thr_k, thr_i);
                        wave_offer<float, int, KS>(tk, k, thr_k, thr_i, cs, (int)crow, lane < 32 && crow < n, lane);
#pragma unroll
                        for (int sl = 0; sl < KS; ++sl) {
                            const int p = sl * 64 + lane;
                            if (p < k) {
                                lkey[lo + p] = tk.key[sl];
                                lid[lo + p] = tk.id[sl];
                            }
                        }
                        if (li == j) thr[u] = thr_k;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
            }
        }
        RAG_TMARK(4)
        cur = cur + 1 == NS ? 0 : cur + 1;
        if (++ks == nk) {
            ks = 0;
            row0 += (long long)S * tile_rows;
        }
    }
    wait_vmcnt<0>();  // no LDS-DMA may outlive the workgroup
#ifdef RAG_MFMA_TIMING
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) atomicAdd(&g_mfma_timing[i], tacc[i]);
        atomicAdd(&g_mfma_timing[5], (unsigned long long)total);
        atomicAdd(&g_mfma_timing[6], 1ULL);
    }
#endif

    // ---- list mode: write the lists of every real query: cand[q][s*WM + wm][k]
    __syncthreads();
    for (int i = tid; i < NWV * BNW * kl; i += T) {
        const int w = i / (BNW * k), rem = i - w * BNW * k;
        const int j = rem / k, p = rem - j * k;   // j = 32*u + lane-in-tile
        const int qi = q0 + BNW * (w % WN) + j;
        if (qi < nq) {
            const long long o = ((long long)qi * (S * WM) + (long long)s * WM + (w / WN)) * k + p;
            cand_key[o] = lkey[i];
            cand_id[o] = lid[i];
        }
    }
}

inline size_t smem_need(int WM, int WN, int MT, int NT, int BK, int NS, int k) {
    const size_t BM = 32 * (size_t)MT * WM, BN = 32 * (size_t)NT * WN, NWV = (size_t)WM * WN;
    return NS * (BM + BN) * BK * 4 + 1024 + NWV * 32 * NT * (size_t)k * 8 + NWV * 128 + BM * 4;
}

struct Args {
    const float* X;
    long long n;
    int ld;  // words per row
    const float* Qp;
    int nq, k, metric;
    float* cand_key;
    int* cand_id;
    const float *xnorm, *qnorm, *zero16;
    hipStream_t st;
    const int* run_flag;     // optional: launch is a no-op when *run_flag == 0
    const float* thr_q;      // filter mode: per-query threshold
    unsigned int* counts;    // filter mode: per-query append counters (zeroed by the caller)
};

template <int WM, int WN, int MT, int NT, int BK, int NS, int KS, bool BF, bool FILT>
int launch(const MfmaPlan& p, const Args& a) {
    const size_t smem = smem_need(WM, WN, MT, NT, BK, NS, FILT ? 0 : a.k);
    auto fn = a.metric == RAG_METRIC_IP ? flat_mfma_kernel<WM, WN, MT, NT, BK, NS, KS, RAG_METRIC_IP, BF, FILT>
                                        : flat_mfma_kernel<WM, WN, MT, NT, BK, NS, KS, RAG_METRIC_L2, BF, FILT>;
    if (smem > 48 * 1024)
        RAG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem));
    fn<<<(unsigned)(p.S * p.col_tiles), 64 * WM * WN, smem, a.st>>>(
        a.X, a.n, a.ld, a.Qp, a.nq, a.k, p.S, p.col_tiles, a.cand_key, a.cand_id, a.xnorm, a.qnorm, a.zero16,
        (long long)p.tile_step, a.run_flag, a.thr_q, a.counts, p.cap);
    RAG_HIP(hipGetLastError());
    return RAG_OK;
}

// The block shapes that are built.  bpc = blocks per CU the grid is sized for.
struct Shape {
    int WM, WN, MT, NT, BK, bpc;
};
constexpr Shape SHAPES[] = {
    {4, 1, 2, 1, 16, 2},  // 1: <= 32 queries, 256 rows
    {2, 2, 4, 1, 16, 2},  // 2: <= 64 queries, 256 rows
    {1, 4, 8, 1, 16, 2},  // 3: <= 128 queries, 256 rows, two decoupled 4-wave blocks per CU
    {1, 8, 8, 1, 16, 1},  // 4: 256 queries, 256 rows, one 8-wave block per CU
    {4, 1, 2, 1, 32, 1},  // 5: <= 32 queries, whole 128-byte lines
    {2, 2, 4, 1, 32, 1},  // 6: <= 64 queries
    {1, 4, 8, 1, 32, 1},  // 7: <= 128 queries
    {1, 8, 8, 1, 32, 1},  // 8: 256 queries
    {2, 4, 4, 2, 16, 1},  // 9: 256 queries x 256 rows, wave tile 128 rows x 64 queries (6 fragment reads per
                          //    8 MFMAs instead of 9); measured equal to shape 4, kept as an option
    {2, 4, 4, 2, 32, 1},  // 10: the same tile with the software-pipelined K loop of flat_mfma_pipe.h
                          //    (filter mode only, rows of whole 128-byte lines)
};
constexpr int NSHAPES = sizeof(SHAPES) / sizeof(SHAPES[0]);

// One translation unit per element type instantiates its kernels behind this dispatcher
// (flat_mfma_f32.hip / flat_mfma_bf16.hip).  List mode: KS = 2 (64 < k <= 128) exists for the
// 4-wave shapes only (eight waves' lists of that length do not fit the LDS).  Filter mode holds
// no lists and is built for the 64-byte-row shapes (1-4, 9) and the 128-byte-row shapes 7, 8.
}  // namespace mfma
}  // namespace rag
#ifndef RAG_PIPE_MFMA16
#define RAG_PIPE_MFMA16 1
#endif
#include "flat_mfma_pipe.h"
namespace rag {
namespace mfma {

// thr_q == nullptr: the sample pass (tile_step > 1; cand_key receives p.cap values per query).
template <bool BF>
int launch_pipe(const MfmaPlan& p, const Args& a) {
    const size_t smem = pipe_smem_need();
    const bool sample = a.thr_q == nullptr;
    constexpr bool M16 = BF && RAG_PIPE_MFMA16 != 0;  // bfloat16 tiles as v_mfma_f32_16x16x32_bf16 (flat_mfma_pipe.h)
    auto fn = a.metric == RAG_METRIC_IP
                  ? (sample ? flat_pipe_kernel<RAG_METRIC_IP, BF, true, M16> : flat_pipe_kernel<RAG_METRIC_IP, BF, false, M16>)
                  : (sample ? flat_pipe_kernel<RAG_METRIC_L2, BF, true, M16> : flat_pipe_kernel<RAG_METRIC_L2, BF, false, M16>);
    RAG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    fn<<<(unsigned)(p.S * p.col_tiles), 512, smem, a.st>>>(a.X, a.n, a.ld, a.Qp, a.nq, p.S, p.col_tiles, a.cand_key,
                                                          a.cand_id, a.xnorm, a.qnorm, a.zero16, (long long)p.tile_step,
                                                          a.run_flag, a.thr_q, a.counts, p.cap);
    RAG_HIP(hipGetLastError());
    return RAG_OK;
}

template <bool BF>
int dispatch(const MfmaPlan& p, const Args& a) {
#define RAG_MFMA_NS(WM, WN, MT, NT, BK, KS, FILT)                               \
    (p.NS >= 3 ? launch<WM, WN, MT, NT, BK, 3, KS, BF, FILT>(p, a) : launch<WM, WN, MT, NT, BK, 2, KS, BF, FILT>(p, a))
#define RAG_MFMA_4W(WM, WN, MT, BK) \
    (p.KS == 2 ? RAG_MFMA_NS(WM, WN, MT, 1, BK, 2, false) : RAG_MFMA_NS(WM, WN, MT, 1, BK, 1, false))
    if (p.filt) {
        switch (p.shape) {
            case 0: return RAG_MFMA_NS(4, 1, 2, 1, 16, 1, true);
            case 1: return RAG_MFMA_NS(2, 2, 4, 1, 16, 1, true);
            case 2: return RAG_MFMA_NS(1, 4, 8, 1, 16, 1, true);
            case 3: return RAG_MFMA_NS(1, 8, 8, 1, 16, 1, true);
            case 6: return RAG_MFMA_NS(1, 4, 8, 1, 32, 1, true);
            case 7: return RAG_MFMA_NS(1, 8, 8, 1, 32, 1, true);
            case 8: return RAG_MFMA_NS(2, 4, 4, 2, 16, 1, true);
            case 9:
                if (a.ld % 32 != 0) return fail(RAG_ERR_INVALID_ARG, "block shape 10 needs rows of whole 128-byte stages");
                return launch_pipe<BF>(p, a);
            default: return fail(RAG_ERR_INVALID_ARG, "no filter-mode kernel for block shape %d", p.shape + 1);
        }
    }
    switch (p.shape) {
        case 0: return RAG_MFMA_4W(4, 1, 2, 16);
        case 1: return RAG_MFMA_4W(2, 2, 4, 16);
        case 2: return RAG_MFMA_4W(1, 4, 8, 16);
        case 3: return RAG_MFMA_NS(1, 8, 8, 1, 16, 1, false);
        case 4: return RAG_MFMA_4W(4, 1, 2, 32);
        case 5: return RAG_MFMA_4W(2, 2, 4, 32);
        case 6: return RAG_MFMA_4W(1, 4, 8, 32);
        case 7: return RAG_MFMA_NS(1, 8, 8, 1, 32, 1, false);
        case 9: return fail(RAG_ERR_INVALID_ARG, "block shape 10 is a filter-mode shape");
        default: return RAG_MFMA_NS(2, 4, 4, 2, 16, 1, false);
    }
#undef RAG_MFMA_4W
#undef RAG_MFMA_NS
}

int dispatch_f32(const MfmaPlan& p, const Args& a);   // flat_mfma_f32.hip
int dispatch_bf16(const MfmaPlan& p, const Args& a);  // flat_mfma_bf16.hip

}  // namespace mfma
}  // namespace rag
