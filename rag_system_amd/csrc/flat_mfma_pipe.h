// flat_mfma_pipe.h -- filter-mode batch kernel with a software-pipelined K loop (block shape 10).
//
// Same contraction, operand layout, swizzle and filter epilogue as flat_mfma_kernel
// (flat_mfma_kernel.h; included from there), for one block shape: 2 x 4 waves, block tile
// 256 rows x 256 queries, wave tile 128 rows x 64 queries (MT = 4, NT = 2: 6 fragment reads per
// 8 MFMAs).  Two things differ.
//
// (1) Whole cache lines.  A K stage is 32 words = one 128-byte line per row, and a wave's LDS-DMA
//     instruction fetches 8 rows x 128 B.  With 64-byte stages every line is requested twice, half
//     a line at a time, and the L2 -> LDS path delivers 42 GB/s per CU (benchmarks/pipe_timing:
//     loads alone 11.5 ms for C3, i.e. the loads, not the MFMAs, bound the 64-byte version at
//     1.36 PFLOP/s).  The stage costs 32 KiB for the rows and 32 KiB for the queries; the CU's
//     160 KiB hold a ring of THREE row stages (rows come from HBM: two stages of prefetch) and TWO
//     query stages (queries come from the L2: one stage), and nothing else -- no sink for filler
//     loads (they go into ring buffers nobody reads any more) and no row-norm tile (the L2 epilogue
//     reads its norms from global memory).
// (2) The schedule inside a stage.  flat_mfma_kernel reads a K slice's fragments and multiplies
//     them in the same phase, so every wave of the CU waits for the LDS at the same time and the
//     matrix pipe idles while they do.  Here the fragments are double buffered in registers and
//     every LDS read is issued one MFMA batch (8 MFMAs = 256 cycles) ahead of its use:
//
//   iteration e  (fa = fragments of stage e, K slice 0, already in flight or landed)
//     issue LDS-DMA: queries of stage e+1, rows of stage e+2 (into the buffers stage e-1 used)
//     fb <- slice 1 ; 8 MFMAs on fa ; fa <- slice 2 ; 8 MFMAs on fb ; fb <- slice 3 ; 8 MFMAs on fa
//     wait: own loads of stage e+1 landed (vmcnt), own reads returned (lgkmcnt) ; s_barrier
//     fa <- slice 0 of stage e+1 ; 8 MFMAs on fb
//     (last stage of a row tile: filter epilogue on the accumulators)
//
// One barrier per 32 MFMAs per wave.  Safety of the rings: every fragment read of stage e has
// returned before the barrier of iteration e, and the loads that overwrite its buffers are issued
// after that barrier (iteration e+1); stage e+1 is read only after the barrier of iteration e,
// before which every wave waited for its own share of it.  K order per accumulator is the same as
// in flat_mfma_kernel, so the scores are bit-identical.
#pragma once

namespace rag {
namespace mfma {

template <int METRIC, bool BF>
__global__ __launch_bounds__(512, 2) void flat_pipe_kernel(
    const float* __restrict__ X, long long n, int ld, const float* __restrict__ Qp, int nq, int S, int col_tiles,
    float* __restrict__ cand_key, int* __restrict__ cand_id, const float* __restrict__ xnorm,
    const float* __restrict__ qnorm, const float* __restrict__ zero16, long long tile_step,
    const int* __restrict__ run_flag, const float* __restrict__ thr_q, unsigned int* __restrict__ counts, int cap) {
    if (run_flag != nullptr && *run_flag == 0) return;  // uniform, before any barrier
    constexpr int WN = 4, MT = 4, NT = 2, BK = 32;
    constexpr int NWV = 8, BMW = 128, BM = 256, BNW = 64, BN = 256;
    constexpr int CH = BK / 4, LDS_LD = BK;
    constexpr int NSX = 3, NSQ = 2;            // ring depths: row stages / query stages
    constexpr int XQ = BM * CH / 64 / NWV;     // 4 LDS-DMA instructions per wave per X stage
    constexpr int QQ = BN * CH / 64 / NWV;     // 4 per Q stage
    constexpr int STG = BM * LDS_LD;           // words per stage buffer (32 KiB), X and Q alike
    static_assert(BM == BN, "one stage size");

    extern __shared__ __align__(16) unsigned char smem_raw[];
    float* Xs = reinterpret_cast<float*>(smem_raw);  // [NSX][BM][LDS_LD]
    float* Qs = Xs + NSX * STG;                      // [NSQ][BN][LDS_LD]   -- 160 KiB in all

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
    const long long ntiles = ((n + BM - 1) / BM + tile_step - 1) / tile_step;
    const long long my_tiles = s < ntiles ? (ntiles - s + S - 1) / S : 0;
    const long long tile_rows = (long long)BM * tile_step;
    const int nk = ld / BK;  // the plan admits only ld % BK == 0
    const long long total = my_tiles * nk;

    int my_q[NT];
    float thr[NT], qn[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        my_q[u] = q0 + BNW * wn + 32 * u + li;
        thr[u] = my_q[u] < nq ? thr_q[my_q[u]] : FLT_MAX;
        qn[u] = 0.f;
        if (METRIC == RAG_METRIC_L2) qn[u] = my_q[u] < nq ? qnorm[my_q[u]] : 0.f;
    }

    // LDS-DMA addressing as in flat_mfma_kernel: LDS slot (16 B) -> (row, physical chunk); the
    // swizzle is applied to the source address.  One wave instruction = 8 rows x one whole 128-byte
    // line each.
    // Addresses are a wave-uniform 64-bit base (scalar registers) plus ONE 32-bit per-lane byte offset
    // that rows and queries share: instruction `it` of a wave covers rows lrow + 64*it (same chunk,
    // same swizzle -- 64 rows further the swizzle repeats), and the 64*it rows go into the base.
    // (Per-lane 64-bit addresses for 8 loads do not fit beside 176 accumulator/fragment registers.)
    const int lrow = wid * 8 + (lane >> 3);  // row of this lane's slot in instruction 0
    const int lcol = 4 * ((lane & 7) ^ swz<BK>(lrow));
    const unsigned lane_off = 4u * (unsigned)(lrow * ld + lcol);  // bytes; < 256 rows of <= 32 KiB
    const char* Xb = reinterpret_cast<const char*>(X);
    const char* Qb = reinterpret_cast<const char*>(Qp) + (long long)q0 * ld * 4;
    const long long row_stride = 256LL * ld;  // bytes between the rows of consecutive instructions
    const bool q_whole = q0 + BN <= nq;
    // two prefetch cursors: rows run two stages ahead of the multiply, queries one
    long long px_tiles_left = my_tiles;
    long long px_row0 = (long long)s * tile_rows;
    int px_ks = 0, px_buf = 0;
    long long pq_left = total;
    int pq_ks = 0, pq_buf = 0;
    // Past the end the loads still issue (zero source, into a ring buffer nobody reads any more) so
    // that the number of outstanding operations per iteration stays what the counted waits assume.
    auto issue_x = [&]() {
        float* xs = Xs + px_buf * STG;
        px_buf = px_buf + 1 == NSX ? 0 : px_buf + 1;
#if defined(RAG_PIPE_EXP) && (RAG_PIPE_EXP & 1)
        if (px_tiles_left < my_tiles) return;  // ablation: no loads after the first row tile
#endif
        if (px_tiles_left > 0) {
            const char* base = Xb + (px_row0 * ld + px_ks * BK) * 4;
            if (px_row0 + BM <= n) {
#pragma unroll
                for (int it = 0; it < XQ; ++it)
                    glds16(base + it * row_stride + (unsigned long long)lane_off, xs + (it * NWV + wid) * 256);
            } else {  // the last, partial row tile: rows past the end re-read the last row (masked in the epilogue)
#pragma unroll
                for (int it = 0; it < XQ; ++it) {
                    long long r = lrow + 64 * it;
                    r = px_row0 + r < n ? r : n - 1 - px_row0;
                    glds16(base + (unsigned long long)(4u * (unsigned)(r * ld + lcol)), xs + (it * NWV + wid) * 256);
                }
            }
            if (++px_ks == nk) {
                px_ks = 0;
                px_row0 += (long long)S * tile_rows;
                --px_tiles_left;
            }
        } else {
#pragma unroll
            for (int it = 0; it < XQ; ++it) glds16(zero16, xs + (it * NWV + wid) * 256);
        }
    };
    auto issue_q = [&]() {
        float* qs = Qs + pq_buf * STG;
        pq_buf ^= 1;
#if defined(RAG_PIPE_EXP) && (RAG_PIPE_EXP & 1)
        if (pq_left + nk < total) return;
#endif
        if (pq_left > 0) {
            const char* base = Qb + pq_ks * BK * 4;
            if (q_whole) {
#pragma unroll
                for (int it = 0; it < QQ; ++it)
                    glds16(base + it * row_stride + (unsigned long long)lane_off, qs + (it * NWV + wid) * 256);
            } else {  // fewer than 256 queries left: the padding columns re-read the last query (never pass)
#pragma unroll
                for (int it = 0; it < QQ; ++it) {
                    int r = lrow + 64 * it;
                    r = q0 + r < nq ? r : nq - 1 - q0;
                    glds16(base + (unsigned long long)(4u * (unsigned)(r * ld + lcol)), qs + (it * NWV + wid) * 256);
                }
            }
            pq_ks = pq_ks + 1 == nk ? 0 : pq_ks + 1;
            --pq_left;
        } else {
#pragma unroll
            for (int it = 0; it < QQ; ++it) glds16(zero16, qs + (it * NWV + wid) * 256);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    // Fragment reads: inline assembly on purpose.  The compiler's own wait insertion treats the
    // loop-carried reads as "wait for everything" (lgkmcnt(0) in front of the MFMA batch that uses
    // them, which exposes the LDS latency); issued this way they are invisible to it and the counted
    // waits below are the only ones.  Addresses are bytes in the LDS aperture.
    const int sw = swz<BK>(li);
    const unsigned xs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)Xs;
    const unsigned qs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)Qs;
    const unsigned xfrag = xs_lds + 4u * (unsigned)((wm * BMW + li) * LDS_LD);
    const unsigned qfrag = qs_lds + 4u * (unsigned)((BNW * wn + li) * LDS_LD);
    unsigned pcb[4];  // byte offset of this lane's chunk in its row, per K slice
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) pcb[kk] = 16u * (unsigned)((kk * 2 + lh) ^ sw);
    f32x4 fa_x[MT], fa_q[NT], fb_x[MT], fb_q[NT];
    constexpr int FR = MT + NT;  // LDS reads per fragment set
    auto read_frags = [&](int xbuf, int qbuf, int kk, f32x4(&ax)[MT], f32x4(&aq)[NT]) {
        const unsigned xa = xfrag + (unsigned)xbuf * (STG * 4u) + pcb[kk];
        const unsigned qa = qfrag + (unsigned)qbuf * (STG * 4u) + pcb[kk];
        static_assert(32 * LDS_LD * 4 == 4096 && MT == 4 && NT == 2, "fragment offsets below");
#if defined(RAG_PIPE_EXP) && (RAG_PIPE_EXP & 2)
        return;  // ablation: no fragment reads
#endif
        asm volatile("ds_read_b128 %0, %1" : "=&v"(aq[0]) : "v"(qa));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=&v"(aq[1]) : "v"(qa));
        asm volatile("ds_read_b128 %0, %1" : "=&v"(ax[0]) : "v"(xa));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=&v"(ax[1]) : "v"(xa));
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=&v"(ax[2]) : "v"(xa));
        asm volatile("ds_read_b128 %0, %1 offset:12288" : "=&v"(ax[3]) : "v"(xa));
    };
    auto multiply = [&](const f32x4(&ax)[MT], const f32x4(&aq)[NT]) {
#if defined(RAG_PIPE_VAR) && (RAG_PIPE_VAR & 2)
        __builtin_amdgcn_s_setprio(1);
#endif
#if defined(RAG_PIPE_EXP) && (RAG_PIPE_EXP & 4)
        acc[0][0][0] += ax[0][0] * aq[0][0];  // ablation: no MFMA
        return;
#endif
        if (BF) {
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, ax[t]), __builtin_bit_cast(bf16x8, aq[u]), acc[t][u], 0, 0, 0);
        } else {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int u = 0; u < NT; ++u)
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[t][cc], aq[u][cc], acc[t][u], 0, 0, 0);
        }
#if defined(RAG_PIPE_VAR) && (RAG_PIPE_VAR & 2)
        __builtin_amdgcn_s_setprio(0);
#endif
    };
#if defined(RAG_PIPE_EXP) && (RAG_PIPE_EXP & 2)
#pragma unroll
    for (int t = 0; t < MT; ++t) fa_x[t] = fb_x[t] = f32x4{1.f * lane, 2.f, 3.f, 4.f};
#pragma unroll
    for (int u = 0; u < NT; ++u) fa_q[u] = fb_q[u] = f32x4{1.f, 2.f * lane, 3.f, 4.f};
#endif

    // prologue: rows of stages 0 and 1 and queries of stage 0 in flight; slice 0 of stage 0 -> fa
    issue_x();
    issue_q();
    issue_x();
    if (total > 0) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XQ));
        __builtin_amdgcn_s_barrier();
        read_frags(0, 0, 0, fa_x, fa_q);
    }

    int xcur = 0, qcur = 0, ks = 0;
    long long row0 = (long long)s * tile_rows;
    for (long long e = 0; e < total; ++e) {
        issue_q();  // queries of stage e+1 -> the buffer stage e-1 used
        issue_x();  // rows of stage e+2    -> the buffer stage e-1 used
        read_frags(xcur, qcur, 1, fb_x, fb_q);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(FR));  // slice 0 has landed, slice 1 may be in flight
        __builtin_amdgcn_sched_barrier(0);
        multiply(fa_x, fa_q);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(xcur, qcur, 2, fa_x, fa_q);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(FR));
        __builtin_amdgcn_sched_barrier(0);
        multiply(fb_x, fb_q);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(xcur, qcur, 3, fb_x, fb_q);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(FR));
        __builtin_amdgcn_sched_barrier(0);
        multiply(fa_x, fa_q);
        __builtin_amdgcn_sched_barrier(0);
#if defined(RAG_PIPE_VAR) && (RAG_PIPE_VAR & 1)
        // variant: the barrier after the last MFMA batch (a full iteration of lead for the query loads;
        // slice 0 of the next stage is then read with its latency exposed)
        asm volatile("s_waitcnt lgkmcnt(0)");
        __builtin_amdgcn_sched_barrier(0);
        multiply(fb_x, fb_q);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XQ));
        __builtin_amdgcn_s_barrier();
        xcur = xcur + 1 == NSX ? 0 : xcur + 1;
        qcur ^= 1;
        read_frags(xcur, qcur, 0, fa_x, fa_q);
#else
        // own loads of stage e+1 have landed (all but the rows of stage e+2 issued above), every
        // fragment read of stage e has returned.  (Assembly, not the builtin: the compiler relaxes a
        // builtin wait that it believes unnecessary.)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(XQ));
#if !defined(RAG_PIPE_EXP) || !(RAG_PIPE_EXP & 8)
        __builtin_amdgcn_s_barrier();
#endif
        xcur = xcur + 1 == NSX ? 0 : xcur + 1;
        qcur ^= 1;
        read_frags(xcur, qcur, 0, fa_x, fa_q);  // past the last stage: filler data, never multiplied
        __builtin_amdgcn_sched_barrier(0);
        multiply(fb_x, fb_q);
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (ks == nk - 1) {
            // fa is in flight: let it land before compiler-scheduled code (spills included) runs
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const bool partial = row0 + BM > n;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int tile_off = wm * BMW + 32 * t;
                const long long rowbase = row0 + tile_off;
                float xn[16];
                if (METRIC == RAG_METRIC_L2) {
                    // this lane's 16 rows: four runs of four (the LDS holds nothing but the rings)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long long row = rowbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        xn[r] = row < n ? xnorm[row] : 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    float key[16];
                    bool any = false;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        float v = acc[t][u][r];
                        if (METRIC == RAG_METRIC_L2) v = (2.f * v - xn[r]) - qn[u];
                        key[r] = v;
                        bool ok = v >= thr[u];
                        if (partial) ok = ok && (rowbase + ro < n);
                        any = any || ok;
                    }
                    if (__ballot(any) != 0ULL && my_q[u] < nq) {
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
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
                }
            }
        }
        if (++ks == nk) {
            ks = 0;
            row0 += (long long)S * tile_rows;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
}

inline size_t pipe_smem_need() { return (size_t)5 * 256 * 128; }  // all of the CU's 160 KiB

}  // namespace mfma
}  // namespace rag
