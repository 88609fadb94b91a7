// flat_mfma_spec.h -- the filter-flow batch kernel with LOADER waves (block shape 11).
//
// Same contraction, LDS image, rings (three row stages + two query stages = 160 KiB), tiled query
// image, filter / sample epilogue and results as flat_pipe_kernel (flat_mfma_pipe.h); what differs is
// WHO issues the LDS-DMA.  There, each of the eight waves of a block issues 8 of a stage's 64 LDS-DMA
// instructions between its own MFMAs, and every one of them holds the wave for 100-185 cycles during
// which the wave issues no MFMA: ablation showed the costs ADD (MFMAs + barrier alone 8.65 ms for C3,
// + fragment reads 10.3, + loads 13.9) -- an in-order wave cannot slip an MFMA into the time its own
// load instruction is stuck in issue, and a tile-blocked row layout that doubles the load path's rate
// changes 13.8 ms into 13.4.  Here a block has 12 waves: waves 0-7 only read fragments and multiply
// (2 per SIMD, 128 rows x 64 queries each), waves 8-11 (one per SIMD) only issue the loads -- 16 per
// stage each -- and wait for them.  A stuck load then holds a wave that has nothing else to do, while
// the SIMD issues the other two waves' MFMAs.
//
// The price is the register budget: three waves per SIMD leave 168 VGPRs per wave, 128 of them
// accumulators, so fragments are single-buffered (24 registers); the LDS latency of a wave's reads is
// covered by its partner's MFMAs instead (8 MFMAs = 256 cycles per K slice).
//
// One barrier of all twelve waves per stage.  Ring safety is that of flat_pipe_kernel: the loads of
// iteration e (queries of stage e+1, rows of stage e+2) go into the buffers stage e-1 used, whose last
// fragment reads returned (lgkmcnt(0)) before the compute waves reached the barrier of iteration e-1;
// stage e+1 is read behind the barrier of iteration e, before which every loader waited for its own
// loads of it (vmcnt(8): all but the eight row loads just issued).
#pragma once

namespace rag {
namespace mfma {

template <int METRIC, bool BF, bool SAMPLE>
__global__ __launch_bounds__(768) void flat_spec_kernel(
    const float* __restrict__ X, long long n, int ld, const float* __restrict__ Qp, int nq, int S, int col_tiles,
    float* __restrict__ cand_key, int* __restrict__ cand_id, const float* __restrict__ xnorm,
    const float* __restrict__ qnorm, const float* __restrict__ zero16, long long tile_step,
    const int* __restrict__ run_flag, const float* __restrict__ thr_q, unsigned int* __restrict__ counts, int cap) {
    if (run_flag != nullptr && *run_flag == 0) return;  // uniform, before any barrier
    constexpr int WN = 4, MT = 4, NT = 2, BK = 32;
    constexpr int BMW = 128, BM = 256, BNW = 64, BN = 256;
    constexpr int LDS_LD = BK;
    constexpr int NSX = 3;             // ring depths: three row stages, two query stages
    constexpr int NLD = 4;             // loader waves
    constexpr int PL = 32 / NLD;       // 1 KiB pieces per loader per operand stage (a stage is 32 pieces)
    constexpr int STG = BM * LDS_LD;   // words per stage buffer (32 KiB), X and Q alike
    static_assert(BM == BN, "one stage size");

    extern __shared__ __align__(16) unsigned char smem_raw[];
    float* Xs = reinterpret_cast<float*>(smem_raw);  // [NSX][BM][LDS_LD]
    float* Qs = Xs + NSX * STG;                      // [2][BN][LDS_LD]   -- 160 KiB in all

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

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
    const int nrows = (int)n;
    const int ntiles = (int)(((n + BM - 1) / BM + tile_step - 1) / tile_step);
    const int my_tiles = s < ntiles ? (ntiles - s + S - 1) / S : 0;
    const int tile_rows = BM * (int)tile_step;
    const int tile_stride = S * tile_rows;  // rows between this block's consecutive tiles
    const int nk = ld / BK;                 // the plan admits only ld % BK == 0
    const int total = my_tiles * nk;

    if (wid >= 8) {
        // ================================================================== loader waves
        __builtin_amdgcn_s_setprio(2);  // their few instructions go out ahead of the compute waves' streams
        const int l = wid - 8;
        // piece p = it * NLD + l of a stage: LDS slots p*64 .. p*64+63 = rows p*8 .. p*8+7, whole 128-byte lines;
        // the swizzle is applied to the source address ((row >> 1) & 7 repeats every 16 rows: the same for every it)
        const int lrow = l * 8 + (lane >> 3);
        const int lcol = 4 * ((lane & 7) ^ swz<BK>(lrow));
        const unsigned lane_off = 4u * (unsigned)(lrow * ld + lcol);
        const char* Xb = reinterpret_cast<const char*>(X);
        const float* Qt = Qp + (long long)c * nk * STG;  // this query tile's stages (tile_queries_kernel)
        const long long piece_stride = (long long)(NLD * 8) * ld * 4;  // bytes between the rows of consecutive pieces
        int px_tiles_left = my_tiles, px_row0 = s * tile_rows, px_ks = 0, px_buf = 0;
        int pq_left = total, pq_ks = 0, pq_buf = 0;
        // Past the end the loads still issue (zero source, into a ring buffer nobody reads any more) so that
        // the number of outstanding operations per iteration stays what the counted wait assumes.
        auto issue_x = [&]() {
            float* xs = Xs + px_buf * STG;
            px_buf = px_buf + 1 == NSX ? 0 : px_buf + 1;
            if (px_tiles_left > 0) {
                const char* base = Xb + ((long long)px_row0 * ld + px_ks * BK) * 4;
                if (px_row0 + BM <= nrows) {
#pragma unroll
                    for (int it = 0; it < PL; ++it)
                        glds16(base + it * piece_stride + (unsigned long long)lane_off, xs + (it * NLD + l) * 256);
                } else {  // the last, partial row tile: rows past the end re-read the last row (masked in the epilogue)
#pragma unroll
                    for (int it = 0; it < PL; ++it) {
                        int r = lrow + NLD * 8 * it;
                        r = px_row0 + r < nrows ? r : nrows - 1 - px_row0;
                        glds16(base + (unsigned long long)(4u * (unsigned)(r * ld + lcol)), xs + (it * NLD + l) * 256);
                    }
                }
                if (++px_ks == nk) {
                    px_ks = 0;
                    px_row0 += tile_stride;
                    --px_tiles_left;
                }
            } else {
#pragma unroll
                for (int it = 0; it < PL; ++it) glds16(zero16, xs + (it * NLD + l) * 256);
            }
        };
        auto issue_q = [&]() {
            float* qs = Qs + pq_buf * STG;
            pq_buf ^= 1;
            if (pq_left > 0) {
                // the stage is the LDS image itself: every piece is 1 KiB of consecutive bytes
                const float* base = Qt + (long long)pq_ks * STG + (l * 64 + lane) * 4;
#pragma unroll
                for (int it = 0; it < PL; ++it) glds16(base + it * NLD * 256, qs + (it * NLD + l) * 256);
                pq_ks = pq_ks + 1 == nk ? 0 : pq_ks + 1;
                --pq_left;
            } else {
#pragma unroll
                for (int it = 0; it < PL; ++it) glds16(zero16, qs + (it * NLD + l) * 256);
            }
        };
        issue_x();
        issue_q();
        issue_x();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PL));  // stage 0 landed, rows of stage 1 in flight
        __builtin_amdgcn_s_barrier();
        for (int e = 0; e < total; ++e) {
            issue_q();  // queries of stage e+1 -> the buffer stage e-1 used
            issue_x();  // rows of stage e+2    -> the buffer stage e-1 used
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PL));  // own loads of stage e+1 have landed
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
        return;
    }

    // ====================================================================== compute waves
    const int li = lane & 31, lh = lane >> 5;
    const int wn = wid % WN, wm = wid / WN;
    float thr[NT], qn[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int q = q0 + BNW * wn + 32 * u + li;
        thr[u] = FLT_MAX;
        if (!SAMPLE) thr[u] = q < nq ? thr_q[q] : FLT_MAX;
        qn[u] = 0.f;
        if (METRIC == RAG_METRIC_L2) qn[u] = q < nq ? qnorm[q] : 0.f;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    // fragment addresses (bytes in the LDS aperture); the K slice's chunk offset is recomputed per slice
    // (two integer operations) rather than held in four registers
    const int sw = swz<BK>(li);
    const unsigned xs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)Xs;
    const unsigned qs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)Qs;
    const unsigned xfrag = xs_lds + 4u * (unsigned)((wm * BMW + li) * LDS_LD);
    const unsigned qfrag = qs_lds + 4u * (unsigned)((BNW * wn + li) * LDS_LD);
    f32x4 fx[MT], fq[NT];
    auto read_frags = [&](int xbuf, int qbuf, int kk) {
        const unsigned pcb = 16u * (unsigned)((kk * 2 + lh) ^ sw);
        const unsigned xa = xfrag + (unsigned)xbuf * (STG * 4u) + pcb;
        const unsigned qa = qfrag + (unsigned)qbuf * (STG * 4u) + pcb;
        static_assert(32 * LDS_LD * 4 == 4096 && MT == 4 && NT == 2, "fragment offsets below");
        asm volatile("ds_read_b128 %0, %1" : "=&v"(fq[0]) : "v"(qa));
        asm volatile("ds_read_b128 %0, %1" : "=&v"(fx[0]) : "v"(xa));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=&v"(fq[1]) : "v"(qa));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=&v"(fx[1]) : "v"(xa));
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=&v"(fx[2]) : "v"(xa));
        asm volatile("ds_read_b128 %0, %1 offset:12288" : "=&v"(fx[3]) : "v"(xa));
    };
    auto multiply = [&]() {
        if (BF) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < NT; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, fx[t]), __builtin_bit_cast(bf16x8, fq[u]), acc[t][u], 0, 0, 0);
        } else {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int u = 0; u < NT; ++u)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(fx[t][cc], fq[u][cc], acc[t][u], 0, 0, 0);
        }
    };

    auto epilogue = [&](int row0) {
        // The lane id is re-read through volatile assembly: everything per-lane below (query numbers, row
        // offsets) is then computed HERE, once per row tile, instead of being hoisted out of the K loop into
        // registers the loop does not have (168 per wave, 152 of them accumulators and fragments).
        int lane_e;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
        const int li = lane_e & 31, lh = lane_e >> 5;
        int my_q[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) my_q[u] = q0 + BNW * wn + 32 * u + li;
        const bool partial = row0 + BM > nrows;
        float best[NT];  // SAMPLE: this lane's best score per query over the wave's 128 rows
#pragma unroll
        for (int u = 0; u < NT; ++u) best[u] = -FLT_MAX;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int rowbase = row0 + wm * BMW + 32 * t;
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                bool any = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rowbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float v = acc[t][u][r];
                    if (METRIC == RAG_METRIC_L2) {
                        const float xn = row < nrows ? xnorm[row] : 0.f;
                        v = (2.f * v - xn) - qn[u];
                        acc[t][u][r] = v;
                    }
                    bool ok = SAMPLE || v >= thr[u];
                    if (partial) ok = ok && row < nrows;
                    if (SAMPLE) best[u] = ok ? fmaxf(best[u], v) : best[u];
                    any = any || ok;
                }
                if (!SAMPLE && __ballot(any) != 0ULL && my_q[u] < nq) {
                    // rare: append every passing row of this lane's query to its buffer
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rowbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const float v = acc[t][u][r];
                        if (v >= thr[u] && row < nrows) {
                            const unsigned int slot = atomicAdd(&counts[my_q[u]], 1u);
                            if (slot < (unsigned int)cap) {
                                const long long o = (long long)my_q[u] * cap + slot;
                                cand_key[o] = v;
                                cand_id[o] = row;
                            }
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
            }
        }
        if (SAMPLE) {
            // the other 16 rows of every 32-row tile live in lane ^ 32; visited tile number = row0 / tile_rows
            const int slot = 2 * (row0 / tile_rows) + wm;
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const float m = fmaxf(best[u], __shfl_xor(best[u], 32, 64));
                if (lh == 0 && my_q[u] < nq) cand_key[(long long)my_q[u] * cap + slot] = m;
            }
        }
    };

    __builtin_amdgcn_s_barrier();  // stage 0 is in the LDS
    int xcur = 0, qcur = 0, ks = 0;
    int row0 = s * tile_rows;
    for (int e = 0; e < total; ++e) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            read_frags(xcur, qcur, kk);
            asm volatile("s_waitcnt lgkmcnt(0)");
            __builtin_amdgcn_sched_barrier(0);
            multiply();
            __builtin_amdgcn_sched_barrier(0);
        }
        // every fragment read of stage e has returned (the MFMAs above consumed them)
        __builtin_amdgcn_s_barrier();
        xcur = xcur + 1 == NSX ? 0 : xcur + 1;
        qcur ^= 1;
        if (ks == nk - 1) epilogue(row0);
        if (++ks == nk) {
            ks = 0;
            row0 += tile_stride;
        }
    }
}

}  // namespace mfma
}  // namespace rag
