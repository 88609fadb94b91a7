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
//     The QUERIES are not read from their row-major array at all: tile_queries_kernel lays them out
//     beforehand as the LDS image itself, [query tile][K stage][256 rows x 128 B, swizzled], so that
//     every LDS-DMA instruction of a query stage reads 1 KiB of consecutive bytes.  The load path of a
//     CU moves such an instruction at 58 B/clk from the L2 and ANY multi-row shape (8 x 128 B, 16 x 64
//     B, 4 x 256 B) at 31 B/clk (benchmarks/glds_pattern.hip), and half of this kernel's LDS fill is
//     queries.  (The rows stay row-major: every other kernel reads that layout.)
// (2) The schedule inside a stage.  flat_mfma_kernel has all eight waves issue their loads, read their
//     fragments and multiply one after the other, so the matrix pipe idles while every wave of a SIMD
//     sits in load issue (an LDS-DMA instruction costs its wave 60-185 cycles) or waits for the LDS.
//     Here a stage is four K slices, each one batch of 8 MFMAs (32x32x16 bf16; fp32: 32 of 32x32x2)
//     whose operands were read from the LDS during the batch before (two fragment sets, fa / fb, filled
//     alternately by inline-assembly ds_read_b128 that the compiler's wait insertion does not see), and
//     the stage's eight LDS-DMA instructions are spread over the batches, two per batch, behind the
//     second and the sixth MFMA:
//
//       batch 1 (slice 0 of stage e):  reads of slice 1 in flight | loads: queries of stage e+1, second half
//       batch 2 (slice 1):             reads of slice 2           | loads: rows of stage e+2, first half
//       batch 3 (slice 2):             reads of slice 3           | loads: rows of stage e+2, second half
//       s_waitcnt vmcnt(4) lgkmcnt(0); s_barrier      (own loads of stage e+1 landed, all reads of e returned)
//       batch 4 (slice 3):             reads of slice 0 of e+1    | loads: queries of stage e+2, first half
//       [once per row tile: the filter epilogue]
//
//     One barrier per stage; all waits are counted (vmcnt(4): everything but the four row loads just
//     issued).  Tried and measured slower or equal: the two wave groups a phase apart ("ping-pong"),
//     different load slots per SIMD, four dedicated loader waves.
//
// Safety of the rings.  Write after read: the loads issued during stage e go into the row buffer and the
// query buffer that stage e-1 used (rows: ring of three, two stages ahead; queries: the half issued in
// batch 1 goes to the buffer of stage e-1, whose last reads returned before the barrier of stage e-1; the
// half issued in batch 4, behind the barrier of stage e, goes to the buffer stage e itself used, whose
// reads all returned before that barrier).  Read after write: stage e+1 is first read behind the barrier
// of stage e, and every wave waited for its own loads of stage e+1 in front of that barrier.  K order per
// accumulator is the same as in flat_mfma_kernel, so the scores are bit-identical.
#pragma once

#include <type_traits>

#ifndef RAG_PIPE_SLOT_A0  // LDS-DMA slots of a batch of 16 MFMAs (behind MFMA number ...): waves 0-3 (A) and 4-7 (B)
#define RAG_PIPE_SLOT_A0 3
#define RAG_PIPE_SLOT_A1 11
#define RAG_PIPE_SLOT_B0 3
#define RAG_PIPE_SLOT_B1 11
#endif
#ifndef RAG_PIPE_TWO_LOOPS
#define RAG_PIPE_TWO_LOOPS 0
#endif
#ifndef RAG_PIPE_LATE_SLEEP
#define RAG_PIPE_LATE_SLEEP 0
#endif
#ifndef RAG_PIPE_LEAN
#define RAG_PIPE_LEAN 1
#endif
#ifndef RAG_PIPE_UNIFORM_WID
#define RAG_PIPE_UNIFORM_WID 1
#endif

namespace rag {
namespace mfma {

// Queries as the pipelined kernel's LDS image: out[((c * nk + ks) * 2048 + slot) * 4 .. +4] (words) =
// 16-byte chunk (slot & 7) ^ swz(row) of K stage ks of query c * 256 + row, row = slot >> 3; queries
// past nq are zero rows.  Qw: nq x ld words (ld % 32 == 0).
template <int BKW>
__global__ __launch_bounds__(256) void tile_queries_kernel(const f32x4* __restrict__ Qw, int nq, int ld, int nk,
                                                           long long total_slots, f32x4* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total_slots) return;
    const int slot = (int)(i & 2047);
    const long long stage = i >> 11;
    const int ks = (int)(stage % nk);
    const int c = (int)(stage / nk);
    const int row = slot >> 3, q = c * 256 + row;
    const int chunk = (slot & 7) ^ swz<BKW>(row);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q < nq) v = Qw[((long long)q * ld + ks * BKW) / 4 + chunk];
    out[i] = v;
}

#ifdef RAG_PIPE_STAMPS
// diagnostic build (benchmarks/pipe_timing.hip -DRAG_PIPE_STAMPS): s_memtime totals per wave around the stage's
// wait and barrier (16x16x32 path): [0] cycles in s_waitcnt vmcnt, [1] in s_barrier, [2] whole K loop, [3] waves,
// [4] stages
__device__ unsigned long long g_pipe_stamps[8];
#endif

#ifdef RAG_PIPE_CHECK
// diagnostic build (benchmarks/pipe_timing.hip -DRAG_PIPE_CHECK): every LDS-DMA source address is checked
// against its array; an address outside is recorded (count, first offender) and replaced by zero16
__device__ unsigned long long g_pipe_check[8];
#define RAG_GLDS(ptr, dst, lo, nbytes, tag)                                                          \
    do {                                                                                             \
        const char* p_ = reinterpret_cast<const char*>(ptr);                                         \
        const char* lo_ = reinterpret_cast<const char*>(lo);                                         \
        const bool bad_ = p_ < lo_ || p_ + 16 > lo_ + (nbytes);                                      \
        if (bad_) {                                                                                  \
            if (atomicAdd(&g_pipe_check[0], 1ULL) == 0) {                                            \
                g_pipe_check[1] = (unsigned long long)(p_ - lo_);                                    \
                g_pipe_check[2] = (unsigned long long)(tag);                                         \
                g_pipe_check[3] = ((unsigned long long)blockIdx.x << 32) | threadIdx.x;              \
            }                                                                                        \
        }                                                                                            \
        glds16(bad_ ? reinterpret_cast<const char*>(zero16) : p_, dst);                              \
    } while (0)
#else
#define RAG_GLDS(ptr, dst, lo, nbytes, tag) glds16(ptr, dst)
#endif

// SAMPLE = false: the filter pass (thr_q, counts, cand_*[q][cap]).
// SAMPLE = true: the sample pass of the filter flow (flat.hip) -- only the row tiles t * tile_step are
//   visited and no threshold exists yet; each wave leaves the BEST score of every query over its 128 rows
//   of every visited tile in cand_key[q * cap + 2 * (visited tile) + wm] (cap = values per query), and the
//   j-th largest of a query's values is its threshold (kth_largest_kernel): j wave tiles hold a row that
//   good, i.e. it is the j-th best sampled row unless two of the best j share a wave tile, which only
//   lowers it.  No lists, no inserts: the pass costs what its tiles cost (it used to be the list-mode
//   tile kernel, whose every list has to fill up first: 1.33 ms of C3's 17.2 for 1.6 % of the rows).
template <int METRIC, bool BF, bool SAMPLE, bool M16>
__global__ __launch_bounds__(512, 2) void flat_pipe_kernel(
    const float* __restrict__ X, long long n, int ld, const float* __restrict__ Qp, int nq, int S, int col_tiles,
    float* __restrict__ cand_key, int* __restrict__ cand_id, const float* __restrict__ xnorm,
    const float* __restrict__ qnorm, const float* __restrict__ zero16, long long tile_step,
    const int* __restrict__ run_flag, const float* __restrict__ thr_q, unsigned int* __restrict__ counts, int cap) {
    if (run_flag != nullptr && *run_flag == 0) return;  // uniform, before any barrier
    constexpr int WN = 4, MT = 4, NT = 2, BK = 32;
    constexpr int NWV = 8, BMW = 128, BM = 256, BNW = 64, BN = 256;
    constexpr int CH = BK / 4, LDS_LD = BK;
    constexpr int NSX = 3;                     // ring depths: three row stages, two query stages
    constexpr int XQ = BM * CH / 64 / NWV;     // 4 LDS-DMA instructions per wave per X stage
    constexpr int QQ = BN * CH / 64 / NWV;     // 4 per Q stage
    constexpr int STG = BM * LDS_LD;           // words per stage buffer (32 KiB), X and Q alike
    static_assert(BM == BN, "one stage size");

    extern __shared__ __align__(16) unsigned char smem_raw[];
    float* Xs = reinterpret_cast<float*>(smem_raw);  // [NSX][BM][LDS_LD]
    float* Qs = Xs + NSX * STG;                      // [NSQ][BN][LDS_LD]   -- 160 KiB in all

    const int tid = threadIdx.x, lane = tid & 63;
#if RAG_PIPE_UNIFORM_WID
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS destinations and M0 need no per-load readfirstlane
#else
    const int wid = tid >> 6;
#endif
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
    // 32-bit row and stage counters throughout (the caller admits fewer than 2^31 rows): the K loop has no
    // registers to spare, scalar ones included
    const int nrows = (int)n;
    const int ntiles = (int)(((n + BM - 1) / BM + tile_step - 1) / tile_step);
    const int my_tiles = s < ntiles ? (ntiles - s + S - 1) / S : 0;
    const int tile_rows = BM * (int)tile_step;
    const int tile_stride = S * tile_rows;  // rows between this block's consecutive tiles (< 2^31: checked by the host)
    const int nk = ld / BK;  // the plan admits only ld % BK == 0
    const int total = my_tiles * nk;

    static_assert(!M16 || BF, "the 16x16x32 shape exists for bfloat16 only");
    constexpr int NQV = M16 ? 4 : NT;  // queries per lane: 16x16x32 tiles put query (lane & 15) of four 16-query blocks in a lane
    const int l16 = lane & 15, lq = lane >> 4;
    int my_q[NQV];
    float thr[NQV], qn[NQV];
#pragma unroll
    for (int u = 0; u < NQV; ++u) {
        my_q[u] = M16 ? q0 + BNW * wn + 16 * u + l16 : q0 + BNW * wn + 32 * u + li;
        thr[u] = FLT_MAX;
        if (!SAMPLE) thr[u] = my_q[u] < nq ? thr_q[my_q[u]] : FLT_MAX;
        qn[u] = 0.f;
        if (METRIC == RAG_METRIC_L2) qn[u] = my_q[u] < nq ? qnorm[my_q[u]] : 0.f;
        // the compiler waits for these two loads HERE, not with a vmcnt(0) in front of their first use in
        // the epilogue (where it would drain the LDS-DMA ring once per row tile)
        asm volatile("" : "+v"(thr[u]), "+v"(qn[u]));
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
    const float* Qt = Qp + (long long)c * (ld / BK) * STG;  // this query tile's stages (tile_queries_kernel)
    const long long row_stride = 256LL * ld;  // bytes between the rows of consecutive instructions
    // two prefetch cursors: rows run two stages ahead of the multiply, queries one.  x_ld(it) / q_ld(it)
    // issue ONE LDS-DMA instruction of the stage under the cursor (the K loop spreads them between its
    // MFMAs); x_next() / q_next() move the cursor on after the stage's last instruction.
    // Past the end the loads still issue (zero source, into a ring buffer nobody reads any more) so
    // that the number of outstanding operations per iteration stays what the counted waits assume.
#if RAG_PIPE_LEAN
    // Lean cursors: everything a load needs besides its lane offset is scalar state that changes once per
    // stage (x_next / q_next), so an x_ld / q_ld between two MFMAs is a few scalar adds, M0 and the load.
    // Past the end nothing advances any more: the filler loads re-read the last stage (rows) or keep
    // cycling through the query tile's stages -- valid addresses, into ring buffers nobody reads any more.
    if (my_tiles == 0) return;  // uniform per workgroup, before any barrier
    int px_tiles_left = my_tiles;
    int px_row0 = s * tile_rows;
    int px_ks = 0, px_buf = 0;
    const char* px_src = Xb + (long long)px_row0 * ld * 4;  // row px_row0, K stage px_ks
    // byte offset of this lane's 16 bytes from px_src, per instruction; in the last, partial row tile the rows
    // past the end re-read the last row (masked in the epilogue)
    unsigned xo[XQ];
    auto set_xo = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < XQ; ++it) {
            int r = lrow + 64 * it;
            r = px_row0 + r < nrows ? r : nrows - 1 - px_row0;
            xo[it] = 4u * (unsigned)(r * ld + lcol);
        }
    };
    set_xo();
    int pq_ks = 0, pq_buf = 0;
    const char* Qtb = reinterpret_cast<const char*>(Qt);
    const char* pq_src = Qtb;
    unsigned qo[QQ];  // this lane's 16 bytes of instruction `it`, from the stage's first byte
#pragma unroll
    for (int it = 0; it < QQ; ++it) {
        qo[it] = 16u * (unsigned)((it * NWV + wid) * 64 + lane);
    }
    auto x_ld = [&](int it) __attribute__((always_inline)) {
        float* dst = Xs + px_buf * STG + (it * NWV + wid) * 256;
#if defined(RAG_PIPE_EXP) && (RAG_PIPE_EXP & 1)
        if (px_tiles_left < my_tiles) return;  // ablation: no loads after the first row tile
#endif
        RAG_GLDS(px_src + (unsigned long long)xo[it], dst, Xb, (long long)n * ld * 4, 1);
    };
    auto x_next = [&]() __attribute__((always_inline)) {
        px_buf = px_buf + 1 == NSX ? 0 : px_buf + 1;
        if (px_tiles_left > 0) {
            if (++px_ks == nk) {
                if (--px_tiles_left > 0) {
                    px_ks = 0;
                    px_row0 += tile_stride;
                    px_src = Xb + (long long)px_row0 * ld * 4;
                    if (px_row0 + BM > nrows) set_xo();  // at most once: the last tile
                } else {
                    --px_ks;  // stay on the last stage
                }
            } else {
                px_src += BK * 4;
            }
        }
    };
    auto q_ld = [&](int it) __attribute__((always_inline)) {
        float* dst = Qs + pq_buf * STG + (it * NWV + wid) * 256;
#if defined(RAG_PIPE_EXP) && (RAG_PIPE_EXP & 1)
        return;
#endif
        // the stage is the LDS image itself: instruction `it` of this wave reads its 1 KiB in lane order
        // (the empty asm keeps the zero extension in this block, where instruction selection folds it into the
        // scalar-base + 32-bit-offset form of the load; hoisted out of the loop it becomes a 64-bit addition)
        unsigned o = qo[it];
        asm volatile("" : "+v"(o));
        RAG_GLDS(pq_src + (unsigned long long)o, dst, Qtb, (long long)nk * STG * 4, 3);
    };
    auto q_next = [&]() __attribute__((always_inline)) {
        pq_buf ^= 1;
        if (++pq_ks == nk) {
            pq_ks = 0;
            pq_src = Qtb;
        } else {
            pq_src += STG * 4;
        }
    };
#else
    int px_tiles_left = my_tiles;
    int px_row0 = s * tile_rows;
    int px_ks = 0, px_buf = 0;
    int pq_left = total;
    int pq_ks = 0, pq_buf = 0;
    auto x_ld = [&](int it) __attribute__((always_inline)) {
        float* dst = Xs + px_buf * STG + (it * NWV + wid) * 256;
#if defined(RAG_PIPE_EXP) && (RAG_PIPE_EXP & 1)
        if (px_tiles_left < my_tiles) return;  // ablation: no loads after the first row tile
#endif
        if (px_tiles_left > 0) {
            const char* base = Xb + ((long long)px_row0 * ld + px_ks * BK) * 4;
            if (px_row0 + BM <= nrows) {
                RAG_GLDS(base + it * row_stride + (unsigned long long)lane_off, dst, Xb, (long long)n * ld * 4, 1);
            } else {  // the last, partial row tile: rows past the end re-read the last row (masked in the epilogue)
                int r = lrow + 64 * it;
                r = px_row0 + r < nrows ? r : nrows - 1 - px_row0;
                RAG_GLDS(base + (unsigned long long)(4u * (unsigned)(r * ld + lcol)), dst, Xb, (long long)n * ld * 4, 2);
            }
        } else {
            glds16(zero16, dst);
        }
    };
    auto x_next = [&]() __attribute__((always_inline)) {
        px_buf = px_buf + 1 == NSX ? 0 : px_buf + 1;
        if (px_tiles_left > 0 && ++px_ks == nk) {
            px_ks = 0;
            px_row0 += tile_stride;
            --px_tiles_left;
        }
    };
    auto q_ld = [&](int it) __attribute__((always_inline)) {
        float* dst = Qs + pq_buf * STG + (it * NWV + wid) * 256;
#if defined(RAG_PIPE_EXP) && (RAG_PIPE_EXP & 1)
        if (pq_left + nk < total) return;
#endif
        // the stage is the LDS image itself: instruction `it` of this wave reads its 1 KiB in lane order
        if (pq_left > 0) RAG_GLDS(Qt + (long long)pq_ks * STG + ((it * NWV + wid) * 64 + lane) * 4, dst, Qt, (long long)nk * STG * 4, 3);
        else glds16(zero16, dst);
    };
    auto q_next = [&]() __attribute__((always_inline)) {
        pq_buf ^= 1;
        if (pq_left > 0) {
            pq_ks = pq_ks + 1 == nk ? 0 : pq_ks + 1;
            --pq_left;
        }
    };

#endif

    f32x16 acc[MT][NT];      // 32x32 tiles (unused with M16)
    f32x4 acc16[8][4];       // M16: 16x16 tiles, [row block of 16][query block of 16] (unused otherwise)
    if (M16) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc16[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    }

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
    // One K slice: 8 MFMAs (fp32: 32), with two LDS-DMA instructions issued in their shadow (ld0 behind
    // the second MFMA, ld1 behind the sixth): all eight issued together at the top of a stage keep every
    // wave of the CU in load issue at the same time (the CU's load path takes ~18-33 cycles per
    // instruction, 64 per stage) while the matrix pipe idles.
    auto multiply = [&](const f32x4(&ax)[MT], const f32x4(&aq)[NT], auto ld0, auto ld1) {
#if defined(RAG_PIPE_EXP) && (RAG_PIPE_EXP & 4)
        acc[0][0][0] += ax[0][0] * aq[0][0];  // ablation: no MFMA
        ld0();
        ld1();
        return;
#endif
        // slot i = behind the i-th MFMA.  (Tried: the waves of different SIMDs using different slots, so that
        // the CU's eight waves do not arrive at the load path together: 15.0 ms against 13.7 for C3.)
        auto slot = [&](int i) {
            if (i == 1) ld0();
            if (i == 5) ld1();
        };
        if (BF) {
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, ax[t]), __builtin_bit_cast(bf16x8, aq[u]), acc[t][u], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    slot(u * MT + t);
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int u = 0; u < NT; ++u)
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[t][cc], aq[u][cc], acc[t][u], 0, 0, 0);
                        if (cc == 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            slot(u * MT + t);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
        }
    };
    // M16: the same wave tile as 8 x 4 tiles of v_mfma_f32_16x16x32_bf16.  A K slice is 32 values (half a
    // stage: 4 chunks of 16 bytes, chunk = 4 * slice + (lane >> 4)); lane l supplies row / query (l & 15) of
    // every block of 16.  A slice's 32 MFMAs run as two batches of 16 (row blocks 0-3, then 4-7, against the
    // same four query fragments), so that a batch again is 256 matrix cycles with the next batch's operands
    // in flight: 4 reads (the other row half) or 8 (the next slice's first row half and its queries) per
    // batch, 24 per stage as with the 32x32x16 shape.  Same LDS image and swizzle: rows r and r + 16 share
    // swz(), and the sixteen lanes of each ds_read_b128 group touch sixteen different 16-byte bank groups.
    const unsigned xfrag16 = xs_lds + 4u * (unsigned)((wm * BMW + l16) * LDS_LD);
    const unsigned qfrag16 = qs_lds + 4u * (unsigned)((BNW * wn + l16) * LDS_LD);
    unsigned pcb16[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) pcb16[kk] = 16u * (unsigned)((kk * 4 + lq) ^ swz<BK>(l16));
    f32x4 pa_x[4], pb_x[4], pa_q[4], pb_q[4];
    static_assert(16 * LDS_LD * 4 == 2048, "fragment offsets below");
    auto read_x_lo = [&](int xbuf, int kk, f32x4(&ax)[4]) __attribute__((always_inline)) {  // row blocks 0-3 of slice kk
        const unsigned xa = xfrag16 + (unsigned)xbuf * (STG * 4u) + pcb16[kk];
#if defined(RAG_PIPE_EXP) && (RAG_PIPE_EXP & 2)
        return;
#endif
        asm volatile("ds_read_b128 %0, %1" : "=&v"(ax[0]) : "v"(xa));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=&v"(ax[1]) : "v"(xa));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=&v"(ax[2]) : "v"(xa));
        asm volatile("ds_read_b128 %0, %1 offset:6144" : "=&v"(ax[3]) : "v"(xa));
    };
    auto read_x_hi = [&](int xbuf, int kk, f32x4(&ax)[4]) __attribute__((always_inline)) {  // row blocks 4-7
        const unsigned xa = xfrag16 + (unsigned)xbuf * (STG * 4u) + pcb16[kk];
#if defined(RAG_PIPE_EXP) && (RAG_PIPE_EXP & 2)
        return;
#endif
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=&v"(ax[0]) : "v"(xa));
        asm volatile("ds_read_b128 %0, %1 offset:10240" : "=&v"(ax[1]) : "v"(xa));
        asm volatile("ds_read_b128 %0, %1 offset:12288" : "=&v"(ax[2]) : "v"(xa));
        asm volatile("ds_read_b128 %0, %1 offset:14336" : "=&v"(ax[3]) : "v"(xa));
    };
    auto read_q16 = [&](int qbuf, int kk, f32x4(&aq)[4]) __attribute__((always_inline)) {
        const unsigned qa = qfrag16 + (unsigned)qbuf * (STG * 4u) + pcb16[kk];
#if defined(RAG_PIPE_EXP) && (RAG_PIPE_EXP & 2)
        return;
#endif
        asm volatile("ds_read_b128 %0, %1" : "=&v"(aq[0]) : "v"(qa));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=&v"(aq[1]) : "v"(qa));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=&v"(aq[2]) : "v"(qa));
        asm volatile("ds_read_b128 %0, %1 offset:6144" : "=&v"(aq[3]) : "v"(qa));
    };
    // one batch: 16 MFMAs of 16 cycles; the two LDS-DMA slots sit where the 32x32x16 batch has them (behind
    // 64 and 192 matrix cycles)
    auto multiply16 = [&](auto half, auto late, const f32x4(&ax)[4], const f32x4(&aq)[4], auto ld0, auto ld1) __attribute__((always_inline)) {
        constexpr int H = decltype(half)::value;
        constexpr int S0 = decltype(late)::value ? RAG_PIPE_SLOT_B0 : RAG_PIPE_SLOT_A0;
        constexpr int S1 = decltype(late)::value ? RAG_PIPE_SLOT_B1 : RAG_PIPE_SLOT_A1;
#if defined(RAG_PIPE_EXP) && (RAG_PIPE_EXP & 4)
        acc16[4 * H][0][0] += ax[0][0] * aq[0][0];  // ablation: no MFMA
        ld0();
        ld1();
        return;
#endif
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc16[4 * H + t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, ax[t]), __builtin_bit_cast(bf16x8, aq[u]), acc16[4 * H + t][u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (u * 4 + t == S0) ld0();
                if (u * 4 + t == S1) ld1();
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    using Half0 = std::integral_constant<int, 0>;
    using Half1 = std::integral_constant<int, 1>;

#if defined(RAG_PIPE_EXP) && (RAG_PIPE_EXP & 2)
#pragma unroll
    for (int t = 0; t < MT; ++t) fa_x[t] = fb_x[t] = f32x4{1.f * lane, 2.f, 3.f, 4.f};
#pragma unroll
    for (int u = 0; u < NT; ++u) fa_q[u] = fb_q[u] = f32x4{1.f, 2.f * lane, 3.f, 4.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        pa_x[t] = pb_x[t] = f32x4{1.f * lane, 2.f, 3.f, 4.f};
        pa_q[t] = pb_q[t] = f32x4{1.f, 2.f * lane, 3.f, 4.f};
    }
#endif

    // The filter epilogue of a finished row tile (rows row0 .. row0 + BM): every accumulator against its
    // query's threshold, passing rows appended through the query's counter, accumulators cleared.
    auto epilogue = [&](int row0) {
        if (!SAMPLE && METRIC == RAG_METRIC_IP) {
            // Filter pass, inner product (as in epilogue16): first the best score per query over the lane's 64 rows
            // -- one v_max3 per two accumulators instead of a comparison, a select and an OR per accumulator; rows
            // past the end of a partial tile hold the last row again and are sorted out below -- and only a
            // (wave, query block) in which some lane has a passing score looks at its elements.
            float m[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) m[u] = -FLT_MAX;
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < NT; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) m[u] = fmaxf(m[u], acc[t][u][r]);
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                if (__ballot(m[u] >= thr[u]) != 0ULL && my_q[u] < nq) {
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        const int rowbase = row0 + wm * BMW + 32 * t;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ro = (r & 3) + 8 * (r >> 2) + 4 * lh;
                            const float key = acc[t][u][r];
                            if (key >= thr[u] && rowbase + ro < nrows) {
                                const unsigned int slot = atomicAdd(&counts[my_q[u]], 1u);
                                if (slot < (unsigned int)cap) {
                                    const long long o = (long long)my_q[u] * cap + slot;
                                    cand_key[o] = key;
                                    cand_id[o] = rowbase + ro;
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < NT; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
            return;
        }
        const bool partial = row0 + BM > nrows;
        float best[NT];  // SAMPLE: this lane's best score per query over the wave's 128 rows
#pragma unroll
        for (int u = 0; u < NT; ++u) best[u] = -FLT_MAX;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int tile_off = wm * BMW + 32 * t;
            const int rowbase = row0 + tile_off;
            float xn[16];
            if (METRIC == RAG_METRIC_L2) {
                // this lane's 16 rows: four runs of four (the LDS holds nothing but the rings)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rowbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    xn[r] = row < nrows ? xnorm[row] : 0.f;
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
                    bool ok = SAMPLE || v >= thr[u];
                    if (partial) ok = ok && (rowbase + ro < nrows);
                    if (SAMPLE) best[u] = ok ? fmaxf(best[u], v) : best[u];
                    any = any || ok;
                }
                if (!SAMPLE && __ballot(any) != 0ULL && my_q[u] < nq) {
                    // rare: append every passing row of this lane's query to its buffer
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (key[r] >= thr[u] && rowbase + ro < nrows) {
                            const unsigned int slot = atomicAdd(&counts[my_q[u]], 1u);
                            if (slot < (unsigned int)cap) {
                                const long long o = (long long)my_q[u] * cap + slot;
                                cand_key[o] = key[r];
                                cand_id[o] = rowbase + ro;
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

    // the same epilogue over 16x16 tiles: acc16[t][u][r] = row (row0 + wm * 128 + 16 t + 4 (lane >> 4) + r) against
    // query my_q[u]
    auto epilogue16 = [&](int row0) __attribute__((always_inline)) {
        const int rb0 = row0 + wm * BMW + 4 * lq;  // this lane's four consecutive rows of row block t: rb0 + 16 t ...
        auto key_of = [&](float v, float xn, int u) __attribute__((always_inline)) {
            return METRIC == RAG_METRIC_L2 ? (2.f * v - xn) - qn[u] : v;
        };
        if (SAMPLE) {
            const bool partial = row0 + BM > nrows;
            float best[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) best[u] = -FLT_MAX;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int rb = rb0 + 16 * t;
                float xn[4] = {0.f, 0.f, 0.f, 0.f};
                if (METRIC == RAG_METRIC_L2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) xn[r] = rb + r < nrows ? xnorm[rb + r] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = key_of(acc16[t][u][r], xn[r], u);
                        best[u] = (!partial || rb + r < nrows) ? fmaxf(best[u], v) : best[u];
                    }
                    acc16[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            // a query's other rows of the wave tile live in lanes ^ 16, ^ 32, ^ 48
            const int slot = 2 * (row0 / tile_rows) + wm;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float m = fmaxf(best[u], __shfl_xor(best[u], 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                if (lq == 0 && my_q[u] < nq) cand_key[(long long)my_q[u] * cap + slot] = m;
            }
            return;
        }
        if (METRIC == RAG_METRIC_L2) {
            // filter pass, L2: every key needs its row's norm; row block by row block
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int rb = rb0 + 16 * t;
                float xn[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) xn[r] = rb + r < nrows ? xnorm[rb + r] : 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float key[4];
                    bool any = false;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        key[r] = key_of(acc16[t][u][r], xn[r], u);
                        any = any || (key[r] >= thr[u] && rb + r < nrows);
                    }
                    if (__ballot(any) != 0ULL && my_q[u] < nq) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (key[r] >= thr[u] && rb + r < nrows) {
                                const unsigned int slot = atomicAdd(&counts[my_q[u]], 1u);
                                if (slot < (unsigned int)cap) {
                                    const long long o = (long long)my_q[u] * cap + slot;
                                    cand_key[o] = key[r];
                                    cand_id[o] = rb + r;
                                }
                            }
                        }
                    }
                    acc16[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            return;
        }
        // Filter pass, inner product.  First the best score per query over the lane's 32 rows (a maximum per
        // element instead of a comparison, a select and an OR; rows past the end of a partial tile hold the last
        // row again and are sorted out below); only a (wave, query block) in which some lane has a passing score
        // looks at its elements.
        float m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) m[u] = -FLT_MAX;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) m[u] = fmaxf(m[u], acc16[t][u][r]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (__ballot(m[u] >= thr[u]) != 0ULL && my_q[u] < nq) {
                // rare: append every passing row of this lane's query to its buffer
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int rb = rb0 + 16 * t;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float key = acc16[t][u][r];
                        if (key >= thr[u] && rb + r < nrows) {
                            const unsigned int slot = atomicAdd(&counts[my_q[u]], 1u);
                            if (slot < (unsigned int)cap) {
                                const long long o = (long long)my_q[u] * cap + slot;
                                cand_key[o] = key;
                                cand_id[o] = rb + r;
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc16[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // prologue: rows of stages 0 and 1 and queries of stage 0 in flight; slice 0 of stage 0 -> fa; first
    // half of the queries of stage 1
#pragma unroll
    for (int it = 0; it < XQ; ++it) x_ld(it);
    x_next();
#pragma unroll
    for (int it = 0; it < QQ; ++it) q_ld(it);
    q_next();
#pragma unroll
    for (int it = 0; it < XQ; ++it) x_ld(it);
    x_next();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XQ));
    __builtin_amdgcn_s_barrier();
    if constexpr (M16) {
        read_x_lo(0, 0, pa_x);
        read_q16(0, 0, pa_q);
        q_ld(0);
        q_ld(1);

        // The two waves that share a SIMD (wave w and w + 4, i.e. wm = 0 and 1) run the same program in step, so
        // with the same slots both would sit in LDS-DMA issue at the same time and leave the matrix pipe idle;
        // the second one issues its loads half a batch later.
        auto k_loop = [&](auto late) __attribute__((always_inline)) {
            int xcur = 0, qcur = 0, ks = 0;
            int row0 = s * tile_rows;
#ifdef RAG_PIPE_STAMPS
            unsigned long long st_vm = 0, st_bar = 0;
            const unsigned long long st_t0 = __builtin_readcyclecounter();
#endif
            for (int e = 0; e < total; ++e) {
                // the LDS-DMA instructions sit in the same four batches as below; the fragment sets are
                // pa (x: row blocks 0-3, q) and pb (x: row blocks 4-7, q of the stage's second slice)
                read_x_hi(xcur, 0, pb_x);
                asm volatile("s_waitcnt lgkmcnt(4)");  // slice 0: first row half and queries have landed
                __builtin_amdgcn_sched_barrier(0);
                multiply16(Half0{}, late, pa_x, pa_q, [&] { q_ld(2); }, [&] { q_ld(3); q_next(); });
                __builtin_amdgcn_sched_barrier(0);
                read_x_lo(xcur, 1, pa_x);
                read_q16(qcur, 1, pb_q);
                asm volatile("s_waitcnt lgkmcnt(8)");
                __builtin_amdgcn_sched_barrier(0);
                multiply16(Half1{}, late, pb_x, pa_q, [&] { x_ld(0); }, [&] { x_ld(1); });
                __builtin_amdgcn_sched_barrier(0);
                read_x_hi(xcur, 1, pb_x);
                asm volatile("s_waitcnt lgkmcnt(4)");
                __builtin_amdgcn_sched_barrier(0);
                multiply16(Half0{}, late, pa_x, pb_q, [&] { x_ld(2); }, [&] { x_ld(3); x_next(); });
                __builtin_amdgcn_sched_barrier(0);
#ifdef RAG_PIPE_STAMPS
                asm volatile("s_waitcnt lgkmcnt(0)");
                const unsigned long long ta_ = __builtin_readcyclecounter();
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XQ));
                const unsigned long long tb_ = __builtin_readcyclecounter();
                __builtin_amdgcn_s_barrier();
                const unsigned long long tc_ = __builtin_readcyclecounter();
                st_vm += tb_ - ta_;
                st_bar += tc_ - tb_;
                __builtin_amdgcn_sched_barrier(0);
#else
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(XQ));
                __builtin_amdgcn_s_barrier();
#endif
#if RAG_PIPE_LATE_SLEEP
                if (wm != 0) __builtin_amdgcn_s_sleep(RAG_PIPE_LATE_SLEEP);  // de-phase the two waves of a SIMD
#endif
                xcur = xcur + 1 == NSX ? 0 : xcur + 1;
                qcur ^= 1;
                read_x_lo(xcur, 0, pa_x);  // past the last stage: filler data, never multiplied
                read_q16(qcur, 0, pa_q);
                __builtin_amdgcn_sched_barrier(0);
                multiply16(Half1{}, late, pb_x, pb_q, [&] { q_ld(0); }, [&] { q_ld(1); });
                __builtin_amdgcn_sched_barrier(0);
                if (ks == nk - 1) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    epilogue16(row0);
                }
                if (++ks == nk) {
                    ks = 0;
                    row0 += tile_stride;
                }
            }
#ifdef RAG_PIPE_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) {
                atomicAdd(&g_pipe_stamps[0], st_vm);
                atomicAdd(&g_pipe_stamps[1], st_bar);
                atomicAdd(&g_pipe_stamps[2], __builtin_readcyclecounter() - st_t0);
                atomicAdd(&g_pipe_stamps[3], 1ULL);
                atomicAdd(&g_pipe_stamps[4], (unsigned long long)total);
            }
#endif
        };
#if RAG_PIPE_TWO_LOOPS
        if (wm == 0) k_loop(std::integral_constant<bool, false>{});
        else k_loop(std::integral_constant<bool, true>{});
#else
        k_loop(std::integral_constant<bool, false>{});
#endif
    } else {
        read_frags(0, 0, 0, fa_x, fa_q);
        q_ld(0);
        q_ld(1);

        int xcur = 0, qcur = 0, ks = 0;
        int row0 = s * tile_rows;
        for (int e = 0; e < total; ++e) {
            // loads of this iteration, in issue order: second half of the queries of stage e+1, rows of stage
            // e+2 (both into the buffers stage e-1 used), and behind the barrier the first half of the queries
            // of stage e+2 (into the buffer stage e used)
            read_frags(xcur, qcur, 1, fb_x, fb_q);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(FR));  // slice 0 has landed, slice 1 may be in flight
            __builtin_amdgcn_sched_barrier(0);
            multiply(fa_x, fa_q, [&] { q_ld(2); }, [&] { q_ld(3); q_next(); });
            __builtin_amdgcn_sched_barrier(0);
            read_frags(xcur, qcur, 2, fa_x, fa_q);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(FR));
            __builtin_amdgcn_sched_barrier(0);
            multiply(fb_x, fb_q, [&] { x_ld(0); }, [&] { x_ld(1); });
            __builtin_amdgcn_sched_barrier(0);
            read_frags(xcur, qcur, 3, fb_x, fb_q);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(FR));
            __builtin_amdgcn_sched_barrier(0);
            multiply(fa_x, fa_q, [&] { x_ld(2); }, [&] { x_ld(3); x_next(); });
            __builtin_amdgcn_sched_barrier(0);
            // own loads of stage e+1 have landed (all but the four row loads of stage e+2 issued above), every
            // fragment read of stage e has returned.  (Assembly, not the builtin: the compiler relaxes a
            // builtin wait that it believes unnecessary.)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(XQ));
            __builtin_amdgcn_s_barrier();
            xcur = xcur + 1 == NSX ? 0 : xcur + 1;
            qcur ^= 1;
            read_frags(xcur, qcur, 0, fa_x, fa_q);  // past the last stage: filler data, never multiplied
            __builtin_amdgcn_sched_barrier(0);
            multiply(fb_x, fb_q, [&] { q_ld(0); }, [&] { q_ld(1); });
            __builtin_amdgcn_sched_barrier(0);
            if (ks == nk - 1) {
                // fa is in flight: let it land before compiler-scheduled code (spills included) runs
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                epilogue(row0);
            }
            if (++ks == nk) {
                ks = 0;
                row0 += tile_stride;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup
}

inline size_t pipe_smem_need() { return (size_t)5 * 256 * 128; }  // all of the CU's 160 KiB

// words of the tiled query image for `col_tiles` query tiles of rows of ld words
inline long long pipe_tiled_words(int col_tiles, int ld) { return (long long)col_tiles * (ld / 32) * 256 * 32; }

inline int pipe_tile_queries(const float* Qw, int nq, int ld, int col_tiles, float* out, hipStream_t st) {
    const long long slots = pipe_tiled_words(col_tiles, ld) / 4;
    tile_queries_kernel<32><<<(unsigned)((slots + 255) / 256), 256, 0, st>>>(reinterpret_cast<const f32x4*>(Qw), nq, ld, ld / 32,
                                                                         slots, reinterpret_cast<f32x4*>(out));
    RAG_HIP(hipGetLastError());
    return RAG_OK;
}

}  // namespace mfma
}  // namespace rag
