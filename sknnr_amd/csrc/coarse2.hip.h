// coarse2.hip.h -- second-generation MFMA pre-filter for feature spaces up to 64 wide (KS <= 4): lists of 2 / 6 / 8 / 16
// per lane (coarse2_supported), 1 .. 31 neighbours searched (more than a list holds: pooled lists, pair_union_rank below).
//
// Same contract as coarse_kernel (coarse.hip.h): for every query, the M smallest ranking values
//   v(q, r) ~= |r'|^2 - 2 q'.r'   seen by each of the two lanes that own the query, with the J-th
// smallest of both lists together (J = neighbours searched + 1) bounding every row that is not listed.
// What changed, and why (measurements: DESIGN.md section 4, profiles/r02_*):
//
//   * On gfx950 a wave's VALU instructions overlap an MFMA only when they do not depend on it and sit
//     behind it in the SAME wave's stream (about four per 32x32x16 MFMA are free; VALU of another wave
//     of the SIMD is not overlapped, whatever the wave priorities).  The sweep is therefore scheduled by
//     hand inside the wave, tile by tile (a unit = one 32-reference tile x one 32-query block, two units
//     per tile): the main products of both units go out back to back and the skip test of the first unit
//     sits behind the MFMAs of the second (tile_issue_and_test).  The |r'|^2 C operand is read from LDS
//     straight into the accumulator registers, which pays for the second accumulator set.  (Pipelining
//     across tiles -- the products of unit u+1 before the test of unit u -- exposes one LDS wait per unit
//     and measured slower: DESIGN.md section 4.2.)
//   * The two correction products (lo.hi + hi.lo, four MFMAs per visited unit) are gone from the sweep.
//     A unit is visited when some value's MAIN product is below threshold + margin (margin >= the size of
//     the correction, as before); such values are queued with their main value, and the batched flush
//     computes the correction of each queued entry exactly once with packed f16 dot products
//     (v_dot2c_f32_f16) on the entry's fragments fetched from the image in L2 -- the two lanes that share
//     a query each hold half of its K range and exchange partial sums.  Rejected values satisfy
//     main >= threshold + margin, hence corrected >= threshold: the same guarantee the certificate uses.
//   * LDS stages hold only hi fragments + |r'|^2 (16 tiles per stage: half the barriers).
//   * Seeding: the first kSeedTiles tiles are swept once with a one-insertion-per-unit rule (each lane
//     inserts its smallest main value) to obtain a valid starting threshold (J-th smallest seed value +
//     2 x margin); the real sweep then starts with a tight threshold instead of taking ~1000 hits per
//     q-block in the first tiles.
//   * WAVES = 16 for the bulk of a call, 4 for the rows of a thin last round and for small calls (host side:
//     launch_coarse2_ks).
// Round 3 (DESIGN.md sections 4.1, 4.2; profiles/r03_*):
//   * the image is in CELL order and the rows of a call are bucketed by cell (bucket.hip.h): a workgroup works on positions,
//     starts its sweep (and takes its seeds) at the stage its middle row's cell names;
//   * (round 3: the flush walked the entries of a query's two lanes as one sequence, two per trip; round 4: wave flush)
//   * thresholds of a rank beyond one list over the two lists of a query kept as one pool (template parameter E):
//     6 .. 31 neighbours on lists of 6 / 8 / 16.
#pragma once
#include "coarse.hip.h"

// Timing experiments that produce WRONG results (sweeps without visits, corrections or stage barriers) compile only in
// development builds that say so; the product build (sknnr_amd/_build.py) never defines any of them.
#if !defined(SKNNR_EXPERIMENTS) && (defined(SKNNR_V2_SWEEP_ONLY) || defined(SKNNR_V2_NO_BARRIER) || defined(SKNNR_V2_NO_SEED))
#error "SKNNR_V2_SWEEP_ONLY / NO_BARRIER / NO_SEED are timing experiments with wrong results: add -DSKNNR_EXPERIMENTS"
#endif

namespace sknnr {

typedef _Float16 half2v __attribute__((ext_vector_type(2)));

__host__ __device__ constexpr int tile2_bytes(int ks) { return ks * 1024 + 128; }
#ifndef SKNNR_V2_WAVES
#define SKNNR_V2_WAVES 16
#endif
#ifndef SKNNR_V2_UNROLL
#define SKNNR_V2_UNROLL 1
#endif
#ifndef SKNNR_V2_TPS
#define SKNNR_V2_TPS (SKNNR_V2_WAVES == 16 ? 16 : 8)
#endif
// 16-wave workgroups: one per CU, 16 tiles per stage; 8-wave workgroups: two per CU (each with its own stages of
// 8 tiles), so that a workgroup waiting at its stage barrier for a flushing wave leaves the SIMDs to the other one
__host__ __device__ constexpr int tiles_per_stage2(int ks) { return ks <= 2 ? SKNNR_V2_TPS : SKNNR_V2_TPS / 2; }
#ifndef SKNNR_SEED_TILES
#define SKNNR_SEED_TILES 64
#endif
constexpr int kSeedTiles = SKNNR_SEED_TILES;
// Tiles of the seed window for an image of n_tiles tiles: at most kSeedTiles and at most an eighth of the image (a
// 10,000-row set has 320 tiles: 64 swept twice would be a fifth of its sweep), whole stages, at least one.
__host__ __device__ constexpr int seed_tiles_for(long n_tiles, int tps) {
    const long want = n_tiles / 8 < kSeedTiles ? n_tiles / 8 : kSeedTiles;
    const long stages = want / tps < 1 ? 1 : want / tps;
    return (int)(stages * tps);
}
constexpr int kCoarse2Waves = SKNNR_V2_WAVES;
constexpr int kCoarse2Nqb = 2;
#ifndef SKNNR_V2_QCAP
#define SKNNR_V2_QCAP 5
#endif
constexpr int kQueueCap = SKNNR_V2_QCAP;  // entries per lane and q-block in LDS ([entry][lane] 8-byte pairs)
#ifndef SKNNR_V2_FLUSH_AT
#define SKNNR_V2_FLUSH_AT 3
#endif
constexpr int kQueueFlushAt = SKNNR_V2_FLUSH_AT;  // a visit ends with a flush once some lane holds this many
// per wave: the queues [q-block][entry][lane] 8-byte pairs, then the row behind every lane's column [q-block][lane] (4 bytes)
__host__ __device__ constexpr int queue2_bytes_per_wave() { return kCoarse2Nqb * kQueueCap * 64 * 8 + kCoarse2Nqb * 64 * 4; }
// measured against coarse_kernel on 4.19M x 50k rows (profiles/r02_v2_vs_v1.txt): 6-entry lists win for KS <= 4, 8-entry
// lists for KS <= 3 (KS = 4 with 8-entry lists spills 88 bytes and loses)
#ifndef SKNNR_V2_M2
#define SKNNR_V2_M2 1  // one neighbour (lists of 2) on the second-generation kernel, up to 32 features
#endif
#ifndef SKNNR_V2_M12
#define SKNNR_V2_M12 1  // 16 .. 23 neighbours on pooled lists of 12 (round 4), scripts/k16_probe.py, 2M x 50k rows: one K-step, 16 waves:
                        // 98 -> 113 Mq/s at k = 20; two K-steps at 12 waves (at 16 the instance spills 20 registers: 65 Mq/s):
                        // k = 16 / 20 / 23: 87 / 85 / 71 -> 89.5 / 90 / 77 Mq/s against lists of 16
#endif
#ifndef SKNNR_V2_M16
#define SKNNR_V2_M16 1  // 16 .. 31 neighbours (lists of 16) on the second-generation kernel, 12 waves
#endif
__host__ __device__ constexpr bool coarse2_supported(int ks, int m) {
    return (m == 6 && ks <= 4) || (m == 8 && ks <= 4) || (SKNNR_V2_M2 && m == 2 && ks <= 2) || (SKNNR_V2_M16 && m == 16 && ks <= 4) ||
           (SKNNR_V2_M12 && m == 12 && ks <= 4);
}
// waves per workgroup of the bulk launch: 16 (4 per SIMD, <= 128 VGPRs); lists of 16, lists of 12 from two K-steps on and lists
// of 8 at four K-steps need 12 (3 per SIMD, <= 168).  Round 4 (end): the three- and four-K-step instances of lists of 8 / 12 / 16
// at 12 waves (6 .. 60 spilled registers) replace the first-generation kernel for 8 .. 31 neighbours at 33 .. 64 features:
// scripts/wide_k_probe.py, 1M x 50k rows: k = 10 / 14 at 64 features 60.5 / 54 -> 79 / 69 Mq/s, k = 20 / 25 at 48 features
// 32 / 30 -> 68.5 / 52, k = 20 / 30 at 64 features 32 / 27 -> 62 / 32
#ifndef SKNNR_V2_M16_WAVES
#define SKNNR_V2_M16_WAVES 12
#endif
__host__ __device__ constexpr int coarse2_waves(int ks, int m) {
    return (m == 16 || (m == 12 && ks >= 2) || (m == 8 && ks >= 4)) ? SKNNR_V2_M16_WAVES : kCoarse2Waves;
}

// sum_j x[j] * y[j] over one 8-element fragment, f32 accumulate (v_dot2c_f32_f16)
__device__ __forceinline__ float dot8(const half8& x, const half8& y, float acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const half2v p = {x[2 * i], x[2 * i + 1]}, q = {y[2 * i], y[2 * i + 1]};
        acc = __builtin_amdgcn_fdot2(p, q, acc, false);
    }
    return acc;
}

// The (M + E)-th smallest entry of the union of the two sorted M-lists owned by lanes l and l+32 (E >= 1: a rank beyond
// the length of one list -- lists of 16 serve up to 30 neighbours):
//     max(a_{E-1}, b_{E-1}, max_{i = E .. M-1} min(a_i, b_{M+E-1-i}))
// (k-th smallest of a union = max over i + j = k - 1 of min(a_i, b_j), entries past the end of a list counting as +inf).
// Both lanes get the same value.  For this to bound what the LISTS let go of as well as what the skip test rejects, the
// two lists of a query are kept as one pool of 2 M entries (flush: what one list lets go of -- the entry an insertion
// displaces, or a candidate it does not take -- is handed to the other list at once unless it is already above the
// threshold), so that everything dropped is >= the final threshold or has M entries below it in BOTH lists: >= the 2M-th
// smallest of the pool, which no rank of the union exceeds.
template <int M, int E>
__device__ __forceinline__ float pair_union_rank(const float (&vals)[M]) {
    if constexpr (E == 0) {
        return pair_union_rank_m<M>(vals);
    } else {
        static_assert(E >= 1 && E <= M, "rank between M + 1 and 2 M");
        float u = fmaxf(vals[E - 1], __shfl_xor(vals[E - 1], 32, 64));
#pragma unroll
        for (int i = E; i < M; ++i) u = fmaxf(u, fminf(vals[i], __shfl_xor(vals[M + E - 1 - i], 32, 64)));
        return u;
    }
}
// Ranks beyond the length of a list (one kernel instance each), by neighbours searched (kk):
//   lists of 6 : rank  9 for kk = 6 .. 7
//   lists of 8 : rank 12 for kk = 8 .. 10, 15 for 11 .. 13, 16 (the whole pool) for 14 .. 15
//   lists of 12: rank 22 for kk = 16 .. 20, 24 (the whole pool) for 21 .. 23
//   lists of 16: rank 22 for kk = 16 .. 20, 27 for 21 .. 25, 31 for 26 .. 30, 32 for 31
// A rank above kk + 1 is a looser threshold, never a wrong one, and the certificate likes the slack: the gaps between a
// query's consecutive neighbour distances shrink with the rank.
__host__ __device__ constexpr int coarse2_rank_extra(int m_list, int kk) {
    if (kk + 1 <= m_list) return 0;
    if (m_list == 6) return 3;
    if (m_list == 8) return kk <= 10 ? 4 : (kk <= 13 ? 7 : 8);
    if (m_list == 12) return kk <= 20 ? 10 : 12;
    return kk <= 20 ? 6 : (kk <= 25 ? 11 : (kk <= 30 ? 15 : 16));
}
constexpr int kCoarse2MaxKK6 = 7, kCoarse2MaxKK8 = 15, kCoarse2MaxKK12 = 23;
constexpr int kCoarse2MaxKK16 = 31;

// The skip test of a unit: the minimum of its sixteen main values, as raw instructions (hipcc neither interleaves
// independent VALU work between dependent MFMAs nor sees through the min instructions, so the tile step below is
// written out in asm: MFMAs back to back, the tree of the first unit behind the MFMAs of the second -- about four
// VALU instructions per MFMA are what the matrix pipe covers, scripts/microbench/mfma_shadow.hip).
//   x[0..15] : main products of the unit
//   g[0..4]  : minima of {0-2}, {3-5}, {6-8}, {9-11}, {12-15};  m: minimum of all sixteen
// Hazard (guide section 5.7): a VALU read of an MFMA result needs 11 wait states after the 8-pass MFMA was issued,
// and nothing pads the inside of an asm statement: every tree is preceded by explicit s_nop padding that covers
// the distance to the last MFMA writing its inputs (11 states in tile_issue_and_test, where one independent MFMA
// sits in between; 16 in step_test_only, where the unit's last MFMA was the previous instruction).
#define SKNNR_TREE_A                                   \
    "v_min3_f32 %[g0], %[x0], %[x1], %[x2]\n\t"        \
    "v_min3_f32 %[g1], %[x3], %[x4], %[x5]\n\t"        \
    "v_min3_f32 %[g2], %[x6], %[x7], %[x8]\n\t"        \
    "v_min3_f32 %[g3], %[x9], %[x10], %[x11]\n\t"
#define SKNNR_TREE_B                                   \
    "v_min3_f32 %[g4], %[x12], %[x13], %[x14]\n\t"     \
    "v_min3_f32 %[m], %[g0], %[g1], %[g2]\n\t"         \
    "v_min_f32 %[g4], %[g4], %[x15]\n\t"               \
    "s_nop 0\n\t"                                      \
    "v_min3_f32 %[m], %[m], %[g3], %[g4]\n\t"
#define SKNNR_TREE_OUT [g0] "=&v"(g[0]), [g1] "=&v"(g[1]), [g2] "=&v"(g[2]), [g3] "=&v"(g[3]), [g4] "=&v"(g[4]), [m] "=&v"(m)
#define SKNNR_TREE_IN                                                                                             \
    [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), \
        [x7] "v"(x[7]), [x8] "v"(x[8]), [x9] "v"(x[9]), [x10] "v"(x[10]), [x11] "v"(x[11]), [x12] "v"(x[12]),       \
        [x13] "v"(x[13]), [x14] "v"(x[14]), [x15] "v"(x[15])

// The min tree alone (second unit of a tile: its MFMAs have just been issued).
__device__ __forceinline__ void step_test_only(const floatx16& x, float (&g)[5], float& m) {
    asm volatile("s_nop 7\n\ts_nop 7\n\t" SKNNR_TREE_A SKNNR_TREE_B : SKNNR_TREE_OUT : SKNNR_TREE_IN);
}
// Both units of a tile, hand scheduled: the main MFMAs of the two units go out back to back and the min tree
// of the FIRST unit sits behind the MFMAs of the second one -- independent VALU work in the shadow of the
// wave's own MFMAs.  `c` holds |r'|^2 on entry and the second unit's values on exit (C operand in place).
template <int KS>
__device__ __forceinline__ void tile_issue_and_test(floatx16& a, floatx16& c, const half8 (&ah)[KS], const half8 (&p)[KS],
                                                    const half8 (&q)[KS], float (&g)[5], float& m) {
    static_assert(KS >= 1 && KS <= 4, "hand-scheduled for one to four K-steps");
    // first unit: KS MFMAs into `a` (C operand = |r'|^2 from `c`), then the first MFMA of the second unit
    asm volatile("v_mfma_f32_32x32x16_f16 %[a], %[h0], %[p0], %[c]\n\t" : [a] "=&v"(a) : [c] "v"(c), [h0] "v"(ah[0]), [p0] "v"(p[0]));
#pragma unroll
    for (int s = 1; s < KS; ++s)
        asm volatile("v_mfma_f32_32x32x16_f16 %[a], %[h], %[p], %[a]\n\t" : [a] "+v"(a) : [h] "v"(ah[s]), [p] "v"(p[s]));
    asm volatile("v_mfma_f32_32x32x16_f16 %[c], %[h0], %[q0], %[c]\n\t"
                 // 11 wait states must separate the last MFMA of `a` from the first read of its result (8-pass
                 // MFMA -> VALU read, guide section 5.7); counting the MFMA in between as ONE, twelve idle slots
                 // make that certain whatever the issue timing (the wave is waiting for its MFMAs anyway)
                 "s_nop 7\n\ts_nop 3\n\t"
                 : [c] "+v"(c)
                 : [h0] "v"(ah[0]), [q0] "v"(q[0]), "v"(a));
    const floatx16& x = a;
    asm volatile(SKNNR_TREE_A
                 : [g0] "=&v"(g[0]), [g1] "=&v"(g[1]), [g2] "=&v"(g[2]), [g3] "=&v"(g[3])
                 : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]),
                   [x6] "v"(x[6]), [x7] "v"(x[7]), [x8] "v"(x[8]), [x9] "v"(x[9]), [x10] "v"(x[10]), [x11] "v"(x[11]));
    if constexpr (KS >= 2)
        asm volatile("v_mfma_f32_32x32x16_f16 %[c], %[h1], %[q1], %[c]\n\t" : [c] "+v"(c) : [h1] "v"(ah[1]), [q1] "v"(q[1]));
    asm volatile("v_min3_f32 %[g4], %[x12], %[x13], %[x14]\n\t"
                 "v_min3_f32 %[m], %[g0], %[g1], %[g2]\n\t"
                 "v_min_f32 %[g4], %[g4], %[x15]\n\t"
                 "s_nop 0\n\t"
                 "v_min3_f32 %[m], %[m], %[g3], %[g4]\n\t"
                 : [g4] "=&v"(g[4]), [m] "=&v"(m)
                 : [x12] "v"(x[12]), [x13] "v"(x[13]), [x14] "v"(x[14]), [x15] "v"(x[15]), [g0] "v"(g[0]), [g1] "v"(g[1]),
                   [g2] "v"(g[2]), [g3] "v"(g[3]));
#pragma unroll
    for (int s = 2; s < KS; ++s)
        asm volatile("v_mfma_f32_32x32x16_f16 %[c], %[h], %[q], %[c]\n\t" : [c] "+v"(c) : [h] "v"(ah[s]), [q] "v"(q[s]));
}

// WAVES = 16 (one workgroup of 1024 query rows per CU) for the bulk of a call; WAVES = 4 (256 rows) for the rows of a
// last, thinly filled round of workgroups and for small calls: spread over four times as many CUs with one wave per
// SIMD, where a wave no longer shares its matrix pipe (host side: launch_coarse2_ks).
// E > 0: thresholds of rank M + E (pair_union_rank), the two lists of a query kept as one pool, no sentinels.
template <int KS, int M, int WAVES = kCoarse2Waves, int E = 0>
__global__ void __launch_bounds__(WAVES * 64, WAVES == 12 ? 3 : (WAVES == 8 || M > 8 ? 2 : 4))
coarse2_kernel(const char* __restrict__ rhi,    // n_stages * TPS records [hi: KS KiB][|r'|^2: 128 B]
               const char* __restrict__ rlo,    // n_stages * TPS records [lo: KS KiB]
               int n_stages,
               const uint4* __restrict__ qimg,  // [row][2][KS][2] 16-B pieces (qimg_index; rows of the chunk)
               const double* __restrict__ qnc,  // [n_qblocks*32] |q'|^2 (0 for padding rows)
               float skip_scale,                // 2^-9 * max|r'| * (1 + slack): margin = skip_scale * |q'|
               int n_sentinel,                  // M - (neighbours searched + 1); E > 0: 0
               float* __restrict__ cand_val,    // [n_qblocks*32][2][M]
               int* __restrict__ cand_idx,
               // Query bucketing (bucket.hip.h): the kernel works on POSITIONS pos0 .. of the chunk; position p holds row
               // qperm[p] (null: the row itself).  A workgroup starts its sweep at the stage its middle row's cell names.
               int pos0, const int* __restrict__ qperm, const unsigned char* __restrict__ qcell,
               const int* __restrict__ cell_stage) {
    constexpr int TPS = tiles_per_stage2(KS);
    constexpr int TB = tile2_bytes(KS);
    constexpr int STAGE = TPS * TB;
    constexpr int NQB = kCoarse2Nqb;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    const int qb0 = (blockIdx.x * WAVES + wave) * NQB;
    const unsigned qwave = lds_addr_of(smem + 2 * STAGE + wave * queue2_bytes_per_wave() + lane * 8);
    // the row behind this lane's column of q-block qb: needed at the start and by every flush -- kept in LDS
    // (two registers more would spill), and the rotation of the sweep (workgroup-uniform)
    const unsigned qrow_lds = lds_addr_of(smem + 2 * STAGE + wave * queue2_bytes_per_wave() + NQB * kQueueCap * 512 + lane * 4);
    int st0 = 0;
    if (qperm) {
        const int mid = qperm[pos0 + (blockIdx.x * WAVES + WAVES / 2) * NQB * 32];
        st0 = __builtin_amdgcn_readfirstlane(cell_stage[qcell[mid]]);
    }
    // 16-B piece (part: 0 hi / 1 lo, K-step s, this lane's K half) of row r
    auto qfrag = [&](int r, int part, int s) {
        return __builtin_bit_cast(half8, qimg[qimg_index(r, part, KS, s, lane >> 5)]);
    };
    auto stage_of = [&](int st) {  // the st-th stage of this workgroup's sweep
        const int i = st0 + st;
        return i >= n_stages ? i - n_stages : i;
    };

    // Queries of this wave: hi fragments resident for the whole sweep; the lo fragments are only needed by
    // the flush, which fetches them again (16 VGPRs that the second accumulator set needs more).
    half8 bh[NQB][KS];
    float qnorm[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const int pos = pos0 + (qb0 + qb) * 32 + (lane & 31);
        const int r = qperm ? qperm[pos] : pos;
        asm volatile("ds_write_b32 %0, %1" ::"v"(qrow_lds + qb * 256), "v"(r) : "memory");
#pragma unroll
        for (int s = 0; s < KS; ++s) bh[qb][s] = qfrag(r, 0, s);
        qnorm[qb] = (float)sqrt(qnc[r]);
        // The flush needs the lo fragments of this lane's column again and again: they are copied once into
        // fragment order by position (this wave's own 1-KiB blocks, written and later read by the same lanes), so that
        // every later fetch is one coalesced 1-KiB load instead of 64 pieces out of 32 rows' lines.
    }

    float vals[NQB][M];
    int idxs[NQB][M];
#ifdef SKNNR_COARSE_COUNTERS
    unsigned ctr[16] = {};
#endif
#ifdef SKNNR_COARSE_TIMERS  // development aid: where one wave's cycles go (s_memtime stamps; the stamps cost time themselves)
    unsigned long long tm[8] = {};
    unsigned long long tk = 0, tk0 = 0;
    TICK();
    const unsigned long long t_begin = tk;
#endif
    float loose[NQB], margin[NQB];  // loose = threshold + margin: what a MAIN product is tested against
    int cnt[NQB];                   // (a lane that had to drop hits sets its loose to NaN: nothing is below NaN, the
                                    //  lane stops working and the query is marked at the end)
    auto reset_lists = [&](int qb) {
#pragma unroll
        for (int i = 0; i < M; ++i) {
            vals[qb][i] = (half == 0 && i < n_sentinel) ? -FLT_MAX : FLT_MAX;
            idxs[qb][i] = -1;
        }
    };
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        cnt[qb] = 0;
        margin[qb] = skip_scale * qnorm[qb] + 1e-30f;
        loose[qb] = FLT_MAX;
        reset_lists(qb);
    }

    // ---- operands of one unit from the staged tile: |r'|^2 lands in the accumulator registers ------------
    auto load_c0 = [&](const char* tb, floatx16& acc) {
        const floatx4* cp = (const floatx4*)(tb + KS * 1024 + half * 64);
        const floatx4 c_0 = cp[0], c_1 = cp[1], c_2 = cp[2], c_3 = cp[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = c_0[i];
            acc[4 + i] = c_1[i];
            acc[8 + i] = c_2[i];
            acc[12 + i] = c_3[i];
        }
    };
    auto load_hi = [&](const char* tb, half8 (&ah)[KS]) {
#pragma unroll
        for (int s = 0; s < KS; ++s) ah[s] = *(const half8*)(tb + s * 1024 + lane * 16);
    };
    auto issue_main = [&](const half8 (&ah)[KS], int qb, floatx16& acc) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[qb][s], acc, 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- wave flush (round 4): the entries of ALL lanes compacted, corrected 32 per trip, then inserted by their owners ----
    // The pair flush of round 3 corrected the entries of a query's two lanes two per trip: a flush is triggered by ONE lane holding
    // kQueueFlushAt entries while the wave holds ~22 on 64 lanes, so it takes 2-3 trips of two dependent L2 round trips each
    // with 8-11 lanes at work (8,800 wave-cycles per flush, profiles/r03_coarse_timers.txt).  Here every lane first moves its
    // entries to consecutive slots of the queue region (prefix sum of the counts; DS operations of a wave execute in order:
    // all reads are done before the first write), tagged with the query column; then lane pair p (lanes p, p + 32: the two
    // K halves) corrects entry 32 t + p whatever column it belongs to -- the column's hi / lo fragments come from the row's
    // own cache line of the query image, the row's fragments from the reference image, ALL of it one round of independent
    // loads -- and writes the corrected value back; finally every lane walks its own entries (now corrected) into its list.
    // The corrected value is the same expression as in that flush (main + (lo.hi + hi.lo over K half 0 + K half 1));
    // which list an entry goes to does not matter to the certificate.  One L2 round trip per 32 entries instead of four to six.
    auto flush_wave = [&](int qb) {
        TSTAMP(1);
        CTR(6, 1);
        const unsigned qbase = qwave - (unsigned)lane * 8u + (unsigned)qb * (kQueueCap * 512u);  // this q-block's region (2,560 B)
        const int c = cnt[qb];
        // exclusive prefix sum of the counts (0 .. kQueueCap < 8: three ballots)
        const unsigned long long b0 = __builtin_amdgcn_ballot_w64((c & 1) != 0), b1 = __builtin_amdgcn_ballot_w64((c & 2) != 0),
                                 b2 = __builtin_amdgcn_ballot_w64((c & 4) != 0);
        auto below = [&](unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };
        const int off = below(b0) + 2 * below(b1) + 4 * below(b2);
        const int total = __builtin_popcountll(b0) + 2 * __builtin_popcountll(b1) + 4 * __builtin_popcountll(b2);
        // (reads and their wait in ONE statement: nothing may touch the destination registers while the data is in flight)
        static_assert(kQueueCap == 5 || kQueueCap == 4, "four or five entries per lane are read here");
        unsigned long long own[5];
        if constexpr (kQueueCap == 5) {
            asm volatile("ds_read_b64 %0, %5\n\tds_read_b64 %1, %5 offset:512\n\tds_read_b64 %2, %5 offset:1024\n\t"
                         "ds_read_b64 %3, %5 offset:1536\n\tds_read_b64 %4, %5 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(own[0]), "=&v"(own[1]), "=&v"(own[2]), "=&v"(own[3]), "=&v"(own[4])
                         : "v"(qbase + (unsigned)lane * 8u)
                         : "memory");
        } else {
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\t"
                         "ds_read_b64 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(own[0]), "=&v"(own[1]), "=&v"(own[2]), "=&v"(own[3])
                         : "v"(qbase + (unsigned)lane * 8u)
                         : "memory");
        }
#pragma unroll
        for (int j = 0; j < kQueueCap; ++j) {
            // (pos < 2^26: the image is addressed with 32-bit offsets, use_coarse2) -- the column rides in the top bits
            const unsigned long long e = own[j] | ((unsigned long long)(unsigned)(lane & 31) << 58);
            if (j < c) asm volatile("ds_write_b64 %0, %1" ::"v"(qbase + (unsigned)(off + j) * 8u), "v"(e) : "memory");
        }
        const unsigned qrow_base = qrow_lds - (unsigned)lane * 4u + (unsigned)qb * 256u;
        for (int t0 = 0; t0 < total; t0 += 32) {
            CTR(7, 1);
            const int gi = t0 + (lane & 31);
            const bool on = gi < total;
            unsigned long long e;
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(e) : "v"(qbase + (unsigned)(on ? gi : 0) * 8u) : "memory");
            CTR(8, __builtin_popcountll(__builtin_amdgcn_ballot_w64(on)));
            const unsigned hi32 = (unsigned)(e >> 32);
            const int pos = (int)(hi32 & 0x03ffffffu), col = (int)(hi32 >> 26);
            int r;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(qrow_base + (unsigned)col * 4u) : "memory");
            const unsigned row = (unsigned)(pos & 31) * 16u + (unsigned)(32 * half) * 16u;
            const unsigned oh = (unsigned)(pos >> 5) * (unsigned)TB + row;
            const unsigned ol = (unsigned)(pos >> 5) * (unsigned)(KS * 1024) + row;
            half8 fl[KS], fh[KS], qh[KS], ql[KS];
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) {
                fl[s2] = *(const half8*)(rlo + ol + s2 * 1024);
                fh[s2] = *(const half8*)(rhi + oh + s2 * 1024);
                qh[s2] = qfrag(r, 0, s2);
                ql[s2] = qfrag(r, 1, s2);
            }
            float acc = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) {
                acc = dot8(fl[s2], qh[s2], acc);
                acc = dot8(fh[s2], ql[s2], acc);
            }
            const float other = __shfl_xor(acc, 32, 64);
            const float cv = __uint_as_float((unsigned)e) + (half ? other + acc : acc + other);  // K half 0 + K half 1
            if (on && half == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(qbase + (unsigned)gi * 8u), "v"(cv) : "memory");
        }
        // every lane's own entries, corrected: into its list
        for (int j = 0; __builtin_amdgcn_ballot_w64(j < c) != 0; ++j) {
            const bool on = j < c;
            unsigned long long e;
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(e) : "v"(qbase + (unsigned)(on ? off + j : 0) * 8u) : "memory");
            const float cv = __uint_as_float((unsigned)e);
            const int pos = (int)((unsigned)(e >> 32) & 0x03ffffffu);
            if constexpr (E == 0) {
                if (on && cv < vals[qb][M - 1]) list_insert<M>(vals[qb], idxs[qb], cv, pos);
            } else {
                // pooled lists (pair_union_rank): what this list lets go of is handed to the partner's at once, unless it is no
                // smaller than `loose` (which stays above the query's final threshold)
                float out_v = FLT_MAX;
                int out_i = -1;
                if (on) {
                    if (cv < vals[qb][M - 1]) {
                        out_v = vals[qb][M - 1];
                        out_i = idxs[qb][M - 1];
                        list_insert<M>(vals[qb], idxs[qb], cv, pos);
                    } else {
                        out_v = cv;
                        out_i = pos;
                    }
                }
                const bool offer = out_v < loose[qb];
                if (__builtin_amdgcn_ballot_w64(offer) != 0) {
                    const float in_v = __shfl_xor(offer ? out_v : FLT_MAX, 32, 64);
                    const int in_i = __shfl_xor(out_i, 32, 64);
                    if (in_v < vals[qb][M - 1]) list_insert<M>(vals[qb], idxs[qb], in_v, in_i);
                }
            }
        }
        cnt[qb] = 0;
        const float tight = pair_union_rank<M, E>(vals[qb]) + margin[qb];
        loose[qb] = loose[qb] != loose[qb] ? loose[qb] : min2f(loose[qb], tight);  // (NaN = poisoned: stays)
        TSTAMP(2);
    };

    // ---- one unit: skip test on the main products, visit = queue every value below threshold + margin ------
    // The visit looks at the registers of the groups that can hold a hit and appends.  The queue is flushed at the END of a
    // visit once a lane holds kQueueFlushAt entries, when the accumulator is dead and its registers are free
    // for the flush's gathers; every visit therefore starts with at least kQueueCap - kQueueFlushAt + 1 free
    // slots per lane.  A lane with more hits in ONE unit than it has room for (exact duplicates among the
    // references, a query whose margin is infinite) poisons its query: the row fails the certificate and is
    // answered by the exact scan.
    auto process = [&](floatx16& acc, const float (&g)[5], float m1, int tile_no, int qb) {
        CTR(0, 1);
        if (__builtin_amdgcn_ballot_w64(m1 < loose[qb]) == 0) return;
        CTR(1, 1);
        TSTAMP(0);  // sweep (everything since the last stamp that is not a visit / flush / barrier)
#ifdef SKNNR_V2_SWEEP_ONLY  // timing experiment: no visits (the margin keeps the sweep alive for the compiler)
        margin[qb] += 1e-30f;
        return;
#endif
        const unsigned qlane = qwave + qb * (kQueueCap * 512);
        // the row id is needed on a visit only: the empty asm pins its computation inside this branch (the
        // compiler would otherwise speculate it into the skip path, one VALU per unit)
        asm volatile("" : "+s"(tile_no));
        const int id_base = tile_no * 32 + 4 * half;
        int want = cnt[qb];  // entries this lane would hold if the queue were unbounded
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            if (__builtin_amdgcn_ballot_w64(g[k] < loose[qb]) == 0) continue;
#pragma unroll
            for (int r = 3 * k; r < (k == 4 ? 16 : 3 * k + 3); ++r) {
                const bool hit = acc[r] < loose[qb];
                if (__builtin_amdgcn_ballot_w64(hit) == 0) continue;
                CTR(3, 1);
                CTR(4, __builtin_popcountll(__builtin_amdgcn_ballot_w64(hit)));
                want += hit ? 1 : 0;
                if (hit && cnt[qb] < kQueueCap) {
                    queue_store(qlane + cnt[qb] * 512, acc[r], id_base + acc_row(r, 0));
                    cnt[qb] += 1;
                }
            }
        }
        if (want > kQueueCap) loose[qb] = __builtin_nanf("");
        if (__builtin_amdgcn_ballot_w64(cnt[qb] >= kQueueFlushAt) != 0) flush_wave(qb);
        else TSTAMP(1);  // visit scan without a flush
    };

    // ---- seeding: a valid starting threshold from the first kSeedTiles tiles ------------------------------
    const int n_tiles = n_stages * TPS;
#ifdef SKNNR_V2_NO_SEED
    if (false) {
#else
    if (n_tiles >= 2 * kSeedTiles) {
#endif
        const int SEED_STAGES = seed_tiles_for(n_tiles, TPS) / TPS;
        stage_copy(rhi + (size_t)stage_of(0) * STAGE, smem, STAGE, wave, lane, WAVES);
        __syncthreads();
        for (int st = 0; st < SEED_STAGES; ++st) {
            const char* cur = smem + (st & 1) * STAGE;
            if (st + 1 < SEED_STAGES)
                stage_copy(rhi + (size_t)stage_of(st + 1) * STAGE, smem + ((st + 1) & 1) * STAGE, STAGE, wave, lane, WAVES);
#pragma unroll 1
            for (int t = 0; t < TPS; ++t) {
                const char* tb = cur + t * TB;
                half8 ah[KS];
                load_hi(tb, ah);
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    floatx16 acc;
                    load_c0(tb, acc);
                    issue_main(ah, qb, acc);
                    const float t0 = first_read(acc[0]);
                    const float a0 = min3f(t0, acc[1], acc[2], t0), a1 = min3f(acc[3], acc[4], acc[5], t0);
                    const float a2 = min3f(acc[6], acc[7], acc[8], t0), a3 = min3f(acc[9], acc[10], acc[11], t0);
                    const float a4 = min2f(min3f(acc[12], acc[13], acc[14], t0), acc[15], t0);
                    const float m1 = min3f(min3f(a0, a1, a2, t0), a3, a4, t0);
                    if (__builtin_amdgcn_ballot_w64(m1 < vals[qb][M - 1]) != 0) {
                        if (m1 < vals[qb][M - 1]) list_insert<M>(vals[qb], idxs[qb], m1, 0);
                    }
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            // J distinct rows have main values <= the J-th smallest seed; their corrected values are at most
            // `margin` larger: a valid bound on the J-th smallest corrected value of the whole sweep
            const float seed = pair_union_rank<M, E>(vals[qb]);
            loose[qb] = seed < FLT_MAX ? seed + 2.0f * margin[qb] : FLT_MAX;
            reset_lists(qb);
        }
        TSTAMP(4);  // seeding pass
    }

    // ---- the sweep ---------------------------------------------------------------------------------------------
    stage_copy(rhi + (size_t)stage_of(0) * STAGE, smem, STAGE, wave, lane, WAVES);
    __syncthreads();
    for (int st = 0; st < n_stages; ++st) {
        const char* cur = smem + (st & 1) * STAGE;
        if (st + 1 < n_stages)
            stage_copy(rhi + (size_t)stage_of(st + 1) * STAGE, smem + ((st + 1) & 1) * STAGE, STAGE, wave, lane, WAVES);
        const int tile0 = stage_of(st) * TPS;
        float g[5], m1;
#pragma unroll SKNNR_V2_UNROLL
        for (int t = 0; t < TPS; ++t) {
            const char* tb = cur + t * TB;
            const int tile_no = tile0 + t;
            floatx16 acc0, acc1;
            half8 ah[KS];
            load_hi(tb, ah);
            load_c0(tb, acc1);
            tile_issue_and_test<KS>(acc0, acc1, ah, bh[0], bh[1], g, m1);  // both units go out, unit 0 is tested
            process(acc0, g, m1, tile_no, 0);
            step_test_only(acc1, g, m1);
            process(acc1, g, m1, tile_no, 1);
        }
        TSTAMP(0);
#ifdef SKNNR_V2_NO_BARRIER  // timing experiment only (results are wrong: a wave may read a stage that is being replaced)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        __syncthreads();  // next stage landed (vmcnt(0)) and everyone is done with `cur`
#endif
        TSTAMP(3);  // stage barrier
    }

#ifdef SKNNR_COARSE_COUNTERS
    if (lane == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&coarse_counters[i], (unsigned long long)ctr[i]);
#endif
    TSTAMP(0);
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        flush_wave(qb);
        // (the lists are filed under the POSITION of the chunk -- the row itself unless the call is bucketed: consecutive lanes
        //  write consecutive lists, and the finaliser reads them without waiting for the position -> row table, round 4)
        const size_t q = (size_t)(pos0 + (qb0 + qb) * 32 + (lane & 31));
        const size_t base = (q * 2 + half) * M;
        // a poisoned query (dropped hits) must fail the certificate: a NaN bound never certifies
        const int mine = loose[qb] != loose[qb] ? 1 : 0;
        const bool bad = (mine | __shfl_xor(mine, 32, 64)) != 0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            cand_val[base + i] = bad ? __builtin_nanf("") : vals[qb][i];
            cand_idx[base + i] = idxs[qb][i];
        }
    }
#ifdef SKNNR_COARSE_TIMERS
    TICK();
    tm[7] = tk - t_begin;
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&coarse_timers[i], tm[i]);
#endif
}

}  // namespace sknnr
