// coarse.hip.h -- the MFMA pre-filter: for every query, the M smallest approximate
// ranking values  v(q, r) ~= |r'|^2 - 2 q'.r'   (primes: centred and scaled features)
// seen by each of the two lanes that own the query, over ALL reference rows.
//
// Replaces the hot double loop of the reference's engine
//   dgemm  M = -2 Xc Yc^T            SKL/metrics/_pairwise_distances_reduction/_middle_term_computer.pyx.tp:440
//   d2 = |x|^2 + M + |y|^2, heap_push SKL/metrics/_pairwise_distances_reduction/_argkmin.pyx.tp:492-510
// as a candidate generator; the float64 finaliser (exact.hip.h) re-scores the
// candidates with the reference's exact expression and certifies the result.
//
// MI355X mapping
//   * contraction on the f16 matrix pipe (v_mfma_f32_32x32x16_f16, 16x the f32 MFMA rate)
//     with every operand split x = hi + lo (two f16 each) and three products
//     hi.hi + lo.hi + hi.lo accumulated in f32  ->  ~2^-22 relative error, i.e. f32-class
//     accuracy at 16/3 of the f32-MFMA throughput; the two correction products are only
//     issued for tiles in which some lane's hi.hi value is within their bound of its threshold;
//   * references are the A operand (rows), queries the B operand (columns): the 32x32
//     accumulator then holds ONE query per lane (col = lane & 31) and 16 references in the
//     lane's registers, so the running top-M is lane-local (no cross-lane traffic in the
//     sweep); lanes l and l+32 share a query, keep one list each and test hits against one
//     threshold, the (k+1)-th best value of their two lists together;
//   * |r'|^2 enters as the C operand of the first MFMA (exact f32, no VALU add);
//   * reference tiles are stored in HBM in MFMA-fragment order and copied to LDS by
//     LDS-DMA (global_load_lds_dwordx4), double buffered; every wave of the workgroup (16, 12
//     or 8 waves by feature width and list length) reads the same staged tile, lane-linear
//     ds_read_b128 (conflict-free);
//   * queries live in registers for the whole sweep (64 or 32 per wave).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

namespace sknnr {

#if !defined(SKNNR_EXPERIMENTS) && defined(SKNNR_ABLATE_NO_CORR)
#error "SKNNR_ABLATE_NO_CORR is a timing experiment with wrong results: add -DSKNNR_EXPERIMENTS"
#endif

// query-row padding of a chunk: a multiple of the rows per workgroup of every pre-filter geometry (2048, 1536, 1024, 768, 512,
// 384, 256)
constexpr long kRowQuantum = 6144;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// One 32-reference tile record in HBM/LDS: [part hi|lo][K-step][lane] 16 B, then the
// C-operand init  [lane half][16] f32.
__host__ __device__ constexpr int tile_frag_bytes(int ks) { return 2 * ks * 1024; }
__host__ __device__ constexpr int tile_bytes(int ks) { return 2 * ks * 1024 + 128; }
// 32-reference tiles per LDS stage.  One 16-wave workgroup per CU (4 waves per SIMD) shares the
// stage: two stage buffers + the waves' candidate queues must fit the 160 KiB of LDS.
__host__ __device__ constexpr int tiles_per_stage(int ks) { return ks <= 2 ? 8 : (ks <= 4 ? 4 : 2); }
// Query image: every row's B-operand fragments side by side -- [row][part: hi | lo][K-step][K half] 16-byte pieces,
// 64 KS bytes per row (round 3; rounds 1-2 kept whole 1-KiB MFMA fragments of 32 rows together).  A lane (column
// `row`, K half `kh`) picks its pieces out of the row's own cache lines, so that bucketed calls, whose q-blocks are
// made of scattered rows, fetch every line they touch completely (the fragment-major layout cost them 4x the bytes).
__host__ __device__ constexpr size_t qimg_index(long row, int part, int ks, int s, int kh) {
    return ((size_t)(row * 2 + part) * ks + s) * 2 + kh;
}
// Row of the 32x32 accumulator held in register r of a lane in half h (guide section 3).
__host__ __device__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Launch geometry by feature width (KS) and list length (M):
//   light  (KS <= 2, M <= 8): 16 waves -- one 1024-thread workgroup per CU, 4 waves per SIMD at <= 128 VGPR;
//   medium (KS 3..6, M <= 8; KS 7, M 6; KS <= 5, M 16): 12 waves -- 3 per SIMD at <= 170 VGPR
//          (lo fragments fetched on demand);
//   heavy  (the rest): 8 waves -- 2 per SIMD, <= 256 VGPR.
//   Two 32-query blocks per wave while the registers allow it (measured for M = 16, KS = 2: 12 waves
//   x 2 q-blocks is 2.2x faster than 16 waves x 1 q-block).
__host__ __device__ constexpr bool coarse_is_light(int ks, int m) { return ks <= 2 && m <= 8; }
__host__ __device__ constexpr bool coarse_is_medium(int ks, int m) {
    return (ks >= 3 && ks <= 6 && m <= 8) || (ks == 7 && m <= 6) || (ks <= 5 && m == 16);
}
__host__ __device__ constexpr int coarse_waves(int ks, int m) {
    return coarse_is_light(ks, m) ? 16 : (coarse_is_medium(ks, m) ? 12 : 8);
}
__host__ __device__ constexpr int coarse_wps(int ks, int m) {
    return coarse_is_light(ks, m) ? 4 : (coarse_is_medium(ks, m) ? 3 : 2);
}
__host__ __device__ constexpr bool coarse_lo_on_demand(int ks, int m) { return coarse_is_medium(ks, m); }
#ifndef SKNNR_M2_NQB
#define SKNNR_M2_NQB 4  // one neighbour (lists of 2): four q-blocks per wave up to 16 features, three up to 32
#endif
#ifndef SKNNR_M2_KS2_NQB
#define SKNNR_M2_KS2_NQB 3
#endif
#ifndef SKNNR_KS1_NQB
#define SKNNR_KS1_NQB 3  // up to 16 features: three q-blocks per wave fit 128 VGPR (d=16, 50k refs: 11 % faster than two)
#endif
__host__ __device__ constexpr int coarse_nqb(int ks, int m) {
    return (m == 2 && ks == 1) ? SKNNR_M2_NQB : (m == 2 && ks == 2) ? SKNNR_M2_KS2_NQB : (ks == 1 && m <= 6) ? SKNNR_KS1_NQB : (((ks <= 4 && m <= 8) || (ks <= 2 && m == 16)) ? 2 : 1);
}

// v_min3_f32 / v_min_f32 as raw instructions: the compiler would put a canonicalising
// v_max in front of every fminf operand that comes out of an MFMA (16 extra VALU per tile).
// HAZARD (guide section 5.7 item 2): hipcc pads the MFMA -> VALU-read wait states only for
// instructions it can see, not for the inside of an asm statement.  Every asm below that
// reads accumulator registers therefore takes `dep`, a value produced by a compiler-visible
// VALU instruction (first_read) that read the same MFMA result: the data dependence keeps the
// asm behind that instruction, and the compiler pads that instruction correctly.
__device__ __forceinline__ float first_read(float a) {
    return __builtin_canonicalizef(a);  // one compiler-visible v_max_f32 a, a: value unchanged
}
__device__ __forceinline__ float min3f(float a, float b, float c, float dep) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c), "v"(dep));
    return r;
}
__device__ __forceinline__ float min2f(float a, float b, float dep) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b), "v"(dep));
    return r;
}
// no accumulator operands: plain register values
__device__ __forceinline__ float min2f(float a, float b) { return min2f(a, b, a); }

#ifndef SKNNR_SHIFT_INSERT_FROM
#define SKNNR_SHIFT_INSERT_FROM 32
#endif
// Sorted (ascending) insertion of (v, id) into a lane-local list whose last entry is
// known to be > v.
template <int M>
__device__ __forceinline__ void list_insert(float (&vals)[M], int (&idxs)[M], float v, int id) {
    if constexpr (M >= SKNNR_SHIFT_INSERT_FROM) {
        // Long lists: every position decides for itself (keep / take v / take the left neighbour), high
        // to low, so that the old left neighbour is still there.  Same result as the bubble below
        // (v settles behind equal values); hipcc turns the 31-stage bubble of M = 32 into
        // index-select chains (35k v_cndmask, ~100k cycles per flush).
#pragma unroll
        for (int i = M - 1; i >= 0; --i) {
            const bool here = v < vals[i];                                    // position i changes
            const bool lower_left = i > 0 ? (v < vals[i - 1]) : false;        // v belongs further left
            const float nv = lower_left ? vals[i > 0 ? i - 1 : 0] : v;
            const int ni = lower_left ? idxs[i > 0 ? i - 1 : 0] : id;
            vals[i] = here ? nv : vals[i];
            idxs[i] = here ? ni : idxs[i];
        }
    } else {
        vals[M - 1] = v;
        idxs[M - 1] = id;
#pragma unroll
        for (int i = M - 1; i > 0; --i) {
            const bool sw = vals[i] < vals[i - 1];
            const float lo = sw ? vals[i] : vals[i - 1];
            const float hi = sw ? vals[i - 1] : vals[i];
            const int ilo = sw ? idxs[i] : idxs[i - 1];
            const int ihi = sw ? idxs[i - 1] : idxs[i];
            vals[i - 1] = lo;
            vals[i] = hi;
            idxs[i - 1] = ilo;
            idxs[i] = ihi;
        }
    }
}

// Candidate queue of one lane: queue_cap(M) (value, index) pairs in LDS, laid out
// [entry][lane] so that the batched flush reads conflict-free.  A hit is appended with a
// handful of instructions; the 44-instruction sorted insertion runs later, for all lanes of
// the wave at once (the flush), instead of once per hit with one or two lanes active.
// Development aid (-DSKNNR_COARSE_COUNTERS): event counts of the sweep, summed over waves into
// coarse_counters[]; read back by sknnr_get_stats and printed to stderr.  Off in the product build.
#ifdef SKNNR_COARSE_TIMERS
static __device__ unsigned long long coarse_timers[8];  // (one copy per kernel translation unit)
#define TICK()                                  \
    do {                                        \
        asm volatile("" ::: "memory");          \
        tk = __builtin_readcyclecounter();      \
        asm volatile("" ::: "memory");          \
    } while (0)
#define TSTAMP(i)            \
    do {                     \
        tk0 = tk;            \
        TICK();              \
        tm[i] += tk - tk0;   \
    } while (0)
#else
#define TICK() ((void)0)
#define TSTAMP(i) ((void)0)
#endif
#ifdef SKNNR_COARSE_COUNTERS
static __device__ unsigned long long coarse_counters[16];
#define CTR_ARG , unsigned (&ctr)[16]
#define CTR_PASS , ctr
#define CTR(i, n) ctr[i] += (unsigned)(n)
#else
#define CTR_ARG
#define CTR_PASS
#define CTR(i, n) ((void)0)
#endif


// Wave priorities: a wave that handles a visit (corrections, hit scan) runs ahead of sweeping waves;
// measured 1.3 % faster than priority 1 for the MFMAs only.
#ifndef SKNNR_PRIO_CORR
#define SKNNR_PRIO_CORR 2
#define SKNNR_PRIO_SCAN 2
#endif
#ifndef SKNNR_PRIO_MAIN
#define SKNNR_PRIO_MAIN 1  // while the main products are issued
#define SKNNR_PRIO_TEST 0  // skip test (and the rest of the sweep)
#endif

// 4 deep, flushed at 3; the 2-entry lists (one neighbour, four q-blocks per wave) make do with 2
__host__ __device__ constexpr int queue_cap(int m) { return m == 2 ? 2 : 4; }
__host__ __device__ constexpr int queue_flush_at(int m) { return m == 2 ? 2 : 3; }
__host__ __device__ constexpr int queue_bytes_per_wave(int nqb, int m) { return nqb * queue_cap(m) * 64 * 8; }

// The three-product split contraction of one 32-ref x 32-query tile, in two stages:
//   main    = |r'|^2 + hi.hi                     (KS MFMAs)
//   correct = main + lo.hi + hi.lo                (2 KS MFMAs)
// |correct - main| <= 2^-9 |q'| |r'| (each lo is at most 2^-11 of its operand and the
// reference operand carries the factor -2), so a tile whose `main` values all exceed the
// lane thresholds by that margin cannot contain a hit and skips the correction MFMAs.
template <int KS>
__device__ __forceinline__ floatx16 contract_main(const half8 (&ah)[KS], const half8 (&bh)[KS], floatx16 c0) {
    floatx16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh[0], c0, 0, 0, 0);
#pragma unroll
    for (int s = 1; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s], acc, 0, 0, 0);
    return acc;
}
template <int KS>
__device__ __forceinline__ floatx16 contract_correct(const half8 (&ah)[KS], const half8 (&al)[KS],
                                                     const half8 (&bh)[KS], const half8 (&bl)[KS],
                                                     floatx16 acc) {
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s], acc, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s], acc, 0, 0, 0);
    return acc;
}
template <int KS>
__device__ __forceinline__ floatx16 split_contract(const half8 (&ah)[KS], const half8 (&al)[KS],
                                                   const half8 (&bh)[KS], const half8 (&bl)[KS],
                                                   floatx16 c0) {
    return contract_correct<KS>(ah, al, bh, bl, contract_main<KS>(ah, bh, c0));
}

// Copy `bytes` (multiple of 16) from global to LDS, lane-linear, by LDS-DMA.  Chunks of
// 1 KiB are dealt round-robin to the workgroup's waves.
__device__ __forceinline__ void stage_copy(const char* __restrict__ gsrc, char* lds_dst, int bytes,
                                           int wave, int lane, int n_waves) {
    const int n_chunks = (bytes + 1023) >> 10;
    for (int c = wave; c < n_chunks; c += n_waves) {
        const int off = (c << 10) + (lane << 4);
        if (off < bytes) {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(gsrc + off),
                (__attribute__((address_space(3))) void*)(lds_dst + (c << 10)), 16, 0, 0);
        }
    }
}

// The queue is accessed with inline-asm DS instructions: hipcc orders a compiler-visible
// ds_write behind the stage's in-flight LDS-DMA (s_waitcnt vmcnt(0) before every append),
// which would drain the prefetch on each hit.  The queue region is disjoint from the stage
// buffers and DS operations of one wave execute in order, so no wait is needed on the write;
// the read carries its own lgkmcnt(0) (guide section 5.7, form (i)).
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ void queue_store(unsigned addr, float v, int id) {
    const unsigned long long pr = ((unsigned long long)(unsigned)id << 32) | (unsigned long long)__float_as_uint(v);
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(pr) : "memory");
}
__device__ __forceinline__ unsigned long long queue_load(unsigned addr) {
    unsigned long long pr;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(pr) : "v"(addr) : "memory");
    return pr;
}

// One hit: append to the lane's queue, or -- queue full -- insert directly.
template <int M>
__device__ __forceinline__ void take_hit(float v, int id, float (&vals)[M], int (&idxs)[M], float& thr,
                                         int& cnt, unsigned qlane CTR_ARG) {
    const bool hit = v < thr;
    if (__builtin_amdgcn_ballot_w64(hit) == 0) return;
    CTR(3, 1);
    CTR(4, __builtin_popcountll(__builtin_amdgcn_ballot_w64(hit)));
    const bool room = cnt < queue_cap(M);
    if (hit && room) {
        queue_store(qlane + cnt * 512, v, id);
        cnt += 1;
    }
    if (__builtin_amdgcn_ballot_w64(hit && !room) != 0) {
        CTR(5, 1);
        if (hit && !room) {
            list_insert<M>(vals, idxs, v, id);
            thr = min2f(thr, vals[M - 1]);
        }
    }
}

// The M-th smallest entry of the union of the two sorted M-lists owned by lanes l and l+32
// (the two lanes that share a query): max_i min(a_i, b_{M-1-i}).  Both lanes get the same value.
template <int M>
__device__ __forceinline__ float pair_union_rank_m(const float (&vals)[M]) {
    float u = fminf(vals[0], __shfl_xor(vals[M - 1], 32, 64));
#pragma unroll
    for (int i = 1; i < M; ++i) u = fmaxf(u, fminf(vals[i], __shfl_xor(vals[M - 1 - i], 32, 64)));
    return u;
}

// Batched insertion of every lane's queued candidates, then the threshold of the two lanes
// that own the same query (l and l+32) is tightened to the M-th smallest entry of their two
// lists together.  The lower lane's list starts with M - J sentinels (-FLT_MAX), so that entry
// is the J-th best value the query has seen (J = neighbours searched + 1): everything either
// lane rejects from now on is >= that value, which is the bound the finaliser's certificate
// recomputes from the stored lists.
template <int M>
__device__ __forceinline__ void flush_queue(float (&vals)[M], int (&idxs)[M], float& thr, int& cnt,
                                            unsigned qlane CTR_ARG) {
    for (int i = 0; __builtin_amdgcn_ballot_w64(i < cnt) != 0; ++i) {
        CTR(7, 1);
        CTR(8, __builtin_popcountll(__builtin_amdgcn_ballot_w64(i < cnt)));
        if (i < cnt) {
            const unsigned long long e = queue_load(qlane + i * 512);
            const float ev = __uint_as_float((unsigned)e);
            CTR(9, __builtin_amdgcn_ballot_w64(ev < vals[M - 1]) != 0);
            CTR(10, __builtin_popcountll(__builtin_amdgcn_ballot_w64(ev < vals[M - 1])));
            if (ev < vals[M - 1]) list_insert<M>(vals, idxs, ev, (int)(e >> 32));
        }
    }
    cnt = 0;
    thr = pair_union_rank_m<M>(vals);
}

// KS : 16-wide K-steps per split part (padded feature count / 16)
// M  : list length per lane (2, 6, 8, 16 or 32: up to 1, 5, 7, 15 or 31 neighbours searched)
template <int KS, int M>
__global__ void __launch_bounds__(coarse_waves(KS, M) * 64, coarse_wps(KS, M))
coarse_kernel(const char* __restrict__ rimg,   // n_stages * TPS tile records
              int n_stages,
              const uint4* __restrict__ qimg,  // [row][2][KS][2] 16-B pieces (qimg_index)
              const double* __restrict__ qnc,  // [n_qblocks*32] |q'|^2 (0 for padding rows)
              float skip_scale,                // 2^-9 * max|r'| * (1 + slack): margin = skip_scale * |q'|
              int n_sentinel,                  // M - (neighbours searched + 1): leading sentinels of the lower lane's list
              float* __restrict__ cand_val,    // [n_qblocks*32][2][M]
              int* __restrict__ cand_idx) {
    constexpr int TPS = tiles_per_stage(KS);
    constexpr int TB = tile_bytes(KS);
    constexpr int STAGE = TPS * TB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    constexpr int WAVES = coarse_waves(KS, M);
    constexpr int NQB = coarse_nqb(KS, M);
    constexpr bool LO_ON_DEMAND = coarse_lo_on_demand(KS, M);
    const int qb0 = (blockIdx.x * WAVES + wave) * NQB;
    const unsigned qwave = lds_addr_of(smem + 2 * STAGE + wave * queue_bytes_per_wave(NQB, M) + lane * 8);

    // Queries of this wave: B fragments, resident for the whole sweep.
    half8 bh[NQB][KS], bl[NQB][KS];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const long row = (long)(qb0 + qb) * 32 + (lane & 31);
            const uint4 uh = qimg[qimg_index(row, 0, KS, s, lane >> 5)];
            const uint4 ul = qimg[qimg_index(row, 1, KS, s, lane >> 5)];
            bh[qb][s] = __builtin_bit_cast(half8, uh);
            bl[qb][s] = __builtin_bit_cast(half8, ul);
        }
    }

    float vals[NQB][M];
    int idxs[NQB][M];
    float thr[NQB];
    float margin[NQB];
    int cnt[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        thr[qb] = FLT_MAX;
        cnt[qb] = 0;
        margin[qb] = skip_scale * (float)sqrt(qnc[(size_t)(qb0 + qb) * 32 + (lane & 31)]) + 1e-30f;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            vals[qb][i] = (half == 0 && i < n_sentinel) ? -FLT_MAX : FLT_MAX;
            idxs[qb][i] = -1;
        }
    }

#ifdef SKNNR_COARSE_COUNTERS
    unsigned ctr[16] = {};
#endif
#ifdef SKNNR_COARSE_TIMERS  // development aid: where one wave's cycles go (s_memtime stamps)
    unsigned long long tm[8] = {};
    unsigned long long tk = 0, tk0 = 0;
    TICK();
    const unsigned long long t_begin = tk;
#endif
    stage_copy(rimg, smem, STAGE, wave, lane, WAVES);
    __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and publishes stage 0

    for (int st = 0; st < n_stages; ++st) {
        char* cur = smem + (st & 1) * STAGE;
        if (st + 1 < n_stages)
            stage_copy(rimg + (size_t)(st + 1) * STAGE, smem + ((st + 1) & 1) * STAGE, STAGE, wave, lane, WAVES);

        // ---- one tile: (1) operands from LDS -----------------------------------------------------------
        auto tile_load = [&](const char* tb, half8 (&ah)[KS], half8 (&al)[KS], floatx16& c0, bool keep_lo) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                ah[s] = *(const half8*)(tb + (0 * KS + s) * 1024 + lane * 16);
                if (keep_lo) al[s] = *(const half8*)(tb + (1 * KS + s) * 1024 + lane * 16);
            }
            const floatx4* cp = (const floatx4*)(tb + tile_frag_bytes(KS) + half * 64);
            const floatx4 c_0 = cp[0], c_1 = cp[1], c_2 = cp[2], c_3 = cp[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c0[i] = c_0[i];
                c0[4 + i] = c_1[i];
                c0[8 + i] = c_2[i];
                c0[12 + i] = c_3[i];
            }
        };
        // ---- (2) main products, skip tests and visits of the tile, q-block by q-block ------------------
        // have_lo: the tile's lo fragments are in `al`; otherwise the first visit fetches them.
        auto tile_process = [&](const char* tb, int tile_no, const floatx16& c0, half8 (&ah)[KS], half8 (&al)[KS],
                                bool have_lo) {
            const int id_base = tile_no * 32 + 4 * half;
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                __builtin_amdgcn_s_setprio(SKNNR_PRIO_MAIN);
                floatx16 acc = contract_main<KS>(ah, bh[qb], c0);
                __builtin_amdgcn_s_setprio(SKNNR_PRIO_TEST);
                // Skip test: minima of five groups of the 16 main-product values ({0-2}, {3-5}, {6-8},
                // {9-11}, {12-15}).  A value can only become a hit after correction if its main value is
                // below thr + margin, so the same five minima later tell which groups to look at.
                const float t0 = first_read(acc[0]);
                float g[5];
                g[0] = min3f(t0, acc[1], acc[2], t0);
                g[1] = min3f(acc[3], acc[4], acc[5], t0);
                g[2] = min3f(acc[6], acc[7], acc[8], t0);
                g[3] = min3f(acc[9], acc[10], acc[11], t0);
                g[4] = min2f(min3f(acc[12], acc[13], acc[14], t0), acc[15], t0);
                const float m1 = min3f(min3f(g[0], g[1], g[2], t0), g[3], g[4], t0);
                const float loose = thr[qb] + margin[qb];
                CTR(0, 1);
                if (__builtin_amdgcn_ballot_w64(m1 < loose) == 0) {
                    TSTAMP(1);  // skip test, no visit
                    continue;
                }
                TSTAMP(2);  // skip test, visit follows
                CTR(1, 1);
                CTR(11, __builtin_popcountll(__builtin_amdgcn_ballot_w64(m1 < loose)));
                CTR(12, __builtin_amdgcn_ballot_w64(m1 < thr[qb]) != 0);
                if (!have_lo) {
#pragma unroll
                    for (int s = 0; s < KS; ++s) al[s] = *(const half8*)(tb + (1 * KS + s) * 1024 + lane * 16);
                    have_lo = true;
                }
                __builtin_amdgcn_s_setprio(SKNNR_PRIO_CORR);
#ifndef SKNNR_ABLATE_NO_CORR  // timing experiment only: results are wrong without the corrections
                acc = contract_correct<KS>(ah, al, bh[qb], bl[qb], acc);
#endif
                __builtin_amdgcn_s_setprio(SKNNR_PRIO_SCAN);
                const unsigned qlane = qwave + qb * (queue_cap(M) * 512);
                // (take_hit's compare is compiler-visible: it is the hazard-padded first reader of the
                // corrected accumulator)
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    if (__builtin_amdgcn_ballot_w64(g[k] < loose) != 0) {
                        CTR(2, 1);
#pragma unroll
                        for (int r = 3 * k; r < (k == 4 ? 16 : 3 * k + 3); ++r)
                            take_hit<M>(acc[r], id_base + acc_row(r, 0), vals[qb], idxs[qb], thr[qb], cnt[qb], qlane CTR_PASS);
                    }
                }
                __builtin_amdgcn_s_setprio(SKNNR_PRIO_TEST);
                TSTAMP(3);  // corrections + hit scan
                if (__builtin_amdgcn_ballot_w64(cnt[qb] >= queue_flush_at(M)) != 0) {
                    CTR(6, 1);
                    flush_queue<M>(vals[qb], idxs[qb], thr[qb], cnt[qb], qlane CTR_PASS);
                    TSTAMP(4);  // flush
                }
            }
        };

#pragma unroll 1
        for (int t = 0; t < TPS; ++t) {
            const char* tb = cur + t * TB;
            half8 ah[KS], al[KS];
            floatx16 c0;
            tile_load(tb, ah, al, c0, !LO_ON_DEMAND);
            TSTAMP(0);  // tile operands requested
            tile_process(tb, st * TPS + t, c0, ah, al, !LO_ON_DEMAND);
        }
        TSTAMP(5);  // loop overhead
        __syncthreads();  // next stage landed (vmcnt(0)) and everyone is done with `cur`
        TSTAMP(6);  // barrier (+ stage issue)
    }

#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        flush_queue<M>(vals[qb], idxs[qb], thr[qb], cnt[qb], qwave + qb * (queue_cap(M) * 512) CTR_PASS);
        const size_t q = (size_t)(qb0 + qb) * 32 + (lane & 31);
        const size_t base = (q * 2 + half) * M;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            cand_val[base + i] = vals[qb][i];
            cand_idx[base + i] = idxs[qb][i];
        }
    }
#ifdef SKNNR_COARSE_COUNTERS
    if (lane == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&coarse_counters[i], (unsigned long long)ctr[i]);
#endif
#ifdef SKNNR_COARSE_TIMERS
    TICK();
    tm[7] = tk - t_begin;
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&coarse_timers[i], tm[i]);
#endif
}

// Diagnostic twin of the production kernel: the same split contraction, every value
// written out.  One wave per (32-ref tile, 32-query block).
template <int KS>
__global__ void __launch_bounds__(64)
coarse_matrix_kernel(const char* __restrict__ rimg, const int* __restrict__ perm, const uint4* __restrict__ qimg,
                     int n_ref, int nq, float* __restrict__ out) {
    const int lane = threadIdx.x;
    const int tile = blockIdx.x, qblk = blockIdx.y;
    const char* tb = rimg + (size_t)tile * tile_bytes(KS);
    half8 ah[KS], al[KS], bh[KS], bl[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        ah[s] = *(const half8*)(tb + (0 * KS + s) * 1024 + lane * 16);
        al[s] = *(const half8*)(tb + (1 * KS + s) * 1024 + lane * 16);
        bh[s] = __builtin_bit_cast(half8, qimg[qimg_index((long)qblk * 32 + (lane & 31), 0, KS, s, lane >> 5)]);
        bl[s] = __builtin_bit_cast(half8, qimg[qimg_index((long)qblk * 32 + (lane & 31), 1, KS, s, lane >> 5)]);
    }
    const int half = lane >> 5;
    floatx16 c0;
    const float* cp = (const float*)(tb + tile_frag_bytes(KS) + half * 64);
#pragma unroll
    for (int i = 0; i < 16; ++i) c0[i] = cp[i];
    const floatx16 acc = split_contract<KS>(ah, al, bh, bl, c0);
    const int q = qblk * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int pos = tile * 32 + acc_row(r, half);
        if (q < nq && pos < n_ref) out[(size_t)q * n_ref + perm[pos]] = acc[r];
    }
}

}  // namespace sknnr
