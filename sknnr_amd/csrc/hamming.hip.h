// hamming.hip.h -- integer pre-filter + exact re-score for the weighted-Hamming search of RFNN / GBNN (round 3).
//
// The reference reaches scipy's cdist(metric="hamming", w=w) through scikit-learn's brute path
// (REF src/sknnr/_weighted_trees.py:53-59, :139-140): d(q, r) = sum_t w_t [q_t != r_t] / sum_t w_t on float64 node
// ids, sums in tree order.  exact_scan_kernel<2> replays that arithmetic pair by pair: one float64 compare and one
// float64 add per (pair, tree) -- 6.3e12 compares/s in round 2.  Node ids are small integers, so here
//
//   1. hamming_pack_kernel      query rows (float64 ids) -> 16-bit ids, two trees per dword, [tree pair][query]
//   2. hamming_coarse_kernel    D^(q, r) = sum_t w^_t [q_t != r_t] with 16-bit integer weights w^_t = round(w_t S):
//                               per TWO trees one v_xor_b32 (ids differ?), one v_pk_min_u16 (-> 0/1 per half) and one
//                               v_dot2_u32_u16 (weighted accumulate): 1.5 VALU lane-ops per compare, exact in integers.
//                               A lane owns a reference row, a workgroup 16 queries (their ids and the weights are
//                               scalar operands), the reference ids stream from L2 once per 16 queries.  Every row
//                               whose D^ is within `band` of the running kk-th smallest D^ of its query becomes a
//                               candidate (appended in index order).
//   3. hamming_rescore_kernel   the candidates' distances in the reference's own float64 arithmetic, selection by
//                               (distance, index) exactly as exact_scan_kernel<2> does it over all rows, post-steps.
//
// Exactness: |D^ - S sum_t w_t [.]| <= T / 2 (rounding of the weights), the float64 evaluation of the true sum is
// off by less than 2^-40 of a weight unit; so D^_a + band < D^_b with band = T + 2 implies d_a < d_b in the reference's
// arithmetic, and a row that is NOT within band of the kk-th smallest D^ has kk rows strictly closer than itself:
// the kk nearest by (distance, index) are always among the candidates.  A query with more candidates than the list
// holds, or with an id that is not a 16-bit integer, goes to exact_scan_kernel<2> (fail list), as uncertified rows do
// on the Euclidean path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "exact.hip.h"

namespace sknnr {

typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));

constexpr int kHamNq = 16;          // queries per workgroup (scalar operands: 16 ids + 1 weight pair per tree pair)
constexpr int kHamWaves = 4;        // 256 reference rows per step
constexpr int kHamCand = 192;       // candidate slots per query
constexpr int kHamMaxKK = 32;

// acc += w.lo [r.lo != q.lo] + w.hi [r.hi != q.hi]  for one dword of two packed 16-bit ids: xor, min(x, 1) per half,
// weighted accumulate.  As raw instructions: hipcc expands a vector min into two compares, two selects and a byte
// permute, and pads every hand-written instruction it cannot see into with an s_nop.  The query ids and the weights are
// workgroup-uniform: scalar registers (one constant-bus operand per instruction).
// (four queries per statement: the compiler pads every asm statement with an s_nop)
__device__ __forceinline__ void ham_step4(unsigned& a0, unsigned& a1, unsigned& a2, unsigned& a3, uint32_t r, uint32_t q0, uint32_t q1,
                                          uint32_t q2, uint32_t q3, uint32_t w, uint32_t ones) {
    uint32_t t0, t1, t2, t3;
    asm("v_xor_b32 %[t0], %[q0], %[r]\n\t"
        "v_xor_b32 %[t1], %[q1], %[r]\n\t"
        "v_xor_b32 %[t2], %[q2], %[r]\n\t"
        "v_xor_b32 %[t3], %[q3], %[r]\n\t"
        "v_pk_min_u16 %[t0], %[t0], %[ones]\n\t"
        "v_pk_min_u16 %[t1], %[t1], %[ones]\n\t"
        "v_pk_min_u16 %[t2], %[t2], %[ones]\n\t"
        "v_pk_min_u16 %[t3], %[t3], %[ones]\n\t"
        "v_dot2_u32_u16 %[a0], %[t0], %[w], %[a0]\n\t"
        "v_dot2_u32_u16 %[a1], %[t1], %[w], %[a1]\n\t"
        "v_dot2_u32_u16 %[a2], %[t2], %[w], %[a2]\n\t"
        "v_dot2_u32_u16 %[a3], %[t3], %[w], %[a3]"
        : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
        : [r] "v"(r), [q0] "v"(q0), [q1] "v"(q1), [q2] "v"(q2), [q3] "v"(q3), [w] "v"(w), [ones] "v"(ones));
}
__device__ __forceinline__ void ham_step(unsigned& acc, uint32_t r, uint32_t q, uint32_t w, uint32_t ones) {
    uint32_t t;
    asm("v_xor_b32 %[t], %[q], %[r]\n\t"
        "v_pk_min_u16 %[t], %[t], %[ones]\n\t"
        "v_dot2_u32_u16 %[acc], %[t], %[w], %[acc]"
        : [acc] "+v"(acc), [t] "=&v"(t)
        : [r] "v"(r), [q] "v"(q), [w] "v"(w), [ones] "v"(ones));
}

// The same four steps with the queries' ids and the weight pair in SCALAR registers (one constant-bus operand per
// instruction): they are workgroup-uniform, a scalar load fetches the sixteen ids of a tree pair in one instruction, and the
// 64 vector registers that held them (four tree pairs in flight) are what kept the kernel at four waves per SIMD -- too few
// to cover the loads it waits for at the top of every trip (0.55 of the instruction roofline, profiles/r03_bench_output.json).
#ifndef SKNNR_HAM_SGPR
#define SKNNR_HAM_SGPR 1
#endif
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void ham_step4s(unsigned& a0, unsigned& a1, unsigned& a2, unsigned& a3, uint32_t r, uint32_t q0, uint32_t q1,
                                           uint32_t q2, uint32_t q3, uint32_t w, uint32_t ones) {
    uint32_t t0, t1, t2, t3;
    asm volatile("v_xor_b32 %[t0], %[q0], %[r]\n\t"  // (volatile: stays behind the s_waitcnt that covers its scalar operands)
        "v_xor_b32 %[t1], %[q1], %[r]\n\t"
        "v_xor_b32 %[t2], %[q2], %[r]\n\t"
        "v_xor_b32 %[t3], %[q3], %[r]\n\t"
        "v_pk_min_u16 %[t0], %[t0], %[ones]\n\t"
        "v_pk_min_u16 %[t1], %[t1], %[ones]\n\t"
        "v_pk_min_u16 %[t2], %[t2], %[ones]\n\t"
        "v_pk_min_u16 %[t3], %[t3], %[ones]\n\t"
        "v_dot2_u32_u16 %[a0], %[t0], %[w], %[a0]\n\t"
        "v_dot2_u32_u16 %[a1], %[t1], %[w], %[a1]\n\t"
        "v_dot2_u32_u16 %[a2], %[t2], %[w], %[a2]\n\t"
        "v_dot2_u32_u16 %[a3], %[t3], %[w], %[a3]"
        : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
        : [r] "v"(r), [q0] "s"(q0), [q1] "s"(q1), [q2] "s"(q2), [q3] "s"(q3), [w] "s"(w), [ones] "v"(ones));
}
// sixteen consecutive dwords (64-byte aligned) and one dword through the scalar cache
__device__ __forceinline__ void ham_sload(u32x16& q, uint32_t& w, const uint32_t* qp, const uint32_t* wp) {
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0" : "=&s"(q), "=&s"(w) : "s"(qp), "s"(wp) : "memory");
}

// many neighbours: the candidate lists are seeded and compacted (hamming_coarse_kernel), which takes 12 KB more LDS
__host__ __device__ constexpr bool ham_compacts(int kk) { return kk >= 8; }
__host__ __device__ constexpr size_t hamming_coarse_lds(int kk) { return ham_compacts(kk) ? (size_t)16 * 192 * sizeof(unsigned) : 0; }

struct HammingArgs {
    const uint32_t* rimg;   // [tp][n_ref_pad] two 16-bit ids per dword
    const uint32_t* wq;     // [tp] two 16-bit weights per dword
    const uint32_t* qimg;   // [tp][nq_pad] packed query ids
    const int* q_bad;       // (nq_pad) 1: the row's ids are not 16-bit integers
    int n_ref, n_ref_pad, tp;
    long nq, nq_pad;
    int kk;
    unsigned band;
    int* cand_cnt;          // (nq) out: candidates, or -1 = overflow / bad ids
    int* cand_id;           // (nq, kHamCand) out, ascending
};

// float64 ids -> packed 16-bit ids [tp][nq_pad]; rows with an id that is not an integer in [0, 65535] are flagged.
#ifdef SKNNR_KERNELS_HAMMING
__global__ void __launch_bounds__(256) hamming_pack_kernel(const double* __restrict__ xq, long nq, long nq_pad, int t, int tp,
                                                           uint32_t* __restrict__ qimg, int* __restrict__ q_bad) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq_pad) return;
    bool bad = false;
    for (int p = 0; p < tp; ++p) {
        uint32_t packed = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 2 * p + h;
            double v = (q < nq && c < t) ? xq[q * t + c] : 0.0;
            const bool ok = v >= 0.0 && v <= 65535.0 && v == floor(v);
            bad |= !ok;
            packed |= (ok ? (uint32_t)v : 0u) << (16 * h);
        }
        qimg[(size_t)p * nq_pad + q] = packed;
    }
    q_bad[q] = (q < nq && bad) ? 1 : 0;
}
#endif  // SKNNR_KERNELS_HAMMING

#ifdef SKNNR_KERNELS_HAMMING
__global__ void __launch_bounds__(kHamWaves * 64) hamming_coarse_kernel(HammingArgs a) {
    __shared__ unsigned dbuf[kHamNq][kHamWaves * 64];        // D^ of the step's 256 rows for the 16 queries
    __shared__ unsigned top[kHamNq][kHamMaxKK];              // the kk smallest D^ so far, ascending
    __shared__ int cnt[kHamNq];
    __shared__ int cand[kHamNq][kHamCand];
    // D^ of every candidate, for the compaction below: dynamic LDS, present only when many neighbours are searched
    // (ham_compacts(kk)) -- a call for a few neighbours never fills its lists and keeps five workgroups per CU (30 KB each)
    extern __shared__ unsigned cval_dyn[];
    unsigned (*cval)[kHamCand] = (unsigned (*)[kHamCand])cval_dyn;
    __shared__ unsigned seed_kth[kHamNq];                     // upper bound of the final kk-th smallest D^ (seeding pass)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long q0 = (long)blockIdx.x * kHamNq;
    for (int i = tid; i < kHamNq * kHamMaxKK; i += kHamWaves * 64) top[i / kHamMaxKK][i % kHamMaxKK] = 0xffffffffu;
    if (tid < kHamNq) {
        cnt[tid] = 0;
        seed_kth[tid] = 0xffffffffu;
    }
    __syncthreads();
    const int KK = a.kk;
    const bool compacts = ham_compacts(KK);
    // Seeding (many neighbours): rows that share no leaf with a query all sit at the SAME largest distance, and while fewer
    // than kk closer rows have been seen every one of them is within the running bound -- the candidate list of a query
    // filled up in the first few hundred rows whatever its length (kk = 16: a fifth of the queries, kk = 32: nearly all fell
    // to the float64 scan).  A first pass over a prefix of the rows only ranks (no candidates): its kk-th smallest value
    // bounds the final one from above, and the real sweep starts with it.
    const int seed_rows = compacts ? min(a.n_ref, max(256, min(4096, a.n_ref / 16)) / 256 * 256) : 0;
    const uint32_t* qbase = a.qimg + q0;  // + p * nq_pad: 16 consecutive dwords, workgroup-uniform
    uint32_t ones = 0x00010001u;
    asm volatile("" : "+v"(ones));  // (a vector register, loaded once)

    for (int phase = seed_rows > 0 ? 0 : 1; phase < 2; ++phase) {
    const int j_end = phase == 0 ? seed_rows : a.n_ref;
    for (int j0 = 0; j0 < j_end; j0 += kHamWaves * 64) {
        const int r = j0 + tid;
        const uint32_t* rcol = a.rimg + (r < a.n_ref_pad ? r : 0);
        unsigned acc[kHamNq];
#pragma unroll
        for (int j = 0; j < kHamNq; ++j) acc[j] = 0;
        int p = 0;
#if SKNNR_HAM_SGPR
        for (; p + 4 <= a.tp; p += 4) {
            // four tree pairs per trip: the reference dwords by vector loads, then per tree pair one scalar load of the sixteen
            // queries' ids (and the weight pair) and 48 instructions of arithmetic; the next pair's scalar load is in flight
            // meanwhile
            uint32_t rv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) rv[u] = rcol[(size_t)(p + u) * a.n_ref_pad];
            u32x16 qs[2];
            uint32_t ws[2];
            ham_sload(qs[0], ws[0], qbase + (size_t)p * a.nq_pad, a.wq + p);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // (scalar loads return out of order: the only wait is for all of them -- so the next pair's load goes out
                //  once this pair's has landed, and travels during this pair's arithmetic)
                // (the loaded registers are operands of the wait: what consumes them below depends on THIS statement, so nothing
                //  the compiler schedules can read them while the load is in flight -- ADVICE r3)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(qs[u & 1]), "+s"(ws[u & 1]) : : "memory");
                if (u + 1 < 4) ham_sload(qs[(u + 1) & 1], ws[(u + 1) & 1], qbase + (size_t)(p + u + 1) * a.nq_pad, a.wq + p + u + 1);
                const u32x16& q = qs[u & 1];
                const uint32_t w = ws[u & 1];
                ham_step4s(acc[0], acc[1], acc[2], acc[3], rv[u], q[0], q[1], q[2], q[3], w, ones);
                ham_step4s(acc[4], acc[5], acc[6], acc[7], rv[u], q[4], q[5], q[6], q[7], w, ones);
                ham_step4s(acc[8], acc[9], acc[10], acc[11], rv[u], q[8], q[9], q[10], q[11], w, ones);
                ham_step4s(acc[12], acc[13], acc[14], acc[15], rv[u], q[12], q[13], q[14], q[15], w, ones);
            }
        }
#else
        for (; p + 4 <= a.tp; p += 4) {
            // four tree pairs per trip: the reference dwords, the weights and the sixteen queries' ids (workgroup-uniform
            // 64-byte rows) of all four are requested together, then 192 instructions of arithmetic
            uint32_t rv[4], wv[4];
            uint4 qv[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                rv[u] = rcol[(size_t)(p + u) * a.n_ref_pad];
                wv[u] = a.wq[p + u];
                const uint4* qp = (const uint4*)(qbase + (size_t)(p + u) * a.nq_pad);
#pragma unroll
                for (int g = 0; g < 4; ++g) qv[u][g] = qp[g];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    ham_step4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3], rv[u], qv[u][g].x, qv[u][g].y, qv[u][g].z,
                              qv[u][g].w, wv[u], ones);
            }
        }
#endif
        for (; p < a.tp; ++p) {
            const uint32_t rvv = rcol[(size_t)p * a.n_ref_pad];
            const uint32_t w = a.wq[p];
            const uint32_t* qp = qbase + (size_t)p * a.nq_pad;
#pragma unroll
            for (int j = 0; j < kHamNq; ++j) ham_step(acc[j], rvv, qp[j], w, ones);
        }
        __syncthreads();  // the previous step's selection is done with dbuf
#pragma unroll
        for (int j = 0; j < kHamNq; ++j) dbuf[j][tid] = r < a.n_ref ? acc[j] : 0xffffffffu;
        __syncthreads();
        // selection: wave w takes queries 4w .. 4w+3; rows in ascending index order
#pragma unroll 1
        for (int jj = 0; jj < kHamNq / kHamWaves; ++jj) {
            const int j = wave * (kHamNq / kHamWaves) + jj;
            if (q0 + j >= a.nq) break;
            unsigned kth = min(top[j][KK - 1], seed_kth[j]);
#pragma unroll 1
            for (int u = 0; u < kHamWaves; ++u) {
                const unsigned v = dbuf[j][64 * u + lane];
                const unsigned lim = kth > 0xffffffffu - a.band ? 0xffffffffu : kth + a.band;
                unsigned long long m = __builtin_amdgcn_ballot_w64(v <= lim && v != 0xffffffffu);
                while (m) {
                    const int bit = __builtin_ctzll(m);
                    m &= m - 1;
                    const unsigned vv = __shfl(v, bit, 64);
                    const unsigned lim2 = kth > 0xffffffffu - a.band ? 0xffffffffu : kth + a.band;
                    if (vv > lim2) continue;  // the bound has dropped since the ballot
                    if (phase == 0) {  // seeding pass: rank only
                        if (lane == 0 && vv < top[j][KK - 1]) {
                            int pos = KK - 1;
                            while (pos > 0 && top[j][pos - 1] > vv) {
                                top[j][pos] = top[j][pos - 1];
                                --pos;
                            }
                            top[j][pos] = vv;
                        }
                        __builtin_amdgcn_wave_barrier();
                        kth = top[j][KK - 1];
                        continue;
                    }
                    if (compacts && cnt[j] == kHamCand) {
                        // The list is full of rows admitted against EARLIER, looser bounds (about kk (1 + ln(n_ref / kk)) rows
                        // pass the running bound of a sweep in index order): keep what is still within band of the current
                        // kk-th smallest value -- in place, order preserved, the whole wave at work.  Only a list that is
                        // still full afterwards overflows.
                        int kept = 0;
#pragma unroll 1
                        for (int c0 = 0; c0 < kHamCand; c0 += 64) {
                            const int id = cand[j][c0 + lane];
                            const unsigned cv = cval[j][c0 + lane];
                            const bool keep = cv <= lim2;
                            const unsigned long long km = __builtin_amdgcn_ballot_w64(keep);
                            const int dst = kept + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u));
                            __builtin_amdgcn_wave_barrier();  // (every lane has read its slot before any slot is rewritten: dst <= c0 + lane)
                            if (keep) {
                                cand[j][dst] = id;
                                cval[j][dst] = cv;
                            }
                            kept += __builtin_popcountll(km);
                            __builtin_amdgcn_wave_barrier();
                        }
                        if (lane == 0) cnt[j] = kept;
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (lane == 0) {
                        const int c = cnt[j];
                        if (c >= 0) {
                            if (c < kHamCand) {
                                cand[j][c] = j0 + 64 * u + bit;
                                if (compacts) cval[j][c] = vv;
                                cnt[j] = c + 1;
                            } else {
                                cnt[j] = -1;  // still full after the compaction: the exact scan answers this query
                            }
                        }
                        if (vv < top[j][KK - 1]) {  // keep the kk smallest, ascending
                            int pos = KK - 1;
                            while (pos > 0 && top[j][pos - 1] > vv) {
                                top[j][pos] = top[j][pos - 1];
                                --pos;
                            }
                            top[j][pos] = vv;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    kth = min(top[j][KK - 1], seed_kth[j]);
                }
            }
        }
    }
    if (phase == 0) {  // the seed's kk-th smallest value becomes the starting bound; the ranking starts over (same rows again)
        __syncthreads();
        if (tid < kHamNq) seed_kth[tid] = top[tid][KK - 1];
        __syncthreads();
        for (int i = tid; i < kHamNq * kHamMaxKK; i += kHamWaves * 64) top[i / kHamMaxKK][i % kHamMaxKK] = 0xffffffffu;
        __syncthreads();
    }
    }  // phases
    __syncthreads();
    // out go the candidates that are still within band of the FINAL kk-th smallest value (the early ones were admitted against
    // looser bounds): ascending order kept, the re-score reads fewer rows
    for (int j = wave; j < kHamNq; j += kHamWaves) {
        const long q = q0 + j;
        if (q >= a.nq) break;
        const int c = a.q_bad[q] ? -1 : cnt[j];
        if (c < 0) {
            if (lane == 0) a.cand_cnt[q] = -1;
            continue;
        }
        const unsigned kth = top[j][KK - 1];  // (the full ranking: at most the seed's bound)
        const unsigned lim = kth > 0xffffffffu - a.band ? 0xffffffffu : kth + a.band;
        int kept = 0;
        for (int c0 = 0; c0 < c; c0 += 64) {
            const int i = c0 + lane;
            const bool keep = i < c && (!compacts || cval[j][i < c ? i : 0] <= lim);
            const unsigned long long km = __builtin_amdgcn_ballot_w64(keep);
            const int dst = kept + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u));
            if (keep) a.cand_id[q * kHamCand + dst] = cand[j][i];
            kept += __builtin_popcountll(km);
        }
        if (lane == 0) a.cand_cnt[q] = kept;
    }
}
#endif  // SKNNR_KERNELS_HAMMING

struct HammingRescoreArgs {
    SelectArgs s;           // s.xq: (nq, d) float64 ids of the queries; s.ref: (n_ref, d) float64 ids; s.hw, s.hw_sum
    const int* cand_cnt;
    const int* cand_id;
    int* fail_list;         // queries for exact_scan_kernel<2> (overflow, bad ids)
    int* fail_count;
    int fail_base;
    const uint32_t* rrow;   // [n_ref][tpr] row-major 16-bit ids of the references (hamming_rows_kernel)
    int tpr;
};

// Row-major copy of the references' 16-bit ids for the re-score: [row][tpr] dwords (two ids each), tpr a multiple of 256
// (one KiB per 512 trees), zeros past the last tree.
__host__ __device__ constexpr int ham_row_dwords(int t) { return (t + 511) / 512 * 256; }
#ifdef SKNNR_KERNELS_HAMMING
__global__ void __launch_bounds__(256) hamming_rows_kernel(const double* __restrict__ x, long n, int t, int tpr, uint32_t* __restrict__ rows) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // one dword per thread
    if (i >= n * tpr) return;
    const long row = i / tpr;
    const int p = (int)(i - row * tpr);
    uint32_t packed = 0;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int c = 2 * p + hh;
        const double v = c < t ? x[row * t + c] : 0.0;
        packed |= ((v >= 0.0 && v <= 65535.0) ? (uint32_t)v : 0u) << (16 * hh);
    }
    rows[i] = packed;
}
#endif  // SKNNR_KERNELS_HAMMING

// One wave per query.  The candidates' distances in the reference's float64 arithmetic -- s += w_t for every tree whose ids
// differ, IN TREE ORDER -- then lane 0 selects by (distance, index) in ascending index order and finishes, exactly as
// exact_scan_kernel<2> does over all rows.
//   phase 1, per candidate, all lanes: which trees differ?  Lane l compares trees 512 c + 8 l .. + 7 of chunk c on the 16-bit
//            ids (equal as float64 iff equal as integers: both sides were validated), one coalesced KiB of the row-major
//            image per chunk, and leaves one byte of flags in LDS;
//   phase 2, lane i = candidate i of the batch of 64: the weights of the flagged trees are summed in tree order (flags and
//            weights from LDS: no memory traffic, the float64 additions are the same chain the reference runs).
// (Until round 3 every lane walked its own candidate's float64 row, 8 bytes per load out of 64 different rows: 19.3 ms of
//  the 105 ms of the 200k x 20k x 500 benchmark call.)
#ifdef SKNNR_KERNELS_HAMMING
__global__ void __launch_bounds__(256) hamming_rescore_kernel(HammingRescoreArgs a) {
    __shared__ double dv[4][kHamCand];
    __shared__ double hv[4][kHamMaxKK + 2];
    __shared__ int hi[4][kHamMaxKK + 2];
    __shared__ int stack[4][2 * kHamMaxKK + 8];
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_lds[];  // [T] float64 weights, then per wave [64][chunks * 64] flag bytes
    const SelectArgs& s = a.s;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int T = s.d, chunks = (T + 511) / 512, fb = chunks * 64;  // flag bytes per candidate
    double* hw_l = (double*)rs_lds;
    unsigned char* flags = rs_lds + (size_t)((T + 1) / 2 * 2) * 8 + (size_t)wave * 64 * fb;
    for (int t = threadIdx.x; t < T; t += 256) hw_l[t] = s.hw[t];
    __syncthreads();
    const long q = (long)blockIdx.x * 4 + wave;
    if (q >= s.nq) return;
    const int c = a.cand_cnt[q];
    if (c < s.kk) {  // overflow, ids outside 16 bits (c = -1), or fewer candidates than rows asked for: the full scan
        if (lane == 0) {
            const int slot = atomicAdd(a.fail_count, 1);
            a.fail_list[slot] = a.fail_base + (int)q;
        }
        return;
    }
    // this lane's eight trees of every chunk of the query, as four dwords of 16-bit ids
    constexpr int kMaxChunks = 8;  // 4,096 trees
    uint4 qv[kMaxChunks];
    const double* x = s.xq + q * T;
#pragma unroll
    for (int k = 0; k < kMaxChunks; ++k) {
        uint32_t w4[4] = {0, 0, 0, 0};
        if (k < chunks) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int t = 512 * k + 8 * lane + i;
                const uint32_t id = t < T ? (uint32_t)x[t] : 0u;
                w4[i >> 1] |= id << (16 * (i & 1));
            }
        }
        qv[k] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
    for (int b0 = 0; b0 < c; b0 += 64) {
        const int nb = min(64, c - b0);
        for (int i = 0; i < nb; ++i) {  // phase 1
            const uint32_t* row = a.rrow + (size_t)a.cand_id[q * kHamCand + b0 + i] * a.tpr;
#pragma unroll
            for (int k = 0; k < kMaxChunks; ++k) {
                if (k >= chunks) break;
                const uint4 rv = *(const uint4*)(row + 256 * k + 4 * lane);
                const uint32_t d0 = rv.x ^ qv[k].x, d1 = rv.y ^ qv[k].y, d2 = rv.z ^ qv[k].z, d3 = rv.w ^ qv[k].w;
                const uint32_t f = ((d0 & 0xffffu) ? 1u : 0u) | ((d0 >> 16) ? 2u : 0u) | ((d1 & 0xffffu) ? 4u : 0u) | ((d1 >> 16) ? 8u : 0u) |
                                   ((d2 & 0xffffu) ? 16u : 0u) | ((d2 >> 16) ? 32u : 0u) | ((d3 & 0xffffu) ? 64u : 0u) | ((d3 >> 16) ? 128u : 0u);
                flags[(size_t)i * fb + 64 * k + lane] = (unsigned char)f;
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane < nb) {  // phase 2
            const unsigned char* fl = flags + (size_t)lane * fb;
            double acc = 0.0;
            for (int t8 = 0; t8 * 8 < T; ++t8) {
                const unsigned f = fl[t8];
                const int t0 = 8 * t8;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (t0 + i < T) acc = (f >> i) & 1u ? acc + hw_l[t0 + i] : acc;
            }
            dv[wave][b0 + lane] = acc / s.hw_sum;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane == 0) {
        const int KK = s.kk;
        for (int e = 0; e < KK; ++e) {
            hv[wave][e] = DBL_MAX;
            hi[wave][e] = 0;
        }
        for (int i = 0; i < c; ++i) {
            const double v = dv[wave][i];
            if (v < hv[wave][KK - 1]) sorted_insert_ref(hv[wave], hi[wave], KK, v, a.cand_id[q * kHamCand + i]);
        }
        scan_finish_query<2>(s, q, hv[wave], hi[wave], stack[wave]);
    }
}
#endif  // SKNNR_KERNELS_HAMMING
__host__ inline size_t hamming_rescore_lds(int t) { return (size_t)((t + 1) / 2 * 2) * 8 + (size_t)4 * 64 * ((t + 511) / 512 * 64); }

// Full float64 distance rows of selected queries: out[i][j] = the weighted Hamming distance between query rows[i] and
// reference row j, in the reference's arithmetic (s += (u != v) * w in tree order, / sum w: scipy's cdist as reached from
// REF _weighted_trees.py:53-59, :139-140 -- the same chain as exact_scan_kernel<2> and hamming_rescore_kernel).  Serves
// sknnr_hamming_distances: the rows whose k-th distance is tied are selected on the host by np.argpartition, exactly as
// SKL/neighbors/_base.py:733-760 (_kneighbors_reduce_func) does, when the caller asks for the reference's own choice
// among exactly tied rows.
struct HammingRowsArgs {
    const double* xq;    // (., d) float64 node ids of the query rows
    const long* rows;    // (n_rows) which rows of xq, or null: rows 0 .. n_rows - 1
    long n_rows;
    const double* refT;  // (d, n_ref) transposed reference ids
    int d, n_ref;
    const double* hw;    // (d) weights
    double hw_sum;       // their sum in index order
    double* out;         // (n_rows, n_ref)
};
#ifdef SKNNR_KERNELS_HAMMING
__global__ void __launch_bounds__(256) hamming_distance_rows_kernel(HammingRowsArgs a) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.n_ref) return;
    for (long i = blockIdx.y; i < a.n_rows; i += gridDim.y) {
        const double* x = a.xq + (a.rows ? a.rows[i] : i) * a.d;
        double acc = 0.0;
        for (int c = 0; c < a.d; ++c) {
            const double r = a.refT[(size_t)c * a.n_ref + j];
            acc = x[c] != r ? acc + a.hw[c] : acc;
        }
        a.out[(size_t)i * a.n_ref + j] = acc / a.hw_sum;
    }
}
#endif  // SKNNR_KERNELS_HAMMING

}  // namespace sknnr
