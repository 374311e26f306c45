// coarse.hip.h -- the MFMA pre-filter: for every query, the M smallest approximate
// ranking values  v(q, r) ~= |r'|^2 - 2 q'.r'   (primes: centred and scaled features)
// seen by each of the two lanes that own the query, over ALL reference rows.
//
// Replaces the hot double loop of the reference's engine
//   dgemm  M = -2 Xc Yc^T            SKL/metrics/_pairwise_distances_reduction/_middle_term_computer.pyx.tp:440
//   d2 = |x|^2 + M + |y|^2, heap_push SKL/metrics/_pairwise_distances_reduction/_argkmin.pyx.tp:492-510
// as a candidate generator; the float64 finaliser (finalize.hip.h) re-scores the
// candidates with the reference's exact expression and certifies the result.
//
// MI355X mapping
//   * contraction on the f16 matrix pipe (v_mfma_f32_32x32x16_f16, 16x the f32 MFMA rate)
//     with every operand split x = hi + lo (two f16 each) and three products
//     hi.hi + lo.hi + hi.lo accumulated in f32  ->  ~2^-22 relative error, i.e. f32-class
//     accuracy at 16/3 of the f32-MFMA throughput;
//   * references are the A operand (rows), queries the B operand (columns): the 32x32
//     accumulator then holds ONE query per lane (col = lane & 31) and 16 references in the
//     lane's registers, so the running top-M is lane-local (no cross-lane traffic in the
//     sweep); lanes l and l+32 share a query and keep one list each;
//   * |r'|^2 enters as the C operand of the first MFMA (exact f32, no VALU add);
//   * reference tiles are stored in HBM in MFMA-fragment order and copied to LDS by
//     LDS-DMA (global_load_lds_dwordx4), double buffered; every wave of the 512-thread
//     workgroup reads the same staged tile, lane-linear ds_read_b128 (conflict-free);
//   * queries live in registers for the whole sweep (64 per wave, 512 per workgroup).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

namespace sknnr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// One 32-reference tile record in HBM/LDS: [part hi|lo][K-step][lane] 16 B, then the
// C-operand init  [lane half][16] f32.
__host__ __device__ constexpr int tile_frag_bytes(int ks) { return 2 * ks * 1024; }
__host__ __device__ constexpr int tile_bytes(int ks) { return 2 * ks * 1024 + 128; }
// 32-reference tiles per LDS stage (stage <= ~33 KiB so two stages fit beside anything).
__host__ __device__ constexpr int tiles_per_stage(int ks) { return ks <= 2 ? 8 : (ks <= 4 ? 4 : 2); }
// Row of the 32x32 accumulator held in register r of a lane in half h (guide section 3).
__host__ __device__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

constexpr int kCoarseWaves = 8;
constexpr int kCoarseThreads = kCoarseWaves * 64;

__device__ __forceinline__ float min3f(float a, float b, float c) {
    return __builtin_fminf(a, __builtin_fminf(b, c));
}

// Sorted (ascending) insertion of (v, id) into a lane-local list whose last entry is
// known to be > v.
template <int M>
__device__ __forceinline__ void list_insert(float (&vals)[M], int (&idxs)[M], float v, int id) {
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

// The three-product split contraction of one 32-ref x 32-query tile.
template <int KS>
__device__ __forceinline__ floatx16 split_contract(const half8 (&ah)[KS], const half8 (&al)[KS],
                                                   const half8 (&bh)[KS], const half8 (&bl)[KS],
                                                   floatx16 c0) {
    floatx16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh[0], c0, 0, 0, 0);
#pragma unroll
    for (int s = 1; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s], acc, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s], acc, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s], acc, 0, 0, 0);
    return acc;
}

// Copy `bytes` (multiple of 16) from global to LDS, lane-linear, by LDS-DMA.  Chunks of
// 1 KiB are dealt round-robin to the workgroup's waves.
__device__ __forceinline__ void stage_copy(const char* __restrict__ gsrc, char* lds_dst, int bytes,
                                           int wave, int lane) {
    const int n_chunks = (bytes + 1023) >> 10;
    for (int c = wave; c < n_chunks; c += kCoarseWaves) {
        const int off = (c << 10) + (lane << 4);
        if (off < bytes) {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(gsrc + off),
                (__attribute__((address_space(3))) void*)(lds_dst + (c << 10)), 16, 0, 0);
        }
    }
}

// KS  : 16-wide K-steps per split part (padded feature count / 16)
// M   : list length per lane
// NQB : 32-query blocks per wave
template <int KS, int M, int NQB>
__global__ void __launch_bounds__(kCoarseThreads, 2)
coarse_kernel(const char* __restrict__ rimg,   // n_stages * TPS tile records
              int n_stages,
              const uint4* __restrict__ qimg,  // [n_qblocks][2][KS][64] 16-B fragments
              float* __restrict__ cand_val,    // [n_qblocks*32][2][M]
              int* __restrict__ cand_idx) {
    constexpr int TPS = tiles_per_stage(KS);
    constexpr int TB = tile_bytes(KS);
    constexpr int STAGE = TPS * TB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    const int qb0 = (blockIdx.x * kCoarseWaves + wave) * NQB;

    // Queries of this wave: B fragments, resident for the whole sweep.
    half8 bh[NQB][KS], bl[NQB][KS];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const uint4 uh = qimg[((size_t)((qb0 + qb) * 2 + 0) * KS + s) * 64 + lane];
            const uint4 ul = qimg[((size_t)((qb0 + qb) * 2 + 1) * KS + s) * 64 + lane];
            bh[qb][s] = __builtin_bit_cast(half8, uh);
            bl[qb][s] = __builtin_bit_cast(half8, ul);
        }
    }

    float vals[NQB][M];
    int idxs[NQB][M];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
#pragma unroll
        for (int i = 0; i < M; ++i) {
            vals[qb][i] = FLT_MAX;
            idxs[qb][i] = -1;
        }
    }

    stage_copy(rimg, smem, STAGE, wave, lane);
    __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and publishes stage 0

    for (int st = 0; st < n_stages; ++st) {
        char* cur = smem + (st & 1) * STAGE;
        if (st + 1 < n_stages)
            stage_copy(rimg + (size_t)(st + 1) * STAGE, smem + ((st + 1) & 1) * STAGE, STAGE, wave, lane);

#pragma unroll 1
        for (int t = 0; t < TPS; ++t) {
            const char* tb = cur + t * TB;
            half8 ah[KS], al[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                ah[s] = *(const half8*)(tb + (0 * KS + s) * 1024 + lane * 16);
                al[s] = *(const half8*)(tb + (1 * KS + s) * 1024 + lane * 16);
            }
            floatx16 c0;
            {
                const floatx4* cp = (const floatx4*)(tb + tile_frag_bytes(KS) + half * 64);
                const floatx4 c_0 = cp[0], c_1 = cp[1], c_2 = cp[2], c_3 = cp[3];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    c0[i] = c_0[i];
                    c0[4 + i] = c_1[i];
                    c0[8 + i] = c_2[i];
                    c0[12 + i] = c_3[i];
                }
            }
            const int id_base = (st * TPS + t) * 32 + 4 * half;

#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                const floatx16 acc = split_contract<KS>(ah, al, bh[qb], bl[qb], c0);
                // lane-local minimum of the 16 new values: 8 v_min3 instead of 16 compares
                float m0 = min3f(acc[0], acc[1], acc[2]);
                float m1 = min3f(acc[3], acc[4], acc[5]);
                float m2 = min3f(acc[6], acc[7], acc[8]);
                float m3 = min3f(acc[9], acc[10], acc[11]);
                float m4 = min3f(acc[12], acc[13], acc[14]);
                m0 = min3f(m0, m1, acc[15]);
                m2 = min3f(m2, m3, m4);
                const float mn = __builtin_fminf(m0, m2);
                if (__builtin_amdgcn_ballot_w64(mn < vals[qb][M - 1]) != 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[r];
                        if (v < vals[qb][M - 1]) list_insert<M>(vals[qb], idxs[qb], v, id_base + acc_row(r, 0));
                    }
                }
            }
        }
        __syncthreads();  // next stage landed (vmcnt(0)) and everyone is done with `cur`
    }

#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const size_t q = (size_t)(qb0 + qb) * 32 + (lane & 31);
        const size_t base = (q * 2 + half) * M;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            cand_val[base + i] = vals[qb][i];
            cand_idx[base + i] = idxs[qb][i];
        }
    }
}

// Diagnostic twin of the production kernel: the same split contraction, every value
// written out.  One wave per (32-ref tile, 32-query block).
template <int KS>
__global__ void __launch_bounds__(64)
coarse_matrix_kernel(const char* __restrict__ rimg, const uint4* __restrict__ qimg, int n_ref,
                     int nq, float* __restrict__ out) {
    const int lane = threadIdx.x;
    const int tile = blockIdx.x, qblk = blockIdx.y;
    const char* tb = rimg + (size_t)tile * tile_bytes(KS);
    half8 ah[KS], al[KS], bh[KS], bl[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        ah[s] = *(const half8*)(tb + (0 * KS + s) * 1024 + lane * 16);
        al[s] = *(const half8*)(tb + (1 * KS + s) * 1024 + lane * 16);
        bh[s] = __builtin_bit_cast(half8, qimg[((size_t)(qblk * 2 + 0) * KS + s) * 64 + lane]);
        bl[s] = __builtin_bit_cast(half8, qimg[((size_t)(qblk * 2 + 1) * KS + s) * 64 + lane]);
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
        const int ref = tile * 32 + acc_row(r, half);
        if (q < nq && ref < n_ref) out[(size_t)q * n_ref + ref] = acc[r];
    }
}

}  // namespace sknnr
