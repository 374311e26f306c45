// Microbenchmark (development aid, round 3): the weighted-Hamming pre-filter on the matrix pipe.
//   Every tree's node id is hashed into 8 buckets and encoded one-hot (8 bytes of int8): agreements under the hash >= true
//   agreements, so  (sum of weights) - A.B  is a LOWER bound of the weighted Hamming distance.  v_mfma_i32_32x32x32_i8
//   takes 4 trees per instruction (K = 32 bytes); the operands are expanded in registers from 3-bit codes (4 bits per
//   tree in LDS), so neither HBM nor LDS ever sees the 16x inflated encoding.
// Part 1 checks the operand lane map of the instruction with exact integer data; part 2 times the inner loop
// (2 reference tiles x 2 query blocks per wave, 4 waves per workgroup, codes in LDS) and prints compares/s.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/microbench/hamming_mfma.hip -o scripts/microbench/hamming_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx16 __attribute__((ext_vector_type(16)));

// ---- part 1: lane map -------------------------------------------------------------------------------------------
// assumed: lane l (r = l & 31, h = l >> 5) holds A[row r][k = 16 h + j], B[k = 16 h + j][col r], j = 0..15 (byte j of the
// 128-bit operand); D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h.
__global__ void map_check(const int8_t* A, const int8_t* B, int* D) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    intx4 a, b;
    int8_t* ab = (int8_t*)&a;
    int8_t* bb = (int8_t*)&b;
    for (int j = 0; j < 16; ++j) {
        ab[j] = A[r * 32 + 16 * h + j];
        bb[j] = B[(16 * h + j) * 32 + r];
    }
    intx16 c = {};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int reg = 0; reg < 16; ++reg) D[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = c[reg];
}

// ---- part 2: the inner loop -----------------------------------------------------------------------------------------
constexpr int kWaves = 4;
// one-hot fragment of this lane's two trees of K-step s from the row's nibble-packed codes (dword s/2 holds 8 trees)
template <bool WEIGHTED>
__device__ __forceinline__ intx4 expand(uint32_t w, int off, uint32_t wgt4, int woff) {
    const uint32_t c0 = __builtin_amdgcn_ubfe(w, off, 3), c1 = __builtin_amdgcn_ubfe(w, off + 4, 3);
    unsigned long long u0 = 1, u1 = 1;
    if (WEIGHTED) {
        u0 = __builtin_amdgcn_ubfe(wgt4, woff, 8);
        u1 = __builtin_amdgcn_ubfe(wgt4, woff + 8, 8);
    }
    const unsigned long long lo = u0 << (8 * c0), hi = u1 << (8 * c1);
    intx4 f;
    f[0] = (int)(unsigned)lo;
    f[1] = (int)(unsigned)(lo >> 32);
    f[2] = (int)(unsigned)hi;
    f[3] = (int)(unsigned)(hi >> 32);
    return f;
}

template <bool WEIGHTED>
__global__ void __launch_bounds__(kWaves * 64, 1)
sweep(const uint32_t* __restrict__ rcodes, const uint32_t* __restrict__ qcodes, const uint32_t* __restrict__ wq, int tq2, int n_ref_tiles, int thr,
      int* out, int* hits) {
    // LDS: reference codes of the tile pair [64 rows][tq2 dwords] (single buffer here), query codes [wave][64 rows][tq2]
    extern __shared__ uint32_t lds[];
    uint32_t* rl = lds;
    uint32_t* ql = lds + 64 * tq2 + threadIdx.x / 64 * 64 * tq2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    for (int i = lane; i < 64 * tq2; i += 64) ql[i] = qcodes[((size_t)(blockIdx.x * kWaves + wave) * 64 * tq2 + i) % (4096 * tq2)];
    int n_hit = 0, sum = 0;
    for (int rt = 0; rt < n_ref_tiles; rt += 2) {
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * tq2; i += kWaves * 64) rl[i] = rcodes[(size_t)rt * 32 * tq2 + i];
        __syncthreads();
        intx16 acc[2][2] = {};
        for (int s2 = 0; s2 < tq2; ++s2) {
            const uint32_t ra = rl[r * tq2 + s2], rb = rl[(32 + r) * tq2 + s2];
            const uint32_t qa = ql[r * tq2 + s2], qb = ql[(32 + r) * tq2 + s2];
            const uint32_t wg0 = WEIGHTED ? wq[2 * s2] : 0, wg1 = WEIGHTED ? wq[2 * s2 + 1] : 0;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int off = 16 * p + 8 * h;
                const intx4 fa = expand<false>(ra, off, 0, 0), fb = expand<false>(rb, off, 0, 0);
                const intx4 ga = expand<WEIGHTED>(qa, off, p ? wg1 : wg0, 16 * h), gb = expand<WEIGHTED>(qb, off, p ? wg1 : wg0, 16 * h);
                acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, ga, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, gb, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb, ga, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb, gb, acc[1][1], 0, 0, 0);
            }
        }
        // skip test: the largest agreement of the lane's sixteen rows against the query's threshold
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                int m = acc[a][b][0];
#pragma unroll
                for (int i = 1; i < 16; ++i) m = max(m, acc[a][b][i]);
                if (__builtin_amdgcn_ballot_w64(m >= thr) != 0) n_hit += 1;
                sum += m;
            }
    }
    if (lane == 0) atomicAdd(hits, n_hit);
    out[blockIdx.x * kWaves * 64 + threadIdx.x] = sum;
}

int main() {
    // part 1
    {
        std::vector<int8_t> A(1024), B(1024);
        srand(3);
        for (auto& v : A) v = (int8_t)(rand() % 7 - 3);
        for (auto& v : B) v = (int8_t)(rand() % 9 - 4);
        int8_t *dA, *dB;
        int* dD;
        (void)hipMalloc(&dA, 1024); (void)hipMalloc(&dB, 1024); (void)hipMalloc(&dD, 4096);
        (void)hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice);
        (void)hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
        map_check<<<1, 64>>>(dA, dB, dD);
        std::vector<int> D(1024);
        (void)hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                int want = 0;
                for (int k = 0; k < 32; ++k) want += (int)A[i * 32 + k] * (int)B[k * 32 + j];
                bad += want != D[i * 32 + j];
            }
        printf("lane map of v_mfma_i32_32x32x32_i8 (A[r][16h+j], B[16h+j][r], D col = l&31): %d of 1024 results differ\n", bad);
    }
    // part 2: 500 trees -> 63 dwords of nibble codes per row (125 K-steps of 4 trees), 20,000 reference rows = 625 tiles
    const int trees = 500, tq2 = (trees + 7) / 8, n_ref_tiles = 624;
    std::vector<uint32_t> rc((size_t)n_ref_tiles * 32 * tq2), qc((size_t)4096 * tq2), wq(2 * tq2);
    for (auto& v : rc) v = ((uint32_t)rand() << 16 ^ (uint32_t)rand()) & 0x77777777u;
    for (auto& v : qc) v = ((uint32_t)rand() << 16 ^ (uint32_t)rand()) & 0x77777777u;
    for (auto& v : wq) v = 0x01010101u * (uint32_t)(1 + rand() % 100);
    uint32_t *drc, *dqc, *dwq;
    int *dout, *dhits;
    (void)hipMalloc(&drc, rc.size() * 4); (void)hipMalloc(&dqc, qc.size() * 4); (void)hipMalloc(&dwq, wq.size() * 4);
    (void)hipMalloc(&dout, 4096 * 256 * 4); (void)hipMalloc(&dhits, 4);
    (void)hipMemcpy(drc, rc.data(), rc.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dqc, qc.data(), qc.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dwq, wq.data(), wq.size() * 4, hipMemcpyHostToDevice);
    const size_t sh = (size_t)(64 + kWaves * 64) * tq2 * 4;
    for (int weighted = 0; weighted < 2; ++weighted) {
        auto kern = weighted ? sweep<true> : sweep<false>;
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        const int wgs = 256 * 3;  // 768 workgroups x 256 queries = 196,608 queries
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipMemset(dhits, 0, 4);
            (void)hipEventRecord(e0);
            kern<<<wgs, kWaves * 64, sh>>>(drc, dqc, dwq, tq2, n_ref_tiles, weighted ? 200 * 60 : 200, dout, dhits);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        int hits = 0;
        (void)hipMemcpy(&hits, dhits, 4, hipMemcpyDeviceToHost);
        const double pairs = (double)wgs * 256 * n_ref_tiles * 32;
        const double cmp = pairs * trees;
        const double mfma = pairs / 1024 * 2 * tq2;
        printf("%s weights: %d workgroups, %.3f ms: %.1fe12 compares/s, %.0f TOP/s int8 executed (%.3f of 5000), %s; visited units %d\n",
               weighted ? "per-tree" : "uniform", wgs, best, cmp / (best * 1e-3) / 1e12, mfma * 65536 / (best * 1e-3) / 1e12,
               mfma * 65536 / (best * 1e-3) / 1e12 / 5000.0, hipGetErrorString(hipGetLastError()), hits);
    }
    return 0;
}
