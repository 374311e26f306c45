// Microbenchmark (development aid): does wave priority change whether MFMAs of one wave overlap
// VALU work of ANOTHER wave on the same SIMD?  (mfma_valu_overlap.hip: with equal priorities they
// do not -- an MFMA-only wave and a VALU-only wave on one SIMD take the sum of their solo times.)
//   role split : waves w and w + 4 share a SIMD (8-wave workgroups, one per CU): waves 0-3 issue
//                MFMA chains, waves 4-7 v_min3 chains; s_setprio per role
//   sweep-like : every wave runs the pre-filter's inner pattern (2 dependent MFMAs, then a 7-op min
//                tree + compare + ballot on the result); 4 waves per SIMD; s_setprio around the MFMAs /
//                around the tree
// build: hipcc --offload-arch=gfx950 -O3 scripts/microbench/mfma_valu_prio.hip -o scripts/microbench/mfma_valu_prio
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int PM, int PV>
__global__ void __launch_bounds__(512) split_roles(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    floatx16 acc = {}, acc2 = {};
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    const bool do_mfma = (wave & 4) == 0;
    if (do_mfma) __builtin_amdgcn_s_setprio(PM); else __builtin_amdgcn_s_setprio(PV);
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc2, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2));
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(v2), "v"(v3));
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(v3), "v"(v0));
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(v0), "v"(v1));
            }
        }
    }
    float s = v0 + v1 + v2 + v3;
    for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// MODE 0: all MFMA (8 waves); MODE 1: all VALU
template <int MODE>
__global__ void __launch_bounds__(512) solo(float* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    floatx16 acc = {}, acc2 = {};
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    const bool active = (threadIdx.x >> 8) == 0;  // only waves 0-3: one wave per SIMD, the partner idles
    for (int it = 0; it < iters && active; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc2, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2));
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(v2), "v"(v3));
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(v3), "v"(v0));
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(v0), "v"(v1));
            }
        }
    }
    float s = v0 + v1 + v2 + v3;
    for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// The pre-filter's inner pattern: NM dependent MFMAs, then a min tree over the 16 results, a compare and a
// ballot (the branch is never taken).  PM / PT: priorities while issuing the MFMAs / the tree (-1 = no s_setprio).
template <int THREADS, int NM, int PM, int PT>
__global__ void __launch_bounds__(THREADS) sweep_like(float* out, int iters, float thr) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.f + i * 0.5f); }
    floatx16 c0;
    for (int i = 0; i < 16; ++i) c0[i] = 1000.f + i;
    int hits = 0;
    for (int it = 0; it < iters; ++it) {
        if (PM >= 0) __builtin_amdgcn_s_setprio(PM);
        floatx16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
#pragma unroll
        for (int j = 1; j < NM; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        if (PT >= 0) __builtin_amdgcn_s_setprio(PT);
        float g0 = fminf(fminf(acc[0], acc[1]), acc[2]);
        float g1 = fminf(fminf(acc[3], acc[4]), acc[5]);
        float g2 = fminf(fminf(acc[6], acc[7]), acc[8]);
        float g3 = fminf(fminf(acc[9], acc[10]), acc[11]);
        float g4 = fminf(fminf(fminf(acc[12], acc[13]), acc[14]), acc[15]);
        const float m = fminf(fminf(fminf(g0, g1), fminf(g2, g3)), g4);
        if (__builtin_amdgcn_ballot_w64(m < thr) != 0) { hits += 1; a[0] = (_Float16)((float)a[0] + 1.f); }
    }
    out[blockIdx.x * THREADS + threadIdx.x] = (float)hits;
}

template <typename K, typename... A>
float timed(K kern, int threads, A... args) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<256, threads>>>(args...);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<256, threads>>>(args...);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 20000;
    printf("solo (one wave per SIMD, partner idle): MFMA %.2f ms, VALU %.2f ms  [8 MFMA / 64 v_min3 per iteration]\n",
           timed(solo<0>, 512, out, iters), timed(solo<1>, 512, out, iters));
    printf("role split, MFMA wave + VALU wave per SIMD (perfect overlap = max of the two solo times, none = their sum):\n");
    printf("  prio MFMA 0 / VALU 0 : %.2f ms\n", timed(split_roles<0, 0>, 512, out, iters));
    printf("  prio MFMA 0 / VALU 3 : %.2f ms\n", timed(split_roles<0, 3>, 512, out, iters));
    printf("  prio MFMA 3 / VALU 0 : %.2f ms\n", timed(split_roles<3, 0>, 512, out, iters));
    printf("  prio MFMA 1 / VALU 2 : %.2f ms\n", timed(split_roles<1, 2>, 512, out, iters));
    printf("sweep-like, 2 MFMAs + min tree per iteration, 4 waves per SIMD (1024 threads), %d iterations:\n", iters);
    printf("  no setprio           : %.2f ms\n", timed(sweep_like<1024, 2, -1, -1>, 1024, out, iters, -1.f));
    printf("  MFMA 1 / tree 0      : %.2f ms\n", timed(sweep_like<1024, 2, 1, 0>, 1024, out, iters, -1.f));
    printf("  MFMA 0 / tree 1      : %.2f ms\n", timed(sweep_like<1024, 2, 0, 1>, 1024, out, iters, -1.f));
    printf("  MFMA 0 / tree 3      : %.2f ms\n", timed(sweep_like<1024, 2, 0, 3>, 1024, out, iters, -1.f));
    printf("  MFMA 3 / tree 0      : %.2f ms\n", timed(sweep_like<1024, 2, 3, 0>, 1024, out, iters, -1.f));
    printf("  bare: 4 waves x 2 MFMA x 32 cycles = 256 cycles per iteration per SIMD = %.2f ms at 2.0 GHz\n", 256.0 * iters / 2.0e9 * 1e3);
    printf("sweep-like at 2 waves per SIMD (512 threads) and 1 wave per SIMD (256):\n");
    printf("  2/SIMD no setprio    : %.2f ms\n", timed(sweep_like<512, 2, -1, -1>, 512, out, iters, -1.f));
    printf("  2/SIMD MFMA 0/tree 1 : %.2f ms\n", timed(sweep_like<512, 2, 0, 1>, 512, out, iters, -1.f));
    printf("  1/SIMD no setprio    : %.2f ms\n", timed(sweep_like<256, 2, -1, -1>, 256, out, iters, -1.f));
    printf("sweep-like with 1 MFMA (d <= 16) and 4 MFMAs (d = 64) per tree, 4 waves per SIMD, no setprio: %.2f / %.2f ms\n",
           timed(sweep_like<1024, 1, -1, -1>, 1024, out, iters, -1.f), timed(sweep_like<1024, 4, -1, -1>, 1024, out, iters, -1.f));
    return 0;
}
