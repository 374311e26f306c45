// Microbenchmark (development aid): do MFMAs of one wave overlap with VALU work of another wave on
// the same SIMD?  Workgroups of 8 waves (2 per SIMD), one workgroup per CU.  mode 0: all waves issue
// dependent MFMA chains; mode 1: all waves issue v_min3 chains; mode 2: waves 0-3 MFMA, waves 4-7
// VALU (SIMD partners do different work); mode 3: every wave alternates MFMA and VALU bursts.
// build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-result scripts/microbench/mfma_valu_overlap.hip -o scripts/microbench/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int MODE, int THREADS = 512>
__global__ void __launch_bounds__(THREADS) k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    floatx16 acc = {}, acc2 = {};
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    const bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && (wave & 4) == 0);
    const bool do_valu = MODE == 1 || MODE == 3 || (MODE == 2 && (wave & 4) != 0);
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // two independent accumulation chains
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc2, 0, 0, 0);
            }
        }
        if (do_valu) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {  // 64 independent-ish VALU ops: 4 chains
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2));
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(v2), "v"(v3));
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(v3), "v"(v0));
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(v0), "v"(v1));
            }
        }
    }
    float s = v0 + v1 + v2 + v3;
    for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int MODE, int THREADS = 512>
float run(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, THREADS><<<256, THREADS>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, THREADS><<<256, THREADS>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 20000;
    const float t0 = run<0>(out, iters), t1 = run<1>(out, iters), t2 = run<2>(out, iters), t3 = run<3>(out, iters);
    // per SIMD: mode 0: 2 waves x 8 MFMA x iters; mode 1: 2 waves x 64 VALU x iters
    printf("all-MFMA   %.2f ms  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", t0, t0 * 1e-3 * 2.4e9 / (2.0 * 8 * iters));
    printf("all-VALU   %.2f ms  (%.1f cycles per VALU per SIMD at 2.4 GHz)\n", t1, t1 * 1e-3 * 2.4e9 / (2.0 * 64 * iters));
    printf("partners   %.2f ms  (one wave MFMA, its SIMD partner VALU; half the work of each pure run: "
           "perfect overlap = max(%.2f, %.2f), none = %.2f)\n", t2, t0 / 2, t1 / 2, t0 / 2 + t1 / 2);
    printf("alternate  %.2f ms  (every wave does both: perfect overlap = max(%.2f, %.2f), none = %.2f)\n", t3, t0, t1, t0 + t1);
    const float u0 = run<0, 1024>(out, iters), u1 = run<1, 1024>(out, iters), u2 = run<2, 1024>(out, iters), u3 = run<3, 1024>(out, iters);
    printf("4 waves/SIMD: all-MFMA %.2f  all-VALU %.2f  partners(2 MFMA + 2 VALU waves per SIMD) %.2f [overlap %.2f, none %.2f]  "
           "alternate %.2f [overlap %.2f, none %.2f]\n", u0, u1, u2, u0 / 2 > u1 / 2 ? u0 / 2 : u1 / 2, u0 / 2 + u1 / 2, u3,
           u0 > u1 ? u0 : u1, u0 + u1);
    const float w0 = run<0, 256>(out, iters), w1 = run<1, 256>(out, iters), w3 = run<3, 256>(out, iters);
    printf("1 wave/SIMD : all-MFMA %.2f  all-VALU %.2f  alternate %.2f [none %.2f]\n", w0, w1, w3, w0 + w1);
    return 0;
}
