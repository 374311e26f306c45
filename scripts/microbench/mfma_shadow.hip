// Microbenchmark (development aid): can a wave's OWN VALU work run in the shadow of its own MFMAs?
// (mfma_valu_overlap / mfma_valu_prio: VALU of ANOTHER wave of the SIMD does not.)
//   serial    : per iteration 2 dependent MFMAs, then the skip test's min tree on THEIR result  (the shipped sweep)
//   pipelined : the 2 MFMAs of iteration i+1 are issued first, then the tree of iteration i's result
//               (two accumulator sets) -- the tree has no dependence on the MFMAs in flight
//   fillers   : 2 MFMAs + n independent VALU (v_min_f32 / v_min3_f32) per iteration, n = 0..12
// at 1, 2 and 4 waves per SIMD.  Reported: cycles per iteration per SIMD at the measured clock
// (s_memtime / s_memrealtime), and ms.
// build: hipcc --offload-arch=gfx950 -O3 scripts/microbench/mfma_shadow.hip -o scripts/microbench/mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float tree(const floatx16& acc) {
    float g0, g1, g2, g3, g4, m;
    const float t0 = __builtin_canonicalizef(acc[0]);
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(g0) : "v"(t0), "v"(acc[1]), "v"(acc[2]));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(g1) : "v"(acc[3]), "v"(acc[4]), "v"(acc[5]), "v"(t0));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(g2) : "v"(acc[6]), "v"(acc[7]), "v"(acc[8]), "v"(t0));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(g3) : "v"(acc[9]), "v"(acc[10]), "v"(acc[11]), "v"(t0));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(g4) : "v"(acc[12]), "v"(acc[13]), "v"(acc[14]), "v"(t0));
    asm("v_min_f32 %0, %1, %2" : "=v"(g4) : "v"(g4), "v"(acc[15]), "v"(t0));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(g0), "v"(g1), "v"(g2));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(g3), "v"(g4));
    return m;
}

template <int THREADS, int MODE>
__global__ void __launch_bounds__(THREADS) sweep(float* out, int iters, float thr, unsigned long long* clk) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.f + i * 0.5f); }
    floatx16 c0;
    for (int i = 0; i < 16; ++i) c0[i] = 1000.f + i;
    int hits = 0;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
            floatx16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            const float m = tree(acc);
            if (__builtin_amdgcn_ballot_w64(m < thr) != 0) { hits += 1; a[0] = (_Float16)((float)a[0] + 1.f); }
        }
    } else if (MODE == 1) {
        // ping-pong between two accumulator sets (unrolled by two: no register copies)
        floatx16 p = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        p = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, p, 0, 0, 0);
        for (int it = 0; it < iters; it += 2) {
            floatx16 q = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            q = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, q, 0, 0, 0);
            const float m = tree(p);
            if (__builtin_amdgcn_ballot_w64(m < thr) != 0) { hits += 1; a[0] = (_Float16)((float)a[0] + 1.f); }
            p = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            p = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, p, 0, 0, 0);
            const float m2 = tree(q);
            if (__builtin_amdgcn_ballot_w64(m2 < thr) != 0) { hits += 1; a[0] = (_Float16)((float)a[0] + 1.f); }
        }
        hits += (int)p[3];
    } else {
        // as MODE 1, with the tree of the previous result placed BETWEEN the two MFMAs of the next one
        floatx16 p = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        p = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, p, 0, 0, 0);
        for (int it = 0; it < iters; it += 2) {
            floatx16 q = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            const float m = tree(p);
            q = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, q, 0, 0, 0);
            if (__builtin_amdgcn_ballot_w64(m < thr) != 0) { hits += 1; a[0] = (_Float16)((float)a[0] + 1.f); }
            p = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            const float m2 = tree(q);
            p = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, p, 0, 0, 0);
            if (__builtin_amdgcn_ballot_w64(m2 < thr) != 0) { hits += 1; a[0] = (_Float16)((float)a[0] + 1.f); }
        }
        hits += (int)p[3];
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    out[blockIdx.x * THREADS + threadIdx.x] = (float)hits;
}

// 2 MFMAs + N independent VALU per iteration; OP 0: v_min_f32 (2 sources), OP 1: v_min3_f32 (3 sources)
template <int THREADS, int N, int OP>
__global__ void __launch_bounds__(THREADS) fillers(float* out, int iters, unsigned long long* clk) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.f + i * 0.5f); }
    floatx16 acc = {};
    float v[4] = {(float)threadIdx.x, 1.f, 2.f, 3.f};
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < N / 2; ++j) {
            if (OP == 0) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[j & 3]) : "v"(v[(j + 1) & 3]));
            else asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[j & 3]) : "v"(v[(j + 1) & 3]), "v"(v[(j + 2) & 3]));
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int j = N / 2; j < N; ++j) {
            if (OP == 0) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[j & 3]) : "v"(v[(j + 1) & 3]));
            else asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[j & 3]) : "v"(v[(j + 1) & 3]), "v"(v[(j + 2) & 3]));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    float s = v[0] + v[1] + v[2] + v[3];
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

static float* g_out;
static unsigned long long* g_clk;
template <typename K, typename... A>
void report(const char* name, int waves_per_simd, K kern, int threads, int iters, A... args) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<<<256, threads>>>(g_out, 200, args..., g_clk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kern<<<256, threads>>>(g_out, iters, args..., g_clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2];
    (void)hipMemcpy(c, g_clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)c[0] / ((double)c[1] * 10.0);  // s_memrealtime ticks at 100 MHz
    printf("%-44s %d waves/SIMD: %7.3f ms, %6.1f cycles per iteration per wave, %6.1f per SIMD-iteration-of-all-waves, clock %.2f GHz\n",
           name, waves_per_simd, ms, (double)c[0] / iters, (double)c[0] / iters, ghz);
}

int main() {
    (void)hipMalloc(&g_out, 256 * 1024 * 4);
    (void)hipMalloc(&g_clk, 64);
    const int iters = 20000;
    printf("cycles per iteration are wave cycles of wave 0 (all waves of a SIMD run the same loop concurrently):\n"
           "divide by the waves per SIMD for the SIMD's throughput cost of one wave-iteration\n");
    report("serial   (2 MFMA -> tree of their result)", 1, sweep<256, 0>, 256, iters, -1.f);
    report("pipelined (2 MFMA of i+1, tree of i)", 1, sweep<256, 1>, 256, iters, -1.f);
    report("serial", 2, sweep<512, 0>, 512, iters, -1.f);
    report("pipelined", 2, sweep<512, 1>, 512, iters, -1.f);
    report("serial", 4, sweep<1024, 0>, 1024, iters, -1.f);
    report("pipelined", 4, sweep<1024, 1>, 1024, iters, -1.f);
    report("pipelined, tree between the MFMAs", 1, sweep<256, 2>, 256, iters, -1.f);
    report("pipelined, tree between the MFMAs", 2, sweep<512, 2>, 512, iters, -1.f);
    report("pipelined, tree between the MFMAs", 4, sweep<1024, 2>, 1024, iters, -1.f);
#define FILL(N)                                                                         \
    report("2 MFMA + " #N " v_min_f32 ", 1, fillers<256, N, 0>, 256, iters);             \
    report("2 MFMA + " #N " v_min3_f32", 1, fillers<256, N, 1>, 256, iters);             \
    report("2 MFMA + " #N " v_min_f32 ", 4, fillers<1024, N, 0>, 1024, iters);           \
    report("2 MFMA + " #N " v_min3_f32", 4, fillers<1024, N, 1>, 1024, iters);
    FILL(0) FILL(4) FILL(8) FILL(12) FILL(16)
    return 0;
}
