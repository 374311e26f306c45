// Microbenchmark (development aid): what v_mfma_f32_32x32x16_f16 rate does the chip SUSTAIN (power / clock) on realistic
// operand data, alone and with the pre-filter's VALU work beside it?  4 waves per SIMD, one workgroup of 1024 threads per CU,
// every CU busy; 2 dependent MFMAs per step.  Operands: zeros, or random f16 of the pre-filter's magnitudes.
// Reported: wall ms, MFMA/s, TFLOP/s (dense), the clock held (s_memtime / s_memrealtime).
// build: hipcc --offload-arch=gfx950 -O3 scripts/microbench/mfma_power.hip -o scripts/microbench/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NVALU>
__global__ void __launch_bounds__(1024) k(const half8* __restrict__ ops, float* out, int iters, unsigned long long* clk) {
    const half8 a0 = ops[threadIdx.x * 4 + 0], a1 = ops[threadIdx.x * 4 + 1], b0 = ops[threadIdx.x * 4 + 2], b1 = ops[threadIdx.x * 4 + 3];
    floatx16 acc;
    float v[4] = {1.f, 2.f, 3.f, (float)threadIdx.x};
    float sum = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a0), "v"(b0));
#pragma unroll
        for (int j = 0; j < NVALU / 2; ++j) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[j & 3]) : "v"(v[(j + 1) & 3]), "v"(v[(j + 2) & 3]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a1), "v"(b1));
#pragma unroll
        for (int j = NVALU / 2; j < NVALU; ++j) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[j & 3]) : "v"(v[(j + 1) & 3]), "v"(v[(j + 2) & 3]));
    }
    asm volatile("s_nop 7\n\ts_nop 7");
    for (int i = 0; i < 16; ++i) sum += acc[i];
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    out[blockIdx.x * 1024 + threadIdx.x] = sum + v[0] + v[1] + v[2] + v[3];
}

// the same FLOPs per step from four v_mfma_f32_16x16x32_f16 (two independent accumulators of four registers)
typedef float floatx4v __attribute__((ext_vector_type(4)));
template <int NVALU>
__global__ void __launch_bounds__(1024) k16(const half8* __restrict__ ops, float* out, int iters, unsigned long long* clk) {
    const half8 a0 = ops[threadIdx.x * 4 + 0], a1 = ops[threadIdx.x * 4 + 1], b0 = ops[threadIdx.x * 4 + 2], b1 = ops[threadIdx.x * 4 + 3];
    floatx4v acc, acc2;
    float v[4] = {1.f, 2.f, 3.f, (float)threadIdx.x};
    float sum = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a0), "v"(b0));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc2) : "v"(a1), "v"(b0));
#pragma unroll
        for (int j = 0; j < NVALU / 2; ++j) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[j & 3]) : "v"(v[(j + 1) & 3]), "v"(v[(j + 2) & 3]));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a0), "v"(b1));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc2) : "v"(a1), "v"(b1));
#pragma unroll
        for (int j = NVALU / 2; j < NVALU; ++j) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[j & 3]) : "v"(v[(j + 1) & 3]), "v"(v[(j + 2) & 3]));
    }
    asm volatile("s_nop 7\n\ts_nop 7");
    for (int i = 0; i < 4; ++i) sum += acc[i] + acc2[i];
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    out[blockIdx.x * 1024 + threadIdx.x] = sum + v[0] + v[1] + v[2] + v[3];
}

template <int NVALU, bool SMALL = false>
void run(const char* name, const half8* ops, float* out, unsigned long long* clk, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto kern = SMALL ? k16<NVALU> : k<NVALU>;
    for (int w = 0; w < 3; ++w) kern<<<256, 1024>>>(ops, out, iters, clk);  // warm up: let the clock settle
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kern<<<256, 1024>>>(ops, out, iters, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2];
    (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    const double n_mfma = 256.0 * 16 * 2.0 * iters;
    printf("%-44s %8.3f ms  %6.2f MFMA/ns  %7.1f TFLOP/s  (%.1f ns per MFMA per SIMD; s_memtime/s_memrealtime = %.2f GHz)\n", name, ms,
           n_mfma / (ms * 1e6), n_mfma * 32768.0 / (ms * 1e-3) / 1e12, ms * 1e6 / (4.0 * 2.0 * iters), (double)c[0] / ((double)c[1] * 10.0));
}

int main() {
    const int iters = 200000;
    std::vector<_Float16> h(1024 * 32);
    half8* ops; float* out; unsigned long long* clk;
    (void)hipMalloc(&ops, h.size() * 2); (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&clk, 64);
    for (auto& x : h) x = (_Float16)0.f;
    (void)hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<0>("zeros, MFMA only", ops, out, clk, iters);
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((rand() % 2001) - 1000) * 0.128f);  // |x| <= 128, full mantissas
    (void)hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<0>("random operands, MFMA only", ops, out, clk, iters);
    run<4>("random operands, 2 MFMA + 4 v_min3", ops, out, clk, iters);
    run<8>("random operands, 2 MFMA + 8 v_min3", ops, out, clk, iters);
    run<12>("random operands, 2 MFMA + 12 v_min3", ops, out, clk, iters);
    run<16>("random operands, 2 MFMA + 16 v_min3", ops, out, clk, iters);
    printf("-- the same FLOPs per step from four v_mfma_f32_16x16x32_f16 (printed per 32x32x16-equivalent)\n");
    run<0, true>("random operands, 16x16x32 only", ops, out, clk, iters);
    run<8, true>("random operands, 16x16x32 + 8 v_min3", ops, out, clk, iters);
    run<12, true>("random operands, 16x16x32 + 12 v_min3", ops, out, clk, iters);
    return 0;
}
