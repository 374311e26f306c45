// Microbenchmark (development aid): how does v_mfma_f32_32x32x16_f16 accumulate?
//
// The split-contraction error budget of the pre-filter needs a model of the
// instruction's internal arithmetic, which the ISA guide does not give:
//   (1) fixed probes: is the 16-term product sum added to C with ONE rounding (wide internal
//       accumulation), per k-group, or term by term?  round-to-nearest-even or truncation?
//   (2) a randomised search for the worst error of one instruction, in units of
//       2^-24 (|C| + sum |a_k b_k|) and of ulp(result), over operand laws that stress it
//       (same-sign products, C of either sign and of much larger / smaller magnitude, values at
//       f16 rounding midpoints).
// build: hipcc --offload-arch=gfx950 -O2 scripts/microbench/mfma_f16_numerics.hip -o scripts/microbench/mfma_f16_numerics
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// One wave = one 32x32x16 product.  a: [32 rows][16 k] f16 (row-major), b: [16 k][32 cols], c/d: [32][32] f32.
__global__ void __launch_bounds__(64) one_mfma(const _Float16* a, const _Float16* b, const float* c, float* d) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const size_t t = blockIdx.x;
    a += t * 512; b += t * 512; c += t * 1024; d += t * 1024;
    half8 fa, fb;
    for (int j = 0; j < 8; ++j) {
        fa[j] = a[r * 16 + 8 * h + j];
        fb[j] = b[(8 * h + j) * 32 + r];
    }
    floatx16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = c[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) d[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

static std::vector<float> run(const std::vector<_Float16>& a, const std::vector<_Float16>& b, const std::vector<float>& c) {
    const size_t tiles = c.size() / 1024;
    _Float16 *da, *db; float *dc, *dd;
    hipMalloc(&da, a.size() * 2); hipMalloc(&db, b.size() * 2); hipMalloc(&dc, c.size() * 4); hipMalloc(&dd, c.size() * 4);
    hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice);
    one_mfma<<<(unsigned)tiles, 64>>>(da, db, dc, dd);
    std::vector<float> d(c.size());
    hipMemcpy(d.data(), dd, c.size() * 4, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dc); hipFree(dd);
    return d;
}

// Probe: row 0 of A holds av[k], column 0 of B holds bv[k], C[0][0] = c0.  Returns D[0][0].
static float probe(const double (&av)[16], const double (&bv)[16], float c0) {
    std::vector<_Float16> a(512, (_Float16)0), b(512, (_Float16)0);
    std::vector<float> c(1024, 0.f);
    for (int k = 0; k < 16; ++k) { a[k] = (_Float16)av[k]; b[k * 32] = (_Float16)bv[k]; }
    c[0] = c0;
    return run(a, b, c)[0];
}

int main() {
    const float two24 = 16777216.f;
    printf("== fixed probes (C = 2^24 has ulp 2; every listed product is exact in f32)\n");
    {
        double av[16], bv[16];
        for (int n : {1, 2, 3, 4, 8, 16}) {
            for (int k = 0; k < 16; ++k) { av[k] = k < n ? 1.0 : 0.0; bv[k] = 1.0; }
            printf("C=2^24 + %2d products of 1.0 in k=0..%2d : D - 2^24 = %g   (exact %d; term-by-term RNE would give 0)\n", n, n - 1,
                   (double)probe(av, bv, two24) - two24, n);
        }
        for (int stride : {2, 4, 8}) {
            for (int k = 0; k < 16; ++k) { av[k] = (k % stride == 0) ? 1.0 : 0.0; bv[k] = 1.0; }
            printf("C=2^24 + products of 1.0 at every %d-th k (%d terms): D - 2^24 = %g\n", stride, 16 / stride,
                   (double)probe(av, bv, two24) - two24);
        }
        for (int k = 0; k < 16; ++k) { av[k] = 0; bv[k] = 1; }
        av[0] = 1.5;
        printf("C=2^24 + one product 1.5: D - 2^24 = %g   (RNE 2, truncation 0)\n", (double)probe(av, bv, two24) - two24);
        av[0] = 3.0;
        printf("C=2^24 + one product 3.0: D - 2^24 = %g   (RNE 4, truncation 2)\n", (double)probe(av, bv, two24) - two24);
        av[0] = 1.0;
        printf("C=2^24 + one product 1.0: D - 2^24 = %g   (RNE tie-to-even 0, round-half-up 2)\n", (double)probe(av, bv, two24) - two24);
        printf("C=-2^24 + one product 1.5: D + 2^24 = %g  (RNE 2, truncation toward zero 2, toward -inf 0)\n",
               (double)probe(av, bv, -two24) + two24 + 0.5);
        // wide internal accumulation: 2^12 - 2^12 + 2^-13 in different k positions
        for (int pos : {1, 2, 4, 8, 15}) {
            for (int k = 0; k < 16; ++k) { av[k] = 0; bv[k] = 1; }
            av[0] = 64.0; bv[0] = 64.0;          // +2^12
            av[pos] = -64.0; bv[pos] = 64.0;     // -2^12
            const int p3 = pos == 15 ? 7 : pos + 1;
            av[p3] = 0.0078125; bv[p3] = 0.015625;  // 2^-7 * 2^-6 = 2^-13
            printf("C=0, 2^12 (k=0) - 2^12 (k=%d) + 2^-13 (k=%d): D = %g   (exact 2^-13 = %g)\n", pos, p3, (double)probe(av, bv, 0.f),
                   std::ldexp(1.0, -13));
        }
        // where does the group's alignment cut?  2^12 - 2^12 + 2^(12-j) inside one group of 8
        for (int j = 18; j <= 27; ++j) {
            for (int k = 0; k < 16; ++k) { av[k] = 0; bv[k] = 1; }
            av[0] = 64.0; bv[0] = 64.0;
            av[1] = -64.0; bv[1] = 64.0;
            av[2] = std::ldexp(1.0, 6 - j / 2); bv[2] = std::ldexp(1.0, 6 - (j - j / 2));
            printf("C=0, 2^12 - 2^12 + 2^(12-%d) in one group: D = %g x 2^(12-%d)\n", j, (double)probe(av, bv, 0.f) / std::ldexp(1.0, 12 - j), j);
        }
        for (int j = 22; j <= 26; ++j) {  // truncation or rounding at the cut: small term 3 x 2^(12-j)
            for (int k = 0; k < 16; ++k) { av[k] = 0; bv[k] = 1; }
            av[0] = 64.0; bv[0] = 64.0;
            av[1] = -64.0; bv[1] = 64.0;
            av[2] = 3.0 * std::ldexp(1.0, 6 - j / 2); bv[2] = std::ldexp(1.0, 6 - (j - j / 2));
            printf("C=0, 2^12 - 2^12 + 3 x 2^(12-%d) in one group: D = %g x 2^(12-%d)\n", j, (double)probe(av, bv, 0.f) / std::ldexp(1.0, 12 - j), j);
            av[2] = -av[2];
            printf("C=0, 2^12 - 2^12 - 3 x 2^(12-%d) in one group: D = %g x 2^(12-%d)\n", j, (double)probe(av, bv, 0.f) / std::ldexp(1.0, 12 - j), j);
        }
        // the same small term in the OTHER group (k >= 8), and with the big pair split over the groups
        for (int j : {20, 24, 25, 26, 30}) {
            for (int k = 0; k < 16; ++k) { av[k] = 0; bv[k] = 1; }
            av[0] = 64.0; bv[0] = 64.0;
            av[1] = -64.0; bv[1] = 64.0;
            av[9] = std::ldexp(1.0, 6 - j / 2); bv[9] = std::ldexp(1.0, 6 - (j - j / 2));
            printf("C=0, (2^12 - 2^12) in group 0, 2^(12-%d) in group 1: D = %g x 2^(12-%d)\n", j, (double)probe(av, bv, 0.f) / std::ldexp(1.0, 12 - j), j);
            av[1] = 0; av[8] = -64.0; bv[8] = 64.0;
            printf("C=0, 2^12 in group 0, -2^12 + 2^(12-%d) in group 1: D = %g x 2^(12-%d)\n", j, (double)probe(av, bv, 0.f) / std::ldexp(1.0, 12 - j), j);
        }
        // does C take part in the alignment?  C = 2^12, products -2^12 and 2^(12-j)
        for (int j : {20, 24, 25, 26, 30}) {
            for (int k = 0; k < 16; ++k) { av[k] = 0; bv[k] = 1; }
            av[0] = -64.0; bv[0] = 64.0;
            av[1] = std::ldexp(1.0, 6 - j / 2); bv[1] = std::ldexp(1.0, 6 - (j - j / 2));
            printf("C=2^12, products -2^12 + 2^(12-%d): D = %g x 2^(12-%d)\n", j, (double)probe(av, bv, 4096.f) / std::ldexp(1.0, 12 - j), j);
        }
        // product sum vs C alignment: C = 1, 16 products of 2^-25 each (sum 2^-21, above half an ulp of 1 = 2^-24)
        for (int k = 0; k < 16; ++k) { av[k] = std::ldexp(1.0, -12); bv[k] = std::ldexp(1.0, -13); }
        printf("C=1 + 16 products of 2^-25: D - 1 = %g   (exact 2^-21 = %g; term-by-term gives 0)\n", (double)probe(av, bv, 1.f) - 1.0,
               std::ldexp(1.0, -21));
        // subnormal f16 operands
        for (int k = 0; k < 16; ++k) { av[k] = 0; bv[k] = 0; }
        av[0] = std::ldexp(1.0, -24); bv[0] = 1.0;
        printf("C=0 + (f16 subnormal 2^-24) * 1: D = %g   (exact %g)\n", (double)probe(av, bv, 0.f), std::ldexp(1.0, -24));
    }

    printf("== randomised worst case of ONE instruction\n");
    std::mt19937_64 rng(12345);
    const int tiles = 4096;
    struct Law { const char* name; int sign_mode; double c_scale; bool midpoint; };
    const Law laws[] = {
        {"random signs, |a|<=256 |b|<=128, C ~ sum", 0, 1.0, false},
        {"all products positive, C positive ~ sum", 1, 1.0, false},
        {"all products positive, C negative ~ -sum/2", 1, -0.5, false},
        {"all products positive, C = 0", 1, 0.0, false},
        {"all positive, C 1000x the sum", 1, 1000.0, false},
        {"all positive, C 1/1000 of the sum", 1, 0.001, false},
        {"products positive, operands with full 11-bit mantissas (odd last bit)", 1, 1.0, true},
        {"random signs, operands with full mantissas, C negative", 0, -1.0, true},
    };
    for (const Law& law : laws) {
        std::vector<_Float16> a((size_t)tiles * 512), b((size_t)tiles * 512);
        std::vector<float> c((size_t)tiles * 1024);
        std::uniform_real_distribution<double> ua(0.0, 256.0), ub(0.0, 128.0), u01(0.0, 1.0);
        auto full = [&](double v) {  // force the last mantissa bit of the f16 to 1
            _Float16 h = (_Float16)v;
            uint16_t bits; std::memcpy(&bits, &h, 2);
            bits |= 1; std::memcpy(&h, &bits, 2);
            return h;
        };
        for (size_t i = 0; i < a.size(); ++i) {
            double va = ua(rng) * (u01(rng) < 0.3 ? std::ldexp(1.0, -(int)(u01(rng) * 12)) : 1.0);
            double vb = ub(rng) * (u01(rng) < 0.3 ? std::ldexp(1.0, -(int)(u01(rng) * 12)) : 1.0);
            if (law.sign_mode == 0) { if (u01(rng) < 0.5) va = -va; if (u01(rng) < 0.5) vb = -vb; }
            a[i] = law.midpoint ? full(va) : (_Float16)va;
            b[i] = law.midpoint ? full(vb) : (_Float16)vb;
        }
        // C: scale times the exact sum of |products| of the element
        std::vector<double> exact((size_t)tiles * 1024), sabs((size_t)tiles * 1024);
        for (int t = 0; t < tiles; ++t)
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double s = 0, sa = 0;
                    for (int k = 0; k < 16; ++k) {
                        const double p = (double)a[(size_t)t * 512 + i * 16 + k] * (double)b[(size_t)t * 512 + k * 32 + j];
                        s += p; sa += std::fabs(p);
                    }
                    const float cv = (float)(law.c_scale * sa * (0.5 + u01(rng)));
                    c[(size_t)t * 1024 + i * 32 + j] = cv;
                    exact[(size_t)t * 1024 + i * 32 + j] = s + (double)cv;   // exact in double: 22-bit products, 16 terms
                    sabs[(size_t)t * 1024 + i * 32 + j] = sa + std::fabs((double)cv);
                }
        const std::vector<float> d = run(a, b, c);
        double worst_units = 0, worst_ulps = 0;
        for (size_t i = 0; i < d.size(); ++i) {
            const double err = std::fabs((double)d[i] - exact[i]);
            worst_units = std::max(worst_units, err / (std::ldexp(1.0, -24) * sabs[i]));
            int e; (void)std::frexp(exact[i] == 0 ? 1e-300 : exact[i], &e);
            const double ulp = std::ldexp(1.0, e - 24);
            if (exact[i] != 0) worst_ulps = std::max(worst_ulps, err / ulp);
        }
        printf("%-72s worst error %.3f x 2^-24 (|C| + sum|ab|), %.3f ulp(result)   [%zu elements]\n", law.name, worst_units, worst_ulps,
               d.size());
    }
    return 0;
}
