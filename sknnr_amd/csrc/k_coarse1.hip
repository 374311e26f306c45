// k_coarse1.hip -- kernel translation unit: the first-generation MFMA pre-filter (coarse.hip.h: more than 64 features,
// lists of 32, small reference sets) and its diagnostic twin, behind launch.hip.h.
#include <cstdio>
#include <cstring>

#include "launch.hip.h"

namespace sknnr {
namespace {

template <int KS, int M>
hipError_t coarse1_ks_m(const launch::Coarse1Launch& L, hipStream_t st) {
    constexpr int NQB = coarse_nqb(KS, M);
    constexpr int WAVES = coarse_waves(KS, M);
    constexpr int QPB = WAVES * NQB * 32;
    constexpr int TPS = tiles_per_stage(KS);
    constexpr size_t sh = 2 * (size_t)TPS * tile_bytes(KS) + (size_t)WAVES * queue_bytes_per_wave(NQB, M);
    static_assert(kRowQuantum % QPB == 0, "query rows are padded to multiples of kRowQuantum");
    static_assert(sh <= 160 * 1024, "LDS budget");
    auto kern = coarse_kernel<KS, M>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return e;
    kern<<<dim3((unsigned)(L.nq_pad / QPB)), dim3(WAVES * 64), sh, st>>>(L.rimg, L.n_stages, L.qimg, L.qnc, L.skip_scale, L.n_sentinel,
                                                                      L.cand_val, L.cand_idx);
    return hipGetLastError();
}

template <int M>
int coarse1_m(int ks, const launch::Coarse1Launch& L, hipStream_t st, hipError_t* err) {
    switch (ks) {
#ifdef SKNNR_DEV_ONLY_KS2_M6  // development builds: two K-steps only (fast compile)
        case 2: *err = coarse1_ks_m<2, M>(L, st); return 0;
#else
        case 1: *err = coarse1_ks_m<1, M>(L, st); return 0;
        case 2: *err = coarse1_ks_m<2, M>(L, st); return 0;
        case 3: *err = coarse1_ks_m<3, M>(L, st); return 0;
        case 4: *err = coarse1_ks_m<4, M>(L, st); return 0;
        case 5: *err = coarse1_ks_m<5, M>(L, st); return 0;
        case 6: *err = coarse1_ks_m<6, M>(L, st); return 0;
        case 7: *err = coarse1_ks_m<7, M>(L, st); return 0;
        case 8: *err = coarse1_ks_m<8, M>(L, st); return 0;
#endif
    }
    return launch::kNoInstance;
}

template <int KS>
void matrix_ks(const char* rimg, const int* perm, const uint4* qimg, int n_ref, long nq, long n_tiles, long nqb, float* out) {
    coarse_matrix_kernel<KS><<<dim3((unsigned)n_tiles, (unsigned)nqb), dim3(64)>>>(rimg, perm, qimg, n_ref, (int)nq, out);
}

}  // namespace

namespace launch {

int coarse1(int ks, int m_list, const Coarse1Launch& L, hipStream_t st, hipError_t* err) {
    switch (m_list) {
        case 2: return coarse1_m<2>(ks, L, st, err);
        case 6: return coarse1_m<6>(ks, L, st, err);
        case 8: return coarse1_m<8>(ks, L, st, err);
        case 16: return coarse1_m<16>(ks, L, st, err);
        case 32: return coarse1_m<32>(ks, L, st, err);
    }
    return kNoInstance;
}

hipError_t coarse_matrix(int ks, const char* rimg, const int* perm, const uint4* qimg, int n_ref, long nq, long n_tiles, long nqb,
                         float* out) {
    switch (ks) {
        case 1: matrix_ks<1>(rimg, perm, qimg, n_ref, nq, n_tiles, nqb, out); break;
        case 2: matrix_ks<2>(rimg, perm, qimg, n_ref, nq, n_tiles, nqb, out); break;
        case 3: matrix_ks<3>(rimg, perm, qimg, n_ref, nq, n_tiles, nqb, out); break;
        case 4: matrix_ks<4>(rimg, perm, qimg, n_ref, nq, n_tiles, nqb, out); break;
        case 5: matrix_ks<5>(rimg, perm, qimg, n_ref, nq, n_tiles, nqb, out); break;
        case 6: matrix_ks<6>(rimg, perm, qimg, n_ref, nq, n_tiles, nqb, out); break;
        case 7: matrix_ks<7>(rimg, perm, qimg, n_ref, nq, n_tiles, nqb, out); break;
        case 8: matrix_ks<8>(rimg, perm, qimg, n_ref, nq, n_tiles, nqb, out); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

void coarse1_dev_report() {
#ifdef SKNNR_COARSE_TIMERS
    {
        unsigned long long c[8] = {};
        (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(sknnr::coarse_timers), sizeof c);
        static const char* names[8] = {"operands_wait", "main_no_visit", "main_then_visit", "correct_and_scan", "flush", "loop_overhead", "barrier", "wave_total"};
        if (c[7])
            for (int i = 0; i < 8; ++i)
                std::fprintf(stderr, "[coarse1-time] %-18s %14llu  %5.1f %%\n", names[i], c[i], 100.0 * (double)c[i] / (double)c[7]);
        std::memset(c, 0, sizeof c);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(sknnr::coarse_timers), c, sizeof c);
    }
#endif
}

}  // namespace launch
}  // namespace sknnr
