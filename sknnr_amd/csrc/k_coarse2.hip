// k_coarse2.hip -- kernel translation unit: the second-generation MFMA pre-filter (coarse2.hip.h), one instance per
// (K-steps, list length, waves per workgroup, rank beyond the list), behind launch.hip.h.  Compiled in two halves
// (-DSKNNR_C2_PART=0: lists of 2 and 6; =1: lists of 8, 12 and 16 at one to four K-steps) so that the halves build in parallel.
#include <cstdio>
#include <cstring>

#include "launch.hip.h"

#ifndef SKNNR_C2_PART
#error "compile with -DSKNNR_C2_PART=0 or 1"
#endif

namespace sknnr {
namespace {

constexpr int kTailWaves = 4;  // the thin-round variant (launch_coarse2_ks in sknnr_hip.hip)

template <int KS, int M, int WAVES, int E>
hipError_t go(const launch::Coarse2Launch& L, hipStream_t st) {
    constexpr int QPB = WAVES * kCoarse2Nqb * 32;
    constexpr size_t sh = 2 * (size_t)tiles_per_stage2(KS) * tile2_bytes(KS) + (size_t)WAVES * queue2_bytes_per_wave();
    static_assert(kRowQuantum % QPB == 0, "query rows are padded to multiples of kRowQuantum");
    static_assert(sh <= 160 * 1024, "LDS budget");
    auto kern = coarse2_kernel<KS, M, WAVES, E>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return e;
    kern<<<dim3((unsigned)(L.rows / QPB)), dim3(WAVES * 64), sh, st>>>(L.rhi, L.rlo, L.n_stages, L.qimg, L.qnc, L.skip_scale, L.n_sentinel,
                                                                   L.cand_val, L.cand_idx, L.pos0, L.qperm, L.qcell, L.cell_stage);
    return hipGetLastError();
}

// the ranks beyond a list that have an instance (coarse2_rank_extra)
template <int KS, int M, int WAVES>
int by_rank(int extra, const launch::Coarse2Launch& L, hipStream_t st, hipError_t* err) {
    if (extra == 0) { *err = go<KS, M, WAVES, 0>(L, st); return 0; }
    if constexpr (M == 16) {
        if (extra == 6) { *err = go<KS, M, WAVES, 6>(L, st); return 0; }
        if (extra == 11) { *err = go<KS, M, WAVES, 11>(L, st); return 0; }
        if (extra == 15) { *err = go<KS, M, WAVES, 15>(L, st); return 0; }
        if (extra == 16) { *err = go<KS, M, WAVES, 16>(L, st); return 0; }
    } else if constexpr (M == 12) {
        if (extra == 10) { *err = go<KS, M, WAVES, 10>(L, st); return 0; }
        if (extra == 12) { *err = go<KS, M, WAVES, 12>(L, st); return 0; }
    } else if constexpr (M == 8) {
        if (extra == 4) { *err = go<KS, M, WAVES, 4>(L, st); return 0; }
        if (extra == 7) { *err = go<KS, M, WAVES, 7>(L, st); return 0; }
        if (extra == 8) { *err = go<KS, M, WAVES, 8>(L, st); return 0; }
    } else if constexpr (M == 6) {
        if (extra == 3) { *err = go<KS, M, WAVES, 3>(L, st); return 0; }
    }
    return launch::kNoInstance;
}

template <int KS, int M>
int by_waves(int waves, int extra, const launch::Coarse2Launch& L, hipStream_t st, hipError_t* err) {
    static_assert(coarse2_supported(KS, M), "no second-generation kernel for this shape");
    if (waves == coarse2_waves(KS, M)) return by_rank<KS, M, coarse2_waves(KS, M)>(extra, L, st, err);
    if (waves == kTailWaves) return by_rank<KS, M, kTailWaves>(extra, L, st, err);
    return launch::kNoInstance;
}

}  // namespace

namespace launch {

#if SKNNR_C2_PART == 0
int coarse2_part1(int ks, int m, int waves, int extra, const Coarse2Launch& L, hipStream_t st, hipError_t* err);  // the other half

static int coarse2_part0(int ks, int m, int waves, int extra, const Coarse2Launch& L, hipStream_t st, hipError_t* err) {
    if (ks == 2 && m == 6) return by_waves<2, 6>(waves, extra, L, st, err);
#ifndef SKNNR_DEV_ONLY_KS2_M6
    if (ks == 1 && m == 2) return by_waves<1, 2>(waves, extra, L, st, err);
    if (ks == 2 && m == 2) return by_waves<2, 2>(waves, extra, L, st, err);
    if (ks == 1 && m == 6) return by_waves<1, 6>(waves, extra, L, st, err);
    if (ks == 3 && m == 6) return by_waves<3, 6>(waves, extra, L, st, err);
    if (ks == 4 && m == 6) return by_waves<4, 6>(waves, extra, L, st, err);
#endif
    return kNoInstance;
}

int coarse2(int ks, int m_list, int waves, int rank_extra, const Coarse2Launch& L, hipStream_t st, hipError_t* err) {
    const int rc = coarse2_part0(ks, m_list, waves, rank_extra, L, st, err);
    return rc == kNoInstance ? coarse2_part1(ks, m_list, waves, rank_extra, L, st, err) : rc;
}

void coarse2_dev_report() {
#ifdef SKNNR_COARSE_TIMERS
    {
        unsigned long long c[8] = {};
        (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(sknnr::coarse_timers), sizeof c);
        static const char* names[8] = {"sweep", "visit_scan", "flush", "stage_barrier", "seeding", "-", "-", "wave_total"};
        for (int i = 0; i < 8; ++i)
            std::fprintf(stderr, "[coarse-time] %-18s %14llu  %5.1f %%\n", names[i], c[i], 100.0 * (double)c[i] / (double)(c[7] ? c[7] : 1));
        std::memset(c, 0, sizeof c);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(sknnr::coarse_timers), c, sizeof c);
    }
#endif
#ifdef SKNNR_COARSE_COUNTERS
    {
        unsigned long long c[16] = {};
        (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(sknnr::coarse_counters), sizeof c);
        static const char* names[16] = {"tile_qblocks", "visits", "group_hits", "member_hits", "hit_lanes", "overflow_inserts",
                                        "flushes", "flush_iters", "flush_iter_lanes", "flush_insert_iters", "flush_insert_lanes",
                                        "visit_lanes", "visits_with_true_hit", "", "", ""};
        for (int i = 0; i < 13; ++i) std::fprintf(stderr, "[coarse] %-22s %llu\n", names[i], c[i]);
        std::memset(c, 0, sizeof c);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(sknnr::coarse_counters), c, sizeof c);
    }
#endif
}
#else
int coarse2_part1(int ks, int m, int waves, int extra, const Coarse2Launch& L, hipStream_t st, hipError_t* err) {
#ifndef SKNNR_DEV_ONLY_KS2_M6
    if (ks == 1 && m == 8) return by_waves<1, 8>(waves, extra, L, st, err);
    if (ks == 2 && m == 8) return by_waves<2, 8>(waves, extra, L, st, err);
    if (ks == 3 && m == 8) return by_waves<3, 8>(waves, extra, L, st, err);
    if (ks == 4 && m == 8) return by_waves<4, 8>(waves, extra, L, st, err);
    if (ks == 3 && m == 12) return by_waves<3, 12>(waves, extra, L, st, err);
    if (ks == 3 && m == 16) return by_waves<3, 16>(waves, extra, L, st, err);
    if (ks == 4 && m == 12) return by_waves<4, 12>(waves, extra, L, st, err);
    if (ks == 4 && m == 16) return by_waves<4, 16>(waves, extra, L, st, err);
    if (ks == 1 && m == 12) return by_waves<1, 12>(waves, extra, L, st, err);
    if (ks == 2 && m == 12) return by_waves<2, 12>(waves, extra, L, st, err);
    if (ks == 1 && m == 16) return by_waves<1, 16>(waves, extra, L, st, err);
    if (ks == 2 && m == 16) return by_waves<2, 16>(waves, extra, L, st, err);
#endif
    return kNoInstance;
}
#endif

}  // namespace launch
}  // namespace sknnr
