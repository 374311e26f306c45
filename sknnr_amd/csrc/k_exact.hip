// k_exact.hip -- kernel translation unit: query preparation, bucketing, finaliser, exact scan / merge, predict and the
// small gather kernels (exact.hip.h, bucket.hip.h), behind the launchers of launch.hip.h.
#define SKNNR_KERNELS_EXACT 1  // this unit defines the non-template kernels of exact.hip.h and bucket.hip.h
#include <cstdint>
#include <cstdlib>

#include "launch.hip.h"

namespace sknnr {
namespace {
__global__ void add_counter_kernel(const int* __restrict__ cnt, long long* __restrict__ total) { *total += *cnt; }

template <int BT>
hipError_t prep_lds_bt(const PrepArgs& a, hipStream_t st) {
    const size_t sh = (size_t)BT * (a.d_in | 1) * 8;
    hipError_t e = hipFuncSetAttribute((const void*)prep_queries_kernel<BT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return e;
    prep_queries_kernel<BT><<<dim3((unsigned)(a.nq_pad / BT)), dim3(BT), sh, st>>>(a);
    return hipGetLastError();
}

template <int FORMULA>
hipError_t exact_scan_f(bool chunked, const ScanArgs& a, long blocks, size_t sh, hipStream_t st) {
    auto kern = chunked ? exact_scan_kernel<FORMULA, true> : exact_scan_kernel<FORMULA, false>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return e;
    kern<<<dim3((unsigned)blocks), dim3(kScanWaves * 64), sh, st>>>(a);
    return hipGetLastError();
}
}  // namespace

namespace launch {

hipError_t row_norms(const double* x, long n, int d, double* out, hipStream_t st) {
    row_norms_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(x, n, d, out);
    return hipGetLastError();
}

template <int KS>
hipError_t prep_direct_ks(PrepArgs a, hipStream_t st) {
    a.lds_wave_bytes = prep_direct_wave_lds(KS, a.xt != nullptr, a.qimg != nullptr);
    const size_t sh = (size_t)4 * a.lds_wave_bytes;
    hipError_t e = hipFuncSetAttribute((const void*)prep_queries_direct_kernel<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return e;
    prep_queries_direct_kernel<KS><<<dim3((unsigned)(a.nq_pad / 256)), dim3(256), sh, st>>>(a);
    return hipGetLastError();
}

hipError_t prep_direct(const PrepArgs& a, hipStream_t st) {
    switch (a.ks) {
        case 1: return prep_direct_ks<1>(a, st);
        case 2: return prep_direct_ks<2>(a, st);
        case 3: return prep_direct_ks<3>(a, st);
        case 4: return prep_direct_ks<4>(a, st);
    }
    return hipErrorInvalidValue;
}

hipError_t prep_lds(int rows_per_block, const PrepArgs& a, hipStream_t st) {
    switch (rows_per_block) {
        case 256: return prep_lds_bt<256>(a, st);
        case 128: return prep_lds_bt<128>(a, st);
        case 64: return prep_lds_bt<64>(a, st);
    }
    return hipErrorInvalidValue;
}

hipError_t check_finite(const double* x, long n_el, int* status, hipStream_t st) {
    const long blocks = n_el + 255 < 256L * 256 * 16 ? (n_el + 255) / 256 : 256L * 16;
    check_finite_kernel<<<dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, st>>>(x, n_el, status);
    return hipGetLastError();
}

hipError_t cell_assign(const CellArgs& a, hipStream_t st) {
    cell_assign_kernel<<<dim3((unsigned)((a.nq + 255) / 256)), dim3(256), 0, st>>>(a);
    return hipGetLastError();
}
hipError_t cell_count(const CellArgs& a, hipStream_t st) {
    cell_count_kernel<<<dim3((unsigned)((a.nq + kBucketBlock * kBucketChunks - 1) / (kBucketBlock * kBucketChunks))), dim3(kBucketBlock), 0, st>>>(a);
    return hipGetLastError();
}
hipError_t cell_scatter(const CellArgs& a, hipStream_t st) {
    cell_scatter_kernel<<<dim3((unsigned)((a.n_pad + kBucketBlock * kBucketChunks - 1) / (kBucketBlock * kBucketChunks))), dim3(kBucketBlock), 0, st>>>(a);
    return hipGetLastError();
}

hipError_t finalize(const FinalizeArgs& f, long n, hipStream_t st) {
    if (f.m_list <= 8) {
        finalize_kernel<8><<<dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), finalize_lds_bytes(8, f.s.d), st>>>(f);
    } else if (f.m_list <= 16) {  // (lists of 12 and 16: 32 lanes per query)
        finalize_kernel<16><<<dim3((unsigned)((n * 32 + 255) / 256)), dim3(256), finalize_lds_bytes(16, f.s.d), st>>>(f);
    } else {
        finalize_kernel<32><<<dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), finalize_lds_bytes(32, f.s.d), st>>>(f);
    }
    return hipGetLastError();
}

hipError_t exact_scan(int formula, bool chunked, const ScanArgs& a, long blocks, size_t lds_bytes, hipStream_t st) {
    switch (formula) {
        case 0: return exact_scan_f<0>(chunked, a, blocks, lds_bytes, st);
        case 1: return exact_scan_f<1>(chunked, a, blocks, lds_bytes, st);
        default: return exact_scan_f<2>(chunked, a, blocks, lds_bytes, st);
    }
}

hipError_t scan_merge(int formula, const ScanArgs& a, long blocks, size_t lds_bytes, int grid_wg_of_scan, int forced_slices,
                      hipStream_t st) {
    const dim3 grid((unsigned)blocks), block(256);
    switch (formula) {
        case 0: scan_merge_kernel<0><<<grid, block, lds_bytes, st>>>(a, grid_wg_of_scan, forced_slices); break;
        case 1: scan_merge_kernel<1><<<grid, block, lds_bytes, st>>>(a, grid_wg_of_scan, forced_slices); break;
        default: scan_merge_kernel<2><<<grid, block, lds_bytes, st>>>(a, grid_wg_of_scan, forced_slices); break;
    }
    return hipGetLastError();
}

hipError_t pack_shards(const double* val, const long* idx, long nq, int n_shards, int kk, double* slice_v, int* slice_i,
                       hipStream_t st) {
    const long heaps = nq * n_shards * kk;
    const long blocks = (heaps + 255) / 256 < 256L * 32 ? (heaps + 255) / 256 : 256L * 32;
    pack_shards_kernel<<<dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, st>>>(val, idx, nq, n_shards, kk, slice_v, slice_i);
    return hipGetLastError();
}

hipError_t predict(const PredictArgs& a, hipStream_t st) {
    const long total = a.nq * a.t;
    predict_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(a);
    return hipGetLastError();
}

hipError_t crosswalk(const long* table, const long* idx, long n, long* out, hipStream_t st) {
    const long want = (n + 255) / 256;
    const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 256L * 16 ? 256L * 16 : want));
    crosswalk_kernel<<<dim3(blocks), dim3(256), 0, st>>>(table, idx, n, out);
    return hipGetLastError();
}

hipError_t add_counter(const int* cnt, long long* total, hipStream_t st) {
    add_counter_kernel<<<dim3(1), dim3(1), 0, st>>>(cnt, total);
    return hipGetLastError();
}

}  // namespace launch
}  // namespace sknnr
