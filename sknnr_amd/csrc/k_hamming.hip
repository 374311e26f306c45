// k_hamming.hip -- kernel translation unit: the weighted-Hamming kernels (hamming.hip.h) behind launch.hip.h.
#define SKNNR_KERNELS_HAMMING 1  // this unit defines the kernels of hamming.hip.h
#include "launch.hip.h"

namespace sknnr {
namespace launch {

hipError_t hamming_pack(const double* xq, long nq, long nq_pad, int t, int tp, uint32_t* qimg, int* q_bad, hipStream_t st) {
    hamming_pack_kernel<<<dim3((unsigned)(nq_pad / 256)), dim3(256), 0, st>>>(xq, nq, nq_pad, t, tp, qimg, q_bad);
    return hipGetLastError();
}

hipError_t hamming_rows(const double* x, long n, int t, int tpr, uint32_t* rows, hipStream_t st) {
    const long n_dw = n * tpr;
    hamming_rows_kernel<<<dim3((unsigned)((n_dw + 255) / 256)), dim3(256), 0, st>>>(x, n, t, tpr, rows);
    return hipGetLastError();
}

hipError_t hamming_coarse(const HammingArgs& a, hipStream_t st) {
    static_assert(hamming_coarse_lds(8) == (size_t)kHamNq * kHamCand * sizeof(unsigned), "one D^ per candidate slot");
    hamming_coarse_kernel<<<dim3((unsigned)((a.nq + kHamNq - 1) / kHamNq)), dim3(kHamWaves * 64), hamming_coarse_lds(a.kk), st>>>(a);
    return hipGetLastError();
}

hipError_t hamming_rescore(const HammingRescoreArgs& a, hipStream_t st) {
    const size_t sh = hamming_rescore_lds(a.s.d);
    hipError_t e = hipFuncSetAttribute((const void*)hamming_rescore_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return e;
    hamming_rescore_kernel<<<dim3((unsigned)((a.s.nq + 3) / 4)), dim3(256), sh, st>>>(a);
    return hipGetLastError();
}

hipError_t hamming_distance_rows(const HammingRowsArgs& a, hipStream_t st) {
    const unsigned gy = (unsigned)(a.n_rows < 1 ? 1 : (a.n_rows > 32768 ? 32768 : a.n_rows));
    hamming_distance_rows_kernel<<<dim3((unsigned)((a.n_ref + 255) / 256), gy), dim3(256), 0, st>>>(a);
    return hipGetLastError();
}

}  // namespace launch
}  // namespace sknnr
