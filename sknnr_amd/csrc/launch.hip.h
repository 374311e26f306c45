// launch.hip.h -- the seam between the host translation unit (sknnr_hip.hip: index build, workspace, host pipeline, C ABI)
// and the kernel translation units (k_exact.hip, k_hamming.hip, k_coarse1.hip, k_coarse2.hip), which are compiled in
// parallel by _build.py.  Every kernel is launched through one of the functions below; each returns the launch's
// hipGetLastError() (the coarse launchers: an int that also says "no such instance").  Argument structs, geometry
// constants and shared-memory sizes live in the kernel headers; a header's non-template kernels are defined only in the unit
// that owns them (SKNNR_KERNELS_EXACT / SKNNR_KERNELS_HAMMING), templates where they are instantiated.
#pragma once
#include <hip/hip_runtime.h>

#include "bucket.hip.h"
#include "coarse2.hip.h"
#include "exact.hip.h"
#include "hamming.hip.h"

namespace sknnr {
namespace launch {

// ---- k_exact.hip: everything around the pre-filter --------------------------------------------------------------------
hipError_t row_norms(const double* x, long n, int d, double* out, hipStream_t st);
hipError_t prep_direct(const PrepArgs& a, hipStream_t st);                  // ks <= 4, register-resident, 256 rows per block
hipError_t prep_lds(int rows_per_block, const PrepArgs& a, hipStream_t st);  // 256 / 128 / 64 rows per block staged through LDS
hipError_t check_finite(const double* x, long n_el, int* status, hipStream_t st);
hipError_t cell_assign(const CellArgs& a, hipStream_t st);
hipError_t cell_count(const CellArgs& a, hipStream_t st);
hipError_t cell_scatter(const CellArgs& a, hipStream_t st);
hipError_t finalize(const FinalizeArgs& f, long n, hipStream_t st);
hipError_t exact_scan(int formula, bool chunked, const ScanArgs& a, long blocks, size_t lds_bytes, hipStream_t st);
hipError_t scan_merge(int formula, const ScanArgs& a, long blocks, size_t lds_bytes, int grid_wg_of_scan, int forced_slices,
                      hipStream_t st);
hipError_t pack_shards(const double* val, const long* idx, long nq, int n_shards, int kk, double* slice_v, int* slice_i,
                       hipStream_t st);
hipError_t predict(const PredictArgs& a, hipStream_t st);
hipError_t crosswalk(const long* table, const long* idx, long n, long* out, hipStream_t st);
hipError_t add_counter(const int* cnt, long long* total, hipStream_t st);

// ---- k_hamming.hip ------------------------------------------------------------------------------------------------------
hipError_t hamming_pack(const double* xq, long nq, long nq_pad, int t, int tp, uint32_t* qimg, int* q_bad, hipStream_t st);
hipError_t hamming_rows(const double* x, long n, int t, int tpr, uint32_t* rows, hipStream_t st);
hipError_t hamming_coarse(const HammingArgs& a, hipStream_t st);
hipError_t hamming_rescore(const HammingRescoreArgs& a, hipStream_t st);
// full float64 distance rows of selected queries (sknnr_hamming_distances)
hipError_t hamming_distance_rows(const HammingRowsArgs& a, hipStream_t st);

// ---- k_coarse1.hip / k_coarse2.hip: the MFMA pre-filters ----------------------------------------------------------------
struct Coarse1Launch {
    const char* rimg;
    int n_stages;
    const uint4* qimg;
    const double* qnc;
    float skip_scale;
    int n_sentinel;
    float* cand_val;
    int* cand_idx;
    long nq_pad;  // rows of the launch (a multiple of the workgroup's rows)
};
struct Coarse2Launch {
    const char* rhi;
    const char* rlo;
    int n_stages;
    const uint4* qimg;
    const double* qnc;
    float skip_scale;
    int n_sentinel;
    float* cand_val;
    int* cand_idx;
    int pos0;
    const int* qperm;
    const unsigned char* qcell;
    const int* cell_stage;
    long rows;  // positions [pos0, pos0 + rows) of the chunk (a multiple of the workgroup's rows)
};
constexpr int kNoInstance = -1;  // the (ks, list length, waves, rank) combination has no compiled kernel
// return 0 and *err = the launch status, or kNoInstance
int coarse1(int ks, int m_list, const Coarse1Launch& L, hipStream_t st, hipError_t* err);
int coarse2(int ks, int m_list, int waves, int rank_extra, const Coarse2Launch& L, hipStream_t st, hipError_t* err);
hipError_t coarse_matrix(int ks, const char* rimg, const int* perm, const uint4* qimg, int n_ref, long nq, long n_tiles,
                         long nqb, float* out);
// development builds (-DSKNNR_COARSE_COUNTERS / -DSKNNR_COARSE_TIMERS): print and clear the device-side tallies
void coarse1_dev_report();
void coarse2_dev_report();

}  // namespace launch
}  // namespace sknnr
