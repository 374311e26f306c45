// exact.hip.h -- float64 stages of the hot path: query preparation (affine transform +
// f16 split image), candidate re-scoring / certification / ordering, the exact
// float64 scan used when a certificate fails, the weighted multi-output mean and the
// dataframe-index crosswalk.
//
// Every dot product is ONE k-ordered float64 fma chain, the same chain as
// oracle/knn_oracle.c (the reference's come from OpenBLAS dgemm/ddot whose order is
// CPU specific; see DESIGN.md "Numerics").  This file is compiled with
// -ffp-contract=off: the only fused operations are the explicit fma() calls.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "bucket.hip.h"
#include "coarse.hip.h"

namespace sknnr {

// ---------------------------------------------------------------------------------------
// prep_queries_kernel: one thread per query row.
//   (1) xt = ((x - center) / scale) @ proj          REF/src/sknnr/_base.py:236-239 and the
//       transformers' transform() (cited in include/sknnr_hip.h)
//   (2) b = s (xt - mu) split into f16 hi/lo, stored in MFMA B-fragment order
//   (3) qnc = |b|^2 (float64), the query term of the certificate
// The block's rows are staged through LDS with coalesced reads (row stride odd ->
// conflict-free ds_read_b64); proj/mu are wave-uniform -> scalar loads.
// ---------------------------------------------------------------------------------------
// Element types of the query rows (include/sknnr_hip.h, sknnr_dtype): widened to float64 by the load, exactly -- what the
// reference's validate_data / float64 promotion does on the host (REF transformers/_cca_transformer.py:78-87).
enum : int { kDtypeF64 = 0, kDtypeF32 = 1, kDtypeI16 = 2, kDtypeU16 = 3, kDtypeU8 = 4, kDtypeI32 = 5, kDtypeCount = 6 };
__host__ __device__ constexpr int dtype_bytes(int dt) {
    return dt == kDtypeF64 ? 8 : (dt == kDtypeF32 || dt == kDtypeI32 ? 4 : (dt == kDtypeU8 ? 1 : 2));
}
// element i of a typed array as float64 (dt is wave-uniform: a scalar branch)
__device__ __forceinline__ double load_as_f64(const void* __restrict__ x, int dt, long i) {
    switch (dt) {
        case kDtypeF32: return (double)((const float*)x)[i];
        case kDtypeI16: return (double)((const short*)x)[i];
        case kDtypeU16: return (double)((const unsigned short*)x)[i];
        case kDtypeU8: return (double)((const unsigned char*)x)[i];
        case kDtypeI32: return (double)((const int*)x)[i];
        default: return ((const double*)x)[i];
    }
}

struct PrepArgs {
    const void* x;         // (nq, d_in) query rows of this launch, elements of x_dtype
    int x_dtype;           // kDtype*: float64 unless the caller handed narrower rows
    long nq;               // live rows
    long nq_pad;           // rows of the fragment image to write (multiple of the block size)
    int d_in;              // columns of x
    int d;                 // transformed feature count
    int ks;                // K-steps: padded feature count dp = 16 * ks
    const double* center;  // (d_in) or null
    const double* scale;   // (d_in) or null
    const double* proj;    // (d_in, dp) zero padded, or null (then d_in == d)
    const double* mu;      // (dp) zero padded centre of the coarse image
    double s;              // power-of-two scale of the coarse image
    double* xt;            // (nq, d) transformed rows out, or null
    uint4* qimg;           // [nq_pad][2][ks][2] 16-byte pieces out (qimg_index), or null (transform only)
    double* qnc;           // (nq) out; +inf marks a row whose image overflows f16 (never certified)
    int* status;           // device word, or null: bit 0 = a query value is NaN, bit 1 = infinite
                           // (SKL/utils/validation.py _assert_all_finite, reached from SKL/neighbors/_base.py:838-845)
    CellTreeDev tree;      // query bucketing (bucket.hip.h): the register-resident kernel also names every row's cell
    unsigned char* cell;   // (nq) out, or null
    int lds_wave_bytes;    // prep_queries_direct_kernel: the wave-private LDS region (prep_direct_wave_lds)
};

// prep_queries_direct_kernel's wave-private LDS tile (round 4): the transformed rows (since round 1) and the image rows of
// the wave's 64 queries are transposed in it, so that every store instruction writes whole KiB of the (contiguous) output
// instead of 16 bytes into each of 64 cache lines; rows sit 16 bytes apart from a power of two (conflict-free 16-byte
// accesses).  (Staging the INPUT rows the same way was built and measured: prep 2.2 -> 4.2 ms -- the tile of 17 KB per wave
// halves the occupancy and the reads through it cost more than the partial-line fetches they replace; not kept.)
__host__ __device__ constexpr int prep_direct_wave_lds(int ks, bool has_xt, bool has_img) {
    const int xt_b = has_xt ? 64 * (8 * ks + 1) * 8 : 0, img_b = has_img ? 64 * (64 * ks + 16) : 0;
    return xt_b > img_b ? xt_b : img_b;
}

// |b| at or above this has no finite f16 image (65504 is the largest f16; the margin keeps hi + lo exact)
constexpr double kImageLimit = 32768.0;

// Wave-level report of non-finite input values: one atomic per wave that saw any.
__device__ __forceinline__ void report_nonfinite(int* status, bool has_nan, bool has_inf) {
    if (!status) return;
    const int bits = (has_nan ? 1 : 0) | (has_inf ? 2 : 0);
    if (bits) atomicOr(status, bits);
}
__device__ __forceinline__ void classify(double v, bool& has_nan, bool& has_inf) {
    has_nan |= v != v;
    has_inf |= fabs(v) == INFINITY;
}

template <int BT>
__global__ void __launch_bounds__(BT) prep_queries_kernel(PrepArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* xs = (double*)smem_raw;
    const int tid = threadIdx.x;
    const long q0 = (long)blockIdx.x * BT;
    const int ldx = a.d_in | 1;
    long n_here = a.nq - q0;
    n_here = n_here < 0 ? 0 : (n_here > BT ? BT : n_here);
    const long n_el = n_here * a.d_in;
    const long e0 = q0 * a.d_in;
    {
        // coalesced walk over the block's contiguous rows; (row, col) advanced incrementally
        int r = tid / a.d_in, c = tid - (tid / a.d_in) * a.d_in;
        const int dr = BT / a.d_in, dc = BT - dr * a.d_in;
        bool has_nan = false, has_inf = false;
        for (long e = tid; e < n_el; e += BT) {
            double v = load_as_f64(a.x, a.x_dtype, e0 + e);
            classify(v, has_nan, has_inf);
            if (a.center) v = v - a.center[c];
            if (a.scale) v = v / a.scale[c];
            xs[r * ldx + c] = v;
            r += dr;
            c += dc;
            if (c >= a.d_in) { c -= a.d_in; ++r; }
        }
        report_nonfinite(a.status, has_nan, has_inf);
    }
    __syncthreads();

    const long q = q0 + tid;
    const bool live = q < a.nq;
    const int dp = 16 * a.ks;
    const double* xrow = xs + tid * ldx;
    double qn = 0.0;
    bool overflow = false;
    for (int jc = 0; jc < 2 * a.ks; ++jc) {
        double acc[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) acc[jj] = 0.0;
        if (live) {
            if (a.proj) {
                const double* pc = a.proj + jc * 8;
                for (int c = 0; c < a.d_in; ++c) {
                    const double xv = xrow[c];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) acc[jj] = fma(xv, pc[(long)c * dp + jj], acc[jj]);
                }
            } else {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int k = jc * 8 + jj;
                    acc[jj] = k < a.d ? xrow[k] : 0.0;
                }
            }
            if (a.xt) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int k = jc * 8 + jj;
                    if (k < a.d) a.xt[q * a.d + k] = acc[jj];
                }
            }
        }
        if (!a.qimg) continue;  // transform-only launch (sknnr_affine_transform)
        half8 hi, lo;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int k = jc * 8 + jj;
            const double b = (live && k < a.d) ? a.s * (acc[jj] - a.mu[k]) : 0.0;
            overflow |= !(fabs(b) < kImageLimit);
            qn = fma(b, b, qn);
            const _Float16 h = (_Float16)(float)b;
            hi[jj] = h;
            lo[jj] = (_Float16)(float)(b - (double)h);
        }
        if (q < a.nq_pad) {
            const int step = jc >> 1, hh = jc & 1;
            a.qimg[qimg_index(q, 0, a.ks, step, hh)] = __builtin_bit_cast(uint4, hi);
            a.qimg[qimg_index(q, 1, a.ks, step, hh)] = __builtin_bit_cast(uint4, lo);
        }
    }
    if (a.qnc && q < a.nq_pad) a.qnc[q] = live ? (overflow ? INFINITY : qn) : 0.0;
}

// Register-resident variant for narrow feature spaces (16*KS <= 64 transformed features): one
// thread per query, all accumulators in VGPRs, no LDS -> occupancy is set by registers only
// (the LDS-staged kernel above holds 264 B of LDS per query and tops out at ~9 waves per CU).
// Each input feature is read once (16-byte loads when rows are 16-byte aligned) and scattered
// into the 16*KS accumulators with wave-uniform (scalar-loaded) projector rows; the fma order
// per output is the same k-ordered chain.
template <int KS>
__global__ void __launch_bounds__(256) prep_queries_direct_kernel(PrepArgs a) {
    constexpr int DP = 16 * KS;
    extern __shared__ __attribute__((aligned(16))) char prep_lds[];
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    const bool live = q < a.nq;
    const int lane_ = threadIdx.x & 63;
    char* wtile = prep_lds + (size_t)(threadIdx.x >> 6) * a.lds_wave_bytes;  // this wave's region: input rows, then output tiles
    double acc[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) acc[j] = 0.0;
    bool has_nan = false, has_inf = false;
    auto transform_row = [&](const void* xb, long xe) {
        if (a.proj) {
            int c = 0;
            if ((a.d_in & 1) == 0 && a.x_dtype == kDtypeF64) {
                const double2* x2 = (const double2*)((const double*)xb + xe);
                for (; c < a.d_in; c += 2) {
                    const double2 xv = x2[c >> 1];
                    double v0 = xv.x, v1 = xv.y;
                    classify(v0, has_nan, has_inf);
                    classify(v1, has_nan, has_inf);
                    if (a.center) { v0 = v0 - a.center[c]; v1 = v1 - a.center[c + 1]; }
                    if (a.scale) { v0 = v0 / a.scale[c]; v1 = v1 / a.scale[c + 1]; }
                    const double* p0 = a.proj + (long)c * DP;
#pragma unroll
                    for (int j = 0; j < DP; ++j) acc[j] = fma(v0, p0[j], acc[j]);
#pragma unroll
                    for (int j = 0; j < DP; ++j) acc[j] = fma(v1, p0[DP + j], acc[j]);
                }
            }
            for (; c < a.d_in; ++c) {
                double v = load_as_f64(xb, a.x_dtype, xe + c);
                classify(v, has_nan, has_inf);
                if (a.center) v = v - a.center[c];
                if (a.scale) v = v / a.scale[c];
                const double* pc = a.proj + (long)c * DP;
#pragma unroll
                for (int j = 0; j < DP; ++j) acc[j] = fma(v, pc[j], acc[j]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < DP; ++k) {
                if (k < a.d) {
                    double v = load_as_f64(xb, a.x_dtype, xe + k);
                    classify(v, has_nan, has_inf);
                    if (a.center) v = v - a.center[k];
                    if (a.scale) v = v / a.scale[k];
                    acc[k] = v;
                }
            }
        }
    };
    if (live) transform_row(a.x, q * a.d_in);
    report_nonfinite(a.status, has_nan, has_inf);
    if (a.cell && live) {
        // cell of the row in the tree over the reference rows' principal axes: float32, on the transformed values at hand
        float z[kCellMaxDepth];
#pragma unroll
        for (int l = 0; l < kCellMaxDepth; ++l) z[l] = 0.f;
#pragma unroll
        for (int k = 0; k < DP; ++k) {
            if (k < a.d) {
                const float v = (float)acc[k] - a.tree.centre[k];
#pragma unroll
                for (int l = 0; l < kCellMaxDepth; ++l)
                    if (l < a.tree.depth) z[l] = fmaf(v, a.tree.axes[l * a.d + k], z[l]);
            }
        }
        a.cell[q] = (unsigned char)cell_of(z, a.tree);
    }
    if (a.xt) {
        // Transformed rows go out through a wave-private LDS tile (64 rows x 8 KS columns, two
        // passes) so that every store instruction writes whole 64..128-byte row segments instead of
        // 64 scattered 8-byte words.  DS operations of one wave execute in order: no barrier.
        constexpr int HC = 8 * KS;
        double* tile = (double*)wtile;  // (the input rows are in registers by now)
        const int lane = threadIdx.x & 63;
        const long q0 = q - lane;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int k = 0; k < HC; ++k) tile[lane * (HC + 1) + k] = acc[h * HC + k];
#pragma unroll
            for (int it = 0; it < HC; ++it) {
                const int e = it * 64 + lane;
                const int row = e / HC, col = e % HC;
                const int kc = h * HC + col;
                if (kc < a.d && q0 + row < a.nq) a.xt[(q0 + row) * a.d + kc] = tile[row * (HC + 1) + col];
            }
        }
    }
    if (!a.qimg) return;
    double qn = 0.0;
    bool overflow = false;
#pragma unroll
    for (int jc = 0; jc < 2 * KS; ++jc) {
        half8 hi, lo;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int k = jc * 8 + jj;
            const double b = (live && k < a.d) ? a.s * (acc[k] - a.mu[k]) : 0.0;
            overflow |= !(fabs(b) < kImageLimit);
            qn = fma(b, b, qn);
            const _Float16 h = (_Float16)(float)b;
            hi[jj] = h;
            lo[jj] = (_Float16)(float)(b - (double)h);
        }
        {
            // the image row (64 KS bytes: [hi | lo][K-step][K half] 16-byte pieces) into the wave's tile, rows 16 bytes apart
            // from a power of two; the 64 image rows of the wave are contiguous in memory and go out below
            const int step = jc >> 1, hh = jc & 1;
            char* irow = wtile + (size_t)lane_ * (64 * KS + 16);
            *(uint4*)(irow + 16 * (int)qimg_index(0, 0, KS, step, hh)) = __builtin_bit_cast(uint4, hi);
            *(uint4*)(irow + 16 * (int)qimg_index(0, 1, KS, step, hh)) = __builtin_bit_cast(uint4, lo);
        }
    }
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int PPR = 4 * KS;  // 16-byte pieces per image row
        const long q0w = q - lane_;   // (a multiple of 64; the launch covers nq_pad rows, a multiple of the block)
        uint4* dst = a.qimg + (size_t)q0w * PPR;
#pragma unroll
        for (int i = 0; i < PPR; ++i) {
            const int p = 64 * i + lane_;
            const int r = p / PPR, o = p % PPR;
            dst[p] = *(const uint4*)(wtile + (size_t)r * (64 * KS + 16) + 16 * o);
        }
    }
    if (a.qnc && q < a.nq_pad) a.qnc[q] = live ? (overflow ? INFINITY : qn) : 0.0;
}

// Finiteness scan of query rows that no prep kernel reads (calls outside the MFMA envelope).
#ifdef SKNNR_KERNELS_EXACT
__global__ void __launch_bounds__(256) check_finite_kernel(const double* __restrict__ x, long n, int* status) {
    bool has_nan = false, has_inf = false;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) classify(x[i], has_nan, has_inf);
    report_nonfinite(status, has_nan, has_inf);
}
#endif  // SKNNR_KERNELS_EXACT

// ---------------------------------------------------------------------------------------
// The reference's pair distance in float64.
// formula 0: |x|^2 + (-2 x.y) + |y|^2, clamped at 0   (_argkmin.pyx.tp:492-502)
// formula 1: sum (x - y)^2, mul and add separately rounded (_dist_metrics.pxd.tp:39-49)
// ---------------------------------------------------------------------------------------
template <int FORMULA>
__device__ __forceinline__ double pair_d2_f(const double* __restrict__ x, const double* __restrict__ r,
                                            int d, double rn) {
    double qn = 0.0, acc = 0.0;
    int c = 0;
    auto step = [&](double xv, double rv) {  // one feature, in ascending k order
        if (FORMULA == 0) {
            qn = fma(xv, xv, qn);
            acc = fma(xv, rv, acc);
        } else {
            const double t = xv - rv;
            acc = acc + t * t;  // -ffp-contract=off keeps the two roundings
        }
    };
    if ((d & 1) == 0) {  // rows start 16-byte aligned: two elements per load
        const double2* x2 = (const double2*)x;
        const double2* r2 = (const double2*)r;
        // eight features per trip, all eight 16-byte loads issued before the first use (sixteen per
        // trip measured slower): the row
        // gathers come from L2 / Infinity Cache and a lane that waits for each load in turn is
        // bound by their latency
        // sixteen features per trip where x comes from LDS (finalize_kernel, round 4): the eight loads of a trip are one whole
        // 128-byte line of the candidate's row, requested before it can fall out of the vector cache (tail 7.5 -> 7.3 ms per
        // 10M rows); with x from memory as well (sixteen loads in flight per lane) the longer trip measured slower
        for (; c + 16 <= d; c += 16) {
            double2 rv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) rv[i] = r2[(c >> 1) + i];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double2 xv = x2[(c >> 1) + i];
                step(xv.x, rv[i].x);
                step(xv.y, rv[i].y);
            }
        }
        for (; c + 8 <= d; c += 8) {
            double2 xv[4], rv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rv[i] = r2[(c >> 1) + i];
                xv[i] = x2[(c >> 1) + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                step(xv[i].x, rv[i].x);
                step(xv[i].y, rv[i].y);
            }
        }
        for (; c < d; c += 2) {
            const double2 xv = x2[c >> 1], rv = r2[c >> 1];
            step(xv.x, rv.x);
            step(xv.y, rv.y);
        }
    }
    for (; c < d; ++c) step(x[c], r[c]);
    if (FORMULA == 0) {
        const double d2 = qn + (-2.0 * acc) + rn;
        return d2 > 0.0 ? d2 : 0.0;
    }
    return acc;
}
__device__ __forceinline__ double pair_d2(const double* __restrict__ x, const double* __restrict__ r,
                                          int d, double rn, int formula) {
    return formula == 0 ? pair_d2_f<0>(x, r, d, rn) : pair_d2_f<1>(x, r, d, rn);
}

struct SelectArgs {
    const double* xq;   // (nq, d) transformed queries
    const double* ref;  // (n_ref, d)
    const double* rn;   // (n_ref) |r|^2 (fma chain)
    long nq;
    int d;
    int n_ref;
    int k;             // neighbours returned
    int kk;            // neighbours searched = k + exclude_self
    int exclude_self;
    int deterministic;
    int formula;
    int pow10_is_divisor;  // decimals < 0
    double pow10;          // 10^|decimals|
    long row_offset;       // global row of query 0 of this launch
    const double* hw;      // formula 2 (weighted Hamming): (d) weights
    double hw_sum;         // ... and their sum in index order (what scipy divides by)
    // Raw candidates of one reference shard (sknnr_shard_candidates): the kk smallest values ascending by
    // (value, index), squared distances as the formula gives them, indices + id_offset; no self exclusion, no
    // square root, no reorder.
    int raw;
    long id_offset;
    double* out_dist;      // (nq, k) or null
    long* out_idx;         // (nq, k)
};

// np.round(x, decimals) as numpy evaluates it (multiply, rint, divide); only the rint'ed
// value is needed for ordering (REF/src/sknnr/_base.py:168-170).
__device__ __forceinline__ double round_key(double x, double p10, int divisor) {
    return divisor ? rint(x / p10) : rint(x * p10);
}

// ---------------------------------------------------------------------------------------
// finalize_kernel: 2M lanes per query, one candidate per lane.
//   exact d2 per candidate -> rank by (d2, index) -> certificate -> drop self (X=None)
//   -> sqrt -> sknnr reorder -> outputs; uncertified queries are queued for exact_scan.
// ---------------------------------------------------------------------------------------
struct FinalizeArgs {
    SelectArgs s;
    const float* cand_val;  // [nq][2][m_list]
    const int* cand_idx;    // image positions (coarse.hip.h sweeps the norm-ordered image)
    const int* perm;        // image position -> reference row
    int m_list;             // entries per lane list written by the coarse kernel (<= M)
    int rank_extra;         // E: the pre-filter's thresholds were of rank m_list + E of the two lists (coarse2.hip.h,
                            //    pair_union_rank; 0: rank m_list)
    const double* qnc;      // (nq)
    double inv_s2;          // 1 / s^2
    double s2;              // s^2
    double inv_s;           // 1 / s
    double eps_c;           // certificate: eps = eps_c * (sqrt(qnc) + ymax)^2 (already * 2^-24)
    double ymax;            // max |s (r - mu)|
    // Rounding noise of the reference's own float64 expression: the value it ranks a
    // pair by differs from the true squared distance by at most noise_a * (|x| + |y|)^2 for the expanded
    // formula (uncentred norms: |x| <= |q'|/s + |mu|) and by noise_a * |x - y|^2 for the direct one (mu2 = 0).
    double noise_a;         // (d + 4) * 2^-53
    double mu2;             // 2 |mu| (expanded formula) or 0 (direct)
    int* fail_list;         // call-relative row ids of uncertified queries
    int fail_base;          // call-relative id of this launch's row 0
    int* fail_count;
    // Bucketed calls (bucket.hip.h): the launch covers positions pos0 .. pos0 + s.nq - 1 of the chunk and position p
    // holds row qperm[p] of the chunk; every per-row array is then indexed by that row.  Null: position = row.
    const int* qperm;
    long pos0;
};

// Lane exchange inside the group of LPQ lanes that share a query.  A group is LPQ/16 DPP rows:
// peer N = 16 b + r is reached by one ds_bpermute (lane ^ 16 b; identical calls are merged by the
// compiler, so there is one per row block and variable) followed by the row rotation `row_ror:r`,
// a VALU move with no LDS traffic and no wait.  N = 1 .. LPQ-1 visits every other lane of the group
// exactly once; only the multiset of peers matters to the callers, which send the peer's lane
// number along the same way when they need it.
template <int LPQ, int N>
__device__ __forceinline__ int peer_i(int v, int c) {
    constexpr int B = N / 16, R = N % 16;
    const int vb = B ? __shfl_xor(v, 16 * B, LPQ) : v;
    if constexpr (R != 0) {
        return __builtin_amdgcn_update_dpp(0, vb, 0x120 + R, 0xf, 0xf, false);
    } else {
        return vb;
    }
}
template <int LPQ, int N>
__device__ __forceinline__ float peer_f(float v, int c) {
    return __int_as_float(peer_i<LPQ, N>(__float_as_int(v), c));
}
template <int LPQ, int N>
__device__ __forceinline__ long peer_l(long v, int c) {
    const int lo = peer_i<LPQ, N>((int)(unsigned)(unsigned long)v, c);
    const int hi = peer_i<LPQ, N>((int)((unsigned long)v >> 32), c);
    return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
}
template <int LPQ, int N>
__device__ __forceinline__ double peer_d(double v, int c) {
    return __longlong_as_double(peer_l<LPQ, N>(__double_as_longlong(v), c));
}
// A value together with its copies from the other row blocks of the group (one ds_bpermute per
// block, fetched once), from which peer N = 16 b + r is a DPP rotation of copy b.
template <int LPQ>
struct PeersI {
    int blk[LPQ / 16];
    __device__ __forceinline__ explicit PeersI(int v) {
        blk[0] = v;
#pragma unroll
        for (int b = 1; b < LPQ / 16; ++b) blk[b] = __shfl_xor(v, 16 * b, LPQ);
    }
    template <int N>
    __device__ __forceinline__ int get() const {
        constexpr int B = N / 16, R = N % 16;
        if constexpr (R != 0) {
            return __builtin_amdgcn_update_dpp(0, blk[B], 0x120 + R, 0xf, 0xf, false);
        } else {
            return blk[B];
        }
    }
};
template <int LPQ>
struct PeersF {
    PeersI<LPQ> p;
    __device__ __forceinline__ explicit PeersF(float v) : p(__float_as_int(v)) {}
    template <int N>
    __device__ __forceinline__ float get() const { return __int_as_float(p.template get<N>()); }
};
template <int LPQ>
struct PeersL {
    PeersI<LPQ> lo, hi;
    __device__ __forceinline__ explicit PeersL(long v)
        : lo((int)(unsigned)(unsigned long)v), hi((int)((unsigned long)v >> 32)) {}
    template <int N>
    __device__ __forceinline__ long get() const {
        return (long)(((unsigned long)(unsigned)hi.template get<N>() << 32) | (unsigned long)(unsigned)lo.template get<N>());
    }
};
template <int LPQ>
struct PeersD {
    PeersL<LPQ> p;
    __device__ __forceinline__ explicit PeersD(double v) : p(__double_as_longlong(v)) {}
    template <int N>
    __device__ __forceinline__ double get() const { return __longlong_as_double(p.template get<N>()); }
};
// f(integral_constant<int, n>) for n = 1 .. LPQ-1
template <int LPQ, int N = 1, typename F>
__device__ __forceinline__ void for_each_peer(F&& f) {
    if constexpr (N < LPQ) {
        f(std::integral_constant<int, N>{});
        for_each_peer<LPQ, N + 1>(f);
    }
}

template <int LPQ, int O = LPQ / 2>
__device__ __forceinline__ double group_min(double v, int c) {
    if constexpr (O > 0) {
        v = fmin(v, peer_d<LPQ, O>(v, c));
        return group_min<LPQ, O / 2>(v, c);
    } else {
        return v;
    }
}
template <int LPQ, int O = LPQ / 2>
__device__ __forceinline__ double group_max(double v, int c) {
    if constexpr (O > 0) {
        v = fmax(v, peer_d<LPQ, O>(v, c));
        return group_max<LPQ, O / 2>(v, c);
    } else {
        return v;
    }
}
template <int LPQ, int O = LPQ / 2>
__device__ __forceinline__ int group_min_i(int v, int c) {
    if constexpr (O > 0) {
        const int w = peer_i<LPQ, O>(v, c);
        return group_min_i<LPQ, O / 2>(w < v ? w : v, c);
    } else {
        return v;
    }
}

// value of the lane that names `dst_lane` (a lane of this wave) as its destination; every destination is named once
template <int LPQ>
__device__ __forceinline__ double scatter_d(double v, int dst_lane) {
    const long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_permute(dst_lane << 2, (int)(unsigned)(unsigned long)bits);
    const int hi = __builtin_amdgcn_ds_permute(dst_lane << 2, (int)((unsigned long)bits >> 32));
    return __longlong_as_double((long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo));
}
// number of lanes of the group starting at wave lane `gbase` that hold the flag
template <int LPQ>
__device__ __forceinline__ int group_count(bool flag, int gbase) {
    const unsigned long long b = __builtin_amdgcn_ballot_w64(flag);
    const unsigned long long mask = LPQ == 64 ? ~0ull : (((1ull << (LPQ & 63)) - 1ull) << gbase);
    return __builtin_popcountll(b & mask);
}
// does any lane of the group starting at wave lane `gbase` hold the flag?
template <int LPQ>
__device__ __forceinline__ bool group_any(bool flag, int gbase) {
    const unsigned long long b = __builtin_amdgcn_ballot_w64(flag);
    const unsigned long long mask = LPQ == 64 ? ~0ull : (((1ull << (LPQ & 63)) - 1ull) << gbase);
    return (b & mask) != 0ull;
}

// (measured per 10M rows: forcing 8 waves per SIMD on the 12-candidate instantiation -- 64 VGPRs, 68 bytes of scratch --
// takes 8.0 ms instead of 4.5; collapsing the row gathers to one row per query saves 0.3 ms; dropping the index
// tie-break from the ranking loop saves nothing: the compiler's 80 VGPRs / 6 waves stay)
// LDS row of one group's query in finalize_kernel: 16 bytes of padding spread the groups of a wave over the banks
__host__ __device__ constexpr size_t finalize_row_bytes(int d) { return (size_t)d * 8 + 16; }
__host__ __device__ constexpr size_t finalize_lds_bytes(int m_tmpl, int d) { return (size_t)(256 / (2 * m_tmpl)) * finalize_row_bytes(d); }

template <int M>
__global__ void __launch_bounds__(256) finalize_kernel(FinalizeArgs a) {
    constexpr int LPQ = 2 * M;
    const SelectArgs& s = a.s;
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    long q = gt / LPQ;
    const int c = (int)(gt % LPQ);
    const bool live = q < s.nq;
    if (!live) q = s.nq - 1;  // keep the lane for the exchanges; it writes nothing
    // the candidate lists are filed by POSITION (coarse2.hip.h): read beside the position -> row table, not behind it
    const long q_list = a.qperm ? a.pos0 + q : q;
    if (a.qperm) q = a.qperm[a.pos0 + q];

    // The query's float64 row, once per GROUP (round 4): every lane of the group re-scores one candidate against the same
    // row, and sixteen lanes fetching the same 256 bytes cost the vector cache as much as sixteen different rows.  The
    // group's lanes load the row together (8 bytes each per trip) into LDS and read it from there.
    extern __shared__ __attribute__((aligned(16))) char fin_lds[];
    double* xrow = (double*)(fin_lds + (size_t)(threadIdx.x / LPQ) * finalize_row_bytes(s.d));
    for (int e = c; e < s.d; e += LPQ) xrow[e] = s.xq[q * s.d + e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // (a group lies inside one wave; DS operations of a wave execute in order)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const int list = c / M, slot = c % M;
    const bool has_slot = slot < a.m_list;
    const long cpos = (q_list * 2 + list) * a.m_list + (has_slot ? slot : 0);
    const int pos_img = has_slot ? a.cand_idx[cpos] : -1;
    const float cv = has_slot ? a.cand_val[cpos] : INFINITY;
    const bool valid = pos_img >= 0 && pos_img < s.n_ref;
    // (gathered for every valid candidate, before the window below is known: asking for the row behind the image position only
    //  once the window is known saves 60 % of these gathers and costs 0.3 ms per 10M rows -- the kernel waits on its chain of
    //  dependent gathers, list -> row id -> row, not on their number)
    const int id = valid ? a.perm[pos_img] : -1;

    // Only candidates whose pre-filter value is within 2 eps of the kk-th smallest one can be
    // among the kk nearest (or tie with them); the others are strictly farther than kk
    // candidates and are not gathered.
    const double qn = a.qnc[q];
    const double nrm = sqrt(qn) + a.ymax;
    const double eps = a.eps_c * nrm * nrm;
    const float cve = valid ? cv : INFINITY;
    // the kk-th smallest pre-filter value of the group: the smallest value that at least kk values do not exceed
    int n_le = 1;
    const PeersF<LPQ> cve_of(cve);
    for_each_peer<LPQ>([&](auto n) { n_le += cve_of.template get<n.value>() <= cve; });
    const double tau_c = group_min<LPQ>(n_le >= s.kk ? (double)cve : INFINITY, c);
    // ... and the reference ranks by its rounded float64 expression: two rows closer than twice its noise
    // may come out in either order, so the window widens by that much (in scaled units).
    const double nr = nrm * a.inv_s + a.mu2;
    const double noise = a.noise_a * nr * nr;
    const bool need = valid && ((double)cv <= tau_c + 2.0 * eps + 2.0 * noise * a.s2);

    double d2 = INFINITY;
    if (need) d2 = pair_d2(xrow, s.ref + (long)id * s.d, s.d, s.rn[id], s.formula);
    const bool usable = need && (d2 == d2) && d2 < INFINITY;
    if (!usable) d2 = INFINITY;
    const int key_id = usable ? id : (0x7fffff00 + c);  // unusable slots sort last, distinct

    // rank by (d2, index)
    int rank = 0;
    const PeersD<LPQ> d2_of(d2);
    const PeersI<LPQ> key_of(key_id);
    for_each_peer<LPQ>([&](auto n) {
        const double dj = d2_of.template get<n.value>();
        const int ij = key_of.template get<n.value>();
        rank += (dj < d2) || (dj == d2 && ij < key_id);
    });
    const int n_usable = group_count<LPQ>(usable, (int)(threadIdx.x & 63) & ~(LPQ - 1));

    // certificate: every reference outside the lists has a float64 d2 above tau
    const double tau = group_min<LPQ>(rank >= s.kk - 1 ? d2 : INFINITY, c);
    // Bound on everything outside the lists: the m_list-th smallest entry of the two lists together
    // (sentinels included), max_i min(a_i, b_{m-1-i}) -- the value the pre-filter's rejections were
    // tested against last (coarse.hip.h, pair_union_rank_m).
    // With rank_extra = E > 0 the rejections were tested against rank m + E,
    //     max(a_{E-1}, b_{E-1}, max_{i = E .. m-1} min(a_i, b_{m+E-1-i})),
    // and the two lists were kept as one pool (coarse2.hip.h, pair_union_rank): what they dropped is >= the final
    // threshold or >= the larger of their last entries, which no rank of the union exceeds.
    const int ex = a.rank_extra;
    const float cv_raw = has_slot ? cv : INFINITY;
    const bool paired = has_slot && slot >= ex;
    const int partner = M + (paired ? a.m_list + ex - 1 - slot : 0);
    const float cv_partner = __shfl(cv_raw, partner, LPQ);
    double t_pair = (list == 0 && paired) ? (double)fminf(cv_raw, cv_partner) : -INFINITY;
    if (ex > 0 && has_slot && slot == ex - 1) t_pair = (double)cv_raw;
    const double t_min = group_max<LPQ>(t_pair, c);
    // true d2 of every outside row >= (qn + t_min - eps) / s^2; the value the reference ranks it by is at
    // most `noise` below that.  qn = +inf (image overflow) and NaN inputs fail the comparison.
    const double bound = (qn + t_min - eps) * a.inv_s2 - noise;
    bool certified = (n_usable >= s.kk) && (tau < INFINITY) && (bound > tau);
    // Exactly tied float64 distances: which tied row the reference keeps at the k-th slot (and,
    // without deterministic ordering, in which order it lists tied rows) depends on its heap's
    // history -- such queries are replayed by exact_scan_kernel.
    // (values scattered to their rank's lane: a tie involves two neighbours of that order.  d2 is ascending in the
    //  rank, unusable slots hold +inf at the end.)
    {
        const int gbase = (int)(threadIdx.x & 63) & ~(LPQ - 1);
        const double by_rank = scatter_d<LPQ>(d2, gbase + rank);          // lane r of the group: the r-th smallest d2
        const double next = __shfl_down(by_rank, 1, LPQ);                // lane r: the (r+1)-th smallest (r = LPQ-1: itself)
        const bool has_next = c + 1 < LPQ;
        const bool tied_here = has_next && by_rank == next && by_rank < INFINITY &&
                               (s.deterministic ? c == s.kk - 1 : c < s.kk);  // across the boundary / involving a kept row
        // (raw shard candidates are listed by (value, index); which of the rows tied at a shard's last slot it lists is
        //  settled by the merge, which sees that the shard's list is full and ends at the boundary value)
        if (!s.raw && group_any<LPQ>(tied_here, gbase)) certified = false;
    }

    // X=None: drop the row's own index, or the first entry when it is absent
    // (SKL/neighbors/_base.py:936-963)
    int sel = rank;
    const long self_id = s.row_offset + q;
    if (s.exclude_self) {
        const bool is_self = usable && (long)id == self_id && rank < s.kk;
        int drop = group_min_i<LPQ>(is_self ? rank : 0x7fffffff, c);
        if (drop == 0x7fffffff) drop = 0;
        sel = rank == drop ? -1 : (rank > drop ? rank - 1 : rank);
    }
    const bool chosen = usable && sel >= 0 && sel < s.k;
    const double dist = s.raw ? d2 : sqrt(d2 > 0.0 ? d2 : 0.0);

    int pos = sel;
    if (s.deterministic) {
        // REF/src/sknnr/_base.py:166-175.  The rounded keys are non-decreasing in `sel`; only equal neighbours can
        // change places, and that is rare (distances equal to 10 decimals): the all-pairs ordering below runs only in
        // waves that hold such a pair.
        const double dmax = group_max<LPQ>(chosen ? dist : 0.0, c);
        const double row_scale = fmax(dmax, 1.0);
        const double k0 = chosen ? round_key(dist / row_scale, s.pow10, s.pow10_is_divisor) : INFINITY;
        const int gbase = (int)(threadIdx.x & 63) & ~(LPQ - 1);
        // slot `sel` of the group for every lane that has one (unique: 0 .. LPQ-2 after a drop), the dropped lane parks at the end
        const double k0_by_sel = scatter_d<LPQ>(k0, gbase + (sel >= 0 ? sel : LPQ - 1));
        const double k0_next = __shfl_down(k0_by_sel, 1, LPQ);
        const bool equal_keys = c + 1 < s.k && c + 1 < LPQ && k0_by_sel == k0_next && k0_by_sel < INFINITY;
        if (__builtin_amdgcn_ballot_w64(equal_keys) != 0) {
            long k1 = (long)id - self_id;
            k1 = k1 < 0 ? -k1 : k1;
            const int tagged = key_id | (chosen ? (int)0x80000000u : 0);  // key_id >= 0: bit 31 carries `chosen`
            pos = 0;
            const PeersD<LPQ> k0_of(k0);
            const PeersL<LPQ> k1_of(k1);
            const PeersI<LPQ> tag_of(tagged);
            for_each_peer<LPQ>([&](auto n) {
                const double k0j = k0_of.template get<n.value>();
                const long k1j = k1_of.template get<n.value>();
                const int tj = tag_of.template get<n.value>();
                const int ij = tj & 0x7fffffff;
                const bool less = (k0j < k0) || (k0j == k0 && (k1j < k1 || (k1j == k1 && ij < key_id)));
                pos += (tj < 0) && less;
            });
        }
    }
    if (live && chosen) {
        if (s.out_dist) s.out_dist[q * s.k + pos] = dist;
        s.out_idx[q * s.k + pos] = id + s.id_offset;
    }
    if (live && c == 0 && !certified) {
        const int slot = atomicAdd(a.fail_count, 1);
        a.fail_list[slot] = a.fail_base + (int)q;
    }
}

// ---------------------------------------------------------------------------------------
// exact_scan_kernel: the reference's engine itself, one workgroup per query.
//
// Every reference row is visited in ascending index order, its float64 distance pushed
// into a fixed-size max-heap that rejects values equal to its root, the heap is sorted with
// the reference's own (unstable) dual quicksort, then the X=None / sqrt / reorder post-steps
// run -- a faithful replay of
//   heap_push          SKL/utils/_heap.pyx:6-85
//   simultaneous_sort  SKL/utils/_sorting.pyx:19-93
//   drop self          SKL/neighbors/_base.py:936-963
// so that even exactly tied distances (duplicate rows, integer-valued features,
// cancellation-quantised clusters) come out as the reference returns them: which of several
// tied rows survives at the k-th slot depends on the heap's history, not on the index.
//
// Used for (a) queries whose MFMA certificate failed, (b) queries with an exact tie at the
// k-th slot (or, without deterministic ordering, any tie among the k), (c) calls outside the
// MFMA envelope (large k, very wide features).
// ---------------------------------------------------------------------------------------
struct ScanArgs {
    SelectArgs s;       // s.ref is unused here; refT below
    const double* refT; // (d, n_ref) transposed reference rows
    const int* list;    // query ids to process, or null = all 0..count-1
    const int* count;   // device count (with list), else null and s.nq is used
    // Few queries (fewer passes than workgroups of the grid): the reference rows of a pass are split into S slices,
    // one workgroup each (scan_slices); the slice heaps go to slice_v / slice_i [query slot][S][kk] and
    // scan_merge_kernel combines them.  Null = never sliced.
    double* slice_v;
    int* slice_i;
    int* list2;         // scan_merge_kernel: queries whose merged heaps are not unique (exact ties) -> sequential scan
    int* count2;
};

constexpr int kScanSliceMaxKK = 32;  // sliced mode serves kk <= 32 (the serial merge is O(S kk) insertions)
constexpr int kScanMaxSlices = 32;
// slices per pass, computed the same way by the scan, the merge and (as an upper bound) the host
__host__ __device__ inline int scan_slices(long n_items, int nq_pass, int n_ref, int kk, int grid_wg) {
    const long passes = (n_items + nq_pass - 1) / nq_pass;
    if (passes <= 0 || kk > kScanSliceMaxKK) return 1;
    const int n_steps = (n_ref + 511) / 512;  // kScanRefs references per step: a slice is at least one step
    long sl = grid_wg / passes;
    if (sl > kScanMaxSlices) sl = kScanMaxSlices;
    if (sl > n_steps) sl = n_steps;
    return sl < 2 ? 1 : (int)sl;
}

__device__ __forceinline__ void heap_push_ref(double* hv, int* hi, int k, double v, int id) {
    // caller has checked v < hv[0]
    int cur = 0;
    for (;;) {
        const int l = 2 * cur + 1, r = l + 1;
        int nxt;
        if (l >= k) break;
        if (r >= k) {
            if (hv[l] > v) nxt = l; else break;
        } else if (hv[l] >= hv[r]) {
            if (v < hv[l]) nxt = l; else break;
        } else {
            if (v < hv[r]) nxt = r; else break;
        }
        hv[cur] = hv[nxt];
        hi[cur] = hi[nxt];
        cur = nxt;
    }
    hv[cur] = v;
    hi[cur] = id;
}

// formula 1 (the kd_tree stand-in) and formula 2 (weighted Hamming): rows ordered by (d2, index), as
// oracle_argkmin_direct keeps them -- a sorted list; a value equal to the current k-th is not admitted,
// equal values stay in index order.  Unfilled slots hold DBL_MAX.
__device__ __forceinline__ void sorted_insert_ref(double* hv, int* hi, int k, double v, int id) {
    // caller has checked v < hv[k-1]
    int pos = k - 1;
    while (pos > 0 && hv[pos - 1] > v) {
        hv[pos] = hv[pos - 1];
        hi[pos] = hi[pos - 1];
        --pos;
    }
    hv[pos] = v;
    hi[pos] = id;
}

__device__ __forceinline__ void swap_slots(double* v, int* x, int a, int b) {
    const double tv = v[a]; v[a] = v[b]; v[b] = tv;
    const int tx = x[a]; x[a] = x[b]; x[b] = tx;
}

// The reference's recursive dual quicksort, run with an explicit stack (the two halves are
// independent, so the visiting order does not change the result).
__device__ void dual_quicksort_ref(double* v0, int* x0, int n0, int* stack) {
    int sp = 0;
    stack[sp++] = 0;
    stack[sp++] = n0;
    while (sp > 0) {
        const int n = stack[--sp];
        const int off = stack[--sp];
        double* v = v0 + off;
        int* x = x0 + off;
        if (n <= 1) continue;
        if (n == 2) {
            if (v[0] > v[1]) swap_slots(v, x, 0, 1);
            continue;
        }
        if (n == 3) {
            if (v[0] > v[1]) swap_slots(v, x, 0, 1);
            if (v[1] > v[2]) {
                swap_slots(v, x, 1, 2);
                if (v[0] > v[1]) swap_slots(v, x, 0, 1);
            }
            continue;
        }
        const int mid = n / 2;
        if (v[0] > v[n - 1]) swap_slots(v, x, 0, n - 1);
        if (v[n - 1] > v[mid]) {
            swap_slots(v, x, n - 1, mid);
            if (v[0] > v[n - 1]) swap_slots(v, x, 0, n - 1);
        }
        const double pivot = v[n - 1];
        int store = 0;
        for (int i = 0; i < n - 1; ++i) {
            if (v[i] < pivot) {
                swap_slots(v, x, i, store);
                ++store;
            }
        }
        swap_slots(v, x, store, n - 1);
        if (store > 1) { stack[sp++] = off; stack[sp++] = store; }
        if (store + 2 < n) { stack[sp++] = off + store + 1; stack[sp++] = n - store - 1; }
    }
}

// The end of a query's replay: its kk heap entries -> sort (expanded formula: the reference's quicksort on the heap
// array) -> drop self (X=None) -> sqrt -> sknnr's reorder -> outputs.  Serial over <= kk entries.
template <int FORMULA>
__device__ void scan_finish_query(const SelectArgs& s, long q, double* hv, int* hi, int* stack) {
    const int KK = s.kk;
    if (s.raw) {
        // shard candidates: ascending by (value, index) -- an insertion sort of the heap array (formulas 1 and 2 keep
        // their lists in that order already) -- values and indices as they are
        for (int e = 1; e < KK; ++e) {
            const double dv = hv[e];
            const int iv = hi[e];
            int jj = e - 1;
            while (jj >= 0 && (hv[jj] > dv || (hv[jj] == dv && hi[jj] > iv))) {
                hv[jj + 1] = hv[jj];
                hi[jj + 1] = hi[jj];
                --jj;
            }
            hv[jj + 1] = dv;
            hi[jj + 1] = iv;
        }
        for (int e = 0; e < s.k; ++e) {
            if (s.out_dist) s.out_dist[q * s.k + e] = hv[e];
            s.out_idx[q * s.k + e] = (long)hi[e] + s.id_offset;
        }
        return;
    }
    if (FORMULA == 0) dual_quicksort_ref(hv, hi, KK, stack);
    const long self_id = s.row_offset + q;
    int drop = -1;
    if (s.exclude_self) {
        drop = 0;
        for (int e = 0; e < KK; ++e)
            if ((long)hi[e] == self_id) { drop = e; break; }
    }
    int n = 0;
    for (int e = 0; e < KK; ++e) {
        if (e == drop) continue;
        hv[n] = FORMULA == 2 ? hv[e] : sqrt(hv[e] > 0.0 ? hv[e] : 0.0);  // (a Hamming distance is not a square)
        hi[n] = hi[e];
        ++n;
    }
    if (n > s.k) n = s.k;
    if (s.deterministic) {
        double dmax = 0.0;
        for (int e = 0; e < n; ++e) dmax = fmax(dmax, hv[e]);
        const double row_scale = fmax(dmax, 1.0);
        // stable insertion sort by (rounded, |idx - row|, idx)  (REF _base.py:166-175)
        for (int e = 1; e < n; ++e) {
            const double dv = hv[e];
            const int iv = hi[e];
            const double k0 = round_key(dv / row_scale, s.pow10, s.pow10_is_divisor);
            long k1 = (long)iv - self_id;
            k1 = k1 < 0 ? -k1 : k1;
            int jj = e - 1;
            while (jj >= 0) {
                const double k0j = round_key(hv[jj] / row_scale, s.pow10, s.pow10_is_divisor);
                long k1j = (long)hi[jj] - self_id;
                k1j = k1j < 0 ? -k1j : k1j;
                const bool greater = (k0j > k0) || (k0j == k0 && (k1j > k1 || (k1j == k1 && hi[jj] > iv)));
                if (!greater) break;
                hv[jj + 1] = hv[jj];
                hi[jj + 1] = hi[jj];
                --jj;
            }
            hv[jj + 1] = dv;
            hi[jj + 1] = iv;
        }
    }
    for (int e = 0; e < n; ++e) {
        if (s.out_dist) s.out_dist[q * s.k + e] = hv[e];
        s.out_idx[q * s.k + e] = hi[e];
    }
}

#ifndef SKNNR_SCAN_WPS
#define SKNNR_SCAN_WPS 3  // 3 workgroups per CU (<= 170 VGPR): 1.5 ms instead of 2.1 ms for 8.5k rows; 4 (128 VGPR, spills) is no faster
#endif
constexpr int kScanWaves = 4;
// queries whose heaps one wave replays: three for the expanded formula (165 VGPRs, no spills: 10-17 % faster than
// two on exact-only calls and tie-heavy data, scripts/scan_probe.py), two for the direct and Hamming formulas
// (three spill there and lose: 20000 x 20000 x 500 trees 31.5 -> 45.9 ms)
__host__ __device__ constexpr int scan_qpw(int formula) { return formula == 0 ? 3 : 2; }
__host__ __device__ constexpr int scan_nq(int formula) { return kScanWaves * scan_qpw(formula); }  // queries per workgroup pass
constexpr int kScanRefs = 2 * kScanWaves * 64;       // references per step: two per thread
constexpr int kScanColChunk = 1024;                  // feature columns of the queries held in LDS at a time

// Workgroup LDS: xs[NQ][dpad] | qn[NQ] | hv[NQ][KK] | hi[NQ][kkp] | stack[NQ][stk] | d2[NQ][kScanRefs]
struct ScanLayout {
    int dpad, kkp, stk;
    size_t xs, qn, hv, hi, stack, d2, total;
};
__host__ __device__ inline ScanLayout scan_layout(int d, int kk, int nq) {
    ScanLayout L;
    L.dpad = ((d < kScanColChunk ? d : kScanColChunk) + 1) & ~1;
    L.kkp = kk + (kk & 1);
    L.stk = (2 * kk + 4 + 1) & ~1;
    size_t b = 0;
    L.xs = b;    b += 8 * (size_t)nq * L.dpad;
    L.qn = b;    b += 8 * (size_t)nq;
    L.hv = b;    b += 8 * (size_t)nq * kk;
    L.hi = b;    b += 4 * (size_t)nq * L.kkp;
    L.stack = b; b += 4 * (size_t)nq * L.stk;
    b = (b + 15) & ~(size_t)15;
    L.d2 = b;    b += 8 * (size_t)nq * kScanRefs;
    L.total = b;
    return L;
}
__host__ __device__ inline size_t scan_block_bytes(int d, int kk, int formula) { return scan_layout(d, kk, scan_nq(formula)).total; }

// One workgroup (4 waves) per pass of scan_nq(FORMULA) queries.  Per step of 512 references every thread
// loads TWO columns of the transposed copy (coalesced) once and evaluates their float64 distances to
// all NQ (8 or 12) queries (query values are LDS broadcasts, each feeding two fma chains), so a query
// costs 1/NQ of a sweep over the reference copy; the NQ x 512 values go to LDS and every wave then
// offers them, in index order, to the heaps of its own QPW queries.  Each distance is the same
// ascending-feature fma chain as before: results do not depend on the grouping.
template <int FORMULA, bool CHUNKED = false>
__global__ void __launch_bounds__(kScanWaves * 64, SKNNR_SCAN_WPS) exact_scan_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NT = kScanWaves * 64;
    constexpr int QPW = scan_qpw(FORMULA);
    constexpr int NQ = scan_nq(FORMULA);
    const SelectArgs& s = a.s;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int KK = s.kk;
    const int d = s.d;
    const ScanLayout L = scan_layout(d, KK, NQ);
    double* xs = (double*)(smem_raw + L.xs);
    double* qns = (double*)(smem_raw + L.qn);
    double* hv_all = (double*)(smem_raw + L.hv);
    int* hi_all = (int*)(smem_raw + L.hi);
    int* stack_all = (int*)(smem_raw + L.stack);
    double* d2buf = (double*)(smem_raw + L.d2);

    const long n_items = a.list ? (long)*a.count : s.nq;
    // sliced mode: workgroup b sweeps slice b % S of pass b / S and leaves its heaps to scan_merge_kernel
    const int S = a.slice_v ? scan_slices(n_items, NQ, s.n_ref, KK, (int)gridDim.x) : 1;
    const int slice = S > 1 ? (int)(blockIdx.x % S) : 0;
    const int n_steps = (s.n_ref + kScanRefs - 1) / kScanRefs;
    const int j_begin = (int)((long)slice * n_steps / S) * kScanRefs;
    const int j_end_raw = (int)((long)(slice + 1) * n_steps / S) * kScanRefs;
    const int j_end = j_end_raw < s.n_ref ? j_end_raw : s.n_ref;
    const long f_first = (S > 1 ? (long)(blockIdx.x / S) : (long)blockIdx.x) * NQ;
    const long f_stride = S > 1 ? (1L << 40) : (long)gridDim.x * NQ;
    for (long f0 = f_first; f0 < n_items; f0 += f_stride) {
        const int n_here = (int)((n_items - f0) < NQ ? (n_items - f0) : NQ);
        __syncthreads();  // previous pass's LDS state is dead
        // padding slots repeat the pass's last query; nothing is written for them
        // query rows of the pass, columns [c0, c0 + kScanColChunk): everything when d fits one chunk (the common
        // case: loaded once per pass), else chunk by chunk inside the sweep (wide node-id matrices of RFNN / GBNN)
        constexpr bool chunked = CHUNKED;  // d > kScanColChunk (the launcher picks the instantiation)
        auto load_chunk = [&](int c0) {
            const int cw = (d - c0) < kScanColChunk ? (d - c0) : kScanColChunk;
            for (int e = tid; e < NQ * cw; e += NT) {
                const int qi = e / cw, c = e - qi * cw;
                const long fi = f0 + (qi < n_here ? qi : n_here - 1);
                const long q = a.list ? (long)a.list[fi] : fi;
                xs[qi * L.dpad + c] = s.xq[q * d + c0 + c];
            }
        };
        if (!chunked) load_chunk(0);
        for (int i = tid; i < NQ * KK; i += NT) {
            const int qi = i / KK, e = i - qi * KK;
            hv_all[qi * KK + e] = DBL_MAX;
            hi_all[qi * L.kkp + e] = 0;
        }
        __syncthreads();
        if (FORMULA == 0 && tid < NQ) {
            double qn = 0.0;
            if (!chunked) {  // the rows are in LDS
                for (int c = 0; c < d; ++c) qn = fma(xs[tid * L.dpad + c], xs[tid * L.dpad + c], qn);
            } else {
                const long fi = f0 + (tid < n_here ? tid : n_here - 1);
                const double* xr = s.xq + (a.list ? (long)a.list[fi] : fi) * d;
                for (int c = 0; c < d; ++c) qn = fma(xr[c], xr[c], qn);
            }
            qns[tid] = qn;
        }
        double root[QPW];
#pragma unroll
        for (int i = 0; i < QPW; ++i) root[i] = DBL_MAX;

        const size_t ld = (size_t)s.n_ref;
        for (int j0 = j_begin; j0 < j_end; j0 += kScanRefs) {
            const int ja = j0 + tid, jb = j0 + NT + tid;
            const double* cola = a.refT + (ja < s.n_ref ? ja : 0);
            const double* colb = a.refT + (jb < s.n_ref ? jb : 0);
            double acc[NQ][2];
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) acc[qi][0] = acc[qi][1] = 0.0;
            for (int c0 = 0; c0 < (chunked ? d : 1); c0 += kScanColChunk) {
            if (chunked) {
                __syncthreads();  // everyone is done with the previous chunk
                load_chunk(c0);
                __syncthreads();
            }
            const int ce = (!chunked || (d - c0) < kScanColChunk) ? d : c0 + kScanColChunk;
            int c = c0;
            for (; c + 8 <= ce; c += 8) {
                double ra[8], rb[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    ra[w] = cola[(size_t)(c + w) * ld];
                    rb[w] = colb[(size_t)(c + w) * ld];
                }
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
#pragma unroll
                    for (int w = 0; w < 8; ++w) {
                        const double x = xs[qi * L.dpad + (c - c0) + w];
                        if (FORMULA == 0) {
                            acc[qi][0] = fma(x, ra[w], acc[qi][0]);
                            acc[qi][1] = fma(x, rb[w], acc[qi][1]);
                        } else if (FORMULA == 2) {
                            // weighted Hamming as scipy's cdist evaluates it on the float64 node ids:
                            // s += (u != v) * w, in tree order  (REF _weighted_trees.py:53-59, :139-140)
                            const double wv = s.hw[c + w];
                            acc[qi][0] = x != ra[w] ? acc[qi][0] + wv : acc[qi][0];
                            acc[qi][1] = x != rb[w] ? acc[qi][1] + wv : acc[qi][1];
                        } else {
                            const double ta = x - ra[w], tb = x - rb[w];
                            acc[qi][0] = acc[qi][0] + ta * ta;
                            acc[qi][1] = acc[qi][1] + tb * tb;
                        }
                    }
                }
            }
            for (; c < ce; ++c) {
                const double rav = cola[(size_t)c * ld], rbv = colb[(size_t)c * ld];
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    const double x = xs[qi * L.dpad + (c - c0)];
                    if (FORMULA == 0) {
                        acc[qi][0] = fma(x, rav, acc[qi][0]);
                        acc[qi][1] = fma(x, rbv, acc[qi][1]);
                    } else if (FORMULA == 2) {
                        const double wv = s.hw[c];
                        acc[qi][0] = x != rav ? acc[qi][0] + wv : acc[qi][0];
                        acc[qi][1] = x != rbv ? acc[qi][1] + wv : acc[qi][1];
                    } else {
                        const double ta = x - rav, tb = x - rbv;
                        acc[qi][0] = acc[qi][0] + ta * ta;
                        acc[qi][1] = acc[qi][1] + tb * tb;
                    }
                }
            }
            }  // column chunks
            const double rna = (FORMULA == 0 && ja < s.n_ref) ? s.rn[ja] : 0.0;
            const double rnb = (FORMULA == 0 && jb < s.n_ref) ? s.rn[jb] : 0.0;
            __syncthreads();  // the previous step's replay has finished reading d2buf (and qns is published)
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                double da = acc[qi][0], db = acc[qi][1];
                if (FORMULA == 0) {
                    const double qn = qns[qi];
                    da = qn + (-2.0 * da) + rna;
                    db = qn + (-2.0 * db) + rnb;
                    da = da > 0.0 ? da : 0.0;
                    db = db > 0.0 ? db : 0.0;
                }
                if (FORMULA == 2) {
                    da = da / s.hw_sum;
                    db = db / s.hw_sum;
                }
                d2buf[qi * kScanRefs + tid] = ja < s.n_ref ? da : INFINITY;
                d2buf[qi * kScanRefs + NT + tid] = jb < s.n_ref ? db : INFINITY;
            }
            __syncthreads();
#pragma unroll 1
            for (int i = 0; i < QPW; ++i) {
                const int qi = wave * QPW + i;
                if (qi >= n_here) break;
                double* hv = hv_all + qi * KK;
                int* hi = hi_all + qi * L.kkp;
                const double* buf = d2buf + qi * kScanRefs;
                double rt = root[i];
#pragma unroll 1
                for (int u = 0; u < kScanRefs / 64; ++u) {
                    const double v64 = buf[64 * u + lane];
                    unsigned long long m = __builtin_amdgcn_ballot_w64(v64 < rt);
                    while (m) {
                        const int bit = __builtin_ctzll(m);
                        m &= m - 1;
                        const double v = __shfl(v64, bit, 64);
                        if (v < rt) {  // the bound may have dropped since the ballot
                            if (lane == 0) {
                                if (FORMULA == 0) heap_push_ref(hv, hi, KK, v, j0 + 64 * u + bit);
                                else sorted_insert_ref(hv, hi, KK, v, j0 + 64 * u + bit);
                            }
                            __builtin_amdgcn_wave_barrier();
                            rt = FORMULA == 0 ? hv[0] : hv[KK - 1];
                        }
                    }
                }
                root[i] = rt;
            }
        }

        if (S > 1) {
            // raw heaps of this slice (unfilled slots hold DBL_MAX), [query slot][slice][kk]
            __syncthreads();
            for (int e = tid; e < n_here * KK; e += NT) {
                const int qi = e / KK, c = e - qi * KK;
                const size_t o = ((size_t)(f0 + qi) * S + slice) * KK + c;
                a.slice_v[o] = hv_all[qi * KK + c];
                a.slice_i[o] = hi_all[qi * L.kkp + c];
            }
            continue;
        }
        if (lane == 0) {
            for (int i = 0; i < QPW; ++i) {
                const int qi = wave * QPW + i;
                if (qi >= n_here) break;
                const long q = a.list ? (long)a.list[f0 + qi] : f0 + qi;
                scan_finish_query<FORMULA>(s, q, hv_all + qi * KK, hi_all + qi * L.kkp, stack_all + qi * L.stk);
            }
        }
    }
}

// Sliced scans: combine the S slice heaps of every query (one wave per query, lane 0 works) and finish the
// query.  Under the direct and Hamming formulas the selection rule is "smallest (d2, index) first", which the
// union of the slices' lists answers exactly.  Under the expanded formula the reference's heap decides
// exact ties by its history: the union is its answer only when that answer is unique -- no tie at the k-th
// value (seen, or possibly dropped inside a full slice), with non-deterministic ordering no equal values among
// the kept rows, and with X=None the row itself among them; every other query goes to list2 for the sequential
// scan.
// (forced_slices > 0: the slices are the candidate lists of that many reference SHARDS -- other handles, other GPUs --
//  laid out the same way: sknnr_merge_shards)
template <int FORMULA>
__global__ void __launch_bounds__(256) scan_merge_kernel(ScanArgs a, int grid_wg_of_scan, int forced_slices) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const SelectArgs& s = a.s;
    const int KK = s.kk;
    const long n_items = a.list ? (long)*a.count : s.nq;
    const int S = forced_slices > 0 ? forced_slices : scan_slices(n_items, scan_nq(FORMULA), s.n_ref, KK, grid_wg_of_scan);
    if (S == 1 && forced_slices <= 0) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long slot = (long)blockIdx.x * 4 + wave;
    if (slot >= n_items || lane != 0) return;
    // per wave: hv[KK + 1] | hi[KK + 1] | stack
    const int kkp = (KK + 2) & ~1, stk = (2 * KK + 4 + 1) & ~1;
    char* base = smem_raw + (size_t)wave * (8 * (size_t)kkp + 4 * (size_t)kkp + 4 * (size_t)stk);
    double* hv = (double*)base;
    int* hi = (int*)(base + 8 * (size_t)kkp);
    int* stack = hi + kkp;
    const long q = a.list ? (long)a.list[slot] : slot;

    int n = 0;  // entries of the merged list, at most KK + 1, ascending by (value, index)
    for (int sl = 0; sl < S; ++sl) {
        const double* sv = a.slice_v + ((size_t)slot * S + sl) * KK;
        const int* si = a.slice_i + ((size_t)slot * S + sl) * KK;
        for (int e = 0; e < KK; ++e) {
            const double v = sv[e];
            if (v == DBL_MAX) continue;  // unfilled slot
            const int id = si[e];
            if (n == KK + 1 && !(v < hv[KK] || (v == hv[KK] && id < hi[KK]))) continue;
            int pos = n < KK + 1 ? n : KK;
            while (pos > 0 && (hv[pos - 1] > v || (hv[pos - 1] == v && hi[pos - 1] > id))) {
                hv[pos] = hv[pos - 1];
                hi[pos] = hi[pos - 1];
                --pos;
            }
            hv[pos] = v;
            hi[pos] = id;
            if (n < KK + 1) ++n;
        }
    }
    bool unique = n >= KK;
    if (FORMULA == 0 && unique) {
        const double vk = hv[KK - 1];
        if (n > KK && hv[KK] == vk) unique = false;  // a tie at the k-th value
        // a FULL slice whose largest kept value is the k-th value overall may have rejected or evicted rows equal
        // to it (whatever a slice rejects or evicts is >= its final maximum: irrelevant when that is above vk; a
        // full slice cannot end below vk, or k rows would lie below the k-th value)
        for (int sl = 0; sl < S && unique; ++sl) {
            const double* sv = a.slice_v + ((size_t)slot * S + sl) * KK;
            int filled = 0;
            double smax = -INFINITY;
            for (int e = 0; e < KK; ++e)
                if (sv[e] != DBL_MAX) { ++filled; smax = fmax(smax, sv[e]); }
            if (filled == KK && smax == vk) unique = false;
        }
        if (unique && !s.deterministic)
            for (int e = 1; e < KK; ++e)
                if (hv[e] == hv[e - 1]) unique = false;  // the heap's history orders equal values
        if (unique && s.exclude_self) {
            bool found = false;
            for (int e = 0; e < KK; ++e) found = found || (long)hi[e] == s.row_offset + q;
            if (!found) unique = false;  // "drop the first entry" depends on the order among equal values
        }
    }
    if (!unique) {
        const int o = atomicAdd(a.count2, 1);
        a.list2[o] = (int)q;
        return;
    }
    scan_finish_query<FORMULA>(s, q, hv, hi, stack);
}

// ---------------------------------------------------------------------------------------
// predict_kernel: one thread per (query, output).
// KNeighborsRegressor.predict  SKL/neighbors/_regression.py:224-268
// _get_weights                 SKL/neighbors/_base.py:81-124
// ---------------------------------------------------------------------------------------
struct PredictArgs {
    const double* y;     // (n_ref, t)
    const double* dist;  // (nq, k) or null (uniform)
    const long* idx;     // (nq, k)
    const double* w;     // (nq, k) explicit weights or null
    long nq;
    int k;
    int t;
    int mode;            // sknnr_weight_mode
    double* out;         // (nq, t)
};

// numpy's pairwise sum of n < 128 doubles (8 partial sums), so that the k-term sums come
// out bit-identical to np.sum(..., axis=1).
template <typename F>
__device__ __forceinline__ double np_sum(int n, F term) {
    if (n < 8) {
        double r = term(0);
        for (int i = 1; i < n; ++i) r = r + term(i);
        return r;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = term(j);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = r[j] + term(i + j);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res = res + term(i);
    return res;
}

// np_sum for n <= 8 terms held in registers (static indexing only).
__device__ __forceinline__ double np_sum_small(int n, const double (&v)[8]) {
    if (n == 8) return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    double r = v[0];
#pragma unroll
    for (int i = 1; i < 7; ++i)
        if (i < n) r = r + v[i];
    return r;
}

#ifdef SKNNR_KERNELS_EXACT
__global__ void __launch_bounds__(256) predict_kernel(PredictArgs a) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.nq * a.t) return;
    const long q = e / a.t;
    const int tt = (int)(e - q * a.t);
    const long* ids = a.idx + q * a.k;
    const double* dd = a.dist ? a.dist + q * a.k : nullptr;
    const double* ww = a.w ? a.w + q * a.k : nullptr;
    if (a.k <= 8) {
        // The common case: all k indices, then all k target rows (random 8-byte gathers from
        // L2 / Infinity Cache) and distances are requested before the first use; a loop that waits
        // for index i, then row i, is bound by 2k memory latencies.  Same summation order as below.
        long id[8];
        double yv[8], wv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) id[i] = i < a.k ? ids[i] : ids[0];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            yv[i] = a.y[id[i] * a.t + tt];
            wv[i] = a.mode == 0 ? 1.0 : (a.mode == 2 ? (i < a.k ? ww[i] : 0.0) : (i < a.k ? dd[i] : 1.0));
        }
        if (a.mode == 0) {
            double acc = yv[0];
#pragma unroll
            for (int i = 1; i < 8; ++i)
                if (i < a.k) acc = acc + yv[i];
            a.out[e] = acc / (double)a.k;
            return;
        }
        if (a.mode == 1) {
            bool any_zero = false;
#pragma unroll
            for (int i = 0; i < 8; ++i) any_zero |= (i < a.k) && (wv[i] == 0.0);
#pragma unroll
            for (int i = 0; i < 8; ++i) wv[i] = any_zero ? (wv[i] == 0.0 ? 1.0 : 0.0) : 1.0 / wv[i];
        }
        double nv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) nv[i] = yv[i] * wv[i];
        a.out[e] = np_sum_small(a.k, nv) / np_sum_small(a.k, wv);
        return;
    }
    if (a.mode == 0) {
        // np.mean(_y[neigh_ind], axis=1): slices added in order, then one division
        double acc = a.y[ids[0] * a.t + tt];
        for (int i = 1; i < a.k; ++i) acc = acc + a.y[ids[i] * a.t + tt];
        a.out[e] = acc / (double)a.k;
        return;
    }
    bool any_zero = false;
    if (a.mode == 1)
        for (int i = 0; i < a.k; ++i) any_zero |= (dd[i] == 0.0);
    auto weight = [&](int i) -> double {
        if (a.mode == 2) return ww[i];
        if (any_zero) return dd[i] == 0.0 ? 1.0 : 0.0;
        return 1.0 / dd[i];
    };
    const double num = np_sum(a.k, [&](int i) { return a.y[ids[i] * a.t + tt] * weight(i); });
    const double den = np_sum(a.k, [&](int i) { return weight(i); });
    a.out[e] = num / den;
}
#endif  // SKNNR_KERNELS_EXACT

// Candidate lists of G shards, (G, nq, kk) float64 values and int64 indices as the ranks' all-gather delivers them,
// into the merge kernel's layout [query][G][kk] (indices as int: a reference set has fewer than 2^31 rows).
#ifdef SKNNR_KERNELS_EXACT
__global__ void __launch_bounds__(256)
pack_shards_kernel(const double* __restrict__ val, const long* __restrict__ idx, long nq, int g_count, int kk,
                   double* __restrict__ slice_v, int* __restrict__ slice_i) {
    const long total = nq * g_count * kk;
    const long stride = (long)gridDim.x * 256;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {
        const long q = o / ((long)g_count * kk);
        const int r = (int)(o - q * (long)g_count * kk);
        const int g = r / kk, e = r - g * kk;
        const long src = ((long)g * nq + q) * kk + e;
        slice_v[o] = val[src];
        slice_i[o] = (int)idx[src];
    }
}
#endif  // SKNNR_KERNELS_EXACT

#ifdef SKNNR_KERNELS_EXACT
__global__ void __launch_bounds__(256)
crosswalk_kernel(const long* __restrict__ table, const long* __restrict__ idx, long n, long* __restrict__ out) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = table[idx[i]];
}
#endif  // SKNNR_KERNELS_EXACT

// Row norms |r|^2 as one fma chain per row (SKL/.../_base.pyx.tp:20-42 uses ddot).
#ifdef SKNNR_KERNELS_EXACT
__global__ void __launch_bounds__(256)
row_norms_kernel(const double* __restrict__ x, long n, int d, double* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
    for (int c = 0; c < d; ++c) acc = fma(x[i * d + c], x[i * d + c], acc);
    out[i] = acc;
}
#endif  // SKNNR_KERNELS_EXACT

}  // namespace sknnr
