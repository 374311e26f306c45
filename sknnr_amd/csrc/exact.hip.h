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
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

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
struct PrepArgs {
    const double* x;       // (nq, d_in) query rows of this launch
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
    uint4* qimg;           // [nq_pad/32][2][ks][64] fragments out, or null (transform only)
    double* qnc;           // (nq) out
};

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
    const double* xsrc = a.x + q0 * a.d_in;
    for (long e = tid; e < n_el; e += BT) {
        const int r = (int)(e / a.d_in);
        const int c = (int)(e - (long)r * a.d_in);
        double v = xsrc[e];
        if (a.center) v = v - a.center[c];
        if (a.scale) v = v / a.scale[c];
        xs[r * ldx + c] = v;
    }
    __syncthreads();

    const long q = q0 + tid;
    const bool live = q < a.nq;
    const long qb = q >> 5;
    const int col = (int)(q & 31);
    const int dp = 16 * a.ks;
    const double* xrow = xs + tid * ldx;
    double qn = 0.0;
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
            qn = fma(b, b, qn);
            const _Float16 h = (_Float16)(float)b;
            hi[jj] = h;
            lo[jj] = (_Float16)(float)(b - (double)h);
        }
        if (q < a.nq_pad) {
            const int step = jc >> 1, hh = jc & 1;
            a.qimg[((size_t)(qb * 2 + 0) * a.ks + step) * 64 + hh * 32 + col] = __builtin_bit_cast(uint4, hi);
            a.qimg[((size_t)(qb * 2 + 1) * a.ks + step) * 64 + hh * 32 + col] = __builtin_bit_cast(uint4, lo);
        }
    }
    if (live && a.qnc) a.qnc[q] = qn;
}

// ---------------------------------------------------------------------------------------
// The reference's pair distance in float64.
// formula 0: |x|^2 + (-2 x.y) + |y|^2, clamped at 0   (_argkmin.pyx.tp:492-502)
// formula 1: sum (x - y)^2, mul and add separately rounded (_dist_metrics.pxd.tp:39-49)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double pair_d2(const double* __restrict__ x, const double* __restrict__ r,
                                          int d, double rn, int formula) {
    if (formula == 0) {
        double qn = 0.0, dot = 0.0;
        for (int c = 0; c < d; ++c) {
            const double xv = x[c];
            qn = fma(xv, xv, qn);
            dot = fma(xv, r[c], dot);
        }
        double d2 = qn + (-2.0 * dot) + rn;
        return d2 > 0.0 ? d2 : 0.0;
    }
    double acc = 0.0;
    for (int c = 0; c < d; ++c) {
        const double t = x[c] - r[c];
        acc = acc + t * t;  // -ffp-contract=off keeps the two roundings
    }
    return acc;
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
    const float* cand_val;  // [nq][2][M]
    const int* cand_idx;
    const double* qnc;      // (nq)
    double inv_s2;          // 1 / s^2
    double eps_c;           // certificate: eps = eps_c * (sqrt(qnc) + ymax)^2 (already * 2^-24)
    double ymax;            // max |s (r - mu)|
    int* fail_list;
    int* fail_count;
};

template <int LPQ>
__device__ __forceinline__ double group_min(double v) {
#pragma unroll
    for (int o = LPQ / 2; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, LPQ));
    return v;
}
template <int LPQ>
__device__ __forceinline__ double group_max(double v) {
#pragma unroll
    for (int o = LPQ / 2; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, LPQ));
    return v;
}
template <int LPQ>
__device__ __forceinline__ int group_min_i(int v) {
#pragma unroll
    for (int o = LPQ / 2; o > 0; o >>= 1) {
        const int w = __shfl_xor(v, o, LPQ);
        v = w < v ? w : v;
    }
    return v;
}

template <int M>
__global__ void __launch_bounds__(256) finalize_kernel(FinalizeArgs a) {
    constexpr int LPQ = 2 * M;
    const SelectArgs& s = a.s;
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    long q = gt / LPQ;
    const int c = (int)(gt % LPQ);
    const bool live = q < s.nq;
    if (!live) q = s.nq - 1;  // keep the lane for the shuffles; it writes nothing

    const int id = a.cand_idx[q * LPQ + c];
    const float cv = a.cand_val[q * LPQ + c];
    const bool valid = id >= 0 && id < s.n_ref;

    double d2 = INFINITY;
    if (valid) d2 = pair_d2(s.xq + q * s.d, s.ref + (long)id * s.d, s.d, s.rn[id], s.formula);
    const bool usable = valid && (d2 == d2) && d2 < INFINITY;
    if (!usable) d2 = INFINITY;
    const int key_id = usable ? id : (0x7fffff00 + c);  // unusable slots sort last, distinct

    // rank by (d2, index)
    int rank = 0, n_usable = 0;
#pragma unroll
    for (int j = 0; j < LPQ; ++j) {
        const double dj = __shfl(d2, j, LPQ);
        const int ij = __shfl(key_id, j, LPQ);
        rank += (dj < d2) || (dj == d2 && ij < key_id);
        n_usable += dj < INFINITY;
    }

    // certificate: every reference outside the lists has a float64 d2 above tau
    const double tau = group_min<LPQ>(rank >= s.kk - 1 ? d2 : INFINITY);
    const float t_last = (c % M == M - 1) ? (valid ? cv : INFINITY) : INFINITY;
    const double t_min = group_min<LPQ>((double)t_last);
    const double qn = a.qnc[q];
    const double nrm = sqrt(qn) + a.ymax;
    const double eps = a.eps_c * nrm * nrm;
    const double bound = (qn + t_min - eps) * a.inv_s2;
    const bool certified = (n_usable >= s.kk) && (tau < INFINITY) && (bound > tau);

    // X=None: drop the row's own index, or the first entry when it is absent
    // (SKL/neighbors/_base.py:936-963)
    int sel = rank;
    const long self_id = s.row_offset + q;
    if (s.exclude_self) {
        const bool is_self = usable && (long)id == self_id && rank < s.kk;
        int drop = group_min_i<LPQ>(is_self ? rank : 0x7fffffff);
        if (drop == 0x7fffffff) drop = 0;
        sel = rank == drop ? -1 : (rank > drop ? rank - 1 : rank);
    }
    const bool chosen = usable && sel >= 0 && sel < s.k;
    const double dist = sqrt(d2 > 0.0 ? d2 : 0.0);

    int pos = sel;
    if (s.deterministic) {
        // REF/src/sknnr/_base.py:166-175
        const double dmax = group_max<LPQ>(chosen ? dist : 0.0);
        const double row_scale = fmax(dmax, 1.0);
        const double k0 = chosen ? round_key(dist / row_scale, s.pow10, s.pow10_is_divisor) : INFINITY;
        long k1 = (long)id - self_id;
        k1 = k1 < 0 ? -k1 : k1;
        pos = 0;
#pragma unroll
        for (int j = 0; j < LPQ; ++j) {
            const double k0j = __shfl(k0, j, LPQ);
            const long k1j = __shfl(k1, j, LPQ);
            const int ij = __shfl(key_id, j, LPQ);
            const bool chj = __shfl((int)chosen, j, LPQ) != 0;
            const bool less = (k0j < k0) || (k0j == k0 && (k1j < k1 || (k1j == k1 && ij < key_id)));
            pos += chj && less;
        }
    }
    if (live && chosen) {
        if (s.out_dist) s.out_dist[q * s.k + pos] = dist;
        s.out_idx[q * s.k + pos] = id;
    }
    if (live && c == 0 && !certified) {
        const int slot = atomicAdd(a.fail_count, 1);
        a.fail_list[slot] = (int)q;
    }
}

// ---------------------------------------------------------------------------------------
// exact_scan_kernel: one workgroup per query, every reference in float64.  Used for the
// queries whose certificate failed and for calls outside the MFMA envelope (large k,
// very wide features).  Per-thread sorted lists in LDS ([slot][thread], conflict-free),
// then KK rounds of a workgroup-wide lexicographic arg-min.
// ---------------------------------------------------------------------------------------
struct ScanArgs {
    SelectArgs s;
    const int* list;   // query ids to process, or null = all 0..count-1
    const int* count;  // device count (with list), else null and s.nq is used
};

__global__ void exact_scan_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const SelectArgs& s = a.s;
    const int T = blockDim.x, tid = threadIdx.x;
    const int KK = s.kk;
    double* xs = (double*)smem_raw;               // d
    double* lv = xs + ((s.d + 1) & ~1);           // [KK][T]
    int* li = (int*)(lv + (size_t)KK * T);        // [KK][T]
    double* rv = (double*)(li + (size_t)KK * T + ((KK * T) & 1));  // [KK] merged values
    int* ri = (int*)(rv + KK);                    // [KK] merged ids
    double* wv = (double*)(ri + KK + (KK & 1));   // [T/64] per-wave winners
    int* wi = (int*)(wv + 4);                     // [T/64]
    int* wo = wi + 4;                             // [T/64]
    volatile int* win_owner_p = wo + 4;           // winner of the current round

    const long n_items = a.list ? (long)*a.count : s.nq;
    for (long f = blockIdx.x; f < n_items; f += gridDim.x) {
        const long q = a.list ? (long)a.list[f] : f;
        for (int c = tid; c < s.d; c += T) xs[c] = s.xq[q * s.d + c];
        for (int i = 0; i < KK; ++i) {
            lv[(size_t)i * T + tid] = INFINITY;
            li[(size_t)i * T + tid] = 0x7fffffff;
        }
        __syncthreads();

        for (int j = tid; j < s.n_ref; j += T) {
            const double d2 = pair_d2(xs, s.ref + (long)j * s.d, s.d, s.rn[j], s.formula);
            if (d2 < lv[(size_t)(KK - 1) * T + tid]) {
                int i = KK - 1;
                while (i > 0 && lv[(size_t)(i - 1) * T + tid] > d2) {
                    lv[(size_t)i * T + tid] = lv[(size_t)(i - 1) * T + tid];
                    li[(size_t)i * T + tid] = li[(size_t)(i - 1) * T + tid];
                    --i;
                }
                lv[(size_t)i * T + tid] = d2;
                li[(size_t)i * T + tid] = j;
            }
        }

        int head = 0;
        for (int round = 0; round < KK; ++round) {
            double v = head < KK ? lv[(size_t)head * T + tid] : INFINITY;
            int id = head < KK ? li[(size_t)head * T + tid] : 0x7fffffff;
            int owner = tid;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double v2 = __shfl_xor(v, o, 64);
                const int id2 = __shfl_xor(id, o, 64);
                const int ow2 = __shfl_xor(owner, o, 64);
                if (v2 < v || (v2 == v && id2 < id)) { v = v2; id = id2; owner = ow2; }
            }
            if ((tid & 63) == 0) { wv[tid >> 6] = v; wi[tid >> 6] = id; wo[tid >> 6] = owner; }
            __syncthreads();
            if (tid == 0) {
                double bv = wv[0]; int bi = wi[0], bo = wo[0];
                for (int w = 1; w < (T >> 6); ++w)
                    if (wv[w] < bv || (wv[w] == bv && wi[w] < bi)) { bv = wv[w]; bi = wi[w]; bo = wo[w]; }
                rv[round] = bv; ri[round] = bi; *win_owner_p = bo;
            }
            __syncthreads();
            if (tid == *win_owner_p) ++head;
        }

        if (tid == 0) {
            // drop self (X=None), sqrt, reorder: serial over <= KK entries
            const long self_id = s.row_offset + q;
            int drop = -1;
            if (s.exclude_self) {
                drop = 0;
                for (int i = 0; i < KK; ++i)
                    if ((long)ri[i] == self_id) { drop = i; break; }
            }
            int n = 0;
            for (int i = 0; i < KK; ++i) {
                if (i == drop) continue;
                rv[n] = sqrt(rv[i] > 0.0 ? rv[i] : 0.0);
                ri[n] = ri[i];
                ++n;
            }
            if (n > s.k) n = s.k;
            if (s.deterministic) {
                double dmax = 0.0;
                for (int i = 0; i < n; ++i) dmax = fmax(dmax, rv[i]);
                const double row_scale = fmax(dmax, 1.0);
                // insertion sort by (rounded, |idx - row|, idx); lv/li of this thread's column are free now
                for (int i = 1; i < n; ++i) {
                    const double dv = rv[i]; const int iv = ri[i];
                    const double k0 = round_key(dv / row_scale, s.pow10, s.pow10_is_divisor);
                    long k1 = (long)iv - self_id; k1 = k1 < 0 ? -k1 : k1;
                    int j = i - 1;
                    while (j >= 0) {
                        const double k0j = round_key(rv[j] / row_scale, s.pow10, s.pow10_is_divisor);
                        long k1j = (long)ri[j] - self_id; k1j = k1j < 0 ? -k1j : k1j;
                        const bool greater = (k0j > k0) || (k0j == k0 && (k1j > k1 || (k1j == k1 && ri[j] > iv)));
                        if (!greater) break;
                        rv[j + 1] = rv[j]; ri[j + 1] = ri[j];
                        --j;
                    }
                    rv[j + 1] = dv; ri[j + 1] = iv;
                }
            }
            for (int i = 0; i < n; ++i) {
                if (s.out_dist) s.out_dist[q * s.k + i] = rv[i];
                s.out_idx[q * s.k + i] = ri[i];
            }
        }
        __syncthreads();
    }
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

__global__ void __launch_bounds__(256) predict_kernel(PredictArgs a) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.nq * a.t) return;
    const long q = e / a.t;
    const int tt = (int)(e - q * a.t);
    const long* ids = a.idx + q * a.k;
    if (a.mode == 0) {
        // np.mean(_y[neigh_ind], axis=1): slices added in order, then one division
        double acc = a.y[ids[0] * a.t + tt];
        for (int i = 1; i < a.k; ++i) acc = acc + a.y[ids[i] * a.t + tt];
        a.out[e] = acc / (double)a.k;
        return;
    }
    const double* dd = a.dist ? a.dist + q * a.k : nullptr;
    const double* ww = a.w ? a.w + q * a.k : nullptr;
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

__global__ void __launch_bounds__(256)
crosswalk_kernel(const long* __restrict__ table, const long* __restrict__ idx, long n, long* __restrict__ out) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = table[idx[i]];
}

// Row norms |r|^2 as one fma chain per row (SKL/.../_base.pyx.tp:20-42 uses ddot).
__global__ void __launch_bounds__(256)
row_norms_kernel(const double* __restrict__ x, long n, int d, double* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
    for (int c = 0; c < d; ++c) acc = fma(x[i * d + c], x[i * d + c], acc);
    out[i] = acc;
}

}  // namespace sknnr
