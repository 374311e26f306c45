// bucket.hip.h -- query bucketing for the second-generation pre-filter (round 3).
//
// The pre-filter's cost beside the matrix work is its visits: a (32-reference tile x 32-query block) unit is
// visited when some value is below the query's running threshold, and the threshold of a query only gets
// tight once rows close to it have been swept.  With the reference image in an order that is unrelated to
// the query (round 2: by norm) a query takes ~20 hits on 50,000 rows (6 ln(N / seed), counters in
// profiles/r03_coarse_counters.txt).  Here the image is ordered by CELLS of a median-split tree over the
// leading principal axes of the reference rows (index build, host), every query row is assigned to its cell
// (cell_hist_kernel), the rows of a call are bucketed by cell (cell_scatter_kernel: a counting sort that yields
// a permutation, nothing is moved), so that the 1,024 rows of a workgroup come from the same neighbourhood,
// and the workgroup starts its sweep -- and takes its seed thresholds -- in the stages around its own cell.
// Simulated and measured: 9-11 hits per query instead of 20-23.
//
// Any assignment is CORRECT: the certificate (exact.hip.h, finalize_kernel) does not depend on the order of the
// sweep or on which rows share a q-block; the assignment only decides how early the thresholds get tight.
// It is computed in float32 for that reason.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sknnr {

constexpr int kCellMaxDepth = 6;                 // up to 64 cells
constexpr int kCellMax = 1 << kCellMaxDepth;
constexpr int kBucketBlock = 1024;               // threads per block of the counting sort
#ifndef SKNNR_BUCKET_CHUNKS
#define SKNNR_BUCKET_CHUNKS 32
#endif
constexpr int kBucketChunks = SKNNR_BUCKET_CHUNKS;  // rows per thread: a block sorts kBucketChunks x kBucketBlock rows

// The tree: `depth` leading principal axes (unit vectors, [depth][d]), the centre they are taken about and the split
// values, node `n` of level `l` at (1 << l) - 1 + n.  All in device memory.
struct CellTreeDev {
    const float* axes;
    const float* centre;
    const float* thr;
    int depth;  // 0: no tree
};

// Cell of one row given its coordinates z[l] along the axes.
__device__ __forceinline__ int cell_of(const float (&z)[kCellMaxDepth], const CellTreeDev& t) {
    int node = 0;
#pragma unroll
    for (int l = 0; l < kCellMaxDepth; ++l)
        if (l < t.depth) node = 2 * node + (z[l] >= t.thr[(1 << l) - 1 + node] ? 1 : 0);  // (NaN coordinates: the `false` branch)
    return node;
}

struct CellArgs {
    const double* xq;      // (nq, d) transformed query rows
    long nq;               // live rows
    long n_pad;            // positions of the permutation (>= nq): positions nq .. n_pad-1 map to themselves
    int d;
    int depth;             // tree levels: 2^depth cells
    const float* axes;     // [depth][d] principal axes (unit vectors)
    const float* centre;   // [d]
    const float* thr;      // [2^depth - 1] split values: node `n` of level `l` at (1 << l) - 1 + n
    unsigned char* cell;   // (n_pad) out: cell of every row (padding rows: the last cell)
    int* hist;             // [kCellMax] rows per cell (zeroed by the caller), then [kCellMax] cursors
    int* perm;             // (n_pad) out: position -> row
};

// Cell of every row, for calls whose rows no prep kernel has classified (prep_queries_direct_kernel does it on the
// transformed values it holds in registers).
#ifdef SKNNR_KERNELS_EXACT
__global__ void __launch_bounds__(256) cell_assign_kernel(CellArgs a) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q < a.nq) {
        const double* row = a.xq + q * a.d;
        float z[kCellMaxDepth];
#pragma unroll
        for (int l = 0; l < kCellMaxDepth; ++l) z[l] = 0.f;
        auto feed = [&](int k, double x) {
            const float v = (float)x - a.centre[k];
#pragma unroll
            for (int l = 0; l < kCellMaxDepth; ++l)
                if (l < a.depth) z[l] = fmaf(v, a.axes[l * a.d + k], z[l]);
        };
        int k = 0;
        if ((a.d & 3) == 0) {  // rows start 32-byte aligned: four 16-byte loads in flight per trip
            const double2* r2 = (const double2*)row;
            for (; k + 8 <= a.d; k += 8) {
                const double2 v0 = r2[k / 2], v1 = r2[k / 2 + 1], v2 = r2[k / 2 + 2], v3 = r2[k / 2 + 3];
                feed(k, v0.x); feed(k + 1, v0.y); feed(k + 2, v1.x); feed(k + 3, v1.y);
                feed(k + 4, v2.x); feed(k + 5, v2.y); feed(k + 6, v3.x); feed(k + 7, v3.y);
            }
        }
        for (; k < a.d; ++k) feed(k, row[k]);
        const CellTreeDev t{a.axes, a.centre, a.thr, a.depth};
        a.cell[q] = (unsigned char)cell_of(z, t);
    }
}
#endif  // SKNNR_KERNELS_EXACT

// Rows per cell.
#ifdef SKNNR_KERNELS_EXACT
__global__ void __launch_bounds__(kBucketBlock) cell_count_kernel(CellArgs a) {
    __shared__ int h[kCellMax];
    if (threadIdx.x < kCellMax) h[threadIdx.x] = 0;
    __syncthreads();
    // (kBucketChunks x 1024 rows per block: the 64 device-wide counters take one atomic per block and cell -- with one chunk per
    //  block those 625k atomics on 64 addresses were the kernel's time at 10M rows)
#pragma unroll
    for (int r = 0; r < kBucketChunks; ++r) {
        const long q = ((long)blockIdx.x * kBucketChunks + r) * kBucketBlock + threadIdx.x;
        if (q < a.nq) atomicAdd(&h[a.cell[q]], 1);
    }
    __syncthreads();
    if (threadIdx.x < kCellMax && h[threadIdx.x] != 0) atomicAdd(&a.hist[threadIdx.x], h[threadIdx.x]);
}
#endif  // SKNNR_KERNELS_EXACT


// Counting sort by cell: perm[first position of the cell + ticket] = row.  One returning atomic per (block, cell)
// reserves the block's range; inside the block the rows take tickets from an LDS counter.  The order inside a cell
// is not deterministic (and does not matter: see the header).
#ifdef SKNNR_KERNELS_EXACT
__global__ void __launch_bounds__(kBucketBlock) cell_scatter_kernel(CellArgs a) {
    __shared__ int cnt[kCellMax], base[kCellMax], first[kCellMax];
    const int n_cells = 1 << a.depth;
    if (threadIdx.x < kCellMax) cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) {  // exclusive prefix of the 64 cell totals (every block for itself: 64 scalar loads)
        int acc = 0;
        for (int c = 0; c < kCellMax; ++c) {
            first[c] = acc;
            acc += c < n_cells ? a.hist[c] : 0;
        }
    }
    __syncthreads();
    int c[kBucketChunks], ticket[kBucketChunks];
#pragma unroll
    for (int r = 0; r < kBucketChunks; ++r) {
        const long q = ((long)blockIdx.x * kBucketChunks + r) * kBucketBlock + threadIdx.x;
        c[r] = 0;
        ticket[r] = 0;
        if (q < a.nq) {
            c[r] = a.cell[q];
            ticket[r] = atomicAdd(&cnt[c[r]], 1);
        } else if (q < a.n_pad) {
            a.perm[q] = (int)q;  // padding rows keep their place behind the live ones
            a.cell[q] = (unsigned char)(n_cells - 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < kCellMax && cnt[threadIdx.x] != 0) base[threadIdx.x] = atomicAdd(&a.hist[kCellMax + threadIdx.x], cnt[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kBucketChunks; ++r) {
        const long q = ((long)blockIdx.x * kBucketChunks + r) * kBucketBlock + threadIdx.x;
        if (q < a.nq) a.perm[first[c[r]] + base[c[r]] + ticket[r]] = (int)q;
    }
}
#endif  // SKNNR_KERNELS_EXACT

}  // namespace sknnr
