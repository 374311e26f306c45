// sknnr_hip.hip -- the C ABI of include/sknnr_hip.h over the gfx950 kernels (host translation unit).
//
// Host side only: index construction (coarse image of the reference rows), workspace,
// the per-chunk launch sequence  prep -> coarse (MFMA) -> finalize -> exact_scan,
// staging for host buffers, error reporting.  No CPU path computes results: if a
// device call fails the function fails.
#include "../../include/sknnr_hip.h"

#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

// kernels live in their own translation units (k_*.hip: each defines SKNNR_KERNELS_* for the non-template kernels it owns);
// this one sees their argument structs and geometry constants only
#include "launch.hip.h"

using namespace sknnr;

// ----------------------------------------------------------------------------------------
// errors
// ----------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(SKNNR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                 \
    } while (0)

// ----------------------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------------------
namespace {

// IEEE binary16 from double, round-to-nearest-even, overflow -> inf.
uint16_t f64_to_f16_bits(double v) {
    const float f = (float)v;  // double rounding is harmless: lo is taken from the actual hi
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t absx = x & 0x7fffffffu;
    if (absx >= 0x7f800000u) return (uint16_t)(sign | (absx > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (absx >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  // rounds to >= 65520 -> inf
    if (absx < 0x33000001u) return (uint16_t)sign;               // < 2^-25 (or exactly) -> 0
    int exp = (int)(absx >> 23) - 127;
    uint32_t man = (absx & 0x7fffffu) | 0x800000u;  // 24-bit significand
    int shift;                                        // bits to drop
    uint32_t hexp;
    if (exp < -14) {  // subnormal half
        shift = 13 + (-14 - exp);
        hexp = 0;
    } else {
        shift = 13;
        hexp = (uint32_t)(exp + 15);
    }
    uint32_t q = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1u);
    const uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) ++q;
    uint32_t out;
    if (hexp == 0) out = q;  // q may reach 0x400 = smallest normal: encoding is continuous
    else out = ((hexp - 1) << 10) + q;  // q in [0x400, 0x800]; carry bumps the exponent
    return (uint16_t)(sign | out);
}

double f16_bits_to_f64(uint16_t h) {
    const int sign = (h >> 15) & 1;
    const int exp = (h >> 10) & 31;
    const int man = h & 1023;
    double v;
    if (exp == 0) v = std::ldexp((double)man, -24);
    else if (exp == 31) v = man ? std::numeric_limits<double>::quiet_NaN() : std::numeric_limits<double>::infinity();
    else v = std::ldexp((double)(man | 1024), exp - 25);
    return sign ? -v : v;
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }  // early returns (HIP_TRY) free what a function allocated
    hipError_t ensure(size_t count) {
        if (count <= n && p) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        hipError_t e = hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

constexpr int kMaxKs = 8;              // coarse path: d <= 128
constexpr long kChunkRows = 1L << 22;  // rows per chunk of host-side staging loops (transform entry point, X=None results)
// Query rows per device chunk of one call (each chunk: prep -> pre-filter -> finalise, padded to kRowQuantum).
// One launch for as many rows as a 4 GiB workspace holds, at most 2^24: every pre-filter launch ends with a
// partly filled last round of workgroups (a workgroup sweeps the whole image: ~1.3 ms at 50k rows), so fewer,
// larger launches waste less (10M x 50k x 32, k=5: 183 -> 188 Mq/s against chunks of 4M rows,
// scripts/chunk_probe.py).  SKNNR_CHUNK_ROWS overrides (A/B runs).
long chunk_rows(int ks, int m_list) {
    static const long forced = [] {
        const char* e = std::getenv("SKNNR_CHUNK_ROWS");
        const long r = e ? std::atol(e) : 0;
        return r >= kRowQuantum ? std::min<long>(r, 1L << 26) : 0L;
    }();
    if (forced) return forced;
    const long per_row = 64L * std::max(ks, 1) + 8 + 16L * std::max(m_list, 2);  // query image + |q'|^2 + two candidate lists
    const long rows = std::min<long>(1L << 24, (4L << 30) / per_row);
    return std::max<long>(kRowQuantum, rows / kRowQuantum * kRowQuantum);
}
constexpr int kScanMaxKK = 192;
// Error bound of the split contraction, in units of 2^-24 (|q'| + max|r'|)^2 -- derived in DESIGN.md
// section 2 from the measured arithmetic of v_mfma_f32_32x32x16_f16 (scripts/microbench/
// mfma_f16_numerics.hip, profiles/r02_mfma_f16_numerics.txt: per instruction two groups of eight exact
// products, each group aligned to its largest product and truncated 24 bits below it, each group sum
// added with one round-to-nearest-even):
//     6.01   dropped terms of the split  (lo.lo, and the residuals a - hi - lo, b - hi - lo)
//   + 3.51   truncation inside the groups  (7 x 2^-24 x sum |products|, sum <= 2 |q'||r'| (1 + 2^-10))
//   + 1.00   |r'|^2 rounded to f32 (the C operand)
//   + 6 ks   two roundings per instruction, 3 ks instructions, each <= 2^-24 x |partial value|
//   + 0.25   f16 subnormal floor of the lo parts
// = 10.77 + 6 ks  ->  11 + 6 ks.  Random operands reach 2.6 (d=8) .. 5.1 (d=100) units
// (tests/test_hip_parity.py::test_coarse_error_budget); adversarial ones (same-sign products at the
// image limits, full mantissas) are tested in test_coarse_error_budget_adversarial.
constexpr double eps_units(int ks) { return 11.0 + 6.0 * ks; }
// coarse2_kernel: only the ks main instructions round in the matrix pipe (2 ks roundings); the correction is a
// packed-f16 dot product in f32 whose own rounding is below 0.1 unit, added with one more rounding:
//   6.01 + 3.51 + 1.00 + (2 ks + 1) + 0.25 + 0.1 = 11.87 + 2 ks  ->  12 + 2 ks.
constexpr double eps_units2(int ks) { return 12.0 + 2.0 * ks; }

}  // namespace

// ----------------------------------------------------------------------------------------
// the handle
// ----------------------------------------------------------------------------------------
constexpr int kHostSlots = 4;  // tiles in flight in the host-buffer pipeline

// One background host thread that runs posted jobs in order (the host-buffer pipeline's copy-in and copy-out legs:
// the staging memcpys used to sit in the enqueueing thread, in series with it -- profiles/r02_host_path_probe.txt).
class HostWorker {
public:
    HostWorker() : th_([this] { run(); }) {}
    ~HostWorker() {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    std::future<int> post(std::function<int()> fn) {
        std::packaged_task<int()> task(std::move(fn));
        std::future<int> f = task.get_future();
        {
            std::lock_guard<std::mutex> l(m_);
            q_.emplace_back(std::move(task));
        }
        cv_.notify_one();
        return f;
    }

private:
    void run() {
        for (;;) {
            std::packaged_task<int()> task;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;  // stop requested and nothing left to do
                task = std::move(q_.front());
                q_.pop_front();
            }
            task();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::packaged_task<int()>> q_;
    bool stop_ = false;
    std::thread th_;  // (last member: the thread starts when everything above exists)
};

struct sknnr_index {
    int device = 0;
    long n_ref = 0;
    int d = 0, t = 0;
    int ks = 0;       // K-steps of the coarse image (0 = coarse path unavailable: d > 128)
    int n_stages = 0;
    double s = 1.0;   // coarse scale
    double ymax = 0.0;
    double mu_norm = 0.0;    // |mu| (upper bound), for the reference formula's rounding-noise term
    std::vector<double> mu;  // (16*ks) zero padded
    std::mutex mtx;          // one call at a time per handle (the workspace below is shared)

    // query-time affine map
    int d_in = 0;
    bool has_affine = false;
    DevBuf<double> center, scale, proj;  // proj padded to (d_in, 16*ks) when ks > 0 else (d_in, d)
    bool has_center = false, has_scale = false, has_proj = false;

    DevBuf<double> hw;       // weighted-Hamming weights (one per column), set by sknnr_index_set_hamming_weights
    double hw_sum = 0.0;
    bool has_hw = false;
    // integer pre-filter of the weighted-Hamming search (hamming.hip.h): 16-bit ids / weights, two trees per dword
    bool h16_ok = false;           // every reference id is an integer in [0, 65535]
    int h_tp = 0, h_ref_pad = 0;   // tree pairs; reference rows padded to the step of the kernel
    DevBuf<uint32_t> h_rimg, h_wq, h_qimg, h_rrow;  // (h_rrow: row-major ids for the re-score)
    DevBuf<int> h_bad, h_cand_cnt, h_cand_id;
    DevBuf<double> ref64, refT, rn64, y64, mu_dev;  // refT: (d, n_ref) transposed copy for the exact scan
    DevBuf<char> rimg;
    DevBuf<char> rhi2, rlo2;  // coarse2_kernel's image: [hi | |r'|^2] records for the LDS stages, lo fragments apart
    int n_stages2 = 0;
    DevBuf<int> perm;  // image position -> reference row (rows are imaged by increasing centred norm)
    // Second-generation image in CELL order (bucket.hip.h): a median-split tree over the leading principal axes
    DevBuf<int> perm2;             // its image position -> reference row
    int cell_depth = 0;            // 0: the second image is in the first one's order, no bucketing
    DevBuf<float> cell_axes, cell_centre, cell_thr;
    DevBuf<int> cell_stage;        // [2^depth] stage at which a workgroup of that cell starts its sweep
    DevBuf<unsigned char> qcell;   // workspace: cell of every query row of the chunk
    DevBuf<int> qperm, cell_hist;  // workspace: position -> row; [2][kCellMax] rows per cell / cursors

    // workspace (one chunk)
    DevBuf<double> xt, qnc, xstage, dist_stage, pred_stage;
    DevBuf<uint4> qimg;
    DevBuf<float> cand_val;
    DevBuf<int> cand_idx, fail_list, fail_count, fail_list2, slice_i;
    DevBuf<double> slice_v;  // sliced exact scans: the slice heaps (exact.hip.h, scan_slices)
    DevBuf<int> status;            // bit 0: a query value was NaN, bit 1: infinite (since the last poll)
    DevBuf<long long> fail_total;  // running count of certificate failures (device)
    DevBuf<long> idx_stage;

    // host-buffer pipeline: pinned staging + device staging, kHostSlots slots; three streams
    struct HostSlot {
        double* pin_x = nullptr;  size_t pin_x_n = 0;
        double* pin_d = nullptr;  size_t pin_d_n = 0;
        long* pin_i = nullptr;    size_t pin_i_n = 0;
        double* pin_p = nullptr;  size_t pin_p_n = 0;
        DevBuf<double> dev_x, dev_d, dev_p;
        DevBuf<long> dev_i;
        hipEvent_t ev_h2d = nullptr, ev_done = nullptr, ev_d2h = nullptr;
    } slot[kHostSlots];
    hipStream_t st_h2d = nullptr, st_run = nullptr, st_d2h = nullptr;
    std::unique_ptr<HostWorker> w_in, w_out;  // copy-in (look-ahead) and copy-out legs of the host pipeline

    // Workspace hand-over between calls on different streams: the last launch of a call records
    // ev_ws; the next call's stream waits for it before it touches the workspace.
    hipEvent_t ev_ws = nullptr;
    bool ws_busy = false;
    // The thin last round of the pre-filter (4-wave workgroups, one wave per SIMD) leaves most of every CU free: the
    // finaliser of the rows that are already done runs beside it on a side stream (fork after the bulk launch, join
    // before the exact scan).
    hipStream_t st_side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    long bulk_rows_done = 0;  // rows whose pre-filter was complete at ev_fork (0: no fork in the last launch)
    hipEvent_t ev_bulk_end = nullptr;  // the call record's end-of-pre-filter event; recorded at the fork when there is one
    bool stream_open = false;  // a sknnr_stream owns the host pipeline's slots

    // Device timing of calls (HIP events on the launch stream), resolved lazily by sknnr_get_stats:
    // a ring of call records so that several calls of one benchmark step are summed, not only the last.
    struct CallTiming {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        std::vector<std::pair<hipEvent_t, hipEvent_t>> coarse;
        size_t coarse_used = 0;
        long coarse_rows = 0;  // rows the timed pre-filter launches of this call processed
        bool pending = false;
    };
    static constexpr int kTimingRing = 64;
    CallTiming timing[kTimingRing];
    int timing_next = 0;
    sknnr_stats stats{};

    ~sknnr_index() {
        w_in.reset();   // (joins: no job may outlive the buffers below)
        w_out.reset();
        (void)hipSetDevice(device);
        for (auto* b : {&center, &scale, &proj, &ref64, &refT, &rn64, &y64, &mu_dev, &xt, &qnc, &xstage,
                        &dist_stage, &pred_stage})
            b->release();
        rimg.release();
        perm.release();
        perm2.release();
        cell_axes.release(); cell_centre.release(); cell_thr.release(); cell_stage.release();
        qcell.release(); qperm.release(); cell_hist.release();
        h_rimg.release(); h_wq.release(); h_qimg.release(); h_rrow.release(); h_bad.release(); h_cand_cnt.release(); h_cand_id.release();
        qimg.release();
        cand_val.release();
        cand_idx.release();
        fail_list.release();
        fail_count.release();
        fail_total.release();
        status.release();
        idx_stage.release();
        for (auto& sl : slot) {
            for (void* hp : {(void*)sl.pin_x, (void*)sl.pin_d, (void*)sl.pin_i, (void*)sl.pin_p})
                if (hp) (void)hipHostFree(hp);
            sl.dev_x.release(); sl.dev_d.release(); sl.dev_p.release(); sl.dev_i.release();
            for (hipEvent_t e : {sl.ev_h2d, sl.ev_done, sl.ev_d2h})
                if (e) (void)hipEventDestroy(e);
        }
        for (hipStream_t h : {st_h2d, st_run, st_d2h})
            if (h) (void)hipStreamDestroy(h);
        if (ev_ws) (void)hipEventDestroy(ev_ws);
        if (st_side) (void)hipStreamDestroy(st_side);
        for (hipEvent_t e : {ev_fork, ev_join})
            if (e) (void)hipEventDestroy(e);
        for (auto& ct : timing) {
            for (hipEvent_t e : {ct.e0, ct.e1})
                if (e) (void)hipEventDestroy(e);
            for (auto& pr : ct.coarse) {
                (void)hipEventDestroy(pr.first);
                (void)hipEventDestroy(pr.second);
            }
        }
    }
};

// ----------------------------------------------------------------------------------------
// misc entry points
// ----------------------------------------------------------------------------------------
extern "C" int32_t sknnr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int32_t sknnr_abi_version(void) { return SKNNR_ABI_VERSION; }

extern "C" const char* sknnr_last_error(void) { return g_last_error.c_str(); }

extern "C" int sknnr_index_shape(const sknnr_index* ix, int64_t* n_ref, int32_t* d, int32_t* t,
                                 int32_t* d_in, int32_t* device) {
    if (!ix) return fail(SKNNR_ERR_INVALID, "index is NULL");
    if (n_ref) *n_ref = ix->n_ref;
    if (d) *d = ix->d;
    if (t) *t = ix->t;
    if (d_in) *d_in = ix->has_affine ? ix->d_in : ix->d;
    if (device) *device = ix->device;
    return SKNNR_OK;
}

// ----------------------------------------------------------------------------------------
// index construction
// ----------------------------------------------------------------------------------------
// Decide the order of the reference image by replaying the pre-filter's visit rule on a sample:
// 64 pseudo-queries (a sampled row displaced by 0.35 x the difference of two others) sweep up to
// 8192 sampled rows in tiles of 32, once in the caller's order and once by increasing centred norm,
// keeping the 6 best per query; the order with fewer visited (tile, 32-query block) pairs wins, an
// order other than the caller's only by a clear margin.
// Squared Mahalanobis norms of the rows about `mu` (covariance from up to 65536 evenly spaced rows,
// a small ridge, Cholesky, one forward substitution per row): the density order of a correlated cloud.
// Skipped (returns false) when it would cost more than ~4e9 multiply-adds on the host.
static bool mahalanobis_norms(const double* ref, int64_t n_ref, int d, const std::vector<double>& mu,
                              std::vector<double>& out) {
    if ((double)n_ref * d * d > 8e9 || n_ref < 2 * (int64_t)d) return false;
    const int64_t n_cov = std::min<int64_t>(n_ref, 65536), stride = n_ref / n_cov;
    std::vector<double> cov((size_t)d * d, 0.0), v((size_t)d);
    for (int64_t i = 0; i < n_cov; ++i) {
        const double* r = ref + i * stride * d;
        for (int a = 0; a < d; ++a) v[(size_t)a] = r[a] - mu[(size_t)a];
        for (int a = 0; a < d; ++a)
            for (int b = 0; b <= a; ++b) cov[(size_t)a * d + b] += v[(size_t)a] * v[(size_t)b];
    }
    double tr = 0.0;
    for (int a = 0; a < d; ++a) {
        for (int b = 0; b <= a; ++b) cov[(size_t)a * d + b] /= (double)n_cov;
        tr += cov[(size_t)a * d + a];
    }
    if (!(tr > 0.0)) return false;
    for (int a = 0; a < d; ++a) cov[(size_t)a * d + a] += 1e-9 * tr / d;
    // in-place Cholesky, lower triangle
    for (int a = 0; a < d; ++a) {
        for (int b = 0; b <= a; ++b) {
            double sum = cov[(size_t)a * d + b];
            for (int k = 0; k < b; ++k) sum -= cov[(size_t)a * d + k] * cov[(size_t)b * d + k];
            if (a == b) {
                if (!(sum > 0.0)) return false;
                cov[(size_t)a * d + a] = std::sqrt(sum);
            } else {
                cov[(size_t)a * d + b] = sum / cov[(size_t)b * d + b];
            }
        }
    }
    out.resize((size_t)n_ref);
    const unsigned n_thr = std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < n_thr; ++t)
        th.emplace_back([&, t] {
            std::vector<double> z((size_t)d);
            for (int64_t i = t; i < n_ref; i += n_thr) {
                const double* r = ref + i * d;
                double m2 = 0.0;
                for (int a = 0; a < d; ++a) {
                    double sum = r[a] - mu[(size_t)a];
                    for (int k = 0; k < a; ++k) sum -= cov[(size_t)a * d + k] * z[(size_t)k];
                    z[(size_t)a] = sum / cov[(size_t)a * d + a];
                    m2 += z[(size_t)a] * z[(size_t)a];
                }
                out[(size_t)i] = m2;
            }
        });
    for (auto& t : th) t.join();
    return true;
}

// Replay of the pre-filter's visit rule on a sample, shared by the choices below: 64 pseudo-queries (a sampled row
// displaced by 0.35 x the difference of two others) sweep up to 8192 sampled rows in tiles of 32, keeping the 6 best
// per query; counted are the (tile, 32-query block) pairs in which some query of the block has a value below its
// threshold as of the start of the tile.
struct OrderSim {
    const double* ref;
    int64_t n_ref;
    int d, S, NQ = 64, J = 6;
    int64_t stride;
    std::vector<int> sample;
    std::vector<double> q;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    OrderSim(const double* ref_, int64_t n_ref_, int d_) : ref(ref_), n_ref(n_ref_), d(d_) {
        S = (int)std::min<int64_t>(n_ref, 8192);
        stride = n_ref / S;
        sample.resize(S);
        for (int i = 0; i < S; ++i) sample[i] = (int)(i * stride);
        q.resize((size_t)NQ * d);
        auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (int64_t)(rng % (uint64_t)n_ref); };
        for (int i = 0; i < NQ; ++i) {
            const int64_t a = next(), b = next(), c = next();
            for (int k = 0; k < d; ++k) q[(size_t)i * d + k] = ref[a * d + k] + 0.35 * (ref[b * d + k] - ref[c * d + k]);
        }
    }
    // `order`: S row ids; `start` (optional): per query, the position of `order` at which its sweep begins (rotation)
    long visits(const std::vector<int>& order, const std::vector<int>* start = nullptr) const {
        std::vector<double> best((size_t)NQ * J, std::numeric_limits<double>::infinity());
        long n_vis = 0;
        for (int t0 = 0; t0 + 32 <= S; t0 += 32) {
            for (int qb = 0; qb < NQ / 32; ++qb) {
                bool any = false;
                for (int qi = qb * 32; qi < qb * 32 + 32; ++qi) {
                    double* bq = &best[(size_t)qi * J];
                    const double thr = bq[J - 1];  // threshold as of the start of the tile, like the kernel's
                    const int base = start ? (*start)[(size_t)qi] : 0;
                    for (int r = t0; r < t0 + 32; ++r) {
                        const double* rr = ref + (int64_t)order[(size_t)((base + r) % S)] * d;
                        double d2 = 0.0;
                        for (int k = 0; k < d; ++k) {
                            const double t = q[(size_t)qi * d + k] - rr[k];
                            d2 += t * t;
                        }
                        if (d2 < thr) {
                            any = true;
                            if (d2 < bq[J - 1]) {
                                int p = J - 1;
                                while (p > 0 && bq[p - 1] > d2) { bq[p] = bq[p - 1]; --p; }
                                bq[p] = d2;
                            }
                        }
                    }
                }
                n_vis += any;
            }
        }
        return n_vis;
    }
};

// returns 0: the caller's order, 1: increasing centred norm, 2: a fixed pseudo-random shuffle (for callers
// whose rows are sorted by something that correlates with the features: the first tiles would then
// cover one corner of the cloud only)
// (... 3: increasing Mahalanobis norm, when `mnorm` is available); *best_visits: the winner's visit count
static int choose_image_order(OrderSim& sim, const std::vector<double>& cnorm, const std::vector<double>* mnorm, long* best_visits) {
    *best_visits = -1;
    if (std::getenv("SKNNR_IMAGE_ORDER")) return std::atoi(std::getenv("SKNNR_IMAGE_ORDER"));
    if (sim.n_ref < 2048) return 0;
    const int S = sim.S;
    std::vector<int> by_norm(sim.sample);
    std::stable_sort(by_norm.begin(), by_norm.end(), [&](int a, int b) { return cnorm[(size_t)a] < cnorm[(size_t)b]; });
    // the sample in a fixed pseudo-random order (a strided walk is already spread over the rows; the
    // shuffle removes what is left of the caller's ordering)
    std::vector<int> shuffled(sim.sample);
    uint64_t rng = sim.rng;
    for (int i = S - 1; i > 0; --i) {
        rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
        std::swap(shuffled[i], shuffled[(int)(rng % (uint64_t)(i + 1))]);
    }
    // caller's order on CONSECUTIVE rows (what the image would really hold tile by tile)
    std::vector<int> head(S);
    for (int i = 0; i < S; ++i) head[i] = i;
    const long v_orig = std::max(sim.visits(sim.sample), sim.visits(head)), v_norm = sim.visits(by_norm), v_shuf = sim.visits(shuffled);
    long best = v_orig;
    int choice = 0;
    if (v_shuf * 100 < best * 93) { best = v_shuf; choice = 2; }
    if (v_norm * 100 < best * 93) { best = v_norm; choice = 1; }
    if (mnorm) {
        std::vector<int> by_mah(sim.sample);
        std::stable_sort(by_mah.begin(), by_mah.end(), [&](int a, int b) { return (*mnorm)[(size_t)a] < (*mnorm)[(size_t)b]; });
        const long v_mah = sim.visits(by_mah);
        if (v_mah * 100 < best * (choice == 0 ? 93 : 97)) { best = v_mah; choice = 3; }
    }
    *best_visits = best;
    return choice;
}

// ----------------------------------------------------------------------------------------
// cell order of the second-generation image (bucket.hip.h)
// ----------------------------------------------------------------------------------------
namespace {

// Eigenvectors of a symmetric d x d matrix (cyclic Jacobi), sorted by decreasing eigenvalue: vec[i * d + k] = k-th
// component of the i-th vector.
void symmetric_eigen(std::vector<double> a, int d, std::vector<double>& val, std::vector<double>& vec) {
    std::vector<double> v((size_t)d * d, 0.0);
    for (int i = 0; i < d; ++i) v[(size_t)i * d + i] = 1.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < d; ++p)
            for (int q = p + 1; q < d; ++q) off += a[(size_t)p * d + q] * a[(size_t)p * d + q];
        if (off < 1e-22) break;
        for (int p = 0; p < d; ++p)
            for (int q = p + 1; q < d; ++q) {
                const double apq = a[(size_t)p * d + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (a[(size_t)q * d + q] - a[(size_t)p * d + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < d; ++k) {  // columns p, q
                    const double akp = a[(size_t)k * d + p], akq = a[(size_t)k * d + q];
                    a[(size_t)k * d + p] = c * akp - sn * akq;
                    a[(size_t)k * d + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < d; ++k) {  // rows p, q
                    const double apk = a[(size_t)p * d + k], aqk = a[(size_t)q * d + k];
                    a[(size_t)p * d + k] = c * apk - sn * aqk;
                    a[(size_t)q * d + k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < d; ++k) {  // accumulate the rotation (columns of v are the vectors)
                    const double vkp = v[(size_t)k * d + p], vkq = v[(size_t)k * d + q];
                    v[(size_t)k * d + p] = c * vkp - sn * vkq;
                    v[(size_t)k * d + q] = sn * vkp + c * vkq;
                }
            }
    }
    std::vector<int> order(d);
    for (int i = 0; i < d; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return a[(size_t)x * d + x] > a[(size_t)y * d + y]; });
    val.resize(d);
    vec.assign((size_t)d * d, 0.0);
    for (int i = 0; i < d; ++i) {
        val[i] = a[(size_t)order[i] * d + order[i]];
        for (int k = 0; k < d; ++k) vec[(size_t)i * d + k] = v[(size_t)k * d + order[i]];
    }
}

struct CellTree {
    int depth = 0;
    std::vector<float> axes, centre, thr;  // [depth][d], [d], [2^depth - 1]
    std::vector<int> code;                 // cell of every reference row
};

// Median-split tree over the `depth` leading principal axes of the centred reference rows: level l splits every
// node at the median of its rows' coordinate along axis l.  The reference rows are compared with the same float32
// arithmetic as the device uses for the queries (cell_hist_kernel), so a query that IS a reference row lands in
// that row's cell.
CellTree build_cell_tree(const double* ref, int64_t n_ref, int d, const std::vector<double>& mu, int depth) {
    CellTree t;
    t.depth = depth;
    const int64_t n_cov = std::min<int64_t>(n_ref, 65536), stride = n_ref / n_cov;
    std::vector<double> cov((size_t)d * d, 0.0), v((size_t)d);
    for (int64_t i = 0; i < n_cov; ++i) {
        const double* r = ref + i * stride * d;
        for (int a = 0; a < d; ++a) v[(size_t)a] = r[a] - mu[(size_t)a];
        for (int a = 0; a < d; ++a)
            for (int b = 0; b <= a; ++b) cov[(size_t)a * d + b] += v[(size_t)a] * v[(size_t)b];
    }
    for (int a = 0; a < d; ++a)
        for (int b = 0; b <= a; ++b) cov[(size_t)b * d + a] = cov[(size_t)a * d + b] = cov[(size_t)a * d + b] / (double)n_cov;
    std::vector<double> val, vec;
    symmetric_eigen(cov, d, val, vec);
    t.axes.resize((size_t)depth * d);
    t.centre.resize((size_t)d);
    for (int l = 0; l < depth; ++l)
        for (int k = 0; k < d; ++k) t.axes[(size_t)l * d + k] = (float)vec[(size_t)l * d + k];
    for (int k = 0; k < d; ++k) t.centre[(size_t)k] = (float)mu[(size_t)k];
    // coordinates of every row along the axes (the device's arithmetic: float32 fma chain over k)
    std::vector<float> z((size_t)n_ref * depth);
    for (int64_t i = 0; i < n_ref; ++i) {
        float acc[kCellMaxDepth] = {};
        for (int k = 0; k < d; ++k) {
            const float x = (float)ref[i * d + k] - t.centre[(size_t)k];
            for (int l = 0; l < depth; ++l) acc[l] = std::fmaf(x, t.axes[(size_t)l * d + k], acc[l]);
        }
        for (int l = 0; l < depth; ++l) z[(size_t)i * depth + l] = acc[l];
    }
    t.thr.assign(((size_t)1 << depth) - 1, 0.f);
    t.code.assign((size_t)n_ref, 0);
    std::vector<std::vector<int>> nodes(1);
    nodes[0].resize((size_t)n_ref);
    for (int64_t i = 0; i < n_ref; ++i) nodes[0][(size_t)i] = (int)i;
    for (int l = 0; l < depth; ++l) {
        std::vector<std::vector<int>> next(nodes.size() * 2);
        for (size_t n = 0; n < nodes.size(); ++n) {
            std::vector<int>& rows = nodes[n];
            float split = 0.f;
            if (!rows.empty()) {
                std::vector<float> zz(rows.size());
                for (size_t j = 0; j < rows.size(); ++j) zz[j] = z[(size_t)rows[j] * depth + l];
                std::nth_element(zz.begin(), zz.begin() + zz.size() / 2, zz.end());
                split = zz[zz.size() / 2];
            }
            t.thr[((size_t)1 << l) - 1 + n] = split;
            for (int r : rows) next[2 * n + (z[(size_t)r * depth + l] >= split ? 1 : 0)].push_back(r);
        }
        nodes.swap(next);
    }
    for (size_t n = 0; n < nodes.size(); ++n)
        for (int r : nodes[n]) t.code[(size_t)r] = (int)n;
    return t;
}

}  // namespace

extern "C" int sknnr_index_create(const double* ref, int64_t n_ref, int32_t d, const double* y,
                                  int32_t t, int32_t device, sknnr_index** out) {
    if (!out) return fail(SKNNR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!ref || n_ref < 1 || d < 1) return fail(SKNNR_ERR_INVALID, "ref must be a non-empty (n_ref, d) matrix");
    if (n_ref > 0x7fffff00L) return fail(SKNNR_ERR_UNSUPPORTED, "n_ref = %ld exceeds 2^31 - 256", (long)n_ref);
    if (y && t < 1) return fail(SKNNR_ERR_INVALID, "t must be >= 1 when y is given");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1)
        return fail(SKNNR_ERR_NO_DEVICE, "no HIP device is visible");
    if (device < 0 || device >= n_dev) return fail(SKNNR_ERR_INVALID, "device %d out of range [0, %d)", device, n_dev);
    HIP_TRY(hipSetDevice(device));
    {
        // every value finite (the reference's fit validates the same way: SKL/utils/validation.py, ensure_all_finite);
        // x - x is 0 for finite x and NaN for NaN and +-inf
        double bad = 0.0;
        const size_t total = (size_t)n_ref * (size_t)d;
        for (size_t i = 0; i < total; ++i) bad += ref[i] - ref[i];
        if (bad != 0.0) {
            for (size_t i = 0; i < total; ++i)
                if (ref[i] != ref[i]) return fail(SKNNR_ERR_NONFINITE, "Input X contains NaN.");
            return fail(SKNNR_ERR_NONFINITE, "Input X contains infinity or a value too large for dtype('float64').");
        }
    }

    sknnr_index* ix = new (std::nothrow) sknnr_index();
    if (!ix) return fail(SKNNR_ERR_INVALID, "out of host memory");
    struct Guard {
        sknnr_index* p;
        ~Guard() { delete p; }
    } guard{ix};
    ix->device = device;
    ix->n_ref = n_ref;
    ix->d = d;
    ix->t = y ? t : 0;

    const size_t nd = (size_t)n_ref * d;
    HIP_TRY(ix->ref64.ensure(nd));
    HIP_TRY(hipMemcpy(ix->ref64.p, ref, nd * sizeof(double), hipMemcpyHostToDevice));
    {
        std::vector<double> tr(nd);
        for (int64_t i = 0; i < n_ref; ++i)
            for (int c = 0; c < d; ++c) tr[(size_t)c * n_ref + i] = ref[(size_t)i * d + c];
        HIP_TRY(ix->refT.ensure(nd));
        HIP_TRY(hipMemcpy(ix->refT.p, tr.data(), nd * sizeof(double), hipMemcpyHostToDevice));
    }
    HIP_TRY(ix->rn64.ensure(n_ref));
    HIP_TRY(launch::row_norms(ix->ref64.p, n_ref, d, ix->rn64.p, nullptr));
    if (y) {
        HIP_TRY(ix->y64.ensure((size_t)n_ref * t));
        HIP_TRY(hipMemcpy(ix->y64.p, y, (size_t)n_ref * t * sizeof(double), hipMemcpyHostToDevice));
    }

    // ---- coarse image: centred, power-of-two scaled, split into f16 hi/lo, fragment order
    const int ks = (d + 15) / 16;
    if (ks <= kMaxKs) {
        ix->ks = ks;
        const int dp = 16 * ks;
        ix->mu.assign(dp, 0.0);
        for (int c = 0; c < d; ++c) {
            long double acc = 0;
            for (int64_t i = 0; i < n_ref; ++i) acc += ref[i * d + c];
            ix->mu[c] = (double)(acc / n_ref);
        }
        double amax = 0.0;
        for (int64_t i = 0; i < n_ref; ++i)
            for (int c = 0; c < d; ++c) amax = std::max(amax, std::fabs(ref[i * d + c] - ix->mu[c]));
        if (!(amax < std::numeric_limits<double>::infinity()))
            return fail(SKNNR_ERR_INVALID, "reference rows contain non-finite values");
        // A = -2 s (r - mu) must stay <= 256 in magnitude: 2 s amax <= 256
        int e = 0;
        if (amax > 0.0) {
            (void)std::frexp(amax, &e);  // amax = f * 2^e, f in [0.5, 1)
            e = 7 - e;                   // s * amax in [64, 128)
        }
        ix->s = std::ldexp(1.0, e);
        const double s = ix->s;

        // Image order: the caller's, or reference rows by increasing centred norm.  In a high-dimensional
        // cloud a row near the centre is, on average, closer to every query than a far one
        // (d2 = |q'|^2 + |r'|^2 - 2 q'.r'): the lists fill with good candidates early and later tiles
        // are visited less often (benchmark law, 32-D: 31.6 % -> 24.6 % of the tile x q-block tests;
        // 10M x 50k x 64: 83 -> 101 Mq/s).  In few dimensions the norms spread widely and the centre
        // rows are poor candidates for most queries (8-D, 100k rows: 193 -> 148 Mq/s), so the choice is
        // made per index by replaying the candidate orders on a sample (choose_image_order).  `perm` maps an
        // image position back to the caller's row index; only the finaliser needs it.
        std::vector<double> cnorm((size_t)n_ref);
        for (int64_t i = 0; i < n_ref; ++i) {
            double yn = 0.0;
            for (int c = 0; c < d; ++c) {
                const double b = s * (ref[i * d + c] - ix->mu[c]);
                yn += b * b;
            }
            cnorm[(size_t)i] = yn;
        }
        std::vector<int> perm((size_t)n_ref);
        for (int64_t i = 0; i < n_ref; ++i) perm[(size_t)i] = (int)i;
        std::vector<double> mnorm;
        const char* forced = std::getenv("SKNNR_IMAGE_ORDER");
        const bool have_mah = (!forced || std::atoi(forced) == 3) && n_ref >= 2048 && mahalanobis_norms(ref, n_ref, d, ix->mu, mnorm);
        OrderSim sim(ref, n_ref, d);
        long order_visits = -1;
        int image_order = choose_image_order(sim, cnorm, have_mah ? &mnorm : nullptr, &order_visits);
        if (image_order == 3 && !have_mah) image_order = 1;
        if (image_order == 1) {
            std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return cnorm[(size_t)a] < cnorm[(size_t)b]; });
        } else if (image_order == 3) {
            std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return mnorm[(size_t)a] < mnorm[(size_t)b]; });
        } else if (image_order == 2) {
            uint64_t rs = 0xD1B54A32D192ED03ull;
            for (int64_t i = n_ref - 1; i > 0; --i) {
                rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
                std::swap(perm[(size_t)i], perm[(size_t)(rs % (uint64_t)(i + 1))]);
            }
        }

        // One 32-reference tile of an image: hi / lo fragments in MFMA A-fragment order and the |r'|^2 block in
        // accumulator order, for the rows order[32 tile .. 32 tile + 31]
        auto fill_tile = [&](const std::vector<int>& order, long tile, uint16_t* frag_hi, uint16_t* frag_lo, float* ci) {
            for (int step = 0; step < ks; ++step)
                for (int lane = 0; lane < 64; ++lane) {
                    const long pos = tile * 32 + (lane & 31);
                    const long row = pos < n_ref ? order[(size_t)pos] : -1;
                    for (int j = 0; j < 8; ++j) {
                        const int k = step * 16 + 8 * (lane >> 5) + j;
                        double a = 0.0;
                        if (row >= 0 && k < d) a = -2.0 * s * (ref[row * d + k] - ix->mu[k]);
                        const uint16_t hi = f64_to_f16_bits(a);
                        frag_hi[((size_t)step * 64 + lane) * 8 + j] = hi;
                        frag_lo[((size_t)step * 64 + lane) * 8 + j] = f64_to_f16_bits(a - f16_bits_to_f64(hi));
                    }
                }
            for (int h = 0; h < 2; ++h)
                for (int r = 0; r < 16; ++r) {
                    const long pos = tile * 32 + acc_row(r, h);
                    ci[h * 16 + r] = pos < n_ref ? (float)cnorm[(size_t)order[(size_t)pos]] : std::numeric_limits<float>::infinity();
                }
        };
        double ymax2 = 0.0;
        for (int64_t i = 0; i < n_ref; ++i) ymax2 = std::max(ymax2, cnorm[(size_t)i]);
        ix->ymax = std::sqrt(ymax2);

        const int tps = tiles_per_stage(ks);
        const long n_tiles = ((n_ref + 31) / 32 + tps - 1) / tps * tps;
        ix->n_stages = (int)(n_tiles / tps);
        const size_t tb = tile_bytes(ks);
        std::vector<char> img(n_tiles * tb);
        for (long tile = 0; tile < n_tiles; ++tile) {
            char* rec = img.data() + tile * tb;
            fill_tile(perm, tile, reinterpret_cast<uint16_t*>(rec), reinterpret_cast<uint16_t*>(rec + (size_t)ks * 1024),
                      reinterpret_cast<float*>(rec + tile_frag_bytes(ks)));
        }
        if (ks <= 4) {
            // second-generation kernel: hi fragments + |r'|^2 per tile (staged through LDS), lo fragments in an
            // array of their own (read from L2 by the flush only)
            const int tps2 = tiles_per_stage2(ks);
            const long n_tiles2 = ((n_ref + 31) / 32 + tps2 - 1) / tps2 * tps2;
            ix->n_stages2 = (int)(n_tiles2 / tps2);
            // ... in CELL order when the kernel will serve this index (bucket.hip.h): rows sorted by the cell of a
            // median-split tree over the leading principal axes (inside a cell: by centred norm), so that a workgroup
            // whose query rows were bucketed by the same tree starts its sweep among their neighbours
            std::vector<int> perm2(perm);
            int depth = 0;
            {
                const char* e = std::getenv("SKNNR_CELLS");
                int want = e ? std::atoi(e) : kCellMaxDepth;
                want = std::min(want, std::min(kCellMaxDepth, (int)d));
                while (want > 0 && (n_ref >> want) < 512) --want;  // cells of at least 512 rows (one 16-tile stage)
                if (n_tiles2 >= 2 * kSeedTiles && want >= 2) depth = want;
            }
            const bool cells_forced = std::getenv("SKNNR_CELLS") != nullptr;
            CellTree tree;
            auto mix = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; };
            if (depth > 0) {
                tree = build_cell_tree(ref, n_ref, d, ix->mu, depth);
                if (!cells_forced && order_visits > 0) {
                    // Does the cell order pay for THIS reference set?  The same replay as for the other orders, every
                    // pseudo-query starting its sweep around its own cell.  Isotropic clouds (whitened Mahalanobis
                    // spaces, uniform hypercubes) have no leading axes to split along: they keep the order chosen above
                    // (measured, 10M x 50k x 64 Mahalanobis: 68 ms by norm, 91 ms by cells).
                    std::vector<int> by_cell(sim.sample);
                    std::stable_sort(by_cell.begin(), by_cell.end(), [&](int a, int b) {
                        if (tree.code[(size_t)a] != tree.code[(size_t)b]) return tree.code[(size_t)a] < tree.code[(size_t)b];
                        return mix((uint32_t)a) < mix((uint32_t)b);
                    });
                    const int n_cells = 1 << depth;
                    std::vector<int> first_s((size_t)n_cells + 1, sim.S);
                    for (int pos = sim.S - 1; pos >= 0; --pos) first_s[(size_t)tree.code[(size_t)by_cell[(size_t)pos]]] = pos;
                    for (int c = n_cells - 1; c >= 0; --c)
                        if (first_s[(size_t)c] == sim.S) first_s[(size_t)c] = first_s[(size_t)c + 1];
                    const int window = std::max<int>(32, (int)((int64_t)seed_tiles_for(n_tiles2, tps2) * 32 * sim.S / n_ref));
                    std::vector<int> start((size_t)sim.NQ);
                    for (int qi = 0; qi < sim.NQ; ++qi) {
                        int node = 0;
                        for (int l = 0; l < depth; ++l) {
                            float z = 0.f;
                            for (int k = 0; k < d; ++k)
                                z = std::fmaf((float)sim.q[(size_t)qi * d + k] - tree.centre[(size_t)k], tree.axes[(size_t)l * d + k], z);
                            node = 2 * node + (z >= tree.thr[((size_t)1 << l) - 1 + node] ? 1 : 0);
                        }
                        const int mid = (first_s[(size_t)node] + first_s[(size_t)node + 1]) / 2;
                        start[(size_t)qi] = ((mid - window / 2) % sim.S + sim.S) % sim.S;
                    }
                    const long v_cell = sim.visits(by_cell, &start);
                    if (v_cell * 100 >= order_visits * 85) depth = 0;
                    if (std::getenv("SKNNR_ORDER_TRACE"))
                        std::fprintf(stderr, "[order] best plain order %d: %ld visits; cell order (depth %d): %ld visits -> %s\n", image_order,
                                     order_visits, tree.depth, v_cell, depth ? "cells" : "plain");
                }
            }
            if (depth > 0) {
                for (int64_t i = 0; i < n_ref; ++i) perm2[(size_t)i] = (int)i;
                // (inside a cell: a fixed pseudo-random order -- sorted by norm, a query's nearest rows would share a few
                //  tiles and overflow the per-lane hit queues there)
                const bool by_norm = std::getenv("SKNNR_CELL_NORM_ORDER") != nullptr;
                std::stable_sort(perm2.begin(), perm2.end(), [&](int a, int b) {
                    if (tree.code[(size_t)a] != tree.code[(size_t)b]) return tree.code[(size_t)a] < tree.code[(size_t)b];
                    if (by_norm) return cnorm[(size_t)a] < cnorm[(size_t)b];
                    return mix((uint32_t)a) < mix((uint32_t)b);
                });
                const int n_cells = 1 << depth;
                std::vector<long> first((size_t)n_cells + 1, n_ref);
                for (int64_t pos = n_ref - 1; pos >= 0; --pos) first[(size_t)tree.code[(size_t)perm2[(size_t)pos]]] = pos;
                for (int c = n_cells - 1; c >= 0; --c)
                    if (first[(size_t)c] == n_ref) first[(size_t)c] = first[(size_t)c + 1];  // empty cell: its successor's place
                std::vector<int> stage((size_t)n_cells);
                for (int c = 0; c < n_cells; ++c) {
                    // the seed window (kSeedTiles tiles) is centred on the cell
                    const long mid_tile = (first[(size_t)c] + first[(size_t)c + 1]) / 2 / 32;
                    const long half = seed_tiles_for(n_tiles2, tps2) / 2;
                    long st = (mid_tile - half) / tps2;
                    if (mid_tile - half < 0) st = ix->n_stages2 + (mid_tile - half - tps2 + 1) / tps2;
                    stage[(size_t)c] = (int)(((st % ix->n_stages2) + ix->n_stages2) % ix->n_stages2);
                }
                HIP_TRY(ix->cell_axes.ensure(tree.axes.size()));
                HIP_TRY(hipMemcpy(ix->cell_axes.p, tree.axes.data(), tree.axes.size() * sizeof(float), hipMemcpyHostToDevice));
                HIP_TRY(ix->cell_centre.ensure(tree.centre.size()));
                HIP_TRY(hipMemcpy(ix->cell_centre.p, tree.centre.data(), tree.centre.size() * sizeof(float), hipMemcpyHostToDevice));
                HIP_TRY(ix->cell_thr.ensure(tree.thr.size()));
                HIP_TRY(hipMemcpy(ix->cell_thr.p, tree.thr.data(), tree.thr.size() * sizeof(float), hipMemcpyHostToDevice));
                HIP_TRY(ix->cell_stage.ensure(stage.size()));
                HIP_TRY(hipMemcpy(ix->cell_stage.p, stage.data(), stage.size() * sizeof(int), hipMemcpyHostToDevice));
                HIP_TRY(ix->cell_hist.ensure(2 * kCellMax));
                ix->cell_depth = depth;
            }
            const size_t tb2 = tile2_bytes(ks), lob = (size_t)ks * 1024;
            std::vector<char> hi2(n_tiles2 * tb2, 0), lo2(n_tiles2 * lob, 0);
            for (long tile = 0; tile < n_tiles2; ++tile)
                fill_tile(perm2, tile, reinterpret_cast<uint16_t*>(hi2.data() + tile * tb2),
                          reinterpret_cast<uint16_t*>(lo2.data() + tile * lob), reinterpret_cast<float*>(hi2.data() + tile * tb2 + lob));
            HIP_TRY(ix->rhi2.ensure(hi2.size()));
            HIP_TRY(hipMemcpy(ix->rhi2.p, hi2.data(), hi2.size(), hipMemcpyHostToDevice));
            HIP_TRY(ix->rlo2.ensure(lo2.size()));
            HIP_TRY(hipMemcpy(ix->rlo2.p, lo2.data(), lo2.size(), hipMemcpyHostToDevice));
            HIP_TRY(ix->perm2.ensure((size_t)n_ref));
            HIP_TRY(hipMemcpy(ix->perm2.p, perm2.data(), (size_t)n_ref * sizeof(int), hipMemcpyHostToDevice));
        }
        {
            double m2 = 0.0;
            for (int c = 0; c < d; ++c) m2 += ix->mu[c] * ix->mu[c];
            ix->mu_norm = std::sqrt(m2) * (1.0 + 1e-12);
        }
        HIP_TRY(ix->rimg.ensure(img.size()));
        HIP_TRY(hipMemcpy(ix->rimg.p, img.data(), img.size(), hipMemcpyHostToDevice));
        HIP_TRY(ix->perm.ensure((size_t)n_ref));
        HIP_TRY(hipMemcpy(ix->perm.p, perm.data(), (size_t)n_ref * sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY(ix->mu_dev.ensure(dp));
        HIP_TRY(hipMemcpy(ix->mu_dev.p, ix->mu.data(), dp * sizeof(double), hipMemcpyHostToDevice));
    }

    HIP_TRY(ix->fail_count.ensure(4));
    HIP_TRY(ix->fail_total.ensure(2));
    HIP_TRY(hipMemset(ix->fail_total.p, 0, 16));
    HIP_TRY(ix->status.ensure(4));
    HIP_TRY(hipMemset(ix->status.p, 0, 16));
    HIP_TRY(hipEventCreateWithFlags(&ix->ev_ws, hipEventDisableTiming));
    HIP_TRY(hipDeviceSynchronize());
    guard.p = nullptr;
    *out = ix;
    return SKNNR_OK;
}

extern "C" void sknnr_index_destroy(sknnr_index* ix) { delete ix; }

extern "C" int sknnr_index_set_affine(sknnr_index* ix, int32_t d_in, const double* center,
                                      const double* scale, const double* proj) {
    if (!ix) return fail(SKNNR_ERR_INVALID, "index is NULL");
    if (d_in < 1) return fail(SKNNR_ERR_INVALID, "d_in must be >= 1");
    if (!proj && d_in != ix->d)
        return fail(SKNNR_ERR_INVALID, "without a projection d_in (%d) must equal d (%d)", d_in, ix->d);
    std::lock_guard<std::mutex> lock(ix->mtx);
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipDeviceSynchronize());  // a call still running may be reading the previous map
    ix->d_in = d_in;
    ix->has_center = center != nullptr;
    ix->has_scale = scale != nullptr;
    ix->has_proj = proj != nullptr;
    if (center) {
        HIP_TRY(ix->center.ensure(d_in));
        HIP_TRY(hipMemcpy(ix->center.p, center, d_in * sizeof(double), hipMemcpyHostToDevice));
    }
    if (scale) {
        HIP_TRY(ix->scale.ensure(d_in));
        HIP_TRY(hipMemcpy(ix->scale.p, scale, d_in * sizeof(double), hipMemcpyHostToDevice));
    }
    if (proj) {
        const int dp = ix->ks > 0 ? 16 * ix->ks : ((ix->d + 15) / 16) * 16;
        std::vector<double> padded((size_t)d_in * dp, 0.0);
        for (int c = 0; c < d_in; ++c)
            for (int j = 0; j < ix->d; ++j) padded[(size_t)c * dp + j] = proj[(size_t)c * ix->d + j];
        HIP_TRY(ix->proj.ensure(padded.size()));
        HIP_TRY(hipMemcpy(ix->proj.p, padded.data(), padded.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    ix->has_affine = true;
    return SKNNR_OK;
}

extern "C" int sknnr_index_set_hamming_weights(sknnr_index* ix, const double* w, int32_t n) {
    if (!ix) return fail(SKNNR_ERR_INVALID, "index is NULL");
    if (!w || n != ix->d) return fail(SKNNR_ERR_INVALID, "w must hold one weight per column (%d), got %d", ix->d, n);
    double total = 0.0;
    for (int i = 0; i < n; ++i) {
        if (!(w[i] >= 0.0) || !(w[i] < std::numeric_limits<double>::infinity()))
            return fail(SKNNR_ERR_INVALID, "Hamming weights must be finite and non-negative");
        total += w[i];  // index order, as scipy's cdist accumulates it
    }
    if (!(total > 0.0)) return fail(SKNNR_ERR_INVALID, "Hamming weights must not sum to zero");
    std::lock_guard<std::mutex> lock(ix->mtx);
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(ix->hw.ensure(n));
    HIP_TRY(hipMemcpy(ix->hw.p, w, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    ix->hw_sum = total;
    ix->has_hw = true;
    // the integer image for the pre-filter (hamming.hip.h): reference ids as 16-bit integers, weights scaled to 16 bits
    ix->h16_ok = false;
    if (!std::getenv("SKNNR_HAMMING_INT") || std::atoi(std::getenv("SKNNR_HAMMING_INT")) != 0) {
        const int tp = (n + 1) / 2;
        const int ref_pad = (int)((ix->n_ref + kHamWaves * 64 - 1) / (kHamWaves * 64) * (kHamWaves * 64));
        double wmax = 0.0;
        for (int i = 0; i < n; ++i) wmax = std::max(wmax, w[i]);
        std::vector<uint32_t> wq((size_t)tp, 0u);
        for (int i = 0; i < n; ++i) {
            const uint32_t q16 = (uint32_t)std::min(65535.0, std::floor(w[i] * (65535.0 / wmax) + 0.5));
            wq[(size_t)i / 2] |= q16 << (16 * (i & 1));
        }
        HIP_TRY(ix->h_wq.ensure((size_t)tp));
        HIP_TRY(hipMemcpy(ix->h_wq.p, wq.data(), (size_t)tp * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(ix->h_rimg.ensure((size_t)tp * ref_pad));
        DevBuf<int> bad;
        HIP_TRY(bad.ensure((size_t)ref_pad));
        HIP_TRY(launch::hamming_pack(ix->ref64.p, ix->n_ref, ref_pad, n, tp, ix->h_rimg.p, bad.p, nullptr));
        std::vector<int> hb((size_t)ref_pad);
        HIP_TRY(hipMemcpy(hb.data(), bad.p, (size_t)ref_pad * sizeof(int), hipMemcpyDeviceToHost));
        bool any_bad = false;
        for (int v : hb) any_bad = any_bad || v != 0;
        const int tpr = ham_row_dwords(n);
        bool rows_ok = n <= 4096 && hamming_rescore_lds(n) <= 150 * 1024;
        if (rows_ok && ix->h_rrow.ensure((size_t)ix->n_ref * tpr) != hipSuccess) {
            // (the row-major image takes a KiB per row and 512 trees: when it does not fit, the float64 scan serves the index)
            (void)hipGetLastError();
            rows_ok = false;
        }
        if (rows_ok) {
            HIP_TRY(launch::hamming_rows(ix->ref64.p, ix->n_ref, n, tpr, ix->h_rrow.p, nullptr));
            HIP_TRY(hipDeviceSynchronize());
        }
        ix->h_tp = tp;
        ix->h_ref_pad = ref_pad;
        ix->h16_ok = !any_bad && rows_ok;  // (more than 4,096 trees: the float64 scan)
    }
    return SKNNR_OK;
}

extern "C" int sknnr_affine_transform(const double* x, int64_t n, int32_t d_in, const double* center,
                                      const double* scale, const double* proj, int32_t d, double* out,
                                      int32_t device) {
    if (!x || !out || n < 0 || d_in < 1 || d < 1) return fail(SKNNR_ERR_INVALID, "bad argument");
    if (!proj && d_in != d) return fail(SKNNR_ERR_INVALID, "without a projection d_in (%d) must equal d (%d)", d_in, d);
    if (n == 0) return SKNNR_OK;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) return fail(SKNNR_ERR_NO_DEVICE, "no HIP device is visible");
    if (device < 0 || device >= n_dev) return fail(SKNNR_ERR_INVALID, "device %d out of range [0, %d)", device, n_dev);
    HIP_TRY(hipSetDevice(device));
    const int ks = (d + 15) / 16, dp = 16 * ks;
    if ((size_t)64 * (d_in | 1) * 8 > 150 * 1024)
        return fail(SKNNR_ERR_UNSUPPORTED, "d_in = %d is too wide for the transform kernel (max 299)", d_in);
    DevBuf<double> dx, dc, dsc, dpj, dout;
    int rc = SKNNR_OK;
    auto cleanup = [&]() { dx.release(); dc.release(); dsc.release(); dpj.release(); dout.release(); };
#define AT_TRY(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            cleanup();                                                                                \
            return fail(SKNNR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));                \
        }                                                                                             \
    } while (0)
    std::vector<double> padded;
    if (proj) {
        padded.assign((size_t)d_in * dp, 0.0);
        for (int c = 0; c < d_in; ++c)
            for (int j = 0; j < d; ++j) padded[(size_t)c * dp + j] = proj[(size_t)c * d + j];
        AT_TRY(dpj.ensure(padded.size()));
        AT_TRY(hipMemcpy(dpj.p, padded.data(), padded.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (center) {
        AT_TRY(dc.ensure(d_in));
        AT_TRY(hipMemcpy(dc.p, center, d_in * sizeof(double), hipMemcpyHostToDevice));
    }
    if (scale) {
        AT_TRY(dsc.ensure(d_in));
        AT_TRY(hipMemcpy(dsc.p, scale, d_in * sizeof(double), hipMemcpyHostToDevice));
    }
    const long rows = std::min<long>(n, kChunkRows);
    AT_TRY(dx.ensure((size_t)rows * d_in));
    AT_TRY(dout.ensure((size_t)rows * d));
    for (long c0 = 0; c0 < n; c0 += rows) {
        const long m = std::min<long>(rows, n - c0);
        AT_TRY(hipMemcpy(dx.p, x + c0 * d_in, (size_t)m * d_in * sizeof(double), hipMemcpyHostToDevice));
        PrepArgs a{};
        a.x = dx.p;
        a.nq = m;
        a.nq_pad = (m + 255) / 256 * 256;
        a.d_in = d_in;
        a.d = d;
        a.ks = ks;
        a.center = center ? dc.p : nullptr;
        a.scale = scale ? dsc.p : nullptr;
        a.proj = proj ? dpj.p : nullptr;
        a.xt = dout.p;
        const int ldx = d_in | 1;
        const int bt = (size_t)256 * ldx * 8 <= 150 * 1024 ? 256 : ((size_t)128 * ldx * 8 <= 150 * 1024 ? 128 : 64);
        AT_TRY(launch::prep_lds(bt, a, nullptr));
        AT_TRY(hipMemcpy(out + c0 * d, dout.p, (size_t)m * d * sizeof(double), hipMemcpyDeviceToHost));
    }
#undef AT_TRY
    cleanup();
    return rc;
}

// ----------------------------------------------------------------------------------------
// stats
// ----------------------------------------------------------------------------------------
// Fold one finished call record into the stats (waits for the call if it is still running).
static void resolve_call(sknnr_index* ix, sknnr_index::CallTiming& ct) {
    if (!ct.pending) return;
    ct.pending = false;
    if (hipEventSynchronize(ct.e1) != hipSuccess) return;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ct.e0, ct.e1) != hipSuccess) return;
    double cms = 0.0;
    for (size_t i = 0; i < ct.coarse_used; ++i) {
        float m = 0.f;
        if (hipEventElapsedTime(&m, ct.coarse[i].first, ct.coarse[i].second) == hipSuccess) cms += m;
    }
    ix->stats.last_kernel_ms = ms;
    ix->stats.last_coarse_ms = cms;
    ix->stats.total_kernel_ms += ms;
    ix->stats.total_coarse_ms += cms;
    ix->stats.timed_calls += 1;
    ix->stats.coarse_rows_timed += ct.coarse_rows;
}
static void resolve_timing(sknnr_index* ix) {
    // oldest first, so that last_* end up describing the newest call
    for (int i = 0; i < sknnr_index::kTimingRing; ++i)
        resolve_call(ix, ix->timing[(ix->timing_next + i) % sknnr_index::kTimingRing]);
}

extern "C" int sknnr_get_stats(const sknnr_index* cix, sknnr_stats* out) {
    if (!cix || !out) return fail(SKNNR_ERR_INVALID, "NULL argument");
    sknnr_index* ix = const_cast<sknnr_index*>(cix);
    std::lock_guard<std::mutex> lock(ix->mtx);
    (void)hipSetDevice(ix->device);
    resolve_timing(ix);
    long long total = 0;
    if (hipMemcpy(&total, ix->fail_total.p, sizeof total, hipMemcpyDeviceToHost) == hipSuccess)
        ix->stats.exact_fallbacks = total;
    *out = ix->stats;
    launch::coarse1_dev_report();  // (development builds only: -DSKNNR_COARSE_COUNTERS / -DSKNNR_COARSE_TIMERS)
    launch::coarse2_dev_report();
    return SKNNR_OK;
}

extern "C" int sknnr_reset_stats(sknnr_index* ix) {
    if (!ix) return fail(SKNNR_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lock(ix->mtx);
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(ix->fail_total.p, 0, 16));
    for (auto& ct : ix->timing) ct.pending = false;
    ix->stats = sknnr_stats{};
    return SKNNR_OK;
}

static int nonfinite_error(int bits) {
    if (bits & 1) return fail(SKNNR_ERR_NONFINITE, "Input X contains NaN.");
    return fail(SKNNR_ERR_NONFINITE, "Input X contains infinity or a value too large for dtype('float64').");
}

// Read and clear the handle's non-finite flag; the caller has synchronised the stream that set it.
static int poll_status(sknnr_index* ix) {
    int bits = 0;
    HIP_TRY(hipMemcpy(&bits, ix->status.p, sizeof bits, hipMemcpyDeviceToHost));
    if (!bits) return SKNNR_OK;
    HIP_TRY(hipMemset(ix->status.p, 0, 16));
    return nonfinite_error(bits);
}

extern "C" int sknnr_check_finite(sknnr_index* ix, void* stream) {
    if (!ix) return fail(SKNNR_ERR_INVALID, "index is NULL");
    std::lock_guard<std::mutex> lock(ix->mtx);
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return poll_status(ix);
}

// ----------------------------------------------------------------------------------------
// launch helpers
// ----------------------------------------------------------------------------------------
namespace {

// `cells`: also name every row's cell (query bucketing); *cells_done tells whether this kernel did (the
// register-resident one does; otherwise the caller runs cell_assign_kernel on the transformed rows)
int launch_prep(sknnr_index* ix, const void* x, long nq, long nq_pad, bool affine, double* xt,
                hipStream_t st, bool check_finite = false, bool cells = false, bool* cells_done = nullptr, int x_dtype = kDtypeF64) {
    PrepArgs a{};
    a.x_dtype = x_dtype;
    if (cells_done) *cells_done = false;
    a.status = check_finite ? ix->status.p : nullptr;
    a.x = x;
    a.nq = nq;
    a.nq_pad = nq_pad;
    a.d_in = affine ? ix->d_in : ix->d;
    a.d = ix->d;
    a.ks = ix->ks;
    a.center = (affine && ix->has_center) ? ix->center.p : nullptr;
    a.scale = (affine && ix->has_scale) ? ix->scale.p : nullptr;
    a.proj = (affine && ix->has_proj) ? ix->proj.p : nullptr;
    a.mu = ix->mu_dev.p;
    a.s = ix->s;
    a.xt = xt;
    a.qimg = ix->qimg.p;
    a.qnc = ix->qnc.p;
    const int ldx = a.d_in | 1;
    const size_t lim = 150 * 1024;
    if (ix->ks <= 4 && !std::getenv("SKNNR_PREP_LDS")) {
        // narrow feature spaces: register-resident kernel (no LDS, high occupancy)
        if (cells && ix->cell_depth > 0) {
            a.tree = CellTreeDev{ix->cell_axes.p, ix->cell_centre.p, ix->cell_thr.p, ix->cell_depth};
            a.cell = ix->qcell.p;
            if (cells_done) *cells_done = true;
        }
        HIP_TRY(launch::prep_direct(a, st));
    } else if ((size_t)256 * ldx * 8 <= lim) {
        HIP_TRY(launch::prep_lds(256, a, st));
    } else if ((size_t)128 * ldx * 8 <= lim) {
        HIP_TRY(launch::prep_lds(128, a, st));
    } else if ((size_t)64 * ldx * 8 <= lim) {
        HIP_TRY(launch::prep_lds(64, a, st));
    } else {
        return fail(SKNNR_ERR_UNSUPPORTED, "d_in = %d is too wide for the query preparation kernel (max 299)", a.d_in);
    }
    return SKNNR_OK;
}

// first-generation pre-filter (k_coarse1.hip)
int launch_coarse(sknnr_index* ix, long nq_pad, int m_list, int kk, hipStream_t st) {
    launch::Coarse1Launch L{ix->rimg.p, ix->n_stages, ix->qimg.p, ix->qnc.p, (float)(std::ldexp(1.0, -9) * ix->ymax * 1.02),
                            m_list - (kk + 1), ix->cand_val.p, ix->cand_idx.p, nq_pad};
    hipError_t e = hipSuccess;
    if (launch::coarse1(ix->ks, m_list, L, st, &e) == launch::kNoInstance)
        return fail(SKNNR_ERR_UNSUPPORTED, "no coarse kernel for ks = %d, list length %d", ix->ks, m_list);
    HIP_TRY(e);
    return SKNNR_OK;
}

constexpr int kCoarse2TailWaves = 4;  // workgroup size (waves) of the thin-round variant
constexpr int kCusPerDevice = 256;

int launch_coarse2_waves(sknnr_index* ix, int m_list, int waves, long row0, long rows, int kk, hipStream_t st) {
    // more neighbours than a list holds: thresholds of rank M + E, no sentinels (coarse2_rank_extra)
    const int extra = coarse2_rank_extra(m_list, kk);
    if (extra != 0 && !((m_list == 16 && kk <= kCoarse2MaxKK16) || (m_list == 12 && kk <= kCoarse2MaxKK12) || (m_list == 8 && kk <= kCoarse2MaxKK8) || (m_list == 6 && kk <= kCoarse2MaxKK6)))
        return fail(SKNNR_ERR_UNSUPPORTED, "lists of %d cannot serve %d neighbours", m_list, kk);
    // positions [row0, row0 + rows) of the chunk (bucketed calls: position -> row through qperm, else the row itself)
    const bool bucketed = ix->cell_depth > 0;
    launch::Coarse2Launch L{ix->rhi2.p, ix->rlo2.p, ix->n_stages2, ix->qimg.p, ix->qnc.p, (float)(std::ldexp(1.0, -9) * ix->ymax * 1.02),
                            extra != 0 ? 0 : m_list - (kk + 1), ix->cand_val.p, ix->cand_idx.p, (int)row0,
                            bucketed ? ix->qperm.p : nullptr, bucketed ? ix->qcell.p : nullptr,
                            bucketed ? ix->cell_stage.p : nullptr, rows};
    hipError_t e = hipSuccess;
    if (launch::coarse2(ix->ks, m_list, waves, extra, L, st, &e) == launch::kNoInstance)
        return fail(SKNNR_ERR_UNSUPPORTED, "no second-generation coarse kernel for ks = %d, list length %d, %d waves, rank + %d", ix->ks,
                    m_list, waves, extra);
    HIP_TRY(e);
    return SKNNR_OK;
}

// A workgroup of 16 waves sweeps the whole image for its 1024 rows (~1.2 ms at 50k reference rows), so the
// last round of a launch leaves most CUs idle when it holds few workgroups (10M rows: 9768 workgroups =
// 38 rounds + 40), and a small call never fills the device.  Rows of such a thin round (at most a quarter of
// the CUs' worth) go to 4-wave workgroups instead: four times as many CUs, one wave per SIMD.
// `rows`: the live rows of the chunk -- its padding rows (up to kRowQuantum - 1 of them, behind the live ones also in a
// bucketed call) get no workgroups of their own: a 262,144-row call is 256 workgroups, not 258 with a thin round of two.
int launch_coarse2(sknnr_index* ix, long rows, int m_list, int kk, hipStream_t st) {
    if (!coarse2_supported(ix->ks, m_list))
        return fail(SKNNR_ERR_UNSUPPORTED, "no second-generation coarse kernel for ks = %d, list length %d", ix->ks, m_list);
    const int BULK_WAVES = coarse2_waves(ix->ks, m_list);
    const long QPB = BULK_WAVES * kCoarse2Nqb * 32;
    const long QPB_TAIL = kCoarse2TailWaves * kCoarse2Nqb * 32;
    static const bool split = [] {
        const char* e = std::getenv("SKNNR_COARSE_TAIL");
        return !(e && std::atoi(e) == 0);
    }();
    const long n_wg = (rows + QPB - 1) / QPB;
    long tail_wg = n_wg % kCusPerDevice;
    if (!split || tail_wg > kCusPerDevice / (BULK_WAVES / kCoarse2TailWaves)) tail_wg = 0;
    const long bulk_rows = (n_wg - tail_wg) * QPB;
    ix->bulk_rows_done = 0;
    if (bulk_rows > 0) {
        int rc = launch_coarse2_waves(ix, m_list, BULK_WAVES, 0, bulk_rows, kk, st);
        if (rc) return rc;
        if (tail_wg > 0 && ix->ev_fork) {  // the caller finalises these rows beside the thin round
            if (ix->ev_bulk_end) {  // the timed region ends here: the thin round shares the device from now on
                HIP_TRY(hipEventRecord(ix->ev_bulk_end, st));
                ix->ev_bulk_end = nullptr;
            }
            HIP_TRY(hipEventRecord(ix->ev_fork, st));
            ix->bulk_rows_done = bulk_rows;
        }
    }
    if (tail_wg > 0) {
        const long tail_rows = std::min(tail_wg * QPB, (rows - bulk_rows + QPB_TAIL - 1) / QPB_TAIL * QPB_TAIL);
        return launch_coarse2_waves(ix, m_list, kCoarse2TailWaves, bulk_rows, tail_rows, kk, st);
    }
    return SKNNR_OK;
}

// The second-generation kernel serves the shapes of coarse2_supported() (SKNNR_COARSE_V2=0 selects the first one).
bool use_coarse2(const sknnr_index* ix, int m_list) {
    static const bool enabled = [] {
        const char* e = std::getenv("SKNNR_COARSE_V2");
        return !(e && std::atoi(e) == 0);
    }();
    // (small reference sets keep the first kernel: the second one needs its seeding pass -- without a starting
    // threshold the first tiles give every lane more hits per unit than its queue holds)
    // (and the flush addresses the image with 32-bit offsets)
    const long tiles2 = (long)ix->n_stages2 * tiles_per_stage2(ix->ks);
    // (... and its flush tags a queued entry's image position, below 2^26, with the query column)
    return enabled && tiles2 >= 2 * kSeedTiles && tiles2 * tile2_bytes(ix->ks) < (1L << 32) && tiles2 * 32 < (1L << 26) &&
           coarse2_supported(ix->ks, m_list);
}

// List length per lane for kk neighbours searched: at least one spare slot keeps the certificate
// cheap.  kk > 31 is outside the MFMA envelope (exact scan for the whole call).
constexpr int kCoarseMaxKK = 31;
int coarse_list_len(int kk) { return kk <= 1 ? 2 : (kk <= 5 ? 6 : (kk <= 7 ? 8 : (kk <= 15 ? 16 : 32))); }
// ... for this handle: where the second-generation kernel serves them, 6 .. 7 neighbours keep lists of 6, 8 .. 15 lists of
// 8, 16 .. 23 lists of 12 and 24 .. 31 lists of 16 (up to 64 features), with thresholds of a rank beyond one list over the two lists of a query kept as one pool
// (coarse2.hip.h, pair_union_rank, coarse2_rank_extra) -- shorter lists are cheaper to keep, lists of 8 run with 16 waves
// per CU and no spills where lists of 16 need 12 waves, and lists of 32 exist on the first-generation kernel only.
int coarse_list_len(const sknnr_index* ix, int kk) {
    static const int enabled = [] {
        const char* e = std::getenv("SKNNR_V2_BIG_K");  // 0: off, 1: lists of 16 only, 2: and lists of 8, 3: and 6 .. 7 on lists of 6, default: all
        return e ? std::atoi(e) : 4;
    }();
    // (6 .. 7 neighbours: lists of 8 hold them where that kernel exists -- measured equal, 169 vs 169 Mq/s at 10M x 50k x 32,
    //  k = 7 -- and lists of 6 serve four K-steps, where lists of 8 would spill: 84 -> 95 Mq/s at 64 features)
    if (enabled >= 3 && kk > 5 && kk <= kCoarse2MaxKK6 && !use_coarse2(ix, 8) && use_coarse2(ix, 6)) return 6;
    if (enabled >= 2 && kk > 7 && kk <= kCoarse2MaxKK8 && use_coarse2(ix, 8)) return 8;
    if (enabled >= 4 && kk > 15 && kk <= kCoarse2MaxKK12 && use_coarse2(ix, 12)) return 12;
    if (enabled >= 1 && kk > 15 && kk <= kCoarse2MaxKK16 && use_coarse2(ix, 16)) return 16;
    return coarse_list_len(kk);
}
int coarse_rank_extra(int m_list, int kk) { return coarse2_rank_extra(m_list, kk); }

void launch_finalize(const FinalizeArgs& f, long n, hipStream_t st) { (void)launch::finalize(f, n, st); }

constexpr int kScanGridWg = 256 * 4;  // workgroups of a scan launch (also what scan_slices splits among the passes)

int launch_scan_formula(sknnr_index* ix, const ScanArgs& a0, long max_items, bool chunked, size_t sh, hipStream_t st) {
    const SelectArgs& s = a0.s;
    const int nq_pass = scan_nq(s.formula);
    ScanArgs a = a0;
    // Few queries: their passes would leave most of the device idle while each sweeps every reference row
    // (0.5 ms at 50k rows whatever the count) -- the kernel then splits the rows of a pass over several
    // workgroups and scan_merge_kernel combines the slice heaps.  The count is on the device when the queries
    // come from the fail list, so the decision is taken there; the host only provides the buffers.
    const bool may_slice = s.kk <= kScanSliceMaxKK && scan_slices(1, nq_pass, s.n_ref, s.kk, kScanGridWg) > 1;
    if (may_slice) {
        // at most kScanGridWg / 2 passes are sliced: (passes * nq_pass) slots x S slices <= kScanGridWg * nq_pass heaps
        const size_t heaps = (size_t)kScanGridWg * nq_pass * s.kk;
        HIP_TRY(ix->slice_v.ensure(heaps));
        HIP_TRY(ix->slice_i.ensure(heaps));
        HIP_TRY(ix->fail_list2.ensure((size_t)kScanGridWg * nq_pass));
        HIP_TRY(hipMemsetAsync(ix->fail_count.p + 2, 0, sizeof(int), st));
        a.slice_v = ix->slice_v.p;
        a.slice_i = ix->slice_i.p;
        a.list2 = ix->fail_list2.p;
        a.count2 = ix->fail_count.p + 2;
    }
    const long passes = (max_items + nq_pass - 1) / nq_pass;
    // (sliced mode needs the whole grid even for one pass; otherwise one workgroup per pass is enough)
    const long blocks = may_slice ? kScanGridWg : std::max<long>(1, std::min<long>(passes, kScanGridWg));
    HIP_TRY(launch::exact_scan(s.formula, chunked, a, blocks, sh, st));
    if (!may_slice) return SKNNR_OK;
    // one wave per sliced query (none if the scan was not sliced: the kernel returns at once)
    const long sliced_max = std::min<long>(max_items, (long)kScanGridWg * nq_pass / 2);
    const int kkp = (s.kk + 2) & ~1, stk = (2 * s.kk + 4 + 1) & ~1;
    const size_t msh = 4 * ((size_t)12 * kkp + (size_t)4 * stk);
    HIP_TRY(launch::scan_merge(s.formula, a, (sliced_max + 3) / 4, msh, (int)blocks, 0, st));
    // queries whose merged heaps are not unique (exact ties): the sequential scan, never sliced
    ScanArgs b = a0;
    b.list = ix->fail_list2.p;
    b.count = ix->fail_count.p + 2;
    b.slice_v = nullptr;
    b.slice_i = nullptr;
    const long blocks2 = std::max<long>(1, std::min<long>((sliced_max + nq_pass - 1) / nq_pass, kScanGridWg));
    HIP_TRY(launch::exact_scan(s.formula, chunked, b, blocks2, sh, st));
    return SKNNR_OK;
}

int launch_scan(sknnr_index* ix, const SelectArgs& s, const int* list, const int* count, long max_items,
                hipStream_t st) {
    const size_t sh = scan_block_bytes(s.d, s.kk, s.formula);
    if (sh > 150 * 1024)
        return fail(SKNNR_ERR_UNSUPPORTED, "n_neighbors = %d with d = %d does not fit the exact scan kernel", s.k, s.d);
    ScanArgs a{s, ix->refT.p, list, count, nullptr, nullptr, nullptr, nullptr};
    const bool chunked = s.d > kScanColChunk;  // wide rows (tree node ids) are swept in column chunks
    return launch_scan_formula(ix, a, max_items, chunked, sh, st);
}

struct CallCtx {
    sknnr_index* ix;
    const sknnr_query_opts* o;
    int kk;
    bool coarse;
};

int validate_call(sknnr_index* ix, const void* q, int64_t nq, const sknnr_query_opts* o, int64_t* out_idx) {
    if (!ix) return fail(SKNNR_ERR_INVALID, "index is NULL");
    if (!o) return fail(SKNNR_ERR_INVALID, "opts is NULL");
    if (o->query_dtype < 0 || o->query_dtype >= kDtypeCount) return fail(SKNNR_ERR_INVALID, "unknown query_dtype %d", o->query_dtype);
    if (o->query_dtype != SKNNR_DTYPE_F64 && q) {
        // narrow rows are widened by the query preparation kernel, which exists inside the MFMA envelope only
        if (ix->ks == 0) return fail(SKNNR_ERR_UNSUPPORTED, "query_dtype %d needs d <= 128 (d = %d): pass float64 rows", o->query_dtype, ix->d);
        if (o->formula == SKNNR_FORMULA_HAMMING) return fail(SKNNR_ERR_UNSUPPORTED, "node ids are float64: query_dtype must be 0 with formula = HAMMING");
    }
    if (nq < 0) return fail(SKNNR_ERR_INVALID, "nq must be >= 0");
    if (!out_idx && nq > 0) return fail(SKNNR_ERR_INVALID, "out_idx is NULL");
    if (o->n_neighbors <= 0) return fail(SKNNR_ERR_INVALID, "Expected n_neighbors > 0. Got %d", o->n_neighbors);
    if (!q && !o->exclude_self) return fail(SKNNR_ERR_INVALID, "q is NULL but exclude_self is not set");
    const long n_fit = ix->n_ref;
    if (o->exclude_self) {
        if (o->n_neighbors + 1 > n_fit)
            return fail(SKNNR_ERR_K_TOO_LARGE,
                        "Expected n_neighbors < n_samples_fit, but n_neighbors = %d, n_samples_fit = %ld, n_samples = %ld",
                        o->n_neighbors, n_fit, (long)nq);
        if (!q && (o->row_offset < 0 || o->row_offset + nq > n_fit))
            return fail(SKNNR_ERR_INVALID, "self query rows [%ld, %ld) outside the %ld reference rows",
                        (long)o->row_offset, (long)(o->row_offset + nq), n_fit);
    } else if (o->n_neighbors > n_fit) {
        return fail(SKNNR_ERR_K_TOO_LARGE,
                    "Expected n_neighbors <= n_samples_fit, but n_neighbors = %d, n_samples_fit = %ld, n_samples = %ld",
                    o->n_neighbors, n_fit, (long)nq);
    }
    if (o->apply_affine && !ix->has_affine)
        return fail(SKNNR_ERR_INVALID, "apply_affine is set but no affine map was installed");
    if (o->formula != SKNNR_FORMULA_EXPANDED && o->formula != SKNNR_FORMULA_DIRECT && o->formula != SKNNR_FORMULA_HAMMING)
        return fail(SKNNR_ERR_INVALID, "unknown formula %d", o->formula);
    if (o->formula == SKNNR_FORMULA_HAMMING) {
        if (!ix->has_hw) return fail(SKNNR_ERR_INVALID, "formula = HAMMING needs sknnr_index_set_hamming_weights first");
        if (o->apply_affine) return fail(SKNNR_ERR_INVALID, "node ids are not mapped by an affine transform");
    }
    if (o->n_neighbors + (o->exclude_self ? 1 : 0) > kScanMaxKK)
        return fail(SKNNR_ERR_UNSUPPORTED, "n_neighbors = %d exceeds the HIP backend's limit of %d", o->n_neighbors,
                    kScanMaxKK - 1);
    if (std::abs(o->decimals) > 300) return fail(SKNNR_ERR_INVALID, "decimals out of range");
    return SKNNR_OK;
}

// Device-resident core: nq rows at xdev (raw if affine else transformed), outputs on device.
// raw / id_offset: shard candidates (sknnr_shard_candidates): squared values ascending by (value, index), indices +
// id_offset, no post-steps
int run_device(sknnr_index* ix, const void* xdev, long nq, const sknnr_query_opts* o, double* d_dist,
               long* d_idx, hipStream_t st, int raw = 0, long id_offset = 0) {
    const int kk = o->n_neighbors + (o->exclude_self ? 1 : 0);
    const bool affine = o->apply_affine != 0 && xdev != nullptr;
    const bool self_rows = xdev == nullptr;
    // rows narrower than float64 (opts->query_dtype): the prep kernel widens them and writes the float64 rows everything
    // after it reads (finaliser, exact scan), as it does for rows that go through the affine map
    const int x_dtype = self_rows ? kDtypeF64 : o->query_dtype;
    const bool to_xt = affine || x_dtype != kDtypeF64;
    static const bool scan_everything = std::getenv("SKNNR_EXACT_ONLY") != nullptr;  // development: time the float64 scan alone
    const bool coarse = ix->ks > 0 && kk <= kCoarseMaxKK && o->formula != SKNNR_FORMULA_HAMMING && !scan_everything;
    const int d_x = affine ? ix->d_in : ix->d;
    if (nq > 0x7fffffffL) return fail(SKNNR_ERR_UNSUPPORTED, "more than 2^31 - 1 query rows in one call");
    if (affine && ix->ks == 0)
        return fail(SKNNR_ERR_UNSUPPORTED, "d = %d > 128 with an affine map is outside the HIP envelope", ix->d);
    // the previous call may have run on another stream: its last kernel must be done with the workspace
    if (ix->ws_busy) HIP_TRY(hipStreamWaitEvent(st, ix->ev_ws, 0));
    const bool check_finite = o->check_finite != 0 && xdev != nullptr;

    // transformed rows of the WHOLE call: the exact scan at the end reads any row of it
    const double* xq_call = self_rows ? ix->ref64.p + o->row_offset * ix->d : (const double*)xdev;
    if (to_xt) {
        HIP_TRY(ix->xt.ensure((size_t)nq * ix->d));
        xq_call = ix->xt.p;
    }
    const long chunk = chunk_rows(ix->ks, coarse_list_len(ix, kk));
    const long cap = std::min(chunk, nq);
    const long cap_pad = (cap + kRowQuantum - 1) / kRowQuantum * kRowQuantum;
    if (coarse || to_xt) {
        HIP_TRY(ix->qimg.ensure((size_t)(cap_pad / 32) * 2 * ix->ks * 64));
        HIP_TRY(ix->qnc.ensure(cap_pad));
    }
    if (coarse && ix->cell_depth > 0) {
        HIP_TRY(ix->qcell.ensure((size_t)cap_pad));
        HIP_TRY(ix->qperm.ensure((size_t)cap_pad));
    }
    if (coarse) {
        HIP_TRY(ix->cand_val.ensure((size_t)cap_pad * 2 * coarse_list_len(ix, kk)));
        HIP_TRY(ix->cand_idx.ensure((size_t)cap_pad * 2 * coarse_list_len(ix, kk)));
        HIP_TRY(ix->fail_list.ensure(nq));
        HIP_TRY(hipMemsetAsync(ix->fail_count.p, 0, 16, st));
    }

    SelectArgs call{};
    call.xq = xq_call;
    call.ref = ix->ref64.p;
    call.rn = ix->rn64.p;
    call.nq = nq;
    call.d = ix->d;
    call.n_ref = (int)ix->n_ref;
    call.k = o->n_neighbors;
    call.kk = kk;
    call.exclude_self = o->exclude_self ? 1 : 0;
    call.deterministic = o->deterministic ? 1 : 0;
    call.formula = o->formula;
    call.pow10_is_divisor = o->decimals < 0;
    call.pow10 = std::pow(10.0, std::abs(o->decimals));
    call.row_offset = o->row_offset;
    call.hw = ix->hw.p;
    call.hw_sum = ix->hw_sum;
    call.raw = raw;
    call.id_offset = id_offset;
    if (raw) call.deterministic = 0;
    call.out_dist = d_dist;
    call.out_idx = d_idx;

    // timing record of this call (a ring: an old record still pending is folded into the totals first)
    sknnr_index::CallTiming& ct = ix->timing[ix->timing_next];
    ix->timing_next = (ix->timing_next + 1) % sknnr_index::kTimingRing;
    resolve_call(ix, ct);
    if (!ct.e0) {
        HIP_TRY(hipEventCreate(&ct.e0));
        HIP_TRY(hipEventCreate(&ct.e1));
    }
    ct.coarse_used = 0;
    ct.coarse_rows = 0;
    HIP_TRY(hipEventRecord(ct.e0, st));
    if (check_finite && !(coarse || to_xt)) {
        // no prep kernel reads the rows on this path: scan them here
        const long n_el = nq * (long)d_x;
        HIP_TRY(launch::check_finite((const double*)xdev, n_el, ix->status.p, st));
    }
    for (long c0 = 0; c0 < nq; c0 += chunk) {
        const long n = std::min(chunk, nq - c0);
        const long n_pad = (n + kRowQuantum - 1) / kRowQuantum * kRowQuantum;
        const void* xin = self_rows ? (const void*)(xq_call + c0 * ix->d)
                                    : (const void*)((const char*)xdev + (size_t)c0 * d_x * dtype_bytes(x_dtype));
        const bool bucketed = coarse && ix->cell_depth > 0 && use_coarse2(ix, coarse_list_len(ix, kk));
        bool cells_done = false;
        if (coarse || to_xt) {
            int rc = launch_prep(ix, xin, n, n_pad, affine, to_xt ? ix->xt.p + c0 * ix->d : nullptr, st, check_finite, bucketed, &cells_done,
                                 x_dtype);
            if (rc) return rc;
        }
        if (!coarse) continue;

        if (ct.coarse_used == ct.coarse.size()) {
            hipEvent_t e0, e1;
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            ct.coarse.emplace_back(e0, e1);
        }
        auto& ev = ct.coarse[ct.coarse_used++];
        const bool v2 = use_coarse2(ix, coarse_list_len(ix, kk));
        if (bucketed) {
            // cell of every row, rows bucketed by cell: position -> row (bucket.hip.h)
            CellArgs ca{};
            ca.xq = xq_call + c0 * ix->d;
            ca.nq = n;
            ca.n_pad = n_pad;
            ca.d = ix->d;
            ca.depth = ix->cell_depth;
            ca.axes = ix->cell_axes.p;
            ca.centre = ix->cell_centre.p;
            ca.thr = ix->cell_thr.p;
            ca.cell = ix->qcell.p;
            ca.hist = ix->cell_hist.p;
            ca.perm = ix->qperm.p;
            HIP_TRY(hipMemsetAsync(ix->cell_hist.p, 0, 2 * kCellMax * sizeof(int), st));
            if (!cells_done) HIP_TRY(launch::cell_assign(ca, st));
            HIP_TRY(launch::cell_count(ca, st));
            HIP_TRY(launch::cell_scatter(ca, st));
        }
        HIP_TRY(hipEventRecord(ev.first, st));
        if (v2 && !ix->st_side && !std::getenv("SKNNR_NO_SIDE_STREAM")) {
            HIP_TRY(hipStreamCreateWithFlags(&ix->st_side, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&ix->ev_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&ix->ev_join, hipEventDisableTiming));
        }
        ix->bulk_rows_done = 0;
        ix->ev_bulk_end = ev.second;
        {
            // matrix work issued / algorithmic (rows of the launch cancel): K-steps x tiles swept over d x n_ref
            const long tiles = v2 ? (long)ix->n_stages2 * tiles_per_stage2(ix->ks) +
                                        seed_tiles_for((long)ix->n_stages2 * tiles_per_stage2(ix->ks), tiles_per_stage2(ix->ks))
                                  : (long)ix->n_stages * tiles_per_stage(ix->ks);
            ix->stats.mfma_executed_ratio = (double)tiles * 32.0 * (16.0 * ix->ks) / ((double)ix->n_ref * ix->d);
        }
        int rc = v2 ? launch_coarse2(ix, n, coarse_list_len(ix, kk), kk, st)
                    : launch_coarse(ix, n_pad, coarse_list_len(ix, kk), kk, st);
        if (rc) return rc;
        if (ix->ev_bulk_end) HIP_TRY(hipEventRecord(ev.second, st));  // (no fork: the whole pre-filter is timed)
        ix->ev_bulk_end = nullptr;
        ct.coarse_rows += ix->bulk_rows_done > 0 ? std::min<long>(ix->bulk_rows_done, n) : n;

        FinalizeArgs f{};
        f.s = call;  // this chunk's window of the call
        f.s.xq = xq_call + c0 * ix->d;
        f.s.nq = n;
        f.s.row_offset = o->row_offset + c0;
        f.s.out_dist = d_dist ? d_dist + c0 * o->n_neighbors : nullptr;
        f.s.out_idx = d_idx + c0 * o->n_neighbors;
        f.cand_val = ix->cand_val.p;
        f.cand_idx = ix->cand_idx.p;
        f.perm = v2 ? ix->perm2.p : ix->perm.p;
        f.qnc = ix->qnc.p;
        f.qperm = bucketed ? ix->qperm.p : nullptr;
        f.m_list = coarse_list_len(ix, kk);
        f.rank_extra = coarse_rank_extra(f.m_list, kk);
        f.inv_s2 = 1.0 / (ix->s * ix->s);
        f.s2 = ix->s * ix->s;
        f.inv_s = 1.0 / ix->s;
        f.eps_c = (v2 ? eps_units2(ix->ks) : eps_units(ix->ks)) * std::ldexp(1.0, -24);
        f.ymax = ix->ymax;
        f.noise_a = (ix->d + 4) * std::ldexp(1.0, -53);
        f.mu2 = o->formula == SKNNR_FORMULA_EXPANDED ? 2.0 * ix->mu_norm : 0.0;
        f.fail_list = ix->fail_list.p;
        f.fail_count = ix->fail_count.p;
        f.fail_base = (int)c0;
        // rows [r0, r0 + rows) of the chunk (the kernel indexes everything by the row inside its window); bucketed
        // calls: POSITIONS [r0, r0 + rows) of the chunk, the kernel maps them to rows of the chunk
        auto finalize_rows = [&](long r0, long rows, hipStream_t s_) {
            FinalizeArgs g = f;
            if (bucketed) {
                g.pos0 = r0;
                g.s.nq = rows;
                launch_finalize(g, rows, s_);
                return;
            }
            g.s.xq = f.s.xq + r0 * ix->d;
            g.s.nq = rows;
            g.s.row_offset = f.s.row_offset + r0;
            g.s.out_dist = f.s.out_dist ? f.s.out_dist + r0 * o->n_neighbors : nullptr;
            g.s.out_idx = f.s.out_idx + r0 * o->n_neighbors;
            g.cand_val = f.cand_val + (size_t)r0 * 2 * f.m_list;
            g.cand_idx = f.cand_idx + (size_t)r0 * 2 * f.m_list;
            g.qnc = f.qnc + r0;
            g.fail_base = f.fail_base + (int)r0;
            launch_finalize(g, rows, s_);
        };
        const long done = v2 ? std::min<long>(ix->bulk_rows_done, n) : 0;
        if (done > 0) {
            // fork: the rows of the bulk launch are finalised on the side stream while the thin round runs here
            HIP_TRY(hipStreamWaitEvent(ix->st_side, ix->ev_fork, 0));
            finalize_rows(0, done, ix->st_side);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(ix->ev_join, ix->st_side));
            if (n > done) finalize_rows(done, n - done, st);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamWaitEvent(st, ix->ev_join, 0));  // join before anything else touches the workspace
        } else {
            finalize_rows(0, n, st);
            HIP_TRY(hipGetLastError());
        }
    }
    // Weighted Hamming on 16-bit ids: integer pre-filter, float64 re-score of the candidates (hamming.hip.h); the queries
    // it cannot serve (too many candidates, ids outside 16 bits) join the fail list of the exact scan below
    const bool ham_int = !coarse && o->formula == SKNNR_FORMULA_HAMMING && ix->h16_ok && kk <= kHamMaxKK;
    if (ham_int) {
        const long chunk_q = 1L << 18;  // 768 candidate bytes per query: 200 MB per chunk
        const long cap_q = std::min(chunk_q, nq);
        const long cap_pad = (cap_q + 255) / 256 * 256;
        HIP_TRY(ix->h_qimg.ensure((size_t)ix->h_tp * cap_pad));
        HIP_TRY(ix->h_bad.ensure((size_t)cap_pad));
        HIP_TRY(ix->h_cand_cnt.ensure((size_t)cap_q));
        HIP_TRY(ix->h_cand_id.ensure((size_t)cap_q * kHamCand));
        HIP_TRY(ix->fail_list.ensure(nq));
        HIP_TRY(hipMemsetAsync(ix->fail_count.p, 0, 16, st));
        for (long c0 = 0; c0 < nq; c0 += chunk_q) {
            const long n = std::min(chunk_q, nq - c0);
            const long n_pad = (n + 255) / 256 * 256;
            HIP_TRY(launch::hamming_pack(xq_call + c0 * ix->d, n, n_pad, ix->d, ix->h_tp, ix->h_qimg.p, ix->h_bad.p, st));
            HammingArgs ha{};
            ha.rimg = ix->h_rimg.p;
            ha.wq = ix->h_wq.p;
            ha.qimg = ix->h_qimg.p;
            ha.q_bad = ix->h_bad.p;
            ha.n_ref = (int)ix->n_ref;
            ha.n_ref_pad = ix->h_ref_pad;
            ha.tp = ix->h_tp;
            ha.nq = n;
            ha.nq_pad = n_pad;
            ha.kk = kk;
            ha.band = (unsigned)ix->d + 2u;
            ha.cand_cnt = ix->h_cand_cnt.p;
            ha.cand_id = ix->h_cand_id.p;
            HIP_TRY(launch::hamming_coarse(ha, st));
            HammingRescoreArgs hr{};
            hr.s = call;
            hr.s.xq = xq_call + c0 * ix->d;
            hr.s.nq = n;
            hr.s.row_offset = o->row_offset + c0;
            hr.s.out_dist = d_dist ? d_dist + c0 * o->n_neighbors : nullptr;
            hr.s.out_idx = d_idx + c0 * o->n_neighbors;
            hr.cand_cnt = ix->h_cand_cnt.p;
            hr.cand_id = ix->h_cand_id.p;
            hr.fail_list = ix->fail_list.p;
            hr.fail_count = ix->fail_count.p;
            hr.fail_base = (int)c0;
            hr.rrow = ix->h_rrow.p;
            hr.tpr = ham_row_dwords(ix->d);
            HIP_TRY(launch::hamming_rescore(hr, st));
        }
    }
    // One exact scan per call: the rows the finaliser could not certify (call-relative ids), or
    // every row when the call is outside the MFMA envelope.
    int rc = (coarse || ham_int) ? launch_scan(ix, call, ix->fail_list.p, ix->fail_count.p, nq, st)
                                 : launch_scan(ix, call, nullptr, nullptr, nq, st);
    if (rc) return rc;
    if (coarse) {
        // keep a running total on the device; sknnr_get_stats reads it (no sync here)
        HIP_TRY(launch::add_counter(ix->fail_count.p, ix->fail_total.p, st));
        ix->stats.coarse_queries += nq;
    } else {
        // (weighted Hamming: the rows the integer pre-filter could not serve -- list overflow, ids beyond 16 bits -- count as
        //  fallbacks too)
        if (ham_int) HIP_TRY(launch::add_counter(ix->fail_count.p, ix->fail_total.p, st));
        ix->stats.exact_only_queries += nq;
    }
    HIP_TRY(hipEventRecord(ct.e1, st));
    ct.pending = true;
    HIP_TRY(hipEventRecord(ix->ev_ws, st));
    ix->ws_busy = true;
    ix->stats.queries += nq;
    return SKNNR_OK;
}

}  // namespace

// ----------------------------------------------------------------------------------------
// host-buffer pipeline
// ----------------------------------------------------------------------------------------
static int launch_predict(sknnr_index* ix, const double* dist, const long* idx, const double* w, long nq, int k,
                          int mode, double* out, hipStream_t st);

namespace {

constexpr long kHostChunkRows = 1L << 20;  // rows per pipeline slot (SKNNR_HOST_CHUNK_ROWS overrides)

long host_chunk_rows() {
    static const long v = [] {
        const char* e = std::getenv("SKNNR_HOST_CHUNK_ROWS");
        const long r = e ? std::atol(e) : 0;
        return r >= 1024 ? r : kHostChunkRows;
    }();
    return v;
}

// memcpy split over a few threads: a single core copies ~10 GB/s, PCIe Gen5 x16 moves ~50
void parallel_copy(void* dst, const void* src, size_t bytes) {
    const size_t kMin = 8u << 20;
    unsigned n = std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency()));
    if (bytes < 2 * kMin) n = 1;
    if (n == 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = ((bytes / n) + 4095) & ~(size_t)4095;
    for (unsigned i = 0; i < n; ++i) {
        const size_t a = (size_t)i * per;
        if (a >= bytes) break;
        const size_t len = std::min(per, bytes - a);
        th.emplace_back([=] { std::memcpy((char*)dst + a, (const char*)src + a, len); });
    }
    for (auto& t : th) t.join();
}

template <typename T>
int ensure_pinned(T*& p, size_t& have, size_t want) {
    if (p && have >= want) return SKNNR_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    have = 0;
    HIP_TRY(hipHostMalloc((void**)&p, std::max<size_t>(want, 1) * sizeof(T), hipHostMallocDefault));
    have = want;
    return SKNNR_OK;
}

// Pageable host arrays in, pageable host arrays out, through kHostSlots pinned/device slots:
//   host thread : copy tile c into its slot's pinned buffer | copy results of tile c - kHostSlots out
//   st_h2d      : pinned -> device                          (PCIe)
//   st_run      : prep / pre-filter / finalise / scan [+ predict]
//   st_d2h      : device -> pinned                          (PCIe)
// so that PCIe in, kernels and PCIe out of neighbouring tiles overlap.  The state lives in a HostPipe so
// that the streamed entry points (sknnr_stream_*) keep the pipeline full ACROSS calls: a pushed tile's
// results leave the slot when the slot is needed again (kHostSlots tiles later) or at flush.
struct HostPipe {
    sknnr_index* ix = nullptr;
    sknnr_query_opts o{};        // o.row_offset advances with every submitted tile
    bool want_dist = false, want_idx = true, want_pred = false;
    int k = 0, t = 0, d_x = 0;
    size_t x_esz = sizeof(double);  // bytes per element of the caller's rows (opts->query_dtype)
    struct Pending {
        bool live = false;
        long n = 0;
        double* od = nullptr;
        long* oi = nullptr;
        double* op = nullptr;
        std::future<int> out_done;  // copy-out job of the slot's tile (w_out)
    } pending[kHostSlots];
    int slot_of = 0;
    // look-ahead copy-in (one-shot calls know their next tile): the job that stages tile `ahead_tile` into slot `ahead_slot`
    std::packaged_task<int()> deferred[kHostSlots];  // SKNNR_PIPE_WORKERS=0: the copy-out jobs, run when the slot is drained
    std::future<int> ahead_done;
    const void* ahead_q = nullptr;
    int ahead_slot = -1;
    int d2h_slot = -1;  // slot whose kernels are enqueued and whose device-to-host copies are not yet (pipe_enqueue_d2h)
    double ms_copy_in = 0, ms_copy_out = 0, ms_wait = 0, ms_enqueue = 0;  // host-thread time per phase (SKNNR_PIPE_TRACE=1)
};

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
bool pipe_trace() {
    static const bool v = std::getenv("SKNNR_PIPE_TRACE") != nullptr;
    return v;
}

int pipe_open(HostPipe& p, sknnr_index* ix, const sknnr_query_opts* o, bool want_dist, bool want_idx, bool want_pred) {
    p = HostPipe{};
    p.ix = ix;
    p.o = *o;
    p.want_dist = want_dist;
    p.want_idx = want_idx;
    p.want_pred = want_pred;
    p.k = o->n_neighbors;
    p.t = ix->t;
    p.d_x = o->apply_affine ? ix->d_in : ix->d;
    p.x_esz = (size_t)dtype_bytes(o->query_dtype);
    // (each object on its own: a creation that failed half way is completed by the next call, never skipped)
    for (hipStream_t* h : {&ix->st_h2d, &ix->st_run, &ix->st_d2h})
        if (!*h) HIP_TRY(hipStreamCreateWithFlags(h, hipStreamNonBlocking));
    for (auto& sl : ix->slot)
        for (hipEvent_t* e : {&sl.ev_h2d, &sl.ev_done, &sl.ev_d2h})
            if (!*e) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    if (!ix->w_in) ix->w_in.reset(new HostWorker());
    if (!ix->w_out) ix->w_out.reset(new HostWorker());
    return SKNNR_OK;
}

// Slot b's previous tile has left (its copy-out job, on w_out, is done): the slot's buffers may be reused.
int pipe_enqueue_d2h(HostPipe& p);
int pipe_drain(HostPipe& p, int b) {
    auto& pd = p.pending[b];
    if (!pd.live) return SKNNR_OK;
    if (p.d2h_slot == b) {  // (its results have not even been sent yet)
        int rc = pipe_enqueue_d2h(p);
        if (rc) return rc;
    }
    const double t0 = now_ms();
    if (p.deferred[b].valid()) {
        p.deferred[b]();
        p.deferred[b] = std::packaged_task<int()>();
    }
    const int rc = pd.out_done.valid() ? pd.out_done.get() : SKNNR_OK;
    p.ms_wait += now_ms() - t0;
    pd.live = false;
    if (rc) return fail(rc, "host pipeline: copying a tile's results out failed (HIP error while waiting for the device-to-host copy)");
    return SKNNR_OK;
}

// Drain slot b and size its pinned / device buffers for a tile of n rows.
int pipe_prepare_slot(HostPipe& p, int b, long n) {
    sknnr_index* ix = p.ix;
    auto& sl = ix->slot[b];
    int rc = pipe_drain(p, b);  // the slot's previous tile must have left before its buffers are reused
    if (rc) return rc;
    const int k = p.k, t = p.t, d_x = p.d_x;
    const size_t x_f64 = ((size_t)n * d_x * p.x_esz + 7) / 8;  // the tile's rows, in 8-byte units
    if ((rc = ensure_pinned(sl.pin_x, sl.pin_x_n, x_f64))) return rc;
    if ((rc = ensure_pinned(sl.pin_i, sl.pin_i_n, (size_t)n * k))) return rc;
    if (p.want_dist && (rc = ensure_pinned(sl.pin_d, sl.pin_d_n, (size_t)n * k))) return rc;
    if (p.want_pred && (rc = ensure_pinned(sl.pin_p, sl.pin_p_n, (size_t)n * t))) return rc;
    HIP_TRY(sl.dev_x.ensure(x_f64));
    HIP_TRY(sl.dev_i.ensure((size_t)n * k));
    HIP_TRY(sl.dev_d.ensure((size_t)n * k));
    if (p.want_pred) HIP_TRY(sl.dev_p.ensure((size_t)n * t));
    return SKNNR_OK;
}

// Device-to-host copies and the copy-out job of the tile whose kernels were enqueued last (slot p.d2h_slot), if any.
// BOTH directions use one stream (st_h2d), tile j's rows in front of tile j-1's results: on this stack a device-to-host
// copy that runs while a host-to-device copy is active is not given a DMA engine but executed by a blit kernel
// (__amd_rocclr_copyBuffer, six pieces of ~0.5 ms per tile in a rocprofv3 trace) that shares the CUs with the hot path
// -- the prep kernel beside it took 1.18 ms instead of 0.30, a tile 7.75 ms instead of 5.8 (profiles/r03_host_pipeline.txt).
// One stream keeps every transfer on the DMA engine; per tile 4.5 ms in + 1.4 ms out still fit beside 5.8 ms of kernels.
int pipe_enqueue_d2h(HostPipe& p) {
    const int b = p.d2h_slot;
    if (b < 0) return SKNNR_OK;
    p.d2h_slot = -1;
    sknnr_index* ix = p.ix;
    auto& sl = ix->slot[b];
    auto& pd = p.pending[b];
    const long n = pd.n;
    const int k = p.k, t = p.t;
    double* od = pd.od;
    long* oi = pd.oi;
    double* op = pd.op;
    static const bool one_stream = [] { const char* e = std::getenv("SKNNR_PIPE_ONE_STREAM"); return !(e && std::atoi(e) == 0); }();
    hipStream_t st = one_stream ? ix->st_h2d : ix->st_d2h;
    HIP_TRY(hipStreamWaitEvent(st, sl.ev_done, 0));
    if (oi) HIP_TRY(hipMemcpyAsync(sl.pin_i, sl.dev_i.p, (size_t)n * k * sizeof(long), hipMemcpyDeviceToHost, st));
    if (od) HIP_TRY(hipMemcpyAsync(sl.pin_d, sl.dev_d.p, (size_t)n * k * sizeof(double), hipMemcpyDeviceToHost, st));
    if (op) HIP_TRY(hipMemcpyAsync(sl.pin_p, sl.dev_p.p, (size_t)n * t * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(sl.ev_d2h, st));
    // the copy-out leg: wait for the tile's device-to-host copies, then pinned -> the caller's arrays (w_out, in order)
    static const bool workers = [] { const char* e = std::getenv("SKNNR_PIPE_WORKERS"); return !(e && std::atoi(e) == 0); }();
    {
        const int device = ix->device;
        hipEvent_t ev = sl.ev_d2h;
        const long* pin_i = sl.pin_i;
        const double *pin_d = sl.pin_d, *pin_p = sl.pin_p;
        auto job = [=]() -> int {
            if (hipSetDevice(device) != hipSuccess || hipEventSynchronize(ev) != hipSuccess) return SKNNR_ERR_HIP;
            if (oi) parallel_copy(oi, pin_i, (size_t)n * k * sizeof(long));
            if (od) parallel_copy(od, pin_d, (size_t)n * k * sizeof(double));
            if (op) parallel_copy(op, pin_p, (size_t)n * t * sizeof(double));
            return SKNNR_OK;
        };
        if (workers) {
            pd.out_done = ix->w_out->post(job);
        } else {  // (A/B: the round-2 behaviour -- the copy runs on this thread when the slot is drained)
            std::packaged_task<int()> task(job);
            pd.out_done = task.get_future();
            p.deferred[b] = std::move(task);
        }
    }
    return SKNNR_OK;
}

// One tile of at most host_chunk_rows() rows.  `q` may be reused by the caller as soon as this returns.
// q_next / n_next: the tile the same call will submit next (or null): its rows are staged into the next slot's pinned
// buffer by the copy-in worker while this tile is enqueued and the caller waits for older results.
int pipe_submit(HostPipe& p, const void* q, long n, double* od, long* oi, double* op, const void* q_next = nullptr,
                long n_next = 0) {
    sknnr_index* ix = p.ix;
    const int b = p.slot_of;
    p.slot_of = (p.slot_of + 1) % kHostSlots;
    auto& sl = ix->slot[b];
    const int k = p.k, d_x = p.d_x;
    int rc;
    const double t_in = now_ms();
    if (p.ahead_slot == b && p.ahead_q == q && p.ahead_done.valid()) {
        rc = p.ahead_done.get();  // staged ahead by the worker (the slot was prepared when the job was posted)
        p.ahead_slot = -1;
        if (rc) return rc;
    } else {
        if ((rc = pipe_prepare_slot(p, b, n))) return rc;
        parallel_copy(sl.pin_x, q, (size_t)n * d_x * p.x_esz);
    }
    const double t_enq = now_ms();
    p.ms_copy_in += t_enq - t_in;
    HIP_TRY(hipMemcpyAsync(sl.dev_x.p, sl.pin_x, (size_t)n * d_x * p.x_esz, hipMemcpyHostToDevice, ix->st_h2d));
    HIP_TRY(hipEventRecord(sl.ev_h2d, ix->st_h2d));
    // the PREVIOUS tile's results travel behind this tile's rows, on the same stream (see pipe_enqueue_d2h)
    if ((rc = pipe_enqueue_d2h(p))) return rc;
    HIP_TRY(hipStreamWaitEvent(ix->st_run, sl.ev_h2d, 0));
    rc = run_device(ix, sl.dev_x.p, n, &p.o, sl.dev_d.p, sl.dev_i.p, ix->st_run);
    if (rc) return rc;
    if (p.want_pred) {
        rc = launch_predict(ix, sl.dev_d.p, sl.dev_i.p, nullptr, n, k, p.o.weight_mode, sl.dev_p.p, ix->st_run);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(sl.ev_done, ix->st_run));
    auto& pd = p.pending[b];
    pd.live = true;
    pd.n = n;
    pd.od = od;
    pd.oi = oi;
    pd.op = op;
    p.d2h_slot = b;  // its device-to-host copies are enqueued behind the next tile's rows, or by the flush
    p.o.row_offset += n;
    p.ms_enqueue += now_ms() - t_enq;
    static const bool workers = [] { const char* e = std::getenv("SKNNR_PIPE_WORKERS"); return !(e && std::atoi(e) == 0); }();
    if (workers && q_next && n_next > 0) {
        // the copy-in leg of the next tile (w_in): its slot is drained and sized here, on this thread
        const int b2 = p.slot_of;
        if ((rc = pipe_prepare_slot(p, b2, n_next))) return rc;
        double* dst = ix->slot[b2].pin_x;
        const size_t bytes = (size_t)n_next * d_x * p.x_esz;
        p.ahead_done = ix->w_in->post([=]() -> int {
            parallel_copy(dst, q_next, bytes);
            return SKNNR_OK;
        });
        p.ahead_q = q_next;
        p.ahead_slot = b2;
    }
    return SKNNR_OK;
}

// Submit `nq` rows in tiles of at most host_chunk_rows().  A pipeline that starts empty ramps up: the device waits for
// the first tile's staging copy and host-to-device transfer (7 ms for a 1M-row tile) with nothing to do, so the first
// tiles are an eighth, a quarter and a half of the regular size.
int pipe_submit_rows(HostPipe& p, const void* q_, long nq, double* od, long* oi, double* op) {
    const char* q = (const char*)q_;
    const size_t row_bytes = (size_t)p.d_x * p.x_esz;
    const long cap = host_chunk_rows();
    bool idle = p.d2h_slot < 0;
    for (const auto& pd : p.pending) idle = idle && !pd.live;
    std::vector<long> cuts;  // tile boundaries
    long c = 0;
    if (idle && nq > cap / 2 && !std::getenv("SKNNR_PIPE_NO_RAMP"))
        for (long part : {cap / 8, cap / 4, cap / 2}) {
            part = std::max<long>(part / kRowQuantum * kRowQuantum, kRowQuantum);
            if (c + part >= nq) break;
            c += part;
            cuts.push_back(c);
        }
    while (c < nq) {
        c = std::min(nq, c + cap);
        cuts.push_back(c);
    }
    long c0 = 0;
    for (size_t i = 0; i < cuts.size(); ++i) {
        const long c1 = cuts[i], n = c1 - c0;
        const long n_next = i + 1 < cuts.size() ? cuts[i + 1] - c1 : 0;
        int rc = pipe_submit(p, q + c0 * row_bytes, n, od ? od + c0 * p.k : nullptr, oi ? oi + c0 * p.k : nullptr,
                             op ? op + c0 * p.t : nullptr, n_next ? q + c1 * row_bytes : nullptr, n_next);
        if (rc) return rc;
        c0 = c1;
    }
    return SKNNR_OK;
}

// Everything submitted so far is in the caller's arrays when this returns; reports non-finite input.
int pipe_flush(HostPipe& p) {
    if (p.ahead_done.valid()) {  // (a look-ahead copy nobody consumed: only after a failed submit)
        (void)p.ahead_done.get();
        p.ahead_slot = -1;
    }
    {
        int rc = pipe_enqueue_d2h(p);  // the last tile's results have nobody behind them
        if (rc) return rc;
    }
    for (int i = 0; i < kHostSlots; ++i) {
        int rc = pipe_drain(p, (p.slot_of + i) % kHostSlots);  // oldest tile first (the next slot to be reused)
        if (rc) return rc;
    }
    if (pipe_trace()) {
        std::fprintf(stderr, "[pipe] host thread: copy-in %.1f ms, copy-out %.1f ms, waiting for results %.1f ms, enqueue %.1f ms\n",
                     p.ms_copy_in, p.ms_copy_out, p.ms_wait, p.ms_enqueue);
        p.ms_copy_in = p.ms_copy_out = p.ms_wait = p.ms_enqueue = 0;
    }
    if (p.o.check_finite) {
        HIP_TRY(hipStreamSynchronize(p.ix->st_run));
        return poll_status(p.ix);
    }
    return SKNNR_OK;
}

// After a failed submit: no posted job may still touch the caller's memory or the slots when the call returns.
void pipe_abort(HostPipe& p) {
    if (p.ahead_done.valid()) (void)p.ahead_done.get();
    p.ahead_slot = -1;
    p.d2h_slot = -1;
    for (int b = 0; b < kHostSlots; ++b) {
        auto& pd = p.pending[b];
        if (p.deferred[b].valid()) {
            p.deferred[b]();
            p.deferred[b] = std::packaged_task<int()>();
        }
        if (pd.out_done.valid()) (void)pd.out_done.get();
        pd.live = false;
    }
}

// Fresh output arrays (numpy's np.empty) are untouched memory: the first write to every page is a fault, and the
// copy-out leg would take them one by one in the middle of the pipeline.  MADV_POPULATE_WRITE (Linux >= 5.14) makes the
// pages present and writable without changing what they hold, so it may run beside the copies; a few threads, because
// the kernel zero-fills the pages it hands out.  Unsupported kernels answer EINVAL: then the copies fault as before.
struct Prefault {
    std::vector<std::thread> th;
    void add(void* ptr, size_t bytes) {
        if (!ptr || bytes < (8u << 20)) return;
        const uintptr_t lo = ((uintptr_t)ptr + 4095) & ~(uintptr_t)4095, hi = ((uintptr_t)ptr + bytes) & ~(uintptr_t)4095;
        if (hi <= lo) return;
        const unsigned n = 4;
        const size_t per = (((hi - lo) / n) + 4095) & ~(size_t)4095;
        for (unsigned i = 0; i < n; ++i) {
            const uintptr_t a = lo + (uintptr_t)i * per;
            if (a >= hi) break;
            const size_t len = std::min<size_t>(per, hi - a);
            th.emplace_back([a, len] { (void)madvise((void*)a, len, 23 /* MADV_POPULATE_WRITE */); });
        }
    }
    ~Prefault() {
        for (auto& t : th) t.join();
    }
};

int run_host_pipeline(sknnr_index* ix, const void* q, long nq, const sknnr_query_opts* o, double* out_dist,
                      long* out_idx, double* out_pred) {
    if (ix->stream_open) return fail(SKNNR_ERR_INVALID, "a query stream is open on this handle: end it first");
    HostPipe p;
    int rc = pipe_open(p, ix, o, out_dist != nullptr, true, out_pred != nullptr);
    Prefault pf;  // (joined when the call returns)
    if (!rc && std::getenv("SKNNR_PREFAULT")) {  // (measured: 96 ms without, 115 ms with -- the populate threads take bandwidth the copies need)
        pf.add(out_idx, (size_t)nq * o->n_neighbors * sizeof(long));
        pf.add(out_dist, (size_t)nq * o->n_neighbors * sizeof(double));
        pf.add(out_pred, (size_t)nq * ix->t * sizeof(double));
    }
    if (!rc) rc = pipe_submit_rows(p, q, nq, out_dist, out_idx, out_pred);
    if (!rc) rc = pipe_flush(p);
    if (rc) {
        const std::string msg = g_last_error;
        pipe_abort(p);
        (void)hipDeviceSynchronize();  // nothing of a failed call may still be in flight on the slots
        g_last_error = msg;
    }
    return rc;
}

// X=None (the query rows are the handle's own reference rows, already on the device): only the results
// travel.  Runs on the default stream, chunk by chunk.
int run_self_rows(sknnr_index* ix, long nq, const sknnr_query_opts* o, double* out_dist, long* out_idx, double* out_pred) {
    hipStream_t st = nullptr;
    const int k = o->n_neighbors;
    for (long c0 = 0; c0 < nq; c0 += kChunkRows) {
        const long n = std::min<long>(kChunkRows, nq - c0);
        HIP_TRY(ix->dist_stage.ensure((size_t)n * k));
        HIP_TRY(ix->idx_stage.ensure((size_t)n * k));
        if (out_pred) HIP_TRY(ix->pred_stage.ensure((size_t)n * ix->t));
        sknnr_query_opts oc = *o;
        oc.row_offset = o->row_offset + c0;
        int rc = run_device(ix, nullptr, n, &oc, ix->dist_stage.p, ix->idx_stage.p, st);
        if (rc) return rc;
        if (out_pred) {
            rc = launch_predict(ix, ix->dist_stage.p, ix->idx_stage.p, nullptr, n, k, o->weight_mode, ix->pred_stage.p, st);
            if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(out_pred + c0 * ix->t, ix->pred_stage.p, (size_t)n * ix->t * sizeof(double), hipMemcpyDeviceToHost, st));
        }
        if (out_dist)
            HIP_TRY(hipMemcpyAsync(out_dist + c0 * k, ix->dist_stage.p, (size_t)n * k * sizeof(double), hipMemcpyDeviceToHost, st));
        if (out_idx)
            HIP_TRY(hipMemcpyAsync(out_idx + c0 * k, ix->idx_stage.p, (size_t)n * k * sizeof(long), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return SKNNR_OK;
}

}  // namespace

// ----------------------------------------------------------------------------------------
// kneighbors
// ----------------------------------------------------------------------------------------
extern "C" int sknnr_kneighbors(sknnr_index* ix, const void* q, int64_t nq, const sknnr_query_opts* o,
                                double* out_dist, int64_t* out_idx, int32_t mem, void* stream) {
    int rc = validate_call(ix, q, nq, o, out_idx);
    if (rc) return rc;
    if (nq == 0) return SKNNR_OK;
    if (mem != SKNNR_MEM_DEVICE && mem != SKNNR_MEM_HOST) return fail(SKNNR_ERR_INVALID, "unknown memspace %d", mem);
    std::lock_guard<std::mutex> lock(ix->mtx);
    HIP_TRY(hipSetDevice(ix->device));
    if (mem == SKNNR_MEM_DEVICE) return run_device(ix, q, nq, o, out_dist, (long*)out_idx, (hipStream_t)stream);
    if (q) return run_host_pipeline(ix, q, nq, o, out_dist, (long*)out_idx, nullptr);
    return run_self_rows(ix, nq, o, out_dist, (long*)out_idx, nullptr);
}


// ----------------------------------------------------------------------------------------
// full weighted-Hamming distance rows (the reference's own choice among exactly tied rows: np.argpartition on the host)
// ----------------------------------------------------------------------------------------
extern "C" int sknnr_hamming_distances(sknnr_index* ix, const double* q, int64_t nq, const int64_t* rows, int64_t n_rows,
                                       double* out, int32_t mem, void* stream) {
    if (!ix) return fail(SKNNR_ERR_INVALID, "index is NULL");
    if (!ix->has_hw) return fail(SKNNR_ERR_INVALID, "sknnr_hamming_distances needs sknnr_index_set_hamming_weights first");
    if (n_rows < 0 || nq < 0) return fail(SKNNR_ERR_INVALID, "negative row count");
    if (n_rows == 0) return SKNNR_OK;
    if (!out) return fail(SKNNR_ERR_INVALID, "out is NULL");
    if (mem != SKNNR_MEM_DEVICE && mem != SKNNR_MEM_HOST) return fail(SKNNR_ERR_INVALID, "unknown memspace %d", mem);
    const long n_src = q ? nq : ix->n_ref;  // rows that `rows` may name
    if (!rows && n_rows > n_src) return fail(SKNNR_ERR_INVALID, "n_rows = %ld exceeds the %ld query rows", (long)n_rows, n_src);
    std::lock_guard<std::mutex> lock(ix->mtx);
    HIP_TRY(hipSetDevice(ix->device));
    HammingRowsArgs a{};
    a.refT = ix->refT.p;
    a.d = ix->d;
    a.n_ref = (int)ix->n_ref;
    a.hw = ix->hw.p;
    a.hw_sum = ix->hw_sum;
    if (mem == SKNNR_MEM_DEVICE) {
        a.xq = q ? q : ix->ref64.p;
        a.rows = (const long*)rows;  // (device memory: the caller vouches for the values)
        a.n_rows = n_rows;
        a.out = out;
        HIP_TRY(launch::hamming_distance_rows(a, (hipStream_t)stream));
        return SKNNR_OK;
    }
    for (int64_t i = 0; rows && i < n_rows; ++i)
        if (rows[i] < 0 || rows[i] >= n_src) return fail(SKNNR_ERR_INVALID, "rows[%ld] = %ld outside [0, %ld)", (long)i, (long)rows[i], n_src);
    // host buffers: the selected rows only travel in, their distance rows out, at most ~256 MB of them at a time
    const long per = std::max<long>(1, std::min<long>(n_rows, (256L << 20) / (8 * std::max<long>(1, ix->n_ref))));
    DevBuf<double> dq, dout;
    DevBuf<long> drows;
    HIP_TRY(dout.ensure((size_t)per * ix->n_ref));
    if (q) HIP_TRY(dq.ensure((size_t)per * ix->d));
    else HIP_TRY(drows.ensure((size_t)per));
    std::vector<double> gathered;
    std::vector<long> sel;
    for (long c0 = 0; c0 < n_rows; c0 += per) {
        const long n = std::min<long>(per, n_rows - c0);
        if (q) {
            gathered.resize((size_t)n * ix->d);
            for (long i = 0; i < n; ++i) {
                const long r = rows ? rows[c0 + i] : c0 + i;
                std::memcpy(&gathered[(size_t)i * ix->d], q + (size_t)r * ix->d, (size_t)ix->d * sizeof(double));
            }
            HIP_TRY(hipMemcpy(dq.p, gathered.data(), gathered.size() * sizeof(double), hipMemcpyHostToDevice));
            a.xq = dq.p;
            a.rows = nullptr;
        } else {
            sel.resize((size_t)n);
            for (long i = 0; i < n; ++i) sel[(size_t)i] = rows ? rows[c0 + i] : c0 + i;
            HIP_TRY(hipMemcpy(drows.p, sel.data(), (size_t)n * sizeof(long), hipMemcpyHostToDevice));
            a.xq = ix->ref64.p;
            a.rows = drows.p;
        }
        a.n_rows = n;
        a.out = dout.p;
        HIP_TRY(launch::hamming_distance_rows(a, nullptr));
        HIP_TRY(hipMemcpy(out + (size_t)c0 * ix->n_ref, dout.p, (size_t)n * ix->n_ref * sizeof(double), hipMemcpyDeviceToHost));
    }
    return SKNNR_OK;
}

// ----------------------------------------------------------------------------------------
// reference-sharded search: per-shard candidates, then the merge
// ----------------------------------------------------------------------------------------
extern "C" int sknnr_shard_candidates(sknnr_index* ix, const double* q, int64_t nq, const sknnr_query_opts* o,
                                      int64_t index_offset, double* out_val, int64_t* out_idx, int32_t mem, void* stream) {
    int rc = validate_call(ix, q, nq, o, out_idx);
    if (rc) return rc;
    if (o->query_dtype != SKNNR_DTYPE_F64) return fail(SKNNR_ERR_UNSUPPORTED, "the sharded entry points take float64 rows (query_dtype = 0)");
    if (!q || o->exclude_self)
        return fail(SKNNR_ERR_INVALID, "shard candidates are searched for given rows: the caller adds the self slot (n_neighbors + 1) and the merge drops it");
    if (!out_val && nq > 0) return fail(SKNNR_ERR_INVALID, "out_val is NULL");
    if (index_offset < 0 || index_offset + ix->n_ref > 0x7fffff00L) return fail(SKNNR_ERR_INVALID, "index_offset out of range");
    if (nq == 0) return SKNNR_OK;
    if (mem != SKNNR_MEM_DEVICE && mem != SKNNR_MEM_HOST) return fail(SKNNR_ERR_INVALID, "unknown memspace %d", mem);
    std::lock_guard<std::mutex> lock(ix->mtx);
    HIP_TRY(hipSetDevice(ix->device));
    if (mem == SKNNR_MEM_DEVICE) return run_device(ix, q, nq, o, out_val, (long*)out_idx, (hipStream_t)stream, 1, index_offset);
    // host buffers: one staged round trip (a shard's candidates are an intermediate of a multi-GPU call, not a hot host path)
    const int k = o->n_neighbors;
    const int d_x = o->apply_affine ? ix->d_in : ix->d;
    DevBuf<double> dq, dv;
    DevBuf<long> di;
    HIP_TRY(dq.ensure((size_t)nq * d_x));
    HIP_TRY(dv.ensure((size_t)nq * k));
    HIP_TRY(di.ensure((size_t)nq * k));
    HIP_TRY(hipMemcpy(dq.p, q, (size_t)nq * d_x * sizeof(double), hipMemcpyHostToDevice));
    rc = run_device(ix, dq.p, nq, o, dv.p, di.p, nullptr, 1, index_offset);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (o->check_finite && (rc = poll_status(ix))) return rc;
    HIP_TRY(hipMemcpy(out_val, dv.p, (size_t)nq * k * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_idx, di.p, (size_t)nq * k * sizeof(long), hipMemcpyDeviceToHost));
    return SKNNR_OK;
}

namespace {

int merge_shards_formula(sknnr_index* ix, const SelectArgs& call, int n_shards, long nq, hipStream_t st) {
    ScanArgs a{call, ix->refT.p, nullptr, nullptr, ix->slice_v.p, ix->slice_i.p, ix->fail_list2.p, ix->fail_count.p + 2};
    const int kkp = (call.kk + 2) & ~1, stk = (2 * call.kk + 4 + 1) & ~1;
    const size_t msh = 4 * ((size_t)12 * kkp + (size_t)4 * stk);
    HIP_TRY(launch::scan_merge(call.formula, a, (nq + 3) / 4, msh, 0, n_shards, st));
    // rows whose merged answer is not unique (exact ties that the reference's heap settles by its history): the
    // sequential scan over ALL reference rows of this handle, as for any other tied row
    const size_t sh = scan_block_bytes(call.d, call.kk, call.formula);
    if (sh > 150 * 1024) return fail(SKNNR_ERR_UNSUPPORTED, "n_neighbors = %d with d = %d does not fit the exact scan kernel", call.k, call.d);
    ScanArgs b{call, ix->refT.p, ix->fail_list2.p, ix->fail_count.p + 2, nullptr, nullptr, nullptr, nullptr};
    const bool chunked = call.d > kScanColChunk;
    const int nq_pass = scan_nq(call.formula);
    const long blocks = std::max<long>(1, std::min<long>((nq + nq_pass - 1) / nq_pass, kScanGridWg));
    HIP_TRY(launch::exact_scan(call.formula, chunked, b, blocks, sh, st));
    return SKNNR_OK;
}

// Device-resident core of sknnr_merge_shards.
int merge_shards_device(sknnr_index* ix, const double* xdev, long nq, const sknnr_query_opts* o, int n_shards,
                        const double* shard_val, const long* shard_idx, double* d_dist, long* d_idx, hipStream_t st) {
    const int kk = o->n_neighbors + (o->exclude_self ? 1 : 0);
    const bool affine = o->apply_affine != 0 && xdev != nullptr;
    const bool self_rows = xdev == nullptr;
    if (nq > 0x7fffffffL) return fail(SKNNR_ERR_UNSUPPORTED, "more than 2^31 - 1 query rows in one call");
    if (affine && ix->ks == 0) return fail(SKNNR_ERR_UNSUPPORTED, "d = %d > 128 with an affine map is outside the HIP envelope", ix->d);
    if (ix->ws_busy) HIP_TRY(hipStreamWaitEvent(st, ix->ev_ws, 0));
    // the transformed rows (the tied rows are re-scanned in them)
    const double* xq_call = self_rows ? ix->ref64.p + o->row_offset * ix->d : xdev;
    if (affine) {
        HIP_TRY(ix->xt.ensure((size_t)nq * ix->d));
        const long chunk = chunk_rows(ix->ks, 2);
        const long cap_pad = (std::min(chunk, nq) + kRowQuantum - 1) / kRowQuantum * kRowQuantum;
        HIP_TRY(ix->qimg.ensure((size_t)(cap_pad / 32) * 2 * ix->ks * 64));
        HIP_TRY(ix->qnc.ensure(cap_pad));
        for (long c0 = 0; c0 < nq; c0 += chunk) {
            const long n = std::min(chunk, nq - c0);
            const long n_pad = (n + kRowQuantum - 1) / kRowQuantum * kRowQuantum;
            int rc = launch_prep(ix, xdev + c0 * ix->d_in, n, n_pad, true, ix->xt.p + c0 * ix->d, st, o->check_finite != 0);
            if (rc) return rc;
        }
        xq_call = ix->xt.p;
    }
    SelectArgs call{};
    call.xq = xq_call;
    call.ref = ix->ref64.p;
    call.rn = ix->rn64.p;
    call.nq = nq;
    call.d = ix->d;
    call.n_ref = (int)ix->n_ref;
    call.k = o->n_neighbors;
    call.kk = kk;
    call.exclude_self = o->exclude_self ? 1 : 0;
    call.deterministic = o->deterministic ? 1 : 0;
    call.formula = o->formula;
    call.pow10_is_divisor = o->decimals < 0;
    call.pow10 = std::pow(10.0, std::abs(o->decimals));
    call.row_offset = o->row_offset;
    call.hw = ix->hw.p;
    call.hw_sum = ix->hw_sum;
    call.out_dist = d_dist;
    call.out_idx = d_idx;
    const size_t heaps = (size_t)nq * n_shards * kk;
    HIP_TRY(ix->slice_v.ensure(heaps));
    HIP_TRY(ix->slice_i.ensure(heaps));
    HIP_TRY(ix->fail_list2.ensure((size_t)nq));
    HIP_TRY(hipMemsetAsync(ix->fail_count.p, 0, 16, st));
    HIP_TRY(launch::pack_shards(shard_val, shard_idx, nq, n_shards, kk, ix->slice_v.p, ix->slice_i.p, st));
    int rc = merge_shards_formula(ix, call, n_shards, nq, st);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ix->ev_ws, st));
    ix->ws_busy = true;
    ix->stats.queries += nq;
    ix->stats.exact_only_queries += nq;
    return SKNNR_OK;
}

}  // namespace

extern "C" int sknnr_merge_shards(sknnr_index* ix, const double* q, int64_t nq, const sknnr_query_opts* o, int32_t n_shards,
                                  const double* shard_val, const int64_t* shard_idx, double* out_dist, int64_t* out_idx,
                                  int32_t mem, void* stream) {
    int rc = validate_call(ix, q, nq, o, out_idx);
    if (rc) return rc;
    if (o->query_dtype != SKNNR_DTYPE_F64) return fail(SKNNR_ERR_UNSUPPORTED, "the sharded entry points take float64 rows (query_dtype = 0)");
    if (n_shards < 1 || n_shards > 64) return fail(SKNNR_ERR_INVALID, "n_shards must be in [1, 64], got %d", n_shards);
    if ((!shard_val || !shard_idx) && nq > 0) return fail(SKNNR_ERR_INVALID, "shard candidate arrays are NULL");
    const int kk = o->n_neighbors + (o->exclude_self ? 1 : 0);
    if (kk > kScanSliceMaxKK) return fail(SKNNR_ERR_UNSUPPORTED, "the shard merge serves n_neighbors (+ self) <= %d", kScanSliceMaxKK);
    if (nq == 0) return SKNNR_OK;
    if (mem != SKNNR_MEM_DEVICE && mem != SKNNR_MEM_HOST) return fail(SKNNR_ERR_INVALID, "unknown memspace %d", mem);
    std::lock_guard<std::mutex> lock(ix->mtx);
    HIP_TRY(hipSetDevice(ix->device));
    if (mem == SKNNR_MEM_DEVICE)
        return merge_shards_device(ix, q, nq, o, n_shards, shard_val, (const long*)shard_idx, out_dist, (long*)out_idx, (hipStream_t)stream);
    const int k = o->n_neighbors;
    const int d_x = o->apply_affine ? ix->d_in : ix->d;
    const size_t n_cand = (size_t)nq * n_shards * kk;
    DevBuf<double> dq, dv, dd;
    DevBuf<long> dsi, di;
    if (q) {
        HIP_TRY(dq.ensure((size_t)nq * d_x));
        HIP_TRY(hipMemcpy(dq.p, q, (size_t)nq * d_x * sizeof(double), hipMemcpyHostToDevice));
    }
    HIP_TRY(dv.ensure(n_cand));
    HIP_TRY(dsi.ensure(n_cand));
    HIP_TRY(dd.ensure((size_t)nq * k));
    HIP_TRY(di.ensure((size_t)nq * k));
    HIP_TRY(hipMemcpy(dv.p, shard_val, n_cand * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dsi.p, shard_idx, n_cand * sizeof(long), hipMemcpyHostToDevice));
    rc = merge_shards_device(ix, q ? dq.p : nullptr, nq, o, n_shards, dv.p, dsi.p, dd.p, di.p, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (o->check_finite && q && (rc = poll_status(ix))) return rc;
    if (out_dist) HIP_TRY(hipMemcpy(out_dist, dd.p, (size_t)nq * k * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_idx, di.p, (size_t)nq * k * sizeof(long), hipMemcpyDeviceToHost));
    return SKNNR_OK;
}

// ----------------------------------------------------------------------------------------
// predict
// ----------------------------------------------------------------------------------------
static int launch_predict(sknnr_index* ix, const double* dist, const long* idx, const double* w, long nq, int k,
                          int mode, double* out, hipStream_t st) {
    PredictArgs a{};
    a.y = ix->y64.p;
    a.dist = dist;
    a.idx = idx;
    a.w = w;
    a.nq = nq;
    a.k = k;
    a.t = ix->t;
    a.mode = mode;
    a.out = out;
    HIP_TRY(launch::predict(a, st));
    // the reduction may read the handle's staging buffers: it is now the workspace's last user
    HIP_TRY(hipEventRecord(ix->ev_ws, st));
    ix->ws_busy = true;
    return SKNNR_OK;
}

extern "C" int sknnr_predict_from_neighbors(sknnr_index* ix, const double* dist, const int64_t* idx, const double* w,
                                            int64_t nq, int32_t k, int32_t mode, double* out_pred, int32_t mem,
                                            void* stream) {
    if (!ix) return fail(SKNNR_ERR_INVALID, "index is NULL");
    if (ix->t < 1) return fail(SKNNR_ERR_NO_TARGETS, "the index was created without targets");
    if (nq < 0 || k < 1 || !idx || !out_pred) return fail(SKNNR_ERR_INVALID, "bad argument");
    if (mode == SKNNR_WEIGHTS_DISTANCE && !dist) return fail(SKNNR_ERR_INVALID, "distance weights need dist");
    if (mode == SKNNR_WEIGHTS_EXPLICIT && !w) return fail(SKNNR_ERR_INVALID, "explicit weights need w");
    if (mode < 0 || mode > 2) return fail(SKNNR_ERR_INVALID, "unknown weight mode %d", mode);
    if (mem != SKNNR_MEM_DEVICE && mem != SKNNR_MEM_HOST) return fail(SKNNR_ERR_INVALID, "unknown memspace %d", mem);
    if (nq == 0) return SKNNR_OK;
    std::lock_guard<std::mutex> lock(ix->mtx);
    HIP_TRY(hipSetDevice(ix->device));
    if (mem == SKNNR_MEM_DEVICE)
        return launch_predict(ix, dist, (const long*)idx, w, nq, k, mode, out_pred, (hipStream_t)stream);
    hipStream_t st = nullptr;
    DevBuf<double> dd, dw, dout;  // freed by their destructors on every return path
    DevBuf<long> di;
    HIP_TRY(di.ensure((size_t)nq * k));
    HIP_TRY(dout.ensure((size_t)nq * ix->t));
    HIP_TRY(hipMemcpy(di.p, idx, (size_t)nq * k * sizeof(long), hipMemcpyHostToDevice));
    if (dist) {
        HIP_TRY(dd.ensure((size_t)nq * k));
        HIP_TRY(hipMemcpy(dd.p, dist, (size_t)nq * k * sizeof(double), hipMemcpyHostToDevice));
    }
    if (w) {
        HIP_TRY(dw.ensure((size_t)nq * k));
        HIP_TRY(hipMemcpy(dw.p, w, (size_t)nq * k * sizeof(double), hipMemcpyHostToDevice));
    }
    int rc = launch_predict(ix, dist ? dd.p : nullptr, di.p, w ? dw.p : nullptr, nq, k, mode, dout.p, st);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out_pred, dout.p, (size_t)nq * ix->t * sizeof(double), hipMemcpyDeviceToHost));
    return SKNNR_OK;
}

extern "C" int sknnr_predict(sknnr_index* ix, const void* q, int64_t nq, const sknnr_query_opts* o, double* out_pred,
                             double* out_dist, int64_t* out_idx, int32_t mem, void* stream) {
    if (!ix) return fail(SKNNR_ERR_INVALID, "index is NULL");
    if (ix->t < 1) return fail(SKNNR_ERR_NO_TARGETS, "the index was created without targets");
    if (!o) return fail(SKNNR_ERR_INVALID, "opts is NULL");
    if (!out_pred && nq > 0) return fail(SKNNR_ERR_INVALID, "out_pred is NULL");
    if (o->weight_mode == SKNNR_WEIGHTS_EXPLICIT)
        return fail(SKNNR_ERR_INVALID, "explicit weights go through sknnr_predict_from_neighbors");
    if (o->weight_mode != SKNNR_WEIGHTS_UNIFORM && o->weight_mode != SKNNR_WEIGHTS_DISTANCE)
        return fail(SKNNR_ERR_INVALID, "unknown weight mode %d", o->weight_mode);
    static int64_t dummy_idx;
    int rc = validate_call(ix, q, nq, o, out_idx ? out_idx : &dummy_idx);
    if (rc) return rc;
    if (nq == 0) return SKNNR_OK;
    if (mem != SKNNR_MEM_DEVICE && mem != SKNNR_MEM_HOST) return fail(SKNNR_ERR_INVALID, "unknown memspace %d", mem);
    std::lock_guard<std::mutex> lock(ix->mtx);
    HIP_TRY(hipSetDevice(ix->device));
    const int k = o->n_neighbors;

    if (mem == SKNNR_MEM_DEVICE) {
        hipStream_t st = (hipStream_t)stream;
        double* dd = out_dist;
        long* di = (long*)out_idx;
        if (!dd || !di) {
            // the staging buffers below belong to the workspace: wait for its previous user first
            if (ix->ws_busy) HIP_TRY(hipStreamWaitEvent(st, ix->ev_ws, 0));
        }
        if (!dd) {
            HIP_TRY(ix->dist_stage.ensure((size_t)nq * k));
            dd = ix->dist_stage.p;
        }
        if (!di) {
            HIP_TRY(ix->idx_stage.ensure((size_t)nq * k));
            di = ix->idx_stage.p;
        }
        rc = run_device(ix, q, nq, o, dd, di, st);
        if (rc) return rc;
        return launch_predict(ix, dd, di, nullptr, nq, k, o->weight_mode, out_pred, st);
    }
    if (q) return run_host_pipeline(ix, q, nq, o, out_dist, (long*)out_idx, out_pred);
    return run_self_rows(ix, nq, o, out_dist, (long*)out_idx, out_pred);
}

// ----------------------------------------------------------------------------------------
// streamed query tiles (raster ingestion)
// ----------------------------------------------------------------------------------------
struct sknnr_stream {
    HostPipe pipe;
    int64_t rows_pushed = 0;
};

extern "C" int sknnr_stream_begin(sknnr_index* ix, const sknnr_query_opts* o, int32_t want_dist, int32_t want_pred,
                                  sknnr_stream** out) {
    if (!out) return fail(SKNNR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    static int64_t dummy_idx;
    static const double dummy_q = 0.0;
    int rc = validate_call(ix, &dummy_q, 0, o, &dummy_idx);
    if (rc) return rc;
    if (o->exclude_self) return fail(SKNNR_ERR_INVALID, "a stream answers pushed rows: exclude_self is not available");
    if (want_pred) {
        if (ix->t < 1) return fail(SKNNR_ERR_NO_TARGETS, "the index was created without targets");
        if (o->weight_mode != SKNNR_WEIGHTS_UNIFORM && o->weight_mode != SKNNR_WEIGHTS_DISTANCE)
            return fail(SKNNR_ERR_INVALID, "a stream predicts with uniform or distance weights only");
    }
    std::lock_guard<std::mutex> lock(ix->mtx);
    if (ix->stream_open) return fail(SKNNR_ERR_INVALID, "a query stream is already open on this handle");
    HIP_TRY(hipSetDevice(ix->device));
    sknnr_stream* s = new (std::nothrow) sknnr_stream();
    if (!s) return fail(SKNNR_ERR_INVALID, "out of host memory");
    rc = pipe_open(s->pipe, ix, o, want_dist != 0, true, want_pred != 0);
    if (rc) {
        delete s;
        return rc;
    }
    ix->stream_open = true;
    *out = s;
    return SKNNR_OK;
}

extern "C" int sknnr_stream_push(sknnr_stream* s, const void* q, int64_t nq, double* out_dist, int64_t* out_idx,
                                 double* out_pred) {
    if (!s) return fail(SKNNR_ERR_INVALID, "stream is NULL");
    if (nq < 0) return fail(SKNNR_ERR_INVALID, "nq must be >= 0");
    if (nq == 0) return SKNNR_OK;
    if (!q) return fail(SKNNR_ERR_INVALID, "q is NULL");
    HostPipe& p = s->pipe;
    if (!out_idx && !out_pred) return fail(SKNNR_ERR_INVALID, "a push needs out_idx or out_pred");
    if (out_dist && !p.want_dist) return fail(SKNNR_ERR_INVALID, "the stream was opened without distances");
    if (out_pred && !p.want_pred) return fail(SKNNR_ERR_INVALID, "the stream was opened without predictions");
    std::lock_guard<std::mutex> lock(p.ix->mtx);
    HIP_TRY(hipSetDevice(p.ix->device));
    int rc = pipe_submit_rows(p, q, nq, out_dist, (long*)out_idx, out_pred);
    if (rc) {
        const std::string msg = g_last_error;
        pipe_abort(p);
        (void)hipDeviceSynchronize();
        g_last_error = msg;
        return rc;
    }
    s->rows_pushed += nq;
    return SKNNR_OK;
}

extern "C" int sknnr_stream_flush(sknnr_stream* s) {
    if (!s) return fail(SKNNR_ERR_INVALID, "stream is NULL");
    std::lock_guard<std::mutex> lock(s->pipe.ix->mtx);
    HIP_TRY(hipSetDevice(s->pipe.ix->device));
    return pipe_flush(s->pipe);
}

extern "C" int sknnr_stream_end(sknnr_stream* s, int64_t* rows_pushed) {
    if (!s) return SKNNR_OK;
    int rc;
    {
        std::lock_guard<std::mutex> lock(s->pipe.ix->mtx);
        rc = hipSetDevice(s->pipe.ix->device) == hipSuccess ? pipe_flush(s->pipe) : fail(SKNNR_ERR_HIP, "hipSetDevice failed");
        s->pipe.ix->stream_open = false;
    }
    if (rows_pushed) *rows_pushed = s->rows_pushed;
    delete s;
    return rc;
}

// ----------------------------------------------------------------------------------------
// crosswalk
// ----------------------------------------------------------------------------------------
extern "C" int sknnr_crosswalk(const int64_t* table, int64_t n_table, const int64_t* idx, int64_t n, int64_t* out,
                               int32_t device, int32_t mem, void* stream) {
    if (!table || !idx || !out || n < 0 || n_table < 1) return fail(SKNNR_ERR_INVALID, "bad argument");
    if (mem != SKNNR_MEM_DEVICE && mem != SKNNR_MEM_HOST) return fail(SKNNR_ERR_INVALID, "unknown memspace %d", mem);
    if (n == 0) return SKNNR_OK;
    HIP_TRY(hipSetDevice(device));
    if (mem == SKNNR_MEM_DEVICE) {
        HIP_TRY(launch::crosswalk((const long*)table, (const long*)idx, n, (long*)out, (hipStream_t)stream));
        return SKNNR_OK;
    }
    DevBuf<long> dt, di, dout;  // freed by their destructors on every return path
    HIP_TRY(dt.ensure(n_table));
    HIP_TRY(di.ensure(n));
    HIP_TRY(dout.ensure(n));
    HIP_TRY(hipMemcpy(dt.p, table, n_table * sizeof(long), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(di.p, idx, n * sizeof(long), hipMemcpyHostToDevice));
    HIP_TRY(launch::crosswalk(dt.p, di.p, n, dout.p, nullptr));
    HIP_TRY(hipMemcpy(out, dout.p, n * sizeof(long), hipMemcpyDeviceToHost));
    return SKNNR_OK;
}

// ----------------------------------------------------------------------------------------
// diagnostics
// ----------------------------------------------------------------------------------------
extern "C" int sknnr_debug_coarse_matrix(sknnr_index* ix, const double* q, int64_t nq, float* out, double* out_qnorm,
                                         double* out_scale, double* out_eps) {
    if (!ix || !q || !out || nq < 1) return fail(SKNNR_ERR_INVALID, "bad argument");
    if (ix->ks == 0) return fail(SKNNR_ERR_UNSUPPORTED, "no coarse image (d > 128)");
    if ((double)nq * (double)ix->n_ref > 16777216.0) return fail(SKNNR_ERR_INVALID, "nq * n_ref must be <= 2^24");
    std::lock_guard<std::mutex> lock(ix->mtx);
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipDeviceSynchronize());  // default stream below; the workspace may still be in use on another one
    const long n_pad = (nq + kRowQuantum - 1) / kRowQuantum * kRowQuantum;
    HIP_TRY(ix->xstage.ensure((size_t)nq * ix->d));
    HIP_TRY(hipMemcpy(ix->xstage.p, q, (size_t)nq * ix->d * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(ix->qimg.ensure((size_t)(n_pad / 32) * 2 * ix->ks * 64));
    HIP_TRY(ix->qnc.ensure(n_pad));
    int rc = launch_prep(ix, ix->xstage.p, nq, n_pad, false, nullptr, nullptr);
    if (rc) return rc;
    DevBuf<float> dout;
    HIP_TRY(dout.ensure((size_t)nq * ix->n_ref));
    const long n_tiles = (ix->n_ref + 31) / 32, nqb = (nq + 31) / 32;
    hipError_t e = launch::coarse_matrix(ix->ks, ix->rimg.p, ix->perm.p, ix->qimg.p, (int)ix->n_ref, nq, n_tiles, nqb, dout.p);
    if (e == hipSuccess) e = hipMemcpy(out, dout.p, (size_t)nq * ix->n_ref * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess && out_qnorm) e = hipMemcpy(out_qnorm, ix->qnc.p, nq * sizeof(double), hipMemcpyDeviceToHost);
    dout.release();
    if (e != hipSuccess) return fail(SKNNR_ERR_HIP, "debug matrix failed: %s", hipGetErrorString(e));
    if (out_scale) *out_scale = ix->s;
    if (out_eps) *out_eps = eps_units(ix->ks) * std::ldexp(1.0, -24);
    return SKNNR_OK;
}
