// front.hip — the host side of IVF.query in front of the device pipeline, exact and fast.
//
// What the reference does per query on the host before any PQ work (ivf.py:125-128,
// fast_pq.py:200-204):
//     q = ascontiguousarray(q, float32);  q /= np.linalg.norm(q)        (angular)
//     q_pq = pad1(q, 8) [ @ R.T ]                                        (R float64)
// np.linalg.norm of a float32 vector is sqrt(x.dot(x)) = sqrtf(cblas_sdot), and
// vector @ matrix is cblas_dgemv(ColMajor, Trans, d_pad, dq, 1, R, d_pad, x, 1, 0, y, 1)
// (numpy matmul.c.src, @TYPE@_gemv via the vector_matrix case).  Their summation orders
// belong to the BLAS build numpy links, so they are not restated: tk_host_blas_bind
// resolves cblas_sdot / cblas_dgemv FROM THAT SAME LIBRARY at run time and a small thread
// pool calls them row by row — bit-identical to numpy by construction (the binding checks
// it against numpy when it loads, tests/test_front_end.py on every run), and a few
// hundred times faster than a Python loop over rows.
//
// tk_stream_*: raw float32 queries on the host in, ids on the host out, as a pipeline of
// sub-batches: exact preparation (threads) -> pinned staging -> H2D on a copy stream ->
// tk_index_query_batch_dev_ex on a compute stream (its last stage copies the ids into
// pinned memory) -> the caller collects.  Preparation and copies of a batch overlap the
// kernels of the batches before it.
#include <dlfcn.h>
#include <math.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tinyknn_hip.h"
#include "kernels.h"

int tk_fail(int code, const std::string &msg);   // api.hip

// ---------------------------------------------------------------------------
// the BLAS numpy links
typedef float (*sdot64_t)(int64_t, const float *, int64_t, const float *, int64_t);
typedef float (*sdot32_t)(int, const float *, int, const float *, int);
typedef void (*dgemv64_t)(int, int, int64_t, int64_t, double, const double *, int64_t,
                          const double *, int64_t, double, double *, int64_t);
typedef void (*dgemv32_t)(int, int, int, int, double, const double *, int, const double *, int,
                          double, double *, int);

static struct HostBlas {
    void *handle = nullptr;
    sdot64_t sdot64 = nullptr;
    sdot32_t sdot32 = nullptr;
    dgemv64_t dgemv64 = nullptr;
    dgemv32_t dgemv32 = nullptr;
    std::string what;
} g_blas;

static inline float blas_sdot(int64_t n, const float *x)
{
    return g_blas.sdot64 ? g_blas.sdot64(n, x, 1, x, 1) : g_blas.sdot32((int)n, x, 1, x, 1);
}

// y (n) = A^T x, A column-major (m, n) with leading dimension m  ==  R (n, m) row-major
static inline void blas_dgemv_t(int64_t m, int64_t n, const double *A, const double *x, double *y)
{
    const int ColMajor = 102, Trans = 112;
    if (g_blas.dgemv64) g_blas.dgemv64(ColMajor, Trans, m, n, 1.0, A, m, x, 1, 0.0, y, 1);
    else g_blas.dgemv32(ColMajor, Trans, (int)m, (int)n, 1.0, A, (int)m, x, 1, 0.0, y, 1);
}

extern "C" int tk_host_blas_bind(const char *path)
{
    if (!path) return tk_fail(TK_ERR_ARG, "bad argument: null library path");
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return tk_fail(TK_ERR_ARG, std::string("bad argument: dlopen failed: ") + dlerror());
    HostBlas b;
    b.handle = h;
    // ILP64 builds suffix the symbols (numpy's wheels: scipy_ prefix + 64_ suffix)
    const char *s64[] = {"scipy_cblas_sdot64_", "cblas_sdot64_", nullptr};
    const char *g64[] = {"scipy_cblas_dgemv64_", "cblas_dgemv64_", nullptr};
    const char *s32[] = {"scipy_cblas_sdot", "cblas_sdot", nullptr};
    const char *g32[] = {"scipy_cblas_dgemv", "cblas_dgemv", nullptr};
    for (int i = 0; s64[i] && !b.sdot64; i++) {
        void *a = dlsym(h, s64[i]), *g = dlsym(h, g64[i]);
        if (a && g) { b.sdot64 = (sdot64_t)a; b.dgemv64 = (dgemv64_t)g; b.what = s64[i]; }
    }
    for (int i = 0; s32[i] && !b.sdot64 && !b.sdot32; i++) {
        void *a = dlsym(h, s32[i]), *g = dlsym(h, g32[i]);
        if (a && g) { b.sdot32 = (sdot32_t)a; b.dgemv32 = (dgemv32_t)g; b.what = s32[i]; }
    }
    if (!b.sdot64 && !b.sdot32) {
        dlclose(h);
        return tk_fail(TK_ERR_ARG, std::string("bad argument: no cblas_sdot/cblas_dgemv in ") + path);
    }
    g_blas = b;      // an earlier handle stays open: another thread may be inside it
    return TK_OK;
}

extern "C" int tk_host_blas_bound(void) { return (g_blas.sdot64 || g_blas.sdot32) ? 1 : 0; }

// ---------------------------------------------------------------------------
// a small pool: parallel_for over row ranges, the caller takes part
namespace {
struct Pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void(int64_t, int64_t)> fn;
    int64_t n = 0, grain = 1;
    std::atomic<int64_t> next{0}, done{0};
    std::atomic<int> in_drain{0};     // workers between reading `gen` and leaving drain()
    uint64_t gen = 0;
    bool stop = false;

    void worker()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
                in_drain.fetch_add(1, std::memory_order_relaxed);   // under the lock
            }
            drain();
            in_drain.fetch_sub(1, std::memory_order_release);
        }
    }
    void drain()
    {
        for (;;) {
            const int64_t a = next.fetch_add(grain, std::memory_order_relaxed);
            if (a >= n) return;
            const int64_t b = a + grain < n ? a + grain : n;
            fn(a, b);
            done.fetch_add(b - a, std::memory_order_release);
        }
    }
    void resize(int threads)
    {
        shutdown();
        stop = false;
        for (int i = 1; i < threads; i++) th.emplace_back([this] { worker(); });
    }
    void shutdown()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto &t : th) t.join();
        th.clear();
    }
    // The caller takes part and waits for the ITEMS, not for the workers: a worker that wakes
    // late finds nothing left and goes back to sleep.
    void run(int64_t count, int64_t g, std::function<void(int64_t, int64_t)> f)
    {
        if (count <= 0) return;
        if (th.empty() || count <= g) {
            f(0, count);
            return;
        }
        {
            std::unique_lock<std::mutex> lk(mu);
            // a late worker of the previous region may still be inside drain(): the job fields
            // change only while none is (workers enter drain under this lock)
            while (in_drain.load(std::memory_order_acquire) != 0) {
                lk.unlock();
                sched_yield();
                lk.lock();
            }
            fn = std::move(f);
            n = count;
            grain = g;
            next.store(0, std::memory_order_relaxed);
            done.store(0, std::memory_order_relaxed);
            gen++;
        }
        cv.notify_all();
        drain();
        while (done.load(std::memory_order_acquire) < count) sched_yield();
    }
    ~Pool() { shutdown(); }
};
Pool *g_pool = nullptr;
std::mutex g_pool_mu;      // one parallel region at a time
int g_threads = 0;

int default_threads()
{
    const char *e = getenv("TINYKNN_HOST_THREADS");
    if (e && atoi(e) > 0) return atoi(e);
    cpu_set_t set;
    int n = 1;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    return n > 32 ? 32 : (n < 1 ? 1 : n);
}
Pool &pool()
{
    if (!g_pool) {
        g_pool = new Pool();
        g_threads = default_threads();
        g_pool->resize(g_threads);
    }
    return *g_pool;
}
}   // namespace

extern "C" int tk_host_threads(int n)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (n > 0) {
        Pool &p = pool();
        if (n != g_threads) {
            g_threads = n > 256 ? 256 : n;
            p.resize(g_threads);
        }
    } else {
        (void)pool();
    }
    return g_threads;
}

// ---------------------------------------------------------------------------
// rows [a, b): ivf.py:125-127 into qn, fast_pq.py:202-204 into q_pq (rotated case only)
static void prepare_rows(const float *src, int64_t a, int64_t b, int d, int angular, float *qn,
                         const double *R, int dq, int d_pad, double *q_pq)
{
    std::vector<double> xs;
    if (R) xs.assign((size_t)d_pad, 0.0);
    for (int64_t r = a; r < b; r++) {
        float *row = qn + r * d;
        if (row != src + r * d) memcpy(row, src + r * d, (size_t)d * 4);
        if (angular) {
            // np.linalg.norm: sqrt(x.dot(x)); FLOAT_dot accumulates the chunk results in a
            // double and casts back, which is the identity for a single chunk
            const float nrm = sqrtf((float)(0.0 + (double)blas_sdot(d, row)));
            for (int t = 0; t < d; t++) row[t] = row[t] / nrm;
        }
        if (R) {
            for (int t = 0; t < d; t++) xs[(size_t)t] = (double)row[t];   // pad1 + cast to float64
            blas_dgemv_t(d_pad, dq, R, xs.data(), q_pq + r * dq);
        }
    }
}

extern "C" int tk_prepare_queries_host(const float *q_raw, int64_t nq, int d, int angular, float *qn,
                                       const double *R, int dq, int d_pad, double *q_pq)
{
    if (!tk_host_blas_bound())
        return tk_fail(TK_ERR_STATE, "tk_host_blas_bind has not been called: the exact host front end "
                                     "needs the BLAS numpy links");
    if (!q_raw || !qn || nq < 0 || d < 1) return tk_fail(TK_ERR_ARG, "bad argument: buffers / sizes");
    if (R && (!q_pq || dq < 1 || d_pad < d)) return tk_fail(TK_ERR_ARG, "bad argument: rotation");
    std::lock_guard<std::mutex> lk(g_pool_mu);
    const int64_t grain = R ? 32 : 256;
    pool().run(nq, grain, [=](int64_t a, int64_t b) {
        prepare_rows(q_raw, a, b, d, angular, qn, R, dq, d_pad, q_pq);
    });
    return TK_OK;
}

// ---------------------------------------------------------------------------
// streaming session
#define FHIP(x)                                                                              \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            char b_[512];                                                                    \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_),      \
                     __FILE__, __LINE__);                                                    \
            return tk_fail(TK_ERR_HIP, b_);                                                  \
        }                                                                                    \
    } while (0)

// How a session moves a batch's rows between page-locked host memory and HBM.
//   0  hipMemcpyAsync (copy engine) on the index's input stream + the pad kernel; ids back by
//      hipMemcpyAsync behind the rescoring kernel
//   1  (default) kernels that read / write the page-locked buffers directly over PCIe: one
//      ingest launch (copy + pad1 in one pass) and one egress launch, so that a stream never
//      alternates between copy-engine commands and kernel dispatches
// A/B in profiles/r02_raw_stream_ab.md
static int stream_copy_mode() { return 1; }     // (0, the copy-engine form, was measured slower: profiles/r02_raw_stream_ab.md)

// rows of page-locked host memory -> q (nq, d) and, when dq > d, the zero-padded table-build
// rows (nq, dq) (fast_pq.py:202 pad1), 16 bytes per lane and load where the row length allows
__global__ void ingest_rows_kernel(const float *__restrict__ src, int64_t nq, int d, int dq,
                                   float *__restrict__ q, float *__restrict__ qpad)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nt = (int64_t)gridDim.x * blockDim.x;
    if ((d & 3) == 0 && (dq & 3) == 0) {
        const int d4 = d >> 2, dq4 = dq >> 2;
        const float4 *s4 = (const float4 *)src;
        float4 *q4 = (float4 *)q, *p4 = (float4 *)qpad;
        for (int64_t i = tid; i < nq * d4; i += nt) {
            const float4 v = s4[i];
            q4[i] = v;
            if (qpad) {
                const int64_t r = i / d4;
                p4[r * dq4 + (i - r * d4)] = v;
            }
        }
        if (qpad)
            for (int64_t i = tid; i < nq * (dq4 - d4); i += nt) {
                const int64_t r = i / (dq4 - d4);
                p4[r * dq4 + d4 + (i - r * (dq4 - d4))] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        return;
    }
    for (int64_t i = tid; i < nq * d; i += nt) {
        const float v = src[i];
        q[i] = v;
        if (qpad) {
            const int64_t r = i / d;
            qpad[r * dq + (i - r * d)] = v;
        }
    }
    if (qpad)
        for (int64_t i = tid; i < nq * (dq - d); i += nt) {
            const int64_t r = i / (dq - d);
            qpad[r * dq + d + (i - r * (dq - d))] = 0.f;
        }
}

// n 8-byte words HBM -> page-locked host memory (or any other pair of device-visible buffers)
__global__ void copy_words_kernel(const uint64_t *__restrict__ src, int64_t n, uint64_t *__restrict__ dst)
{
    const int64_t nt = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nt) dst[i] = src[i];
}

void tk_launch_copy_words(const void *src, int64_t n_words, void *dst, hipStream_t st)
{
    if (n_words <= 0) return;
    int64_t g = (n_words + 255) / 256;
    if (g > 128) g = 128;
    hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)g), dim3(256), 0, st, (const uint64_t *)src,
                       n_words, (uint64_t *)dst);
}

struct StreamSlot {
    float *h_q = nullptr;
    double *h_qp = nullptr;
    int64_t *h_out = nullptr;
    float *d_q = nullptr;
    void *d_qp = nullptr;
    int64_t *d_out = nullptr;
    hipEvent_t in_done = nullptr, out_done = nullptr;
    int64_t nq = 0;
    int64_t *user_out = nullptr;
    int64_t ticket = -1;
    bool busy = false;
};

struct tk_stream {
    tk_index *ix = nullptr;
    int d = 0, dq = 0, k = 0, n_probes = 0, pass_1 = 0, angular = 0, d_pad = 0;
    bool rotated = false;
    std::vector<double> R;
    int64_t max_nq = 0;
    std::vector<StreamSlot> slots;
    hipStream_t comp_st = nullptr;
    int64_t submitted = 0;
    double prep_s = 0;      // host preparation time, summed over submits
};

static void stream_free(tk_stream *s)
{
    if (!s) return;
    for (StreamSlot &x : s->slots) {
        if (x.h_q) (void)hipHostFree(x.h_q);
        if (x.h_qp) (void)hipHostFree(x.h_qp);
        if (x.h_out) (void)hipHostFree(x.h_out);
        if (x.d_q) (void)hipFree(x.d_q);
        if (x.d_qp) (void)hipFree(x.d_qp);
        if (x.d_out) (void)hipFree(x.d_out);
        if (x.in_done) (void)hipEventDestroy(x.in_done);
        if (x.out_done) (void)hipEventDestroy(x.out_done);
    }
    if (s->comp_st) (void)hipStreamDestroy(s->comp_st);
    delete s;
}

extern "C" tk_stream *tk_stream_create(tk_index *ix, int64_t max_nq, int k, int n_probes, int pass_1,
                                       int angular, const double *R, int d_pad, int n_slots)
{
    int64_t info[8];
    if (!ix || tk_index_info(ix, info) != TK_OK) {
        tk_fail(TK_ERR_ARG, "bad argument: index");
        return nullptr;
    }
    if (!tk_host_blas_bound()) {
        tk_fail(TK_ERR_STATE, "tk_host_blas_bind has not been called: the exact host front end needs "
                              "the BLAS numpy links");
        return nullptr;
    }
    const int d = (int)info[0], dq = (int)info[1];
    if (max_nq < 1 || k < 1 || n_probes < 1 || n_slots < 2 || n_slots > 64 ||
        (R ? d_pad < d : dq < d)) {
        tk_fail(TK_ERR_ARG, "bad argument: tk_stream_create sizes");
        return nullptr;
    }
    // one sub-batch per submit: the index splits larger batches, and the completion event
    // belongs to one sub-batch
    if (tk_index_reserve(ix, max_nq, k, n_probes, pass_1) != TK_OK) return nullptr;
    if (max_nq > tk_index_max_sub_batch(ix, k, n_probes, pass_1)) {
        tk_fail(TK_ERR_ARG, "bad argument: max_nq exceeds one sub-batch of this index/n_probes "
                            "(tk_index_max_sub_batch)");
        return nullptr;
    }
    tk_stream *s = new tk_stream();
    s->ix = ix; s->d = d; s->dq = dq; s->k = k; s->n_probes = n_probes; s->pass_1 = pass_1;
    s->angular = angular; s->rotated = R != nullptr; s->d_pad = R ? d_pad : dq; s->max_nq = max_nq;
    if (R) s->R.assign(R, R + (size_t)dq * d_pad);
    s->slots.resize((size_t)n_slots);
    // The scan chain of the session runs on the NULL stream: HIP maps streams onto four
    // hardware queues and the pipelined index already owns three (front + two replay streams);
    // a stream of our own would be a fifth whenever anything in the process has touched the
    // NULL stream (torch does), and was seen to land on the queue of a replay stream, which
    // serialised the two replays: 1.03 ms per batch instead of 0.7 (profiles/r02_raw_stream_ab.md).
    bool ok = true;
    for (StreamSlot &x : s->slots) {
        if (!ok) break;
        ok = hipHostMalloc((void **)&x.h_q, (size_t)max_nq * d * 4, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void **)&x.h_out, (size_t)max_nq * k * 8, hipHostMallocDefault) == hipSuccess &&
             hipMalloc((void **)&x.d_q, (size_t)max_nq * d * 4) == hipSuccess &&
             hipMalloc((void **)&x.d_out, (size_t)max_nq * k * 8) == hipSuccess &&
             hipEventCreateWithFlags(&x.in_done, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&x.out_done, hipEventDisableTiming) == hipSuccess;
        if (ok && s->rotated)
            ok = hipHostMalloc((void **)&x.h_qp, (size_t)max_nq * dq * 8, hipHostMallocDefault) == hipSuccess &&
                 hipMalloc(&x.d_qp, (size_t)max_nq * dq * 8) == hipSuccess;
        else if (ok && dq > d)
            ok = hipMalloc(&x.d_qp, (size_t)max_nq * dq * 4) == hipSuccess;
    }
    if (!ok) {
        tk_fail(TK_ERR_HIP, "tk_stream_create: pinned/device allocation failed");
        stream_free(s);
        return nullptr;
    }
    return s;
}

static int stream_collect(tk_stream *s, StreamSlot &x)
{
    if (!x.busy) return TK_OK;
    // the batch's last stage is enqueued up to three calls after its submit: make sure it is
    const int64_t newer = s->submitted - 1 - x.ticket;
    if (newer < tk_index_pending(s->ix)) {
        int r = tk_index_join(s->ix, s->comp_st);
        if (r != TK_OK) return r;
    }
    FHIP(hipEventSynchronize(x.out_done));
    if (x.user_out) memcpy(x.user_out, x.h_out, (size_t)x.nq * s->k * 8);
    x.busy = false;
    return TK_OK;
}

// stage 2 of a submit: slot x holds the prepared rows in pinned memory
static int64_t stream_enqueue(tk_stream *s, StreamSlot &x, int64_t nq, int64_t *out_ids)
{
    // inputs go in on the stream where the batch's first kernel runs (no stream of our own:
    // HIP has four hardware queues and the pipelined index uses them all)
    hipStream_t in_st = (hipStream_t)tk_index_input_stream(s->ix);
    if (!in_st) in_st = s->comp_st;
    const void *qp = x.d_q;
    const bool pad = !s->rotated && s->dq > s->d;     // pad1: zeros behind the row, exact
    if (stream_copy_mode() == 1) {
        const int64_t items = nq * (int64_t)s->d / 4;
        int64_t g = (items + 255) / 256;
        g = g < 1 ? 1 : (g > 256 ? 256 : g);
        hipLaunchKernelGGL(ingest_rows_kernel, dim3((unsigned)g), dim3(256), 0, in_st, x.h_q, nq, s->d,
                           s->dq, x.d_q, pad ? (float *)x.d_qp : nullptr);
        if (s->rotated)
            tk_launch_copy_words(x.h_qp, nq * (int64_t)s->dq, x.d_qp, in_st);
    } else {
        FHIP(hipMemcpyAsync(x.d_q, x.h_q, (size_t)nq * s->d * 4, hipMemcpyHostToDevice, in_st));
        if (s->rotated)
            FHIP(hipMemcpyAsync(x.d_qp, x.h_qp, (size_t)nq * s->dq * 8, hipMemcpyHostToDevice, in_st));
        if (pad) tk_launch_prepare_queries(x.d_q, nq, s->d, nullptr, s->dq, s->dq, x.d_qp, in_st);
    }
    if (pad || s->rotated) qp = x.d_qp;
    // No hand-over to the compute stream: in pipelined mode the rows arrive on the index's front
    // stream, where the batch's first kernel (the table build) is enqueued next, and every
    // later consumer of the batch is ordered behind that kernel by the index's own events.
    // (Making the compute stream wait for them put each scan launch behind the previous
    // batch's coarse replay: 0.5 ms of idle chip per batch, profiles/r02_raw_stream_ab.md.)
    // mode 1: the ids leave through copy_words_kernel instead of the copy engine
    tk_index_host_out_by_kernel(s->ix, stream_copy_mode() == 1);
    int r = tk_index_query_batch_dev_ex(s->ix, x.d_q, qp, s->rotated ? 1 : 0, nq, s->k, s->n_probes,
                                        s->pass_1, x.d_out, x.h_out, x.out_done, s->comp_st);
    if (r != TK_OK) return r;
    x.nq = nq;
    x.user_out = out_ids;
    x.ticket = s->submitted;
    x.busy = true;
    return s->submitted++;
}

extern "C" int64_t tk_stream_submit(tk_stream *s, const float *q_raw, int64_t nq, int64_t *out_ids)
{
    if (!s || !q_raw || !out_ids || nq < 1 || nq > s->max_nq)
        return tk_fail(TK_ERR_ARG, "bad argument: tk_stream_submit (1 <= nq <= max_nq)");
    StreamSlot &x = s->slots[(size_t)(s->submitted % (int64_t)s->slots.size())];
    int r = stream_collect(s, x);
    if (r != TK_OK) return r;
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    r = tk_prepare_queries_host(q_raw, nq, s->d, s->angular, x.h_q, s->rotated ? s->R.data() : nullptr,
                                s->dq, s->d_pad, x.h_qp);
    if (r != TK_OK) return r;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    s->prep_s += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    return stream_enqueue(s, x, nq, out_ids);
}

extern "C" int64_t tk_stream_submit_prepared(tk_stream *s, const float *qn, const void *q_pq,
                                             int64_t nq, int64_t *out_ids)
{
    if (!s || !qn || !out_ids || nq < 1 || nq > s->max_nq || (s->rotated && !q_pq))
        return tk_fail(TK_ERR_ARG, "bad argument: tk_stream_submit_prepared (1 <= nq <= max_nq)");
    StreamSlot &x = s->slots[(size_t)(s->submitted % (int64_t)s->slots.size())];
    int r = stream_collect(s, x);
    if (r != TK_OK) return r;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        const int d = s->d, dq = s->dq;
        float *hq = x.h_q;
        double *hqp = s->rotated ? x.h_qp : nullptr;
        const double *src_qp = (const double *)q_pq;
        pool().run(nq, 1024, [=](int64_t a, int64_t b) {
            memcpy(hq + a * d, qn + a * d, (size_t)(b - a) * d * 4);
            if (hqp) memcpy(hqp + a * dq, src_qp + a * dq, (size_t)(b - a) * dq * 8);
        });
    }
    return stream_enqueue(s, x, nq, out_ids);
}

extern "C" int tk_stream_wait(tk_stream *s, int64_t ticket)
{
    if (!s || ticket < 0 || ticket >= s->submitted)
        return tk_fail(TK_ERR_ARG, "bad argument: tk_stream_wait ticket");
    StreamSlot &x = s->slots[(size_t)(ticket % (int64_t)s->slots.size())];
    if (!x.busy || x.ticket != ticket) return TK_OK;     // collected already
    return stream_collect(s, x);
}

extern "C" int tk_stream_drain(tk_stream *s)
{
    if (!s) return tk_fail(TK_ERR_ARG, "bad argument: null stream");
    int r = tk_index_join(s->ix, s->comp_st);
    for (StreamSlot &x : s->slots)
        if (r == TK_OK) r = stream_collect(s, x);
    return r;
}

// n_probes / pass_1 of the submits that follow (drains first; the page-locked staging and the
// device buffers of the session do not depend on them, only the index's workspaces do)
extern "C" int tk_stream_set_probes(tk_stream *s, int n_probes, int pass_1)
{
    if (!s || n_probes < 1) return tk_fail(TK_ERR_ARG, "bad argument: tk_stream_set_probes");
    int r = tk_stream_drain(s);
    if (r != TK_OK) return r;
    if (n_probes == s->n_probes && pass_1 == s->pass_1) return TK_OK;
    r = tk_index_reserve(s->ix, s->max_nq, s->k, n_probes, pass_1);
    if (r != TK_OK) return r;
    if (s->max_nq > tk_index_max_sub_batch(s->ix, s->k, n_probes, pass_1))
        return tk_fail(TK_ERR_ARG, "bad argument: max_nq exceeds one sub-batch of this index at these "
                                   "n_probes (tk_index_max_sub_batch)");
    s->n_probes = n_probes;
    s->pass_1 = pass_1;
    return TK_OK;
}

extern "C" double tk_stream_prepare_seconds(tk_stream *s) { return s ? s->prep_s : 0.0; }

extern "C" void tk_stream_destroy(tk_stream *s)
{
    if (!s) return;
    (void)tk_stream_drain(s);
    (void)hipStreamSynchronize(s->comp_st);      // (the NULL stream when comp_st is null)
    stream_free(s);
}
