// host_sanitize_test.cc — drives front.hip (compiled as host C++ against hoststub/hip/hip_runtime.h) under
// AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer: the thread pool, the BLAS binding, the exact host
// preparation (ivf.py:125-127, fast_pq.py:202-204) and the streaming sessions' slot bookkeeping over a FAKE index
// whose "pipeline" completes a batch only a few calls after its submit (as the real pipelined index does).
// TEST INFRASTRUCTURE ONLY (tests/test_host_sanitizers.py runs it); prints "host sanitize: OK" and exits 0.
#include <math.h>
#include <stdio.h>

#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/tinyknn_hip.h"
#include "../kernels.h"

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

static std::string g_err;
int tk_fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
void tk_launch_prepare_queries(const float *, int64_t, int, const double *, int, int, void *, hipStream_t) {}

void tk_index_host_out_by_kernel(tk_index *ix, bool on);      // (C++ linkage: kernels.h)
// ---- the fake index: ids of a query = (sum of its row as int64) + j; a batch lands in its host buffer only when
// ---- `depth` later batches have been enqueued, or at a join (the pipelined index's behaviour seen from front.hip)
struct Pending {
    std::vector<int64_t> ids;
    int64_t *host_out;
    int64_t *dev_out;
};
struct tk_index {
    int d = 24, dq = 24, depth = 3;
    std::deque<Pending> q;
    std::mutex mu;
    int host_out_by_kernel = 0;
    void land(size_t keep)
    {
        while (q.size() > keep) {
            Pending &p = q.front();
            memcpy(p.dev_out, p.ids.data(), p.ids.size() * 8);
            memcpy(p.host_out, p.ids.data(), p.ids.size() * 8);
            q.pop_front();
        }
    }
};
void tk_index_host_out_by_kernel(tk_index *ix, bool on) { ix->host_out_by_kernel = on; }
extern "C" {
int tk_index_info(tk_index *ix, int64_t *out8)
{
    for (int i = 0; i < 8; i++) out8[i] = 0;
    out8[0] = ix->d;
    out8[1] = ix->dq;
    return TK_OK;
}
int tk_index_reserve(tk_index *, int64_t, int, int, int) { return TK_OK; }
int64_t tk_index_max_sub_batch(tk_index *, int, int, int) { return 1 << 20; }
int tk_index_pending(tk_index *ix)
{
    std::lock_guard<std::mutex> lk(ix->mu);
    return (int)ix->q.size();
}
int tk_index_join(tk_index *ix, void *)
{
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->land(0);
    return TK_OK;
}
void *tk_index_input_stream(tk_index *) { return nullptr; }
int tk_index_query_batch_dev_ex(tk_index *ix, const float *q_dev, const void *q_pq_dev, int q_pq_is_f64, int64_t nq,
                                int k, int, int, int64_t *out_ids_dev, int64_t *out_ids_host, void *done_event, void *)
{
    if (!q_dev || !q_pq_dev || !out_ids_dev || !out_ids_host || !done_event) return tk_fail(TK_ERR_ARG, "fake index: null");
    std::lock_guard<std::mutex> lk(ix->mu);
    Pending p;
    p.ids.resize((size_t)nq * k);
    for (int64_t r = 0; r < nq; r++) {
        double s = 0;
        for (int t = 0; t < ix->d; t++) s += (double)q_dev[r * ix->d + t] * 1024.0;
        // the table-build rows must be the padded / rotated ones: fold their last element in
        const double last = q_pq_is_f64 ? ((const double *)q_pq_dev)[r * ix->dq + ix->dq - 1]
                                        : (double)((const float *)q_pq_dev)[r * ix->dq + ix->dq - 1];
        for (int j = 0; j < k; j++) p.ids[(size_t)r * k + j] = (int64_t)llround(s) + j + (int64_t)llround(last * 16.0);
    }
    p.host_out = out_ids_host;
    p.dev_out = out_ids_dev;
    ix->q.push_back(std::move(p));
    ix->land((size_t)ix->depth);
    return TK_OK;
}
}

#define CHECK(c)                                                                              \
    do {                                                                                      \
        if (!(c)) {                                                                           \
            fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, g_err.c_str()); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

static uint32_t rng_state = 12345;
static float frand()
{
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float)((rng_state >> 8) & 0xffff) / 65536.0f - 0.5f;
}

int main(int argc, char **argv)
{
    CHECK(argc >= 2);      // path of a library exporting cblas_sdot / cblas_dgemv (numpy's, or any CBLAS)
    CHECK(tk_prepare_queries_host(nullptr, 0, 1, 0, nullptr, nullptr, 0, 0, nullptr) == TK_ERR_STATE);
    CHECK(tk_host_blas_bind("/nonexistent.so") == TK_ERR_ARG);
    CHECK(tk_host_blas_bind(argv[1]) == TK_OK);
    CHECK(tk_host_blas_bound() == 1);

    // -- exact preparation: every row against a scalar restatement, pool resized between regions, two caller threads
    const int d = 24, dq = 24, d_pad = 24;
    const int64_t nq = 5000;
    std::vector<float> raw((size_t)nq * d);
    for (float &v : raw) v = frand();
    std::vector<double> R((size_t)dq * d_pad);
    for (double &v : R) v = (double)frand();
    for (int threads : {1, 3, 8, 2}) {
        CHECK(tk_host_threads(threads) == threads);
        auto one = [&](int angular, bool rotate) {
            std::vector<float> qn((size_t)nq * d);
            std::vector<double> qp((size_t)nq * dq);
            CHECK(tk_prepare_queries_host(raw.data(), nq, d, angular, qn.data(), rotate ? R.data() : nullptr, dq, d_pad,
                                          rotate ? qp.data() : nullptr) == TK_OK);
            for (int64_t r = 0; r < nq; r += 97) {
                double nrm2 = 0;
                for (int t = 0; t < d; t++) nrm2 += (double)raw[r * d + t] * raw[r * d + t];
                for (int t = 0; t < d; t++) {
                    const float want = angular ? raw[r * d + t] / sqrtf((float)nrm2) : raw[r * d + t];
                    CHECK(fabsf(qn[r * d + t] - want) <= 1e-6f * (1.0f + fabsf(want)));
                }
                if (rotate) {
                    for (int j = 0; j < dq; j++) {
                        double acc = 0;
                        for (int t = 0; t < d; t++) acc += R[(size_t)j * d_pad + t] * (double)qn[r * d + t];
                        CHECK(fabs(qp[r * dq + j] - acc) <= 1e-9 * (1.0 + fabs(acc)));
                    }
                }
            }
        };
        std::thread a([&] { one(1, true); one(0, false); });
        std::thread b([&] { one(0, true); one(1, false); });
        a.join();
        b.join();
    }

    // -- streaming sessions over the fake index: batches of every size, more in flight than slots, waits out of order
    for (int rotated = 0; rotated < 2; rotated++) {
        tk_index ix;
        const int k = 10, n_slots = 4;
        const int64_t max_nq = 700;
        tk_stream *s = tk_stream_create(&ix, max_nq, k, 5, 0, 1, rotated ? R.data() : nullptr, d_pad, n_slots);
        CHECK(s != nullptr);
        CHECK(tk_stream_submit(s, raw.data(), max_nq + 1, nullptr) < 0);
        const int n_batches = 23;
        std::vector<std::vector<int64_t>> outs(n_batches);
        std::vector<int64_t> tickets, sizes, firsts;
        int64_t row = 0;
        for (int b = 0; b < n_batches; b++) {
            const int64_t n = 1 + (int64_t)((b * 131) % max_nq);
            if (row + n > nq) row = 0;
            outs[b].assign((size_t)n * k, -99);
            const int64_t t = tk_stream_submit(s, raw.data() + row * d, n, outs[b].data());
            CHECK(t == b);
            tickets.push_back(t);
            sizes.push_back(n);
            firsts.push_back(row);
            row += n;
            if (b % 5 == 4) CHECK(tk_stream_wait(s, tickets[b - 2]) == TK_OK);
        }
        CHECK(tk_stream_set_probes(s, 7, 0) == TK_OK);      // (drains)
        std::vector<float> qn((size_t)nq * d);
        std::vector<double> qp((size_t)nq * dq);
        CHECK(tk_prepare_queries_host(raw.data(), nq, d, 1, qn.data(), rotated ? R.data() : nullptr, dq, d_pad,
                                      rotated ? qp.data() : nullptr) == TK_OK);
        for (int b = 0; b < n_batches; b++)
            for (int64_t r = 0; r < sizes[b]; r++) {
                const int64_t g = firsts[b] + r;
                double sum = 0;
                for (int t = 0; t < d; t++) sum += (double)qn[g * d + t] * 1024.0;
                const double last = rotated ? qp[g * dq + dq - 1] : (double)qn[g * d + d - 1];
                for (int j = 0; j < k; j++)
                    CHECK(outs[b][(size_t)r * k + j] == (int64_t)llround(sum) + j + (int64_t)llround(last * 16.0));
            }
        // prepared rows in, two batches left in flight at destroy
        std::vector<int64_t> o1((size_t)100 * k), o2((size_t)50 * k);
        CHECK(tk_stream_submit_prepared(s, qn.data(), rotated ? (const void *)qp.data() : nullptr, 100, o1.data()) >= 0);
        CHECK(tk_stream_submit_prepared(s, qn.data() + 100 * d, rotated ? (const void *)(qp.data() + 100 * dq) : nullptr, 50,
                                        o2.data()) >= 0);
        CHECK(tk_stream_prepare_seconds(s) >= 0.0);
        tk_stream_destroy(s);
        CHECK(ix.q.empty());
    }
    CHECK(tk_host_threads(1) == 1);      // (joins the workers: nothing left running at exit)
    printf("host sanitize: OK\n");
    return 0;
}
