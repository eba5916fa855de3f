// api.hip — the C ABI of libtinyknn_hip.so (declared in include/tinyknn_hip.h).
// Host-pointer entry points stage their arguments into HBM, run the gfx950
// kernels and copy results back; the tk_index_* entry points keep the index
// resident and only enqueue kernels.  There is no CPU implementation behind any of
// these calls.
#include <algorithm>
#include <cmath>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/tinyknn_hip.h"
#include "kernels.h"

static thread_local std::string g_err;

static int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
int tk_fail(int code, const std::string &msg) { return fail(code, msg); }   // front.hip

#define HIPCHECK(x)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            char b_[512];                                                                    \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_),      \
                     __FILE__, __LINE__);                                                    \
            return fail(TK_ERR_HIP, b_);                                                     \
        }                                                                                    \
    } while (0)

#define ARGCHECK(cond, msg)                                                                  \
    do {                                                                                     \
        if (!(cond)) return fail(TK_ERR_ARG, std::string("bad argument: ") + msg);           \
    } while (0)

extern "C" const char *tk_last_error(void) { return g_err.c_str(); }
extern "C" int tk_version(void) { return 1; }

extern "C" int tk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int tk_set_device(int device)
{
    HIPCHECK(hipSetDevice(device));
    return TK_OK;
}

static int require_gpu()
{
    if (tk_device_count() <= 0)
        return fail(TK_ERR_HIP, "no HIP device visible: libtinyknn_hip has no CPU fallback");
    return TK_OK;
}

// growable device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return TK_OK;
        if (p) HIPCHECK(hipFree(p));
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        HIPCHECK(hipMalloc(&p, want));
        cap = want;
        return TK_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T *as() const { return (T *)p; }
};

#define TRY(x)                         \
    do {                               \
        int r_ = (x);                  \
        if (r_ != TK_OK) return r_;    \
    } while (0)

#define TK_DBG_SYNC(tag)                                                                       \
    do {                                                                                       \
        static const int on_ = getenv("TINYKNN_DEBUG_SYNC") ? 1 : 0;                           \
        if (on_) {                                                                             \
            fprintf(stderr, "[dbg] %s ...", tag);                                              \
            fflush(stderr);                                                                    \
            hipError_t e_ = hipDeviceSynchronize();                                            \
            fprintf(stderr, " %s\n", hipGetErrorString(e_));                                   \
            fflush(stderr);                                                                    \
        }                                                                                      \
    } while (0)

// ---------------------------------------------------------------------------
// scratch of the host-pointer entry points
struct Scratch {
    std::mutex mu;
    DevBuf ref, tiled, tables, out, hidx, hval, labels, slots_i, slots_l, q, rows, cand, pos,
        centers, shift, scale, tabs8, mins;
};
static Scratch &scratch()
{
    static Scratch s;
    return s;
}

static int stage_codes(Scratch &S, const uint64_t *data, int64_t chunks, int M, hipStream_t st)
{
    const int P = M / 2;
    size_t ref_bytes = (size_t)chunks * M * 8;
    TRY(S.ref.ensure(ref_bytes));
    TRY(S.tiled.ensure((size_t)tk_tiled_uint4s(chunks, P) * 16));
    HIPCHECK(hipMemcpyAsync(S.ref.p, data, ref_bytes, hipMemcpyHostToDevice, st));
    tk_launch_retile(S.ref.as<uint4>(), S.tiled.as<uint4>(), chunks, P, st);
    return TK_OK;
}

extern "C" int tk_estimate_pq_batch(const uint64_t *data, int64_t chunks, int M,
                                    const uint64_t *tables, int64_t nq, uint64_t *out, int signd,
                                    int order)
{
    TRY(require_gpu());
    ARGCHECK(chunks >= 0 && nq >= 0, "negative size");
    ARGCHECK(M >= 2 && M % 2 == 0, "M must be even");
    ARGCHECK(order == TK_ORDER_SSE || order == TK_ORDER_AVX, "order");
    ARGCHECK(nq <= 65535, "at most 65535 tables per call");
    if (chunks == 0 || nq == 0) return TK_OK;
    Scratch &S = scratch();
    std::lock_guard<std::mutex> lk(S.mu);
    hipStream_t st = 0;
    TRY(stage_codes(S, data, chunks, M, st));
    TRY(S.tables.ensure((size_t)nq * M * 16));
    TRY(S.out.ensure((size_t)nq * chunks * 16));
    HIPCHECK(hipMemcpyAsync(S.tables.p, tables, (size_t)nq * M * 16, hipMemcpyHostToDevice, st));
    tk_launch_scan_flat(S.tiled.as<uint4>(), chunks, M, S.tables.as<uint4>(), nq,
                        S.out.as<uint4>(), chunks, nullptr, 0, signd, order, st);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(out, S.out.p, (size_t)nq * chunks * 16, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    return TK_OK;
}

extern "C" int tk_estimate_pq(const uint64_t *data, int64_t chunks, int M, const uint64_t *tables,
                              uint64_t *out, int signd, int order)
{
    return tk_estimate_pq_batch(data, chunks, M, tables, 1, out, signd, order);
}

extern "C" int tk_query_pq(const uint64_t *data, int64_t chunks, int M, int64_t n,
                           const uint64_t *tables, int64_t *indices, int32_t *vals, int R,
                           int signd, const int64_t *labels, int order)
{
    TRY(require_gpu());
    ARGCHECK(chunks >= 0 && R >= 1, "sizes");
    ARGCHECK(M >= 2 && M % 2 == 0, "M must be even");
    ARGCHECK(order == TK_ORDER_SSE || order == TK_ORDER_AVX, "order");
    ARGCHECK(chunks < (1ll << 31) / 16, "list too long");
    ARGCHECK((size_t)R * 12 + 16 <= 64 * 1024, "heap larger than 64 KiB of LDS (R <= 5460)");
    if (chunks == 0) return TK_OK;
    Scratch &S = scratch();
    std::lock_guard<std::mutex> lk(S.mu);
    hipStream_t st = 0;
    TRY(stage_codes(S, data, chunks, M, st));
    TRY(S.tables.ensure((size_t)M * 16));
    TRY(S.out.ensure((size_t)chunks * 16));
    TRY(S.hidx.ensure((size_t)R * 8));
    TRY(S.hval.ensure((size_t)R * 4));
    TRY(S.slots_i.ensure(3 * sizeof(int)));
    TRY(S.slots_l.ensure(sizeof(int64_t)));
    // labels beyond n are never read (pos < n is tested first, _fast_pq_256.pyx:111-114)
    int64_t nlab = n < 16 * chunks ? n : 16 * chunks;
    if (nlab < 0) nlab = 0;
    if (labels && nlab > 0) {
        TRY(S.labels.ensure((size_t)nlab * 8));
        HIPCHECK(hipMemcpyAsync(S.labels.p, labels, (size_t)nlab * 8, hipMemcpyHostToDevice, st));
    }
    HIPCHECK(hipMemcpyAsync(S.tables.p, tables, (size_t)M * 16, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(S.hidx.p, indices, (size_t)R * 8, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(S.hval.p, vals, (size_t)R * 4, hipMemcpyHostToDevice, st));
    int64_t nclamp = n > (int64_t)0x7fffffff ? 0x7fffffff : (n < 0 ? 0 : n);
    int si[3] = {0, (int)chunks, (int)nclamp};
    int64_t sl[1] = {labels ? 0 : -1};
    HIPCHECK(hipMemcpyAsync(S.slots_i.p, si, sizeof si, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(S.slots_l.p, sl, sizeof sl, hipMemcpyHostToDevice, st));
    const int64_t cap_min = (chunks + 15) / 16 * 16;
    TRY(S.mins.ensure((size_t)cap_min));
    tk_launch_scan_flat(S.tiled.as<uint4>(), chunks, M, S.tables.as<uint4>(), 1, S.out.as<uint4>(),
                        chunks, S.mins.as<uint8_t>(), cap_min, signd, order, st);
    tk_launch_heap_replay(S.out.as<uint4>(), chunks, 1, S.slots_i.as<int>(),
                          S.slots_i.as<int>() + 2, S.slots_l.as<int64_t>(), 1,
                          S.labels.as<int64_t>(), S.hidx.as<int64_t>(), S.hval.as<int32_t>(), R,
                          signd, 1, nullptr, st, S.mins.as<uint8_t>(), cap_min);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(indices, S.hidx.p, (size_t)R * 8, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipMemcpyAsync(vals, S.hval.p, (size_t)R * 4, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    return TK_OK;
}

// init_heap (_fast_pq.pyx:311-315): two HOST arrays filled with constants — nothing for the device to
// do (a fill kernel plus two copies back cost 45 us per call); like every entry point it still refuses
// to run on a machine without a GPU
extern "C" int tk_init_heap(int64_t *indices, int32_t *vals, int R, int signd)
{
    static const int have_gpu = tk_device_count();
    if (have_gpu <= 0) return require_gpu();
    ARGCHECK(R >= 0, "R");
    ARGCHECK(R == 0 || (indices && vals), "null heap");
    for (int i = 0; i < R; i++) {
        indices[i] = -1;
        vals[i] = signd ? 127 : 255;
    }
    return TK_OK;
}

static int heap_insert_host(int64_t *indices, int32_t *vals, int R, int64_t i, int32_t v, int is)
{
    TRY(require_gpu());
    ARGCHECK(R >= 1, "R");
    ARGCHECK((size_t)R * 12 + 16 <= 64 * 1024, "heap larger than 64 KiB of LDS (R <= 5460)");
    Scratch &S = scratch();
    std::lock_guard<std::mutex> lk(S.mu);
    hipStream_t st = 0;
    TRY(S.hidx.ensure((size_t)R * 8));
    TRY(S.hval.ensure((size_t)R * 4));
    HIPCHECK(hipMemcpyAsync(S.hidx.p, indices, (size_t)R * 8, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(S.hval.p, vals, (size_t)R * 4, hipMemcpyHostToDevice, st));
    tk_launch_heap_insert(S.hidx.as<int64_t>(), S.hval.as<int32_t>(), R, i, v, is, st);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(indices, S.hidx.p, (size_t)R * 8, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipMemcpyAsync(vals, S.hval.p, (size_t)R * 4, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    return TK_OK;
}

extern "C" int tk_heap_insert(int64_t *indices, int32_t *vals, int R, int64_t i, int32_t v)
{
    return heap_insert_host(indices, vals, R, i, v, 0);
}
extern "C" int tk_heap_insert_is(int64_t *indices, int32_t *vals, int R, int64_t i, int32_t v)
{
    return heap_insert_host(indices, vals, R, i, v, 1);
}

extern "C" int tk_build_tables(const float *centers, int dq, int dpb, int f_order, const void *q,
                               int q_is_f64, int64_t nq, double aux0, double aux1, int signd,
                               uint8_t *tables, void *shift, double *scale)
{
    TRY(require_gpu());
    ARGCHECK(dpb >= 1 && dpb <= 32, "dims_per_block must be in 1..32");
    ARGCHECK(dq >= dpb && dq % dpb == 0, "dq must be a multiple of dims_per_block");
    ARGCHECK(nq >= 0, "nq");
    const int M = dq / dpb;
    const size_t esz = q_is_f64 ? 8 : 4;
    ARGCHECK(((size_t)16 * M + 640) * esz <= 64 * 1024 && 16 * M <= 8192,
             "too many blocks for the LDS table");
    if (nq == 0) return TK_OK;
    Scratch &S = scratch();
    std::lock_guard<std::mutex> lk(S.mu);
    hipStream_t st = 0;
    TRY(S.centers.ensure((size_t)16 * dq * 4));
    TRY(S.q.ensure((size_t)nq * dq * esz));
    TRY(S.tabs8.ensure((size_t)nq * M * 16));
    TRY(S.shift.ensure((size_t)nq * esz));
    TRY(S.scale.ensure((size_t)nq * 8));
    HIPCHECK(hipMemcpyAsync(S.centers.p, centers, (size_t)16 * dq * 4, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(S.q.p, q, (size_t)nq * dq * esz, hipMemcpyHostToDevice, st));
    tk_launch_build_tables(S.centers.as<float>(), dq, dpb, f_order, S.q.p, q_is_f64, nq, aux0, aux1,
                           signd, S.tabs8.as<uint8_t>(), S.shift.p, S.scale.as<double>(), st);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(tables, S.tabs8.p, (size_t)nq * M * 16, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipMemcpyAsync(shift, S.shift.p, (size_t)nq * esz, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipMemcpyAsync(scale, S.scale.p, (size_t)nq * 8, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    return TK_OK;
}

extern "C" int64_t tk_knn_brute1(const void *x, int x_is_f64, const void *Y, int y_is_f64,
                                 int64_t n, int d, int64_t k, int64_t *out_pos)
{
    const size_t xsz = x_is_f64 ? 8 : 4, ysz = y_is_f64 ? 8 : 4;
    int r = require_gpu();
    if (r != TK_OK) return r;
    if (n < 0 || d < 1 || k < 0) return fail(TK_ERR_ARG, "bad argument: sizes");
    if (n > 4096) return fail(TK_ERR_ARG, "bad argument: at most 4096 candidate rows");
    int64_t kk = k < n ? k : n;
    if (kk == 0) return 0;
    Scratch &S = scratch();
    std::lock_guard<std::mutex> lk(S.mu);
    hipStream_t st = 0;
    std::vector<int64_t> cand((size_t)n);
    for (int64_t i = 0; i < n; i++) cand[i] = i;
#define HC(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) return fail(TK_ERR_HIP, hipGetErrorString(e_));      \
    } while (0)
    if ((r = S.q.ensure((size_t)d * xsz)) || (r = S.rows.ensure((size_t)n * d * ysz)) ||
        (r = S.cand.ensure((size_t)n * 8)) || (r = S.pos.ensure((size_t)kk * 8)))
        return r;
    HC(hipMemcpyAsync(S.q.p, x, (size_t)d * xsz, hipMemcpyHostToDevice, st));
    HC(hipMemcpyAsync(S.rows.p, Y, (size_t)n * d * ysz, hipMemcpyHostToDevice, st));
    HC(hipMemcpyAsync(S.cand.p, cand.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
    tk_launch_rescore(S.q.p, x_is_f64, d, S.rows.p, y_is_f64, n, S.cand.as<int64_t>(), (int)n, 1,
                      (int)kk, 0, S.pos.as<int64_t>(), nullptr, st);
    HC(hipGetLastError());
    HC(hipMemcpyAsync(out_pos, S.pos.p, (size_t)kk * 8, hipMemcpyDeviceToHost, st));
    HC(hipStreamSynchronize(st));
#undef HC
    return kk;
}

// ---------------------------------------------------------------------------
// device-resident code array (a TransformedData kept in HBM)
struct tk_codes {
    DevBuf tiled;
    int64_t chunks = 0;
    int M = 0;
    // scratch of the calls on this array
    DevBuf tables, out, hidx, hval, labels, slots_i, slots_l, pair_off, unit_prefix, pair_q,
        pair_f0, chunk_off, mins, cdist, cblock;
    void *pin = nullptr;   // 64 KiB of pinned host memory: tables in, heap out, of the one-launch replay
};

extern "C" tk_codes *tk_codes_upload(const uint64_t *data, int64_t chunks, int M)
{
    if (require_gpu() != TK_OK) return nullptr;
    if (chunks < 0 || M < 2 || M % 2) {
        fail(TK_ERR_ARG, "bad argument: chunks / M");
        return nullptr;
    }
    tk_codes *c = new tk_codes();
    DevBuf stage;
    const int P = M / 2;
    bool ok = c->tiled.ensure((size_t)tk_tiled_uint4s(chunks, P) * 16 + 16) == TK_OK;
    if (ok && chunks > 0) {
        ok = stage.ensure((size_t)chunks * M * 8) == TK_OK &&
             hipMemcpy(stage.p, data, (size_t)chunks * M * 8, hipMemcpyHostToDevice) == hipSuccess;
        if (ok) {
            tk_launch_retile(stage.as<uint4>(), c->tiled.as<uint4>(), chunks, P, 0);
            ok = hipDeviceSynchronize() == hipSuccess;
        }
    }
    stage.release();
    if (!ok) {
        fail(TK_ERR_HIP, "tk_codes_upload: device allocation or copy failed");
        c->tiled.release();
        delete c;
        return nullptr;
    }
    c->chunks = chunks;
    c->M = M;
    return c;
}

extern "C" void tk_codes_free(tk_codes *c)
{
    if (!c) return;
    DevBuf *b[] = {&c->tiled, &c->tables, &c->out, &c->hidx, &c->hval, &c->labels, &c->slots_i,
                   &c->slots_l, &c->pair_off, &c->unit_prefix, &c->pair_q, &c->pair_f0,
                   &c->chunk_off, &c->mins, &c->cdist, &c->cblock};
    for (DevBuf *x : b) x->release();
    if (c->pin) (void)hipHostFree(c->pin);
    delete c;
}

// nq tables against the resident array; tables/out are DEVICE pointers.  out:
// (nq, chunks) uint4.  Four queries per pass (list-major kernel) when nq >= 4.
extern "C" int tk_codes_estimate_dev(tk_codes *c, const void *tables_dev, int64_t nq,
                                     void *out_dev, int signd, int order, void *stream)
{
    ARGCHECK(c, "null codes handle");
    ARGCHECK(nq >= 0 && nq <= 65535, "0 <= nq <= 65535");
    ARGCHECK(order == TK_ORDER_SSE || order == TK_ORDER_AVX, "order");
    if (nq == 0 || c->chunks == 0) return TK_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool units = nq >= 4 && (double)nq / 4 * c->chunks < 2.0e9;
    if (units) {
        TRY(c->pair_off.ensure(8));
        TRY(c->unit_prefix.ensure(tk_unit_prefix_ints(1) * 4));
        TRY(c->pair_q.ensure(((size_t)nq + 4) * 4));
        TRY(c->pair_f0.ensure(((size_t)nq + 4) * 4));
        if (!c->chunk_off.p) {
            int64_t cco[2] = {0, c->chunks};
            TRY(c->chunk_off.ensure(sizeof cco));
            HIPCHECK(hipMemcpy(c->chunk_off.p, cco, sizeof cco, hipMemcpyHostToDevice));
        }
        tk_launch_identity_pairs(nq, (int)c->chunks, c->pair_off.as<int>(),
                                 c->unit_prefix.as<int>(), c->pair_q.as<int>(),
                                 c->pair_f0.as<int>(), st);
        tk_launch_scan_units(c->tiled.as<uint4>(), c->M, (const uint4 *)tables_dev, nq, 1, 1,
                             c->chunk_off.as<int64_t>(), c->pair_off.as<int>(),
                             c->unit_prefix.as<int>(), c->pair_q.as<int>(), c->pair_f0.as<int>(),
                             (uint4 *)out_dev, c->chunks, nullptr, 0, signd, order, 768, st);
    } else {
        tk_launch_scan_flat(c->tiled.as<uint4>(), c->chunks, c->M, (const uint4 *)tables_dev, nq,
                            (uint4 *)out_dev, c->chunks, nullptr, 0, signd, order, st);
    }
    HIPCHECK(hipGetLastError());
    return TK_OK;
}

// estimate_pq on the resident array, host tables/out (what estimate_distances needs)
extern "C" int tk_codes_estimate(tk_codes *c, const uint64_t *tables, int64_t nq, uint64_t *out,
                                 int signd, int order)
{
    ARGCHECK(c, "null codes handle");
    ARGCHECK(nq >= 0 && nq <= 65535, "0 <= nq <= 65535");
    if (nq == 0 || c->chunks == 0) return TK_OK;
    TRY(c->tables.ensure((size_t)nq * c->M * 16));
    TRY(c->out.ensure((size_t)nq * c->chunks * 16));
    HIPCHECK(hipMemcpyAsync(c->tables.p, tables, (size_t)nq * c->M * 16, hipMemcpyHostToDevice, 0));
    TRY(tk_codes_estimate_dev(c, c->tables.p, nq, c->out.p, signd, order, nullptr));
    HIPCHECK(hipMemcpyAsync(out, c->out.p, (size_t)nq * c->chunks * 16, hipMemcpyDeviceToHost, 0));
    HIPCHECK(hipStreamSynchronize(0));
    return TK_OK;
}

// query_pq on the resident array (host heap in/out, optional host labels)
extern "C" int tk_codes_query(tk_codes *c, int64_t n, const uint64_t *tables, int64_t *indices,
                              int32_t *vals, int R, int signd, const int64_t *labels, int order)
{
    ARGCHECK(c, "null codes handle");
    ARGCHECK(R >= 1 && (size_t)R * 12 + 16 <= 64 * 1024, "1 <= R <= 5460");
    ARGCHECK(order == TK_ORDER_SSE || order == TK_ORDER_AVX, "order");
    ARGCHECK(c->chunks < (1ll << 31) / 16, "list too long");
    const int64_t chunks = c->chunks;
    if (chunks == 0) return TK_OK;
    hipStream_t st = 0;
    TRY(c->tables.ensure((size_t)c->M * 16));
    TRY(c->out.ensure((size_t)chunks * 16));
    const int64_t cap_min = (chunks + 15) / 16 * 16;
    TRY(c->mins.ensure((size_t)cap_min));
    // Rows far longer than a FRESH heap of at most 64 entries (top() of one query over a whole data
    // set): one table copy from pinned memory, the scan, ONE launch that replays with the heap in
    // registers and writes it to pinned memory (heap.hip, flat_top_one_kernel).  The head is
    // ~sqrt(R chunks) blocks: about as many again pass its bound.
    bool fresh = !labels && chunks >= 4096 && R <= 64 && (size_t)c->M * 16 <= 32 * 1024;
    for (int i = 0; i < R && fresh; i++) fresh = indices[i] == -1 && vals[i] == (signd ? 127 : 255);
    if (fresh) {
        if (!c->pin) HIPCHECK(hipHostMalloc(&c->pin, 64 * 1024, hipHostMallocDefault));
        TRY(c->cdist.ensure((size_t)chunks * 16));
        TRY(c->cblock.ensure((size_t)chunks * 5));
        unsigned char *pin = (unsigned char *)c->pin;
        int64_t *pidx = (int64_t *)(pin + 32 * 1024);
        int32_t *pval = (int32_t *)(pin + 32 * 1024 + 64 * 8);
        memcpy(pin, tables, (size_t)c->M * 16);
        HIPCHECK(hipMemcpyAsync(c->tables.p, pin, (size_t)c->M * 16, hipMemcpyHostToDevice, st));
        tk_launch_scan_flat(c->tiled.as<uint4>(), chunks, c->M, c->tables.as<uint4>(), 1,
                            c->out.as<uint4>(), chunks, c->mins.as<uint8_t>(), cap_min, signd, order, st);
        int64_t h = ((int64_t)std::sqrt((double)R * (double)chunks) + 63) / 64 * 64;
        h = h < 64 ? 64 : (h > chunks / 16 * 16 ? chunks / 16 * 16 : h);
        const int64_t nn = n < 0 ? 0 : n;
        tk_launch_flat_top_one(c->out.as<uint4>(), c->mins.as<uint8_t>(), (int)chunks, (int)h, nn, R, signd,
                               c->cdist.as<uint4>(), c->cblock.as<int>(), pidx, pval, st);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipStreamSynchronize(st));
        memcpy(indices, pidx, (size_t)R * 8);
        memcpy(vals, pval, (size_t)R * 4);
#ifdef TK_FLAT_CLOCK
        {
            const int64_t *g = pidx + 1024;
            fprintf(stderr, "flat clock: head %lld cyc %.1f us | compact %lld cyc %.1f us | tail %lld cyc %.1f us | kept %lld h %lld\n",
                    (long long)(g[2] - g[0]), (g[3] - g[1]) / 100.0, (long long)(g[4] - g[2]), (g[5] - g[3]) / 100.0,
                    (long long)(g[6] - g[4]), (g[7] - g[5]) / 100.0, (long long)g[8], (long long)h);
        }
#endif
        return TK_OK;
    }
    TRY(c->hidx.ensure((size_t)R * 8));
    TRY(c->hval.ensure((size_t)R * 4));
    TRY(c->slots_i.ensure(3 * sizeof(int)));
    TRY(c->slots_l.ensure(sizeof(int64_t)));
    int64_t nlab = n < 16 * chunks ? n : 16 * chunks;
    if (nlab < 0) nlab = 0;
    if (labels && nlab > 0) {
        TRY(c->labels.ensure((size_t)nlab * 8));
        HIPCHECK(hipMemcpyAsync(c->labels.p, labels, (size_t)nlab * 8, hipMemcpyHostToDevice, st));
    }
    HIPCHECK(hipMemcpyAsync(c->tables.p, tables, (size_t)c->M * 16, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(c->hidx.p, indices, (size_t)R * 8, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(c->hval.p, vals, (size_t)R * 4, hipMemcpyHostToDevice, st));
    int64_t nclamp = n > (int64_t)0x7fffffff ? 0x7fffffff : (n < 0 ? 0 : n);
    int si[3] = {0, (int)chunks, (int)nclamp};
    int64_t sl[1] = {labels ? 0 : -1};
    HIPCHECK(hipMemcpyAsync(c->slots_i.p, si, sizeof si, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(c->slots_l.p, sl, sizeof sl, hipMemcpyHostToDevice, st));
    // the scan also writes each block's minimum: the replay walks 1024 blocks per step on them
    tk_launch_scan_flat(c->tiled.as<uint4>(), chunks, c->M, c->tables.as<uint4>(), 1,
                        c->out.as<uint4>(), chunks, c->mins.as<uint8_t>(), cap_min, signd, order, st);
    tk_launch_heap_replay(c->out.as<uint4>(), chunks, 1, c->slots_i.as<int>(),
                          c->slots_i.as<int>() + 2, c->slots_l.as<int64_t>(), 1,
                          c->labels.as<int64_t>(), c->hidx.as<int64_t>(), c->hval.as<int32_t>(), R,
                          signd, 1, nullptr, st, c->mins.as<uint8_t>(), cap_min);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(indices, c->hidx.p, (size_t)R * 8, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipMemcpyAsync(vals, c->hval.p, (size_t)R * 4, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    return TK_OK;
}

// ---------------------------------------------------------------------------
// offline build path (build.hip): host buffers in, host buffers out, rows in slabs
extern "C" int tk_encode_pq(const float *centers, int dq, int dpb, const void *data,
                            int data_is_f64, int64_t n, uint8_t *labels)
{
    TRY(require_gpu());
    ARGCHECK(centers && data && labels, "null buffer");
    ARGCHECK(dpb >= 1 && dpb <= 32 && dq >= dpb && dq % dpb == 0, "dq/dpb");
    ARGCHECK(n >= 0, "n");
    const int M = dq / dpb;
    ARGCHECK(16 % dpb == 0, "dims_per_block must divide 16 for the device encoder");
    ARGCHECK((size_t)16 * M * (dpb + 1) * 4 + 4 * 64 * 17 * 8 + 4 * 64 * (size_t)M <= 160 * 1024,
             "codebook larger than the LDS budget");
    const size_t esz = data_is_f64 ? 8 : 4;
    const int64_t slab = 1 << 20;
    DevBuf dc, dd, dl;
    TRY(dc.ensure((size_t)16 * dq * 4));
    HIPCHECK(hipMemcpy(dc.p, centers, (size_t)16 * dq * 4, hipMemcpyHostToDevice));
    int rc = TK_OK;
    for (int64_t o = 0; o < n && rc == TK_OK; o += slab) {
        const int64_t m = n - o < slab ? n - o : slab;
        if ((rc = dd.ensure((size_t)m * dq * esz)) != TK_OK) break;
        if ((rc = dl.ensure((size_t)m * M)) != TK_OK) break;
        hipError_t e = hipMemcpy(dd.p, (const char *)data + (size_t)o * dq * esz,
                                 (size_t)m * dq * esz, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            if (tk_launch_encode_pq(dc.as<float>(), dq, dpb, dd.p, data_is_f64, m, dl.as<uint8_t>(), 0))
                rc = fail(TK_ERR_HIP, "encode_pq_kernel: LDS budget / attribute");
            e = hipGetLastError();
        }
        if (e == hipSuccess)
            e = hipMemcpy(labels + (size_t)o * M, dl.p, (size_t)m * M, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(TK_ERR_HIP, hipGetErrorString(e));
    }
    dc.release(); dd.release(); dl.release();
    return rc;
}

extern "C" int tk_assign_lists(const float *X, int64_t n, int d, int normalise, const void *Y,
                               int y_is_f64, const void *ynorm2, int64_t L, int k,
                               int64_t *nearest)
{
    TRY(require_gpu());
    ARGCHECK(X && Y && ynorm2 && nearest, "null buffer");
    ARGCHECK(n >= 0 && d >= 1 && L >= 1 && L < (1ll << 31), "sizes");
    ARGCHECK(k >= 1 && k <= 9 && k <= L, "k must be 1 .. 9 (the range examples/bench.py:108-111 sweeps)");
    ARGCHECK(d <= 384, "d > 384: OpenBLAS splits K there and the FMA chain no longer holds");
    ARGCHECK(!normalise || d <= 128, "row normalisation on the device needs d <= 128");
    const size_t ysz = y_is_f64 ? 8 : 4;
    // Y (L, d) -> Yt (d, L): a wave reads consecutive centres
    std::vector<char> yt((size_t)L * d * ysz);
    for (int64_t j = 0; j < L; j++)
        for (int t = 0; t < d; t++)
            memcpy(&yt[((size_t)t * L + j) * ysz], (const char *)Y + ((size_t)j * d + t) * ysz, ysz);
    const int64_t slab = 1 << 20;
    DevBuf dy, dn, dx, dxn, dout;
    TRY(dy.ensure(yt.size()));
    TRY(dn.ensure((size_t)L * ysz));
    HIPCHECK(hipMemcpy(dy.p, yt.data(), yt.size(), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dn.p, ynorm2, (size_t)L * ysz, hipMemcpyHostToDevice));
    int rc = TK_OK;
    for (int64_t o = 0; o < n && rc == TK_OK; o += slab) {
        const int64_t m = n - o < slab ? n - o : slab;
        if ((rc = dx.ensure((size_t)m * d * 4)) != TK_OK) break;
        if ((rc = dout.ensure((size_t)m * k * 8)) != TK_OK) break;
        if (normalise && (rc = dxn.ensure((size_t)m * d * 4)) != TK_OK) break;
        hipError_t e = hipMemcpy(dx.p, X + (size_t)o * d, (size_t)m * d * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            const float *xs = dx.as<float>();
            if (normalise) {
                tk_launch_normalise_rows(dx.as<float>(), m, d, dxn.as<float>(), 0);
                xs = dxn.as<float>();
            }
            tk_launch_assign(xs, m, d, dy.p, dn.p, y_is_f64, (int)L, k, dout.as<int64_t>(), 0);
            e = hipGetLastError();
        }
        if (e == hipSuccess)
            e = hipMemcpy(nearest + (size_t)o * k, dout.p, (size_t)m * k * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(TK_ERR_HIP, hipGetErrorString(e));
    }
    dy.release(); dn.release(); dx.release(); dxn.release(); dout.release();
    return rc;
}

// ---------------------------------------------------------------------------
// device-resident index
// buffers of ONE batch in flight
struct Work {
    DevBuf tables, shift, scale, cdist, cheap_idx, cheap_val, probes, slot_prefix, slot_chunk0,
        slot_n, slot_loff, dist, heap_idx, heap_val, repeat_flag, cmins, mins, u_count, u_cursor,
        u_pair_off, u_unit_prefix, u_pair_q, u_pair_f0, c_pair_off, c_unit_prefix, c_pair_q,
        c_pair_f0, spos, rpos, smins, pair_cnt, pair_off, scan_tmp, tally, usage, pos_lens, pos_off,
        qlim, slot_exact, p_count, p_cursor, p_pair_off, p_unit_prefix, p_pair_q, p_pair_f0, flag_list, p_unit_desc,
        plain0, h_count, h_cursor, h_pair_off, h_unit_prefix, h_pair_q, h_pair_f0,   // plain_scan.hip
        plain_q;                                                                       // two-phase sharded scan
    // list-sharded batch: what tk_index_shard_scan_dev left for the filtered exchange
    const int64_t *shard_probes = nullptr;
    int64_t shard_nq = 0, shard_capacity = 0;
    bool shard_first = false;       // tk_index_shard_scan_first_dev ran: _rest_dev is owed
    // pipelined mode (depth > 1): hand-offs between the caller's stream and a latency stream
    hipEvent_t tables_done = nullptr, coarse_scanned = nullptr, front_done = nullptr,
               scanned = nullptr, done = nullptr;
    bool busy = false;                                 // `done` has been recorded
    uint64_t td_seq = 0, fd_seq = 0;                   // order in which tables_done / front_done were last recorded
    // plain_scan.hip: how many queries of the workspace's last plain batch were flagged — a
    // page-locked word the device writes and an event behind it, polled (never waited for) when a
    // later call looks at the workspace
    int *flag_host = nullptr;
    hipEvent_t plain_ev = nullptr;
    bool plain_pending = false;
    bool last_plain = false;    // the last batch that used this workspace went the plain way
    int64_t plain_nq = 0;
    void release()
    {
        if (flag_host) (void)hipHostFree(flag_host);
        flag_host = nullptr;
        if (plain_ev) (void)hipEventDestroy(plain_ev);
        plain_ev = nullptr;
        plain_pending = false;
        DevBuf *b[] = {&tables, &shift, &scale, &cdist, &cheap_idx, &cheap_val, &probes,
                       &slot_prefix, &slot_chunk0, &slot_n, &slot_loff, &dist, &heap_idx, &heap_val,
                       &repeat_flag, &cmins, &mins, &u_count, &u_cursor, &u_pair_off, &u_unit_prefix,
                       &u_pair_q, &u_pair_f0, &c_pair_off, &c_unit_prefix, &c_pair_q, &c_pair_f0,
                       &spos, &rpos, &smins, &pair_cnt, &pair_off, &scan_tmp, &tally, &usage, &pos_lens, &pos_off,
                       &qlim, &slot_exact, &p_count, &p_cursor, &p_pair_off, &p_unit_prefix, &p_pair_q, &p_pair_f0, &flag_list, &p_unit_desc,
                       &plain0, &h_count, &h_cursor, &h_pair_off, &h_unit_prefix, &h_pair_q, &h_pair_f0,
                       &plain_q};
        for (DevBuf *x : b) x->release();
        hipEvent_t *evs[] = {&tables_done, &coarse_scanned, &front_done, &scanned, &done};
        for (hipEvent_t *e : evs) {
            if (*e) (void)hipEventDestroy(*e);
            *e = nullptr;
        }
        busy = false;
    }
};

#define IXLOCK(ix_)                                               \
    std::unique_lock<std::recursive_mutex> ixlock_;               \
    if (ix_) ixlock_ = std::unique_lock<std::recursive_mutex>((ix_)->mu)

struct Pending;
static int flush_pending(struct tk_index *ix);
static int batch_epilogue(const struct Pending &b, hipStream_t st);

struct tk_index {
    // one caller at a time: every entry point takes this lock (the reference's kernels are
    // nogil and re-entrant on distinct buffers; calls on one handle from several threads are
    // serialised here instead of corrupting the pipeline state)
    std::recursive_mutex mu;
    // FastPQ
    DevBuf pq_centers;
    int dq = 0, dpb = 0, M = 0, f_order = 0, order = TK_ORDER_AVX;
    double sqrt_nb = 0;
    // coarse
    DevBuf active_centers, center_codes;
    int64_t n_lists = 0, center_chunks = 0;
    int d = 0;
    // lists
    DevBuf list_chunk_off, list_n, ids_off, ids, codes, ids32;
    bool have_ids32 = false;   // every label fits int32: the lane kernel can run the duplicate test
    int64_t total_chunks = 0, total_ids = 0;
    int max_list_chunks = 0;
    bool ids_unique = false;   // no label occurs twice => the lane-per-query replay is exact
    int heap_mode = 0;         // 0 auto (lanes, else packed wave), 1 general wave, 2 packed wave
    bool have_pq = false, have_centers = false, have_lists = false, have_data = false;
    // list-sharded index: this rank stores the codes of the lists it owns;
    // list_chunk_off stays the GLOBAL layout (every rank derives the same distance rows),
    // local_chunk_off addresses this rank's code storage (lists of other ranks: empty)
    DevBuf owner, local_chunk_off;
    DevBuf rot_t;            // fast mode: R transposed (d_pad, dq) float64, or empty
    DevBuf br_ynorm, br_vals, br_tau, br_cand, br_count, br_out, br_q, br_sample;   // tk_index_knn_brute
    int rot_d_pad = 0;
    int rank = 0, world = 1;
    bool sharded = false;
    // vectors
    DevBuf data;
    int64_t N = 0;
    int data_is_f64 = 0;
    // index-static descriptors of the coarse stage, staging buffers of the host API
    DevBuf cslots_i, cslots_l, c_chunk_off, q, qpq, stage;
    int scan_mode = 0;         // 0 auto, 1 query-major kernel, 2 list-major (units) kernel
    // tk_index_set_option
    int opt_scan_form = 0;             // exact list-major kernel: 0 per-lane table-row loads, 1 / 2 rows staged in LDS
    int opt_rescore_form = 2;          // rescoring: 2 / 1 rows staged through LDS in tiles of 32 / 64, 0 lane per row
    int opt_plain_limit = 0x7fffffff;  // a cap on every query's table limit (tests: provokes the re-scan path)
    bool flat_plain_ok = true;         // tk_index_top_centers: the plain path has not failed on this index
    int plain_state = 0;       // PLAIN_PROBE .. PLAIN_OFF (see plain_poll)
    int plain_skip = 0;        // OFF: batches left before the next probe
    int plain_backoff = 256;   // OFF: length of the next pause (doubled by a failed probe, up to 4096)
    int plain_wait = 0;        // WAIT: batches seen while no verdict is pending (a probe that was abandoned)
    bool capturing = false;    // the current call is being captured into a hipGraph: no event queries
    int plain_mode = 0;        // 0 auto: probed lists behind the first ones as plain sums on the matrix
                               // cores where the lemma of plain_scan.hip allows AND few queries need the
                               // re-scan (plain_poll); 1: exact kernel only; 2: plain always
    bool host_out_kernel = false;   // a batch's pinned host copy of the ids is written by a kernel
    // per-batch workspaces: `depth` batches may be in flight (tk_index_set_pipeline),
    // each on its own internal stream
    std::vector<Work> works;
    int depth = 1;
    uint64_t calls = 0;
    std::vector<hipStream_t> lat_streams;    // `depth` of them (pipelined mode)
    hipEvent_t ev_in = nullptr;              // caller's stream -> a batch's stream
    std::vector<struct Pending *> pending;   // calls whose list scan is still to be enqueued (<= 2)
    uint64_t ev_seq = 0;                     // counts the records of tables_done / front_done (pipeline_step's merged wait)
    int coalesce = 1;                        // 2: two consecutive calls run as ONE batch (tk_index_set_coalesce)
    struct Pending *held = nullptr;          // ... the first of such a pair, its inputs copied, waiting for the second
    int held_n_probes = 0, held_pass_1 = 0, held_f64 = 0;
    int64_t held_rows = 0;                   // rows its staging buffers hold
    hipStream_t held_stt = nullptr, held_caller = nullptr;
    hipStream_t front_stream = nullptr;      // coarse replays + descriptors of all batches, in order
    // profiling: one set of 8 events per recorded batch, read back on demand
    int profiling = 0;
    uint64_t prof_seen = 0;
    std::vector<hipEvent_t> evs;   // 8 per set
    std::vector<hipStream_t> ev_streams;
    std::vector<char> ev_plain;    // the set's batch ran the plain kernel (events 8, 9 recorded)
    size_t ev_used = 0;            // sets recorded since the last read
    int last_S = 0, last_R = 0, last_work = 0;
    int64_t last_nq = 0;
};

extern "C" tk_index *tk_index_create(void)
{
    if (require_gpu() != TK_OK) return nullptr;
    tk_index *ix = new tk_index();
    ix->works.resize(1);
    // A/B: TINYKNN_PLAIN_SCAN=2 starts every index in mode 2 (plain always, repeating labels too)
    const char *e = getenv("TINYKNN_PLAIN_SCAN");
    if (e && e[0] == '2') ix->plain_mode = 2;
    return ix;
}

extern "C" void tk_index_destroy(tk_index *ix)
{
    if (!ix) return;
    (void)flush_pending(ix);
    (void)hipDeviceSynchronize();
    DevBuf *bufs[] = {&ix->pq_centers, &ix->active_centers, &ix->center_codes, &ix->list_chunk_off,
                      &ix->list_n, &ix->ids_off, &ix->ids, &ix->codes, &ix->ids32, &ix->data, &ix->cslots_i,
                      &ix->cslots_l, &ix->c_chunk_off, &ix->q, &ix->qpq, &ix->stage, &ix->owner,
                      &ix->local_chunk_off, &ix->rot_t, &ix->br_ynorm, &ix->br_vals, &ix->br_tau,
                      &ix->br_cand, &ix->br_count, &ix->br_out, &ix->br_q, &ix->br_sample};
    for (DevBuf *b : bufs) b->release();
    for (Work &w : ix->works) w.release();
    for (hipStream_t st : ix->lat_streams) (void)hipStreamDestroy(st);
    if (ix->front_stream) (void)hipStreamDestroy(ix->front_stream);
    if (ix->ev_in) (void)hipEventDestroy(ix->ev_in);
    for (auto &e : ix->evs) (void)hipEventDestroy(e);
    delete ix;
}

extern "C" int tk_index_set_pq(tk_index *ix, const float *centers, int dq, int dpb, int f_order,
                               double sqrt_n_blocks, int order)
{
    IXLOCK(ix);
    ARGCHECK(ix, "null index");
    ARGCHECK(dpb >= 1 && dpb <= 32 && dq % dpb == 0, "dq/dpb");
    ARGCHECK((dq / dpb) % 2 == 0, "number of blocks must be even");
    ARGCHECK(dq / dpb <= 512, "at most 512 blocks");
    ARGCHECK(order == TK_ORDER_SSE || order == TK_ORDER_AVX, "order");
    TRY(ix->pq_centers.ensure((size_t)16 * dq * 4));
    HIPCHECK(hipMemcpy(ix->pq_centers.p, centers, (size_t)16 * dq * 4, hipMemcpyHostToDevice));
    ix->dq = dq; ix->dpb = dpb; ix->M = dq / dpb; ix->f_order = f_order;
    ix->sqrt_nb = sqrt_n_blocks; ix->order = order;
    ix->have_pq = true;
    return TK_OK;
}

static int upload_tiled(DevBuf &dst, DevBuf &stage, const uint64_t *codes, int64_t chunks, int M)
{
    const int P = M / 2;
    size_t tiled_bytes = (size_t)tk_tiled_uint4s(chunks, P) * 16;
    TRY(dst.ensure(tiled_bytes > 0 ? tiled_bytes : 16));
    if (chunks == 0) return TK_OK;
    size_t ref_bytes = (size_t)chunks * M * 8;
    TRY(stage.ensure(ref_bytes));
    HIPCHECK(hipMemcpy(stage.p, codes, ref_bytes, hipMemcpyHostToDevice));
    tk_launch_retile(stage.as<uint4>(), dst.as<uint4>(), chunks, P, 0);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipDeviceSynchronize());
    return TK_OK;
}

extern "C" int tk_index_set_centers(tk_index *ix, const float *active_centers, int64_t n_lists,
                                    int d, const uint64_t *center_codes, int64_t center_chunks)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->have_pq, "set_pq first");
    ARGCHECK(n_lists >= 1 && d >= 1, "sizes");
    ARGCHECK(center_chunks == (n_lists + 15) / 16, "center_chunks must be ceil(n_lists/16)");
    TRY(ix->active_centers.ensure((size_t)n_lists * d * 4));
    HIPCHECK(hipMemcpy(ix->active_centers.p, active_centers, (size_t)n_lists * d * 4,
                       hipMemcpyHostToDevice));
    TRY(upload_tiled(ix->center_codes, ix->stage, center_codes, center_chunks, ix->M));
    ix->n_lists = n_lists; ix->d = d; ix->center_chunks = center_chunks;
    int64_t cco[2] = {0, center_chunks};
    TRY(ix->c_chunk_off.ensure(sizeof cco));
    HIPCHECK(hipMemcpy(ix->c_chunk_off.p, cco, sizeof cco, hipMemcpyHostToDevice));
    int ci[3] = {0, (int)center_chunks, (int)n_lists};
    int64_t cl[1] = {-1};
    TRY(ix->cslots_i.ensure(sizeof ci));
    TRY(ix->cslots_l.ensure(sizeof cl));
    HIPCHECK(hipMemcpy(ix->cslots_i.p, ci, sizeof ci, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ix->cslots_l.p, cl, sizeof cl, hipMemcpyHostToDevice));
    ix->have_centers = true;
    return TK_OK;
}

// `owner` == NULL: the whole index; otherwise `codes` holds only the lists with
// owner[l] == rank, concatenated in list order
static int set_lists_impl(tk_index *ix, const int64_t *list_sizes, const uint64_t *codes,
                          const int64_t *ids, const int32_t *owner, int rank, int world)
{
    ARGCHECK(ix && ix->have_centers, "set_centers first");
    const int64_t L = ix->n_lists;
    std::vector<int64_t> coff(L + 1, 0), ioff(L + 1, 0), loff(L + 1, 0);
    int64_t maxc = 0;
    for (int64_t i = 0; i < L; i++) {
        ARGCHECK(list_sizes[i] >= 0, "negative list size");
        ARGCHECK(!owner || (owner[i] >= 0 && owner[i] < world), "owner out of range");
        int64_t c = (list_sizes[i] + 15) / 16;
        coff[i + 1] = coff[i] + c;
        loff[i + 1] = loff[i] + ((!owner || owner[i] == rank) ? c : 0);
        ioff[i + 1] = ioff[i] + list_sizes[i];
        if (c > maxc) maxc = c;
    }
    ARGCHECK(maxc < (1ll << 26), "list too long");
    // are the labels pairwise distinct?  (IVF.build(n_probes=1): every point in one list)
    {
        bool uniq = true;
        const int64_t T = ioff[L];
        int64_t mn = 0, mx = -1;
        for (int64_t i = 0; i < T; i++) {
            if (i == 0 || ids[i] < mn) mn = ids[i];
            if (i == 0 || ids[i] > mx) mx = ids[i];
        }
        if (T > 0 && mn >= 0 && mx < 64 * T + 1024) {
            std::vector<uint64_t> seen((size_t)(mx / 64 + 1), 0);
            for (int64_t i = 0; i < T && uniq; i++) {
                uint64_t bit = 1ull << (ids[i] & 63);
                if (seen[(size_t)(ids[i] >> 6)] & bit) uniq = false;
                seen[(size_t)(ids[i] >> 6)] |= bit;
            }
        } else if (T > 0) {
            std::vector<int64_t> tmp(ids, ids + T);
            std::sort(tmp.begin(), tmp.end());
            for (int64_t i = 1; i < T && uniq; i++) uniq = tmp[i] != tmp[i - 1];
            if (mn < 0) uniq = false;  // a label -1 would match the heap's sentinel
        }
        ix->ids_unique = uniq;
    }
    TRY(upload_tiled(ix->codes, ix->stage, codes, loff[L], ix->M));
    ix->sharded = owner != nullptr;
    ix->rank = owner ? rank : 0;
    ix->world = owner ? world : 1;
    if (owner) {
        TRY(ix->owner.ensure((size_t)L * 4));
        TRY(ix->local_chunk_off.ensure((size_t)(L + 1) * 8));
        HIPCHECK(hipMemcpy(ix->owner.p, owner, (size_t)L * 4, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(ix->local_chunk_off.p, loff.data(), (size_t)(L + 1) * 8,
                           hipMemcpyHostToDevice));
    }
    TRY(ix->list_chunk_off.ensure((size_t)(L + 1) * 8));
    TRY(ix->ids_off.ensure((size_t)(L + 1) * 8));
    TRY(ix->list_n.ensure((size_t)L * 8));
    TRY(ix->ids.ensure((size_t)(ioff[L] > 0 ? ioff[L] : 1) * 8));
    HIPCHECK(hipMemcpy(ix->list_chunk_off.p, coff.data(), (size_t)(L + 1) * 8, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ix->ids_off.p, ioff.data(), (size_t)(L + 1) * 8, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ix->list_n.p, list_sizes, (size_t)L * 8, hipMemcpyHostToDevice));
    if (ioff[L] > 0)
        HIPCHECK(hipMemcpy(ix->ids.p, ids, (size_t)ioff[L] * 8, hipMemcpyHostToDevice));
    {   // int32 copy of the labels for the lane kernel's duplicate test
        bool fits = true;
        for (int64_t i = 0; i < ioff[L] && fits; i++) fits = ids[i] >= 0 && ids[i] < 0x7fffffff;
        ix->have_ids32 = false;
        if (fits && ioff[L] > 0 && !ix->ids_unique) {
            std::vector<int32_t> tmp((size_t)ioff[L]);
            for (int64_t i = 0; i < ioff[L]; i++) tmp[(size_t)i] = (int32_t)ids[i];
            TRY(ix->ids32.ensure(tmp.size() * 4));
            HIPCHECK(hipMemcpy(ix->ids32.p, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice));
            ix->have_ids32 = true;
        }
    }
    ix->total_chunks = coff[L];
    ix->total_ids = ioff[L];
    ix->max_list_chunks = (int)maxc;
    ix->have_lists = true;
    return TK_OK;
}

extern "C" int tk_index_set_lists(tk_index *ix, const int64_t *list_sizes, const uint64_t *codes,
                                  const int64_t *ids)
{
    IXLOCK(ix);
    return set_lists_impl(ix, list_sizes, codes, ids, nullptr, 0, 1);
}

extern "C" int tk_index_set_lists_shard(tk_index *ix, const int64_t *list_sizes,
                                        const int32_t *owner, int rank, int world,
                                        const uint64_t *codes_owned, const int64_t *ids)
{
    IXLOCK(ix);
    ARGCHECK(owner, "owner");
    ARGCHECK(world >= 1 && rank >= 0 && rank < world, "rank/world");
    return set_lists_impl(ix, list_sizes, codes_owned, ids, owner, rank, world);
}

extern "C" int tk_index_set_data(tk_index *ix, const void *data, int data_is_f64, int64_t N,
                                 int d)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->have_centers, "set_centers first");
    ARGCHECK(d == ix->d, "data dimension differs from the centres'");
    ARGCHECK(N >= 1, "N");
    const size_t esz = data_is_f64 ? 8 : 4;
    TRY(ix->data.ensure((size_t)N * d * esz));
    HIPCHECK(hipMemcpy(ix->data.p, data, (size_t)N * d * esz, hipMemcpyHostToDevice));
    ix->N = N;
    ix->data_is_f64 = data_is_f64;
    ix->have_data = true;
    return TK_OK;
}

struct Plan {
    int kc, rescore, R, S;
    int64_t cap;       // uint4 per query in the distance buffer
    int64_t cap_min;   // bytes per query in the block-minimum buffer (multiple of 16)
    int64_t ccap_min;  // same for the coarse stage
};

static int make_plan(const tk_index *ix, int k, int n_probes, int pass_1, Plan &p)
{
    ARGCHECK(ix && ix->have_pq && ix->have_centers && ix->have_lists && ix->have_data,
             "index not fully populated (pq, centers, lists, data)");
    ARGCHECK(k >= 1 && n_probes >= 1, "k and n_probes must be >= 1");
    int64_t kc = n_probes < ix->n_lists ? n_probes : ix->n_lists;              // fast_pq.py:291
    int64_t rescore = 2 * kc + 10 < ix->n_lists ? 2 * kc + 10 : ix->n_lists;   // :293-294
    int64_t R = pass_1 > 0 ? pass_1 : (int64_t)(n_probes + 1) * k + 1;         // ivf.py:135-136
    ARGCHECK(R * 12 + 16 <= 64 * 1024 && rescore * 12 + 16 <= 64 * 1024,
             "heap larger than 64 KiB of LDS (pass_1 <= 5460)");
    ARGCHECK(R * (ix->data_is_f64 ? 16 : 12) + (int64_t)ix->d * (ix->data_is_f64 ? 8 : 4) + 16 <= 64 * 1024,
             "rescoring tile larger than 64 KiB of LDS");
    p.kc = (int)kc; p.rescore = (int)rescore; p.R = (int)R; p.S = (int)kc;
    p.cap = (int64_t)kc * ix->max_list_chunks;
    if (p.cap < 1) p.cap = 1;
    ARGCHECK(p.cap < (1ll << 31), "probed chunk range overflows int32");
    p.cap_min = (p.cap + 15) / 16 * 16;
    p.ccap_min = (ix->center_chunks + 15) / 16 * 16;
    return TK_OK;
}

// Plain sums on the matrix cores for the probed lists behind the first ones (plain_scan.hip):
// signed tables (all of IVF.query), at most 26 block pairs, a replay that starts from fresh heaps
// on packed entries (the lane kernels check the lemma's condition per query and flag the queries
// to re-scan), an unsharded index.  TINYKNN_PLAIN_SCAN=0 / tk_index_set_plain_scan(ix, 1): off.
static bool plain_env_on()
{
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("TINYKNN_PLAIN_SCAN");
        on = !(e && e[0] == '0');
    }
    return on != 0;
}
static bool plain_possible(const tk_index *ix, const Plan &p)
{
    if (ix->plain_mode == 1 || !plain_env_on() || ix->sharded || p.S < 2 || !tk_plain_fits(ix->M)) return false;
    if (ix->heap_mode != 0 || ix->scan_mode == 1 || p.cap * 16 > 0xffffff) return false;
    if (ix->ids_unique) return p.R <= TK_LANES_MAX_R;
    // repeating labels (build n_probes >= 2): such a batch is bound by the replay with the duplicate
    // test (1.56 ms alone for 10 000 queries, two in flight), not by the scan, and the plain kernel
    // beside it only stretches that replay — same box, glove-like build_probes = 2: 1.33 ms per batch
    // on the exact kernel, 1.55 ms with the plain path (profiles/r03/ab_build_probes2.txt).  Only on
    // request (mode 2), which is how the tests reach this branch.
    return ix->plain_mode == 2 && ix->have_ids32 && tk_lanes_dedupe_fits(p.R, p.S) && ix->total_ids < (1ll << 31);
}

// The plain path is exact for every query, but a FLAGGED query is scanned twice and replayed
// twice: on data without structure (iid vectors: the first rows a query scans are no nearer than
// the rest, the bound stays above the table's limit) 42 % of the queries were flagged and the batch
// took 3 x as long as on the exact kernel alone.  So the path proves itself first, and a caller
// that enqueues far ahead of the device (the pipelined mode never waits) cannot pile up plain
// batches before the first verdict is in:
//     PROBE    the next batch goes the plain way, then ->
//     WAIT     exact kernel only until that batch's flagged count is known (read — never waited
//              for — from a page-locked word behind an event): <= 1 % flagged -> ON, else -> OFF
//     ON       plain for every batch; any completed batch above 1 % -> OFF
//     OFF      exact kernel only for `plain_backoff` batches (256, doubling up to 4096 on every
//              failed probe in a row), then -> PROBE
// tk_index_set_plain_scan(ix, 2): always plain (A/B, the tests' forced re-scans).  Results never
// depend on any of this.
enum { PLAIN_PROBE = 0, PLAIN_WAIT = 1, PLAIN_ON = 2, PLAIN_OFF = 3 };
static bool plain_adaptive(const tk_index *ix)
{
    return ix->plain_mode == 0 && ix->opt_plain_limit == 0x7fffffff;
}
static void plain_poll(tk_index *ix)
{
    if (ix->capturing) return;      // (no event queries inside a stream capture)
    for (Work &w : ix->works)
        if (w.plain_pending && w.plain_ev && hipEventQuery(w.plain_ev) == hipSuccess) {
            w.plain_pending = false;
            const bool bad = w.flag_host && w.plain_nq > 0 && (double)*w.flag_host > 0.01 * (double)w.plain_nq;
            if (bad) {
                if (ix->plain_state == PLAIN_WAIT)          // a failed probe: wait longer before the next
                    ix->plain_backoff = ix->plain_backoff < 4096 ? ix->plain_backoff * 2 : 4096;
                if (ix->plain_state != PLAIN_OFF) ix->plain_skip = ix->plain_backoff;
                ix->plain_state = PLAIN_OFF;
            } else if (ix->plain_state == PLAIN_WAIT) {
                ix->plain_state = PLAIN_ON;
                ix->plain_backoff = 256;
            }
        }
    (void)hipGetLastError();       // (hipErrorNotReady of a query is not an error)
}
// plain for THIS batch?
static bool plain_now(tk_index *ix, const Plan &p)
{
    if (!plain_possible(ix, p)) return false;
    if (!plain_adaptive(ix)) return true;
    if (ix->capturing) return ix->plain_state == PLAIN_ON;     // a captured graph keeps what it was captured with
    plain_poll(ix);
    switch (ix->plain_state) {
    case PLAIN_ON: return true;
    case PLAIN_PROBE:
        ix->plain_state = PLAIN_WAIT;
        ix->plain_wait = 0;
        return true;
    case PLAIN_WAIT: {
        // the probe's verdict is recorded behind its replay, up to three calls after this point; a
        // probe batch that was abandoned before that (a failed reserve, a HIP error) never reports:
        // after 16 batches with no verdict pending anywhere, probe again
        bool pending = !ix->pending.empty() || ix->held != nullptr;      // (a batch not yet replayed may be the probe)
        for (const Work &w : ix->works) pending |= w.plain_pending;
        if (!pending && ++ix->plain_wait >= 16) ix->plain_state = PLAIN_PROBE;
        return false;
    }
    default:
        if (--ix->plain_skip <= 0) ix->plain_state = PLAIN_PROBE;
        return false;
    }
}

// Chunk pairs per unit of the plain kernel (one wave per unit): a multiple of 4 — the wave's output
// tile leaves every fourth chunk pair — of at least 12 (a unit's 26 table-row loads want amortising),
// more where a batch holds far more than ~6 units per resident wave (long lists: 100M x 128).
static int plain_k(const tk_index *ix, int64_t nq, const Plan &p)
{
    const double iters = (double)nq * p.S / 32.0 * ((double)ix->total_chunks / (double)ix->n_lists) / 2.0;
    int k = (int)(iters / (2048.0 * 6.0));
    k = (k + 3) & ~3;
    return k < 12 ? 12 : (k > 64 ? 64 : k);
}

static size_t plain_desc_bytes(const tk_index *ix, int64_t nq, const Plan &p)
{
    return (size_t)tk_plain_units_bound(nq * p.S, ix->n_lists, ix->total_chunks, ix->max_list_chunks,
                                        plain_k(ix, nq, p)) * 16;
}

static int reserve(tk_index *ix, Work &w, int64_t nq, int k, const Plan &p)
{
    const int M = ix->M;
    TRY(w.tables.ensure((size_t)nq * M * 16));
    TRY(w.shift.ensure((size_t)nq * 8));
    TRY(w.scale.ensure((size_t)nq * 8));
    TRY(w.cdist.ensure((size_t)nq * ix->center_chunks * 16));
    TRY(w.cheap_idx.ensure((size_t)nq * p.rescore * 8));
    TRY(w.cheap_val.ensure((size_t)nq * p.rescore * 4));
    TRY(w.probes.ensure((size_t)nq * p.kc * 8));
    TRY(w.slot_prefix.ensure((size_t)nq * (p.S + 1) * 4));
    TRY(w.slot_chunk0.ensure((size_t)nq * p.S * 8));
    TRY(w.slot_n.ensure((size_t)nq * p.S * 4));
    TRY(w.slot_loff.ensure((size_t)nq * p.S * 8));
    TRY(w.dist.ensure((size_t)nq * p.cap * 16));
    TRY(w.heap_idx.ensure((size_t)nq * p.R * 8));
    TRY(w.heap_val.ensure((size_t)nq * p.R * 4));
    (void)k;
    TRY(w.repeat_flag.ensure((size_t)nq));
    TRY(w.mins.ensure((size_t)nq * p.cap_min));
    TRY(w.cmins.ensure((size_t)nq * p.ccap_min));
    const size_t L = (size_t)ix->n_lists;
    {
        const void *before = w.u_count.p;
        TRY(w.u_count.ensure(L * 4));
        // the per-list pair counters are zero between batches (the scan kernel re-zeroes
        // them after reading); a fresh buffer must start that way
        if (w.u_count.p != before) HIPCHECK(hipMemset(w.u_count.p, 0, w.u_count.cap));
    }
    TRY(w.u_cursor.ensure(L * 4));
    TRY(w.u_pair_off.ensure((L + 1) * 4));
    TRY(w.u_unit_prefix.ensure(tk_unit_prefix_ints((int64_t)L) * 4));
    TRY(w.u_pair_q.ensure(((size_t)nq * p.S + 4 * L) * 4));
    TRY(w.u_pair_f0.ensure(((size_t)nq * p.S + 4 * L) * 4));
    TRY(w.c_pair_off.ensure(8));
    TRY(w.c_unit_prefix.ensure(tk_unit_prefix_ints(1) * 4));
    TRY(w.c_pair_q.ensure(((size_t)nq + 4) * 4));
    TRY(w.c_pair_f0.ensure(((size_t)nq + 4) * 4));
    if (plain_possible(ix, p)) {
        TRY(w.qlim.ensure((size_t)nq * 4));
        TRY(w.slot_exact.ensure((size_t)nq * 4));
        const void *before = w.p_count.p;
        TRY(w.p_count.ensure(L * 4));
        if (w.p_count.p != before) HIPCHECK(hipMemset(w.p_count.p, 0, w.p_count.cap));
        TRY(w.p_cursor.ensure(L * 4));
        TRY(w.p_pair_off.ensure((L + 1) * 4));
        TRY(w.p_unit_prefix.ensure(tk_unit_prefix_ints((int64_t)L) * 4));
        TRY(w.p_pair_q.ensure(((size_t)nq * p.S + 4) * 4));
        TRY(w.p_pair_f0.ensure(((size_t)nq * p.S + 4) * 4));
        TRY(w.flag_list.ensure(((size_t)nq + 1) * 4));
        TRY(w.p_unit_desc.ensure(plain_desc_bytes(ix, nq, p)));
        TRY(w.plain0.ensure((size_t)nq * 4));
        if (!w.flag_host) {
            HIPCHECK(hipHostMalloc((void **)&w.flag_host, 64, hipHostMallocDefault));
            *w.flag_host = 0;
        }
        if (!w.plain_ev) HIPCHECK(hipEventCreateWithFlags(&w.plain_ev, hipEventDisableTiming));
        const void *hb = w.h_count.p;
        TRY(w.h_count.ensure(L * 4));
        if (w.h_count.p != hb) HIPCHECK(hipMemset(w.h_count.p, 0, w.h_count.cap));
        TRY(w.h_cursor.ensure(L * 4));
        TRY(w.h_pair_off.ensure((L + 1) * 4));
        TRY(w.h_unit_prefix.ensure(tk_unit_prefix_ints((int64_t)L) * 4));
        TRY(w.h_pair_q.ensure(((size_t)nq + 4 * L) * 4));
        TRY(w.h_pair_f0.ensure(((size_t)nq + 4 * L) * 4));
    }
    return TK_OK;
}

static const int64_t MAX_SUB = 32768;  // gridDim.y limit of the scan kernels is 65535
// a sharded batch only runs the list-major kernels (no gridDim.y); what bounds it is int32 unit
// counts and the 16 GB of distance rows per rank, both checked in shard_args
static const int64_t MAX_SHARD_BATCH = 131072;

// Queries per sub-batch: the distance buffer is nq * cap * 17 bytes (16 int8 + 1 minimum
// per chunk, cap = n_probes * longest list); one workspace keeps it under 16 GB (env
// TINYKNN_WORKSPACE_GB; up to depth + 5 workspaces exist — sized for 288 GB of HBM: at
// 100M x 128 with 10 000 lists a 4 GB workspace cut a batch of 10 000 queries in two, and the
// list-major scan then found 5 instead of 10 queries per list to share a fetched chunk).
static double workspace_bytes()
{
    static double b = 0;
    if (b == 0) {
        const char *e = getenv("TINYKNN_WORKSPACE_GB");
        const double g = e ? atof(e) : 0.0;
        b = (g >= 0.25 ? g : 16.0) * 1.0e9;
    }
    return b;
}

static int64_t sub_batch(const Plan &p)
{
    int64_t s = (int64_t)(workspace_bytes() / ((double)p.cap * 17.0));
    s = s < 64 ? 64 : s;
    return s < MAX_SUB ? s : MAX_SUB;
}

extern "C" int tk_index_reserve(tk_index *ix, int64_t nq, int k, int n_probes, int pass_1)
{
    IXLOCK(ix);
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    TRY(flush_pending(ix));
    const int64_t ms = sub_batch(p);
    for (Work &w : ix->works) TRY(reserve(ix, w, nq < ms ? nq : ms, k, p));
    return TK_OK;
}

// stage timers of one batch (tk_index_set_profiling)
#define TK_PROF_EVENTS 10     // per recorded batch: 8 stage marks + 2 around the plain kernel alone
struct Prof {
    const std::vector<hipEvent_t> *evs = nullptr;   // the index's event pool (it may grow)
    size_t base = 0;
    int evi = 0;
    int set = -1;
    bool plain_marked = false;
    int mark(hipStream_t st)
    {
        if (evs) HIPCHECK(hipEventRecord((*evs)[base + (size_t)evi++], st));
        return TK_OK;
    }
    int mark_plain(int which, hipStream_t st)       // 0: in front of the plain kernel, 1: behind it
    {
        if (evs) {
            HIPCHECK(hipEventRecord((*evs)[base + 8 + (size_t)which], st));
            plain_marked = true;
        }
        return TK_OK;
    }
};

static int prof_begin(tk_index *ix, Work &w, int64_t nq, const Plan &p, hipStream_t st, Prof &pf)
{
    // profiling = n: every n-th batch is timed (1 = every batch)
    if (ix->profiling == 0 || ix->ev_used >= 4096 || (ix->prof_seen++ % (uint64_t)ix->profiling) != 0)
        return TK_OK;
    while (ix->evs.size() < (ix->ev_used + 1) * TK_PROF_EVENTS) {
        hipEvent_t e;
        HIPCHECK(hipEventCreate(&e));
        ix->evs.push_back(e);
    }
    pf.evs = &ix->evs;
    pf.base = ix->ev_used * TK_PROF_EVENTS;
    if (ix->ev_streams.size() <= ix->ev_used) ix->ev_streams.resize(ix->ev_used + 1);
    if (ix->ev_plain.size() <= ix->ev_used) ix->ev_plain.resize(ix->ev_used + 1);
    ix->ev_streams[ix->ev_used] = st;
    ix->ev_plain[ix->ev_used] = 0;
    pf.set = (int)ix->ev_used;
    ix->ev_used++;
    ix->last_S = p.S; ix->last_R = p.R; ix->last_nq = nq;
    ix->last_work = (int)(&w - &ix->works[0]);
    return TK_OK;
}

// list-major scan (4 queries per pass over a chunk) when lists are shared by enough
// queries and the unit count fits int32; otherwise one query per wave
static bool use_units(const tk_index *ix, int64_t nq, const Plan &p)
{
    return ix->scan_mode == 2 ||
           (ix->scan_mode == 0 && nq * p.S >= 8 * ix->n_lists &&
            (double)nq * p.S / 4 * ix->max_list_chunks + (double)ix->total_chunks < 2.0e9);
}

// Stage 1 of a batch: distance tables (+ the descriptors of the list-major coarse scan).
static bool coarse_units(const tk_index *ix, int64_t nq)
{
    return ix->scan_mode != 1 && nq >= 16 && (double)nq / 4 * ix->center_chunks < 2.0e9;
}

static int stage_tables(tk_index *ix, Work &w, const void *qpq_dev, int qpq_f64, int64_t nq,
                        hipStream_t st, Prof &pf, bool plain = false, TkSecond qpq2 = TkSecond())
{
    TRY(pf.mark(st));
    // 1. distance tables                                   fast_pq.py:186-222
    tk_launch_build_tables(ix->pq_centers.as<float>(), ix->dq, ix->dpb, ix->f_order, qpq_dev,
                           qpq_f64, nq, ix->sqrt_nb, 0.0, 1, w.tables.as<uint8_t>(), w.shift.p,
                           w.scale.as<double>(), st, qpq2);
    if (plain)      // per query: below which value clamp(plain sum) is the saturated value
        tk_launch_table_limits(w.tables.as<uint4>(), ix->M, ix->order, nq, w.qlim.as<int>(), st, ix->opt_plain_limit);
    if (coarse_units(ix, nq))
        // every query scans the one list of coded centres: list-major, no idle lanes
        tk_launch_identity_pairs(nq, (int)ix->center_chunks, w.c_pair_off.as<int>(),
                                 w.c_unit_prefix.as<int>(), w.c_pair_q.as<int>(),
                                 w.c_pair_f0.as<int>(), st);
    TRY(pf.mark(st));
    return TK_OK;
}

// the coarse scan as a job of the list-major kernel
static TkScanJob coarse_job(const tk_index *ix, const Work &w, const Plan &p)
{
    TkScanJob j;
    j.codes = ix->center_codes.as<uint4>();
    j.tables = w.tables.as<uint4>();
    j.list_chunk_off = ix->c_chunk_off.as<int64_t>();
    j.n_lists = 1;
    j.unit_prefix = w.c_unit_prefix.as<int>();
    j.pair_off = w.c_pair_off.as<int>();
    j.pair_q = w.c_pair_q.as<int>();
    j.pair_f0 = w.c_pair_f0.as<int>();
    j.dist = w.cdist.as<uint4>();
    j.cap = ix->center_chunks;
    j.mins = w.cmins.as<uint8_t>();
    j.min_stride = p.ccap_min;
    return j;
}

static TkScanJob list_job(const tk_index *ix, const Work &w, const Plan &p)
{
    TkScanJob j;
    j.codes = ix->codes.as<uint4>();
    j.tables = w.tables.as<uint4>();
    j.list_chunk_off = ix->list_chunk_off.as<int64_t>();
    j.n_lists = (int)ix->n_lists;
    j.unit_prefix = w.u_unit_prefix.as<int>();
    j.pair_off = w.u_pair_off.as<int>();
    j.pair_q = w.u_pair_q.as<int>();
    j.pair_f0 = w.u_pair_f0.as<int>();
    j.dist = w.dist.as<uint4>();
    j.cap = p.cap;
    j.mins = w.mins.as<uint8_t>();
    j.min_stride = p.cap_min;
    return j;
}

static TkScanJob plain_job(const tk_index *ix, const Work &w, const Plan &p)
{
    TkScanJob j = list_job(ix, w, p);
    j.unit_prefix = w.p_unit_prefix.as<int>();
    j.pair_off = w.p_pair_off.as<int>();
    j.pair_q = w.p_pair_q.as<int>();
    j.pair_f0 = w.p_pair_f0.as<int>();
    j.unit_desc4 = w.p_unit_desc.as<int>();
    return j;
}

// Rows a query scans with the exact kernel before the plain sums take over, in heap sizes: the heap
// is then full of real values and its bound a low quantile of what it has seen.  2 where labels are
// distinct (a flagged query is then re-played by the packed kernel without the duplicate test:
// ~0.15 ms of one wave), 4 where they repeat (build_probes >= 2: with 2 about one query in 10 000
// was still flagged, and ONE flagged query costs its batch a 0.9 ms wave-per-query replay with
// the duplicate test; with 4 none in the bench batches).
static int head_rows(const tk_index *ix, const Plan &p)
{
    return (ix->ids_unique ? 2 : 4) * p.R;
}
// head pairs: the first ceil(head_rows / 16) chunks of the first probed list of a query in head mode
static int head_chunks(const tk_index *ix, const Plan &p) { return (head_rows(ix, p) + 15) >> 4; }

static TkScanJob head_job(const tk_index *ix, const Work &w, const Plan &p)
{
    TkScanJob j = list_job(ix, w, p);
    j.unit_prefix = w.h_unit_prefix.as<int>();
    j.pair_off = w.h_pair_off.as<int>();
    j.pair_q = w.h_pair_q.as<int>();
    j.pair_f0 = w.h_pair_f0.as<int>();
    j.max_chunks = head_chunks(ix, p);
    return j;
}

// persistent workgroups of the plain kernel: two per CU (58 KB of LDS, 256 registers per lane)
static int plain_blocks() { return 512; }

// 2a. coarse scan = the scan of dtable.top(centers)          ivf.py:131, fast_pq.py:284-312
static void launch_coarse_scan(tk_index *ix, Work &w, int64_t nq, const Plan &p, hipStream_t st,
                               const uint4 *tables = nullptr)
{
    const int M = ix->M;
    if (!tables) tables = w.tables.as<uint4>();
    if (coarse_units(ix, nq))
        tk_launch_scan_units(ix->center_codes.as<uint4>(), M, tables, nq, 1, 1,
                             ix->c_chunk_off.as<int64_t>(), w.c_pair_off.as<int>(),
                             w.c_unit_prefix.as<int>(), w.c_pair_q.as<int>(),
                             w.c_pair_f0.as<int>(), w.cdist.as<uint4>(), ix->center_chunks,
                             w.cmins.as<uint8_t>(), p.ccap_min, 1, ix->order, 768, st);
    else
        tk_launch_scan_flat(ix->center_codes.as<uint4>(), ix->center_chunks, M,
                            tables, nq, w.cdist.as<uint4>(), ix->center_chunks,
                            w.cmins.as<uint8_t>(), p.ccap_min, 1, ix->order, st);
}

// 2b. rest of the coarse stage: heap replay over the coded centres, probe lists, per-slot
// descriptors.  `pair_count`: per-list (query, slot) pair counters for the list-major scan
// (or NULL); with `owner` only the lists owned by `me` are counted (list-sharded index).
// `probes_out`: (nq, kc) int64, the probe lists (ivf.py:131) — w.probes, or a caller's buffer.
static int coarse_replay_probes(tk_index *ix, Work &w, const float *q_dev, int64_t nq, const Plan &p,
                                int64_t *probes_out, hipStream_t st, Prof &pf, TkSecond q2 = TkSecond())
{
    TRY(pf.mark(st));
    // positions of one list against a fresh heap are distinct labels: lane-per-query
    const bool fast_c = ix->heap_mode != 1 && ix->center_chunks * 16 <= 0xffffff;
    const bool lanes_c = fast_c && ix->heap_mode == 0 && p.rescore <= TK_LANES_MAX_R;
    if (fast_c && !lanes_c) {
        tk_launch_heap_replay_packed(w.cdist.as<uint4>(), ix->center_chunks, nq,
                                     ix->cslots_i.as<int>(), ix->cslots_i.as<int>() + 2,
                                     ix->cslots_l.as<int64_t>(), 1, nullptr,
                                     w.cheap_idx.as<int64_t>(), w.cheap_val.as<int32_t>(),
                                     p.rescore, 1, 1, nullptr, 0, 0, st);
    } else if (lanes_c) {
        if (tk_launch_heap_replay_lanes(w.cdist.as<uint4>(), ix->center_chunks, nq,
                                        ix->cslots_i.as<int>(), ix->cslots_i.as<int>() + 2,
                                        ix->cslots_l.as<int64_t>(), 1, nullptr,
                                        w.cheap_idx.as<int64_t>(), w.cheap_val.as<int32_t>(),
                                        p.rescore, 1, 1, nullptr, w.cmins.as<uint8_t>(),
                                        p.ccap_min, nullptr, st))
            return fail(TK_ERR_HIP, "hipFuncSetAttribute(LDS size) failed");
    } else {
        tk_launch_heap_fill(w.cheap_idx.as<int64_t>(), w.cheap_val.as<int32_t>(),
                            nq * p.rescore, 127, st);
        tk_launch_heap_replay(w.cdist.as<uint4>(), ix->center_chunks, nq,
                              ix->cslots_i.as<int>(), ix->cslots_i.as<int>() + 2,
                              ix->cslots_l.as<int64_t>(), 1, nullptr, w.cheap_idx.as<int64_t>(),
                              w.cheap_val.as<int32_t>(), p.rescore, 1, 1, nullptr, st);
    }
    TRY(pf.mark(st));
    tk_launch_rescore(q_dev, 0, ix->d, ix->active_centers.p, 0, ix->n_lists,
                      w.cheap_idx.as<int64_t>(), p.rescore, nq, p.kc, 0, probes_out, nullptr, st, ix->opt_rescore_form, q2);
    return TK_OK;
}

// per-slot descriptors of the probed lists of `nq` queries
static void coarse_slots(tk_index *ix, Work &w, const int64_t *probes, int64_t nq, const Plan &p,
                         int *pair_count, const int *owner, int me, hipStream_t st, bool plain = false)
{
    tk_launch_make_slots(probes, nullptr, p.S, nq, ix->n_lists, ix->list_chunk_off.as<int64_t>(),
                         ix->list_n.as<int64_t>(), ix->ids_off.as<int64_t>(),
                         w.slot_prefix.as<int>(), w.slot_chunk0.as<int64_t>(), w.slot_n.as<int>(),
                         w.slot_loff.as<int64_t>(), w.repeat_flag.as<unsigned char>(), pair_count,
                         owner, me, st, plain ? w.qlim.as<int>() : nullptr, head_rows(ix, p),
                         plain ? w.slot_exact.as<int>() : nullptr, plain ? w.p_count.as<int>() : nullptr,
                         plain ? w.plain0.as<int>() : nullptr, plain ? w.h_count.as<int>() : nullptr);
}

// the pair lists of a batch: one set for the exact list-major kernel, with `plain` a second one
// (the slots behind slot_exact[q]) for the plain kernel
static void unit_pairs(tk_index *ix, Work &w, int64_t nq, const Plan &p, bool plain, hipStream_t st)
{
    if (!plain) {
        tk_launch_unit_pairs(nq, w.probes.as<int64_t>(), p.S, ix->n_lists,
                             ix->list_chunk_off.as<int64_t>(), w.slot_prefix.as<int>(),
                             w.u_count.as<int>(), w.u_pair_off.as<int>(),
                             w.u_unit_prefix.as<int>(), w.u_cursor.as<int>(),
                             w.u_pair_q.as<int>(), w.u_pair_f0.as<int>(),
                             nq * p.S + 4 * ix->n_lists, st);
        return;
    }
    TkPairSet ex{w.u_count.as<int>(), w.u_cursor.as<int>(), w.u_pair_off.as<int>(),
                 w.u_unit_prefix.as<int>(), w.u_pair_q.as<int>(), w.u_pair_f0.as<int>()};
    TkPairSet pl{w.p_count.as<int>(), w.p_cursor.as<int>(), w.p_pair_off.as<int>(),
                 w.p_unit_prefix.as<int>(), w.p_pair_q.as<int>(), w.p_pair_f0.as<int>(),
                 w.p_unit_desc.as<int>(), plain_k(ix, nq, p)};
    TkPairSet hd{w.h_count.as<int>(), w.h_cursor.as<int>(), w.h_pair_off.as<int>(),
                 w.h_unit_prefix.as<int>(), w.h_pair_q.as<int>(), w.h_pair_f0.as<int>()};
    tk_launch_unit_pairs2(nq, w.probes.as<int64_t>(), p.S, ix->n_lists, ix->list_chunk_off.as<int64_t>(),
                          w.slot_prefix.as<int>(), w.slot_exact.as<int>(), ex, pl, hd, head_chunks(ix, p), st);
}

static int stage_coarse_rest(tk_index *ix, Work &w, const float *q_dev, int64_t nq, const Plan &p,
                             int *pair_count, const int *owner, int me, hipStream_t st, Prof &pf,
                             bool plain = false, TkSecond q2 = TkSecond())
{
    TRY(coarse_replay_probes(ix, w, q_dev, nq, p, w.probes.as<int64_t>(), st, pf, q2));
    coarse_slots(ix, w, w.probes.as<int64_t>(), nq, p, pair_count, owner, me, st, plain);
    return TK_OK;
}

// Stages 3b-4: the heap replay over the distance rows of queries [q0, q0 + nq) of the
// batch's slot arrays (dist/mins/heaps: `nq` rows starting at row 0), then the exact
// rescoring.  q_dev: row 0 = query q0.
// the queries the lane replay flagged (bound above the table's limit at the first plain block:
// plain_scan.hip): every probed list again with the exact kernel, then the replay again
static void rescan_flagged(tk_index *ix, Work &w, int64_t q0, int64_t nq, const Plan &p, hipStream_t st)
{
    int *list = w.flag_list.as<int>();
    tk_launch_flagged_list(w.repeat_flag.as<unsigned char>() + q0, nq, list, st, w.flag_host);
    if (w.plain_ev && !ix->capturing && hipEventRecord(w.plain_ev, st) == hipSuccess) {
        w.plain_pending = true;
        w.plain_nq = nq;
    }
    tk_launch_scan_probes(ix->codes.as<uint4>(), ix->M, w.tables.as<uint4>() + q0 * ix->M, nq,
                          w.slot_prefix.as<int>() + q0 * (p.S + 1), w.slot_chunk0.as<int64_t>() + q0 * p.S,
                          p.S, (int)p.cap, w.dist.as<uint4>(), p.cap, w.mins.as<uint8_t>(), p.cap_min, 1,
                          ix->order, st, list);
}

static int stage_back(tk_index *ix, Work &w, const float *q_dev, int64_t q0, int64_t nq, int k,
                      const Plan &p, int64_t *out_dev, hipStream_t st, Prof &pf, bool plain = false,
                      TkSecond q2 = TkSecond(), TkSecond out2 = TkSecond())
{
    const int *slot_exact = plain ? w.plain0.as<int>() + q0 : nullptr;     // (first plain chunk per query)
    const int *qlim = plain ? w.qlim.as<int>() + q0 : nullptr;
    const int *slot_prefix = w.slot_prefix.as<int>() + q0 * (p.S + 1);
    const int *slot_n = w.slot_n.as<int>() + q0 * p.S;
    const int64_t *slot_loff = w.slot_loff.as<int64_t>() + q0 * p.S;
    unsigned char *repeat_flag = w.repeat_flag.as<unsigned char>() + q0;
    // heaps start fresh here, so packed entries apply.  Distinct labels: one query per
    // lane (or per wave for big heaps), and the few queries whose probe list wrapped a -1
    // (a list may then be scanned twice) re-run with the duplicate test.  Repeating labels
    // (build n_probes >= 2): the packed wave kernel with the duplicate test for everybody.
    const bool packed_ok = ix->heap_mode != 1 && p.cap * 16 <= 0xffffff;
    if (packed_ok && ix->ids_unique) {
        const bool lanes = ix->heap_mode == 0 && p.R <= TK_LANES_MAX_R;
        if (!lanes)
            tk_launch_heap_replay_packed(w.dist.as<uint4>(), p.cap, nq, slot_prefix, slot_n,
                                         slot_loff, p.S, ix->ids.as<int64_t>(),
                                         w.heap_idx.as<int64_t>(), w.heap_val.as<int32_t>(), p.R,
                                         1, 0, repeat_flag, 0, 0, st);
        else if (tk_launch_heap_replay_lanes(w.dist.as<uint4>(), p.cap, nq, slot_prefix, slot_n,
                                             slot_loff, p.S, ix->ids.as<int64_t>(),
                                             w.heap_idx.as<int64_t>(), w.heap_val.as<int32_t>(),
                                             p.R, 1, 0, repeat_flag, w.mins.as<uint8_t>(),
                                             p.cap_min, nullptr, st, slot_exact, qlim))
            return fail(TK_ERR_HIP, "hipFuncSetAttribute(LDS size) failed");
        if (plain) {
            // flag 2 = the lane replay's "bound above the limit at the first plain block": exact
            // re-scan, then the packed kernel from a fresh heap (labels are distinct: no duplicate test)
            rescan_flagged(ix, w, q0, nq, p, st);
            tk_launch_heap_replay_packed(w.dist.as<uint4>(), p.cap, nq, slot_prefix, slot_n, slot_loff,
                                         p.S, ix->ids.as<int64_t>(), w.heap_idx.as<int64_t>(),
                                         w.heap_val.as<int32_t>(), p.R, 1, 0, repeat_flag, 2, 0, st);
        }
        tk_launch_heap_replay_packed(w.dist.as<uint4>(), p.cap, nq, slot_prefix, slot_n, slot_loff,
                                     p.S, ix->ids.as<int64_t>(), w.heap_idx.as<int64_t>(),
                                     w.heap_val.as<int32_t>(), p.R, 1, 0, repeat_flag, 1, 1, st);
    } else if (packed_ok && ix->have_ids32 && ix->heap_mode == 0 && tk_lanes_dedupe_fits(p.R, p.S) &&
               ix->total_ids < (1ll << 31)) {
        // repeating labels that fit int32: one query per lane with the duplicate test
        // (plain: the queries whose probe list wrapped are left to the packed kernel below too)
        if (tk_launch_heap_replay_lanes(w.dist.as<uint4>(), p.cap, nq, slot_prefix, slot_n,
                                        slot_loff, p.S, ix->ids.as<int64_t>(),
                                        w.heap_idx.as<int64_t>(), w.heap_val.as<int32_t>(), p.R, 1,
                                        0, plain ? repeat_flag : nullptr, w.mins.as<uint8_t>(), p.cap_min,
                                        ix->ids32.as<int32_t>(), st, slot_exact, qlim))
            return fail(TK_ERR_HIP, "hipFuncSetAttribute(LDS size) failed");
        if (plain) {
            rescan_flagged(ix, w, q0, nq, p, st);
            tk_launch_heap_replay_packed(w.dist.as<uint4>(), p.cap, nq, slot_prefix, slot_n, slot_loff,
                                         p.S, ix->ids.as<int64_t>(), w.heap_idx.as<int64_t>(),
                                         w.heap_val.as<int32_t>(), p.R, 1, 0, repeat_flag, 1, 1, st);
        }
    } else if (packed_ok) {
        tk_launch_heap_replay_packed(w.dist.as<uint4>(), p.cap, nq, slot_prefix, slot_n, slot_loff,
                                     p.S, ix->ids.as<int64_t>(), w.heap_idx.as<int64_t>(),
                                     w.heap_val.as<int32_t>(), p.R, 1, 0, nullptr, 0, 1, st);
    } else {
        tk_launch_heap_fill(w.heap_idx.as<int64_t>(), w.heap_val.as<int32_t>(), nq * p.R, 127, st);
        tk_launch_heap_replay(w.dist.as<uint4>(), p.cap, nq, slot_prefix, slot_n, slot_loff, p.S,
                              ix->ids.as<int64_t>(), w.heap_idx.as<int64_t>(),
                              w.heap_val.as<int32_t>(), p.R, 1, 0, nullptr, st);
    }
    TRY(pf.mark(st));
    // 4. strip sentinels, exact rescoring                   ivf.py:154-163
    tk_launch_rescore(q_dev, 0, ix->d, ix->data.p, ix->data_is_f64, ix->N, w.heap_idx.as<int64_t>(), p.R, nq, k, 1,
                      out_dev, nullptr, st, ix->opt_rescore_form, q2, out2);
    TRY(pf.mark(st));
    return TK_OK;
}

// One sub-batch.
//
// depth == 1: seven stages back to back on the caller's stream.
//
// depth  > 1 (tk_index_set_pipeline): two kinds of kernels make up a batch — chip-filling,
// VALU-bound scans and latency-bound rest (table build: many small workgroups; heap replays:
// 157 waves per 10 000 queries; rescoring; descriptors).  Two scans at once only stretch each
// other, so ALL scans run on the caller's stream, in order; the table builds and the coarse
// replays + descriptors of all batches on one internal "front" stream; the heap replay +
// rescoring of a batch on one of `depth` more; handed over by events.  Call c enqueues
//     tables(c)                                      front stream
//     [ list scan(c-3)  +  coarse scan(c-1) ]        ONE launch on the caller's stream: one
//                                                    pool of 64-unit blocks, drawn by tickets
//     coarse replay(c-1), probes, descriptors(c-1)   front stream
//     heap replay(c-3), rescoring(c-3)               replay stream (c-3) mod depth
// so the caller's stream is a chain of scan launches that never waits — every launch finds
// its tables (built one call earlier) and its descriptors (two calls earlier) finished —
// while the replays of the previous batches overlap all of it.
// tk_index_join enqueues the launches still owed and re-joins.
struct Pending {
    Work *w;
    const float *q_dev;
    int64_t nq;
    int k;
    Plan p;
    int64_t *out_dev;
    bool units;
    bool plain;             // probed lists behind the first ones by the plain kernel (plain_scan.hip)
    bool coarse_launched;   // its coarse scan has been enqueued
    int64_t *host_out;      // pinned host copy of the ids, enqueued behind the rescoring (or NULL)
    bool host_out_kernel;   // ... written by copy_words_kernel instead of the copy engine
    hipEvent_t user_ev;     // recorded behind that copy (or NULL)
    Prof pf;
    hipStream_t st, sf, sl;   // scans (+ tables) / coarse replay + descriptors / replay + rescoring
    // coalesced calls: the batch is the rows of n_subs calls, each read from and written to the call's
    // OWN buffers (no staging copies: the three kernels that touch them take a second base pointer)
    struct Sub {
        int64_t *out_dev;
        int64_t nq;
        int64_t *host_out;
        bool host_out_kernel;
        hipEvent_t user_ev;
        const float *q_dev;
        const void *qpq_dev;
    } subs[2];
    int n_subs = 0;
    TkSecond q2, qpq2, out2;   // rows of the second call (empty: one call)
};

// depth == 1
static int run_batch_inline(tk_index *ix, Pending &b, const void *qpq_dev, int qpq_f64)
{
    Work &w = *b.w;
    const Plan &p = b.p;
    const int M = ix->M;
    hipStream_t st = b.st;
    TRY(prof_begin(ix, w, b.nq, p, st, b.pf));
    b.units = use_units(ix, b.nq, p);
    b.plain = b.units && plain_now(ix, p);
    w.last_plain = b.plain;
    TRY(stage_tables(ix, w, qpq_dev, qpq_f64, b.nq, st, b.pf, b.plain, b.qpq2));
    launch_coarse_scan(ix, w, b.nq, p, st);
    TRY(stage_coarse_rest(ix, w, b.q_dev, b.nq, p, b.units ? w.u_count.as<int>() : nullptr, nullptr,
                          0, st, b.pf, b.plain, b.q2));
    if (b.units) unit_pairs(ix, w, b.nq, p, b.plain, st);
    TRY(b.pf.mark(st));
    // 3. probed lists through ONE heap, in probe order      ivf.py:135-150
    // (plain first: the exact kernel then overwrites the head chunks of the lists in head mode)
    if (b.plain) TRY(b.pf.mark_plain(0, st));
    if (b.plain && tk_launch_scan_plain(plain_job(ix, w, p), M, ix->order, plain_blocks(), st))
        return fail(TK_ERR_HIP, "scan_plain_kernel: LDS attribute / unsupported M");
    if (b.plain) TRY(b.pf.mark_plain(1, st));
    if (b.plain && b.pf.evs && b.pf.set >= 0) ix->ev_plain[(size_t)b.pf.set] = 1;
    if (b.plain) {
        TkScanJob none;
        memset(&none, 0, sizeof none);
        const TkScanJob hj = head_job(ix, w, p);
        tk_launch_scan_units2(list_job(ix, w, p), none, M, ix->order, 768, st, &hj, ix->opt_scan_form);
    } else if (b.units)
        tk_launch_scan_units(ix->codes.as<uint4>(), M, w.tables.as<uint4>(), b.nq, p.S, ix->n_lists,
                             ix->list_chunk_off.as<int64_t>(), w.u_pair_off.as<int>(),
                             w.u_unit_prefix.as<int>(), w.u_pair_q.as<int>(),
                             w.u_pair_f0.as<int>(), w.dist.as<uint4>(), p.cap,
                             w.mins.as<uint8_t>(), p.cap_min, 1, ix->order, 768, st, ix->opt_scan_form);
    else
        tk_launch_scan_probes(ix->codes.as<uint4>(), M, w.tables.as<uint4>(), b.nq,
                              w.slot_prefix.as<int>(), w.slot_chunk0.as<int64_t>(), p.S,
                              (int)p.cap, w.dist.as<uint4>(), p.cap, w.mins.as<uint8_t>(),
                              p.cap_min, 1, ix->order, st);
    TRY(b.pf.mark(st));
    TRY(stage_back(ix, w, b.q_dev, 0, b.nq, b.k, p, b.out_dev, st, b.pf, b.plain, b.q2, b.out2));
    TRY(batch_epilogue(b, st));
    HIPCHECK(hipGetLastError());
    return TK_OK;
}

// Persistent workgroups of the fused scan launch in the pipelined mode: 512 = two per CU.
// Three per CU (768) is the residency the kernel's 145 VGPRs allow on an EMPTY chip and is the
// faster grid for a launch that runs alone; next to the other batches' kernels a CU that hosts a
// replay wave (120+ VGPRs) has no room for a third scan workgroup, which then waits for a slot
// while its share of the work is drawn by others — measured per 10 000 queries: 768 -> 0.705 ms,
// 640 -> 0.703, 576 -> 0.682, 512 -> 0.657, 448 -> 0.678, 384 -> 0.743 (profiles/r02_scan_grid.md).
// Long launches (100M x 128: 15 M units, 3 ms) amortise that wait and prefer more resident
// waves: 512 -> 4.15 ms per batch, 576 -> 3.95, 640 -> 3.89, 704 -> 3.84, 768 -> 4.03
// (profiles/r02_scan_grid.md), so the grid is 704 above ~6 M estimated units.
static int scan_blocks_pipelined(double est_units) { return est_units > 6.0e6 ? 704 : 512; }

// depth > 1: the launch on the caller's stream that carries the list scan of `prev` (may be
// NULL) and the coarse scan of `cur` (may be NULL), and what follows each on its stream.
// (Measured and dropped in rounds 2-3, profiles/HISTORY.md: table builds on a replay stream or on the
// scan stream, two front streams, descriptors on the scan stream, a high-priority front stream, replay
// streams confined to a CU mask — none moved the batch.)
static int pipeline_step(tk_index *ix, Pending *prev, Pending *cur)
{
    const int M = ix->M;
    hipStream_t st = prev ? prev->st : cur->st;
    // what the launch waits for lives on the front stream, in order: ..., front_done(c-3),
    // tables_done(c-1), ... — the later event covers the earlier one, and every hand-over
    // between streams is a barrier packet the command processor spends microseconds on
    if (cur) {
        HIPCHECK(hipStreamWaitEvent(st, cur->w->tables_done, 0));
        cur->coarse_launched = true;
    }
    if (prev) {
        // (the shortcut holds only if front_done(prev) was recorded BEFORE tables_done(cur), which the
        // three-call distance guarantees — but not a drain of two batches: there the coarse rest of
        // `prev` was enqueued in the same call as, and behind, the table build of `cur`)
        const bool covered = cur && prev->sf == cur->sf && prev->w->fd_seq < cur->w->td_seq;
        if (!covered) HIPCHECK(hipStreamWaitEvent(st, prev->w->front_done, 0));
        TRY(prev->pf.mark(st));
    }
    const bool fuse_prev = prev && prev->units;
    const bool fuse_cur = cur && coarse_units(ix, cur->nq);
    if (prev && !fuse_prev)
        tk_launch_scan_probes(ix->codes.as<uint4>(), M, prev->w->tables.as<uint4>(), prev->nq,
                              prev->w->slot_prefix.as<int>(), prev->w->slot_chunk0.as<int64_t>(),
                              prev->p.S, (int)prev->p.cap, prev->w->dist.as<uint4>(), prev->p.cap,
                              prev->w->mins.as<uint8_t>(), prev->p.cap_min, 1, ix->order, st);
    if (cur && !fuse_cur) launch_coarse_scan(ix, *cur->w, cur->nq, cur->p, st);
    // (plain first: the exact kernel then overwrites the head chunks of the lists in head mode)
    if (prev && prev->plain) {
        TRY(prev->pf.mark_plain(0, st));
        if (tk_launch_scan_plain(plain_job(ix, *prev->w, prev->p), M, ix->order, plain_blocks(), st))
            return fail(TK_ERR_HIP, "scan_plain_wave_kernel: LDS attribute / unsupported M");
        TRY(prev->pf.mark_plain(1, st));
        if (prev->pf.evs && prev->pf.set >= 0) ix->ev_plain[(size_t)prev->pf.set] = 1;
    }
    if (fuse_prev || fuse_cur) {
        TkScanJob none;
        memset(&none, 0, sizeof none);
        TkScanJob hj = none;
        if (prev && prev->plain) hj = head_job(ix, *prev->w, prev->p);
        tk_launch_scan_units2(fuse_prev ? list_job(ix, *prev->w, prev->p) : none,
                              fuse_cur ? coarse_job(ix, *cur->w, cur->p) : none, M, ix->order,
                              scan_blocks_pipelined(fuse_prev ? (double)prev->nq * prev->p.S / 4.0 *
                                                    ((double)ix->total_chunks / (double)ix->n_lists) : 0.0),
                              st, &hj, ix->opt_scan_form);
    }
    TK_DBG_SYNC("step: scans");
    if (prev) {
        // heap replay + rescoring of the previous batch on its stream
        TRY(prev->pf.mark(st));
        HIPCHECK(hipEventRecord(prev->w->scanned, st));
        HIPCHECK(hipStreamWaitEvent(prev->sl, prev->w->scanned, 0));
        TRY(stage_back(ix, *prev->w, prev->q_dev, 0, prev->nq, prev->k, prev->p, prev->out_dev,
                       prev->sl, prev->pf, prev->plain, prev->q2, prev->out2));
        TRY(batch_epilogue(*prev, prev->sl));
        HIPCHECK(hipEventRecord(prev->w->done, prev->sl));
        prev->w->busy = true;
        TK_DBG_SYNC("step: back");
    }
    if (cur) {
        // rest of the coarse stage + scan descriptors of this batch on its stream (one event
        // behind the launch serves both consumers)
        Work &w = *cur->w;
        if (prev) {
            HIPCHECK(hipStreamWaitEvent(cur->sf, prev->w->scanned, 0));
        } else {
            HIPCHECK(hipEventRecord(w.coarse_scanned, st));
            HIPCHECK(hipStreamWaitEvent(cur->sf, w.coarse_scanned, 0));
        }
        TRY(stage_coarse_rest(ix, w, cur->q_dev, cur->nq, cur->p, cur->units ? w.u_count.as<int>() : nullptr,
                              nullptr, 0, cur->sf, cur->pf, cur->plain, cur->q2));
        if (cur->units) unit_pairs(ix, w, cur->nq, cur->p, cur->plain, cur->sf);
        HIPCHECK(hipEventRecord(w.front_done, cur->sf));
        w.fd_seq = ++ix->ev_seq;
        TK_DBG_SYNC("step: coarse rest");
    }
    HIPCHECK(hipGetLastError());
    return TK_OK;
}

// the launch a call (or a flush) owes: the list scan of the oldest call once three are
// pending (`drain`: of the oldest call in any case) + the coarse scan of the newest call
// that has not had one
static int pipeline_advance(tk_index *ix, bool drain)
{
    Pending *coarse = nullptr;
    for (Pending *b : ix->pending)
        if (!b->coarse_launched) { coarse = b; break; }
    Pending *scan = nullptr;
    if (!ix->pending.empty() && ix->pending.front()->coarse_launched &&
        (drain || ix->pending.size() >= 3))
        scan = ix->pending.front();
    if (!scan && !coarse) return TK_OK;
    int r = pipeline_step(ix, scan, coarse);
    if (scan) {
        ix->pending.erase(ix->pending.begin());
        delete scan;
    }
    return r;
}

// what the caller of tk_index_query_batch_dev_ex asked for behind a batch's last kernel
static int batch_epilogue(const Pending &b, hipStream_t st)
{
    if (b.n_subs > 0) {
        for (int i = 0; i < b.n_subs; i++) {      // (the rescoring wrote each call's ids to its own buffer)
            const Pending::Sub &u = b.subs[i];
            if (u.host_out && u.host_out_kernel)
                tk_launch_copy_words(u.out_dev, u.nq * b.k, u.host_out, st);
            else if (u.host_out)
                HIPCHECK(hipMemcpyAsync(u.host_out, u.out_dev, (size_t)u.nq * b.k * 8, hipMemcpyDeviceToHost, st));
            if (u.user_ev) HIPCHECK(hipEventRecord(u.user_ev, st));
        }
        return TK_OK;
    }
    if (b.host_out && b.host_out_kernel)
        tk_launch_copy_words(b.out_dev, b.nq * b.k, b.host_out, st);
    else if (b.host_out)
        HIPCHECK(hipMemcpyAsync(b.host_out, b.out_dev, (size_t)b.nq * b.k * 8, hipMemcpyDeviceToHost, st));
    if (b.user_ev) HIPCHECK(hipEventRecord(b.user_ev, st));
    return TK_OK;
}

static int launch_held(tk_index *ix);

static int flush_pending(tk_index *ix)
{
    int r = TK_OK;
    if (ix->held) r = launch_held(ix);
    while (!ix->pending.empty() && r == TK_OK) r = pipeline_advance(ix, true);
    for (Pending *b : ix->pending) delete b;
    ix->pending.clear();
    return r;
}

// The front stream's chain (table build, coarse replay + rescoring, descriptors: ten short kernels per
// batch, each waiting for the one before) is the pipeline's critical path once the scans overlap: its
// kernels go first when CU slots free up.  Same box, ms per 10 000 queries: 0.430 default priority,
// 0.417 high, 0.443 low; the replay streams high as well: 0.425 (profiles/r04/ab_front_prio.txt)
static hipError_t make_front_stream(hipStream_t *st)
{
    int lo = 0, hi = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (e != hipSuccess) return e;
    return hipStreamCreateWithPriority(st, hipStreamNonBlocking, hi);
}

// Pipelined mode, first half of enqueuing a batch: internal streams and events exist, the batch has
// its workspace and streams, and `stt` — the stream its table build will run on — waits for the
// caller's work so far and for the workspace's previous batch.
static int pipe_begin(tk_index *ix, Pending &b, hipStream_t caller, hipStream_t &stt_out)
{
    Work &w = *b.w;
    while ((int)ix->lat_streams.size() < ix->depth) {
        hipStream_t st;
        HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        ix->lat_streams.push_back(st);
    }
    if (!ix->front_stream) HIPCHECK(make_front_stream(&ix->front_stream));
    b.sf = ix->front_stream;
    b.sl = ix->lat_streams[ix->calls % (uint64_t)ix->depth];
    ix->calls++;
    hipEvent_t *evs[] = {&w.tables_done, &w.coarse_scanned, &w.front_done, &w.scanned, &w.done};
    for (hipEvent_t *e : evs)
        if (!*e) HIPCHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    if (!ix->ev_in) HIPCHECK(hipEventCreateWithFlags(&ix->ev_in, hipEventDisableTiming));
    // the table build of this call goes to the front stream now (after the caller's work
    // so far — its inputs — and once the workspace is free); its coarse scan rides in the
    // NEXT call's launch, its list scan in the launch three calls later
    hipStream_t stt = b.sf;
    if (stt != caller) {
        HIPCHECK(hipEventRecord(ix->ev_in, caller));
        HIPCHECK(hipStreamWaitEvent(stt, ix->ev_in, 0));
    }
    if (w.busy) HIPCHECK(hipStreamWaitEvent(stt, w.done, 0));
    stt_out = stt;
    return TK_OK;
}

// ... second half: workspace sized, tables built, this call's launch enqueued, the batch pending
static int pipe_launch(tk_index *ix, Pending &b, const void *qpq, int q_pq_is_f64, hipStream_t stt)
{
    Work &w = *b.w;
    const int64_t sub = b.nq;
    const int k = b.k;
    const Plan &p = b.p;
    TRY(reserve(ix, w, sub, k, p));
    TRY(prof_begin(ix, w, b.nq, p, b.sl, b.pf));
    b.units = use_units(ix, b.nq, p);
    b.plain = b.units && plain_now(ix, p);
    w.last_plain = b.plain;
    TK_DBG_SYNC("launch: reserved");
    TRY(stage_tables(ix, w, qpq, q_pq_is_f64, b.nq, stt, b.pf, b.plain, b.qpq2));
    HIPCHECK(hipEventRecord(w.tables_done, stt));
    w.td_seq = ++ix->ev_seq;
    TK_DBG_SYNC("launch: tables");
    // this call's launch: list scan of call c-3 + coarse scan of call c-1
    TRY(pipeline_advance(ix, false));
    ix->pending.push_back(new Pending(b));
    return TK_OK;
}

// ---- two consecutive calls as ONE batch (tk_index_set_coalesce(ix, 2), pipelined mode) ----
// The kernels that leave most of the chip idle — the two heap replays (157 waves of 64 queries for
// 10 000 queries, a dependent chain per wave), the nine small kernels of the front stream — take as
// long for 20 000 queries as for 10 000, and the plain kernel's tiles fill better with twice the
// pairs per list.  The first call of a pair is only HELD; the second call joins it and the pair runs
// through the pipeline as one batch of nq_a + nq_b queries.  Nothing is copied: the three kernels
// that touch the callers' buffers (table build, the two rescorings) take a second base pointer for
// the rows of the second call, and each call's ids are written straight to its own buffer (its
// pinned copy and completion event follow behind the last kernel).  The buffers of a call are the
// caller's until tk_index_join, as in the pipelined mode without pairs (tinyknn_hip.h).
// Same kernels on the same rows: results do not change.  A held call is launched alone by
// tk_index_join / quiesce / set_* and when the next call cannot join it (other k / n_probes /
// pass_1 / stream, or too many rows).
static int launch_held(tk_index *ix)
{
    Pending *h = ix->held;
    ix->held = nullptr;
    if (!h) return TK_OK;
    Pending b = *h;
    delete h;
    int64_t rows = 0;
    for (int i = 0; i < b.n_subs; i++) rows += b.subs[i].nq;
    b.nq = rows;
    b.q_dev = b.subs[0].q_dev;
    b.out_dev = b.subs[0].out_dev;
    if (b.n_subs == 2) {
        const int64_t n_a = b.subs[0].nq;
        b.q2 = TkSecond{b.subs[1].q_dev, n_a};
        b.qpq2 = TkSecond{b.subs[1].qpq_dev, n_a};
        b.out2 = TkSecond{b.subs[1].out_dev, n_a};
    }
    TK_DBG_SYNC("launch_held");
    int r_ = pipe_launch(ix, b, b.subs[0].qpq_dev, ix->held_f64, ix->held_stt);
    TK_DBG_SYNC("launch_held done");
    return r_;
}

static int coalesce_call(tk_index *ix, const Plan &p, const float *q_dev, const void *q_pq_dev, int q_pq_is_f64,
                         int64_t nq, int k, int n_probes, int pass_1, int64_t *out_ids_dev,
                         int64_t *out_ids_pinned, hipEvent_t done_ev, hipStream_t caller)
{
    const Pending::Sub sub{out_ids_dev, nq, out_ids_pinned, ix->host_out_kernel, done_ev, q_dev, q_pq_dev};
    if (ix->held) {
        Pending &h = *ix->held;
        const bool joins = h.k == k && ix->held_n_probes == n_probes && ix->held_pass_1 == pass_1 &&
                           ix->held_f64 == q_pq_is_f64 && ix->held_caller == caller &&
                           h.subs[0].nq + nq <= ix->held_rows;
        if (joins) {
            hipStream_t stt = ix->held_stt;
            if (stt != caller) {            // the second call's inputs: the caller's work so far
                HIPCHECK(hipEventRecord(ix->ev_in, caller));
                HIPCHECK(hipStreamWaitEvent(stt, ix->ev_in, 0));
            }
            h.subs[1] = sub;
            h.n_subs = 2;
            return launch_held(ix);
        }
        TRY(launch_held(ix));
    }
    // first of a pair: workspace, streams, inputs staged; rows for a second call of the same size
    Work &w = ix->works[ix->calls % ix->works.size()];
    Pending b;
    b.w = &w;
    b.q_dev = nullptr;
    b.nq = nq;
    b.k = k;
    b.p = p;
    b.out_dev = nullptr;
    b.units = b.plain = b.coarse_launched = false;
    b.host_out = nullptr;
    b.host_out_kernel = false;
    b.user_ev = nullptr;
    b.st = b.sf = b.sl = caller;
    b.subs[0] = sub;
    b.n_subs = 1;
    hipStream_t stt = nullptr;
    TRY(pipe_begin(ix, b, caller, stt));
    const int64_t ms = sub_batch(p);
    const int64_t rows = 2 * nq <= ms ? 2 * nq : nq;
    ix->held = new Pending(b);
    ix->held_n_probes = n_probes;
    ix->held_pass_1 = pass_1;
    ix->held_f64 = q_pq_is_f64;
    ix->held_rows = rows;
    ix->held_stt = stt;
    ix->held_caller = caller;
    if (rows == nq) return launch_held(ix);     // (no room for a second call: alone, at once)
    return TK_OK;
}

static int query_batch_dev_impl(tk_index *ix, const float *q_dev, const void *q_pq_dev,
                                int q_pq_is_f64, int64_t nq, int k, int n_probes, int pass_1,
                                int64_t *out_ids_dev, int64_t *out_ids_pinned, hipEvent_t done_ev,
                                void *stream)
{
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    ARGCHECK(nq >= 0, "nq");
    ARGCHECK(!ix->sharded, "list-sharded index: use tk_index_shard_scan_dev / _finish_dev");
    hipStream_t caller = (hipStream_t)stream;
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        ix->capturing = caller != nullptr && hipStreamIsCapturing(caller, &cs) == hipSuccess &&
                        cs != hipStreamCaptureStatusNone;
        (void)hipGetLastError();
    }
    const size_t esz = q_pq_is_f64 ? 8 : 4;
    const int64_t ms = sub_batch(p);
    ARGCHECK(!(out_ids_pinned || done_ev) || (nq >= 1 && nq <= ms),
             "a completion event / host copy belongs to ONE sub-batch (tk_index_max_sub_batch)");
    // (repeating labels — IVF.build(n_probes >= 2) — run the duplicate-test replay: 32 queries and 70 KB
    //  of LDS per wave, two waves per CU; a doubled batch would not fit the chip in one round of waves:
    //  5.1 M queries/s paired against 7.7 M alone, profiles/r04/bench_full_first.json)
    if (ix->depth > 1 && ix->coalesce == 2 && ix->ids_unique && nq >= 1 && nq <= ms)
        return coalesce_call(ix, p, q_dev, q_pq_dev, q_pq_is_f64, nq, k, n_probes, pass_1, out_ids_dev,
                             out_ids_pinned, done_ev, caller);
    if (ix->held) TRY(launch_held(ix));
    for (int64_t o = 0; o < nq; o += ms) {
        int64_t sub = nq - o < ms ? nq - o : ms;
        Work &w = ix->works[ix->calls % ix->works.size()];
        Pending b;
        b.w = &w;
        b.q_dev = q_dev + o * ix->d;
        b.nq = sub;
        b.k = k;
        b.p = p;
        b.out_dev = out_ids_dev + o * k;
        b.units = false;
        b.plain = false;
        b.coarse_launched = false;
        b.host_out = out_ids_pinned;
        b.host_out_kernel = ix->host_out_kernel;
        b.user_ev = done_ev;
        b.st = b.sf = b.sl = caller;
        const void *qpq = (const char *)q_pq_dev + (size_t)o * ix->dq * esz;
        if (ix->depth == 1) {
            ix->calls++;
            TRY(reserve(ix, w, sub, k, p));
            TRY(run_batch_inline(ix, b, qpq, q_pq_is_f64));
            continue;
        }
        hipStream_t stt = nullptr;
        TRY(pipe_begin(ix, b, caller, stt));
        TRY(pipe_launch(ix, b, qpq, q_pq_is_f64, stt));
    }
    return TK_OK;
}

extern "C" int tk_index_query_batch_dev(tk_index *ix, const float *q_dev, const void *q_pq_dev,
                                        int q_pq_is_f64, int64_t nq, int k, int n_probes,
                                        int pass_1, int64_t *out_ids_dev, void *stream)
{
    IXLOCK(ix);
    return query_batch_dev_impl(ix, q_dev, q_pq_dev, q_pq_is_f64, nq, k, n_probes, pass_1,
                                out_ids_dev, nullptr, nullptr, stream);
}

extern "C" int tk_index_query_batch_dev_ex(tk_index *ix, const float *q_dev, const void *q_pq_dev,
                                           int q_pq_is_f64, int64_t nq, int k, int n_probes,
                                           int pass_1, int64_t *out_ids_dev,
                                           int64_t *out_ids_pinned, void *done_event, void *stream)
{
    IXLOCK(ix);
    return query_batch_dev_impl(ix, q_dev, q_pq_dev, q_pq_is_f64, nq, k, n_probes, pass_1,
                                out_ids_dev, out_ids_pinned, (hipEvent_t)done_event, stream);
}

void tk_index_host_out_by_kernel(tk_index *ix, bool on) { ix->host_out_kernel = on; }   // front.hip

extern "C" int64_t tk_index_max_sub_batch(tk_index *ix, int k, int n_probes, int pass_1)
{
    IXLOCK(ix);
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    return sub_batch(p);
}

// Stream on which a caller should copy a batch's inputs in: the front stream in pipelined
// mode (the table build, a batch's first kernel, runs there; a fifth stream of the caller's
// would share one of HIP's four hardware queues and serialise with a replay stream), NULL
// = the stream the batch is enqueued on.
extern "C" void *tk_index_input_stream(tk_index *ix)
{
    IXLOCK(ix);
    if (!ix || ix->depth <= 1) return nullptr;
    if (!ix->front_stream && make_front_stream(&ix->front_stream) != hipSuccess) return nullptr;
    return ix->front_stream;
}

// calls whose last stage has not been enqueued yet
extern "C" int tk_index_pending(tk_index *ix)
{
    if (!ix) return 0;
    int n = ix->held ? ix->held->n_subs : 0;
    for (const Pending *b : ix->pending) n += b->n_subs > 0 ? b->n_subs : 1;
    return n;
}

extern "C" int tk_index_info(tk_index *ix, int64_t *info8)
{
    IXLOCK(ix);
    ARGCHECK(ix && info8, "null index / buffer");
    ARGCHECK(ix->have_pq && ix->have_centers, "set_pq and set_centers first");
    info8[0] = ix->d; info8[1] = ix->dq; info8[2] = ix->M; info8[3] = ix->n_lists;
    info8[4] = ix->rot_d_pad; info8[5] = ix->depth; info8[6] = ix->N; info8[7] = ix->total_chunks;
    return TK_OK;
}

// ---------------------------------------------------------------------------
// list-sharded batch = shard_scan -> all-to-all of the send buffer -> shard_finish
static int reserve_shard(tk_index *ix, Work &w, int64_t nq, int64_t qh, const Plan &p)
{
    const int M = ix->M;
    TRY(w.tables.ensure((size_t)nq * M * 16));
    TRY(w.shift.ensure((size_t)nq * 8));
    TRY(w.scale.ensure((size_t)nq * 8));
    TRY(w.cdist.ensure((size_t)nq * ix->center_chunks * 16));
    TRY(w.cheap_idx.ensure((size_t)nq * p.rescore * 8));
    TRY(w.cheap_val.ensure((size_t)nq * p.rescore * 4));
    TRY(w.probes.ensure((size_t)nq * p.kc * 8));
    TRY(w.slot_prefix.ensure((size_t)nq * (p.S + 1) * 4));
    TRY(w.slot_chunk0.ensure((size_t)nq * p.S * 8));
    TRY(w.slot_n.ensure((size_t)nq * p.S * 4));
    TRY(w.slot_loff.ensure((size_t)nq * p.S * 8));
    TRY(w.repeat_flag.ensure((size_t)nq));
    TRY(w.cmins.ensure((size_t)nq * p.ccap_min));
    TRY(w.spos.ensure((size_t)nq * p.S * 4));
    TRY(w.rpos.ensure((size_t)qh * p.S * 4));
    // rows of the home queries only
    TRY(w.dist.ensure((size_t)qh * p.cap * 16));
    TRY(w.mins.ensure((size_t)qh * p.cap_min));
    TRY(w.heap_idx.ensure((size_t)qh * p.R * 8));
    TRY(w.heap_val.ensure((size_t)qh * p.R * 4));
    const size_t L = (size_t)ix->n_lists;
    {
        const void *before = w.u_count.p;
        TRY(w.u_count.ensure(L * 4));
        if (w.u_count.p != before) HIPCHECK(hipMemset(w.u_count.p, 0, w.u_count.cap));
    }
    TRY(w.u_cursor.ensure(L * 4));
    TRY(w.u_pair_off.ensure((L + 1) * 4));
    TRY(w.u_unit_prefix.ensure(tk_unit_prefix_ints((int64_t)L) * 4));
    TRY(w.u_pair_q.ensure(((size_t)nq * p.S + 4 * L) * 4));
    TRY(w.u_pair_f0.ensure(((size_t)nq * p.S + 4 * L) * 4));
    TRY(w.c_pair_off.ensure(8));
    TRY(w.c_unit_prefix.ensure(tk_unit_prefix_ints(1) * 4));
    TRY(w.c_pair_q.ensure(((size_t)nq + 4) * 4));
    TRY(w.c_pair_f0.ensure(((size_t)nq + 4) * 4));
    return TK_OK;
}

static int shard_args(tk_index *ix, int slot, int64_t nq, int64_t capacity, const Plan &p,
                      int64_t &qh)
{
    ARGCHECK(ix->sharded, "not a list-sharded index (tk_index_set_lists_shard)");
    ARGCHECK(slot >= 0 && slot < ix->depth, "slot must be < the pipeline depth");
    ARGCHECK(nq >= 1 && nq <= MAX_SHARD_BATCH, "1 <= nq <= 131072 per sharded batch");
    ARGCHECK(capacity >= 1 && capacity * ix->world < (1ll << 31), "capacity");
    qh = (nq + ix->world - 1) / ix->world;
    ARGCHECK((double)nq * p.S / 4 * ix->max_list_chunks + (double)ix->total_chunks < 2.0e9,
             "too many scan units for one batch");
    ARGCHECK((double)qh * p.cap * 17.0 < 16.0e9, "distance rows of the home queries exceed 16 GB");
    return TK_OK;
}

// Coarse stage sharded by HOME rank: tables for all nq queries (every rank scores segments of
// every query), coarse scan + replay + rescoring only for this rank's ceil(nq/world) home
// queries; the caller all-gathers the probe lists and hands them to tk_index_shard_scan_dev.
extern "C" int tk_index_shard_coarse_dev(tk_index *ix, int slot, const float *q_dev,
                                         const void *q_pq_dev, int q_pq_is_f64, int64_t nq, int k,
                                         int n_probes, int pass_1, int64_t *probes_home_dev,
                                         void *stream)
{
    IXLOCK(ix);
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    int64_t qh = 0;
    TRY(shard_args(ix, slot, nq, 1, p, qh));
    ARGCHECK(probes_home_dev, "probes buffer");
    Work &w = ix->works[(size_t)slot];
    hipStream_t st = (hipStream_t)stream;
    TRY(reserve_shard(ix, w, nq, qh, p));
    Prof pf;
    TRY(stage_tables(ix, w, q_pq_dev, q_pq_is_f64, nq, st, pf));
    const int64_t q0 = (int64_t)ix->rank * qh;
    int64_t nqh = nq - q0;
    nqh = nqh < 0 ? 0 : (nqh > qh ? qh : nqh);
    // rows past nq: list 0 (never read by a consumer; defined for the all-gather)
    HIPCHECK(hipMemsetAsync(probes_home_dev, 0, (size_t)qh * p.kc * 8, st));
    if (nqh > 0) {
        if (coarse_units(ix, nqh))     // identity pairs of the home range (stage_tables: of all nq)
            tk_launch_identity_pairs(nqh, (int)ix->center_chunks, w.c_pair_off.as<int>(),
                                     w.c_unit_prefix.as<int>(), w.c_pair_q.as<int>(),
                                     w.c_pair_f0.as<int>(), st);
        launch_coarse_scan(ix, w, nqh, p, st, w.tables.as<uint4>() + q0 * ix->M);
        TRY(coarse_replay_probes(ix, w, q_dev + q0 * ix->d, nqh, p, probes_home_dev, st, pf));
    }
    HIPCHECK(hipGetLastError());
    return TK_OK;
}

extern "C" int tk_index_shard_scan_dev(tk_index *ix, int slot, const float *q_dev,
                                       const void *q_pq_dev, int q_pq_is_f64, int64_t nq, int k,
                                       int n_probes, int pass_1, const int64_t *probes_all_dev,
                                       int64_t capacity, void *send_dev, int *flag_dev, void *stream)
{
    IXLOCK(ix);
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    int64_t qh = 0;
    TRY(shard_args(ix, slot, nq, capacity, p, qh));
    ARGCHECK(send_dev && flag_dev, "send/flag buffers");
    Work &w = ix->works[(size_t)slot];
    hipStream_t st = (hipStream_t)stream;
    TRY(reserve_shard(ix, w, nq, qh, p));
    TRY(w.smins.ensure((size_t)ix->world * capacity + 16));
    TRY(w.usage.ensure((size_t)ix->world * 2 * 8));
    Prof pf;
    const int *owner = ix->owner.as<int>();
    const int64_t *probes = probes_all_dev;
    if (probes) {
        // the tables of this slot were built by tk_index_shard_coarse_dev; the probe lists of
        // all queries arrive gathered from their home ranks
        coarse_slots(ix, w, probes, nq, p, w.u_count.as<int>(), owner, ix->rank, st);
    } else {
        // replicated coarse stage: every rank derives every probe list itself
        TRY(stage_tables(ix, w, q_pq_dev, q_pq_is_f64, nq, st, pf));
        launch_coarse_scan(ix, w, nq, p, st);
        TRY(stage_coarse_rest(ix, w, q_dev, nq, p, w.u_count.as<int>(), owner, ix->rank, st, pf));
        probes = w.probes.as<int64_t>();
    }
    {
        const int64_t n1 = nq * p.S + (int64_t)ix->world * qh * p.S + 1;
        ARGCHECK(n1 < (1ll << 31), "too many (query, list) entries for one sharded batch");
        TRY(w.pos_lens.ensure((size_t)n1 * 8));
        TRY(w.pos_off.ensure((size_t)n1 * 8));
        size_t tmp_bytes = 0;
        ARGCHECK(tk_scan_exclusive64(nullptr, &tmp_bytes, w.pos_lens.as<long long>(),
                                     w.pos_off.as<long long>(), n1, st) == 0,
                 "hipcub scan (size query) failed");
        TRY(w.scan_tmp.ensure(tmp_bytes + 16));
        if (tk_launch_shard_positions(probes, w.slot_prefix.as<int>(), p.S, nq, ix->n_lists, owner,
                                      ix->rank, ix->world, qh, capacity, w.spos.as<int>(),
                                      w.rpos.as<int>(), flag_dev, w.usage.as<long long>(),
                                      w.pos_lens.as<long long>(), w.pos_off.as<long long>(),
                                      w.scan_tmp.p, tmp_bytes, st))
            return fail(TK_ERR_HIP, "hipcub scan failed");
    }
    tk_launch_pairs_scan(w.u_count.as<int>(), ix->local_chunk_off.as<int64_t>(), ix->n_lists,
                         w.u_pair_off.as<int>(), w.u_unit_prefix.as<int>(), w.u_cursor.as<int>(),
                         w.u_pair_q.as<int>(), st);
    tk_launch_shard_pairs_fill(probes, p.S, nq, ix->n_lists, owner, ix->rank,
                               w.spos.as<int>(), w.u_pair_off.as<int>(), w.u_cursor.as<int>(),
                               w.u_pair_q.as<int>(), w.u_pair_f0.as<int>(), st);
    // the owned segments are scored straight into the send buffer (row stride 0, the
    // record's offset is the segment's position); the minima are rebuilt by the receiver
    tk_launch_scan_units(ix->codes.as<uint4>(), ix->M, w.tables.as<uint4>(), nq, p.S, ix->n_lists,
                         ix->local_chunk_off.as<int64_t>(), w.u_pair_off.as<int>(),
                         w.u_unit_prefix.as<int>(), w.u_pair_q.as<int>(), w.u_pair_f0.as<int>(),
                         (uint4 *)send_dev, 0, w.smins.as<uint8_t>(), 0, 1, ix->order, 768, st);
    w.shard_probes = probes;
    w.shard_nq = nq;
    w.shard_capacity = capacity;
    HIPCHECK(hipGetLastError());
    return TK_OK;
}

// ---- the same scan in two phases, the second on the matrix cores (plain_scan.hip) ----
// tk_index_shard_scan_dev scores every owned (query, list) segment with the exact kernel.  The
// plain kernel is 3 x faster, and exact for a query from the point where its heap's bound is at
// most the limit C of its table — and B1, the bound after the query's FIRST probed list, is a
// number every rank can know before it scans the rest: the owner of that list replays it by value
// (shard_first_bound_kernel, the filtered exchange's), a 1-byte MIN all-reduce spreads it.  So:
//   tk_index_shard_scan_first_dev   as _scan_dev up to the positions; exact scan of the FIRST
//                                   slots this rank owns into send_dev; bound_dev[nq] = B1 of the
//                                   queries whose first list it owns, 255 elsewhere
//   all-reduce(MIN, uint8)          by the caller
//   tk_index_shard_scan_rest_dev    the slots behind the first: of the queries with B1 <= C on the
//                                   plain kernel, of the others on the exact one — no query is
//                                   ever scanned twice and nothing has to be repaired afterwards:
//                                   the replay at home (tk_index_shard_finish_dev, or the filtered
//                                   exchange with this very bound) is the reference's by the lemma
// tk_index_shard_plain: does this apply to (k, n_probes, pass_1)?  Replicated state only (every
// rank answers alike): M <= 52, n_probes >= 2, distinct labels (or tk_index_set_plain_scan(ix, 2)),
// not switched off.  Otherwise callers use tk_index_shard_scan_dev.
static bool shard_plain_possible(const tk_index *ix, const Plan &p)
{
    if (ix->plain_mode == 1 || !plain_env_on() || !ix->sharded || p.S < 2 || !tk_plain_fits(ix->M)) return false;
    return ix->ids_unique || ix->plain_mode == 2;
}

extern "C" int tk_index_shard_plain(tk_index *ix, int k, int n_probes, int pass_1)
{
    IXLOCK(ix);
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    return shard_plain_possible(ix, p) ? 1 : 0;
}

static int reserve_shard_plain(tk_index *ix, Work &w, int64_t nq, const Plan &p)
{
    const size_t L = (size_t)ix->n_lists;
    TRY(w.qlim.ensure((size_t)nq * 4));
    TRY(w.plain_q.ensure((size_t)nq + 16));
    DevBuf *zeroed[] = {&w.p_count, &w.h_count};
    for (DevBuf *b : zeroed) {
        const void *before = b->p;
        TRY(b->ensure(L * 4));
        if (b->p != before) HIPCHECK(hipMemset(b->p, 0, b->cap));
    }
    TRY(w.p_cursor.ensure(L * 4));
    TRY(w.p_pair_off.ensure((L + 1) * 4));
    TRY(w.p_unit_prefix.ensure(tk_unit_prefix_ints((int64_t)L) * 4));
    TRY(w.p_pair_q.ensure(((size_t)nq * p.S + 4) * 4));
    TRY(w.p_pair_f0.ensure(((size_t)nq * p.S + 4) * 4));
    TRY(w.p_unit_desc.ensure(plain_desc_bytes(ix, nq, p)));
    TRY(w.h_cursor.ensure(L * 4));
    TRY(w.h_pair_off.ensure((L + 1) * 4));
    TRY(w.h_unit_prefix.ensure(tk_unit_prefix_ints((int64_t)L) * 4));
    TRY(w.h_pair_q.ensure((4 * L + 4) * 4));
    TRY(w.h_pair_f0.ensure((4 * L + 4) * 4));
    return TK_OK;
}

extern "C" int tk_index_shard_scan_first_dev(tk_index *ix, int slot, const float *q_dev,
                                             const void *q_pq_dev, int q_pq_is_f64, int64_t nq, int k,
                                             int n_probes, int pass_1, const int64_t *probes_all_dev,
                                             int64_t capacity, void *send_dev, int *flag_dev,
                                             uint8_t *bound_dev, void *stream)
{
    IXLOCK(ix);
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    int64_t qh = 0;
    TRY(shard_args(ix, slot, nq, capacity, p, qh));
    ARGCHECK(send_dev && flag_dev && bound_dev, "send/flag/bound buffers");
    ARGCHECK(shard_plain_possible(ix, p), "tk_index_shard_plain says no: use tk_index_shard_scan_dev");
    Work &w = ix->works[(size_t)slot];
    hipStream_t st = (hipStream_t)stream;
    TRY(reserve_shard(ix, w, nq, qh, p));
    TRY(reserve_shard_plain(ix, w, nq, p));
    TRY(w.smins.ensure((size_t)ix->world * capacity + 16));
    TRY(w.usage.ensure((size_t)ix->world * 2 * 8));
    Prof pf;
    const int *owner = ix->owner.as<int>();
    const int64_t *probes = probes_all_dev;
    if (probes) {
        coarse_slots(ix, w, probes, nq, p, nullptr, owner, ix->rank, st);
    } else {
        TRY(stage_tables(ix, w, q_pq_dev, q_pq_is_f64, nq, st, pf));
        launch_coarse_scan(ix, w, nq, p, st);
        TRY(stage_coarse_rest(ix, w, q_dev, nq, p, nullptr, owner, ix->rank, st, pf));
        probes = w.probes.as<int64_t>();
    }
    // the limits C of all nq tables (built by _shard_coarse_dev or just above)
    tk_launch_table_limits(w.tables.as<uint4>(), ix->M, ix->order, nq, w.qlim.as<int>(), st, ix->opt_plain_limit);
    {
        const int64_t n1 = nq * p.S + (int64_t)ix->world * qh * p.S + 1;
        ARGCHECK(n1 < (1ll << 31), "too many (query, list) entries for one sharded batch");
        TRY(w.pos_lens.ensure((size_t)n1 * 8));
        TRY(w.pos_off.ensure((size_t)n1 * 8));
        size_t tmp_bytes = 0;
        ARGCHECK(tk_scan_exclusive64(nullptr, &tmp_bytes, w.pos_lens.as<long long>(),
                                     w.pos_off.as<long long>(), n1, st) == 0,
                 "hipcub scan (size query) failed");
        TRY(w.scan_tmp.ensure(tmp_bytes + 16));
        if (tk_launch_shard_positions(probes, w.slot_prefix.as<int>(), p.S, nq, ix->n_lists, owner,
                                      ix->rank, ix->world, qh, capacity, w.spos.as<int>(),
                                      w.rpos.as<int>(), flag_dev, w.usage.as<long long>(),
                                      w.pos_lens.as<long long>(), w.pos_off.as<long long>(),
                                      w.scan_tmp.p, tmp_bytes, st))
            return fail(TK_ERR_HIP, "hipcub scan failed");
    }
    // first slots: exact, straight into the send buffer
    tk_launch_shard_count_first(probes, p.S, nq, ix->n_lists, owner, ix->rank, w.u_count.as<int>(), st);
    tk_launch_pairs_scan(w.u_count.as<int>(), ix->local_chunk_off.as<int64_t>(), ix->n_lists,
                         w.u_pair_off.as<int>(), w.u_unit_prefix.as<int>(), w.u_cursor.as<int>(),
                         w.u_pair_q.as<int>(), st);
    tk_launch_shard_pairs_fill(probes, p.S, nq, ix->n_lists, owner, ix->rank, w.spos.as<int>(),
                               w.u_pair_off.as<int>(), w.u_cursor.as<int>(), w.u_pair_q.as<int>(),
                               w.u_pair_f0.as<int>(), st, 0, 1);
    tk_launch_scan_units(ix->codes.as<uint4>(), ix->M, w.tables.as<uint4>(), nq, p.S, ix->n_lists,
                         ix->local_chunk_off.as<int64_t>(), w.u_pair_off.as<int>(),
                         w.u_unit_prefix.as<int>(), w.u_pair_q.as<int>(), w.u_pair_f0.as<int>(),
                         (uint4 *)send_dev, 0, w.smins.as<uint8_t>(), 0, 1, ix->order, 768, st);
    w.shard_probes = probes;
    w.shard_nq = nq;
    w.shard_capacity = capacity;
    w.shard_first = true;
    // B1 of the queries whose first list lies here (ivf.py:137-152 over that list alone, by value)
    tk_launch_shard_first_bound(probes, w.slot_prefix.as<int>(), w.slot_n.as<int>(), p.S, nq,
                                ix->n_lists, owner, ix->rank, w.spos.as<int>(), (const uint4 *)send_dev,
                                w.smins.as<uint8_t>(), p.R, bound_dev, st);
    HIPCHECK(hipGetLastError());
    return TK_OK;
}

extern "C" int tk_index_shard_scan_rest_dev(tk_index *ix, int slot, int64_t nq, int k, int n_probes,
                                            int pass_1, int64_t capacity, void *send_dev,
                                            const uint8_t *bound_dev, void *stream)
{
    IXLOCK(ix);
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    int64_t qh = 0;
    TRY(shard_args(ix, slot, nq, capacity, p, qh));
    ARGCHECK(send_dev && bound_dev, "send/bound buffers");
    Work &w = ix->works[(size_t)slot];
    ARGCHECK(w.shard_first && w.shard_probes && w.shard_nq == nq && w.shard_capacity == capacity,
             "tk_index_shard_scan_first_dev of this slot (same nq and capacity) comes first");
    w.shard_first = false;
    hipStream_t st = (hipStream_t)stream;
    const int *owner = ix->owner.as<int>();
    const int64_t *probes = w.shard_probes;
    // a segment that overflowed its region is scored by the plain kernel to the tail of the
    // buffer (the batch is repeated anyway): the longest list must fit there
    const int64_t tail = (int64_t)ix->world * capacity - ix->max_list_chunks;
    const int allow = tail >= 0 ? 1 : 0;
    tk_launch_shard_count_rest(probes, p.S, nq, ix->n_lists, owner, ix->rank, bound_dev,
                               w.qlim.as<int>(), allow, w.plain_q.as<uint8_t>(), w.u_count.as<int>(),
                               w.p_count.as<int>(), st);
    TkPairSet ex{w.u_count.as<int>(), w.u_cursor.as<int>(), w.u_pair_off.as<int>(),
                 w.u_unit_prefix.as<int>(), w.u_pair_q.as<int>(), w.u_pair_f0.as<int>()};
    TkPairSet pl{w.p_count.as<int>(), w.p_cursor.as<int>(), w.p_pair_off.as<int>(),
                 w.p_unit_prefix.as<int>(), w.p_pair_q.as<int>(), w.p_pair_f0.as<int>(),
                 w.p_unit_desc.as<int>(), plain_k(ix, nq, p)};
    TkPairSet hd{w.h_count.as<int>(), w.h_cursor.as<int>(), w.h_pair_off.as<int>(),
                 w.h_unit_prefix.as<int>(), w.h_pair_q.as<int>(), w.h_pair_f0.as<int>()};   // (stays empty)
    tk_launch_pairs_scan3(ex, pl, hd, ix->local_chunk_off.as<int64_t>(), ix->n_lists, 0, st);
    tk_launch_plain_desc(pl, ix->local_chunk_off.as<int64_t>(), ix->n_lists, st);
    tk_launch_shard_pairs_fill(probes, p.S, nq, ix->n_lists, owner, ix->rank, w.spos.as<int>(),
                               w.u_pair_off.as<int>(), w.u_cursor.as<int>(), w.u_pair_q.as<int>(),
                               w.u_pair_f0.as<int>(), st, 1, p.S, w.plain_q.as<uint8_t>(), 0);
    tk_launch_shard_pairs_fill(probes, p.S, nq, ix->n_lists, owner, ix->rank, w.spos.as<int>(),
                               w.p_pair_off.as<int>(), w.p_cursor.as<int>(), w.p_pair_q.as<int>(),
                               w.p_pair_f0.as<int>(), st, 1, p.S, w.plain_q.as<uint8_t>(), 1,
                               (int)(tail > 0 ? tail : 0));
    TkScanJob pj;
    pj.codes = ix->codes.as<uint4>();
    pj.tables = w.tables.as<uint4>();
    pj.list_chunk_off = ix->local_chunk_off.as<int64_t>();
    pj.n_lists = (int)ix->n_lists;
    pj.unit_prefix = w.p_unit_prefix.as<int>();
    pj.pair_off = w.p_pair_off.as<int>();
    pj.pair_q = w.p_pair_q.as<int>();
    pj.pair_f0 = w.p_pair_f0.as<int>();
    pj.unit_desc4 = w.p_unit_desc.as<int>();
    pj.dist = (uint4 *)send_dev;
    pj.cap = 0;
    pj.mins = w.smins.as<uint8_t>();
    pj.min_stride = 0;
    if (tk_launch_scan_plain(pj, ix->M, ix->order, plain_blocks(), st))
        return fail(TK_ERR_HIP, "scan_plain_kernel: LDS attribute / unsupported M");
    tk_launch_scan_units(ix->codes.as<uint4>(), ix->M, w.tables.as<uint4>(), nq, p.S, ix->n_lists,
                         ix->local_chunk_off.as<int64_t>(), w.u_pair_off.as<int>(),
                         w.u_unit_prefix.as<int>(), w.u_pair_q.as<int>(), w.u_pair_f0.as<int>(),
                         (uint4 *)send_dev, 0, w.smins.as<uint8_t>(), 0, 1, ix->order, 768, st);
    HIPCHECK(hipGetLastError());
    return TK_OK;
}

// Books of the slot's last tk_index_shard_scan_rest_dev (synchronises): out4 = (query, list) pairs
// this rank scored on the plain kernel, tiles of 32 of them, pair records of the exact kernel for
// the slots behind the first (padded to groups of 4), queries (of all nq) that went the plain way.
extern "C" int tk_index_shard_plain_stats(tk_index *ix, int slot, int64_t *out4)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->sharded && out4, "sharded index and an output buffer");
    ARGCHECK(slot >= 0 && slot < ix->depth, "slot must be < the pipeline depth");
    Work &w = ix->works[(size_t)slot];
    for (int i = 0; i < 4; i++) out4[i] = 0;
    if (!w.p_pair_off.p || !w.plain_q.p || !w.shard_probes || w.shard_first) return TK_OK;
    HIPCHECK(hipDeviceSynchronize());
    const int64_t L = ix->n_lists;
    int v[3] = {0, 0, 0};
    HIPCHECK(hipMemcpy(&v[0], w.p_pair_off.as<int>() + L, 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(&v[1], w.p_unit_prefix.as<int>() + L, 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(&v[2], w.u_pair_off.as<int>() + L, 4, hipMemcpyDeviceToHost));
    std::vector<uint8_t> pq((size_t)w.shard_nq);
    HIPCHECK(hipMemcpy(pq.data(), w.plain_q.p, pq.size(), hipMemcpyDeviceToHost));
    int64_t n = 0;
    for (uint8_t b : pq) n += b != 0;
    int64_t tiles = 0;          // (v[1] units = (tile, range of chunk pairs): a tile's first range starts at 0)
    if (v[1] > 0) {
        std::vector<int> desc((size_t)v[1] * 4);
        HIPCHECK(hipMemcpy(desc.data(), w.p_unit_desc.p, desc.size() * 4, hipMemcpyDeviceToHost));
        for (int u = 0; u < v[1]; u++) tiles += desc[(size_t)4 * u + 2] == 0;
    }
    out4[0] = v[0]; out4[1] = tiles; out4[2] = v[2]; out4[3] = n;
    return TK_OK;
}

// Longest (source -> home) stream of the slot's last tk_index_shard_scan_dev, in uint4, whether
// it fitted the regions or not: what a caller sizes `capacity` by (max-reduce it over the ranks).
// Synchronises with the device.
extern "C" int tk_index_shard_usage(tk_index *ix, int slot, int64_t *max_stream_uint4)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->sharded && max_stream_uint4, "sharded index and an output pointer");
    ARGCHECK(slot >= 0 && slot < ix->depth, "slot must be < the pipeline depth");
    Work &w = ix->works[(size_t)slot];
    ARGCHECK(w.shard_probes && w.usage.p, "tk_index_shard_scan_dev of this slot comes first");
    std::vector<long long> u((size_t)ix->world * 2);
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(u.data(), w.usage.p, u.size() * 8, hipMemcpyDeviceToHost));
    long long m = 0;
    for (long long x : u) m = x > m ? x : m;
    *max_stream_uint4 = m;
    return TK_OK;
}

// (signed tables only, as everything behind IVF.query: ivf.py:128 builds distance_table, never
// udistance_table; the kernels' 0x7f fill and int8 minima assume it)
// ---- filtered exchange: bound -> [min all-reduce] -> filter ->
// [all-to-all of the counts and of the records] -> finish_filtered.  `scan_dev` is the buffer
// tk_index_shard_scan_dev of the same slot filled (it stays on the rank), with the same nq,
// k, n_probes, pass_1 and capacity; the probe lists handed to that call must still be alive.
static int filtered_args(tk_index *ix, Work &w, int64_t nq, int64_t capacity)
{
    ARGCHECK(w.shard_probes && w.shard_nq == nq && w.shard_capacity == capacity,
             "tk_index_shard_scan_dev of this slot (same nq and capacity) comes first");
    return TK_OK;
}

extern "C" int tk_index_shard_bound_dev(tk_index *ix, int slot, int64_t nq, int k, int n_probes,
                                        int pass_1, int64_t capacity, const void *scan_dev,
                                        uint8_t *bound_dev, void *stream)
{
    IXLOCK(ix);
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    int64_t qh = 0;
    TRY(shard_args(ix, slot, nq, capacity, p, qh));
    ARGCHECK(scan_dev && bound_dev, "scan/bound buffers");
    Work &w = ix->works[(size_t)slot];
    TRY(filtered_args(ix, w, nq, capacity));
    tk_launch_shard_first_bound(w.shard_probes, w.slot_prefix.as<int>(), w.slot_n.as<int>(), p.S, nq,
                                ix->n_lists, ix->owner.as<int>(), ix->rank, w.spos.as<int>(),
                                (const uint4 *)scan_dev, w.smins.as<uint8_t>(), p.R, bound_dev,
                                (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
    return TK_OK;
}

static int shard_filter_impl(tk_index *ix, int slot, int64_t nq, int k, int n_probes,
                             int pass_1, int64_t capacity, const void *scan_dev,
                             const uint8_t *bound_dev, int32_t *counts_dev,
                             int32_t *records_dev, int64_t region, int *flag_dev, int64_t *acc_dev,
                             void *stream)
{
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    int64_t qh = 0;
    TRY(shard_args(ix, slot, nq, capacity, p, qh));
    ARGCHECK(scan_dev && bound_dev && counts_dev && records_dev, "scan/bound/counts/records buffers");
    Work &w = ix->works[(size_t)slot];
    TRY(filtered_args(ix, w, nq, capacity));
    hipStream_t st = (hipStream_t)stream;
    const int64_t np1 = nq * p.S + 1;
    ARGCHECK(np1 < (1ll << 31), "too many (query, list) pairs");
    TRY(w.pair_cnt.ensure((size_t)np1 * 4));
    TRY(w.pair_off.ensure((size_t)np1 * 4));
    size_t tmp_bytes = 0;
    ARGCHECK(tk_scan_exclusive(nullptr, &tmp_bytes, w.pair_cnt.as<int>(), w.pair_off.as<int>(), np1,
                               st) == 0, "hipcub scan (size query) failed");
    TRY(w.scan_tmp.ensure(tmp_bytes + 16));
    TRY(w.tally.ensure((size_t)ix->world * 256 * 4));
    HIPCHECK(hipMemsetAsync(w.tally.p, 0, (size_t)ix->world * 256 * 4, st));
    HIPCHECK(hipMemsetAsync(counts_dev, 0, (size_t)ix->world * 3 * 4, st));
    HIPCHECK(hipMemsetAsync(w.pair_cnt.as<int>() + (np1 - 1), 0, 4, st));
    if (tk_launch_shard_filter(w.shard_probes, w.slot_prefix.as<int>(), p.S, nq, ix->n_lists,
                               ix->owner.as<int>(), ix->rank, ix->world, qh, p.cap,
                               w.spos.as<int>(), (const uint4 *)scan_dev, w.smins.as<uint8_t>(),
                               bound_dev, w.pair_cnt.as<int>(), w.pair_off.as<int>(), w.scan_tmp.p,
                               tmp_bytes, w.tally.as<int>(), counts_dev, records_dev, st, (int)region,
                               flag_dev, (long long *)acc_dev))
        return fail(TK_ERR_HIP, "hipcub scan failed");
    HIPCHECK(hipGetLastError());
    return TK_OK;
}

extern "C" int tk_index_shard_filter_dev(tk_index *ix, int slot, int64_t nq, int k, int n_probes,
                                         int pass_1, int64_t capacity, const void *scan_dev,
                                         const uint8_t *bound_dev, int32_t *counts_dev,
                                         int32_t *records_dev, void *stream)
{
    IXLOCK(ix);
    return shard_filter_impl(ix, slot, nq, k, n_probes, pass_1, capacity, scan_dev, bound_dev, counts_dev,
                             records_dev, 0, nullptr, nullptr, stream);
}

// The same with the records of home rank h at records_dev[h * region_records ...] (room for world *
// region_records records): the all-to-all that follows has EQUAL splits, so no rank has to read
// a count on the host before it can enqueue it — counts_dev[0, world) travel beside the records and
// the home rank reads them on the device (tk_index_shard_finish_regions_dev).  More than
// region_records records for one home rank: the rest is dropped and *flag_dev |= 1, the overflow
// flag of the batch (the caller repeats it with larger regions, as with `capacity`).
// acc_dev (or NULL): three int64 the caller keeps across batches — [0] = largest counts_dev[h] seen
// (atomic max: what the regions have to hold), [1] += records, [2] += blocks scored.
extern "C" int tk_index_shard_filter_regions_dev(tk_index *ix, int slot, int64_t nq, int k, int n_probes,
                                                 int pass_1, int64_t capacity, const void *scan_dev,
                                                 const uint8_t *bound_dev, int32_t *counts_dev,
                                                 int32_t *records_dev, int64_t region_records,
                                                 int *flag_dev, int64_t *acc_dev, void *stream)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->sharded, "not a list-sharded index");
    ARGCHECK(region_records >= 1 && region_records * ix->world < (1ll << 31) && flag_dev,
             "region_records (x world must stay below 2^31) / flag buffer");
    return shard_filter_impl(ix, slot, nq, k, n_probes, pass_1, capacity, scan_dev, bound_dev, counts_dev,
                             records_dev, region_records, flag_dev, acc_dev, stream);
}

static int shard_finish_filtered_impl(tk_index *ix, int slot, const float *q_dev,
                                      int64_t nq, int k, int n_probes, int pass_1,
                                      const int32_t *records_dev, int64_t n_records,
                                      const int32_t *counts_recv_dev, int64_t region,
                                      int64_t *out_ids_home_dev, int *flag_dev, void *stream)
{
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    int64_t qh = 0;
    TRY(shard_args(ix, slot, nq, 1, p, qh));
    ARGCHECK(out_ids_home_dev && flag_dev && n_records >= 0 && (records_dev || n_records == 0),
             "records/out/flag buffers");
    Work &w = ix->works[(size_t)slot];
    ARGCHECK(w.shard_probes && w.shard_nq == nq, "tk_index_shard_scan_dev of this slot comes first");
    hipStream_t st = (hipStream_t)stream;
    const int64_t q0 = (int64_t)ix->rank * qh;
    int64_t nqh = nq - q0;
    nqh = nqh < 0 ? 0 : (nqh > qh ? qh : nqh);
    HIPCHECK(hipMemsetAsync(out_ids_home_dev, 0xff, (size_t)qh * k * 8, st));   // -1 rows
    if (nqh > 0) {
        tk_launch_shard_expand(records_dev, n_records, w.slot_prefix.as<int>() + q0 * (p.S + 1), p.S,
                               nqh, w.dist.as<uint4>(), p.cap, w.mins.as<uint8_t>(), p.cap_min,
                               flag_dev, st, counts_recv_dev, (int)region);
        Prof pf;
        TRY(stage_back(ix, w, q_dev + q0 * ix->d, q0, nqh, k, p, out_ids_home_dev, st, pf));
    }
    HIPCHECK(hipGetLastError());
    return TK_OK;
}

extern "C" int tk_index_shard_finish_filtered_dev(tk_index *ix, int slot, const float *q_dev,
                                                  int64_t nq, int k, int n_probes, int pass_1,
                                                  const int32_t *records_dev, int64_t n_records,
                                                  int64_t *out_ids_home_dev, int *flag_dev,
                                                  void *stream)
{
    IXLOCK(ix);
    return shard_finish_filtered_impl(ix, slot, q_dev, nq, k, n_probes, pass_1, records_dev, n_records,
                                      nullptr, 0, out_ids_home_dev, flag_dev, stream);
}

// records_dev: world regions of region_records records as the equal-split all-to-all delivered
// them (region s from source rank s); counts_recv_dev[s] of them are real (the all-to-all of the
// senders' counts_dev[0, world), on the device: no host synchronisation anywhere in the batch)
extern "C" int tk_index_shard_finish_regions_dev(tk_index *ix, int slot, const float *q_dev,
                                                 int64_t nq, int k, int n_probes, int pass_1,
                                                 const int32_t *records_dev,
                                                 const int32_t *counts_recv_dev,
                                                 int64_t region_records, int64_t *out_ids_home_dev,
                                                 int *flag_dev, void *stream)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->sharded, "not a list-sharded index");
    ARGCHECK(counts_recv_dev && region_records >= 1 && region_records * ix->world < (1ll << 31),
             "counts / region_records");
    return shard_finish_filtered_impl(ix, slot, q_dev, nq, k, n_probes, pass_1, records_dev,
                                      region_records * ix->world, counts_recv_dev, region_records,
                                      out_ids_home_dev, flag_dev, stream);
}

extern "C" int tk_index_shard_finish_dev(tk_index *ix, int slot, const float *q_dev, int64_t nq,
                                         int k, int n_probes, int pass_1, int64_t capacity,
                                         const void *recv_dev, int64_t *out_ids_home_dev,
                                         void *stream)
{
    IXLOCK(ix);
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    int64_t qh = 0;
    TRY(shard_args(ix, slot, nq, capacity, p, qh));
    ARGCHECK(recv_dev && out_ids_home_dev, "recv/out buffers");
    Work &w = ix->works[(size_t)slot];
    hipStream_t st = (hipStream_t)stream;
    const int64_t q0 = (int64_t)ix->rank * qh;
    int64_t nqh = nq - q0;
    nqh = nqh < 0 ? 0 : (nqh > qh ? qh : nqh);
    HIPCHECK(hipMemsetAsync(out_ids_home_dev, 0xff, (size_t)qh * k * 8, st));   // -1 rows
    if (nqh > 0) {
        tk_launch_shard_unpack((const uint4 *)recv_dev, w.rpos.as<int>(),
                               w.slot_prefix.as<int>() + q0 * (p.S + 1), p.S, nqh,
                               w.dist.as<uint4>(), p.cap, w.mins.as<uint8_t>(), p.cap_min, 1, st);
        Prof pf;
        TRY(stage_back(ix, w, q_dev + q0 * ix->d, q0, nqh, k, p, out_ids_home_dev, st, pf));
    }
    HIPCHECK(hipGetLastError());
    return TK_OK;
}

// GB/s of a kernel that only reads `bytes` of HBM (measurement plumbing for bench.py)
extern "C" int tk_measure_read_bandwidth(int64_t bytes, int reps, double *gbps)
{
    TRY(require_gpu());
    ARGCHECK(bytes >= (1 << 20) && reps >= 1 && gbps, "bytes >= 1 MiB, reps >= 1");
    DevBuf buf, out;
    int rc = buf.ensure((size_t)bytes);
    if (rc == TK_OK) rc = out.ensure(16);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (rc == TK_OK) {
        hipError_t e = hipMemset(buf.p, 1, (size_t)bytes);
        if (e == hipSuccess) e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        if (e == hipSuccess) {
            tk_launch_read_only(buf.p, bytes / 16, out.as<uint32_t>(), nullptr);
            e = hipDeviceSynchronize();
        }
        if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
        for (int r = 0; r < reps && e == hipSuccess; r++) tk_launch_read_only(buf.p, bytes / 16, out.as<uint32_t>(), nullptr);
        if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && ms > 0) *gbps = (double)bytes * reps / (ms * 1e-3) / 1e9;
        else rc = fail(TK_ERR_HIP, e == hipSuccess ? "zero elapsed time" : hipGetErrorString(e));
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    buf.release();
    out.release();
    return rc;
}

// ---------------------------------------------------------------------------
// device-resident build (devbuild.hip): IVF.build for vectors that live in HBM
static int upload_rotation(tk_index *ix, const double *R, int d_pad)
{
    std::vector<double> rt((size_t)d_pad * ix->dq);
    for (int j = 0; j < ix->dq; j++)
        for (int t = 0; t < d_pad; t++) rt[(size_t)t * ix->dq + j] = R[(size_t)j * d_pad + t];
    TRY(ix->rot_t.ensure(rt.size() * 8));
    HIPCHECK(hipMemcpy(ix->rot_t.p, rt.data(), rt.size() * 8, hipMemcpyHostToDevice));
    ix->rot_d_pad = d_pad;
    return TK_OK;
}

extern "C" float *tk_index_alloc_data(tk_index *ix, int64_t N, int d)
{
    IXLOCK(ix);
    if (!ix || !ix->have_pq || N < 1 || d < 1) {
        fail(TK_ERR_ARG, "bad argument: tk_index_alloc_data (set_pq first, N >= 1, d >= 1)");
        return nullptr;
    }
    if (ix->data.ensure((size_t)N * d * 4) != TK_OK) return nullptr;
    ix->N = N;
    ix->d = d;
    ix->data_is_f64 = 0;
    ix->have_data = ix->have_centers = ix->have_lists = false;   // until tk_index_build_dev
    return ix->data.as<float>();
}

static int synth_centres(DevBuf &buf, const float *centres, int n_centres, int d, const float **dev)
{
    *dev = nullptr;
    if (!centres || n_centres <= 0) return TK_OK;
    TRY(buf.ensure((size_t)n_centres * d * 4));
    HIPCHECK(hipMemcpy(buf.p, centres, (size_t)n_centres * d * 4, hipMemcpyHostToDevice));
    *dev = buf.as<float>();
    return TK_OK;
}

extern "C" int tk_index_synth_data(tk_index *ix, int64_t row0, int64_t n, uint64_t seed,
                                   const float *centres, int n_centres, float sigma)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->data.p && ix->N > 0, "tk_index_alloc_data first");
    ARGCHECK(row0 >= 0 && n >= 0 && row0 + n <= ix->N, "row range");
    const float *cd = nullptr;
    TRY(synth_centres(ix->stage, centres, n_centres, ix->d, &cd));
    tk_launch_synth_rows(ix->data.as<float>() + row0 * ix->d, row0, n, ix->d, seed, cd, n_centres, sigma, 0);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipDeviceSynchronize());
    return TK_OK;
}

// the same generator into host memory (queries, training samples)
extern "C" int tk_synth_rows(float *out, int64_t row0, int64_t n, int d, uint64_t seed,
                             const float *centres, int n_centres, float sigma)
{
    TRY(require_gpu());
    ARGCHECK(out && n >= 0 && d >= 1 && row0 >= 0, "buffers / sizes");
    DevBuf cb, xb;
    const float *cd = nullptr;
    int rc = synth_centres(cb, centres, n_centres, d, &cd);
    const int64_t slab = 1 << 20;
    for (int64_t o = 0; o < n && rc == TK_OK; o += slab) {
        const int64_t m = n - o < slab ? n - o : slab;
        if ((rc = xb.ensure((size_t)m * d * 4)) != TK_OK) break;
        tk_launch_synth_rows(xb.as<float>(), row0 + o, m, d, seed, cd, n_centres, sigma, 0);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpy(out + (size_t)o * d, xb.p, (size_t)m * d * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(TK_ERR_HIP, hipGetErrorString(e));
    }
    cb.release();
    xb.release();
    return rc;
}

// labels (m, M) of `m` float32 rows (m, d) on the device: pad1 / rotation (float64 FMA chain,
// as the device front end) into `rows`, then the nearest centroid per block
static int encode_rows_dev(tk_index *ix, const float *x, int64_t m, DevBuf &rows, uint8_t *labels)
{
    const bool rot = ix->rot_d_pad > 0;
    TRY(rows.ensure((size_t)m * ix->dq * (rot ? 8 : 4)));
    tk_launch_prepare_queries(x, m, ix->d, rot ? ix->rot_t.as<double>() : nullptr, ix->dq,
                              rot ? ix->rot_d_pad : ix->dq, rows.p, 0);
    if (tk_launch_encode_pq(ix->pq_centers.as<float>(), ix->dq, ix->dpb, rows.p, rot ? 1 : 0, m, labels, 0))
        return fail(TK_ERR_HIP, "encode_pq_kernel: LDS budget / attribute");
    HIPCHECK(hipGetLastError());
    return TK_OK;
}

extern "C" int tk_index_build_dev(tk_index *ix, int normalise, const float *all_centers,
                                  const float *search_centers, const float *ynorm2, int64_t C,
                                  int n_probes, const double *R, int d_pad, int64_t *n_active_out)
{
    IXLOCK(ix);
    if (!search_centers) search_centers = all_centers;
    ARGCHECK(n_probes >= 1 && n_probes <= 9 && n_probes <= C, "n_probes must be 1 .. 9");
    const int kp = n_probes;
    ARGCHECK(ix && ix->have_pq && ix->data.p && ix->N > 0, "set_pq and tk_index_alloc_data first");
    ARGCHECK(all_centers && ynorm2 && C >= 1 && C < (1ll << 31), "centres");
    ARGCHECK(ix->N * kp < (1ll << 31), "N * n_probes < 2^31");
    ARGCHECK(ix->d <= 384 && (!normalise || ix->d <= 128), "d <= 384 (128 with normalisation)");
    ARGCHECK(16 % ix->dpb == 0, "dims_per_block must divide 16 for the device encoder");
    ARGCHECK(R ? (d_pad >= ix->d && d_pad <= 16384) : ix->dq >= ix->d, "rotation / padding");
    TRY(flush_pending(ix));
    const int64_t N = ix->N;
    const int d = ix->d, M = ix->M;
    float *X = ix->data.as<float>();
    if (R) TRY(upload_rotation(ix, R, d_pad));
    else { ix->rot_t.release(); ix->rot_d_pad = 0; }
    const int64_t slab = 1 << 20;
    DevBuf yt, yn, near, keys, rows, keys2, rows2, count, remap, labels, rot, tmp, zero, crow, clab;
    struct Cleanup {
        std::vector<DevBuf *> v;
        ~Cleanup() { for (DevBuf *b : v) b->release(); }
    } cl{{&yt, &yn, &near, &keys, &rows, &keys2, &rows2, &count, &remap, &labels, &rot, &tmp, &zero, &crow, &clab}};
    // ---- 1. data = X / |X| (ivf.py:78-79), nearest centre per row (ivf.py:85)
    {
        std::vector<float> ytv((size_t)C * d);
        for (int64_t j = 0; j < C; j++)
            for (int t = 0; t < d; t++) ytv[(size_t)t * C + j] = search_centers[(size_t)j * d + t];
        TRY(yt.ensure(ytv.size() * 4));
        TRY(yn.ensure((size_t)C * 4));
        HIPCHECK(hipMemcpy(yt.p, ytv.data(), ytv.size() * 4, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(yn.p, ynorm2, (size_t)C * 4, hipMemcpyHostToDevice));
    }
    const int64_t T = N * kp;           // (list, row) pairs: every row sits in kp lists
    TRY(near.ensure((size_t)slab * kp * 8));
    TRY(keys.ensure((size_t)T * 4));
    TRY(rows.ensure((size_t)T * 4));
    TRY(count.ensure((size_t)C * 4));
    HIPCHECK(hipMemset(count.p, 0, (size_t)C * 4));
    for (int64_t o = 0; o < N; o += slab) {
        const int64_t m = N - o < slab ? N - o : slab;
        if (normalise) tk_launch_normalise_rows(X + o * d, m, d, X + o * d, 0);
        tk_launch_assign(X + o * d, m, d, yt.p, yn.p, 0, (int)C, kp, near.as<int64_t>(), 0);
        tk_launch_keys_count(near.as<int64_t>(), m, kp, o, N, keys.as<int>(), rows.as<int>(), count.as<int>(), 0);
        HIPCHECK(hipGetLastError());
    }
    HIPCHECK(hipDeviceSynchronize());
    // ---- 2. active centres (ivf.py:91: all_centers[np.unique(nearest)]) and the CSR offsets
    std::vector<int> cnt((size_t)C), rm((size_t)C, -1);
    HIPCHECK(hipMemcpy(cnt.data(), count.p, (size_t)C * 4, hipMemcpyDeviceToHost));
    std::vector<float> act;
    std::vector<int64_t> sizes;
    for (int64_t j = 0; j < C; j++)
        if (cnt[(size_t)j] > 0) {
            rm[(size_t)j] = (int)sizes.size();
            sizes.push_back(cnt[(size_t)j]);
            act.insert(act.end(), all_centers + (size_t)j * d, all_centers + (size_t)(j + 1) * d);
        }
    const int64_t L = (int64_t)sizes.size();
    {   // the reference groups the rows by RAW centre id into n_active lists and asserts
        // max(index) < n_active (utils.py:128, IVF.build -> group_data_by_indices): it only builds
        // when no empty centre precedes a used one.  Same contract here (the host build asserts too).
        int64_t last = -1;
        for (int64_t j = 0; j < C; j++)
            if (cnt[(size_t)j] > 0) last = j;
        ARGCHECK(last < L, "a centre that received no row precedes one that did: the reference's "
                           "group_data_by_indices asserts max(index) < n_active (utils.py:128)");
    }
    std::vector<int64_t> coff((size_t)L + 1, 0), ioff((size_t)L + 1, 0);
    int64_t maxc = 0;
    for (int64_t i = 0; i < L; i++) {
        const int64_t c = (sizes[(size_t)i] + 15) / 16;
        coff[(size_t)i + 1] = coff[(size_t)i] + c;
        ioff[(size_t)i + 1] = ioff[(size_t)i] + sizes[(size_t)i];
        if (c > maxc) maxc = c;
    }
    ARGCHECK(maxc < (1ll << 26), "list too long");
    TRY(remap.ensure((size_t)C * 4));
    HIPCHECK(hipMemcpy(remap.p, rm.data(), (size_t)C * 4, hipMemcpyHostToDevice));
    tk_launch_remap_keys(keys.as<int>(), T, remap.as<int>(), 0);
    // ---- 3. rows grouped by list: stable sort of (list, row); with two lists per row the
    //         column-0 pairs precede the column-1 pairs of every list (utils.py:131-150)
    int bits = 1;
    while ((1ll << bits) < L) bits++;
    TRY(keys2.ensure((size_t)T * 4));
    TRY(rows2.ensure((size_t)T * 4));
    size_t tmp_bytes = 0;
    if (tk_sort_pairs(nullptr, &tmp_bytes, keys.as<int>(), keys2.as<int>(), rows.as<int>(), rows2.as<int>(), T, bits, 0))
        return fail(TK_ERR_HIP, "radix sort: size query failed");
    TRY(tmp.ensure(tmp_bytes > 0 ? tmp_bytes : 16));
    if (tk_sort_pairs(tmp.p, &tmp_bytes, keys.as<int>(), keys2.as<int>(), rows.as<int>(), rows2.as<int>(), T, bits, 0))
        return fail(TK_ERR_HIP, "radix sort failed");
    HIPCHECK(hipDeviceSynchronize());
    keys.release(); rows.release(); keys2.release(); tmp.release(); near.release();
    // ---- 4. PQ codes of every row (a row's code does not depend on its list), of the zero
    //         vector (list padding, fast_pq.py:165) and of the active centres (ivf.py:92-96)
    TRY(labels.ensure((size_t)N * M));
    for (int64_t o = 0; o < N; o += slab) {
        const int64_t m = N - o < slab ? N - o : slab;
        TRY(encode_rows_dev(ix, X + o * d, m, rot, labels.as<uint8_t>() + (size_t)o * M));
    }
    const int64_t L16 = (L + 15) / 16 * 16;
    TRY(crow.ensure((size_t)(L16 + 16) * d * 4));
    TRY(clab.ensure((size_t)(L16 + 16) * M));
    HIPCHECK(hipMemset(crow.p, 0, (size_t)(L16 + 16) * d * 4));
    HIPCHECK(hipMemcpy(crow.p, act.data(), (size_t)L * d * 4, hipMemcpyHostToDevice));
    TRY(encode_rows_dev(ix, crow.as<float>(), L16 + 16, rot, clab.as<uint8_t>()));
    const uint8_t *zero_code = clab.as<uint8_t>() + (size_t)L16 * M;     // code of a zero row
    // ---- 5. the index: centres
    TRY(ix->active_centers.ensure((size_t)L * d * 4));
    HIPCHECK(hipMemcpy(ix->active_centers.p, act.data(), (size_t)L * d * 4, hipMemcpyHostToDevice));
    const int64_t center_chunks = L16 / 16;
    const int P = M / 2;
    TRY(ix->center_codes.ensure((size_t)tk_tiled_uint4s(center_chunks, P) * 16));
    HIPCHECK(hipMemset(ix->center_codes.p, 0, (size_t)tk_tiled_uint4s(center_chunks, P) * 16));
    int64_t cco[2] = {0, center_chunks};
    int64_t cio[2] = {0, L};
    int64_t cn[1] = {L};
    TRY(ix->c_chunk_off.ensure(sizeof cco));
    HIPCHECK(hipMemcpy(ix->c_chunk_off.p, cco, sizeof cco, hipMemcpyHostToDevice));
    TRY(ix->stage.ensure(64));
    HIPCHECK(hipMemcpy(ix->stage.p, cio, sizeof cio, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy((char *)ix->stage.p + 32, cn, sizeof cn, hipMemcpyHostToDevice));
    tk_launch_pack_lists(clab.as<uint8_t>(), M, nullptr, ix->stage.as<int64_t>(), ix->c_chunk_off.as<int64_t>(),
                         (const int64_t *)((char *)ix->stage.p + 32), 1, zero_code,
                         ix->center_codes.as<uint4>(), center_chunks, 0);
    ix->n_lists = L; ix->center_chunks = center_chunks;
    int ci[3] = {0, (int)center_chunks, (int)L};
    int64_t cl1[1] = {-1};
    TRY(ix->cslots_i.ensure(sizeof ci));
    TRY(ix->cslots_l.ensure(sizeof cl1));
    HIPCHECK(hipMemcpy(ix->cslots_i.p, ci, sizeof ci, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ix->cslots_l.p, cl1, sizeof cl1, hipMemcpyHostToDevice));
    // ---- 6. the index: lists
    TRY(ix->list_chunk_off.ensure((size_t)(L + 1) * 8));
    TRY(ix->ids_off.ensure((size_t)(L + 1) * 8));
    TRY(ix->list_n.ensure((size_t)L * 8));
    TRY(ix->ids.ensure((size_t)T * 8));
    HIPCHECK(hipMemcpy(ix->list_chunk_off.p, coff.data(), (size_t)(L + 1) * 8, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ix->ids_off.p, ioff.data(), (size_t)(L + 1) * 8, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ix->list_n.p, sizes.data(), (size_t)L * 8, hipMemcpyHostToDevice));
    const size_t tiled_bytes = (size_t)tk_tiled_uint4s(coff[(size_t)L], P) * 16;
    TRY(ix->codes.ensure(tiled_bytes));
    HIPCHECK(hipMemset(ix->codes.p, 0, tiled_bytes));
    tk_launch_pack_lists(labels.as<uint8_t>(), M, rows2.as<int>(), ix->ids_off.as<int64_t>(),
                         ix->list_chunk_off.as<int64_t>(), ix->list_n.as<int64_t>(), (int)L, zero_code,
                         ix->codes.as<uint4>(), coff[(size_t)L], 0);
    tk_launch_widen_ids(rows2.as<int>(), T, ix->ids.as<int64_t>(), 0);
    HIPCHECK(hipGetLastError());
    ix->ids_unique = kp == 1;       // one list per row: no label can repeat
    ix->have_ids32 = false;
    if (kp > 1) {                   // the lane replay's duplicate test reads the labels as int32
        TRY(ix->ids32.ensure((size_t)T * 4));
        HIPCHECK(hipMemcpyAsync(ix->ids32.p, rows2.p, (size_t)T * 4, hipMemcpyDeviceToDevice, 0));
        ix->have_ids32 = true;
    }
    HIPCHECK(hipDeviceSynchronize());
    ix->sharded = false; ix->rank = 0; ix->world = 1;
    ix->total_chunks = coff[(size_t)L];
    ix->total_ids = T;
    ix->max_list_chunks = (int)maxc;
    ix->have_centers = ix->have_lists = ix->have_data = true;
    if (n_active_out) *n_active_out = L;
    return TK_OK;
}

// A complete unsharded index (tk_index_build_dev, or the host upload) becomes this rank's shard
// of a list-sharded index IN PLACE: the codes of the lists with owner[l] == rank are compacted
// into the rank's own array, everything else (centres, ids, vectors) stays replicated.
extern "C" int tk_index_shard_resident(tk_index *ix, const int32_t *owner, int rank, int world)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->have_lists && !ix->sharded, "a complete unsharded index");
    ARGCHECK(owner && world >= 1 && rank >= 0 && rank < world, "owner / rank / world");
    TRY(flush_pending(ix));
    HIPCHECK(hipDeviceSynchronize());
    const int64_t L = ix->n_lists;
    std::vector<int64_t> sizes((size_t)L), coff((size_t)L + 1, 0), loff((size_t)L + 1, 0);
    HIPCHECK(hipMemcpy(sizes.data(), ix->list_n.p, (size_t)L * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < L; i++) {
        ARGCHECK(owner[i] >= 0 && owner[i] < world, "owner out of range");
        const int64_t c = (sizes[(size_t)i] + 15) / 16;
        coff[(size_t)i + 1] = coff[(size_t)i] + c;
        loff[(size_t)i + 1] = loff[(size_t)i] + (owner[i] == rank ? c : 0);
    }
    const int P = ix->M / 2;
    TRY(ix->owner.ensure((size_t)L * 4));
    TRY(ix->local_chunk_off.ensure((size_t)(L + 1) * 8));
    HIPCHECK(hipMemcpy(ix->owner.p, owner, (size_t)L * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ix->local_chunk_off.p, loff.data(), (size_t)(L + 1) * 8, hipMemcpyHostToDevice));
    DevBuf mine;
    const size_t bytes = (size_t)tk_tiled_uint4s(loff[(size_t)L], P) * 16;
    TRY(mine.ensure(bytes > 0 ? bytes : 16));
    HIPCHECK(hipMemset(mine.p, 0, bytes > 0 ? bytes : 16));
    tk_launch_compact_tiled(ix->codes.as<uint4>(), mine.as<uint4>(), P, ix->list_chunk_off.as<int64_t>(),
                            ix->local_chunk_off.as<int64_t>(), (int)L, loff[(size_t)L], 0);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipDeviceSynchronize());
    ix->codes.release();
    ix->codes = mine;            // (DevBuf is a plain pointer + capacity)
    ix->sharded = true;
    ix->rank = rank;
    ix->world = world;
    return TK_OK;
}

// what a built index holds, back on the host in the reference's formats: list_sizes
// (n_lists,), codes (total chunks, M) uint64 Quick-ADC layout, ids (sum sizes,) — any NULL
extern "C" int tk_index_export_lists(tk_index *ix, int64_t *list_sizes, uint64_t *codes, int64_t *ids)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->have_lists && !ix->sharded, "an unsharded index with lists");
    if (list_sizes)
        HIPCHECK(hipMemcpy(list_sizes, ix->list_n.p, (size_t)ix->n_lists * 8, hipMemcpyDeviceToHost));
    if (ids && ix->total_ids > 0)
        HIPCHECK(hipMemcpy(ids, ix->ids.p, (size_t)ix->total_ids * 8, hipMemcpyDeviceToHost));
    if (codes && ix->total_chunks > 0) {
        const int P = ix->M / 2;
        const int64_t n4 = tk_tiled_uint4s(ix->total_chunks, P);
        std::vector<uint4> t((size_t)n4);
        HIPCHECK(hipMemcpy(t.data(), ix->codes.p, (size_t)n4 * 16, hipMemcpyDeviceToHost));
        uint4 *ref = (uint4 *)codes;
        for (int64_t c = 0; c < ix->total_chunks; c++)
            for (int p = 0; p < P; p++) ref[c * P + p] = t[(size_t)(((c >> 3) * P + p) * 8 + (c & 7))];
    }
    return TK_OK;
}

// active_centers (n_lists, d) float32, center_codes (ceil(n_lists/16), M) uint64 — any NULL
extern "C" int tk_index_export_centers(tk_index *ix, float *active_centers, uint64_t *center_codes)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->have_centers, "an index with centres");
    if (active_centers)
        HIPCHECK(hipMemcpy(active_centers, ix->active_centers.p, (size_t)ix->n_lists * ix->d * 4,
                           hipMemcpyDeviceToHost));
    if (center_codes) {
        const int P = ix->M / 2;
        const int64_t n4 = tk_tiled_uint4s(ix->center_chunks, P);
        std::vector<uint4> t((size_t)n4);
        HIPCHECK(hipMemcpy(t.data(), ix->center_codes.p, (size_t)n4 * 16, hipMemcpyDeviceToHost));
        uint4 *ref = (uint4 *)center_codes;
        for (int64_t c = 0; c < ix->center_chunks; c++)
            for (int p = 0; p < P; p++) ref[c * P + p] = t[(size_t)(((c >> 3) * P + p) * 8 + (c & 7))];
    }
    return TK_OK;
}

// rows of IVF.data by id (float32 vectors), e.g. the candidates a checker wants to rescore
extern "C" int tk_index_read_rows(tk_index *ix, const int64_t *rows, int64_t n, float *out)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->have_data && !ix->data_is_f64, "an index with float32 vectors");
    ARGCHECK(n >= 0 && (n == 0 || (rows && out)), "buffers");
    for (int64_t i = 0; i < n; i++) ARGCHECK(rows[i] >= 0 && rows[i] < ix->N, "row id out of range");
    if (n == 0) return TK_OK;
    DevBuf r, o;
    int rc = r.ensure((size_t)n * 8);
    if (rc == TK_OK) rc = o.ensure((size_t)n * ix->d * 4);
    if (rc == TK_OK) {
        hipError_t e = hipMemcpy(r.p, rows, (size_t)n * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            tk_launch_gather_rows(ix->data.as<float>(), ix->d, r.as<int64_t>(), n, o.as<float>(), 0);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpy(out, o.p, (size_t)n * ix->d * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(TK_ERR_HIP, hipGetErrorString(e));
    }
    r.release();
    o.release();
    return rc;
}

// ---------------------------------------------------------------------------
// device front end ("fast mode")
extern "C" int tk_index_set_rotation(tk_index *ix, const double *R, int d_pad)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->have_pq && ix->have_centers, "set_pq and set_centers first");
    if (!R) {
        ix->rot_t.release();
        ix->rot_d_pad = 0;
        return TK_OK;
    }
    ARGCHECK(d_pad >= ix->d && d_pad <= 16384, "d_pad");
    return upload_rotation(ix, R, d_pad);
}

extern "C" int tk_index_prepare_dev(tk_index *ix, const float *q_raw_dev, int64_t nq, int angular,
                                    float *qn_dev, void *q_pq_dev, void *stream)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->have_pq && ix->have_centers, "set_pq and set_centers first");
    ARGCHECK(nq >= 0 && q_raw_dev && qn_dev && q_pq_dev, "buffers");
    ARGCHECK(!angular || ix->d <= 128, "device normalisation needs d <= 128");
    ARGCHECK(ix->rot_d_pad > 0 || ix->dq >= ix->d, "unrotated PQ: dq >= d");
    hipStream_t st = (hipStream_t)stream;
    if (angular)
        tk_launch_normalise_rows(q_raw_dev, nq, ix->d, qn_dev, st);
    else if (qn_dev != q_raw_dev)
        HIPCHECK(hipMemcpyAsync(qn_dev, q_raw_dev, (size_t)nq * ix->d * 4, hipMemcpyDeviceToDevice, st));
    tk_launch_prepare_queries(qn_dev, nq, ix->d, ix->rot_d_pad ? ix->rot_t.as<double>() : nullptr,
                              ix->dq, ix->rot_d_pad ? ix->rot_d_pad : ix->dq, q_pq_dev, st);
    HIPCHECK(hipGetLastError());
    return TK_OK;
}

extern "C" int tk_index_query_batch_raw(tk_index *ix, const float *q_raw, int64_t nq, int angular,
                                        int k, int n_probes, int pass_1, int64_t *out_ids)
{
    IXLOCK(ix);
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    ARGCHECK(nq >= 0 && q_raw && out_ids, "buffers");
    if (nq == 0) return TK_OK;
    const int f64 = ix->rot_d_pad > 0;
    DevBuf raw, outbuf;
    TRY(raw.ensure((size_t)nq * ix->d * 4));
    TRY(ix->q.ensure((size_t)nq * ix->d * 4));
    TRY(ix->qpq.ensure((size_t)nq * ix->dq * (f64 ? 8 : 4)));
    TRY(outbuf.ensure((size_t)nq * k * 8));
    HIPCHECK(hipMemcpy(raw.p, q_raw, (size_t)nq * ix->d * 4, hipMemcpyHostToDevice));
    int r = tk_index_prepare_dev(ix, raw.as<float>(), nq, angular, ix->q.as<float>(), ix->qpq.p, nullptr);
    if (r == TK_OK)
        r = tk_index_query_batch_dev(ix, ix->q.as<float>(), ix->qpq.p, f64, nq, k, n_probes, pass_1,
                                     outbuf.as<int64_t>(), nullptr);
    if (r == TK_OK) r = flush_pending(ix);
    if (r == TK_OK) {
        hipError_t e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(out_ids, outbuf.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) r = fail(TK_ERR_HIP, hipGetErrorString(e));
    }
    raw.release();
    outbuf.release();
    return r;
}

// ---------------------------------------------------------------------------
// exact k nearest vectors of IVF.data: the ground truth of recall (brute.hip)
extern "C" int tk_index_knn_brute(tk_index *ix, const float *q, int64_t nq, int k, int64_t *out_ids)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->have_data, "set_data first");
    ARGCHECK(!ix->data_is_f64, "float32 vectors only");
    ARGCHECK(ix->d <= 128, "d <= 128");
    ARGCHECK(nq >= 0 && q && out_ids, "buffers");
    ARGCHECK(k >= 1 && k <= 1024 && k <= ix->N, "1 <= k <= min(1024, N)");
    ARGCHECK(ix->N < (1ll << 31), "N < 2^31");
    if (nq == 0) return TK_OK;
    TRY(flush_pending(ix));
    const int64_t ns = ix->N < 8192 ? ix->N : 8192;
    const int cap = 8192;
    TRY(ix->br_ynorm.ensure((size_t)ix->N * 4));
    TRY(ix->br_tau.ensure((size_t)nq * 4));
    TRY(ix->br_vals.ensure((size_t)nq * ns * 4));
    TRY(ix->br_cand.ensure((size_t)nq * cap * 8));
    TRY(ix->br_count.ensure((size_t)nq * 4 + 4));
    TRY(ix->br_out.ensure((size_t)nq * k * 8));
    TRY(ix->br_q.ensure((size_t)nq * ix->d * 4));
    TRY(ix->br_sample.ensure((size_t)ns * (ix->d + 1) * 4));
    HIPCHECK(hipMemcpy(ix->br_q.p, q, (size_t)nq * ix->d * 4, hipMemcpyHostToDevice));
    int *overflow = ix->br_count.as<int>() + nq;
    if (tk_launch_knn_brute(ix->br_q.as<float>(), nq, ix->d, ix->data.as<float>(), ix->N, k,
                            ix->br_ynorm.as<float>(), ix->br_vals.as<float>(), ns,
                            ix->br_tau.as<float>(), ix->br_cand.as<unsigned long long>(), cap,
                            ix->br_count.as<int>(), overflow, ix->br_out.as<int64_t>(),
                            ix->br_sample.as<float>(), nullptr))
        return fail(TK_ERR_HIP, "tk_launch_knn_brute: unsupported size / LDS attribute");
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipDeviceSynchronize());
    int ov = 0;
    HIPCHECK(hipMemcpy(&ov, overflow, 4, hipMemcpyDeviceToHost));
    if (ov) return fail(TK_ERR_HIP, "knn_brute: candidate list overflow (a 2^20-row segment holds more than 8192 rows within the running k-th distance)");
    HIPCHECK(hipMemcpy(out_ids, ix->br_out.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
    return TK_OK;
}

// _FastDistanceTable.top (fast_pq.py:284-312) for a BATCH of queries against the coded rows the
// index holds as its "centres" (tk_index_set_pq + tk_index_set_centers(rows, packed codes) are
// all it needs): per query a heap of rescore = min(2k + 10, n) PQ estimates over all rows, then
// the exact distances of those candidates, k best in ascending order — the coarse stage of
// IVF.query (ivf.py:131) is exactly this call, so the same three kernels run.  Host buffers;
// queries are processed in chunks whose distance rows fit one workspace.
// Rows far longer than the heap (n >= 2^16 rows): the scan runs on the matrix cores (plain_scan.hip)
// behind an exact HEAD — the first n/64 rows — after which the heap is full of real values and its
// bound far below the table's limit C; the lane replay checks exactly that per query (bound at the
// first plain block <= C) and fetches only the blocks whose minimum passes its bound (LAZY).  A chunk
// of queries in which any query fails the check is answered again by the exact kernel alone, and an
// index on which more than 1 % fail (rows without structure) stays on the exact kernel.
extern "C" int tk_index_top_centers(tk_index *ix, const float *q, const void *q_pq, int q_pq_is_f64,
                                    int64_t nq, int k, int64_t *out_ids)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->have_pq && ix->have_centers, "set_pq and set_centers first");
    ARGCHECK(nq >= 0 && k >= 1 && (nq == 0 || (q && q_pq && out_ids)), "buffers / sizes");
    if (nq == 0) return TK_OK;
    TRY(flush_pending(ix));
    Plan p;
    const int64_t kc = k < ix->n_lists ? k : ix->n_lists;                        // fast_pq.py:263
    const int64_t rescore = 2 * kc + 10 < ix->n_lists ? 2 * kc + 10 : ix->n_lists;   // :264-265
    ARGCHECK(rescore * 12 + 16 <= 64 * 1024, "heap larger than 64 KiB of LDS");
    p.kc = (int)kc; p.rescore = (int)rescore; p.R = (int)rescore; p.S = 1;
    p.cap = 1; p.cap_min = 16;
    p.ccap_min = (ix->center_chunks + 15) / 16 * 16;
    int64_t chunk = (int64_t)(workspace_bytes() / ((double)ix->center_chunks * 17.0));
    chunk = chunk < 16 ? 16 : (chunk > MAX_SUB ? MAX_SUB : chunk);
    chunk = chunk < nq ? chunk : nq;
    Work &w = ix->works[0];
    const int M = ix->M;
    const size_t esz = q_pq_is_f64 ? 8 : 4;
    const bool lanes = ix->heap_mode == 0 && ix->center_chunks * 16 <= 0xffffff && p.rescore <= TK_LANES_MAX_R;
    const bool lazy = lanes && ix->center_chunks >= 1024;
    const int hc = (int)(ix->center_chunks / 64 < 16 ? 16 : ix->center_chunks / 64);     // exact head, in chunks
    bool flat_plain = lanes && ix->plain_mode != 1 && plain_env_on() && tk_plain_fits(M) && ix->flat_plain_ok &&
                      ix->center_chunks >= 4096 && coarse_units(ix, chunk);
    TkPairSet pl;
    TRY(w.tables.ensure((size_t)chunk * M * 16));
    TRY(w.shift.ensure((size_t)chunk * 8));
    TRY(w.scale.ensure((size_t)chunk * 8));
    TRY(w.cdist.ensure((size_t)chunk * ix->center_chunks * 16));
    TRY(w.cmins.ensure((size_t)chunk * p.ccap_min));
    TRY(w.cheap_idx.ensure((size_t)chunk * p.rescore * 8));
    TRY(w.cheap_val.ensure((size_t)chunk * p.rescore * 4));
    TRY(w.probes.ensure((size_t)chunk * p.kc * 8));
    TRY(w.c_pair_off.ensure(8));
    TRY(w.c_unit_prefix.ensure(tk_unit_prefix_ints(1) * 4));
    TRY(w.c_pair_q.ensure(((size_t)chunk + 4) * 4));
    TRY(w.c_pair_f0.ensure(((size_t)chunk + 4) * 4));
    TRY(ix->q.ensure((size_t)chunk * ix->d * 4));
    TRY(ix->qpq.ensure((size_t)chunk * ix->dq * esz));
    if (flat_plain) {
        const int K = 64;
        const int64_t nsub = ((ix->center_chunks + 1) / 2 + K - 1) / K;
        TRY(w.qlim.ensure((size_t)chunk * 4));
        TRY(w.plain0.ensure((size_t)chunk * 4));
        TRY(w.repeat_flag.ensure((size_t)chunk));
        TRY(w.flag_list.ensure(((size_t)chunk + 1) * 4));
        TRY(w.p_pair_off.ensure(8));
        TRY(w.p_unit_prefix.ensure(tk_unit_prefix_ints(1) * 4));
        TRY(w.p_pair_q.ensure(((size_t)chunk + 4) * 4));
        TRY(w.p_pair_f0.ensure(((size_t)chunk + 4) * 4));
        TRY(w.p_unit_desc.ensure((size_t)((chunk + 31) / 32) * nsub * 16 + 64));
        if (!w.flag_host) {
            HIPCHECK(hipHostMalloc((void **)&w.flag_host, 64, hipHostMallocDefault));
            *w.flag_host = 0;
        }
        pl = TkPairSet{nullptr, nullptr, w.p_pair_off.as<int>(), w.p_unit_prefix.as<int>(), w.p_pair_q.as<int>(),
                       w.p_pair_f0.as<int>(), w.p_unit_desc.as<int>(), K};
        std::vector<int> h((size_t)chunk, hc);      // every query: plain sums from flat chunk hc on
        HIPCHECK(hipMemcpy(w.plain0.p, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    // heap replay over the centre rows (fresh heap, positions as labels) + exact rescoring -> w.probes
    auto replay_rescore = [&](int64_t m, bool plain) -> int {
        if (!lanes) {
            Prof pf;
            return coarse_replay_probes(ix, w, ix->q.as<float>(), m, p, w.probes.as<int64_t>(), nullptr, pf);
        }
        if (tk_launch_heap_replay_lanes(w.cdist.as<uint4>(), ix->center_chunks, m, ix->cslots_i.as<int>(),
                                        ix->cslots_i.as<int>() + 2, ix->cslots_l.as<int64_t>(), 1, nullptr,
                                        w.cheap_idx.as<int64_t>(), w.cheap_val.as<int32_t>(), p.rescore, 1, 1,
                                        plain ? w.repeat_flag.as<unsigned char>() : nullptr, w.cmins.as<uint8_t>(),
                                        p.ccap_min, nullptr, nullptr, plain ? w.plain0.as<int>() : nullptr,
                                        plain ? w.qlim.as<int>() : nullptr, lazy ? 1 : 0))
            return fail(TK_ERR_HIP, "hipFuncSetAttribute(LDS size) failed");
        tk_launch_rescore(ix->q.as<float>(), 0, ix->d, ix->active_centers.p, 0, ix->n_lists,
                          w.cheap_idx.as<int64_t>(), p.rescore, m, p.kc, 0, w.probes.as<int64_t>(), nullptr, nullptr,
                          ix->opt_rescore_form);
        return TK_OK;
    };
    for (int64_t o = 0; o < nq; o += chunk) {
        const int64_t m = nq - o < chunk ? nq - o : chunk;
        HIPCHECK(hipMemcpy(ix->q.p, q + o * ix->d, (size_t)m * ix->d * 4, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(ix->qpq.p, (const char *)q_pq + (size_t)o * ix->dq * esz, (size_t)m * ix->dq * esz,
                           hipMemcpyHostToDevice));
        Prof pf;
        bool exact = !flat_plain;
        if (flat_plain) {
            TRY(stage_tables(ix, w, ix->qpq.p, q_pq_is_f64, m, nullptr, pf, true));
            HIPCHECK(hipMemsetAsync(w.repeat_flag.p, 0, (size_t)m, nullptr));
            // plain sums of every row first; the exact kernel then overwrites the head chunks
            tk_launch_plain_identity(m, (int)ix->center_chunks, pl, nullptr);
            TkScanJob pj = coarse_job(ix, w, p);
            pj.unit_prefix = pl.unit_prefix; pj.pair_off = pl.pair_off; pj.pair_q = pl.pair_q; pj.pair_f0 = pl.pair_f0;
            pj.unit_desc4 = pl.unit_desc;
            if (tk_launch_scan_plain(pj, M, ix->order, plain_blocks(), nullptr))
                return fail(TK_ERR_HIP, "scan_plain_wave_kernel: LDS attribute / unsupported M");
            tk_launch_identity_pairs(m, hc, w.c_pair_off.as<int>(), w.c_unit_prefix.as<int>(),
                                     w.c_pair_q.as<int>(), w.c_pair_f0.as<int>(), nullptr);
            TkScanJob hj = coarse_job(ix, w, p), none;
            memset(&none, 0, sizeof none);
            hj.max_chunks = hc;
            tk_launch_scan_units2(hj, none, M, ix->order, 768, nullptr, nullptr, 0);
            TRY(replay_rescore(m, true));
            tk_launch_flagged_list(w.repeat_flag.as<unsigned char>(), m, w.flag_list.as<int>(), nullptr, w.flag_host);
            HIPCHECK(hipGetLastError());
            HIPCHECK(hipDeviceSynchronize());
            const int flagged = *w.flag_host;
            if (flagged > 0) exact = true;                       // (this chunk again, exactly)
            if ((double)flagged > 0.01 * (double)m) {            // rows without structure: not again on this index
                ix->flat_plain_ok = false;
                flat_plain = false;
            }
        }
        if (exact) {
            TRY(stage_tables(ix, w, ix->qpq.p, q_pq_is_f64, m, nullptr, pf));
            launch_coarse_scan(ix, w, m, p, nullptr);
            TRY(replay_rescore(m, false));
            HIPCHECK(hipGetLastError());
            HIPCHECK(hipDeviceSynchronize());
        }
        if (p.kc == k) {
            HIPCHECK(hipMemcpy(out_ids + o * k, w.probes.p, (size_t)m * k * 8, hipMemcpyDeviceToHost));
        } else {    // fewer rows than k: rows of kc ids into rows of k, padded with -1
            std::vector<int64_t> tmp((size_t)m * p.kc);
            HIPCHECK(hipMemcpy(tmp.data(), w.probes.p, tmp.size() * 8, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < m; i++)
                for (int t = 0; t < k; t++)
                    out_ids[(o + i) * k + t] = t < p.kc ? tmp[(size_t)i * p.kc + t] : -1;
        }
    }
    return TK_OK;
}

extern "C" int tk_index_set_pipeline(tk_index *ix, int depth)
{
    IXLOCK(ix);
    ARGCHECK(ix, "null index");
    ARGCHECK(depth >= 1 && depth <= 8, "depth must be in 1..8");
    TRY(flush_pending(ix));
    HIPCHECK(hipDeviceSynchronize());
    ix->capturing = false;
    plain_poll(ix);         // (every verdict is in: none is lost with a workspace released below)
    if (ix->plain_state == PLAIN_WAIT) ix->plain_state = PLAIN_PROBE;
    // depth replays in flight + three calls waiting for their list scan + slack
    const size_t n_works = depth > 1 ? (size_t)depth + 5 : 1;
    while (ix->works.size() > n_works) {
        ix->works.back().release();
        ix->works.pop_back();
    }
    ix->works.resize(n_works);
    for (Work &w : ix->works) w.busy = false;
    ix->depth = depth;
    ix->calls = 0;
    return TK_OK;
}

extern "C" int tk_index_set_coalesce(tk_index *ix, int n)
{
    IXLOCK(ix);
    ARGCHECK(ix, "null index");
    ARGCHECK(n == 1 || n == 2, "1 (every call its own batch) or 2 (pairs of calls as one batch)");
    TRY(flush_pending(ix));
    ix->coalesce = n;
    return TK_OK;
}

extern "C" int tk_index_join(tk_index *ix, void *stream)
{
    IXLOCK(ix);
    ARGCHECK(ix, "null index");
    TRY(flush_pending(ix));
    if (ix->depth > 1)
        for (Work &w : ix->works)
            if (w.busy) HIPCHECK(hipStreamWaitEvent((hipStream_t)stream, w.done, 0));
    return TK_OK;
}

// Everything enqueued so far has completed and no workspace remembers an event of it: what a
// stream capture of the pipelined mode needs first (a captured call must not wait on an event
// recorded outside the capture).  Synchronises the device.
extern "C" int tk_index_quiesce(tk_index *ix)
{
    IXLOCK(ix);
    ARGCHECK(ix, "null index");
    TRY(flush_pending(ix));
    HIPCHECK(hipDeviceSynchronize());
    ix->capturing = false;
    plain_poll(ix);
    if (ix->plain_state == PLAIN_WAIT) ix->plain_state = PLAIN_PROBE;
    for (Work &w : ix->works) {
        w.busy = false;
        w.plain_pending = false;
    }
    return TK_OK;
}

extern "C" int tk_index_query_batch(tk_index *ix, const float *q, const void *q_pq,
                                    int q_pq_is_f64, int64_t nq, int k, int n_probes, int pass_1,
                                    int64_t *out_ids, int64_t *out_probes, int64_t *out_heap_idx,
                                    int32_t *out_heap_val)
{
    IXLOCK(ix);
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    ARGCHECK(nq >= 0, "nq");
    if (nq == 0) return TK_OK;
    ARGCHECK(!(out_probes || out_heap_idx || out_heap_val) || nq <= sub_batch(p),
             "debug outputs need the batch to fit one sub-batch");
    const size_t esz = q_pq_is_f64 ? 8 : 4;
    TRY(ix->q.ensure((size_t)nq * ix->d * 4));
    TRY(ix->qpq.ensure((size_t)nq * ix->dq * esz));
    DevBuf outbuf;  // separate from the sub-batch `out` workspace
    TRY(outbuf.ensure((size_t)nq * k * 8));
    HIPCHECK(hipMemcpy(ix->q.p, q, (size_t)nq * ix->d * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ix->qpq.p, q_pq, (size_t)nq * ix->dq * esz, hipMemcpyHostToDevice));
    int r = tk_index_query_batch_dev(ix, ix->q.as<float>(), ix->qpq.p, q_pq_is_f64, nq, k, n_probes,
                                     pass_1, outbuf.as<int64_t>(), nullptr);
    if (r == TK_OK) r = flush_pending(ix);
    const Work &lw = ix->works[(ix->calls + ix->works.size() - 1) % ix->works.size()];   // last used
    if (r == TK_OK) {
        hipError_t e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(out_ids, outbuf.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess && out_probes)
            e = hipMemcpy(out_probes, lw.probes.p, (size_t)nq * p.kc * 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess && out_heap_idx)
            e = hipMemcpy(out_heap_idx, lw.heap_idx.p, (size_t)nq * p.R * 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess && out_heap_val)
            e = hipMemcpy(out_heap_val, lw.heap_val.p, (size_t)nq * p.R * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) r = fail(TK_ERR_HIP, hipGetErrorString(e));
    }
    outbuf.release();
    return r;
}

extern "C" int tk_index_set_heap_mode(tk_index *ix, int mode)
{
    IXLOCK(ix);
    ARGCHECK(ix, "null index");
    ARGCHECK(mode >= 0 && mode <= 2, "mode");
    TRY(flush_pending(ix));
    ix->heap_mode = mode;
    return TK_OK;
}

extern "C" int tk_index_set_option(tk_index *ix, int option, int value)
{
    IXLOCK(ix);
    ARGCHECK(ix, "null index");
    TRY(flush_pending(ix));
    switch (option) {
    case TK_OPT_SCAN_FORM:
        ARGCHECK(value >= 0 && value <= 2, "TK_OPT_SCAN_FORM: 0, 1 or 2");
        ix->opt_scan_form = value;
        return TK_OK;
    case TK_OPT_RESCORE_FORM:
        ARGCHECK(value >= 0 && value <= 2, "TK_OPT_RESCORE_FORM: 0, 1 or 2");
        ix->opt_rescore_form = value;
        return TK_OK;
    case TK_OPT_PLAIN_LIMIT:
        ix->opt_plain_limit = value;
        return TK_OK;
    default:
        return fail(TK_ERR_ARG, "bad argument: unknown option");
    }
}

extern "C" int tk_index_set_scan_mode(tk_index *ix, int mode)
{
    IXLOCK(ix);
    ARGCHECK(ix, "null index");
    ARGCHECK(mode >= 0 && mode <= 2, "mode");
    TRY(flush_pending(ix));
    ix->scan_mode = mode;
    return TK_OK;
}

extern "C" int tk_index_set_plain_scan(tk_index *ix, int mode)
{
    IXLOCK(ix);
    ARGCHECK(ix, "null index");
    ARGCHECK(mode >= 0 && mode <= 2, "mode");
    TRY(flush_pending(ix));
    ix->plain_mode = mode;
    ix->plain_state = PLAIN_PROBE;
    ix->plain_skip = 0;
    ix->plain_backoff = 256;
    return TK_OK;
}

// What the plain path did for the LAST batch enqueued (synchronises): out8 = plain units (tiles of
// 32 pairs), plain pairs, exact pair records (whole lists, padded to groups of 4), head pair
// records, queries flagged for the re-scan, sum over the plain units of the list's chunk pairs
// (x 26 MFMAs of 32 x 32 x 32 = the matrix-core work), the adaptive state (0 probe, 1 wait, 2 on,
// 3 paused: plain_poll), batches left of the pause.  The first six are zero when the LAST batch
// went the exact way.
extern "C" int tk_index_plain_stats(tk_index *ix, int64_t *out8)
{
    IXLOCK(ix);
    ARGCHECK(ix && out8, "null index / buffer");
    for (int i = 0; i < 8; i++) out8[i] = 0;
    TRY(flush_pending(ix));
    HIPCHECK(hipDeviceSynchronize());
    if (plain_adaptive(ix)) plain_poll(ix);
    out8[6] = ix->plain_state;
    out8[7] = ix->plain_state == PLAIN_OFF ? ix->plain_skip : 0;
    const Work &w = ix->works[(ix->calls + ix->works.size() - 1) % ix->works.size()];
    if (!w.last_plain || !w.p_unit_prefix.p || !w.flag_list.p || ix->n_lists < 1) return TK_OK;
    const int64_t L = ix->n_lists;
    int v[4] = {0, 0, 0, 0};
    HIPCHECK(hipMemcpy(&v[0], w.p_unit_prefix.as<int>() + L, 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(&v[1], w.p_pair_off.as<int>() + L, 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(&v[2], w.u_pair_off.as<int>() + L, 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(&v[3], w.h_pair_off.as<int>() + L, 4, hipMemcpyDeviceToHost));
    int flagged = 0;
    HIPCHECK(hipMemcpy(&flagged, w.flag_list.p, 4, hipMemcpyDeviceToHost));
    out8[1] = v[1]; out8[2] = v[2]; out8[3] = v[3]; out8[4] = flagged;
    if (v[0] > 0) {
        // v[0] units = (tile, range of chunk pairs); a tile's first range starts at chunk pair 0
        std::vector<int> desc((size_t)v[0] * 4);
        HIPCHECK(hipMemcpy(desc.data(), w.p_unit_desc.p, desc.size() * 4, hipMemcpyDeviceToHost));
        int64_t cps = 0, tiles = 0;
        for (int u = 0; u < v[0]; u++) {
            tiles += desc[(size_t)4 * u + 2] == 0;
            cps += desc[(size_t)4 * u + 3] - desc[(size_t)4 * u + 2];
        }
        out8[0] = tiles;
        out8[5] = cps;
    }
    return TK_OK;
}

extern "C" int tk_index_set_profiling(tk_index *ix, int on)
{
    IXLOCK(ix);
    ARGCHECK(ix, "null index");
    ix->profiling = on < 0 ? 0 : on;
    ix->prof_seen = 0;
    ix->ev_used = 0;
    return TK_OK;
}

extern "C" int tk_index_last_profile(tk_index *ix, float *ms8, double *scan_bytes, int *batches)
{
    IXLOCK(ix);
    ARGCHECK(ix, "null index");
    float *ms7 = ms8;
    for (int i = 0; i < 8; i++) ms8[i] = 0;
    *scan_bytes = 0;
    TRY(flush_pending(ix));
    *batches = (int)ix->ev_used;
    if (ix->ev_used == 0) return TK_OK;
    for (size_t b = 0; b < ix->ev_used; b++) HIPCHECK(hipStreamSynchronize(ix->ev_streams[b]));
    for (size_t b = 0; b < ix->ev_used; b++)
        for (int i = 0; i < 7; i++) {
            float ms = 0;
            HIPCHECK(hipEventElapsedTime(&ms, ix->evs[b * TK_PROF_EVENTS + i], ix->evs[b * TK_PROF_EVENTS + i + 1]));
            ms7[i] += ms / (float)ix->ev_used;
        }
    {   // the plain kernel alone (events on the stream it is launched on), over the sets that ran it
        int n_plain = 0;
        for (size_t b = 0; b < ix->ev_used; b++)
            if (b < ix->ev_plain.size() && ix->ev_plain[b]) {
                float ms = 0;
                HIPCHECK(hipEventElapsedTime(&ms, ix->evs[b * TK_PROF_EVENTS + 8], ix->evs[b * TK_PROF_EVENTS + 9]));
                ms8[7] += ms;
                n_plain++;
            }
        if (n_plain) ms8[7] /= (float)n_plain;
    }
    // algorithmic bytes of the list scan of the most recent sub-batch:
    // per query  sum over probed lists ceil(n/16)*M*8  +  16*M (table)  +  12*R (heap)
    const int S = ix->last_S;
    std::vector<int> pre((size_t)ix->last_nq * (S + 1));
    HIPCHECK(hipMemcpy(pre.data(), ix->works[ix->last_work].slot_prefix.p, pre.size() * 4,
                       hipMemcpyDeviceToHost));
    double bytes = 0;
    for (int64_t i = 0; i < ix->last_nq; i++)
        bytes += (double)pre[(size_t)i * (S + 1) + S] * ix->M * 8 + 16.0 * ix->M + 12.0 * ix->last_R;
    if (ix->depth > 1)
        // pipelined mode: the timed launch also carries the next batch's coarse scan
        bytes += (double)ix->last_nq * ((double)ix->center_chunks * ix->M * 8 + 16.0 * ix->M +
                                        12.0 * (2 * S + 10 < ix->n_lists ? 2 * S + 10 : (int)ix->n_lists));
    *scan_bytes = bytes;
    ix->ev_used = 0;
    return TK_OK;
}
