// api.hip — the C ABI of libtinyknn_hip.so (declared in include/tinyknn_hip.h).
// Host-pointer entry points stage their arguments into HBM, run the gfx950
// kernels and copy results back; the tk_index_* entry points keep the index
// resident and only enqueue kernels.  There is no CPU implementation behind any of
// these calls.  This file: errors, device selection, the host-pointer drop-in entry points
// (tk_estimate_pq, tk_query_pq, heap primitives, tables, brute force), resident code arrays (tk_codes_*)
// and the offline encode / assign entry points.  The resident index: api_index.hip, api_shard.hip,
// api_build.hip.
#include "api_internal.h"

static thread_local std::string g_err;

int tk_fail(int code, const std::string &msg)     // (also front.hip)
{
    g_err = msg;
    return code;
}

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

int tk_require_gpu()
{
    if (tk_device_count() <= 0)
        return fail(TK_ERR_HIP, "no HIP device visible: libtinyknn_hip has no CPU fallback");
    return TK_OK;
}




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
    // A FRESH heap of up to 513 entries (init_heap's arrays: the first query_pq of a DistanceTable.top / IVF.query): the
    // wave-per-query replay with the heap in registers (heap.hip: heap_replay_pair_kernel; position entries without labels,
    // (value, label64) entries with the reference's duplicate test with them) instead of the general kernel's LDS heap —
    // ~300 instead of ~2 000 cycles per insert.  Same arrays out.
    bool fresh_heap = R <= TK_PAIR_MAX_R && chunks * 16 <= 0xffffff;
    for (int i = 0; i < R && fresh_heap; i++) fresh_heap = indices[i] == -1 && vals[i] == (signd ? 127 : 255);
    if (fresh_heap) {
        if (tk_launch_heap_replay_pair(S.out.as<uint4>(), chunks, 1, S.mins.as<uint8_t>(), cap_min, S.slots_i.as<int>(),
                                       S.slots_i.as<int>() + 2, S.slots_l.as<int64_t>(), 1, S.labels.as<int64_t>(),
                                       S.hidx.as<int64_t>(), S.hval.as<int32_t>(), R, signd, 1, nullptr, labels ? 1 : 0, st))
            return fail(TK_ERR_HIP, "heap_replay_pair_kernel launch failed");
    } else {
        tk_launch_heap_replay(S.out.as<uint4>(), chunks, 1, S.slots_i.as<int>(),
                              S.slots_i.as<int>() + 2, S.slots_l.as<int64_t>(), 1,
                              S.labels.as<int64_t>(), S.hidx.as<int64_t>(), S.hval.as<int32_t>(), R,
                              signd, 1, nullptr, st, S.mins.as<uint8_t>(), cap_min);
    }
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
    bool fresh_heap = !labels && R <= TK_PAIR_MAX_R && (size_t)c->M * 16 <= 32 * 1024;
    for (int i = 0; i < R && fresh_heap; i++) fresh_heap = indices[i] == -1 && vals[i] == (signd ? 127 : 255);
    const bool fresh = fresh_heap && chunks >= 4096 && R <= 64;
    if (fresh_heap && !fresh && chunks * 16 <= 0xffffff) {
        // A fresh heap of up to 513 entries over a shorter array (DistanceTable.top of one query over a few thousand to a
        // million rows, examples/example.py: 16 000 rows, R = 2 k + 10 = 30): table in through pinned memory, the scan, the
        // wave-per-query replay with the heap in registers (heap.hip: heap_replay_pair_kernel, position entries — positions
        // ARE the labels here), the heap written straight to pinned memory: 4 operations instead of 11, ~300 instead of
        // ~2 000 cycles per insert.
        if (!c->pin) HIPCHECK(hipHostMalloc(&c->pin, 64 * 1024, hipHostMallocDefault));
        TRY(c->slots_i.ensure(3 * sizeof(int)));
        TRY(c->slots_l.ensure(sizeof(int64_t)));
        unsigned char *pin = (unsigned char *)c->pin;
        int64_t *pidx = (int64_t *)(pin + 32 * 1024);
        int32_t *pval = (int32_t *)(pin + 32 * 1024 + (size_t)TK_PAIR_MAX_R * 8);
        int *psl = (int *)(pin + 32 * 1024 + (((size_t)TK_PAIR_MAX_R * 12 + 15) & ~(size_t)15));      // slot table: prefix[0..1], n, label offset
        memcpy(pin, tables, (size_t)c->M * 16);
        const int64_t nclamp = n > (int64_t)0x7fffffff ? 0x7fffffff : (n < 0 ? 0 : n);
        psl[0] = 0; psl[1] = (int)chunks; psl[2] = (int)nclamp; psl[3] = 0;
        ((int64_t *)(psl + 4))[0] = -1;
        HIPCHECK(hipMemcpyAsync(c->tables.p, pin, (size_t)c->M * 16, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(c->slots_i.p, psl, 3 * sizeof(int), hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(c->slots_l.p, psl + 4, sizeof(int64_t), hipMemcpyHostToDevice, st));
        tk_launch_scan_flat(c->tiled.as<uint4>(), chunks, c->M, c->tables.as<uint4>(), 1,
                            c->out.as<uint4>(), chunks, c->mins.as<uint8_t>(), cap_min, signd, order, st);
        if (tk_launch_heap_replay_pair(c->out.as<uint4>(), chunks, 1, c->mins.as<uint8_t>(), cap_min, c->slots_i.as<int>(),
                                       c->slots_i.as<int>() + 2, c->slots_l.as<int64_t>(), 1, nullptr, pidx, pval, R, signd,
                                       1, nullptr, 0, st))
            return fail(TK_ERR_HIP, "heap_replay_pair_kernel launch failed");
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipStreamSynchronize(st));
        memcpy(indices, pidx, (size_t)R * 8);
        memcpy(vals, pval, (size_t)R * 4);
        return TK_OK;
    }
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
