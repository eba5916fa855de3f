// api_index.hip — the device-resident index behind tk_index_*: plan, workspaces, the stages of a batch,
// the pipelined mode and pairs of calls, options and statistics.  (Split from api.hip in round 4; the
// shared structures are in api_internal.h.)
#include "api_internal.h"

// ---------------------------------------------------------------------------
// device-resident index

struct Pending;
static int batch_epilogue(const struct Pending &b, hipStream_t st);


extern "C" tk_index *tk_index_create(void)
{
    if (require_gpu() != TK_OK) return nullptr;
    tk_index *ix = new tk_index();
    ix->works.resize(1);
    // A/B: TINYKNN_PLAIN_SCAN=2 starts every index in mode 2 (plain always, repeating labels too)
    const char *e = getenv("TINYKNN_PLAIN_SCAN");
    if (e && e[0] == '2') ix->plain_mode = 2;
    // the default of TK_OPT_PAIR_NQ (the test suite starts with a small one so that batches of a few hundred queries keep
    // exercising the lane kernels)
    e = getenv("TINYKNN_PAIR_NQ");
    if (e && e[0] >= '0' && e[0] <= '9') ix->opt_pair_nq = atoi(e);
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
                      &ix->br_cand, &ix->br_count, &ix->br_out, &ix->br_q, &ix->br_sample, &ix->replay_counters,
                      &ix->twin_list, &ix->twin_off};
    for (DevBuf *b : bufs) b->release();
    for (Work &w : ix->works) w.release();
    // (the internal streams belong to the process: shared_streams below)
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

// twins.hip's table for an index whose lists, ids_off and int32 labels are in place.  The table is an
// ACCELERATOR (the TWIN replay; without it the label-based duplicate test runs): an allocation or HIP failure
// in here means "no table", never a failed upload — the temporaries go on every path, the error is cleared.
struct TwinScratch {
    DevBuf cnt, where;
    ~TwinScratch() { cnt.release(); where.release(); }
};

static int build_twins_try(tk_index *ix, int64_t label_bound, TwinScratch &t)
{
    const int64_t T = ix->total_ids;
    TRY(t.cnt.ensure((size_t)(label_bound + 1) * 4));
    HIPCHECK(hipMemsetAsync(t.cnt.p, 0, (size_t)(label_bound + 1) * 4, 0));
    int *cnt_max = t.cnt.as<int>() + label_bound;
    tk_launch_twin_count(ix->ids32.as<int32_t>(), T, t.cnt.as<int>(), cnt_max, 0);
    int b = 0;
    HIPCHECK(hipMemcpy(&b, cnt_max, 4, hipMemcpyDeviceToHost));
    if (!(b >= 2 && b <= 17 && T * (b - 1) < (1ll << 31)))     // (the replay indexes the table with 32-bit arithmetic)
        return TK_OK;
    const int w = b - 1;
    // transient label_bound * b * 4 bytes + 2 * T * w * 4 persistent: only where the device has them to spare
    size_t free_b = 0, total_b = 0;
    HIPCHECK(hipMemGetInfo(&free_b, &total_b));
    const size_t need = (size_t)label_bound * b * 4 + 2 * (size_t)T * w * 4;
    if (need + need / 4 + (512u << 20) > free_b) return TK_OK;
    TRY(t.where.ensure((size_t)label_bound * b * 4));
    TRY(ix->twin_list.ensure((size_t)T * w * 4));
    TRY(ix->twin_off.ensure((size_t)T * w * 4));
    HIPCHECK(hipMemsetAsync(t.cnt.p, 0, (size_t)label_bound * 4, 0));
    tk_launch_twin_fill(ix->ids32.as<int32_t>(), T, t.cnt.as<int>(), t.where.as<int>(), b,
                        ix->ids_off.as<int64_t>(), (int)ix->n_lists, ix->twin_list.as<int32_t>(),
                        ix->twin_off.as<int32_t>(), 0);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipDeviceSynchronize());
    if (!ix->sharded) {      // every list's codes are here: the table's two premises, checked (twins.hip)
        HIPCHECK(hipMemsetAsync(t.cnt.p, 0, 4, 0));
        tk_launch_twin_verify(ix->codes.as<uint4>(), ix->M / 2, ix->list_chunk_off.as<int64_t>(),
                              ix->ids_off.as<int64_t>(), (int)ix->n_lists, ix->twin_list.as<int32_t>(),
                              ix->twin_off.as<int32_t>(), w, T, t.cnt.as<int>(), 0);
        int bad = 0;
        HIPCHECK(hipMemcpy(&bad, t.cnt.p, 4, hipMemcpyDeviceToHost));
        if (bad) return TK_OK;   // copies with different codes, or two in one list: the label-based test only
    }
    ix->twin_w = w;              // (set last: a failure above leaves "no table")
    ix->twin_unverified = ix->sharded;
    return TK_OK;
}

int build_twins(tk_index *ix, int64_t label_bound)
{
    ix->twin_w = 0;
    ix->twin_unverified = false;
    ix->twin_list.release();
    ix->twin_off.release();
    const int64_t T = ix->total_ids;
    if (!ix->have_ids32 || ix->ids_unique || T <= 0 || T >= (1ll << 31) || label_bound <= 0 ||
        label_bound > 8 * T + 1024 || ix->n_lists >= (1ll << 31))
        return TK_OK;
    TwinScratch t;
    const int rc = build_twins_try(ix, label_bound, t);
    if (rc != TK_OK || ix->twin_w == 0) {
        ix->twin_w = 0;
        ix->twin_unverified = false;
        ix->twin_list.release();
        ix->twin_off.release();
    }
    if (rc != TK_OK) (void)hipGetLastError();      // (an out-of-memory answer is sticky until read)
    return TK_OK;
}

extern "C" int tk_index_twin_table(tk_index *ix, int64_t *rows, int *w, int32_t *list_out, int32_t *off_out)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->have_lists, "an index with its lists");
    if (rows) *rows = ix->total_ids;
    if (w) *w = ix->twin_w;
    const size_t bytes = (size_t)ix->total_ids * (size_t)ix->twin_w * 4;
    if (list_out && bytes) HIPCHECK(hipMemcpy(list_out, ix->twin_list.p, bytes, hipMemcpyDeviceToHost));
    if (off_out && bytes) HIPCHECK(hipMemcpy(off_out, ix->twin_off.p, bytes, hipMemcpyDeviceToHost));
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
    {   // labels that repeat: where every row's other copies are (the lane replay's duplicate test)
        int64_t mx = -1, mn = 0;
        for (int64_t i = 0; i < ioff[L]; i++) {
            mx = ids[i] > mx ? ids[i] : mx;
            mn = ids[i] < mn ? ids[i] : mn;
        }
        ix->labels24 = mn >= 0 && mx < 0x00ffffff;
        TRY(build_twins(ix, mx + 1));
    }
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


int make_plan(const tk_index *ix, int k, int n_probes, int pass_1, Plan &p)
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
bool plain_env_on()
{
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("TINYKNN_PLAIN_SCAN");
        on = !(e && e[0] == '0');
    }
    return on != 0;
}
// the TWIN form of the lane replay applies (heap.hip): repeating labels with a twin table, fresh heaps on packed
// position entries
bool twin_replay(const tk_index *ix, const Plan &p)
{
    return !ix->ids_unique && ix->twin_w > 0 && ix->opt_replay_twin == 1 && ix->heap_mode == 0 &&
           (!ix->twin_unverified || ix->twin_vouched) &&
           ix->total_ids < (1ll << 31) && p.cap * 16 <= 0xffffff && tk_lanes_twin_fits(p.R, p.S, ix->n_lists);
}

// the wave-per-query replay with the heap in registers (heap.hip: heap_replay_pair_kernel): small batches — ONE query
// per call above all — where the lane kernel would spend a wave's round on a few lanes; heap_mode 3 forces it
static bool pair_replay(const tk_index *ix, int64_t nq, int R)
{
    if (R > TK_PAIR_MAX_R) return false;
    if (ix->heap_mode == 3) return true;
    // One batch at a time the register heap wins up to ~10 000 queries (8 000: 0.30 against 0.69 ms,
    // profiles/r06/query1_and_small_batches.txt) — but it fills every SIMD's issue slots, where the lane kernel leaves
    // the chip to the scans of the batches beside it: with batches in flight (pipelined mode) the headline batch LOSES
    // 38 % on it (25.9 -> 16.1 M queries/s).  With batches in flight it pays up to ~4 000 queries per LAUNCH (a pair of
    // calls of 2 000): calls of 500 / 1 000 / 2 000 queries 2.7 -> 7.3, 5.1 -> 11.6, 9.5 -> 14.1 M queries/s, calls of
    // 4 000 (launches of 8 000) 15.9 -> 15.5 (profiles/r06/queries_per_call_sweep.txt).  So: the option's value one
    // batch at a time (default 8 192), at most 4 096 per launch when pipelined (TINYKNN_PAIR_NQ_PIPE: A/B).
    // (a list-sharded index's depth is its number of workspace slots: the caller's batches, the option's value)
    static const int64_t pipe_cap = [] { const char *e = getenv("TINYKNN_PAIR_NQ_PIPE"); return e ? atoll(e) : 4096ll; }();
    const int64_t limit = ix->depth > 1 && !ix->sharded ? (ix->opt_pair_nq < pipe_cap ? ix->opt_pair_nq : pipe_cap) : ix->opt_pair_nq;
    return ix->heap_mode == 0 && nq <= limit;
}

static bool plain_possible(const tk_index *ix, const Plan &p)
{
    if (ix->plain_mode == 1 || !plain_env_on() || ix->sharded || p.S < 2 || !tk_plain_fits(ix->M)) return false;
    // (heap_mode 3: the register heap makes the lemma's check as the lane kernel does)
    if ((ix->heap_mode != 0 && !(ix->heap_mode == 3 && p.R <= TK_PAIR_MAX_R)) || ix->scan_mode == 1 || p.cap * 16 > 0xffffff)
        return false;
    if (ix->ids_unique) return p.R <= TK_LANES_MAX_R;
    // repeating labels with the TWIN form of the lane replay: as with distinct labels
    if (twin_replay(ix, p)) return true;
    // repeating labels with the hash set (build n_probes >= 2): such a batch is bound by the replay with the duplicate
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
int plain_k(const tk_index *ix, int64_t nq, const Plan &p)
{
    const double iters = (double)nq * p.S / 32.0 * ((double)ix->total_chunks / (double)ix->n_lists) / 2.0;
    int k = (int)(iters / (2048.0 * 6.0));
    k = (k + 3) & ~3;
    return k < 12 ? 12 : (k > 64 ? 64 : k);
}

size_t plain_desc_bytes(const tk_index *ix, int64_t nq, const Plan &p)
{
    return (size_t)tk_plain_units_bound(nq * p.S, ix->n_lists, ix->total_chunks, ix->max_list_chunks,
                                        plain_k(ix, nq, p)) * 16;
}

int reserve_slots_pool(Work &w)
{
    if (w.slots_pinned) return TK_OK;
    TRY(w.slots_desc.ensure(sizeof(TkSlotsOut) * TK_SLOTS_POOL));
    HIPCHECK(hipHostMalloc((void **)&w.slots_pinned, sizeof(TkSlotsOut) * TK_SLOTS_POOL, hipHostMallocDefault));
    w.slots_n = 0;
    w.slots_need_upload = 0;
    return TK_OK;
}

static int reserve(tk_index *ix, Work &w, int64_t nq, int k, const Plan &p)
{
    const int M = ix->M;
    if (!ix->capturing) TRY(reserve_slots_pool(w));      // (no pool yet inside a capture: the unfused launch)
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


// Queries per sub-batch: the distance buffer is nq * cap * 17 bytes (16 int8 + 1 minimum
// per chunk, cap = n_probes * longest list); one workspace keeps it under 16 GB (env
// TINYKNN_WORKSPACE_GB; up to depth + 5 workspaces exist — sized for 288 GB of HBM: at
// 100M x 128 with 10 000 lists a 4 GB workspace cut a batch of 10 000 queries in two, and the
// list-major scan then found 5 instead of 10 queries per list to share a fetched chunk).
double workspace_bytes()
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
bool coarse_units(const tk_index *ix, int64_t nq)
{
    return ix->scan_mode != 1 && nq >= 16 && (double)nq / 4 * ix->center_chunks < 2.0e9;
}

int stage_tables(tk_index *ix, Work &w, const void *qpq_dev, int qpq_f64, int64_t nq,
                        hipStream_t st, Prof &pf, bool plain, TkSecond qpq2, int64_t row0)
{
    TRY(pf.mark(st));
    // 1. distance tables                                   fast_pq.py:186-222
    // (+ on the side, by the wave that holds a query's table: its limit for the plain kernel, and the
    //  descriptors of the list-major coarse scan — every query scans the one list of coded centres: no idle
    //  lanes; two short kernels less at the head of the front stream's chain)
    TkTablesExtra ex;
    if (plain) {        // per query: below which value clamp(plain sum) is the saturated value
        ex.qlim = w.qlim.as<int>() + row0;
        ex.lim_avx = ix->order == TK_ORDER_AVX;
        ex.lim_m_used = ex.lim_avx ? (ix->M & ~3) : ix->M;       // the AVX kernels read block pairs two at a time
        ex.lim_force = ix->opt_plain_limit;
    }
    if (coarse_units(ix, nq)) {
        ex.c_pair_off = w.c_pair_off.as<int>();
        ex.c_unit_prefix = w.c_unit_prefix.as<int>();
        ex.c_pair_q = w.c_pair_q.as<int>();
        ex.c_pair_f0 = w.c_pair_f0.as<int>();
        ex.c_chunks = (int)ix->center_chunks;
        ex.c_nq = nq;
    }
    tk_launch_build_tables(ix->pq_centers.as<float>(), ix->dq, ix->dpb, ix->f_order, qpq_dev,
                           qpq_f64, nq, ix->sqrt_nb, 0.0, 1, w.tables.as<uint8_t>() + (size_t)row0 * ix->M * 16, w.shift.p,
                           w.scale.as<double>(), st, qpq2, &ex);
    TRY(pf.mark(st));
    return TK_OK;
}

// the coarse scan as a job of the list-major kernel
TkScanJob coarse_job(const tk_index *ix, const Work &w, const Plan &p)
{
    TkScanJob j;
    j.codes = ix->center_codes.as<uint4>();
    j.tables = tables_of(w);
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
    j.tables = tables_of(w);
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
int head_chunks_of(const tk_index *ix, const Plan &p) { return head_chunks(ix, p); }

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
int plain_blocks() { return 512; }
// ... of a list-sharded rank's scans (eight batches in flight, each with its own replays: LDS space again — one rank
// through RCCL 20.4 -> 21.6 M queries/s on 320, same box)
int shard_plain_blocks() { return 320; }

// 2a. coarse scan = the scan of dtable.top(centers)          ivf.py:131, fast_pq.py:284-312
void launch_coarse_scan(tk_index *ix, Work &w, int64_t nq, const Plan &p, hipStream_t st,
                               const uint4 *tables)
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
int coarse_replay_probes(tk_index *ix, Work &w, const float *q_dev, int64_t nq, const Plan &p,
                                int64_t *probes_out, hipStream_t st, Prof &pf, TkSecond q2,
                                const TkSlotsOut *slots, int *slots_written)
{
    TRY(pf.mark(st));
    // positions of one list against a fresh heap are distinct labels: lane-per-query
    const bool fast_c = ix->heap_mode != 1 && ix->center_chunks * 16 <= 0xffffff;
    const bool lanes_c = fast_c && ix->heap_mode == 0 && p.rescore <= TK_LANES_MAX_R;
    if (fast_c && pair_replay(ix, nq, p.rescore)) {
        tk_launch_heap_replay_pair(w.cdist.as<uint4>(), ix->center_chunks, nq, w.cmins.as<uint8_t>(), p.ccap_min,
                                   ix->cslots_i.as<int>(), ix->cslots_i.as<int>() + 2, ix->cslots_l.as<int64_t>(), 1,
                                   nullptr, w.cheap_idx.as<int64_t>(), w.cheap_val.as<int32_t>(), p.rescore, 1, 1,
                                   nullptr, 0, st);
    } else if (fast_c && !lanes_c) {
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
    const int fused = tk_launch_rescore(q_dev, 0, ix->d, ix->active_centers.p, 0, ix->n_lists,
                                        w.cheap_idx.as<int64_t>(), p.rescore, nq, p.kc, 0, probes_out, nullptr, st,
                                        ix->opt_rescore_form, q2, TkSecond(), slots);
    if (slots_written) *slots_written = fused;
    return TK_OK;
}

// per-slot descriptors of the probed lists of `nq` queries
// (the same arguments as a structure: the coarse rescoring writes the descriptors itself where it can)
static TkSlotsOut slots_out(tk_index *ix, Work &w, const Plan &p, int *pair_count, const int *owner, int me,
                            bool plain)
{
    TkSlotsOut so;
    memset(&so, 0, sizeof so);        // (compared byte by byte with the workspace's device copy)
    so.n_lists = ix->n_lists;
    so.list_chunk_off = ix->list_chunk_off.as<int64_t>();
    so.list_n = ix->list_n.as<int64_t>();
    so.ids_off = ix->ids_off.as<int64_t>();
    so.slot_prefix = w.slot_prefix.as<int>();
    so.slot_chunk0 = w.slot_chunk0.as<int64_t>();
    so.slot_n = w.slot_n.as<int>();
    so.slot_label_off = w.slot_loff.as<int64_t>();
    so.repeat_flag = w.repeat_flag.as<unsigned char>();
    so.pair_count = pair_count;
    so.owner = owner;
    so.me = me;
    so.qlim = plain ? w.qlim.as<int>() : nullptr;
    so.R = head_rows(ix, p);
    so.slot_exact = plain ? w.slot_exact.as<int>() : nullptr;
    so.pair_count2 = plain ? w.p_count.as<int>() : nullptr;
    so.plain0 = plain ? w.plain0.as<int>() : nullptr;
    so.pair_count3 = plain ? w.h_count.as<int>() : nullptr;
    if (!so.qlim || !so.pair_count2 || !so.plain0 || !so.pair_count3) so.slot_exact = nullptr;
    return so;
}

void coarse_slots(tk_index *ix, Work &w, const int64_t *probes, int64_t nq, const Plan &p,
                         int *pair_count, const int *owner, int me, hipStream_t st, bool plain)
{
    const TkSlotsOut so = slots_out(ix, w, p, pair_count, owner, me, plain);
    tk_launch_make_slots(probes, nullptr, p.S, nq, so.n_lists, so.list_chunk_off, so.list_n, so.ids_off,
                         so.slot_prefix, so.slot_chunk0, so.slot_n, so.slot_label_off, so.repeat_flag,
                         so.pair_count, so.owner, so.me, st, so.qlim, so.R, so.slot_exact, so.pair_count2,
                         so.plain0, so.pair_count3);
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

int stage_coarse_rest(tk_index *ix, Work &w, const float *q_dev, int64_t nq, const Plan &p,
                             int *pair_count, const int *owner, int me, hipStream_t st, Prof &pf,
                             bool plain, TkSecond q2)
{
    // the slot descriptors: written by the coarse rescoring's own waves where that kernel can (p.S == p.kc)
    const TkSlotsOut so = slots_out(ix, w, p, pair_count, owner, me, plain);
    const TkSlotsOut *so_dev = nullptr;
    if (p.S == p.kc && p.kc <= 64 && w.slots_pinned) {      // (same box: 0.415 -> 0.408 ms per 10 000 queries, profiles/r04/ab_fused_slots.txt)
        // the structure lives in device memory, in a pool of immutable entries found by content
        // (api_internal.h): the upload's source is page-locked memory that outlives this call, so
        // the copy may become a node of a hipGraph, and no later batch rewrites what a graph reads
        int hit = -1;
        for (int i = 0; i < w.slots_n && hit < 0; i++)
            if (memcmp(&so, &w.slots_pinned[i], sizeof so) == 0) hit = i;
        bool upload = hit >= 0 && !ix->capturing && ((w.slots_need_upload >> hit) & 1u);
        if (hit < 0 && w.slots_n < TK_SLOTS_POOL) {
            hit = w.slots_n++;
            w.slots_pinned[hit] = so;
            upload = true;
        }
        if (hit >= 0) {
            TkSlotsOut *dst = w.slots_desc.as<TkSlotsOut>() + hit;
            if (upload) {
                HIPCHECK(hipMemcpyAsync(dst, &w.slots_pinned[hit], sizeof so, hipMemcpyHostToDevice, st));
                // (inside a capture the copy only runs when the graph does: a later stream-launched
                //  batch with this descriptor uploads it again)
                if (ix->capturing) w.slots_need_upload |= 1u << hit;
                else w.slots_need_upload &= ~(1u << hit);
            }
            so_dev = dst;
        }
    }
    int written = 0;
    TRY(coarse_replay_probes(ix, w, q_dev, nq, p, w.probes.as<int64_t>(), st, pf, q2, so_dev, &written));
    if (!written) coarse_slots(ix, w, w.probes.as<int64_t>(), nq, p, pair_count, owner, me, st, plain);
    return TK_OK;
}

// Stages 3b-4: the heap replay over the distance rows of queries [q0, q0 + nq) of the
// batch's slot arrays (dist/mins/heaps: `nq` rows starting at row 0), then the exact
// rescoring.  q_dev: row 0 = query q0.
// the queries the lane replay flagged (bound above the table's limit at the first plain block:
// plain_scan.hip): every probed list again with the exact kernel, then the replay again
// the flagged count is on its way to the page-locked word: the event the state machine polls behind
static void plain_verdict_event(tk_index *ix, Work &w, int64_t nq, hipStream_t st)
{
    if (w.plain_ev && !ix->capturing && hipEventRecord(w.plain_ev, st) == hipSuccess) {
        w.plain_pending = true;
        w.plain_nq = nq;
    }
}

// list_built: the lane replay compiled the list of flagged queries itself (tk_launch_heap_replay_lanes' flag_list); the
// count then reaches the host through the packed kernel behind the re-scan (plain_verdict_event there)
static void rescan_flagged(tk_index *ix, Work &w, int64_t q0, int64_t nq, const Plan &p, hipStream_t st,
                           bool list_built = false)
{
    int *list = w.flag_list.as<int>();
    if (!list_built) {
        tk_launch_flagged_list(w.repeat_flag.as<unsigned char>() + q0, nq, list, st, w.flag_host);
        plain_verdict_event(ix, w, nq, st);
    }
    tk_launch_scan_probes(ix->codes.as<uint4>(), ix->M, w.tables.as<uint4>() + q0 * ix->M, nq,
                          w.slot_prefix.as<int>() + q0 * (p.S + 1), w.slot_chunk0.as<int64_t>() + q0 * p.S,
                          p.S, (int)p.cap, w.dist.as<uint4>(), p.cap, w.mins.as<uint8_t>(), p.cap_min, 1,
                          ix->order, st, list);
}

// The tail behind a lane (or register-heap) replay that rode with the plain kernel: the queries it flagged — the lemma's
// check failed, or the probe list names a list twice — have been scanned again exactly (rescan_flagged) and are replayed
// from fresh heaps with the reference's duplicate test; the flagged count goes to the page-locked word the host polls.
// One wave per query either way; heaps of up to 129 entries take the register heap (a flagged query of the 100M x 128
// index is ~1 300 inserts over 6 250 blocks: the packed kernel's LDS heap with its label scan per insert made ONE such
// query a 1.2 ms tail behind a 1.9 ms replay of the other 9 999, profiles/r06/c5_flagged_tail.txt)
static void replay_flagged_tail(tk_index *ix, Work &w, int64_t nq, const Plan &p, const int *slot_prefix, const int *slot_n,
                                const int64_t *slot_loff, unsigned char *repeat_flag, hipStream_t st)
{
    if (p.R <= TK_PAIR_MAX_R && (ix->heap_mode == 0 || ix->heap_mode == 3) && ix->opt_pair_nq > 0) {
        (void)tk_launch_heap_replay_pair(w.dist.as<uint4>(), p.cap, nq, w.mins.as<uint8_t>(), p.cap_min, slot_prefix, slot_n,
                                         slot_loff, p.S, ix->ids.as<int64_t>(), w.heap_idx.as<int64_t>(),
                                         w.heap_val.as<int32_t>(), p.R, 1, 0, repeat_flag, (ix->labels24 && ix->opt_labels24) ? 2 : 0, st, nullptr,
                                         nullptr, nullptr, 1, w.flag_list.as<int>(), w.flag_host);
        return;
    }
    tk_launch_heap_replay_packed(w.dist.as<uint4>(), p.cap, nq, slot_prefix, slot_n, slot_loff,
                                 p.S, ix->ids.as<int64_t>(), w.heap_idx.as<int64_t>(),
                                 w.heap_val.as<int32_t>(), p.R, 1, 0, repeat_flag, -1, 1, st,
                                 w.flag_list.as<int>(), w.flag_host);
}

int stage_back(tk_index *ix, Work &w, const float *q_dev, int64_t q0, int64_t nq, int k,
                      const Plan &p, int64_t *out_dev, hipStream_t st, Prof &pf, bool plain,
                      TkSecond q2, TkSecond out2, int *plain_flag)
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
    // Lists far longer than the heap (100M x 128: a query's ten lists hold 6 250 blocks, its heap 111 entries):
    // staging every block through LDS is what the replay then costs (16 row-per-lane loads + 16 LDS writes per
    // segment and lane), while only the few blocks whose minimum passes the bound are ever looked at — the
    // LAZY form fetches just those (heap.hip; what made FlatTop's replay 4 x faster in round 4).  Short
    // lists keep the staged form: there most blocks of the first lists pass, and a dependent fetch each loses.
    // (TWIN form, labels that repeat: staged up to 40 heap sizes — its look-ahead for a candidate's row of the twin
    //  table needs the block in LDS; GloVe-shaped build(n_probes=2), 2 722 blocks against 111 entries: 15.0 M
    //  queries/s staged, 12.6 M lazy, profiles/r05/b2_twin_replay.txt)
    const double blocks_per_query = (double)p.S * (double)ix->total_chunks / (double)ix->n_lists;
    const int lazy = ix->opt_replay_lazy >= 0 ? ix->opt_replay_lazy
                                              : (blocks_per_query >= (twin_replay(ix, p) ? 40.0 : 8.0) * p.R);
    if (packed_ok && pair_replay(ix, nq, p.R)) {
        // one wave per query, heap in registers: position entries where labels are distinct (the queries whose probe
        // list names a list twice: the duplicate test on labels, as every query of an index whose labels repeat).
        // Behind the plain kernel: the lemma's check per query, as the lane kernel makes it (it does not depend on how the
        // duplicate test is made: below the limit the plain values ARE the reference's); the queries that fail it and
        // those flagged beforehand are scanned again exactly and replayed by the packed kernel with the duplicate test —
        // or, on a list-sharded rank (plain_flag: the codes are elsewhere), raise the batch's flag word
        if (tk_launch_heap_replay_pair(w.dist.as<uint4>(), p.cap, nq, w.mins.as<uint8_t>(), p.cap_min, slot_prefix, slot_n,
                                       slot_loff, p.S, ix->ids.as<int64_t>(), w.heap_idx.as<int64_t>(),
                                       w.heap_val.as<int32_t>(), p.R, 1, 0, (plain || ix->ids_unique) ? repeat_flag : nullptr,
                                       (ix->ids_unique ? 0 : 1) | ((ix->labels24 && ix->opt_labels24) ? 2 : 0), st, slot_exact, qlim,
                                       plain && !plain_flag ? w.flag_list.as<int>() : nullptr))
            return fail(TK_ERR_HIP, "hipMemsetAsync(flag list) failed");
        if (plain && plain_flag) {
            tk_launch_shard_flag_plain(repeat_flag, nq, plain_flag, st);
            tk_launch_heap_replay_packed(w.dist.as<uint4>(), p.cap, nq, slot_prefix, slot_n, slot_loff,
                                         p.S, ix->ids.as<int64_t>(), w.heap_idx.as<int64_t>(),
                                         w.heap_val.as<int32_t>(), p.R, 1, 0, repeat_flag, 1, 1, st);
        } else if (plain) {
            rescan_flagged(ix, w, q0, nq, p, st, true);
            replay_flagged_tail(ix, w, nq, p, slot_prefix, slot_n, slot_loff, repeat_flag, st);
            plain_verdict_event(ix, w, nq, st);
        }
    } else if (packed_ok && ix->ids_unique) {
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
                                             p.cap_min, nullptr, st, slot_exact, qlim, lazy,
                                             ix->opt_replay_count ? ix->replay_counters.as<unsigned long long>() : nullptr,
                                             nullptr, lanes && plain && !plain_flag ? w.flag_list.as<int>() : nullptr))
            return fail(TK_ERR_HIP, "hipFuncSetAttribute(LDS size) failed");
        if (plain && plain_flag) {
            tk_launch_shard_flag_plain(repeat_flag, nq, plain_flag, st);
        } else if (plain && lanes) {
            // flag 2 = the lane replay's "bound above the limit at the first plain block", flag 1 = a probe list that
            // names a list twice: exact re-scan of both kinds (the lane replay listed them), then ONE launch of the
            // packed kernel with the duplicate test from fresh heaps (where labels are distinct the test never fires)
            rescan_flagged(ix, w, q0, nq, p, st, true);
            replay_flagged_tail(ix, w, nq, p, slot_prefix, slot_n, slot_loff, repeat_flag, st);
            plain_verdict_event(ix, w, nq, st);
        } else if (plain) {
            rescan_flagged(ix, w, q0, nq, p, st);
            tk_launch_heap_replay_packed(w.dist.as<uint4>(), p.cap, nq, slot_prefix, slot_n, slot_loff,
                                         p.S, ix->ids.as<int64_t>(), w.heap_idx.as<int64_t>(),
                                         w.heap_val.as<int32_t>(), p.R, 1, 0, repeat_flag, 2, 0, st);
        }
        if (!(plain && lanes && !plain_flag))
        tk_launch_heap_replay_packed(w.dist.as<uint4>(), p.cap, nq, slot_prefix, slot_n, slot_loff,
                                     p.S, ix->ids.as<int64_t>(), w.heap_idx.as<int64_t>(),
                                     w.heap_val.as<int32_t>(), p.R, 1, 0, repeat_flag, 1, 1, st);
    } else if (packed_ok && twin_replay(ix, p)) {
        // repeating labels, every copy of a label with ONE value (IVF.build(n_probes >= 2)): one query per lane,
        // position entries, `insert`'s duplicate test decided from the twin table (heap.hip, TWIN form).  The
        // queries that probe a list twice (repeat_flag 1) and, with `plain`, those the lemma's check flags (2)
        // go to the packed kernel with the reference's scan of the labels
        TkTwins tw;
        tw.list = ix->twin_list.as<int32_t>();
        tw.off = ix->twin_off.as<int32_t>();
        tw.w = ix->twin_w;
        // (a list-sharded index replays its home queries: the batch's probe lists are the scan stage's)
        tw.probes = (ix->sharded && w.shard_probes ? w.shard_probes : w.probes.as<int64_t>()) + q0 * p.S;
        tw.bm_words = tk_lanes_twin_bm_words(ix->n_lists);
        if (tk_launch_heap_replay_lanes(w.dist.as<uint4>(), p.cap, nq, slot_prefix, slot_n,
                                        slot_loff, p.S, ix->ids.as<int64_t>(),
                                        w.heap_idx.as<int64_t>(), w.heap_val.as<int32_t>(), p.R, 1,
                                        0, repeat_flag, w.mins.as<uint8_t>(), p.cap_min, nullptr, st,
                                        slot_exact, qlim, lazy,
                                        ix->opt_replay_count ? ix->replay_counters.as<unsigned long long>() : nullptr,
                                        &tw, plain && !plain_flag ? w.flag_list.as<int>() : nullptr))
            return fail(TK_ERR_HIP, "hipFuncSetAttribute(LDS size) failed");
        if (plain && plain_flag) {
            tk_launch_shard_flag_plain(repeat_flag, nq, plain_flag, st);
            tk_launch_heap_replay_packed(w.dist.as<uint4>(), p.cap, nq, slot_prefix, slot_n, slot_loff,
                                         p.S, ix->ids.as<int64_t>(), w.heap_idx.as<int64_t>(),
                                         w.heap_val.as<int32_t>(), p.R, 1, 0, repeat_flag, 1, 1, st);
        } else if (plain) {         // (flags 1 and 2 alike: re-scan, one launch of the packed kernel)
            rescan_flagged(ix, w, q0, nq, p, st, true);
            replay_flagged_tail(ix, w, nq, p, slot_prefix, slot_n, slot_loff, repeat_flag, st);
            plain_verdict_event(ix, w, nq, st);
        } else {
            tk_launch_heap_replay_packed(w.dist.as<uint4>(), p.cap, nq, slot_prefix, slot_n, slot_loff,
                                         p.S, ix->ids.as<int64_t>(), w.heap_idx.as<int64_t>(),
                                         w.heap_val.as<int32_t>(), p.R, 1, 0, repeat_flag, 1, 1, st);
        }
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
// ... and of the plain kernel beside the other batches' kernels: what the pipelined batch runs out of is LDS SPACE
//  — two paired lane replays, the coarse replay and 512 plain workgroups ask for more than the chip's
// 41 MB, and whoever comes last waits.  The plain kernel is as fast on 256 workgroups as on 512 there (its waves wait
// on latencies, not on each other), and with the lane replay's 8-block segments (heap.hip) everything fits: same
// box, headline batch 25.4-25.8 M queries/s against 24.4-24.7 M (320 or 256; 192: 25.0; 512 with 8-block segments:
// 24.0-25.4, bimodal).  Long launches keep 512: 100M x 128 loses 5 % on 320, build(n_probes=2) 4 %.
// (256 against 320, seven runs each in turn on one box: 25.5-25.75 M every time against 25.4-25.6 M with one run
//  of 23.4 M — a whole process in a slow mode, one in six on another box too.)
static int plain_blocks_pipelined(double est_units) { return est_units > 6.0e6 ? 512 : 256; }

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
        if (tk_launch_scan_plain(plain_job(ix, *prev->w, prev->p), M, ix->order,
                                 plain_blocks_pipelined((double)prev->nq * prev->p.S / 4.0 *
                                                        ((double)ix->total_chunks / (double)ix->n_lists)), st))
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

int flush_pending(tk_index *ix)
{
    int r = TK_OK;
    if (ix->held) r = launch_held(ix);
    while (!ix->pending.empty() && r == TK_OK) r = pipeline_advance(ix, true);
    for (Pending *b : ix->pending) delete b;
    ix->pending.clear();
    return r;
}

// The internal streams of the pipelined mode — one "front" stream and up to eight replay streams — belong
// to the PROCESS (per device), not to an index.  HIP maps streams onto its four hardware queues in creation
// order: the streams a second index created for itself landed on the queues of the first one's — its
// front stream on the caller's queue — and that index ran 10 % slower than alone (the sweep points of
// bench.py behind the headline index: 7.2 against 8.0 M queries/s, scripts/r04_b2_sweep_check.py).  Shared
// streams are in-order, so indexes used in turn or from several threads only see extra ordering.
//
// The front stream's chain (table build, coarse replay + rescoring, descriptors: short kernels each
// waiting for the one before) is the pipeline's critical path once the scans overlap: its kernels go
// first when CU slots free up.  Same box, ms per 10 000 queries: 0.430 default priority, 0.417 high,
// 0.443 low; the replay streams high as well: 0.425 (profiles/r04/ab_front_prio.txt)
struct SharedStreams {
    hipStream_t front = nullptr;
    std::vector<hipStream_t> lat;
};
static std::mutex g_streams_mu;
static SharedStreams g_streams[64];     // by device ordinal

static hipError_t shared_front_stream(hipStream_t *st)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_streams_mu);
    SharedStreams &S = g_streams[dev & 63];
    if (!S.front) {
        int lo = 0, hi = 0;
        e = hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (e != hipSuccess) return e;
        e = hipStreamCreateWithPriority(&S.front, hipStreamNonBlocking, hi);
        if (e != hipSuccess) return e;
    }
    *st = S.front;
    return hipSuccess;
}

static hipError_t shared_replay_stream(int i, hipStream_t *st)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_streams_mu);
    SharedStreams &S = g_streams[dev & 63];
    while ((int)S.lat.size() <= i) {
        hipStream_t n;
        e = hipStreamCreateWithFlags(&n, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        S.lat.push_back(n);
    }
    *st = S.lat[(size_t)i];
    return hipSuccess;
}

// The process-wide internal streams, for a host that runs its own stage pipeline over them (multi_gpu.py: the
// list-sharded batches' coarse stages / exchanges + replays; role 0 = the high-priority front stream, role 1 =
// replay stream i < 8).  The same streams tk_index_query_batch_dev uses in pipelined mode: HIP maps streams onto
// four hardware queues in creation order, and a host that created four more of its own would share them.
extern "C" void *tk_shared_stream(int role, int i)
{
    if (require_gpu() != TK_OK) return nullptr;
    hipStream_t st = nullptr;
    const hipError_t e = role == 0 ? shared_front_stream(&st) : (i >= 0 && i < 8 ? shared_replay_stream(i, &st) : hipErrorInvalidValue);
    if (e != hipSuccess) {
        (void)fail(TK_ERR_HIP, "tk_shared_stream: stream creation failed (or a replay stream index beyond 7)");
        return nullptr;
    }
    return (void *)st;
}

// Pipelined mode, first half of enqueuing a batch: internal streams and events exist, the batch has
// its workspace and streams, and `stt` — the stream its table build will run on — waits for the
// caller's work so far and for the workspace's previous batch.
static int pipe_begin(tk_index *ix, Pending &b, hipStream_t caller, hipStream_t &stt_out)
{
    Work &w = *b.w;
    while ((int)ix->lat_streams.size() < ix->depth) {
        hipStream_t st;
        HIPCHECK(shared_replay_stream((int)ix->lat_streams.size(), &st));
        ix->lat_streams.push_back(st);
    }
    if (!ix->front_stream) HIPCHECK(shared_front_stream(&ix->front_stream));
    b.sf = ix->front_stream;
    b.sl = ix->lat_streams[ix->calls % (uint64_t)ix->depth];
    if (ix->capturing) {      // (stream-launched, this arrangement loses a fifth: 20.7 against 25.7 M queries/s, same box)
        // A captured pipeline is replayed by ROCm's graph executor on TWO hardware queues whatever it captured
        // (profiles/r05/hipgraph_replay_queues.txt), one of them 96 % busy with four captured chains.  Captured on THREE —
        // the front chain on the caller's stream in front of the scans, the two replay streams as they are — the same
        // executor gives 16.2 instead of 14.7 M queries/s (identical rows); on two (one replay stream) 13.9 M; with three
        // or four replay streams less (profiles/r06/hipgraph_captured_streams.txt).
        // TINYKNN_GRAPH_STREAMS (A/B): 4 = as stream-launched, 3 (default) = front chain on the caller's stream, 2 = ... and
        // one replay stream
        static const int gs = [] { const char *e = getenv("TINYKNN_GRAPH_STREAMS"); return e ? atoi(e) : 3; }();
        if (gs == 2 || gs == 3) b.sf = caller;
        if (gs == 2 || gs == 5) b.sl = ix->lat_streams[0];      // (5: front stream as it is, one replay stream)
    }
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
    // (repeating labels — IVF.build(n_probes >= 2) — pair up where the TWIN form of the lane replay applies; the
    //  hash-set form runs 32 queries and 70 KB of LDS per wave, two waves per CU, and a doubled batch would not
    //  fit the chip in one round of waves: 5.1 M queries/s paired against 7.7 M alone,
    //  profiles/r04/bench_full_first.json)
    if (ix->depth > 1 && ix->coalesce == 2 && (ix->ids_unique || twin_replay(ix, p)) && nq >= 1 && nq <= ms)
        return coalesce_call(ix, p, q_dev, q_pq_dev, q_pq_is_f64, nq, k, n_probes, pass_1, out_ids_dev,
                             out_ids_pinned, done_ev, caller);
    if (ix->held) TRY(launch_held(ix));
    // a batch beyond one workspace goes in EQUAL parts (30 000 queries at 100M x 128, 22 500 to a workspace: 2 x 15 000,
    // not 22 500 + 7 500 — the small rest fell below the list-major scan's threshold and took the query-major kernel)
    const int64_t parts = nq > ms ? (nq + ms - 1) / ms : 1;
    const int64_t part = (nq + parts - 1) / parts;
    for (int64_t o = 0; o < nq; o += part) {
        int64_t sub = nq - o < part ? nq - o : part;
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
    if (!ix->front_stream && shared_front_stream(&ix->front_stream) != hipSuccess) return nullptr;
    return ix->front_stream;
}

// calls whose last stage has not been enqueued yet
extern "C" int tk_index_pending(tk_index *ix)
{
    if (!ix) return 0;
    IXLOCK(ix);
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
    // the workspaces and replay streams are taken in the order of the calls since the last
    // tk_index_set_pipeline / tk_index_quiesce: a capture that follows repeats the warm-up's calls ON THE
    // WORKSPACES the warm-up sized (a call that met a fresh workspace would hipMalloc inside the capture
    // and invalidate it: 5 captured calls at depth 2 behind 5 warm-up calls used to land on workspaces 5, 6, 0, 1, 2)
    ix->calls = 0;
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
    ARGCHECK(mode >= 0 && mode <= 3, "mode");
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
    case TK_OPT_PAIR_NQ:
        ARGCHECK(value >= 0, "TK_OPT_PAIR_NQ: >= 0");
        ix->opt_pair_nq = value;
        return TK_OK;
    case TK_OPT_LABELS24:
        ix->opt_labels24 = value != 0;
        return TK_OK;
    case TK_OPT_REPLAY_COUNT:
        ix->opt_replay_count = value != 0;
        if (value) {
            TRY(ix->replay_counters.ensure(64));
            HIPCHECK(hipMemset(ix->replay_counters.p, 0, 64));
        }
        return TK_OK;
    case TK_OPT_REPLAY_LAZY:
        ARGCHECK(value >= -1 && value <= 1, "TK_OPT_REPLAY_LAZY: -1, 0 or 1");
        ix->opt_replay_lazy = value;
        return TK_OK;
    case TK_OPT_REPLAY_TWIN:
        ARGCHECK(value == 0 || value == 1, "TK_OPT_REPLAY_TWIN: 0 or 1");
        ix->opt_replay_twin = value;
        return TK_OK;
    case TK_OPT_TWIN_VOUCH:
        ARGCHECK(value == 0 || value == 1, "TK_OPT_TWIN_VOUCH: 0 or 1");
        ix->twin_vouched = value != 0;
        return TK_OK;
    default:
        return fail(TK_ERR_ARG, "bad argument: unknown option");
    }
}

extern "C" int tk_index_set_scan_mode(tk_index *ix, int mode)
{
    IXLOCK(ix);
    ARGCHECK(ix, "null index");
    ARGCHECK(mode >= 0 && mode <= 3, "mode");
    TRY(flush_pending(ix));
    ix->scan_mode = mode;
    return TK_OK;
}

extern "C" int tk_index_set_plain_scan(tk_index *ix, int mode)
{
    IXLOCK(ix);
    ARGCHECK(ix, "null index");
    ARGCHECK(mode >= 0 && mode <= 3, "mode");
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

// TK_OPT_REPLAY_COUNT: what the lane replays of the probed lists did since the option was set / the last call
// (synchronises; zeroes the counters): out4 = insert rounds summed over the waves, the most rounds any wave ran
// (the kernel's critical path: a round is one dependent insert step of a wave), waves, 16-block segments walked.
extern "C" int tk_index_replay_stats(tk_index *ix, int64_t *out4)
{
    IXLOCK(ix);
    ARGCHECK(ix && out4, "null index / buffer");
    for (int i = 0; i < 4; i++) out4[i] = 0;
    if (!ix->replay_counters.p) return TK_OK;
    TRY(flush_pending(ix));
    HIPCHECK(hipDeviceSynchronize());
    unsigned long long v[4];
    HIPCHECK(hipMemcpy(v, ix->replay_counters.p, sizeof v, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemset(ix->replay_counters.p, 0, 64));
    for (int i = 0; i < 4; i++) out4[i] = (int64_t)v[i];
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
