// api_shard.hip — list-sharded entry points (tk_index_shard_*): what multi_gpu.py calls between the
// collectives of one sharded batch.  (Split from api.hip in round 4.)
#include "api_internal.h"

// ---------------------------------------------------------------------------
// list-sharded batch = shard_scan -> all-to-all of the send buffer -> shard_finish
static int reserve_shard(tk_index *ix, Work &w, int64_t nq, int64_t qh, const Plan &p)
{
    const int M = ix->M;
    TRY(reserve_slots_pool(w));
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

static bool shard_plain_possible(const tk_index *ix, const Plan &p)
{
    if (ix->plain_mode == 1 || !plain_env_on() || !ix->sharded || p.S < 2 || !tk_plain_fits(ix->M)) return false;
    return ix->ids_unique || twin_replay(ix, p) || ix->plain_mode == 2;
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
    TRY(w.h_pair_q.ensure(((size_t)nq + 4 * L + 4) * 4));       // (one-phase form: a head pair per query)
    TRY(w.h_pair_f0.ensure(((size_t)nq + 4 * L + 4) * 4));
    TRY(w.slot_exact.ensure((size_t)nq * 4));
    TRY(w.plain0.ensure((size_t)nq * 4));
    return TK_OK;
}

// Coarse stage sharded by HOME rank: tables for all nq queries (every rank scores segments of
// every query), coarse scan + replay + rescoring only for this rank's ceil(nq/world) home
// queries; the caller all-gathers the probe lists and hands them to tk_index_shard_scan_dev.
static int shard_coarse_impl(tk_index *ix, int slot, const float *q_dev, const void *q_pq_dev, int q_pq_is_f64,
                             int64_t nq, int k, int n_probes, int pass_1, int64_t *probes_home_dev, void *stream)
{
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    int64_t qh = 0;
    TRY(shard_args(ix, slot, nq, 1, p, qh));
    ARGCHECK(probes_home_dev, "probes buffer");
    Work &w = ix->works[(size_t)slot];
    hipStream_t st = (hipStream_t)stream;
    TRY(reserve_shard(ix, w, nq, qh, p));
    Prof pf;
    // (the limits C of all nq tables ride in the table launch's shadow where a plain form may follow)
    const bool limits = shard_plain_possible(ix, p);
    if (limits) TRY(reserve_shard_plain(ix, w, nq, p));
    const int64_t q0 = (int64_t)ix->rank * qh;
    int64_t nqh = nq - q0;
    nqh = nqh < 0 ? 0 : (nqh > qh ? qh : nqh);
    TRY(stage_tables(ix, w, q_pq_dev, q_pq_is_f64, nq, st, pf, limits));
    // rows past nq: list 0 (never read by a consumer; defined for the all-gather)
    HIPCHECK(hipMemsetAsync(probes_home_dev, 0, (size_t)qh * p.kc * 8, st));
    if (nqh > 0) {
        if (coarse_units(ix, nqh))     // identity pairs of the home range (stage_tables wrote those of all nq)
            tk_launch_identity_pairs(nqh, (int)ix->center_chunks, w.c_pair_off.as<int>(),
                                     w.c_unit_prefix.as<int>(), w.c_pair_q.as<int>(),
                                     w.c_pair_f0.as<int>(), st);
        launch_coarse_scan(ix, w, nqh, p, st, w.tables.as<uint4>() + q0 * ix->M);
        TRY(coarse_replay_probes(ix, w, q_dev + q0 * ix->d, nqh, p, probes_home_dev, st, pf));
    }
    HIPCHECK(hipGetLastError());
    return TK_OK;
}

extern "C" int tk_index_shard_coarse_dev(tk_index *ix, int slot, const float *q_dev,
                                         const void *q_pq_dev, int q_pq_is_f64, int64_t nq, int k,
                                         int n_probes, int pass_1, int64_t *probes_home_dev,
                                         void *stream)
{
    IXLOCK(ix);
    return shard_coarse_impl(ix, slot, q_dev, q_pq_dev, q_pq_is_f64, nq, k, n_probes, pass_1, probes_home_dev, stream);
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
    tk_launch_scan_units(ix->codes.as<uint4>(), ix->M, tables_of(w), nq, p.S, ix->n_lists,
                         ix->local_chunk_off.as<int64_t>(), w.u_pair_off.as<int>(),
                         w.u_unit_prefix.as<int>(), w.u_pair_q.as<int>(), w.u_pair_f0.as<int>(),
                         (uint4 *)send_dev, 0, w.smins.as<uint8_t>(), 0, 1, ix->order, 768, st);
    w.shard_probes = probes;
    w.shard_nq = nq;
    w.shard_capacity = capacity;
    w.shard_plain = false;
    HIPCHECK(hipGetLastError());
    return TK_OK;
}

static bool shard_one_phase_possible(const tk_index *ix, const Plan &p, int64_t capacity)
{
    const int64_t tail = (int64_t)ix->world * capacity - ix->max_list_chunks;
    return shard_plain_possible(ix, p) && (ix->ids_unique || twin_replay(ix, p)) && tail >= 0 && p.R <= TK_LANES_MAX_R &&
           ix->heap_mode == 0 && p.cap * 16 <= 0xffffff;
}

// ---- ... and with ONE byte per query exchanged first, for data on which the check at home would fail ----
// The one-phase scan is optimistic: one home query in 20 000 of the 100M x 128 index meets its first plain
// block with a bound above its table's limit, and a sharded batch of 120 000 queries then fails as a whole.
// tk_index_shard_scan_head_dev + all-reduce(MIN, uint8) + tk_index_shard_scan_plain_dev(bound_dev) decide per
// query BEFORE the scan, at the price of the rows the exact kernel scores anyway: the HEAD of every first
// probed list this rank owns (two heap sizes of rows) goes exactly into send_dev, the bound after it is
// replayed by value on the spot (bound_dev[nq]: order key, 255 where the first list lies elsewhere), the
// ranks min-reduce the byte, and the scan proper keeps every query whose bound is above its limit on the
// exact kernel.  Against the two-phase form (tk_index_shard_scan_first_dev: WHOLE first lists exactly, the
// bound after them) phase 1 touches 14 chunks per query instead of a whole list (313 at 25M x 128 / 5 000
// lists).  Where the one-phase form does not apply this call is tk_index_shard_scan_dev (bound_dev = 255)
// and the plain call behind it does nothing.
extern "C" int tk_index_shard_scan_head_dev(tk_index *ix, int slot, const float *q_dev,
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
    Work &w = ix->works[(size_t)slot];
    hipStream_t st = (hipStream_t)stream;
    if (!shard_one_phase_possible(ix, p, capacity)) {
        HIPCHECK(hipMemsetAsync(bound_dev, 0xff, (size_t)nq, st));
        TRY(tk_index_shard_scan_dev(ix, slot, q_dev, q_pq_dev, q_pq_is_f64, nq, k, n_probes, pass_1,
                                    probes_all_dev, capacity, send_dev, flag_dev, stream));
        w.shard_head = true;
        return TK_OK;
    }
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
        TRY(stage_tables(ix, w, q_pq_dev, q_pq_is_f64, nq, st, pf, true));
        launch_coarse_scan(ix, w, nq, p, st);
        TRY(stage_coarse_rest(ix, w, q_dev, nq, p, nullptr, owner, ix->rank, st, pf));
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
    // heads of the first slots this rank owns: exact, straight into the send buffer
    const int64_t *lco = ix->local_chunk_off.as<int64_t>();
    const int hc = head_chunks_of(ix, p);
    TkPairSet ex{w.u_count.as<int>(), w.u_cursor.as<int>(), w.u_pair_off.as<int>(),
                 w.u_unit_prefix.as<int>(), w.u_pair_q.as<int>(), w.u_pair_f0.as<int>()};
    TkPairSet pl{w.p_count.as<int>(), w.p_cursor.as<int>(), w.p_pair_off.as<int>(),
                 w.p_unit_prefix.as<int>(), w.p_pair_q.as<int>(), w.p_pair_f0.as<int>(),
                 w.p_unit_desc.as<int>(), plain_k(ix, nq, p)};
    TkPairSet hd{w.h_count.as<int>(), w.h_cursor.as<int>(), w.h_pair_off.as<int>(),
                 w.h_unit_prefix.as<int>(), w.h_pair_q.as<int>(), w.h_pair_f0.as<int>()};
    tk_launch_shard_count_first(probes, p.S, nq, ix->n_lists, owner, ix->rank, w.h_count.as<int>(), st);
    tk_launch_pairs_scan3(ex, pl, hd, lco, ix->n_lists, hc, st);      // (sets 0 and 1 are empty here)
    tk_launch_shard_pairs_fill(probes, p.S, nq, ix->n_lists, owner, ix->rank, w.spos.as<int>(),
                               w.h_pair_off.as<int>(), w.h_cursor.as<int>(), w.h_pair_q.as<int>(),
                               w.h_pair_f0.as<int>(), st, 0, 1);
    TkScanJob hj, none;
    memset(&none, 0, sizeof none);
    hj.codes = ix->codes.as<uint4>();
    hj.tables = tables_of(w);
    hj.list_chunk_off = lco;
    hj.n_lists = (int)ix->n_lists;
    hj.unit_prefix = w.h_unit_prefix.as<int>();
    hj.pair_off = w.h_pair_off.as<int>();
    hj.pair_q = w.h_pair_q.as<int>();
    hj.pair_f0 = w.h_pair_f0.as<int>();
    hj.dist = (uint4 *)send_dev;
    hj.cap = 0;
    hj.mins = w.smins.as<uint8_t>();
    hj.min_stride = 0;
    hj.max_chunks = hc;
    tk_launch_scan_units2(none, none, ix->M, ix->order, 512, st, &hj, 0);
    // the bound after those heads, by value (ivf.py:137-152 over the list's first rows)
    tk_launch_shard_first_bound(probes, w.slot_prefix.as<int>(), w.slot_n.as<int>(), p.S, nq, ix->n_lists,
                                owner, ix->rank, w.spos.as<int>(), (const uint4 *)send_dev,
                                w.smins.as<uint8_t>(), p.R, bound_dev, st, hc);
    w.shard_probes = probes;
    w.shard_nq = nq;
    w.shard_capacity = capacity;
    w.shard_plain = false;
    w.shard_head = true;
    HIPCHECK(hipGetLastError());
    return TK_OK;
}

// ---- the scan in ONE phase with the matrix-core kernel, as the unsharded pipeline runs it ----
// tk_index_shard_scan_dev scores every owned segment exactly; the two-phase form below first learns the
// bound B1 after every query's first list (a replay by value on its owner + a MIN all-reduce: one more
// kernel chain and one more collective on the batch's critical path).  The unsharded index does neither:
// it keeps only the first rows a query scans (the head of its first list: 2 heap sizes) on the exact
// kernel, runs everything else as plain sums on the matrix cores, and lets the REPLAY check the lemma's
// condition per query (bound at the first plain block <= the table's limit C; plain_scan.hip).  The
// same here: every rank scores its segments of every query that way straight into the send buffer —
// head chunks exact, the rest plain — and the home rank's replay (tk_index_shard_finish_dev) does the
// check.  A query that fails it cannot be scanned again at home (the codes are elsewhere): bit 4 of the
// batch's flag word is raised, it travels with the ids, and the caller repeats the batch through
// tk_index_shard_scan_dev.  On the bench batches no query fails.  Same ids as every other form.
// Applies where tk_index_shard_plain says yes AND the labels are distinct AND world * capacity holds the
// longest list (the plain kernel scores an overflowed segment to the buffer's tail); otherwise this
// call is tk_index_shard_scan_dev.
extern "C" int tk_index_shard_scan_plain_dev(tk_index *ix, int slot, const float *q_dev,
                                             const void *q_pq_dev, int q_pq_is_f64, int64_t nq, int k,
                                             int n_probes, int pass_1, const int64_t *probes_all_dev,
                                             int64_t capacity, void *send_dev, int *flag_dev,
                                             const uint8_t *bound_dev, void *stream)
{
    IXLOCK(ix);
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    int64_t qh = 0;
    TRY(shard_args(ix, slot, nq, capacity, p, qh));
    ARGCHECK(send_dev && flag_dev, "send/flag buffers");
    const int64_t tail = (int64_t)ix->world * capacity - ix->max_list_chunks;      // (>= 0 where the form applies)
    Work &w = ix->works[(size_t)slot];
    if (!shard_one_phase_possible(ix, p, capacity)) {
        // (behind tk_index_shard_scan_head_dev, which then was tk_index_shard_scan_dev itself: nothing is owed)
        if (bound_dev && w.shard_head) {
            w.shard_head = false;
            return TK_OK;
        }
        return tk_index_shard_scan_dev(ix, slot, q_dev, q_pq_dev, q_pq_is_f64, nq, k, n_probes, pass_1,
                                       probes_all_dev, capacity, send_dev, flag_dev, stream);
    }
    hipStream_t st = (hipStream_t)stream;
    const bool behind_head = bound_dev != nullptr;
    ARGCHECK(!behind_head || (w.shard_head && w.shard_probes && w.shard_nq == nq && w.shard_capacity == capacity),
             "bound_dev: tk_index_shard_scan_head_dev of this slot (same nq and capacity) comes first");
    w.shard_head = false;
    TRY(reserve_shard(ix, w, nq, qh, p));
    TRY(reserve_shard_plain(ix, w, nq, p));
    TRY(w.smins.ensure((size_t)ix->world * capacity + 16));
    TRY(w.usage.ensure((size_t)ix->world * 2 * 8));
    Prof pf;
    const int *owner = ix->owner.as<int>();
    const int64_t *probes = probes_all_dev;
    if (behind_head) {
        // the head call left tables, limits, probe lists and positions; queries whose bound after the head
        // is above their table's limit leave the plain path (all their lists exact) — the rest is the
        // one-phase scan, which now cannot fail its check at home
        probes = w.shard_probes;
        tk_launch_shard_mask_limits(bound_dev, nq, w.qlim.as<int>(), st);
        coarse_slots(ix, w, probes, nq, p, w.u_count.as<int>(), owner, ix->rank, st, true);
    } else if (probes) {
        // tables and their limits: tk_index_shard_coarse_dev; the probe lists arrive gathered
        coarse_slots(ix, w, probes, nq, p, w.u_count.as<int>(), owner, ix->rank, st, true);
    } else {
        TRY(stage_tables(ix, w, q_pq_dev, q_pq_is_f64, nq, st, pf, true));
        launch_coarse_scan(ix, w, nq, p, st);
        TRY(stage_coarse_rest(ix, w, q_dev, nq, p, w.u_count.as<int>(), owner, ix->rank, st, pf, true));
        probes = w.probes.as<int64_t>();
    }
    if (!behind_head) {
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
    // three pair sets over the lists this rank owns (whole lists exact / plain tiles / heads), the
    // records' row offsets = positions in the send buffer
    const int64_t *lco = ix->local_chunk_off.as<int64_t>();
    TkPairSet ex{w.u_count.as<int>(), w.u_cursor.as<int>(), w.u_pair_off.as<int>(),
                 w.u_unit_prefix.as<int>(), w.u_pair_q.as<int>(), w.u_pair_f0.as<int>()};
    TkPairSet pl{w.p_count.as<int>(), w.p_cursor.as<int>(), w.p_pair_off.as<int>(),
                 w.p_unit_prefix.as<int>(), w.p_pair_q.as<int>(), w.p_pair_f0.as<int>(),
                 w.p_unit_desc.as<int>(), plain_k(ix, nq, p)};
    TkPairSet hd{w.h_count.as<int>(), w.h_cursor.as<int>(), w.h_pair_off.as<int>(),
                 w.h_unit_prefix.as<int>(), w.h_pair_q.as<int>(), w.h_pair_f0.as<int>()};
    const int hc = head_chunks_of(ix, p);
    tk_launch_unit_pairs2(nq, probes, p.S, ix->n_lists, lco, w.slot_prefix.as<int>(), w.slot_exact.as<int>(),
                          ex, pl, hd, hc, st, w.spos.as<int>(), owner, ix->rank, (int)tail);
    TkScanJob lj;
    lj.codes = ix->codes.as<uint4>();
    lj.tables = tables_of(w);
    lj.list_chunk_off = lco;
    lj.n_lists = (int)ix->n_lists;
    lj.unit_prefix = w.u_unit_prefix.as<int>();
    lj.pair_off = w.u_pair_off.as<int>();
    lj.pair_q = w.u_pair_q.as<int>();
    lj.pair_f0 = w.u_pair_f0.as<int>();
    lj.dist = (uint4 *)send_dev;
    lj.cap = 0;
    lj.mins = w.smins.as<uint8_t>();
    lj.min_stride = 0;
    TkScanJob pj = lj, hj = lj, none;
    memset(&none, 0, sizeof none);
    pj.unit_prefix = w.p_unit_prefix.as<int>();
    pj.pair_off = w.p_pair_off.as<int>();
    pj.pair_q = w.p_pair_q.as<int>();
    pj.pair_f0 = w.p_pair_f0.as<int>();
    pj.unit_desc4 = w.p_unit_desc.as<int>();
    hj.unit_prefix = w.h_unit_prefix.as<int>();
    hj.pair_off = w.h_pair_off.as<int>();
    hj.pair_q = w.h_pair_q.as<int>();
    hj.pair_f0 = w.h_pair_f0.as<int>();
    hj.max_chunks = hc;
    // (plain first: the exact kernel then overwrites the head chunks of the lists in head mode)
    if (tk_launch_scan_plain(pj, ix->M, ix->order, shard_plain_blocks(), st))
        return fail(TK_ERR_HIP, "scan_plain_kernel: LDS attribute / unsupported M");
    tk_launch_scan_units2(lj, none, ix->M, ix->order, 768, st, &hj, ix->opt_scan_form);
    w.shard_probes = probes;
    w.shard_nq = nq;
    w.shard_capacity = capacity;
    w.shard_plain = true;
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
extern "C" int tk_index_shard_plain(tk_index *ix, int k, int n_probes, int pass_1)
{
    IXLOCK(ix);
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    return shard_plain_possible(ix, p) ? 1 : 0;
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
    // the limits C of all nq tables (tk_index_shard_coarse_dev computed them beside its tables)
    if (!probes_all_dev)
        tk_launch_table_limits(tables_of(w), ix->M, ix->order, nq, w.qlim.as<int>(), st, ix->opt_plain_limit);
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
    tk_launch_scan_units(ix->codes.as<uint4>(), ix->M, tables_of(w), nq, p.S, ix->n_lists,
                         ix->local_chunk_off.as<int64_t>(), w.u_pair_off.as<int>(),
                         w.u_unit_prefix.as<int>(), w.u_pair_q.as<int>(), w.u_pair_f0.as<int>(),
                         (uint4 *)send_dev, 0, w.smins.as<uint8_t>(), 0, 1, ix->order, 768, st);
    w.shard_probes = probes;
    w.shard_nq = nq;
    w.shard_capacity = capacity;
    w.shard_plain = false;
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
    pj.tables = tables_of(w);
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
    if (tk_launch_scan_plain(pj, ix->M, ix->order, shard_plain_blocks(), st))
        return fail(TK_ERR_HIP, "scan_plain_kernel: LDS attribute / unsupported M");
    tk_launch_scan_units(ix->codes.as<uint4>(), ix->M, tables_of(w), nq, p.S, ix->n_lists,
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

// records_dev: the blocks below the bound as 20-byte records grouped by home rank.  region_records == 0: compact
// (variable splits: counts_dev read on the host before the all-to-all; flag_dev / acc_dev may be NULL).
// region_records >= 1: the records of home rank h at records_dev[h * region_records ...] (room for world *
// region_records records): the all-to-all that follows has EQUAL splits, so no rank has to read a count on the
// host before it can enqueue it — counts_dev[0, world) travel beside the records and the home rank reads them on
// the device (tk_index_shard_finish_filtered_dev with counts_recv_dev).  More than region_records records for one
// home rank: the rest is dropped and *flag_dev |= 1, the overflow flag of the batch (the caller repeats it with
// larger regions, as with `capacity`).  acc_dev (or NULL): three int64 the caller keeps across batches — [0] =
// largest counts_dev[h] seen (atomic max: what the regions have to hold), [1] += records, [2] += blocks scored.
extern "C" int tk_index_shard_filter_dev(tk_index *ix, int slot, int64_t nq, int k, int n_probes,
                                         int pass_1, int64_t capacity, const void *scan_dev,
                                         const uint8_t *bound_dev, int32_t *counts_dev,
                                         int32_t *records_dev, int64_t region_records,
                                         int *flag_dev, int64_t *acc_dev, void *stream)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->sharded, "not a list-sharded index");
    ARGCHECK(region_records == 0 || (region_records >= 1 && region_records * ix->world < (1ll << 31) && flag_dev),
             "region_records (x world must stay below 2^31) / flag buffer");
    if (region_records == 0)
        return shard_filter_impl(ix, slot, nq, k, n_probes, pass_1, capacity, scan_dev, bound_dev, counts_dev,
                                 records_dev, 0, nullptr, nullptr, stream);
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

// counts_recv_dev == NULL: records_dev holds n_records compact records.  Otherwise: `world` regions of region_records
// records as the equal-split all-to-all delivered them (region s from source rank s); counts_recv_dev[s] of them are
// real (the all-to-all of the senders' counts_dev[0, world), on the device: no host synchronisation anywhere in the
// batch); n_records is ignored.
extern "C" int tk_index_shard_finish_filtered_dev(tk_index *ix, int slot, const float *q_dev,
                                                  int64_t nq, int k, int n_probes, int pass_1,
                                                  const int32_t *records_dev, int64_t n_records,
                                                  const int32_t *counts_recv_dev, int64_t region_records,
                                                  int64_t *out_ids_home_dev, int *flag_dev,
                                                  void *stream)
{
    IXLOCK(ix);
    ARGCHECK(ix && ix->sharded, "not a list-sharded index");
    if (!counts_recv_dev)
        return shard_finish_filtered_impl(ix, slot, q_dev, nq, k, n_probes, pass_1, records_dev, n_records,
                                          nullptr, 0, out_ids_home_dev, flag_dev, stream);
    ARGCHECK(region_records >= 1 && region_records * ix->world < (1ll << 31), "counts / region_records");
    return shard_finish_filtered_impl(ix, slot, q_dev, nq, k, n_probes, pass_1, records_dev,
                                      region_records * ix->world, counts_recv_dev, region_records,
                                      out_ids_home_dev, flag_dev, stream);
}

extern "C" int tk_index_shard_finish_dev(tk_index *ix, int slot, const float *q_dev, int64_t nq,
                                         int k, int n_probes, int pass_1, int64_t capacity,
                                         const void *recv_dev, int64_t *out_ids_home_dev,
                                         int *flag_dev, void *stream)
{
    IXLOCK(ix);
    Plan p;
    TRY(make_plan(ix, k, n_probes, pass_1, p));
    int64_t qh = 0;
    TRY(shard_args(ix, slot, nq, capacity, p, qh));
    ARGCHECK(recv_dev && out_ids_home_dev, "recv/out buffers");
    Work &w = ix->works[(size_t)slot];
    ARGCHECK(!w.shard_plain || flag_dev, "the one-phase plain scan needs the batch's flag word at the finish");
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
        // (one-phase plain scan: the replay checks the lemma per home query and flags the batch)
        TRY(stage_back(ix, w, q_dev + q0 * ix->d, q0, nqh, k, p, out_ids_home_dev, st, pf, w.shard_plain,
                       TkSecond(), TkSecond(), w.shard_plain ? flag_dev : nullptr));
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

// GB/s of a kernel that gathers random rows of `row_bytes` out of `table_bytes` of HBM the way the rescoring
// kernel does (measurement plumbing for bench.py's roofline.rescore)
extern "C" int tk_measure_gather_bandwidth(int64_t table_bytes, int row_bytes, int64_t n_gather, int reps, double *gbps)
{
    TRY(require_gpu());
    ARGCHECK(row_bytes >= 16 && row_bytes <= 1024 && row_bytes % 16 == 0 && table_bytes >= (1 << 20) && n_gather >= 1 &&
             reps >= 1 && gbps, "16 <= row_bytes <= 1024 (a multiple of 16), table_bytes >= 1 MiB, n_gather, reps >= 1");
    DevBuf buf, out;
    const int64_t n_rows = table_bytes / row_bytes;
    int rc = buf.ensure((size_t)n_rows * row_bytes);
    if (rc == TK_OK) rc = out.ensure(16);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (rc == TK_OK) {
        hipError_t e = hipMemset(buf.p, 1, (size_t)n_rows * row_bytes);
        if (e == hipSuccess) e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        if (e == hipSuccess) {
            tk_launch_gather_rows(buf.p, n_rows, row_bytes, n_gather, out.as<uint32_t>(), nullptr);
            e = hipDeviceSynchronize();
        }
        if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
        for (int r = 0; r < reps && e == hipSuccess; r++)
            tk_launch_gather_rows(buf.p, n_rows, row_bytes, n_gather, out.as<uint32_t>(), nullptr);
        if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && ms > 0) *gbps = (double)n_gather * row_bytes * reps / (ms * 1e-3) / 1e9;
        else rc = fail(TK_ERR_HIP, e == hipSuccess ? "zero elapsed time" : hipGetErrorString(e));
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    buf.release();
    out.release();
    return rc;
}

// the library's exclusive prefix sum on host arrays (test plumbing: tests/test_scan_gpu.py)
extern "C" int tk_scan_exclusive_host(const void *in_host, void *out_host, int64_t n, int is64)
{
    TRY(require_gpu());
    ARGCHECK(n >= 0 && (n == 0 || (in_host && out_host)), "arrays");
    if (n == 0) return TK_OK;
    const size_t esz = is64 ? 8 : 4;
    DevBuf in, out, tmp;
    size_t tmp_bytes = 0;
    int rc = in.ensure((size_t)n * esz);
    if (rc == TK_OK) rc = out.ensure((size_t)n * esz);
    if (rc == TK_OK)
        rc = (is64 ? tk_scan_exclusive64(nullptr, &tmp_bytes, nullptr, nullptr, n, 0)
                   : tk_scan_exclusive(nullptr, &tmp_bytes, nullptr, nullptr, n, 0)) ? fail(TK_ERR_HIP, "scan size query") : TK_OK;
    if (rc == TK_OK) rc = tmp.ensure(tmp_bytes);
    if (rc == TK_OK && hipMemcpy(in.p, in_host, (size_t)n * esz, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(TK_ERR_HIP, "copy in");
    if (rc == TK_OK) {
        const int e = is64 ? tk_scan_exclusive64(tmp.p, &tmp_bytes, in.as<long long>(), out.as<long long>(), n, 0)
                           : tk_scan_exclusive(tmp.p, &tmp_bytes, in.as<int>(), out.as<int>(), n, 0);
        if (e || hipDeviceSynchronize() != hipSuccess) rc = fail(TK_ERR_HIP, "scan");
    }
    if (rc == TK_OK && hipMemcpy(out_host, out.p, (size_t)n * esz, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(TK_ERR_HIP, "copy out");
    in.release(); out.release(); tmp.release();
    return rc;
}
