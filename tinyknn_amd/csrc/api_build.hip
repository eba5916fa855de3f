// api_build.hip — device-side build of an index, raw-query preparation, exact brute force and the batch
// form of _FastDistanceTable.top over one flat array (FlatTop).  (Split from api.hip in round 4.)
#include "api_internal.h"

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
    ix->labels24 = ix->N < 0x00ffffff;      // (labels are row numbers)
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
    TRY(build_twins(ix, N));
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

// Another rank's shard of a complete unsharded index on the same device (tinyknn_hip.h): the clone borrows
// the replicated arrays and owns the compacted codes of its lists.
extern "C" tk_index *tk_index_clone_shard(tk_index *src, const int32_t *owner, int rank, int world)
{
    if (!src || !owner || world < 1 || rank < 0 || rank >= world || !src->have_lists || !src->have_data ||
        src->sharded) {
        (void)fail(TK_ERR_ARG, "bad argument: tk_index_clone_shard wants a complete unsharded index, an owner map and rank < world");
        return nullptr;
    }
    IXLOCK(src);
    if (flush_pending(src) != TK_OK || hipDeviceSynchronize() != hipSuccess) return nullptr;
    tk_index *ix = tk_index_create();
    if (!ix) return nullptr;
    ix->pq_centers.borrow(src->pq_centers);
    ix->dq = src->dq; ix->dpb = src->dpb; ix->M = src->M; ix->f_order = src->f_order; ix->order = src->order;
    ix->sqrt_nb = src->sqrt_nb;
    ix->active_centers.borrow(src->active_centers);
    ix->center_codes.borrow(src->center_codes);
    ix->n_lists = src->n_lists; ix->center_chunks = src->center_chunks; ix->d = src->d;
    ix->list_chunk_off.borrow(src->list_chunk_off);
    ix->list_n.borrow(src->list_n);
    ix->ids_off.borrow(src->ids_off);
    ix->ids.borrow(src->ids);
    ix->ids32.borrow(src->ids32);
    ix->have_ids32 = src->have_ids32;
    ix->labels24 = src->labels24;
    ix->twin_list.borrow(src->twin_list);
    ix->twin_off.borrow(src->twin_off);
    ix->twin_w = src->twin_w;
    ix->twin_unverified = src->twin_unverified;
    ix->twin_vouched = src->twin_vouched;
    ix->opt_replay_twin = src->opt_replay_twin;
    ix->total_chunks = src->total_chunks; ix->total_ids = src->total_ids;
    ix->max_list_chunks = src->max_list_chunks;
    ix->ids_unique = src->ids_unique;
    ix->cslots_i.borrow(src->cslots_i);
    ix->cslots_l.borrow(src->cslots_l);
    ix->c_chunk_off.borrow(src->c_chunk_off);
    ix->rot_t.borrow(src->rot_t);
    ix->rot_d_pad = src->rot_d_pad;
    ix->data.borrow(src->data);
    ix->N = src->N; ix->data_is_f64 = src->data_is_f64;
    ix->have_pq = ix->have_centers = ix->have_lists = ix->have_data = true;
    ix->plain_mode = src->plain_mode;
    // this rank's codes
    const int64_t L = ix->n_lists;
    std::vector<int64_t> sizes((size_t)L), loff((size_t)L + 1, 0);
    bool ok = hipMemcpy(sizes.data(), src->list_n.p, (size_t)L * 8, hipMemcpyDeviceToHost) == hipSuccess;
    for (int64_t i = 0; ok && i < L; i++) {
        ok = owner[i] >= 0 && owner[i] < world;
        loff[(size_t)i + 1] = loff[(size_t)i] + (owner[i] == rank ? (sizes[(size_t)i] + 15) / 16 : 0);
    }
    const int P = ix->M / 2;
    const size_t bytes = (size_t)tk_tiled_uint4s(loff[(size_t)L], P) * 16;
    ok = ok && ix->owner.ensure((size_t)L * 4) == TK_OK && ix->local_chunk_off.ensure((size_t)(L + 1) * 8) == TK_OK &&
         ix->codes.ensure(bytes > 0 ? bytes : 16) == TK_OK;
    ok = ok && hipMemcpy(ix->owner.p, owner, (size_t)L * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(ix->local_chunk_off.p, loff.data(), (size_t)(L + 1) * 8, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemset(ix->codes.p, 0, bytes > 0 ? bytes : 16) == hipSuccess;
    if (ok) {
        tk_launch_compact_tiled(src->codes.as<uint4>(), ix->codes.as<uint4>(), P, ix->list_chunk_off.as<int64_t>(),
                                ix->local_chunk_off.as<int64_t>(), (int)L, loff[(size_t)L], 0);
        ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    }
    if (!ok) {
        (void)fail(TK_ERR_HIP, "tk_index_clone_shard: owner out of range, or a HIP call failed");
        tk_index_destroy(ix);
        return nullptr;
    }
    ix->sharded = true;
    ix->rank = rank;
    ix->world = world;
    return ix;
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
