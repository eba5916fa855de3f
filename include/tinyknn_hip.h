/*
 * tinyknn_hip.h — C ABI of libtinyknn_hip.so, the MI355X (gfx950) implementation
 * of the ONE hot path of thomasahle/tinyknn: the Quick-ADC 4-bit PQ code scan, the
 * bounded order-dependent top-R heap, the per-query distance table and the exact
 * rescoring that sit behind FastPQ / _FastDistanceTable / IVF.query.
 *
 * Every entry point names the reference interface it replaces (file:line relative
 * to the reference repository).  Plain pointers and sizes only; no torch types.
 * Conventions
 *   - return value: 0 = ok, <0 = error; tk_last_error() gives the text
 *     (the reference kernels are void/nogil and cannot fail; its Python layer
 *     raises AssertionError — the Python binding maps error codes to that).
 *   - pointers are HOST pointers unless the parameter name ends in _dev;
 *     buffers are caller-owned, written in place, nothing is retained
 *     (same ownership rule as the Cython memoryview arguments).
 *   - `order`: accumulation order of the saturating int8 sum.
 *     TK_ORDER_SSE = _fast_pq.pyx:209-236 (one accumulator),
 *     TK_ORDER_AVX = _fast_pq_256.pyx:126-156 (two accumulators, merged last);
 *     the reference's public API uses AVX (fast_pq.py:21-24).
 *   - `stream`: a hipStream_t passed as void* (NULL = the default stream).  Device
 *     entry points only enqueue work; host-pointer entry points synchronise.
 * There is no CPU fallback: without a usable GPU every compute call fails.
 */
#ifndef TINYKNN_HIP_H
#define TINYKNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TK_ORDER_SSE 0
#define TK_ORDER_AVX 1

#define TK_OK 0
#define TK_ERR_ARG (-1)     /* bad argument (shape/size contract violated)   */
#define TK_ERR_HIP (-2)     /* HIP runtime error                              */
#define TK_ERR_STATE (-3)   /* index not fully populated for this call        */

/* ---- library -------------------------------------------------------------- */
const char *tk_last_error(void);
int tk_version(void);
/* number of visible GPUs; 0 when there is none (compute calls then fail) */
int tk_device_count(void);
int tk_set_device(int device);

/* ---- drop-in kernels on host buffers -------------------------------------- */

/* estimate_pq_sse(data, tables, out, signd)  _fast_pq.pyx:101-111
 * estimate_pq_avx(...)                       _fast_pq_256.pyx:52-62
 * data: uint64 (chunks, M) Quick-ADC layout (_transform.py:4-77); tables: uint64
 * (2M,) (_transform.py:114-138); out: uint64 (>= 2*chunks,), 16 int8/uint8 per chunk. */
int tk_estimate_pq(const uint64_t *data, int64_t chunks, int M, const uint64_t *tables,
                   uint64_t *out, int signd, int order);

/* Batched form of the same call: nq tables (nq, 2M) against one code array,
 * out (nq, 2*chunks).  What examples/example.py:60-66 does in a Python loop. */
int tk_estimate_pq_batch(const uint64_t *data, int64_t chunks, int M,
                         const uint64_t *tables, int64_t nq, uint64_t *out, int signd,
                         int order);

/* query_pq_sse(data, n, tables, indices, vals, signd, labels=None)  _fast_pq.pyx:114-206
 * query_pq_avx(...)                                                 _fast_pq_256.pyx:65-123
 * indices int64 (R,), vals int32 (R,) are read AND written (one heap is threaded
 * through several calls, ivf.py:137-150).  labels: int64 (>= n,) or NULL. */
int tk_query_pq(const uint64_t *data, int64_t chunks, int M, int64_t n,
                const uint64_t *tables, int64_t *indices, int32_t *vals, int R, int signd,
                const int64_t *labels, int order);

/* init_heap(indices, vals, signd)   _fast_pq.pyx:240-252 */
int tk_init_heap(int64_t *indices, int32_t *vals, int R, int signd);
/* insert(indices, vals, i, v)       _fast_pq.pyx:274-307 */
int tk_heap_insert(int64_t *indices, int32_t *vals, int R, int64_t i, int32_t v);
/* insert_is(indices, vals, i, v)    _fast_pq.pyx:256-271 */
int tk_heap_insert_is(int64_t *indices, int32_t *vals, int R, int64_t i, int32_t v);

/* FastPQ.distance_table / udistance_table   fast_pq.py:186-222 / :224-252
 * centers: float32 (16, dq) row-major; f_order != 0 when the caller's array was
 * F-ordered (dims_per_block == 1 leaves such a view, fast_pq.py:99-101; numpy then
 * sums the mean in that order).  q: (nq, dq) padded (and rotated) queries, float32
 * (q_is_f64 == 0) or float64.  aux0/aux1: signed: sqrt_n_blocks, unused;
 * unsigned: log(n_blocks), sqrt(n_blocks).  Outputs: tables uint8 (nq, M, 16) (the
 * transform_tables byte image), shift (nq,) in q's dtype, scale float64 (nq,). */
int tk_build_tables(const float *centers, int dq, int dpb, int f_order, const void *q,
                    int q_is_f64, int64_t nq, double aux0, double aux1, int signd,
                    uint8_t *tables, void *shift, double *scale);

/* knn_brute1(x, Y, k)  utils.py:89-92 (+ bottom_k :22-25): squared distances of
 * the n rows of Y (n, d) to x (d,), each float32 or float64 (flags; float64
 * arithmetic if either is, as numpy promotes `Y - x`), positions of the k smallest
 * in ascending order (ties: lower position first).  k >= n returns arange(n).
 * out_pos: int64 (min(k, n),).  Returns the count written, or <0. */
int64_t tk_knn_brute1(const void *x, int x_is_f64, const void *Y, int y_is_f64, int64_t n, int d,
                      int64_t k, int64_t *out_pos);

/* ---- device-resident code array ---------------------------------------------
 * A TransformedData (fast_pq.py:30,184) kept in HBM so that repeated
 * estimate_distances / top calls on the same data (examples/example.py:60-66 runs
 * 1000 of them) do not re-send the codes over PCIe. */
/* (a tk_codes handle carries the scratch of the calls made on it: one caller at a time per
 * handle; the host-buffer entry points above share one mutex-guarded scratch) */
typedef struct tk_codes tk_codes;
tk_codes *tk_codes_upload(const uint64_t *data, int64_t chunks, int M);
void tk_codes_free(tk_codes *c);
/* estimate_pq_* against the resident array: nq tables (nq, 2M) -> out (nq, 2*chunks) */
int tk_codes_estimate(tk_codes *c, const uint64_t *tables, int64_t nq, uint64_t *out, int signd,
                      int order);
/* same with device-resident tables (nq, M, 16) uint8 / out (nq, chunks*16) bytes, enqueued
 * on `stream`; four queries share each pass over the codes when nq >= 4 */
int tk_codes_estimate_dev(tk_codes *c, const void *tables_dev, int64_t nq, void *out_dev,
                          int signd, int order, void *stream);
/* query_pq_* against the resident array (indices / vals / labels are host buffers) */
int tk_codes_query(tk_codes *c, int64_t n, const uint64_t *tables, int64_t *indices,
                   int32_t *vals, int R, int signd, const int64_t *labels, int order);

/* ---- offline build path: the step before the hot path -------------
 * Both restate numpy's rounding (einsum order; OpenBLAS GEMM = FMA chain over k for the
 * shapes involved; argpartition's dumb_select) and reproduce the compiled reference's codes
 * and list memberships on the golden fixtures.
 *
 * tk_encode_pq: the per-block nearest-centroid search of FastPQ.transform
 * (fast_pq.py:174-181: knn_brute(col, code, 1) for every block).  data: (n, dq) rows after
 * the reference's padding (utils.py pad2) and, for a rotated PQ, after `data @ R.T` (a BLAS
 * GEMM that stays on the host; then float64).  labels: (n, M) uint8, to be packed by
 * transform_data (_transform.py:4-77). */
int tk_encode_pq(const float *centers, int dq, int dpb, const void *data, int data_is_f64,
                 int64_t n, uint8_t *labels);
/* tk_assign_lists: knn_brute(X, Y, k, metric) of IVF.build (ivf.py:85, utils.py:66-86) for
 * k <= 2 and d <= 384.  X: (n, d) float32 rows — whole 100-row chunks only (n % 100 rows at
 * the end of a data set are a differently shaped GEMM in numpy: the caller keeps numpy for
 * them); normalise != 0: X is divided by its row norms first (angular metric,
 * utils.py:73-74; d <= 128).  Y: (L, d) centres AFTER the caller's numpy normalisation,
 * float32 or float64, ynorm2 = np.einsum("ij,ij->i", Y, Y) in Y's dtype.
 * nearest: (n, k) int64 in numpy's argpartition order. */
int tk_assign_lists(const float *X, int64_t n, int d, int normalise, const void *Y, int y_is_f64,
                    const void *ynorm2, int64_t L, int k, int64_t *nearest);

/* ---- device-resident IVF index --------------------------------------------
 * Holds what the reference's IVF object holds after fit+build (ivf.py:14-17,
 * 77-102) in HBM, re-tiled for coalesced scans, and answers batches of queries
 * with the kernel pipeline that replaces IVF.query (ivf.py:106-163). */
typedef struct tk_index tk_index;

tk_index *tk_index_create(void);
void tk_index_destroy(tk_index *ix);

/* FastPQ state: centers (16, dq) float32, dims_per_block, sqrt_n_blocks
 * (fast_pq.py:99-102); f_order as in tk_build_tables; order = TK_ORDER_*. */
int tk_index_set_pq(tk_index *ix, const float *centers, int dq, int dpb, int f_order,
                    double sqrt_n_blocks, int order);
/* active_centers (n_lists, d) float32 and their packed codes
 * pq_transformed_centers.packed (center_chunks, M)   ivf.py:91-96 */
int tk_index_set_centers(tk_index *ix, const float *active_centers, int64_t n_lists, int d,
                         const uint64_t *center_codes, int64_t center_chunks);
/* pq_transformed_points / ids (ivf.py:100-102) flattened list-major:
 * list_sizes (n_lists,) true rows; codes (sum ceil(size/16), M) packed chunks;
 * ids (sum size,) labels. */
int tk_index_set_lists(tk_index *ix, const int64_t *list_sizes, const uint64_t *codes,
                       const int64_t *ids);
/* IVF.data: the (normalised) vectors used for rescoring, float32 or float64 as the
 * caller's X was (ivf.py:77-79 keeps X's dtype)  */
int tk_index_set_data(tk_index *ix, const void *data, int data_is_f64, int64_t N, int d);

/* _FastDistanceTable.top(transformed_data, data, k) (fast_pq.py:284-312) for a batch of queries
 * against the rows the index holds as its centres — tk_index_set_pq + tk_index_set_centers(rows,
 * packed codes) is all this call needs: per query a heap of rescore = min(2k + 10, n) PQ
 * estimates over ALL rows, then the exact float32 distances of those candidates (knn_brute1,
 * utils.py:89-92), the k best in ascending order.  (The coarse stage of IVF.query, ivf.py:131,
 * is this call on the coded centres.)  q: (nq, d) float32 rows as distance_table takes them,
 * q_pq: their padded / rotated form; out_ids: (nq, k) int64, -1 beyond n. */
int tk_index_top_centers(tk_index *ix, const float *q, const void *q_pq, int q_pq_is_f64, int64_t nq,
                         int k, int64_t *out_ids);

/* Largest batch the workspace is currently sized for grows on demand; this call
 * pre-sizes it (so that tk_index_query_batch_dev never allocates, e.g. under
 * stream capture). */
int tk_index_reserve(tk_index *ix, int64_t nq, int k, int n_probes, int pass_1);

/* IVF.query for a batch  ivf.py:106-163.
 * q: (nq, d) float32 queries AFTER the metric's normalisation (ivf.py:125-127 is
 * done by the host binding with numpy, as the reference does);
 * q_pq: (nq, dq) padded (rotated) queries for the table build, float32 or float64.
 * pass_1 <= 0 selects (n_probes+1)*k+1 (ivf.py:135-136).
 * out_ids: int64 (nq, k), rows padded with -1 when fewer than k ids exist.
 * Optional debug outputs (NULL to skip): out_probes int64 (nq, min(n_probes,n_lists)),
 * out_heap_idx int64 (nq, pass_1), out_heap_val int32 (nq, pass_1). */
int tk_index_query_batch(tk_index *ix, const float *q, const void *q_pq, int q_pq_is_f64,
                         int64_t nq, int k, int n_probes, int pass_1, int64_t *out_ids,
                         int64_t *out_probes, int64_t *out_heap_idx, int32_t *out_heap_val);

/* Same with device-resident inputs/outputs, enqueued on `stream`, no sync.  Large
 * batches are processed in sub-batches whose distance buffers stay under 16 GB (TINYKNN_WORKSPACE_GB):
 * EQUAL parts of at most tk_index_max_sub_batch queries, each through the pipeline by itself (calls that fit one
 * workspace are what pairs up under tk_index_set_coalesce).
 * An index handle serves one caller at a time: its entry points take a per-handle lock, so
 * calls from several threads are serialised (the reference's kernels are nogil and re-entrant
 * on distinct buffers; use one handle per thread for concurrency). */
int tk_index_query_batch_dev(tk_index *ix, const float *q_dev, const void *q_pq_dev,
                             int q_pq_is_f64, int64_t nq, int k, int n_probes, int pass_1,
                             int64_t *out_ids_dev, void *stream);

/* Batches in flight.  depth = 1 (default): tk_index_query_batch_dev enqueues the whole
 * pipeline on the caller's stream.  depth > 1: ALL code scans and the table builds run on
 * the caller's stream, in order — call c launches ONE kernel there that carries the list
 * scan of call c-2 and the coarse scan of call c as one pool of work — the coarse heap
 * replays + descriptors of all batches on one internal stream, and the heap replay (157
 * waves per 10 000 queries) + rescoring of a batch on one of `depth` more, handed over by
 * events and overlapping the scans of the later batches.  depth = 2 keeps the total at the
 * four hardware queues HIP maps streams onto (more streams share queues and serialise).
 * tk_index_join enqueues the list scans still owed and their replays.  The caller must not
 * reuse the input/output buffers of a call before tk_index_join(ix, stream), which also
 * makes `stream` wait for every batch in flight.  Stream capture (hipGraph) of a batch
 * needs depth = 1: the pipelined mode hands work to streams outside the capture. */
int tk_index_set_pipeline(tk_index *ix, int depth);
int tk_index_join(tk_index *ix, void *stream);
/* n = 2 (pipelined mode only): two consecutive tk_index_query_batch_dev[_ex] calls with the same
 * k / n_probes / pass_1 / stream run through the pipeline as ONE batch of nq_a + nq_b queries.
 * The kernels that leave most of the chip idle (two heap replays of 157 dependent waves per
 * 10 000 queries, the small kernels of the coarse stage) take as long for 20 000 queries as for
 * 10 000.  The first call is held and returns; the second joins it and enqueues the pair.  Nothing is
 * copied: the kernels read each call's queries from, and write its ids to, the call's own buffers
 * (which stay the library's until tk_index_join, as above); each call's completion event is recorded
 * behind the pair's last kernel.  Results are those of separate calls
 * (the same kernels over the same rows).  A held call is launched alone by tk_index_join /
 * tk_index_quiesce / any tk_index_set_*, or when the next call cannot join it.
 * A HELD call has enqueued NOTHING yet: its completion event (tk_index_query_batch_dev_ex) is not
 * recorded and its pinned copy not ordered until its partner arrives or tk_index_join runs — wait on
 * such an event only after one of the two (tk_index_pending() > 0 says calls are still owed;
 * hipEventSynchronize on a never-recorded event returns at once).  n = 1 (default):
 * every call its own batch.  ivf.py:106-163 answers one query per call; the batch forms here and
 * everything about their scheduling are this library's. */
int tk_index_set_coalesce(tk_index *ix, int n);
/* hipGraph of the pipelined mode: tk_index_quiesce (device-synchronising; forgets the completion
 * events of earlier calls), then capture on a non-NULL stream any number of
 * tk_index_query_batch_dev calls followed by tk_index_join on that stream.  The internal streams
 * fork from the capturing stream through the events the calls record on it and are all joined
 * again by tk_index_join, so the capture is one self-contained graph: replaying it runs the
 * captured batches at the pipelined rate (workspaces, events and streams must exist beforehand:
 * run the same calls once uncaptured, tk_index_set_profiling off). */
int tk_index_quiesce(tk_index *ix);

/* Heap replay strategy.  0 = automatic: when the replay can skip `insert`'s
 * duplicate-label scan without changing the result (fresh heaps and pairwise
 * distinct labels, i.e. IVF.build(n_probes=1)) each LANE replays one query (64
 * queries per wave, heap columns in LDS; pass_1 <= 574), or, for larger heaps, a
 * wave replays one query on packed 32-bit entries; otherwise the general kernel
 * (int64 labels + duplicate scan, one query per wave) runs.  1 = always the
 * general kernel.  2 = the packed wave kernel instead of the lane kernel.  3 = the
 * wave-per-query register heap (heaps of <= 513 entries; automatic for small batches,
 * TK_OPT_PAIR_NQ) for every batch size.  All are
 * bit-exact replays of _fast_pq_256.pyx:73-123; the switch exists for A/B timing
 * and for the parity tests. */
int tk_index_set_heap_mode(tk_index *ix, int mode);

/* Probed-list scan strategy.  0 = automatic: list-major (each chunk scored for
 * four queries per pass, pairs grouped by list on the device) when a batch has at
 * least 8 (query, probe) pairs per list, else query-major (one query per wave).
 * 1 = always query-major, 2 = always list-major.  Identical outputs. */
int tk_index_set_scan_mode(tk_index *ix, int mode);

/* Probed lists behind the first ones as PLAIN integer sums on the int8 matrix cores
 * (v_mfma_i32_32x32x32_i8: one-hot(code) x table), clamped to int8.  For a query whose table
 * bounds the negative mass of each of the reference's saturating chains by 128, clamp(plain sum)
 * equals the value compute_block_dists_avx (_fast_pq_256.pyx:126-156) returns wherever that value
 * is below C = 127 - (negative mass) and is >= C elsewhere; the replay inserts only rows below
 * its bound, which never rises (_fast_pq_256.pyx:73-123), so once the bound is <= C the replay
 * over the plain sums IS the replay over the reference's values.  The first probed lists of a
 * query (until they hold 2 * pass_1 rows) stay on the exact kernel; the lane replay checks the
 * condition per query and the queries that fail it are scanned again exactly and replayed again:
 * results are identical to mode 1 in every case (tests/test_plain_scan_gpu.py).
 * 0 = automatic (list-major batches, signed tables, <= 52 blocks, lane replay): a flagged query
 * costs two scans and two replays, so the path first proves itself on ONE probe batch (the batches
 * behind it stay exact until its flagged count has come back — read from a page-locked word, never
 * waited for), is used while at most 1 % of a batch's queries are flagged, and otherwise pauses
 * for 256..4096 batches before the next probe (data without structure: 42 % flagged on iid
 * vectors); 1 = off; 2 = always (no pausing: tests, A/B).  Every mode returns identical results.
 * Environment TINYKNN_PLAIN_SCAN=0 switches it off process-wide. */
int tk_index_set_plain_scan(tk_index *ix, int mode);
/* Accounting of the plain path for the last batch enqueued (synchronises): out8 = plain units
 * (tiles of 32 pairs), plain pairs, exact pair records, head pair records, queries flagged for the
 * re-scan, sum over the plain units of the list's chunk pairs, state of mode 0 (0 probe next,
 * 1 waiting for the probe's count, 2 on, 3 paused), batches left of the pause. */
int tk_index_plain_stats(tk_index *ix, int64_t *out8);
/* Per-index options (no process-wide state: the reference's entry points carry none either,
 * _fast_pq.pyx:101-307 are nogil and re-entrant).  Results never depend on an option; they exist for
 * A/B measurements and for the tests.
 *   TK_OPT_SCAN_FORM     table rows of the exact list-major kernel: 0 = per-lane global loads (DEFAULT;
 *                        fastest, profiles/r02_scan_forms.md), 1 / 2 = staged per block in LDS (a launch
 *                        that carries the exact heads of a plain batch always runs form 0: only that
 *                        form knows row limits)
 *   TK_OPT_RESCORE_FORM  candidate rows of the rescoring: 2 = staged through LDS in tiles of 32 rows
 *                        (DEFAULT), 1 = tiles of 64, 0 = every lane walks its own row (always for float64
 *                        operands and d % 4 != 0)
 *   TK_OPT_PLAIN_LIMIT   a cap on every query's table limit C (plain_scan.hip's lemma; INT_MAX = none):
 *                        -128 sends EVERY query of a plain batch through the exact re-scan + second
 *                        replay (tests/test_plain_scan_gpu.py) */
#define TK_OPT_SCAN_FORM 1
#define TK_OPT_RESCORE_FORM 2
#define TK_OPT_PLAIN_LIMIT 3
/*   TK_OPT_REPLAY_LAZY   the lane replay of the probed lists: 1 = nothing is staged, a lane reads its row's block
 *                        minima and fetches a block only where its minimum passes the bound (as FlatTop's
 *                        replay); 0 = every block staged through LDS; -1 = by the index (DEFAULT: lazy where a
 *                        query's probed lists hold more than 8 heap sizes of blocks — 100M x 128: 6 250 blocks
 *                        against a heap of 111).  Identical results. */
#define TK_OPT_REPLAY_LAZY 4
/*   TK_OPT_REPLAY_COUNT  1: the lane replays of the probed lists count their insert rounds (measurement plumbing
 *                        for bench.py's roofline.replay; read and zeroed by tk_index_replay_stats); 0 (DEFAULT): off */
#define TK_OPT_REPLAY_COUNT 5
/*   TK_OPT_REPLAY_TWIN   labels that repeat (IVF.build(n_probes >= 2), ivf.py:53): 1 (DEFAULT) = the lane replay
 *                        decides `insert`'s duplicate test (_fast_pq.pyx:284-287) from the positions of a row's
 *                        other copies (twins.hip's table, heap.hip TWIN form: 64 queries per wave, position
 *                        entries, no set of labels); 0 = the hash set of the labels in the heap (32 queries per
 *                        wave).  Identical results. */
#define TK_OPT_REPLAY_TWIN 6
/*   TK_OPT_TWIN_VOUCH    the TWIN form rests on two facts about the lists: the copies of a label lie in different lists
 *                        and carry the same code — true of IVF.build's lists (ivf.py:77-102: a list's codes are the codes
 *                        of data[ids]).  tk_index_set_lists and tk_index_build_dev hold every list's codes and CHECK
 *                        them on the device (an index that fails keeps the label-based test); a rank that was handed
 *                        only its own lists' codes (tk_index_set_lists_shard) cannot, and uses the table only behind
 *                        1 = "the caller has checked" (tinyknn_amd.DeviceIndex does, on the host).  0 (DEFAULT). */
#define TK_OPT_TWIN_VOUCH 7
/*   TK_OPT_PAIR_NQ       batches of up to this many queries (DEFAULT 8192 one batch at a time, at most 4096 per launch — a pair
 *                        of calls of 2048 — in pipelined mode; 0 = never; environment TINYKNN_PAIR_NQ sets the default of new indexes) replay their heaps
 *                        one query per WAVE with the heap in registers, two / four / eight nodes per lane (heaps of <= 129 /
 *                        257 / 513 entries: IVF.query's 111 and its coarse top's 30 at the reference's bench settings,
 *                        examples/bench.py:118-137): one query per call is ~760 dependent inserts, 0.47 of 0.54 ms in the
 *                        lane kernel, 0.12 of 0.19 ms here.  Same heap arrays. */
#define TK_OPT_PAIR_NQ 8
/*   TK_OPT_LABELS24      the register heap's duplicate test (labels that repeat, IVF.build(n_probes >= 2); a probe list
 *                        that names a list twice): 1 (DEFAULT) = on entries value8 << 24 | label24 — one register per
 *                        slot, as with positions — where every label of the index is below 0xffffff (16.7 M rows);
 *                        0 = always on (value, label64) entries, three registers per slot (what larger indexes get). */
#define TK_OPT_LABELS24 9
int tk_index_set_option(tk_index *ix, int option, int value);
/* The table behind TK_OPT_REPLAY_TWIN (diagnostics, tests): *rows = stored rows (the length of the concatenated
 * ids), *w = other copies listed per row (0: no table — labels distinct, not int32, or one label stored more
 * than 17 times); list_out / off_out (rows x w int32 each, or NULL): list and offset inside that list of a
 * row's u-th other copy, -1 where a label has fewer copies.  What the reference finds by scanning the heap's
 * labels in `insert` (_fast_pq.pyx:284-287) follows from these positions (heap.hip, TWIN form). */
int tk_index_twin_table(tk_index *ix, int64_t *rows, int *w, int32_t *list_out, int32_t *off_out);

/* Stage timing.  on = n > 0: every n-th (sub-)batch records HIP events on its streams
 * around the stages (no synchronisation in the query call; 1 = every batch).
 * tk_index_last_profile synchronises that stream and returns the mean
 * milliseconds per stage over the batches recorded since the last read
 * [tables, coarse_scan, coarse_heap, coarse_rescore+slots, scan, heap, rescore, plain kernel alone
 * (HIP events in front of and behind scan_plain_wave_kernel on the stream it is launched on; 0 when
 * no recorded batch ran it)], their count, and the algorithmic bytes the list-scan kernel streamed for the
 * most recent batch (code bytes + table + heap per query; with depth > 1 the
 * timed launch also carries the next batch's coarse scan, whose bytes are included). */
int tk_index_set_profiling(tk_index *ix, int on);
int tk_index_last_profile(tk_index *ix, float *ms8, double *scan_bytes, int *batches);

/* measurement plumbing (bench.py): GB/s of a kernel that only reads `bytes` of HBM, every byte
 * once with the flat scan's access pattern — the streaming-read ceiling of the box */
/* out4 = insert rounds summed over the replay waves, the most rounds one wave ran, waves, 16-block segments walked,
 * since TK_OPT_REPLAY_COUNT was set / the last call (synchronises, zeroes) */
int tk_index_replay_stats(tk_index *ix, int64_t *out4);
int tk_measure_read_bandwidth(int64_t bytes, int reps, double *gbps);
/* ... and of a kernel that gathers n_gather random rows of row_bytes (16 .. 1024, a multiple of 16) out of
 * table_bytes of HBM with the rescoring kernel's access pattern (knn_brute1's `data[indices]`, utils.py:89-92):
 * the ceiling bench.py's roofline.rescore is quoted against. */
int tk_measure_gather_bandwidth(int64_t table_bytes, int row_bytes, int64_t n_gather, int reps, double *gbps);
/* The library's own exclusive prefix sum (one launch, decoupled look-back: devbuild.hip) on host arrays — the
 * per-batch sums of a list-sharded rank go through it on the device; this entry exists so that a test can check it
 * against numpy at any length.  is64: int64 elements, else int32.  in_host == out_host is allowed. */
int tk_scan_exclusive_host(const void *in_host, void *out_host, int64_t n, int is64);

/* ---- device-resident build ----------------------------------------
 * IVF.build(X, n_probes = 1 or 2) (ivf.py:77-102) for float32 vectors that are produced IN HBM
 * and never visit the host — how the 100M x 128 configuration is assembled ("per-GPU
 * generation on device (seeded per shard), codes produced by the build's encoder").
 * tk_index_alloc_data: after tk_index_set_pq; the index allocates its (N, d) float32
 *   rescoring vectors and returns the device pointer for the caller to fill (NULL on error).
 * tk_index_synth_data: fills rows [row0, row0 + n) with centres[c(row)] + sigma * N(0, 1)
 *   from a counter-based generator — a pure function of (seed, row, column), independent of
 *   how the rows are split over calls or ranks; centres: host (n_centres, d) float32 or NULL
 *   (noise only).  tk_synth_rows: the same generator into a host buffer (queries, samples).
 * tk_index_build_dev: normalise != 0 divides every row by its norm first (ivf.py:78-79);
 *   the n_probes <= 2 nearest of the C centres per row (knn_brute(data, all_centers, n_probes),
 *   ivf.py:85; with 2 every row sits in two lists, column-0 members of a list before its column-1
 *   members as group_data_by_indices appends them, and the replay runs its duplicate test):
 *   search_centers (C, d) float32 = all_centers after knn_brute's own normalisation for the
 *   angular metric (utils.py:75; NULL: all_centers themselves), ynorm2 its einsum norms),
 *   active centres = the rows of all_centers that own a vector, in id order (ivf.py:91), PQ codes of
 *   every row and of the active centres (R (dq, d_pad) float64 = FastPQ.R or NULL; the
 *   rotation is the float64 FMA chain of the device front end, not numpy's DGEMM), rows
 *   grouped by list in ASCENDING ROW ORDER (numpy's unstable argsort leaves the order inside
 *   a list unspecified, utils.py:131), packed into the Quick-ADC layout.  The index is then
 *   complete (centres, lists, data).  n_active_out: number of lists.
 * tk_index_export_lists / _centers / tk_index_read_rows: what a built index holds, back on
 *   the host in the reference's formats (list_sizes (n_lists,), packed codes (sum
 *   ceil(size/16), M) uint64, ids (sum size,); active centres and their packed codes; rows of
 *   IVF.data by id) — any output may be NULL. */
float *tk_index_alloc_data(tk_index *ix, int64_t N, int d);
int tk_index_synth_data(tk_index *ix, int64_t row0, int64_t n, uint64_t seed, const float *centres,
                        int n_centres, float sigma);
int tk_synth_rows(float *out, int64_t row0, int64_t n, int d, uint64_t seed, const float *centres,
                  int n_centres, float sigma);
int tk_index_build_dev(tk_index *ix, int normalise, const float *all_centers,
                       const float *search_centers, const float *ynorm2, int64_t C, int n_probes,
                       const double *R, int d_pad, int64_t *n_active_out);
int tk_index_export_lists(tk_index *ix, int64_t *list_sizes, uint64_t *codes, int64_t *ids);
int tk_index_export_centers(tk_index *ix, float *active_centers, uint64_t *center_codes);
int tk_index_read_rows(tk_index *ix, const int64_t *rows, int64_t n, float *out);

/* ---- device front end, "fast mode"  -----------------------------------
 * What IVF.query does on the host before the table build (ivf.py:125-128,
 * fast_pq.py:200-204): float32 normalisation for the angular metric, zero padding to dq,
 * rotation by R.  NOT bit-identical to the host path, which is the default everywhere:
 * numpy normalises with a BLAS dot and rotates with a BLAS GEMV whose summation orders are
 * not restated.  Here the norm is numpy's pairwise float32 sum and the rotation a float64
 * FMA chain; both results are within 1 ulp of the host's, and the ids differ only where
 * that flips a quantised table entry or a rescoring tie (measured in DESIGN.md).
 * tk_index_set_rotation: R (dq, d_pad) float64 row-major = FastPQ.R, or NULL (no rotation).
 * tk_index_prepare_dev: q_raw_dev (nq, d) float32 -> qn_dev (nq, d) float32 (normalised if
 * `angular`, else a copy; may alias q_raw_dev) and q_pq_dev (nq, dq) float64 when a rotation
 * is set, float32 otherwise — the two inputs of tk_index_query_batch_dev. */
int tk_index_set_rotation(tk_index *ix, const double *R, int d_pad);
int tk_index_prepare_dev(tk_index *ix, const float *q_raw_dev, int64_t nq, int angular,
                         float *qn_dev, void *q_pq_dev, void *stream);
/* raw float32 queries on the host -> ids on the host through the fast front end (H2D,
 * prepare, pipeline, D2H; synchronous) */
int tk_index_query_batch_raw(tk_index *ix, const float *q_raw, int64_t nq, int angular, int k,
                             int n_probes, int pass_1, int64_t *out_ids);

/* ---- exact host front end + streaming sessions (front.hip) ------------------------------
 * The host side of IVF.query (ivf.py:125-128, fast_pq.py:200-204): float32 normalisation
 * by np.linalg.norm = sqrtf(cblas_sdot(x, x)), pad1, and the float64 rotation
 * q @ R.T = cblas_dgemv(ColMajor, Trans, d_pad, dq, 1, R, d_pad, x, 1, 0, y, 1).  The
 * summation orders of those two calls belong to the BLAS build numpy links and are not
 * restated: tk_host_blas_bind resolves both symbols from THAT library (path of the shared
 * object numpy loaded; ILP64 "64_"-suffixed and "scipy_"-prefixed names are tried first),
 * and tk_prepare_queries_host calls them row by row from a thread pool — the reference's
 * arithmetic, bit for bit, without the Python loop.  No restated CPU arithmetic: without a
 * bound BLAS the call fails (TK_ERR_STATE).
 * tk_host_threads(n): pool size (n <= 0: keep / create the default = min(cores, 32), env
 * TINYKNN_HOST_THREADS); returns the size in use.
 * tk_prepare_queries_host: q_raw (nq, d) float32 -> qn (nq, d) float32 (normalised when
 * `angular`; may alias q_raw: the reference normalises in place) and, with R (dq, d_pad)
 * float64 row-major = FastPQ.R, q_pq (nq, dq) float64 = pad1(qn) @ R.T.  Without R the
 * table-build query is pad1(qn) (zeros appended), which needs no arithmetic. */
int tk_host_blas_bind(const char *blas_shared_object_path);
int tk_host_blas_bound(void);
int tk_host_threads(int n);
int tk_prepare_queries_host(const float *q_raw, int64_t nq, int d, int angular, float *qn,
                            const double *R, int dq, int d_pad, double *q_pq);

/* tk_index_query_batch_dev with a completion hook: behind the batch's last kernel the ids
 * are copied into out_ids_pinned (page-locked host memory, or NULL) and done_event (a
 * hipEvent_t, or NULL) is recorded.  With tk_index_set_pipeline(depth > 1) that happens up
 * to three calls later (or at tk_index_join); tk_index_pending = number of calls whose last
 * stage is not enqueued yet.  nq must fit one sub-batch (tk_index_max_sub_batch).
 * With tk_index_set_coalesce(ix, 2) the first call of a pair is only held: done_event is valid
 * (recorded) only once the partner call has been made or tk_index_join has run.
 * The internal front / replay streams of the pipelined mode belong to the PROCESS (one set per
 * device, shared by every index).  While a hipGraph capture of one index is open (between
 * tk_index_quiesce and the end of the capture) no other index of the process may be used: its work
 * would be pulled into, or invalidate, the foreign capture. */
int tk_index_query_batch_dev_ex(tk_index *ix, const float *q_dev, const void *q_pq_dev,
                                int q_pq_is_f64, int64_t nq, int k, int n_probes, int pass_1,
                                int64_t *out_ids_dev, int64_t *out_ids_pinned, void *done_event,
                                void *stream);
int64_t tk_index_max_sub_batch(tk_index *ix, int k, int n_probes, int pass_1);
/* hipStream_t on which to copy a batch's inputs in (pipelined mode: the index's front stream,
 * where the batch's first kernel runs; NULL: use the stream the batch is enqueued on) */
void *tk_index_input_stream(tk_index *ix);
int tk_index_pending(tk_index *ix);
/* info8 = {d, dq, M, n_lists, rotation d_pad (0: none), pipeline depth, N, total chunks} */
int tk_index_info(tk_index *ix, int64_t *info8);

/* Streaming session: IVF.query for a stream of batches, raw float32 queries on the host in,
 * ids on the host out (ivf.py:106-163 per row).  submit = exact host preparation on the
 * pool (above) into page-locked staging, H2D on a copy stream, the device pipeline on a
 * compute stream, D2H of the ids behind the last kernel; it returns a ticket at once, so
 * that the preparation and the copies of a batch overlap the kernels of the batches before
 * it.  out_ids (nq, k) is written by tk_stream_wait(ticket) / tk_stream_drain (or by a later
 * submit that reuses the slot: n_slots batches may be outstanding) and must stay valid
 * until then.  R / d_pad as in tk_prepare_queries_host (NULL: unrotated PQ).
 * tk_stream_submit_prepared: the same from already prepared rows (qn, q_pq as
 * tk_index_query_batch takes them).  A session owns its index while batches are
 * outstanding; one thread at a time. */
typedef struct tk_stream tk_stream;
tk_stream *tk_stream_create(tk_index *ix, int64_t max_nq, int k, int n_probes, int pass_1,
                            int angular, const double *R, int d_pad, int n_slots);
int64_t tk_stream_submit(tk_stream *s, const float *q_raw, int64_t nq, int64_t *out_ids);
int64_t tk_stream_submit_prepared(tk_stream *s, const float *qn, const void *q_pq, int64_t nq,
                                  int64_t *out_ids);
/* n_probes / pass_1 of the submits that follow (drains the session first; its staging buffers
 * are kept, so that a sweep over n_probes does not re-allocate page-locked memory) */
int tk_stream_set_probes(tk_stream *s, int n_probes, int pass_1);
int tk_stream_wait(tk_stream *s, int64_t ticket);
int tk_stream_drain(tk_stream *s);
double tk_stream_prepare_seconds(tk_stream *s);
void tk_stream_destroy(tk_stream *s);

/* ---- exact k nearest vectors: the ground truth of recall --------------
 * knn_brute(q, IVF.data, k, "euclidean") (utils.py:66-86, as examples/bench.py:85 uses it) on
 * the f32 matrix cores: part = (|q|^2 + |y|^2) - (2q).y with numpy's einsum norms and the
 * GEMM as the f32 FMA chain (v_mfma_f32_32x32x2_f32) — numpy's part values bit for bit — and
 * the k smallest in ascending (part, row) order (numpy's argpartition leaves the order of
 * the first k, and the choice among exact ties at the k-th value, unspecified).
 * q: (nq, d) float32 host, already normalised for the angular metric like IVF.data;
 * out_ids: (nq, k) int64 host.  float32 vectors, d <= 128. */
int tk_index_knn_brute(tk_index *ix, const float *q, int64_t nq, int k, int64_t *out_ids);

/* ---- list-sharded index over `world` ranks, one process per GPU --------
 * The inverted lists (ivf.py:100-102) are partitioned by cluster id: owner[l] is the rank
 * that stores list l's packed codes.  PQ, coarse centres, ids and the rescoring vectors are
 * replicated (set_pq / set_centers / set_data as usual), so every rank derives the same
 * probe order without communication.  tk_index_set_lists_shard replaces tk_index_set_lists:
 * list_sizes and ids describe ALL lists, codes_owned only the lists with owner[l] == rank,
 * concatenated in list order.
 *
 * The reference chains all probed lists of a query through ONE order-dependent heap
 * (ivf.py:137-150), so partial top-k's do not merge; the exchange carries the int8 distance
 * bytes to the query's home rank (query i lives on rank i / ceil(nq/world)), which replays
 * the heap exactly as the unsharded index does.  One batch =
 *   tk_index_shard_coarse_dev  distance tables for all nq queries, then the coarse stage
 *                              (dtable.top(centers), ivf.py:131) of this rank's HOME queries
 *                              only; probes_home_dev: int64 (ceil(nq/world), min(n_probes,
 *                              n_lists)), rows past nq = 0;
 *   all-gather of the probes   -> probes_all (world * ceil(nq/world), ...) = the probe lists
 *                              of all queries in query order (RCCL, by the caller);
 *   tk_index_shard_scan_dev    the owned (query, list) segments scored straight into
 *                              `send_dev`: `world` regions of `capacity` uint4 (16 distances
 *                              each), region h = my segments of rank h's queries, in (query,
 *                              probe slot) order; *flag_dev |= 1 if a region overflowed
 *                              (repeat the batch with a larger capacity).  probes_all_dev ==
 *                              NULL: the replicated form — this call builds the tables and
 *                              runs the coarse stage for ALL queries itself (no coarse call,
 *                              no probe all-gather; 1/3 of a batch's work is then not
 *                              divided by `world`);
 *   all-to-all(send -> recv)   equal splits of capacity*16 bytes (RCCL, by the caller);
 *   tk_index_shard_finish_dev  received segments -> distance rows of the home queries,
 *                              heap replay, rescoring; out_ids_home_dev: int64
 *                              (ceil(nq/world), k), rows past nq and missing ids = -1; flag_dev:
 *                              the batch's flag word (may be NULL behind tk_index_shard_scan_dev /
 *                              the two-phase scan; required behind tk_index_shard_scan_plain_dev,
 *                              which may raise bit 4 in it, see there);
 *   all-gather of the id rows  (by the caller).
 * All calls enqueue on `stream` and use workspace `slot` (< pipeline depth), so that
 * several batches can be in flight on different streams.  nq <= 131072 per batch. */
int tk_index_set_lists_shard(tk_index *ix, const int64_t *list_sizes, const int32_t *owner,
                             int rank, int world, const uint64_t *codes_owned,
                             const int64_t *ids);
/* a complete unsharded index (host upload or tk_index_build_dev) -> this rank's shard, in place:
 * the codes of the lists with owner[l] == rank are compacted on the device, the rest is dropped */
int tk_index_shard_resident(tk_index *ix, const int32_t *owner, int rank, int world);
int tk_index_shard_coarse_dev(tk_index *ix, int slot, const float *q_dev, const void *q_pq_dev,
                              int q_pq_is_f64, int64_t nq, int k, int n_probes, int pass_1,
                              int64_t *probes_home_dev, void *stream);
int tk_index_shard_scan_dev(tk_index *ix, int slot, const float *q_dev, const void *q_pq_dev,
                            int q_pq_is_f64, int64_t nq, int k, int n_probes, int pass_1,
                            const int64_t *probes_all_dev, int64_t capacity, void *send_dev,
                            int *flag_dev, void *stream);
int tk_index_shard_finish_dev(tk_index *ix, int slot, const float *q_dev, int64_t nq, int k,
                              int n_probes, int pass_1, int64_t capacity, const void *recv_dev,
                              int64_t *out_ids_home_dev, int *flag_dev, void *stream);
/* tk_index_shard_scan_dev in the form the unsharded pipeline scans (ivf.py:137-150 through
 * _fast_pq_256.pyx:73-156): of every owned segment only the rows a query scans first — the head of
 * its first probed list, two heap sizes — go through the exact kernel, everything else is scored as
 * plain sums on the int8 matrix cores, and the REPLAY at the query's home rank checks the condition
 * under which those are the reference's values (its bound at the first plain block <= the limit C of
 * the query's table; tk_index_set_plain_scan).  One kernel chain, no extra collective.  A home query
 * that fails the check raises bit 4 of *flag_dev in tk_index_shard_finish_dev (the codes to scan it
 * again exactly are on other ranks): the flag word travels with the ids, every rank sees it, and
 * the batch is repeated behind tk_index_shard_scan_head_dev (below).  Arguments as
 * tk_index_shard_scan_dev + bound_dev (NULL here); falls back to tk_index_shard_scan_dev by itself
 * where the form does not apply (tk_index_shard_plain says 0, heaps beyond the lane replay, world *
 * capacity shorter than the longest list, labels that repeat WITHOUT a twin table).  Labels that repeat
 * (IVF.build(n_probes >= 2)) take this form through the lane replay's TWIN test where the index holds
 * the twin table (tk_index_twin_table: width > 0) and its premises are verified — on a rank that was
 * handed only its own lists' codes (tk_index_set_lists_shard) the device cannot check them: the host
 * wrapper checks over all lists and vouches (tk_index_set_option TK_OPT_TWIN_VOUCH).
 *
 * The same with ONE byte per query exchanged first — for data on which the optimistic form fails (one
 * query in 20 000 of the 100M x 128 index; a sharded batch holds 120 000):
 *   tk_index_shard_scan_head_dev   tables / probe lists / segment positions as tk_index_shard_scan_dev;
 *                                  the HEAD (two heap sizes of rows) of every first probed list this
 *                                  rank owns scored exactly into send_dev; bound_dev[nq] = the bound
 *                                  after it (order key, as tk_index_shard_bound_dev; 255 elsewhere);
 *   all-reduce(MIN, uint8)         by the caller;
 *   tk_index_shard_scan_plain_dev(bound_dev)
 *                                  the one-phase scan, with every query whose bound is above its table's
 *                                  limit kept on the exact kernel: the check at home cannot fail.
 * Against tk_index_shard_scan_first_dev / _rest_dev, phase 1 scores ~14 chunks per query instead of a
 * whole list, and replays as many. */
int tk_index_shard_scan_plain_dev(tk_index *ix, int slot, const float *q_dev, const void *q_pq_dev,
                                  int q_pq_is_f64, int64_t nq, int k, int n_probes, int pass_1,
                                  const int64_t *probes_all_dev, int64_t capacity, void *send_dev,
                                  int *flag_dev, const uint8_t *bound_dev, void *stream);
int tk_index_shard_scan_head_dev(tk_index *ix, int slot, const float *q_dev, const void *q_pq_dev,
                                 int q_pq_is_f64, int64_t nq, int k, int n_probes, int pass_1,
                                 const int64_t *probes_all_dev, int64_t capacity, void *send_dev,
                                 int *flag_dev, uint8_t *bound_dev, void *stream);
/* The library's process-wide internal streams (one set per device): role 0 = the front stream (high
 * priority: table builds, coarse stages), role 1 = replay stream i (0 <= i < 8).  For hosts that run a
 * stage pipeline of their own over the list-sharded entry points — which only enqueue on the stream
 * they are given — without creating streams that would share HIP's four hardware queues with these.
 * NULL on error.  Never destroy them. */
void *tk_shared_stream(int role, int i);
/* Another rank's shard of a complete UNSHARDED index on the same device: a new handle that borrows
 * the source's replicated arrays (PQ, centres, list tables, ids, vectors — the source must outlive
 * it and must not be re-populated meanwhile) and owns only the codes of the lists with
 * owner[l] == rank, compacted on the device.  The source stays usable and unsharded.  This is how
 * several ranks of a list partition are played on ONE GPU (tests, bench.py's one-rank share of a
 * W = 8 partition): a 100M x 128 index holds its 51 GB of vectors once, not once per rank. */
tk_index *tk_index_clone_shard(tk_index *src, const int32_t *owner, int rank, int world);

/* Longest (source -> home) stream of the slot's last tk_index_shard_scan_dev in uint4, fitted or
 * not: max-reduce it over the ranks and size `capacity` by it (the regions of the dense exchange
 * travel whole).  Synchronises with the device. */
int tk_index_shard_usage(tk_index *ix, int slot, int64_t *max_stream_uint4);

/* Filtered exchange — the same batch with a fraction of the bytes on
 * the links.  Every insert of a 16-code block is below the bound captured at the block's start
 * (_fast_pq_256.pyx:73,111-123), so the bound never increases from block to block: a distance
 * that is not below B1, the bound after the query's FIRST probed list, can never enter the
 * heap.  After tk_index_shard_scan_dev (whose buffer then stays on the rank: `scan_dev`):
 *   tk_index_shard_bound_dev   bound_dev[nq] bytes: B1 (order key = distance byte ^ 0x80) of
 *                              the queries whose first probed list this rank owns — replayed
 *                              from the fresh heap over that list — and 255 elsewhere
 *                              (computed from the heap's VALUES alone, which is exact as long
 *                              as the ids inside one list are distinct: a row joins a list
 *                              once, ivf.py:85-102, utils.py:131-150);
 *   all-reduce(MIN, uint8)     -> B1 of every query on every rank (by the caller);
 *   tk_index_shard_filter_dev  the blocks that travel — all of a query's first list, of the
 *                              later lists those whose minimum is below B1 — as records of 5
 *                              int32 {row * cap + block (in the home rank's distance rows), 16
 *                              distance bytes}, grouped by home rank in records_dev (room for
 *                              world * capacity records); counts_dev: 3 * world int32,
 *                              [0, world) = records per home rank, [2 world, 3 world) = all
 *                              blocks this rank scored per home rank (what the dense exchange
 *                              carries), the middle third is scratch;
 *   all-to-all of the counts, then of the records with those splits (by the caller);
 *   tk_index_shard_finish_filtered_dev
 *                              home rows filled with the largest value — a block that did not
 *                              travel then behaves like the real one: no byte of it is below
 *                              any bound — received blocks dropped in, replay, rescoring as
 *                              tk_index_shard_finish_dev; *flag_dev |= 2 on a record that does
 *                              not belong to this rank's rows (never written).
 * Same slot, nq, k, n_probes, pass_1 and capacity as the scan call; the probe lists handed to
 * it must still be alive.  Results are identical to the dense exchange. */
int tk_index_shard_bound_dev(tk_index *ix, int slot, int64_t nq, int k, int n_probes, int pass_1,
                             int64_t capacity, const void *scan_dev, uint8_t *bound_dev,
                             void *stream);
/* ... and WITHOUT a host synchronisation (region_records >= 1; 0 = the compact form above, flag_dev / acc_dev
 * may then be NULL): the records of home rank h go to records_dev[h * region_records, ...) (room for world *
 * region_records records of 5 int32), so the all-to-all of the records has equal splits and can be enqueued
 * before any count is known; the counts (counts_dev[0, world)) travel in their own equal-split all-to-all and
 * are read by the home rank ON THE DEVICE: tk_index_shard_finish_filtered_dev with counts_recv_dev[world] takes
 * the received regions (region s from source rank s; n_records is then ignored), with counts_recv_dev == NULL
 * n_records compact records.  A home rank's records beyond region_records are dropped and *flag_dev |= 1 — the
 * batch's overflow flag, handled like a `capacity` overflow (region_records = capacity can never overflow;
 * callers trim it to what the batches need, multi_gpu.py).  Wire bytes: world * region_records * 20 per rank
 * instead of the exact 20 * records.  acc_dev (or NULL): int64[3] kept by the caller across batches, updated
 * atomically — [0] largest per-home count seen (what the regions must hold), [1] += records, [2] += blocks
 * scored.  (Until round 6 the two forms were four entry points: ..._filter_regions_dev / ..._finish_regions_dev.) */
int tk_index_shard_filter_dev(tk_index *ix, int slot, int64_t nq, int k, int n_probes, int pass_1,
                              int64_t capacity, const void *scan_dev, const uint8_t *bound_dev,
                              int32_t *counts_dev, int32_t *records_dev, int64_t region_records,
                              int *flag_dev, int64_t *acc_dev, void *stream);
int tk_index_shard_finish_filtered_dev(tk_index *ix, int slot, const float *q_dev, int64_t nq,
                                       int k, int n_probes, int pass_1, const int32_t *records_dev,
                                       int64_t n_records, const int32_t *counts_recv_dev,
                                       int64_t region_records, int64_t *out_ids_home_dev, int *flag_dev,
                                       void *stream);
/* The sharded scan in two phases, the second on the int8 matrix cores (tk_index_set_plain_scan's
 * kernel).  The plain sums are the reference's values for a query from the point where its heap's
 * bound is at most the limit C of its table, and B1 — the bound after the query's FIRST probed list
 * (ivf.py:137-152 over that list alone) — is known to every rank before the rest is scanned:
 *   tk_index_shard_scan_first_dev  as tk_index_shard_scan_dev up to the segment positions; the
 *                                  FIRST slots this rank owns scored exactly into send_dev;
 *                                  bound_dev[nq] = B1 (order key, as tk_index_shard_bound_dev) of
 *                                  the queries whose first list this rank owns, 255 elsewhere;
 *   all-reduce(MIN, uint8)         by the caller;
 *   tk_index_shard_scan_rest_dev   the slots behind the first: queries with B1 <= C on the plain
 *                                  kernel, the others on the exact one.  No query is scanned twice,
 *                                  nothing is repaired afterwards; tk_index_shard_finish_dev / the
 *                                  filtered exchange (with this bound: no tk_index_shard_bound_dev)
 *                                  follow as after tk_index_shard_scan_dev and return the same ids.
 * tk_index_shard_plain: 1 if the two-phase form applies to (k, n_probes, pass_1) — M <= 52,
 * n_probes >= 2, distinct labels or labels that repeat under a (verified or vouched) twin table, or
 * tk_index_set_plain_scan(ix, 2); not switched off; replicated
 * state only, so every rank answers alike — else 0 (use tk_index_shard_scan_dev), < 0 on error. */
int tk_index_shard_plain(tk_index *ix, int k, int n_probes, int pass_1);
int tk_index_shard_scan_first_dev(tk_index *ix, int slot, const float *q_dev, const void *q_pq_dev,
                                  int q_pq_is_f64, int64_t nq, int k, int n_probes, int pass_1,
                                  const int64_t *probes_all_dev, int64_t capacity, void *send_dev,
                                  int *flag_dev, uint8_t *bound_dev, void *stream);
int tk_index_shard_scan_rest_dev(tk_index *ix, int slot, int64_t nq, int k, int n_probes, int pass_1,
                                 int64_t capacity, void *send_dev, const uint8_t *bound_dev,
                                 void *stream);
/* Books of the slot's last tk_index_shard_scan_rest_dev (synchronises): out4 = pairs this rank
 * scored on the plain kernel, tiles of 32 of them, exact pair records of the slots behind the
 * first, queries (of all nq) whose bound let them go the plain way. */
int tk_index_shard_plain_stats(tk_index *ix, int slot, int64_t *out4);

#ifdef __cplusplus
}
#endif
#endif
