// kernels.h — launch wrappers of the gfx950 kernels (internal to libtinyknn_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TK_ORDER_SSE 0
#define TK_ORDER_AVX 1

// Device code layout ("tiled"): chunk c (16 rows), block pair p (16 bytes holding
// code[16c+r][2p] | code[16c+r][2p+1] << 4 for r = 0..15, i.e. the reference's
// 16-byte group, _transform.py:4-77) lives at uint4 index
//     ((c / 8) * P + p) * 8 + (c % 8),      P = M / 2
// so that 8 consecutive chunks share one 128-byte line per pair and a wave whose
// lanes own consecutive chunks reads whole lines.  The chunk count is padded to a
// multiple of 8 with zero chunks.
// ints in a unit_prefix array for n_lists lists: the prefix table, then (128-byte aligned)
// the 8 work counters of tk_launch_scan_units, 128 bytes apart
static inline size_t tk_unit_prefix_ints(int64_t n_lists)
{
    return (size_t)(((n_lists + 1 + 31) / 32 + 1) * 32 + 8 * 32);
}

static inline int64_t tk_tiled_uint4s(int64_t chunks, int P)
{
    return ((chunks + 7) / 8) * 8 * (int64_t)P;
}

// reference layout (chunks, M) uint64  ->  tiled layout, `chunk0` = first
// destination chunk (must be a multiple of 8 unless the caller owns the tile)
void tk_launch_retile(const uint4 *src_ref, uint4 *dst_tiled, int64_t chunks, int P,
                      hipStream_t s);

// Flat scan: every chunk of one code array against nq tables.
// tables: (nq, M) uint4 (16 table bytes per block); out: (nq, out_stride) uint4,
// 16 int8/uint8 distances per chunk.
// mins (optional): (nq, min_stride) bytes, the minimum of each chunk's 16 distances.
void tk_launch_scan_flat(const uint4 *codes, int64_t chunks, int M, const uint4 *tables,
                         int64_t nq, uint4 *out, int64_t out_stride, uint8_t *mins,
                         int64_t min_stride, int signd, int order, hipStream_t s);

// Probed-list scan: query q scans the lists slot 0..S-1 named by
// slot_chunk0[q][s] (first global chunk) and slot_prefix[q][s..s+1] (flat chunk
// range inside the query's distance row).  dist: (nq, cap) uint4.
// only ([count, q_0, q_1, ...] from tk_launch_flagged_list, or NULL): score only these queries
void tk_launch_scan_probes(const uint4 *codes, int M, const uint4 *tables, int64_t nq,
                           const int *slot_prefix, const int64_t *slot_chunk0, int S,
                           int max_flat_chunks, uint4 *dist, int64_t cap, uint8_t *mins,
                           int64_t min_stride, int signd, int order, hipStream_t s,
                           const int *only = nullptr);
// list (nq + 1 ints): list[0] = number of flagged queries, list[1..] = their ids in order
// host_count (page-locked host word or NULL) receives the count too
void tk_launch_flagged_list(const unsigned char *flags, int64_t nq, int *list, hipStream_t s,
                            int *host_count = nullptr);

// List-major form of the probed-list scan for large batches.  tk_launch_unit_pairs
// groups the (query, slot) pairs by list (scan + fill kernels; `count` comes from
// tk_launch_make_slots and is left zeroed again);
// tk_launch_scan_units scores each chunk for four queries per pass.  Same outputs
// as tk_launch_scan_probes.  Work arrays: count, cursor (n_lists), pair_off,
// unit_prefix (n_lists+1), pair_q, pair_f0 (max_records >= nq*S + 4*n_lists).
void tk_launch_unit_pairs(int64_t nq, const int64_t *probes, int S, int64_t n_lists,
                          const int64_t *list_chunk_off, const int *slot_prefix, int *count,
                          int *pair_off, int *unit_prefix, int *cursor, int *pair_q, int *pair_f0,
                          int64_t max_records, hipStream_t s);
// descriptors for one list scanned by every query (pair_off/unit_prefix: 2 ints,
// pair_q/pair_f0: nq rounded up to a multiple of 4)
void tk_launch_identity_pairs(int64_t nq, int chunks, int *pair_off, int *unit_prefix, int *pair_q,
                              int *pair_f0, hipStream_t s);
// one job of the list-major scan (what tk_launch_scan_units takes); unit_prefix == NULL: absent
struct TkScanJob {
    const uint4 *codes;
    const uint4 *tables;
    const int64_t *list_chunk_off;
    int n_lists;
    const int *unit_prefix, *pair_off, *pair_q, *pair_f0;
    uint4 *dist;
    int64_t cap;
    uint8_t *mins;
    int64_t min_stride;
    int max_chunks = 0;               // list-major kernel: only the first max_chunks chunks of a list (0: all)
    const int *unit_desc4 = nullptr;  // plain kernel: int4 (list, tile, first chunk pair, end chunk pair) of every unit
};
// ---- plain-sum scan on the int8 matrix cores (plain_scan.hip) ----
// qlim[q]: the bound below which clamp(plain sum) IS the reference's saturated value for query q
// (C of the lemma in plain_scan.hip), or TK_PLAIN_NEVER when the query's table rules it out
#define TK_PLAIN_NEVER (-(1 << 30))
#define TK_PLAIN_COUNTER_OFF(n_lists) ((((n_lists) + 1 + 31) / 32 + 1) * 32)
// force: a cap on every query's limit (the tests' way to provoke the re-scan path); INT_MAX = none
void tk_launch_table_limits(const uint4 *tables, int M, int order, int64_t nq, int *qlim, hipStream_t s,
                            int force = 0x7fffffff);
int tk_plain_fits(int M);
// TkScanJob with unit_prefix = tiles of 32 pairs before each list (+ the work counter at
// TK_PLAIN_COUNTER_OFF); pair records are not padded.  Returns -1 for unsupported M.
int tk_launch_scan_plain(const TkScanJob &j, int M, int order, int n_blocks, hipStream_t s);
// the second pair set (plain pairs) beside the first in ONE pass: see tk_launch_unit_pairs2
struct TkPairSet {
    int *count, *cursor, *pair_off, *unit_prefix, *pair_q, *pair_f0;
    int *unit_desc = nullptr;      // plain set: int4 (list, tile, first chunk pair, end chunk pair) per unit
    int plain_k = 12;              // plain set: chunk pairs per unit, a multiple of 4 (tk_plain_units_bound)
};
// descriptors of the plain kernel for ONE list scanned by every query (pair i = query i, row offset 0):
// pair_off / unit_prefix of one list, pair_q / pair_f0: nq ints, unit_desc: ceil(nq/32) * ceil(CP/K) int4
void tk_launch_plain_identity(int64_t nq, int chunks, const TkPairSet &pl, hipStream_t s);
// the plain set's unit descriptors alone (tk_launch_unit_pairs2 writes them itself; callers of
// tk_launch_pairs_scan3 that fill the records their own way call this behind it)
void tk_launch_plain_desc(const TkPairSet &pl, const int64_t *list_chunk_off, int64_t n_lists, hipStream_t s);
// upper bound of the plain set's units for nq * S pairs over n_lists lists
static inline int64_t tk_plain_units_bound(int64_t pairs, int64_t n_lists, int64_t total_chunks, int max_list_chunks, int K)
{
    const int64_t nsub_max = (max_list_chunks / 2 + 1 + K - 1) / K + 1;
    return (pairs / 32 + 1) * nsub_max + total_chunks / (2 * (int64_t)K) + 2 * n_lists + 8;
}
// (query, slot) pairs grouped by list, split at slot_exact[q]: slots below it -> set `ex` (units
// of the list-major exact kernel), the others -> set `pl` (tiles of the plain kernel)
// hd (head pairs: the first probed list of a query in head mode, slot_exact[q] == 0): units of the
// exact kernel over the first head_chunks chunks of a list only; such a pair is ALSO in set pl
void tk_launch_unit_pairs2(int64_t nq, const int64_t *probes, int S, int64_t n_lists,
                           const int64_t *list_chunk_off, const int *slot_prefix,
                           const int *slot_exact, const TkPairSet &ex, const TkPairSet &pl,
                           const TkPairSet &hd, int head_chunks, hipStream_t s, const int *pos = nullptr,
                           const int *owner = nullptr, int me = 0, int ovf_pos = 0);
// pos / owner / me / ovf_pos: list-sharded index — records only for the lists `me` owns, row offsets =
// positions in the send buffer (see pairs_fill3_kernel)

// two jobs in ONE launch sharing one pool of 64-unit blocks (pipelined mode: the list scan
// of one batch and the coarse scan of the next); signed tables
void tk_launch_scan_units2(const TkScanJob &a, const TkScanJob &b, int M, int order, int n_blocks,
                           hipStream_t s, const TkScanJob *c = nullptr, int form = 0);
void tk_launch_scan_units(const uint4 *codes, int M, const uint4 *tables, int64_t nq, int S,
                          int64_t n_lists, const int64_t *list_chunk_off, const int *pair_off,
                          const int *unit_prefix, const int *pair_q, const int *pair_f0,
                          uint4 *dist, int64_t cap, uint8_t *mins, int64_t min_stride, int signd,
                          int order, int n_blocks, hipStream_t s, int form = 0);

// Exact replay of the reference's sequential heap over precomputed distances.
// One wave per query.  slot_n: true rows per slot; slot_label_off: offset into
// `labels` or -1 (label = position).  heap_idx/heap_val: (nq, R) in/out.
// slots_uniform: the slot arrays hold ONE row that every query uses.
// only_flagged (nq bytes or NULL): replay only the flagged queries, each from a
// fresh heap (the second pass behind tk_launch_heap_replay_lanes).
void tk_launch_heap_replay(const uint4 *dist, int64_t cap, int64_t nq, const int *slot_prefix,
                           const int *slot_n, const int64_t *slot_label_off, int S,
                           const int64_t *labels, int64_t *heap_idx, int32_t *heap_val, int R,
                           int signd, int slots_uniform, const unsigned char *only_flagged,
                           hipStream_t s, const uint8_t *mins = nullptr, int64_t cap_min = 0);
// mins (optional): the scan's per-block minima, (nq, cap_min) bytes, cap_min a multiple of 16:
// slots that start at a multiple of 16 blocks are then walked 1024 blocks per step

// one query, a FRESH heap of R <= 64 entries, rows far longer than it: one workgroup — head of h blocks
// (a multiple of 16) replayed with the heap in registers, the later blocks whose minimum is below the
// bound reached there compacted in order, those replayed; the heap goes to out_idx / out_val (which may
// be pinned host memory).  cdist: chunks uint4, cblock: chunks int + chunks bytes.  See heap.hip
void tk_launch_flat_top_one(const uint4 *dist, const uint8_t *mins, int chunks, int h, int64_t n, int R, int signd,
                            uint4 *cdist, int *cblock, int64_t *out_idx, int32_t *out_val, hipStream_t s);

// twins.hip's table + the probe lists of the batch, for the TWIN form of the lane replay
struct TkTwins {
    const int32_t *list = nullptr;    // (total rows, w): list of the u-th other copy of a stored row, -1 = none
    const int32_t *off = nullptr;     // ... its offset inside that list
    int w = 0;
    const int64_t *probes = nullptr;  // (nq, S) the probed lists of every query (ivf.py:131)
    int bm_words = 0;                 // tk_lanes_twin_bm_words(n_lists): dwords of the per-query list bitmap, 0 = none
};

// Lane-per-query form of the same replay: 64 queries per wave.  Preconditions
// (checked by the caller): heaps start fresh (-1 / 127|255), no label can repeat
// among a query's lists (so `insert`'s duplicate test cannot fire), R*256 B of LDS
// <= 160 KiB, cap*16 < 2^24 codes per query.  Writes (nq, R) heaps.  Returns 0.
// skip (nq bytes or NULL): queries to leave untouched.  mins: (nq, cap_min) per-block
// minima written by the scan kernels, cap_min a multiple of 16.
#define TK_LANES_MAX_R 574
// with labels32 (labels may repeat: duplicate test on 32-bit label slots + a two-choice hash
// set of the labels in the heap, 64 KiB) the LDS budget is (2R+2)*256 + 64 KiB + 16 KiB + the
// slot table
#define TK_LANES_MAX_R_DEDUPE 149
int tk_lanes_dedupe_fits(int R, int S);     // ... and the slot table of S probed lists fits too
int tk_lanes_twin_bm_words(int64_t n_lists);
int tk_lanes_twin_fits(int R, int S, int64_t n_lists);   // the TWIN form: heap columns + slot table + probe list + list bitmap
// plain0 / qlim (both nq ints, or NULL): the blocks from flat chunk plain0[q] on carry clamp(plain
// sums) (plain_scan.hip); a query whose bound at its first such block is above qlim[q] gets
// skip[q] = 1 written (skip must then be writable) and is to be re-scanned exactly and replayed again.
int tk_launch_heap_replay_lanes(const uint4 *dist, int64_t cap, int64_t nq, const int *slot_prefix,
                                const int *slot_n, const int64_t *slot_label_off, int S,
                                const int64_t *labels, int64_t *heap_idx, int32_t *heap_val, int R,
                                int signd, int slots_uniform, unsigned char *skip,
                                const uint8_t *mins, int64_t cap_min, const int32_t *labels32,
                                hipStream_t s, const int *plain0 = nullptr, const int *qlim = nullptr,
                                int lazy = 0, unsigned long long *counters = nullptr,
                                const TkTwins *twins = nullptr, int *flag_list = nullptr);
// flag_list (nq + 1 ints, or NULL): [0] = how many queries this replay leaves to the kernels behind it (flagged in
// `skip` before it, or by its own check), then their numbers in any order — tk_launch_scan_probes' `only`
// lazy: blocks are fetched only where their minimum passes (rows far longer than the heap; distinct labels)
// twins (labels32 == NULL): labels may repeat, every label's copies carry ONE value, and the duplicate test is
// decided from the twin table (heap.hip, TWIN form); `skip` must flag the queries that probe a list twice

// Wave-per-query replay on packed 32-bit entries from FRESH heaps (R*4 B of LDS, or
// R*12 with `dedupe`: int64 labels per slot + the reference's duplicate-label test,
// for labels that can repeat).  flags/run_if: only queries with flags[q] == run_if
// (flags may be NULL).  cap*16 < 2^24 (no dedupe) / R < 2^24.
// Wave-per-query replay with the heap in registers, two / four / eight nodes per lane for heaps of up to 129 / 257 / 513
// entries (heap.hip): R <= TK_PAIR_MAX_R, fresh heaps;
// position entries where labels are distinct (cap * 16 <= 0xffffff), the reference's duplicate test on (value, label)
// entries for the queries with flags[q] != 0 or for every query (dedupe_all).  The kernel of one query per call.
// plain0 / qlim / flag_list: the per-query check of plain_scan.hip's lemma, as tk_launch_heap_replay_lanes makes it.
#define TK_PAIR_MAX_R 513
int tk_launch_heap_replay_pair(const uint4 *dist, int64_t cap, int64_t nq, const uint8_t *mins, int64_t cap_min,
                               const int *slot_prefix, const int *slot_n, const int64_t *slot_label_off, int S,
                               const int64_t *labels, int64_t *heap_idx, int32_t *heap_val, int R, int signd,
                               int slots_uniform, unsigned char *flags, int dedupe_all, hipStream_t s,
                               const int *plain0 = nullptr, const int *qlim = nullptr, int *flag_list = nullptr,
                               int only_flagged = 0, const int *count_src = nullptr, int *host_count = nullptr);
void tk_launch_heap_replay_packed(const uint4 *dist, int64_t cap, int64_t nq, const int *slot_prefix,
                                  const int *slot_n, const int64_t *slot_label_off, int S,
                                  const int64_t *labels, int64_t *heap_idx, int32_t *heap_val,
                                  int R, int signd, int slots_uniform, const unsigned char *flags,
                                  int run_if, int dedupe, hipStream_t s, const int *flag_list = nullptr,
                                  int *host_count = nullptr);
// run_if < 0: every query with a non-zero flag.  flag_list + host_count (page-locked): *host_count = flag_list[0]

void tk_launch_heap_fill(int64_t *heap_idx, int32_t *heap_val, int64_t count, int32_t v,
                         hipStream_t s);

// twins.hip: the other copies of every stored row (labels that repeat: IVF.build(n_probes >= 2))
void tk_launch_twin_count(const int32_t *ids32, int64_t T, int *cnt, int *cnt_max, hipStream_t s);
// *flag |= 1: two copies of a label with different codes; |= 2: two copies in one list (the TWIN form then does not apply)
void tk_launch_twin_verify(const uint4 *codes, int P, const int64_t *list_chunk_off, const int64_t *ids_off, int n_lists,
                           const int32_t *twin_list, const int32_t *twin_off, int w, int64_t T, int *flag, hipStream_t s);
void tk_launch_twin_fill(const int32_t *ids32, int64_t T, int *cursor, int *where, int b, const int64_t *ids_off,
                         int n_lists, int32_t *twin_list, int32_t *twin_off, hipStream_t s);
// single insert on a device heap (insertion-sort variant when `is`)
void tk_launch_heap_insert(int64_t *heap_idx, int32_t *heap_val, int R, int64_t i, int32_t v,
                           int is, hipStream_t s);

// Distance tables (fast_pq.py:186-252).  q: (nq, dq) float or double.
// Rows of a batch made of two calls (tk_index_set_coalesce): rows [n_a, nq) live in a second buffer
// whose row 0 is row n_a of the batch.  Empty (b == nullptr): one buffer.
struct TkSecond {
    const void *b = nullptr;
    int64_t n_a = 0;
};

// what the table build can do on the side for the batch pipeline (every wave has its query's table at hand): the
// limit C of plain_scan.hip's lemma per query (tk_launch_table_limits' result: qlim, avx order?, blocks the
// reference's kernel reads, cap), and the descriptors of "every query scans the one list of coded centres"
// (tk_launch_identity_pairs' result for c_nq <= nq queries over c_chunks chunks) — two kernels less at the head
// of the front stream's chain
#define TK_UNIT_Q 4          // queries a lane of the list-major exact kernel scores per pass (records padded to it)
struct TkTablesExtra {
    int *qlim = nullptr;
    int lim_avx = 0, lim_m_used = 0, lim_force = 0x7fffffff;
    int *c_pair_off = nullptr, *c_unit_prefix = nullptr, *c_pair_q = nullptr, *c_pair_f0 = nullptr;
    int c_chunks = 0;
    int64_t c_nq = 0;
};
void tk_launch_build_tables(const float *centers, int dq, int dpb, int f_order, const void *q,
                            int q_is_f64, int64_t nq, double aux0, double aux1, int signd,
                            uint8_t *tables, void *shift, double *scale, hipStream_t s, TkSecond q2 = TkSecond(),
                            const TkTablesExtra *extra = nullptr);

// Exact rescoring + ascending top-k.
// cand: (nq, R) int64 candidate ids (heap order).  strip: drop -1 entries first
// (ivf.py:154-155) — without it, negative ids address rows from the end
// (fast_pq.py:311).  If the candidate count <= k the ids are returned in heap
// order (ivf.py:158-159 / fast_pq.py:307-308).  out: (nq, k) padded with -1;
// out_count (nq,) optional.
// q / rows: float32 or float64 (flags); float64 arithmetic if either is float64.
// make_slots (below) as the epilogue of the coarse rescoring: the wave that ranked a query's probed
// lists writes their scan descriptors and counts the (query, list) pairs — one kernel less on the
// stream whose chain of short kernels the pipelined mode waits for.  Arguments as tk_launch_make_slots.
struct TkSlotsOut {
    int64_t n_lists;
    const int64_t *list_chunk_off, *list_n, *ids_off;
    int *slot_prefix;
    int64_t *slot_chunk0;
    int *slot_n;
    int64_t *slot_label_off;
    unsigned char *repeat_flag;
    int *pair_count;
    const int *owner;
    int me;
    const int *qlim;
    int R;
    int *slot_exact, *pair_count2, *plain0, *pair_count3;
};
// `slots`: a copy of the structure in DEVICE memory (17 pointers as kernel arguments would sit in scalar
// registers for the whole kernel).  Returns 1 if it was given AND written by the rescoring kernel
// (float32 operands staged through LDS, k <= 64), else 0: the caller then launches tk_launch_make_slots
int tk_launch_rescore(const void *q, int q_is_f64, int d, const void *rows, int rows_is_f64,
                      int64_t n_rows, const int64_t *cand, int R, int64_t nq, int k, int strip,
                      int64_t *out, int *out_count, hipStream_t s, int form = 2, TkSecond q2 = TkSecond(),
                      TkSecond out2 = TkSecond(), const TkSlotsOut *slots = nullptr);

// probes (nq, kc) list ids -> per-slot scan descriptors; pair_count (n_lists, zeroed, or
// NULL) receives the number of (query, slot) pairs per list — with `owner` (n_lists ranks,
// list-sharded index) only for the lists owned by `me`
// qlim / R (= the rows the exact kernel keeps: 2 x the heap size) / slot_exact / plain0 / pair_count2 /
// pair_count3 (all or none): the exact kernel keeps
// what a query scans until the heap is full of real values and its bound far below the table's
// limit — R rows.  First list at least that long ("head mode", slot_exact[q] = 0): its first
// ceil(R / 16) chunks (pair counted in pair_count3 AND, for the plain kernel, in pair_count2);
// otherwise the leading slot_exact[q] >= 1 lists that hold R rows together.  All of them for a query
// whose table rules the plain sums out (qlim = TK_PLAIN_NEVER) or whose probe list wrapped.
// plain0[q] = first flat chunk of the query's row that carries plain sums.
void tk_launch_make_slots(const int64_t *probes, const int *probe_count, int kc, int64_t nq,
                          int64_t n_lists, const int64_t *list_chunk_off, const int64_t *list_n,
                          const int64_t *ids_off, int *slot_prefix, int64_t *slot_chunk0,
                          int *slot_n, int64_t *slot_label_off, unsigned char *repeat_flag,
                          int *pair_count, const int *owner, int me, hipStream_t s,
                          const int *qlim = nullptr, int R = 0, int *slot_exact = nullptr,
                          int *pair_count2 = nullptr, int *plain0 = nullptr, int *pair_count3 = nullptr);

// the exclusive scans of tk_launch_unit_pairs alone (the caller fills the records)
void tk_launch_pairs_scan(int *count, const int64_t *list_chunk_off, int64_t n_lists, int *pair_off,
                          int *unit_prefix, int *cursor, int *pair_q, hipStream_t s);

// ---- list-sharded index (shard.hip) ----
// positions of the (query, slot) segments in the all-to-all buffers: W regions of C uint4;
// home rank of query i = i / qh.  spos: (nq, S), rpos: (qh, S); *flag |= 1 on overflow.
// lens / P: nq*S + W*qh*S + 1 int64 each; tmp: room for tk_scan_exclusive64 over that many;
// usage: 2 * W stream lengths (uint4) or NULL.  Returns -1 if the prefix sum could not run.
int tk_scan_exclusive64(void *tmp, size_t *tmp_bytes, const long long *in, long long *out, int64_t n,
                        hipStream_t s);
int tk_launch_shard_positions(const int64_t *probes, const int *slot_prefix, int S, int64_t nq,
                              int64_t n_lists, const int *owner, int me, int W, int64_t qh,
                              int64_t C, int *spos, int *rpos, int *flag, long long *usage,
                              long long *lens, long long *P, void *tmp, size_t tmp_bytes,
                              hipStream_t s);
void tk_launch_shard_pairs_fill(const int64_t *probes, int S, int64_t nq, int64_t n_lists,
                                const int *owner, int me, const int *spos, const int *pair_off,
                                int *cursor, int *pair_q, int *pair_f0, hipStream_t s, int s_lo = 0,
                                int s_hi = 0x7fffffff, const uint8_t *sel = nullptr, int want = 0,
                                int ovf_pos = -1);
// two-phase scan: pairs per list of the first slots; of the slots behind them by kernel (plain_q)
void tk_launch_shard_count_first(const int64_t *probes, int S, int64_t nq, int64_t n_lists,
                                 const int *owner, int me, int *count, hipStream_t s);
void tk_launch_shard_count_rest(const int64_t *probes, int S, int64_t nq, int64_t n_lists,
                                const int *owner, int me, const uint8_t *bound, const int *qlim,
                                int allow, uint8_t *plain_q, int *count_exact, int *count_plain,
                                hipStream_t s);
// the exclusive scans of tk_launch_unit_pairs2 alone (the caller fills the records)
void tk_launch_pairs_scan3(const TkPairSet &ex, const TkPairSet &pl, const TkPairSet &hd,
                           const int64_t *list_chunk_off, int64_t n_lists, int head_chunks,
                           hipStream_t s);
// flags[q] == 2 for any q < nq: *flag |= 4 (a home query the plain path's lemma does not cover)
void tk_launch_shard_flag_plain(const unsigned char *flags, int64_t nq, int *flag, hipStream_t s);
// slot_prefix: rows of the home queries; dist/mins: (nq_home, cap) / (nq_home, min_stride)
void tk_launch_shard_unpack(const uint4 *recv, const int *rpos, const int *slot_prefix, int S,
                            int64_t nq_home, uint4 *dist, int64_t cap, uint8_t *mins,
                            int64_t min_stride, int signd, hipStream_t s);
// filtered exchange: bound after the first probed list (order key = byte ^ 0x80,
// 255 where this rank does not own the query's first list), blocks below it as 5-int records
// grouped by destination (counts: 3 * W ints, zeroed by the caller), rows rebuilt at home
void tk_launch_shard_first_bound(const int64_t *probes, const int *slot_prefix, const int *slot_n,
                                 int S, int64_t nq, int64_t n_lists, const int *owner, int me,
                                 const int *spos, const uint4 *scan, const uint8_t *smins, int R,
                                 uint8_t *bound, hipStream_t s, int max_chunks = 0);
// qlim[q] = TK_PLAIN_NEVER where the (order-key) bound is above it
void tk_launch_shard_mask_limits(const uint8_t *bound, int64_t nq, int *qlim, hipStream_t s);
// pair_cnt / pair_off: nq * S + 1 ints (pair_cnt's last entry zeroed by the caller); tmp: room
// for tk_scan_exclusive over that many; tally: 256 * W ints, zeroed by the caller
int tk_scan_exclusive(void *tmp, size_t *tmp_bytes, const int *in, int *out, int64_t n, hipStream_t s);
int tk_launch_shard_filter(const int64_t *probes, const int *slot_prefix, int S, int64_t nq,
                           int64_t n_lists, const int *owner, int me, int W, int64_t qh,
                           int64_t cap, const int *spos, const uint4 *scan, const uint8_t *smins,
                           const uint8_t *bound, int *pair_cnt, int *pair_off, void *tmp,
                           size_t tmp_bytes, int *tally, int *counts, int *rec, hipStream_t s,
                           int region = 0, int *flag = nullptr, long long *acc = nullptr);
void tk_launch_shard_expand(const int *rec, int64_t n_rec, const int *slot_prefix, int S,
                            int64_t nq_home, uint4 *dist, int64_t cap, uint8_t *mins,
                            int64_t min_stride, int *bad, hipStream_t s,
                            const int *counts_recv = nullptr, int region = 0);

// ---- offline build path (build.hip) ----
// labels (n, M) uint8 = FastPQ.transform's per-block nearest centroid; data: (n, dq) padded
// (rotated) rows on the device.  Returns -1 when the codebook does not fit 64 KiB of LDS.
int tk_launch_encode_pq(const float *centers, int dq, int dpb, const void *data, int is_f64,
                        int64_t n, uint8_t *labels, hipStream_t s);
// out = X / np.linalg.norm(X, axis=1, keepdims=True), float32 rows, d <= 128
// front.hip <-> api.hip (not part of the C ABI)
struct tk_index;
void tk_launch_copy_words(const void *src, int64_t n_words, void *dst, hipStream_t st);
void tk_index_host_out_by_kernel(tk_index *ix, bool on);
void tk_launch_normalise_rows(const float *X, int64_t n, int d, float *out, hipStream_t s);
// knn_brute(X, Y, k <= 2, "euclidean"): Yt (d, L) = Y transposed, ynorm2 (L,) = einsum |y|^2,
// float32 or float64 both; nearest (n, k)
void tk_launch_assign(const float *X, int64_t n, int d, const void *Yt, const void *ynorm2,
                      int y_is_f64, int L, int k, int64_t *nearest, hipStream_t s);
// device front end (fast mode): padded (Rt == NULL: float32 (n, dq)) or rotated (float64 (n, dq),
// Rt = R transposed (d_pad, dq)) table-build queries from raw/normalised float32 rows
void tk_launch_prepare_queries(const float *X, int64_t n, int d, const double *Rt, int dq, int d_pad,
                               void *out, hipStream_t s);

// ---- device-resident build + seeded generator (devbuild.hip) ----
void tk_launch_synth_rows(float *X, int64_t row0, int64_t n, int d, uint64_t seed, const float *centres,
                          int n_centres, float sigma, hipStream_t s);
void tk_launch_keys_count(const int64_t *nearest, int64_t n, int kp, int64_t row0, int64_t N, int *keys,
                          int *rows, int *count, hipStream_t s);
void tk_launch_remap_keys(int *keys, int64_t n, const int *remap, hipStream_t s);
int tk_sort_pairs(void *tmp, size_t *tmp_bytes, const int *keys_in, int *keys_out, const int *vals_in,
                  int *vals_out, int64_t n, int bits, hipStream_t s);
void tk_launch_widen_ids(const int *rows, int64_t n, int64_t *ids, hipStream_t s);
void tk_launch_pack_lists(const uint8_t *labels, int M, const int *rows_sorted, const int64_t *ids_off,
                          const int64_t *chunk_off, const int64_t *list_n, int n_lists,
                          const uint8_t *zero_code, uint4 *tiled, int64_t total_chunks, hipStream_t s);
void tk_launch_gather_rows(const float *X, int d, const int64_t *rows, int64_t n, float *out, hipStream_t s);
void tk_launch_read_only(const void *src, int64_t n_uint4, uint32_t *out, hipStream_t s);
// n_gather random rows of row_bytes (a multiple of 16, <= 1024) out of n_rows, read as the rescoring kernel reads
void tk_launch_gather_rows(const void *src, int64_t n_rows, int row_bytes, int64_t n_gather, uint32_t *out, hipStream_t s);
void tk_launch_compact_tiled(const uint4 *src, uint4 *dst, int P, const int64_t *global_off,
                             const int64_t *local_off, int n_lists, int64_t local_chunks, hipStream_t s);

// ---- exact k nearest rows on the f32 matrix cores (brute.hip) ----
// see brute.hip for the work buffers; returns -1 on unsupported sizes
int tk_launch_knn_brute(const float *X, int64_t nq, int d, const float *Y, int64_t N, int k,
                        float *ynorm2, float *vals, int64_t ns, float *tau,
                        unsigned long long *cand, int cap, int *count, int *overflow, int64_t *out,
                        float *sample, hipStream_t s);
