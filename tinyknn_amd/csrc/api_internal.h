// api_internal.h — what the translation units of the C ABI share (api.hip: host-pointer entry points and
// resident code arrays; api_index.hip: the resident index and its pipeline; api_shard.hip: the list-sharded
// entry points; api_build.hip: device build, raw queries, brute force, FlatTop).  Not part of the ABI.
#pragma once
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

int tk_fail(int code, const std::string &msg);      // sets tk_last_error() of this thread, returns code
static inline int fail(int code, const std::string &msg) { return tk_fail(code, msg); }

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

int tk_require_gpu();     // TK_OK, or the "no HIP device: no CPU fallback" error
static inline int require_gpu() { return tk_require_gpu(); }

// growable device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool borrowed = false;      // memory of another index (tk_index_clone_shard): never freed or grown here
    void borrow(const DevBuf &o)
    {
        release();
        p = o.p;
        cap = o.cap;
        borrowed = p != nullptr;
    }
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return TK_OK;
        if (borrowed) return fail(TK_ERR_ARG, "bad argument: a borrowed array of a cloned shard cannot grow");
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
        if (p && !borrowed) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        borrowed = false;
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

// buffers of ONE batch in flight
struct Work {
    DevBuf tables, shift, scale, cdist, cheap_idx, cheap_val, probes, slot_prefix, slot_chunk0,
        slot_n, slot_loff, dist, heap_idx, heap_val, repeat_flag, cmins, mins, u_count, u_cursor,
        u_pair_off, u_unit_prefix, u_pair_q, u_pair_f0, c_pair_off, c_unit_prefix, c_pair_q,
        c_pair_f0, spos, rpos, smins, pair_cnt, pair_off, scan_tmp, tally, usage, pos_lens, pos_off,
        qlim, slot_exact, p_count, p_cursor, p_pair_off, p_unit_prefix, p_pair_q, p_pair_f0, flag_list, p_unit_desc,
        plain0, h_count, h_cursor, h_pair_off, h_unit_prefix, h_pair_q, h_pair_f0,   // plain_scan.hip
        plain_q,                                                                       // two-phase sharded scan
        slots_desc;        // pool of TkSlotsOut for the coarse rescoring's epilogue (device copies)
    // The descriptors the fused epilogue reads are IMMUTABLE once written: a small pool per workspace,
    // entry i = slots_pinned[i] (page-locked host memory that outlives every call: the source of the
    // upload, also when that upload is recorded as a node of a hipGraph) and slots_desc[i] (device).
    // A batch looks its descriptor up by content; a new one takes the next free entry; a full pool
    // sends the batch to the unfused make_slots launch.  Nothing a captured graph refers to changes.
    TkSlotsOut *slots_pinned = nullptr;
    int slots_n = 0;
    unsigned slots_need_upload = 0;     // bit i: entry i was only uploaded inside a capture so far
    // list-sharded batch: what tk_index_shard_scan_dev left for the filtered exchange
    const int64_t *shard_probes = nullptr;
    int64_t shard_nq = 0, shard_capacity = 0;
    // list-sharded batch whose tables were built by the queries' HOME ranks and all-gathered by the caller
    // (tk_index_shard_coarse_home_dev / tk_index_shard_set_tables_dev): the gathered rows, in the caller's buffer
    bool shard_first = false;       // tk_index_shard_scan_first_dev ran: _rest_dev is owed
    bool shard_plain = false;       // tk_index_shard_scan_plain_dev filled the send buffer: the home replay checks the lemma
    bool shard_head = false;        // tk_index_shard_scan_head_dev ran: _scan_plain_dev(bound_dev) is owed
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
        if (slots_pinned) (void)hipHostFree(slots_pinned);
        slots_pinned = nullptr;
        slots_n = 0;
        slots_need_upload = 0;
        DevBuf *b[] = {&tables, &shift, &scale, &cdist, &cheap_idx, &cheap_val, &probes,
                       &slot_prefix, &slot_chunk0, &slot_n, &slot_loff, &dist, &heap_idx, &heap_val,
                       &repeat_flag, &cmins, &mins, &u_count, &u_cursor, &u_pair_off, &u_unit_prefix,
                       &u_pair_q, &u_pair_f0, &c_pair_off, &c_unit_prefix, &c_pair_q, &c_pair_f0,
                       &spos, &rpos, &smins, &pair_cnt, &pair_off, &scan_tmp, &tally, &usage, &pos_lens, &pos_off,
                       &qlim, &slot_exact, &p_count, &p_cursor, &p_pair_off, &p_unit_prefix, &p_pair_q, &p_pair_f0, &flag_list, &p_unit_desc,
                       &plain0, &h_count, &h_cursor, &h_pair_off, &h_unit_prefix, &h_pair_q, &h_pair_f0,
                       &plain_q, &slots_desc};
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
    bool labels24 = false;     // every label is in [0, 0xffffff): the register heap's duplicate test on one register per slot
    int opt_labels24 = 1;      // TK_OPT_LABELS24: 0 = (value, label64) entries even where the labels would fit (tests)
    // repeating labels: where the other copies of every stored row are (twins.hip); twin_w = copies - 1, 0 = no table
    DevBuf twin_list, twin_off;
    int twin_w = 0;
    // the table rests on two facts the index build checks where it holds every list's codes (twins.hip: copies of a
    // label lie in different lists and carry the same code); a rank that was handed only its own lists' codes
    // (tk_index_set_lists_shard) cannot, and uses the table only if the caller vouches (TK_OPT_TWIN_VOUCH)
    bool twin_unverified = false, twin_vouched = false;
    int opt_replay_twin = 1;   // TK_OPT_REPLAY_TWIN: 1 = the lane replay decides `insert`'s duplicate test from the twin table
    int64_t total_chunks = 0, total_ids = 0;
    int max_list_chunks = 0;
    bool ids_unique = false;   // no label occurs twice => the lane-per-query replay is exact
    int heap_mode = 0;         // 0 auto (pair for small batches, lanes, else packed wave), 1 general wave, 2 packed wave, 3 pair
    int opt_pair_nq = 8192;    // TK_OPT_PAIR_NQ: batches up to this many queries take the wave-per-query register heap
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
    DevBuf replay_counters;            // TK_OPT_REPLAY_COUNT: 4 x uint64 the list replays add to (tk_index_replay_stats)
    bool opt_replay_count = false;
    int opt_replay_lazy = -1;          // lane replay of the lists: 1 lazy (blocks fetched where their minimum passes), 0 staged, -1 auto
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

struct Plan {
    int kc, rescore, R, S;
    int64_t cap;       // uint4 per query in the distance buffer
    int64_t cap_min;   // bytes per query in the block-minimum buffer (multiple of 16)
    int64_t ccap_min;  // same for the coarse stage
};

#define TK_SLOTS_POOL 16
int reserve_slots_pool(Work &w);      // api_index.hip: the pool above exists (not inside a capture)

static const int64_t MAX_SUB = 32768;  // gridDim.y limit of the scan kernels is 65535
// a sharded batch only runs the list-major kernels (no gridDim.y); what bounds it is int32 unit
// counts and the 16 GB of distance rows per rank, both checked in shard_args
static const int64_t MAX_SHARD_BATCH = 131072;

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

// the distance tables a batch's scans read: the workspace's, or the gathered rows of a list-sharded batch
inline const uint4 *tables_of(const Work &w) { return w.tables.as<uint4>(); }

// ---- api_index.hip, used by the other files
int flush_pending(tk_index *ix);
// the twin table of an index whose int32 labels (all in [0, label_bound)) are in place; no table (twin_w = 0) where
// the labels are distinct, too sparse, or one label has more than 16 copies
int build_twins(tk_index *ix, int64_t label_bound);
bool twin_replay(const tk_index *ix, const struct Plan &p);
int make_plan(const tk_index *ix, int k, int n_probes, int pass_1, Plan &p);
bool plain_env_on();
int plain_k(const tk_index *ix, int64_t nq, const Plan &p);
size_t plain_desc_bytes(const tk_index *ix, int64_t nq, const Plan &p);
double workspace_bytes();
bool coarse_units(const tk_index *ix, int64_t nq);
// row0: the tables (and limits) of rows [row0, row0 + nq) of the workspace; qpq_dev points at row row0's query
int stage_tables(tk_index *ix, Work &w, const void *qpq_dev, int qpq_f64, int64_t nq, hipStream_t st, Prof &pf,
                 bool plain = false, TkSecond qpq2 = TkSecond(), int64_t row0 = 0);
TkScanJob coarse_job(const tk_index *ix, const Work &w, const Plan &p);
int plain_blocks();
int shard_plain_blocks();
void launch_coarse_scan(tk_index *ix, Work &w, int64_t nq, const Plan &p, hipStream_t st,
                        const uint4 *tables = nullptr);
int coarse_replay_probes(tk_index *ix, Work &w, const float *q_dev, int64_t nq, const Plan &p,
                         int64_t *probes_out, hipStream_t st, Prof &pf, TkSecond q2 = TkSecond(),
                         const TkSlotsOut *slots = nullptr, int *slots_written = nullptr);
void coarse_slots(tk_index *ix, Work &w, const int64_t *probes, int64_t nq, const Plan &p,
                  int *pair_count, const int *owner, int me, hipStream_t st, bool plain = false);
int stage_coarse_rest(tk_index *ix, Work &w, const float *q_dev, int64_t nq, const Plan &p,
                      int *pair_count, const int *owner, int me, hipStream_t st, Prof &pf,
                      bool plain = false, TkSecond q2 = TkSecond());
// plain_flag (list-sharded home rank): queries the plain path's lemma does not cover raise bit 4 of
// *plain_flag instead of being scanned again (the codes are on other ranks: the batch is repeated)
int stage_back(tk_index *ix, Work &w, const float *q_dev, int64_t q0, int64_t nq, int k,
               const Plan &p, int64_t *out_dev, hipStream_t st, Prof &pf, bool plain = false,
               TkSecond q2 = TkSecond(), TkSecond out2 = TkSecond(), int *plain_flag = nullptr);
int head_chunks_of(const tk_index *ix, const Plan &p);    // chunks of a first probed list the exact kernel keeps (head mode)
