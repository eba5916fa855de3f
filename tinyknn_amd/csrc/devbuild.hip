// devbuild.hip — IVF.build for vectors that already live in HBM, and the seeded generator
// of the synthetic 100M x 128 configuration ("per-GPU generation on device
// (seeded per shard), codes produced by the build's encoder"; 8f.1).
//
// What IVF.build does (ivf.py:77-102, build n_probes = 1) without the vectors ever visiting
// the host: nearest centre per row (assign kernels of build.hip), nearest centroid per block
// (encode_pq_kernel), rows grouped by list (stable radix sort of (list, row): a list holds its
// rows in ascending row order, where numpy's unstable argsort leaves the order unspecified),
// 16-row chunks packed into the Quick-ADC byte layout (_transform.py:4-77) directly in the
// tiled order the scan kernels read.  api.hip drives these kernels (tk_index_build_dev).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>      // radix_sort_pairs of the offline build only (ROCm's own primitives library)

#include "kernels.h"

// ---------------------------------------------------------------------------
// counter-based generator: every value is a pure function of (seed, row, column pair), so a
// data set does not depend on how its rows are split over calls, slabs or ranks
__device__ __forceinline__ uint64_t tk_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// X[i] = centres[c(row)] + sigma * N(0, 1), c(row) uniform in [0, n_centres) (no centres: the
// noise alone).  One thread per (row, pair of columns): Box-Muller gives two normals.
__global__ void synth_rows_kernel(float *__restrict__ X, int64_t row0, int64_t n, int d, uint64_t seed,
                                  const float *__restrict__ centres, int n_centres, float sigma)
{
    const int hp = (d + 1) >> 1;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * hp) return;
    const int64_t r = i / hp;
    const int pr = (int)(i - r * hp);
    const uint64_t row = (uint64_t)(row0 + r);
    const uint64_t u = tk_mix64(seed ^ tk_mix64(row * 0x100000001B3ull + (uint64_t)pr));
    const float u1 = ((float)(uint32_t)(u >> 32) + 1.0f) * 2.3283064365386963e-10f;   // (0, 1]
    const float u2 = (float)(uint32_t)u * 2.3283064365386963e-10f;                    // [0, 1)
    const float rad = sqrtf(-2.0f * logf(u1));
    float z0 = rad * cosf(6.283185307179586f * u2), z1 = rad * sinf(6.283185307179586f * u2);
    float b0 = 0.f, b1 = 0.f;
    const int c0 = 2 * pr, c1 = 2 * pr + 1;
    if (centres) {
        const int64_t c = (int64_t)(tk_mix64(seed * 31 + row) % (uint64_t)n_centres);
        b0 = centres[c * d + c0];
        if (c1 < d) b1 = centres[c * d + c1];
    }
    X[r * d + c0] = b0 + sigma * z0;
    if (c1 < d) X[r * d + c1] = b1 + sigma * z1;
}

void tk_launch_synth_rows(float *X, int64_t row0, int64_t n, int d, uint64_t seed, const float *centres,
                          int n_centres, float sigma, hipStream_t s)
{
    // slabs: a launch addresses at most 2^32 - 1 threads (grid x block)
    const int hp = (d + 1) / 2;
    const int64_t slab = (int64_t)1 << 22;
    for (int64_t o = 0; o < n; o += slab) {
        const int64_t m = n - o < slab ? n - o : slab;
        hipLaunchKernelGGL(synth_rows_kernel, dim3((unsigned)((m * hp + 255) / 256)), dim3(256), 0, s,
                           X + o * d, row0 + o, m, d, seed, centres, n_centres, sigma);
    }
}

// ---------------------------------------------------------------------------
// nearest (n, kp) int64 of a slab -> sort keys / values + the per-centre histogram.  Column j of
// the rows goes to block j of the pair arrays (block stride N): group_data_by_indices
// (utils.py:131-150) appends a list's column-0 members first, then its column-1 members, and a
// stable sort of the pairs by list keeps exactly that order.
__global__ void keys_count_kernel(const int64_t *__restrict__ nearest, int64_t n, int kp, int64_t row0,
                                  int64_t N, int *__restrict__ keys, int *__restrict__ rows,
                                  int *__restrict__ count)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * kp) return;
    const int64_t r = i / kp;
    const int j = (int)(i - r * kp);
    const int k = (int)nearest[i];
    keys[(int64_t)j * N + row0 + r] = k;
    rows[(int64_t)j * N + row0 + r] = (int)(row0 + r);
    atomicAdd(&count[k], 1);
}

void tk_launch_keys_count(const int64_t *nearest, int64_t n, int kp, int64_t row0, int64_t N, int *keys,
                          int *rows, int *count, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(keys_count_kernel, dim3((unsigned)((n * kp + 255) / 256)), dim3(256), 0, s, nearest, n,
                       kp, row0, N, keys, rows, count);
}

// centre id -> active-list id (ivf.py:91: active_centers = all_centers[np.unique(nearest)])
__global__ void remap_keys_kernel(int *__restrict__ keys, int64_t n, const int *__restrict__ remap)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = remap[keys[i]];
}

void tk_launch_remap_keys(int *keys, int64_t n, const int *remap, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(remap_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, keys, n, remap);
}

// stable sort of (list, row) pairs by list; tmp == NULL: size query.  Offline build only (tk_index_build_dev).
int tk_sort_pairs(void *tmp, size_t *tmp_bytes, const int *keys_in, int *keys_out, const int *vals_in,
                  int *vals_out, int64_t n, int bits, hipStream_t s)
{
    size_t bytes = *tmp_bytes;
    hipError_t e = rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                             (unsigned)bits, s);
    *tmp_bytes = bytes;
    return e == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------
// Exclusive prefix sum in ONE launch (the per-batch sums of a list-sharded rank: segment positions, pair offsets).
// Single pass with decoupled look-back: workgroups take tiles in ticket order (so every predecessor of a tile is
// already running: the waits below always end), a tile publishes its AGGREGATE as soon as it is reduced and its
// inclusive PREFIX once it knows what lies in front; a tile finds that by walking back 64 tiles at a time with
// wave 0 — the first tile with a prefix ends the walk, the aggregates on the way are added.  State of a tile =
// one 64-bit word (flag in the top two bits, value below: a single relaxed atomic publishes both).
//   tmp layout: [0] ticket (unsigned), [8 ...) one uint64 per tile; zeroed by a memset in front of the kernel.
#define TK_SCAN_THREADS 256
#define TK_SCAN_ITEMS 8
#define TK_SCAN_TILE (TK_SCAN_THREADS * TK_SCAN_ITEMS)
#define TK_SCAN_AGG (1ull << 62)
#define TK_SCAN_PFX (2ull << 62)
#define TK_SCAN_VAL ((1ull << 62) - 1ull)

template <typename T>
__global__ __launch_bounds__(TK_SCAN_THREADS) void scan_exclusive_kernel(const T *__restrict__ in, T *__restrict__ out,
                                                                         int64_t n, unsigned *ticket,
                                                                         unsigned long long *state)
{
    __shared__ unsigned tile_s;
    __shared__ long long wave_sum[TK_SCAN_THREADS / 64];
    __shared__ long long excl_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) tile_s = atomicAdd(ticket, 1u);
    __syncthreads();
    const int64_t tile = tile_s;
    const int64_t base = tile * TK_SCAN_TILE + (int64_t)tid * TK_SCAN_ITEMS;
    long long v[TK_SCAN_ITEMS];
    long long mine = 0;
#pragma unroll
    for (int i = 0; i < TK_SCAN_ITEMS; i++) {
        v[i] = base + i < n ? (long long)in[base + i] : 0;
        mine += v[i];
    }
    // inclusive scan of the threads' sums inside the wave
    long long inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const long long up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    long long wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < TK_SCAN_THREADS / 64; w++) {
        wave_off += w < wave ? wave_sum[w] : 0;
        total += wave_sum[w];
    }
    if (wave == 0) {
        long long excl = 0;
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(&state[0], TK_SCAN_PFX | ((unsigned long long)total & TK_SCAN_VAL),
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&state[tile], TK_SCAN_AGG | ((unsigned long long)total & TK_SCAN_VAL),
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int64_t j = tile - 1;; j -= 64) {
                const int64_t idx = j - lane;
                unsigned long long st = TK_SCAN_PFX;                    // (in front of tile 0: a prefix of zero)
                if (idx >= 0) {
                    do {
                        st = __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } while ((st >> 62) == 0);
                }
                const unsigned long long has_pfx = __ballot((st >> 62) == 2);
                const int first = has_pfx ? __builtin_ctzll(has_pfx) : 64;
                // sign-extend the 62-bit value (sums of non-negative counts here, but keep the arithmetic exact)
                long long val = (long long)((st & TK_SCAN_VAL) << 2) >> 2;
                val = lane <= first ? val : 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) val += __shfl_xor(val, o, 64);
                excl += val;
                if (has_pfx) break;
            }
            if (lane == 0) __hip_atomic_store(&state[tile],
                                              TK_SCAN_PFX | ((unsigned long long)(excl + total) & TK_SCAN_VAL),
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) excl_s = excl;
    }
    __syncthreads();
    long long run = excl_s + wave_off + (inc - mine);
#pragma unroll
    for (int i = 0; i < TK_SCAN_ITEMS; i++) {
        if (base + i < n) out[base + i] = (T)run;
        run += v[i];
    }
}

static size_t scan_tmp_bytes(int64_t n)
{
    const int64_t tiles = (n + TK_SCAN_TILE - 1) / TK_SCAN_TILE;
    return 8 + (size_t)(tiles > 0 ? tiles : 1) * 8;
}

template <typename T>
static int scan_exclusive_impl(void *tmp, size_t *tmp_bytes, const T *in, T *out, int64_t n, hipStream_t s)
{
    const size_t need = scan_tmp_bytes(n);
    if (!tmp) {
        *tmp_bytes = need;
        return 0;
    }
    if (*tmp_bytes < need) return -1;
    if (n <= 0) return 0;
    if (hipMemsetAsync(tmp, 0, need, s) != hipSuccess) return -1;
    const int64_t tiles = (n + TK_SCAN_TILE - 1) / TK_SCAN_TILE;
    hipLaunchKernelGGL(scan_exclusive_kernel<T>, dim3((unsigned)tiles), dim3(TK_SCAN_THREADS), 0, s, in, out, n,
                       (unsigned *)tmp, (unsigned long long *)((char *)tmp + 8));
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// exclusive prefix sum of n ints; tmp == NULL: size query.  in == out is allowed (a thread reads its items before
// any thread of the grid writes them: tiles do not overlap).
int tk_scan_exclusive(void *tmp, size_t *tmp_bytes, const int *in, int *out, int64_t n, hipStream_t s)
{
    return scan_exclusive_impl<int>(tmp, tmp_bytes, in, out, n, s);
}

int tk_scan_exclusive64(void *tmp, size_t *tmp_bytes, const long long *in, long long *out, int64_t n,
                        hipStream_t s)
{
    return scan_exclusive_impl<long long>(tmp, tmp_bytes, in, out, n, s);
}

__global__ void widen_ids_kernel(const int *__restrict__ rows, int64_t n, int64_t *__restrict__ ids)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ids[i] = rows[i];
}

void tk_launch_widen_ids(const int *rows, int64_t n, int64_t *ids, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(widen_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rows, n, ids);
}

// ---------------------------------------------------------------------------
// labels (rows, M) uint8 -> tiled Quick-ADC chunks of all lists.  One thread per (global chunk
// c, block pair p): byte r of its 16-byte group = code[row_r][2p] | code[row_r][2p+1] << 4,
// row_r = the r-th row of the chunk in list order (rows_sorted; NULL: rows are already in list
// order); rows past the list's end carry the zero vector's code (pad2 + transform,
// fast_pq.py:165).
__global__ void pack_lists_kernel(const uint8_t *__restrict__ labels, int M,
                                  const int *__restrict__ rows_sorted,
                                  const int64_t *__restrict__ ids_off,
                                  const int64_t *__restrict__ chunk_off,
                                  const int64_t *__restrict__ list_n, int n_lists,
                                  const uint8_t *__restrict__ zero_code, uint4 *__restrict__ tiled,
                                  int64_t total_chunks)
{
    const int P = M >> 1;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_chunks * P) return;
    const int64_t c = i / P;
    const int p = (int)(i - c * P);
    int lo = 0, hi = n_lists;   // chunk_off[lo] <= c < chunk_off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (chunk_off[mid] <= c) lo = mid; else hi = mid;
    }
    const int64_t j = c - chunk_off[lo];
    const int64_t base = ids_off[lo] + 16 * j;
    const int64_t left = list_n[lo] - 16 * j;
    const uint32_t zb = (uint32_t)zero_code[2 * p] | ((uint32_t)zero_code[2 * p + 1] << 4);
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 16; r++) {
        uint32_t b = zb;
        if (r < left) {
            const int64_t row = rows_sorted ? (int64_t)rows_sorted[base + r] : base + r;
            const uint8_t *lab = labels + row * M + 2 * p;
            b = (uint32_t)lab[0] | ((uint32_t)lab[1] << 4);
        }
        w[r >> 2] |= b << (8 * (r & 3));
    }
    tiled[((c >> 3) * P + p) * 8 + (c & 7)] = make_uint4(w[0], w[1], w[2], w[3]);
}

void tk_launch_pack_lists(const uint8_t *labels, int M, const int *rows_sorted, const int64_t *ids_off,
                          const int64_t *chunk_off, const int64_t *list_n, int n_lists,
                          const uint8_t *zero_code, uint4 *tiled, int64_t total_chunks, hipStream_t s)
{
    const int64_t items = total_chunks * (M / 2);
    if (items <= 0) return;
    hipLaunchKernelGGL(pack_lists_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, labels, M,
                       rows_sorted, ids_off, chunk_off, list_n, n_lists, zero_code, tiled, total_chunks);
}

// rows of a float32 (N, d) matrix gathered by id (rescoring vectors for the checker)
__global__ void gather_rows_kernel(const float *__restrict__ X, int d, const int64_t *__restrict__ rows,
                                   int64_t n, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * d) return;
    const int64_t r = i / d;
    out[i] = X[rows[r] * d + (i - r * d)];
}

void tk_launch_gather_rows(const float *X, int d, const int64_t *rows, int64_t n, float *out, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n * d + 255) / 256)), dim3(256), 0, s, X, d, rows,
                       n, out);
}

// chunks of the lists a rank owns, copied from the whole index's tiled code array into the
// rank's own (compact) tiled array: local chunk cl of list l = global chunk coff[l] + cl - loff[l]
__global__ void compact_tiled_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int P,
                                     const int64_t *__restrict__ global_off,
                                     const int64_t *__restrict__ local_off, int n_lists,
                                     int64_t local_chunks)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= local_chunks * P) return;
    const int64_t cl = i / P;
    const int p = (int)(i - cl * P);
    int lo = 0, hi = n_lists;   // local_off[lo] <= cl < local_off[hi]; empty (foreign) lists are skipped
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (local_off[mid] <= cl) lo = mid; else hi = mid;
    }
    const int64_t cg = global_off[lo] + (cl - local_off[lo]);
    dst[((cl >> 3) * P + p) * 8 + (cl & 7)] = src[((cg >> 3) * P + p) * 8 + (cg & 7)];
}

void tk_launch_compact_tiled(const uint4 *src, uint4 *dst, int P, const int64_t *global_off,
                             const int64_t *local_off, int n_lists, int64_t local_chunks, hipStream_t s)
{
    const int64_t items = local_chunks * P;
    if (items <= 0) return;
    hipLaunchKernelGGL(compact_tiled_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, src, dst,
                       P, global_off, local_off, n_lists, local_chunks);
}

// ---------------------------------------------------------------------------
// measurement plumbing: a kernel that only READS `n` uint4 (every byte once, 16 B per lane and
// load, a wave 1 KiB contiguous — the access pattern of the flat scan), for the streaming-read
// ceiling bench.py prints beside the HBM-scale scan leg
__global__ __launch_bounds__(256) void read_only_kernel(const uint4 *__restrict__ src, int64_t n,
                                                        uint32_t *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;      // never true for the test pattern: keeps the loads
}

void tk_launch_read_only(const void *src, int64_t n_uint4, uint32_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(read_only_kernel, dim3(65536), dim3(256), 0, s, (const uint4 *)src, n_uint4, out);
}

// ... and one that reads `n_gather` ROWS of `row16` 16-byte pieces each, chosen at random among n_rows (a hash
// of the gather index), the way the rescoring kernel reads its candidates: consecutive lanes = consecutive
// pieces of a row, a wave's load instruction covers 64 / lanes_per_row whole rows, 16 loads in flight per lane.
// The ceiling of `roofline.rescore` (MI355X_MICROARCH.md quotes 5.5-5.6 TB/s for 1 152-byte rows; GloVe's are 400).
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4 *__restrict__ src, int64_t n_rows, int row16,
                                                          int lpr_log, int64_t n_gather, uint32_t *__restrict__ out)
{
    const int lpr = 1 << lpr_log;
    const int64_t lane_rows = ((int64_t)gridDim.x * blockDim.x) >> lpr_log;       // rows in flight per load round
    const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> lpr_log;
    const int pc = threadIdx.x & (lpr - 1);
    uint32_t acc = 0;
    for (int64_t g = g0; g < n_gather; g += 16 * lane_rows) {
        uint4 v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const uint64_t gi = (uint64_t)(g + u * lane_rows);
            uint64_t h = gi * 0x9E3779B97F4A7C15ull;
            h ^= h >> 29;
            h *= 0xBF58476D1CE4E5B9ull;
            h ^= h >> 32;
            const int64_t row = (int64_t)(h % (uint64_t)n_rows);
            v[u] = (pc < row16 && gi < (uint64_t)n_gather) ? src[row * row16 + pc] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 16; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

void tk_launch_gather_rows(const void *src, int64_t n_rows, int row_bytes, int64_t n_gather, uint32_t *out, hipStream_t s)
{
    const int row16 = row_bytes / 16;
    const int lpr_log = row16 <= 16 ? 4 : (row16 <= 32 ? 5 : 6);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(8192), dim3(256), 0, s, (const uint4 *)src, n_rows, row16, lpr_log,
                       n_gather, out);
}
