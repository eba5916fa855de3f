// mfma_scan.hip — round-3 microbenchmark (go/no-go for the matrix-core scan of the probed lists
// behind the first one): plain int32 sums of table entries as a one-hot(code) x
// table int8 contraction on v_mfma_i32_32x32x32_i8, saturated to int8 on the way out.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bin/mfma_scan mfma_scan.hip && bin/mfma_scan
//
// One wave = one unit (list, tile of 32 (query, list) pairs, range of chunk pairs):
//   B operand: lane (q = lane & 31, h = lane >> 5) holds the 16-byte table row of block 2p + h of
//              its pair's query for every block pair p — 26 x 4 VGPRs, loaded once per unit;
//   A operand: rows = the 32 rows of two consecutive 16-row chunks; lane (r, h) turns nibble h of
//              byte r of the chunk's 16-byte group p into a 16-byte one-hot through a 256-byte
//              LDS table (one ds_read_b128: distinct entries sit on distinct banks);
//   D: 32 rows x 32 queries of int32 sums; lane (q, h) holds rows {0-3, 8-11, 16-19, 24-27} + 4h,
//      clamps them to int8, and after two v_permlane32_swap holds the whole 16-byte block of
//      (chunk 2cp + h, query q): one 16-byte store + one minimum byte per lane.
// Checked against a host restatement on every output byte.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

// the kernel under test is the library's own (one source of truth)
#include "../../tinyknn_amd/csrc/plain_scan.hip"

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

static double now_ms(hipEvent_t a, hipEvent_t b)
{
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main(int argc, char **argv)
{
    const int P = 26, M = 52;
    int n_lists = argc > 1 ? atoi(argv[1]) : 1087;
    int C = argc > 2 ? atoi(argv[2]) : 69;           // chunks per list
    int nq = argc > 3 ? atoi(argv[3]) : 10000;
    int S = argc > 4 ? atoi(argv[4]) : 9;            // lists probed per query
    int cpu = argc > 5 ? atoi(argv[5]) : 512;        // persistent workgroups
    srand(1);
    // codes: random bytes, tiled layout; every list C chunks (+/- a few)
    std::vector<int64_t> coff(n_lists + 1, 0);
    for (int l = 0; l < n_lists; l++) coff[l + 1] = coff[l] + std::max(1, C + (rand() % 7) - 3);
    const int64_t total = coff[n_lists];
    const int64_t tiled = (total + 7) / 8 * 8 * P;
    std::vector<uint8_t> codes((size_t)tiled * 16);
    for (auto &b : codes) b = (uint8_t)(rand() & 0xff);
    std::vector<int8_t> tables((size_t)nq * M * 16);
    for (auto &b : tables) b = (int8_t)((rand() % 28) - 4);
    for (int q = 0; q < nq; q += 97)                 // some queries with large entries: clamps
        for (int i = 0; i < M * 16; i++) tables[(size_t)q * M * 16 + i] = (int8_t)((rand() % 256) - 128);
    // probes: S distinct lists per query; rows laid out back to back
    std::vector<std::vector<int>> by_list(n_lists);
    std::vector<int> q_f0((size_t)nq * S);
    int64_t cap = 0;
    for (int q = 0; q < nq; q++) {
        int f = 0;
        for (int s = 0; s < S; s++) {
            int l;
            bool dup;
            do {
                l = rand() % n_lists;
                dup = false;
                for (int s2 = 0; s2 < s; s2++) dup |= q_f0[(size_t)q * S + s2] == l;
            } while (dup);
            q_f0[(size_t)q * S + s] = l;
            by_list[l].push_back(q * S + s);
            (void)f;
        }
    }
    std::vector<int> rowpos((size_t)nq * S);
    for (int q = 0; q < nq; q++) {
        int f = 0;
        for (int s = 0; s < S; s++) {
            const int l = q_f0[(size_t)q * S + s];
            rowpos[(size_t)q * S + s] = f;
            f += (int)(coff[l + 1] - coff[l]);
        }
        cap = std::max<int64_t>(cap, f);
    }
    if (argc > 6) cap = std::max<int64_t>(cap, atoll(argv[6]));      // row stride override (chunks)
    const int64_t min_stride = (cap + 15) / 16 * 16;
    std::vector<int> pair_off(n_lists + 1, 0), unit_prefix(tk_unit_prefix_ints(n_lists), 0), pair_q, pair_f0;
    for (int l = 0; l < n_lists; l++) {
        for (int id : by_list[l]) {
            pair_q.push_back(id / S);
            pair_f0.push_back(rowpos[id]);
        }
        pair_off[l + 1] = (int)pair_q.size();
        const int cnt = (int)by_list[l].size();
        const int CP = (int)((coff[l + 1] - coff[l] + 1) / 2);
        unit_prefix[l + 1] = unit_prefix[l] + ((cnt + 31) / 32); (void)CP;
    }
    const int U = unit_prefix[n_lists];
    std::vector<int> unit_desc;
    for (int l = 0; l < n_lists; l++)
        for (int t = 0; t < unit_prefix[l + 1] - unit_prefix[l]; t++) { unit_desc.push_back(l); unit_desc.push_back(t); }
    // one wave per unit (round 4): (list, tile, chunk pairs [a, b)), K chunk pairs per unit
    const int K = argc > 7 ? atoi(argv[7]) : 12;
    const int variant = argc > 8 ? atoi(argv[8]) : 1;      // 0: workgroup per tile (round 3), 1: wave per unit
    std::vector<int> unit_prefix4(tk_unit_prefix_ints(n_lists), 0), unit_desc4;
    for (int l = 0; l < n_lists; l++) {
        const int CP = (int)((coff[l + 1] - coff[l] + 1) / 2);
        const int tiles = unit_prefix[l + 1] - unit_prefix[l];
        const int nsub = (CP + K - 1) / K;
        for (int t = 0; t < tiles; t++)
            for (int sb = 0; sb < nsub; sb++) {
                unit_desc4.push_back(l); unit_desc4.push_back(t);
                unit_desc4.push_back(sb * K); unit_desc4.push_back(std::min(CP, (sb + 1) * K));
            }
        unit_prefix4[l + 1] = (int)(unit_desc4.size() / 4);
    }
    printf("wave form: %d units of <= %d chunk pairs\n", unit_prefix4[n_lists], K);
    double pairs_chunks = 0;
    for (int l = 0; l < n_lists; l++) pairs_chunks += (double)by_list[l].size() * (coff[l + 1] - coff[l]);
    printf("lists %d x ~%d chunks, %d queries x %d probes: %d units, %.2f M (chunk, query) pairs, cap %lld\n",
           n_lists, C, nq, S, U, pairs_chunks / 1e6, (long long)cap);

    uint4 *d_codes, *d_tables, *d_dist;
    uint8_t *d_mins;
    int64_t *d_coff;
    int *d_up, *d_po, *d_pq, *d_pf, *d_ud;
    CHECK(hipMalloc(&d_codes, codes.size()));
    CHECK(hipMalloc(&d_tables, tables.size()));
    CHECK(hipMalloc(&d_dist, (size_t)nq * cap * 16));
    CHECK(hipMalloc(&d_mins, (size_t)nq * min_stride));
    CHECK(hipMalloc(&d_coff, coff.size() * 8));
    CHECK(hipMalloc(&d_up, unit_prefix.size() * 4));
    CHECK(hipMalloc(&d_po, pair_off.size() * 4));
    CHECK(hipMalloc(&d_ud, unit_desc.size() * 4 + 8));
    CHECK(hipMemcpy(d_ud, unit_desc.data(), unit_desc.size() * 4, hipMemcpyHostToDevice));
    int *d_ud4, *d_up4;
    CHECK(hipMalloc(&d_ud4, unit_desc4.size() * 4 + 16));
    CHECK(hipMemcpy(d_ud4, unit_desc4.data(), unit_desc4.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_up4, unit_prefix4.size() * 4));
    CHECK(hipMemcpy(d_up4, unit_prefix4.data(), unit_prefix4.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_pq, pair_q.size() * 4 + 4));
    CHECK(hipMalloc(&d_pf, pair_f0.size() * 4 + 4));
    CHECK(hipMemcpy(d_codes, codes.data(), codes.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_tables, tables.data(), tables.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_coff, coff.data(), coff.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_up, unit_prefix.data(), unit_prefix.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_po, pair_off.data(), pair_off.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_pq, pair_q.data(), pair_q.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_pf, pair_f0.data(), pair_f0.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_dist, 0xee, (size_t)nq * cap * 16));
    CHECK(hipMemset(d_mins, 0xee, (size_t)nq * min_stride));
    TkScanJob j;
    j.codes = d_codes; j.tables = d_tables; j.list_chunk_off = d_coff; j.n_lists = n_lists;
    j.unit_prefix = d_up; j.pair_off = d_po; j.pair_q = d_pq; j.pair_f0 = d_pf;
    j.dist = d_dist; j.cap = cap; j.mins = d_mins; j.min_stride = min_stride;
    j.unit_desc4 = d_ud4;       // (list, tile, chunk pairs [a, b)): one wave per unit
    j.unit_prefix = d_up4;
    (void)variant;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 20;
    auto launch = [&]() {
        CHECK(hipMemsetAsync(const_cast<int *>(j.unit_prefix) + TK_PLAIN_COUNTER_OFF(n_lists), 0, 8 * 32 * 4, 0));
        if (tk_launch_scan_plain(j, M, TK_ORDER_AVX, cpu, 0)) { fprintf(stderr, "launch failed\n"); exit(1); }
    };
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    const double ms = now_ms(e0, e1) / reps;
    printf("scan_plain_kernel: %.4f ms per launch = %.2f G (chunk, query)/s, %.1f GB/s algorithmic (26 B per (query, code)), "
           "%.1f cycles per (chunk, query) per SIMD at 2.4 GHz\n",
           ms, pairs_chunks / ms / 1e6, pairs_chunks * 416 / ms / 1e6, ms * 1e-3 * 2.4e9 * 1024 / pairs_chunks);

#ifdef TK_PLAIN_CLOCK
    {
        unsigned long long z[4] = {0, 0, 0, 0}, ck[4];
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(tk_plain_clock), z, sizeof z));
        for (int i = 0; i < 5; i++) launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpyFromSymbol(ck, HIP_SYMBOL(tk_plain_clock), sizeof ck));
        printf("in-kernel clock: %.3f GHz (s_memtime / s_memrealtime x 100 MHz), %.0f cycles = %.1f us per wave, %.0f waves per launch\n",
               (double)ck[0] / (double)ck[1] * 0.1, (double)ck[0] / (double)ck[2], (double)ck[1] / (double)ck[2] / 100.0, ck[2] / 5.0);
    }
#endif
#ifdef TK_PLAIN_STAMPS
    {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st[8];
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(tk_plain_stamps), z, sizeof z));
        launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpyFromSymbol(st, HIP_SYMBOL(tk_plain_stamps), sizeof st));
        const double it = (double)st[6], w = 2048.0;
        printf("stamps, cycles per chunk pair and wave (%.0f chunk pairs, %.1f per wave): staging + fetch %.0f | LDS reads + MFMA chain + clamp %.0f | "
               "pack + swaps + LDS tile %.0f | flush %.0f | unit prologue %.0f | between units %.0f || per wave: kernel %.0f\n",
               it, it / w, st[1] / it, st[2] / it, st[3] / it, st[4] / it, st[0] / it, st[5] / it, st[7] / w);
    }
#endif
    // ---- check every byte against the host
    std::vector<uint8_t> dist((size_t)nq * cap * 16), mins((size_t)nq * min_stride);
    CHECK(hipMemcpy(dist.data(), d_dist, dist.size(), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(mins.data(), d_mins, mins.size(), hipMemcpyDeviceToHost));
    long long bad = 0, checked = 0;
    for (int l = 0; l < n_lists && bad < 10; l++) {
        const int Cl = (int)(coff[l + 1] - coff[l]);
        for (size_t k = 0; k < by_list[l].size(); k += (l % 16 == 0 ? 1 : 7)) {
            const int q = pair_q[pair_off[l] + k], f0 = pair_f0[pair_off[l] + k];
            for (int c = 0; c < Cl; c++) {
                const int64_t gc = coff[l] + c;
                int mn = 127;
                for (int row = 0; row < 16; row++) {
                    int sum = 0;
                    for (int p = 0; p < P; p++) {
                        const uint8_t byte = codes[((size_t)(((gc >> 3) * P + p) * 8 + (gc & 7))) * 16 + row];
                        sum += tables[((size_t)q * M + 2 * p) * 16 + (byte & 15)];
                        sum += tables[((size_t)q * M + 2 * p + 1) * 16 + (byte >> 4)];
                    }
                    const int o = sum < -128 ? -128 : (sum > 127 ? 127 : sum);
                    mn = std::min(mn, o);
                    const int got = (int8_t)dist[((size_t)q * cap + f0 + c) * 16 + row];
                    checked++;
                    if (got != o && bad++ < 10) printf("MISMATCH list %d q %d chunk %d row %d: got %d want %d\n", l, q, c, row, got, o);
                }
                if ((int8_t)mins[(size_t)q * min_stride + f0 + c] != mn && bad++ < 10)
                    printf("MIN MISMATCH list %d q %d chunk %d: got %d want %d\n", l, q, c,
                           (int8_t)mins[(size_t)q * min_stride + f0 + c], mn);
            }
        }
    }
    printf("checked %lld bytes: %s\n", checked, bad ? "MISMATCHES" : "all identical to the host's clamp(plain sum)");
    return bad ? 1 : 0;
}
