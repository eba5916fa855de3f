// tickets.h — work distribution of the persistent scan kernels (adc_scan.hip, plain_scan.hip):
// every wave takes one block of work statically, the rest are drawn from a few work counters.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// work counters of the list-major scan: TK_TICKETS ints, 128 bytes apart, behind the
// (n_lists+1)-entry unit_prefix table in the same allocation
#define TK_TICKETS 8
#define TK_TICKET_OFF(n_lists) ((((n_lists) + 1 + 31) / 32 + 1) * 32)

// Blocks of 64 consecutive units.  Every wave takes one block statically; the rest are
// drawn from TK_TICKETS work counters (behind the prefix table, a cache line each, zeroed
// by the kernel that wrote the table; a wave starts at its home counter and moves on when a
// range is used up), the draw for the NEXT block being issued before the current block's
// work.  With other batches' heap replays sharing some SIMDs a static split leaves the
// kernel waiting for its slowest waves.  One counter would not do: same-address atomics
// retire at ~60 M/s and this kernel wants 70 M blocks/s.
// `work(blk)` processes block blk of NB.
template <typename F>
__device__ __forceinline__ void ticketed_blocks(int NB, int *ticket, F work)
{
    const int NW = gridDim.x * 4 < NB ? gridDim.x * 4 : NB;   // blocks handed out statically
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int dyn = NB - NW;
    int tried = 0, tk = wid & (TK_TICKETS - 1);
    // synchronous draw: the search through the other ranges once a wave's range is used up
    auto draw = [&]() -> int {
        while (tried < TK_TICKETS) {
            const int lo = (int)((int64_t)dyn * tk / TK_TICKETS);
            const int len = (int)((int64_t)dyn * (tk + 1) / TK_TICKETS) - lo;
            int got = len;
            if ((threadIdx.x & 63) == 0) {
                int *t = ticket + tk * 32;
                if (__atomic_load_n(t, __ATOMIC_RELAXED) < len) got = atomicAdd(t, 1);
            }
            got = __builtin_amdgcn_readfirstlane(got);
            if (got < len) return NW + lo + got;
            tried++;
            tk = (tk + 1) & (TK_TICKETS - 1);
        }
        return NB;
    };
    int blk = wid < NW ? wid : NB;
    while (blk < NB) {
        // draw the block after this one now; the answer is looked at after this block's work
        int nlo = 0, nlen = 0, ngot = 0;
        const bool drawn = dyn > 0 && tried < TK_TICKETS;
        if (drawn) {
            nlo = (int)((int64_t)dyn * tk / TK_TICKETS);
            nlen = (int)((int64_t)dyn * (tk + 1) / TK_TICKETS) - nlo;
            if ((threadIdx.x & 63) == 0) ngot = atomicAdd(ticket + tk * 32, 1);
        }
        work(__builtin_amdgcn_readfirstlane(blk));   // wave-uniform, and known to be
        // next block
        blk = NB;
        if (!drawn) break;
        ngot = __builtin_amdgcn_readfirstlane(ngot);
        if (ngot < nlen) {
            blk = NW + nlo + ngot;
            continue;
        }
        tried++;
        tk = (tk + 1) & (TK_TICKETS - 1);
        blk = draw();
    }
}
