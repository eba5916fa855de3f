// valu_rate.hip — issue rate of the scan kernel's VALU instructions on gfx950.
// Question: does a wave64 v_perm_b32 / v_pk_add_i16 clamp / v_and_or_b32
// issue in 2 cycles (SIMD-32, >= 2 waves per SIMD) or 4?  Each wave runs ITER x 16 independent
// instructions of one kind between two s_memtime stamps; W waves per SIMD on every CU.
// Prints cycles per wave-instruction per SIMD = stamp delta x (1 / (ITER*16)) / W ... per W.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define ITER 2048

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(unsigned long long *out, unsigned *sink, unsigned seed)
{
    unsigned a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = seed * (i + 1) + threadIdx.x;
    unsigned b = seed ^ 0x12345678u, sel = 0x07020501u + (threadIdx.x & 3);
    unsigned msk = 0x04040404u;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (KIND == 0) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b), "v"(sel));
            if (KIND == 1) asm volatile("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            if (KIND == 2) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "s"(msk), "v"(b));
            if (KIND == 3) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            if (KIND == 4) asm volatile("v_pk_min_i16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            if (KIND == 5) asm volatile("v_lshrrev_b32 %0, 4, %1" : "=v"(a[i]) : "v"(a[i]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r ^= a[i];
    if (r == 0xdeadbeefu) sink[0] = r;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int cus)
{
    unsigned long long *out;
    unsigned *sink;
    hipMalloc(&out, sizeof(unsigned long long) * cus * 8 * 4);
    hipMalloc(&sink, 4);
    for (int W = 1; W <= 8; W *= 2) {
        // W workgroups of 256 threads per CU = W waves per SIMD (registers/LDS allow 8)
        int blocks = cus * W;
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, sink, 12345u);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, sink, 12345u);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        double med = (double)h[h.size() / 2];
        // s_memtime ticks at 100 MHz-derived "shader cycles"? the guide: tick = shader cycle
        double per_inst_wave = med / (ITER * 16.0);
        printf("%-22s W=%d waves/SIMD: median %9.0f ticks/wave, %.3f ticks per wave-instruction, "
               "%.3f ticks per instruction per SIMD; kernel %.3f ms => %.2f G wave-instr/s/SIMD-equivalent clock %.2f GHz if 1 tick=1 cycle\n",
               name, W, med, per_inst_wave, per_inst_wave / W, ms,
               (double)ITER * 16 * W / (ms * 1e6), med / (ms * 1e6));
    }
    hipFree(out); hipFree(sink);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    run<0>("v_perm_b32", cus);
    run<1>("v_pk_add_i16 clamp", cus);
    run<2>("v_and_or_b32 (sgpr)", cus);
    run<3>("v_add_u32", cus);
    run<4>("v_pk_min_i16", cus);
    run<5>("v_lshrrev_b32", cus);
    return 0;
}
