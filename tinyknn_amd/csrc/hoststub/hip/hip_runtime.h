// hoststub/hip/hip_runtime.h — a SYNCHRONOUS CPU stand-in for the few HIP calls the host-side translation units
// make, so that they compile with g++ and run under AddressSanitizer / ThreadSanitizer (`make asan tsan`;
// GPU sanitizers are not available on the pool).  TEST INFRASTRUCTURE ONLY: never linked into libtinyknn_hip.so.
// "Device" memory is heap memory (so ASan sees every out-of-bounds access of a kernel's host replay), a kernel
// launch runs the kernel body for every (block, thread) in order on the calling thread, streams and events
// complete at once.  What it can show: heap errors, leaks and data races of the host code (thread pool, BLAS
// binding, session bookkeeping).  What it cannot: anything that depends on asynchrony of the device.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
struct tk_stub_stream;
typedef tk_stub_stream *hipStream_t;
typedef int *hipEvent_t;
enum { hipHostMallocDefault = 0, hipEventDisableTiming = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };

struct uint4 { uint32_t x, y, z, w; };
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__
#endif

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                      \
    do {                                                                                 \
        const dim3 g_ = (grid), b_ = (block);                                            \
        gridDim = g_;                                                                    \
        blockDim = b_;                                                                   \
        for (unsigned bx_ = 0; bx_ < g_.x; bx_++)                                        \
            for (unsigned tx_ = 0; tx_ < b_.x; tx_++) {                                  \
                blockIdx = dim3(bx_);                                                    \
                threadIdx = dim3(tx_);                                                   \
                kernel(__VA_ARGS__);                                                     \
            }                                                                            \
    } while (0)

static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (int *)malloc(sizeof(int)); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t e) { return e ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { return e ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "stub error"; }
