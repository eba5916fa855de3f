// smfmac_probe.hip — round 4: operand layout and rate of v_smfmac_i32_32x32x64_i8 on gfx950.
// The plain-sum scan multiplies a one-hot(code) operand by table rows; one-hot blocks of 16 are 2:4
// sparse by construction (at most one non-zero in every group of four), so the sparse matrix
// instruction could carry 64 k-values (four blocks) in the time the dense one carries 32.
// This program finds, by brute force against host models, how the instruction maps
//   A (4 VGPRs: 16 stored bytes per lane), idx (1 VGPR), B (8 VGPRs: 32 bytes per lane)
// to logical (row, k) / (k, column), and times a dependent chain of them.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bin/smfmac_probe smfmac_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void one(const v4i *A, const v8i *B, const int *idx, v16i *C)
{
    const int l = threadIdx.x;
    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    acc = __builtin_amdgcn_smfmac_i32_32x32x64_i8(A[l], B[l], acc, idx[l], 0, 0);
    C[l] = acc;
}

__global__ void chain(const v4i *A, const v8i *B, const int *idx, v16i *C, int n, long long *cyc)
{
    const int l = threadIdx.x & 63;
    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const v4i a = A[l];
    const v8i b = B[l];
    const int ix = idx[l];
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
        acc = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a, b, acc, ix, 0, 0);
        acc = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a, b, acc, ix, 0, 0);
        acc = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a, b, acc, ix, 0, 0);
        acc = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a, b, acc, ix, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    C[l] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void chain_dense(const v4i *A, const v4i *B, v16i *C, int n, long long *cyc)
{
    const int l = threadIdx.x & 63;
    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const v4i a = A[l], b = B[l];
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    C[l] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    std::vector<int8_t> A(64 * 16), B(64 * 32);
    std::vector<uint32_t> idx(64);
    std::vector<int> C(64 * 16);
    srand(3);
    for (auto &x : A) x = (int8_t)(rand() % 7 - 3);
    for (auto &x : B) x = (int8_t)(rand() % 15 - 7);
    for (auto &x : idx) x = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    v4i *dA; v8i *dB; int *dI; v16i *dC; long long *dcyc;
    CHECK(hipMalloc(&dA, 64 * 16)); CHECK(hipMalloc(&dB, 64 * 32)); CHECK(hipMalloc(&dI, 64 * 4));
    CHECK(hipMalloc(&dC, 64 * 64)); CHECK(hipMalloc(&dcyc, 8 * 2048));
    CHECK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dI, idx.data(), 64 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, dA, dB, dI, dC);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(C.data(), dC, 64 * 64, hipMemcpyDeviceToHost));
    // C layout of the 32x32 accumulators (as the dense instruction): lane l, reg i: col = l % 32,
    // row = (i / 4) * 8 + (l / 32) * 4 + i % 4
    auto gotC = [&](int m, int n) { const int i = (m / 8) * 4 + m % 4, l = n + 32 * ((m / 4) & 1); return C[l * 16 + i]; };
    int found = 0;
    for (int am = 0; am < 2; am++)          // A: k-half by lane / 32 (0) or interleaved by 16 (1)
        for (int bm = 0; bm < 3; bm++)      // B: k = kh * 32 + t (0) | k = (t / 16) * 32 + kh * 16 + t % 16 (1) | k = (t/8)*16 + kh*8 + t%8 (2)
            for (int im = 0; im < 2; im++) {    // idx: bits 4g + 2e (0) | bits 2 * j (same) ; im=1: e-major
                long long bad = 0;
                for (int m = 0; m < 32; m++)
                    for (int n = 0; n < 32; n++) {
                        int sum = 0;
                        for (int kh = 0; kh < 2; kh++) {
                            const int la = m + 32 * kh;
                            for (int j = 0; j < 16; j++) {
                                const int g = j / 2, e = j % 2;
                                const int sel = im == 0 ? (idx[la] >> (4 * g + 2 * e)) & 3 : (idx[la] >> (16 * e + 2 * g)) & 3;
                                int k;      // logical k of stored byte j of lane la
                                if (am == 0) k = kh * 32 + 4 * g + sel;
                                else k = (g / 4) * 32 + kh * 16 + 4 * (g % 4) + sel;
                                // B element (k, n)
                                int lb, t;
                                if (bm == 0) { lb = n + 32 * (k / 32); t = k % 32; }
                                else if (bm == 1) { lb = n + 32 * ((k / 16) & 1); t = (k / 32) * 16 + k % 16; }
                                else { lb = n + 32 * ((k / 8) & 1); t = (k / 16) * 8 + k % 8; }
                                sum += (int)A[la * 16 + j] * (int)B[lb * 32 + t];
                            }
                        }
                        bad += sum != gotC(m, n);
                    }
                printf("model A%d B%d I%d: %lld of 1024 outputs differ\n", am, bm, im, bad);
                found += bad == 0;
            }
    printf("%s\n", found ? "LAYOUT FOUND" : "no model matches");
    // rate: dependent chains, 4 waves per workgroup, 2 workgroups per CU
    const int n = 4096;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(chain, dim3(512), dim3(256), 0, 0, dA, dB, dI, dC, n, dcyc);
        CHECK(hipDeviceSynchronize());
    }
    long long cy[512];
    CHECK(hipMemcpy(cy, dcyc, sizeof cy, hipMemcpyDeviceToHost));
    printf("sparse 32x32x64 i8: %.1f cycles per instruction per wave (2 waves per SIMD share the pipe)\n", (double)cy[0] / (4.0 * n));
    hipLaunchKernelGGL(chain_dense, dim3(512), dim3(256), 0, 0, dA, (const v4i *)dB, dC, n, dcyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(cy, dcyc, sizeof cy, hipMemcpyDeviceToHost));
    printf("dense 32x32x32 i8: %.1f cycles per instruction per wave\n", (double)cy[0] / (4.0 * n));
    hipLaunchKernelGGL(chain, dim3(256), dim3(256), 0, 0, dA, dB, dI, dC, n, dcyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(cy, dcyc, sizeof cy, hipMemcpyDeviceToHost));
    printf("sparse, one wave per SIMD: %.1f cycles per instruction\n", (double)cy[0] / (4.0 * n));
    hipLaunchKernelGGL(chain_dense, dim3(256), dim3(256), 0, 0, dA, (const v4i *)dB, dC, n, dcyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(cy, dcyc, sizeof cy, hipMemcpyDeviceToHost));
    printf("dense, one wave per SIMD: %.1f cycles per instruction\n", (double)cy[0] / (4.0 * n));
    return 0;
}
