// Streaming READ bandwidth of HBM on this part, the way the flat scan reads its codes: every lane
// 16 bytes per load, a wave 1 KiB contiguous, each byte once.  Prints GB/s for a few grid shapes.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/bin/read_bw scripts/micro/read_bw.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const uint4 *__restrict__ src, int64_t n, uint32_t *out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n; i += stride) { uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    const int64_t bytes = (int64_t)1 << 30, n = bytes / 16;
    uint4 *buf; uint32_t *out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4);
    hipMemset(buf, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[] = {2048, 8192, 16384, 65536};
    for (int g : grids)
        for (int un = 1; un <= 4; un *= 2) {
            auto launch = [&]() {
                if (un == 1) hipLaunchKernelGGL(read_kernel<1>, dim3(g), dim3(256), 0, 0, buf, n, out);
                else if (un == 2) hipLaunchKernelGGL(read_kernel<2>, dim3(g), dim3(256), 0, 0, buf, n, out);
                else hipLaunchKernelGGL(read_kernel<4>, dim3(g), dim3(256), 0, 0, buf, n, out);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 20; r++) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("1 GiB read, grid %6d x 256, %d loads in flight per lane: %.3f ms = %.0f GB/s\n", g, un, ms / 20,
                   bytes / (ms / 20 * 1e-3) / 1e9);
        }
    return 0;
}
