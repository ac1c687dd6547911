// Cost model of a 64-lane 16-byte gather on gfx950 as a function of how many 128-byte lines the
// lanes touch (the sweeps' second limiter).  Every lane reads ITERS float4 at
// (base + lane * stride_q / 4 + it * 64) mod table; the table (tab_kb KiB) is L1- or L2-resident.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gather_cost.hip -o gpurun_out/gather_cost && gpurun_out/gather_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ tab, uint32_t mask, uint32_t stride_q,
                                                int iters, float* out) {
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    uint32_t idx = wave * 977u + ((lane * stride_q) >> 2);
    float acc = 0.f;
    for (int it = 0; it < iters; it += 4) {
        float4 a = tab[(idx) & mask], b = tab[(idx + 64u * 7u) & mask], c = tab[(idx + 128u * 7u) & mask],
               d = tab[(idx + 192u * 7u) & mask];
        acc += a.x + b.y + c.z + d.w;
        idx += 256u * 7u + 3u;
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const uint32_t sizes_kb[] = {16, 2048, 262144};
    const uint32_t strides[] = {0, 4, 8, 16, 32, 64, 128};   // lane stride in quarter-elements: 4 = consecutive float4
    float* out; hipMalloc(&out, 4);
    for (uint32_t kb : sizes_kb) {
        const uint32_t n = kb * 1024u / 16u;
        float4* tab; hipMalloc(&tab, (size_t)n * 16); hipMemset(tab, 0, (size_t)n * 16);
        for (uint32_t st : strides) {
            const int iters = 256, blocks = 256 * 8 * 4;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            k_gather<<<blocks, 256>>>(tab, n - 1, st, iters, out);
            hipEventRecord(a);
            k_gather<<<blocks, 256>>>(tab, n - 1, st, iters, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double instr = (double)blocks * 4 * iters;          // wave-level gather instructions
            const double per_cu_clk = ms * 1e-3 * 2.4e9 / (instr / 256.0);
            const double lines = st == 0 ? 1 : (st * 64.0 * 4 / 128.0 < 1 ? 1 : st * 64.0 * 4.0 / 128.0);
            printf("table %7u KiB  lane stride %5.1f float4 (~%4.0f lines/gather): %.3f ms  %.1f clk per gather per CU\n",
                   kb, st / 4.0, lines > 64 ? 64.0 : lines, ms, per_cu_clk);
        }
        hipFree(tab);
    }
    return 0;
}
