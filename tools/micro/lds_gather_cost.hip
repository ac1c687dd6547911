// Cost of a 64-lane 16-byte (and 12-byte) gather from LDS on gfx950 as a function of how the lanes' indices
// are spread over a window of W records -- the LDS counterpart of gather_cost.hip.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_gather_cost.hip -o ab/lds_gather_cost && ab/lds_gather_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>   // 0: float4 records (ds_read_b128), 1: SoA x[], y[], z[] (3 x ds_read_b32), 2: float4, lanes consecutive
__global__ __launch_bounds__(256) void k_lds(const uint32_t* __restrict__ idx_tab, uint32_t window, int iters, float* out) {
    __shared__ float4 rec[2048];
    __shared__ float sx[2048], sy[2048], sz[2048];
    for (uint32_t t = threadIdx.x; t < 2048; t += 256) {
        rec[t] = make_float4((float)t, 1.f, 2.f, 3.f);
        sx[t] = (float)t; sy[t] = 1.f; sz[t] = 2.f;
    }
    __syncthreads();
    float acc = 0.f;
    uint32_t k = blockIdx.x * 256u + threadIdx.x;
    for (int it = 0; it < iters; it += 4) {
        uint32_t j[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            j[u] = MODE == 2 ? ((threadIdx.x & 63u) + (uint32_t)(it + u) * 3u) & (window - 1u) : idx_tab[(k + (uint32_t)(it + u) * 64u) & 65535u] & (window - 1u);
        }
        if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += sx[j[u]] + sy[j[u]] + sz[j[u]];
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) { const float4 v = rec[j[u]]; acc += v.x + v.y + v.z + v.w; }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int MODE>
static void run(const char* name, const uint32_t* d_idx, float* out) {
    for (uint32_t window : {64u, 256u, 1024u, 2048u}) {
        const int iters = 512, blocks = 256 * 8 * 2;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        k_lds<MODE><<<blocks, 256>>>(d_idx, window, iters, out);
        hipEventRecord(a);
        k_lds<MODE><<<blocks, 256>>>(d_idx, window, iters, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double instr = (double)blocks * 4 * iters;       // wave-level gathers
        printf("%-34s window %4u records: %.3f ms, %.1f clk per wave-gather per CU (incl. the index load)\n", name, window, ms,
               ms * 1e-3 * 2.4e9 / (instr / 256.0));
    }
}

int main() {
    uint32_t* h = (uint32_t*)malloc(65536 * 4);
    uint32_t s = 12345u;
    for (int i = 0; i < 65536; ++i) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; h[i] = s; }
    uint32_t* d; hipMalloc(&d, 65536 * 4); hipMemcpy(d, h, 65536 * 4, hipMemcpyHostToDevice);
    float* out; hipMalloc(&out, 4);
    run<0>("random float4 (ds_read_b128)", d, out);
    run<1>("random SoA (3 x ds_read_b32)", d, out);
    run<2>("consecutive float4 (ds_read_b128)", d, out);
    return 0;
}
