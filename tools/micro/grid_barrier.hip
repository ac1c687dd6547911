// What a phase boundary costs on gfx950 for a scene that fits the chip at once (DESIGN.md section 6,
// small scenes): (a) dependent kernel launches on one stream, (b) grid-wide barriers inside ONE
// cooperative launch (device-scope release / acquire around an atomic counter), each phase touching
// `kb` KiB per workgroup so that the release has dirty lines to write back.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier.hip -o ab/grid_barrier && ab/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned nblocks, unsigned& epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        ++epoch;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * nblocks;
        // every workgroup of the grid is resident (cooperative launch) and reaches this point: the loop ends
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__device__ __forceinline__ void phase_work(float4* buf, unsigned words_per_block, unsigned phase) {
    // read the block "to the right"'s slice written in the previous phase, write the own slice
    const unsigned nb = gridDim.x, b = blockIdx.x, src = (b + 1) % nb;
    for (unsigned w = threadIdx.x; w < words_per_block; w += blockDim.x) {
        float4 v = buf[(size_t)src * words_per_block + w + (size_t)(phase & 1) * nb * words_per_block];
        v.x += 1.f;
        buf[(size_t)b * words_per_block + w + (size_t)((phase + 1) & 1) * nb * words_per_block] = v;
    }
}

__global__ __launch_bounds__(256) void k_coop(float4* buf, unsigned words_per_block, int phases, unsigned* ctr) {
    unsigned epoch = 0;
    for (int p = 0; p < phases; ++p) {
        phase_work(buf, words_per_block, (unsigned)p);
        grid_barrier(ctr, gridDim.x, epoch);
    }
}

__global__ __launch_bounds__(256) void k_phase(float4* buf, unsigned words_per_block, unsigned p) {
    phase_work(buf, words_per_block, p);
}

int main() {
    const int phases = 14, reps = 50;
    unsigned* ctr; CK(hipMalloc(&ctr, 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (unsigned blocks : {64u, 250u, 256u, 512u}) {
        for (unsigned kb : {0u, 4u, 16u, 64u}) {
            const unsigned wpb = kb * 1024u / 16u;
            float4* buf; CK(hipMalloc(&buf, (size_t)2 * blocks * (wpb ? wpb : 1) * 16)); CK(hipMemset(buf, 0, (size_t)2 * blocks * (wpb ? wpb : 1) * 16));
            // (a) dependent launches
            for (int w = 0; w < 2; ++w) {
                CK(hipEventRecord(a, s));
                for (int r = 0; r < reps; ++r)
                    for (int p = 0; p < phases; ++p) k_phase<<<blocks, 256, 0, s>>>(buf, wpb, (unsigned)p);
                CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            }
            float ms_l; CK(hipEventElapsedTime(&ms_l, a, b));
            // (b) one cooperative launch per "sub-step"
            float ms_c = -1.f;
            {
                int ph = phases; unsigned w_ = wpb;
                void* args[] = {&buf, &w_, &ph, &ctr};
                bool ok = true;
                for (int w = 0; w < 2 && ok; ++w) {
                    CK(hipEventRecord(a, s));
                    for (int r = 0; r < reps; ++r) {
                        CK(hipMemsetAsync(ctr, 0, 4, s));
                        hipError_t e = hipLaunchCooperativeKernel((const void*)k_coop, dim3(blocks), dim3(256), args, 0, s);
                        if (e != hipSuccess) { printf("  cooperative launch of %u blocks refused: %s\n", blocks, hipGetErrorString(e)); ok = false; break; }
                    }
                    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
                }
                if (ok) CK(hipEventElapsedTime(&ms_c, a, b));
            }
            printf("blocks %4u  %3u KiB/block/phase: %d dependent launches %.2f us per phase | cooperative %.2f us per phase (barrier incl.)\n",
                   blocks, kb, phases, ms_l * 1e3 / (reps * phases), ms_c < 0 ? -1.0 : ms_c * 1e3 / (reps * phases));
            CK(hipFree(buf));
        }
    }
    return 0;
}
