// Issue rate of the VALU instruction kinds the sweeps are made of, on gfx950: clocks per wave64 instruction per SIMD
// for v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_fma_f64, v_mul_f64, v_cvt_f64_f32, v_cvt_f32_f64, v_rsq_f32, with
// eight independent chains per lane and eight waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o ab/valu_rates && ab/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(int iters, float seed, float* out) {
    float a[8];
    v2f p[8];
    double d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        a[u] = seed + (float)(threadIdx.x + u);
        p[u] = v2f{a[u], a[u] + 1.f};
        d[u] = (double)a[u];
    }
    const float m = 1.0000001f, c = 1e-9f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(m), "v"(c));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[u]) : "v"(v2f{m, m}), "v"(v2f{c, c}));
            if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[u]) : "v"(v2f{m, m}));
            if (KIND == 3) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[u]) : "v"((double)m), "v"((double)c));
            if (KIND == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[u]) : "v"((double)m));
            if (KIND == 5) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[u]) : "v"(a[u]));
            if (KIND == 6) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[u]) : "v"(d[u]));
            if (KIND == 7) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[u]));
            if (KIND == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[u]) : "v"(v2f{c, c}));
            if (KIND == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[u]) : "v"(m));
            if (KIND == 10) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[u]) : "v"((double)c));
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += a[u] + p[u].x + p[u].y + (float)d[u];
    if (acc == 123.456f) out[0] = acc;
}

template <int KIND>
static void run(const char* name, float* out) {
    const int iters = 4096, blocks = 256 * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_rate<KIND><<<blocks, 256>>>(iters, 1.f, out);
    hipEventRecord(a);
    k_rate<KIND><<<blocks, 256>>>(iters, 1.f, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double per_simd = 8.0 * iters * 8;   // wave instructions per SIMD
    printf("%-16s %.3f ms: %.2f clk per wave64 instruction per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd);
}

int main() {
    float* out; hipMalloc(&out, 4);
    run<0>("v_fma_f32", out);
    run<9>("v_mul_f32", out);
    run<1>("v_pk_fma_f32", out);
    run<2>("v_pk_mul_f32", out);
    run<8>("v_pk_add_f32", out);
    run<3>("v_fma_f64", out);
    run<4>("v_mul_f64", out);
    run<10>("v_max_f64", out);
    run<5>("v_cvt_f64_f32", out);
    run<6>("v_cvt_f32_f64", out);
    run<7>("v_rsq_f32", out);
    return 0;
}
