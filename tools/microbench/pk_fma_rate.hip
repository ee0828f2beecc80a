#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void scalar_k(float *o, float a, float b, int iters) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void packed_k(float *o, float a, float b, int iters) {
    f2 x[8];
    f2 av = {a, a}, bv = {b, b};
    for (int i = 0; i < 8; ++i) x[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(bv));
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void packed_mul_k(float *o, float a, float b, int iters) {
    f2 x[8];
    f2 av = {a, a};
    for (int i = 0; i < 8; ++i) x[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(av));
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float *o; hipMalloc(&o, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int waves = 1; waves <= 8; waves *= 2)
    for (int which = 0; which < 3; ++which) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            dim3 g(256 * 4), b(64 * waves);   // 4 WGs per CU x waves per WG  -> waves per SIMD = waves
            if (which == 0) scalar_k<<<g, b>>>(o, 0.999f, 0.001f, iters);
            else if (which == 1) packed_k<<<g, b>>>(o, 0.999f, 0.001f, iters);
            else packed_mul_k<<<g, b>>>(o, 0.999f, 0.001f, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        // per SIMD: waves * iters * n_inst instructions
        const int ninst = which == 0 ? 16 : 8;
        double cyc = best * 1e-3 * 2.4e9 / ((double)waves * iters * ninst);
        double flops = (which == 2 ? 1.0 : 2.0) * 16.0 * iters * 64.0 * waves * 256 * 4 / (best * 1e-3) / 1e12;
        printf("%s waves/SIMD=%d  %.3f ms  %.2f cyc/inst/SIMD  %.1f TFLOP/s\n", which == 0 ? "v_fma_f32   " : which == 1 ? "v_pk_fma_f32" : "v_pk_mul_f32", waves, best, cyc, flops);
    }
    return 0;
}
