// How many independent VALU instructions hide behind one matrix instruction of the SAME wave (gfx950)?
// Each wave runs a stream of [1 MFMA + k v_fma_f32] groups (4 independent accumulators, 8 independent VALU chains), at 1, 2 and
// 3 waves per SIMD.  Reported: time per group relative to k = 0, and the VALU-only time of the same k (no MFMA) -- if the
// fillers hide, T(k) stays at T(0) until the gap is full; if the pipes are shared, T(k) = T(0) + T(valu only).
//   hipcc --offload-arch=gfx950 -O3 mfma_fillers.hip -o /tmp/fill && /tmp/fill
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int K, bool MFMA>   // KIND 0: fp32 16x16x4, 1: bf16 16x16x32;  K fillers per MFMA
__global__ void k(float *o, int iters) {
    floatx4 a4[4];
    for (int i = 0; i < 4; ++i) a4[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    const float x = threadIdx.x * 0.001f, y = 1.0f - x;
    uint4 u = {threadIdx.x * 3u + 1u, threadIdx.x * 5u + 7u, 0x3f803f80u, 0x3f803f80u};
    const bf16x8 bx = __builtin_bit_cast(bf16x8, u), by = bx;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    const float m = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MFMA) {
                if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a4[i]) : "v"(x), "v"(y));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a4[i]) : "v"(bx), "v"(by));
            }
#pragma unroll
            for (int j = 0; j < K; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i * K + j) & 7]) : "v"(m), "v"(c));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += a4[i][0] + a4[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int K, bool MFMA>
float run(float *o, int iters, int waves_per_simd) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        k<KIND, K, MFMA><<<dim3(256), dim3(256 * waves_per_simd)>>>(o, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

template <int KIND, int K>
void line(float *o, int iters, int w, float t0) {
    const float both = run<KIND, K, true>(o, iters, w), valu = K ? run<KIND, K, false>(o, iters, w) : 0.f;
    // cycles per group per SIMD at 2.4 GHz: iters * 4 groups per wave, w waves per SIMD
    const double cyc = both * 1e-3 * 2.4e9 / (iters * 4.0 * w);
    printf("  k=%d  mfma+valu %.3f ms (%.1f cyc/group/wave-slot)  valu only %.3f ms  mfma only %.3f ms  -> %s\n", K, both, cyc, valu, t0,
           both < t0 + 0.5f * valu ? "fillers (mostly) hidden" : "added on top");
}

template <int KIND>
void sweep(float *o, int iters) {
    for (int w = 1; w <= 3; ++w) {
        printf("%s, %d wave(s) per SIMD\n", KIND ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_16x16x4_f32", w);
        const float t0 = run<KIND, 0, true>(o, iters, w);
        line<KIND, 0>(o, iters, w, t0);
        line<KIND, 1>(o, iters, w, t0);
        line<KIND, 2>(o, iters, w, t0);
        line<KIND, 3>(o, iters, w, t0);
        line<KIND, 4>(o, iters, w, t0);
        line<KIND, 6>(o, iters, w, t0);
        line<KIND, 8>(o, iters, w, t0);
    }
}

int main() {
    float *o;
    (void)hipMalloc(&o, sizeof(float) * 256 * 1024);
    const int iters = 20000;
    sweep<1>(o, iters);
    sweep<0>(o, iters);
    return 0;
}
