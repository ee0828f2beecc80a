// Does a matrix instruction stream of one wave overlap with the VALU stream of another wave on the same SIMD (gfx950)?
// Workgroups of 8 waves = 2 waves per SIMD (wave w and w + 4 share SIMD w % 4).  Waves 0-3 run MFMAs of one kind, waves 4-7
// run dependent-free v_fma_f32 chains; each half is also timed alone (the other half exits at once).  If the two streams
// overlapped, T(both) ~ max(T(mfma), T(valu)); if they share the pipe, T(both) ~ T(mfma) + T(valu).
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0: fp32 16x16x4, 1: bf16 16x16x32
__global__ void k(float *o, int iters, int run_mfma, int run_valu) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        if (!run_mfma) return;
        floatx4 a4[4];
        for (int i = 0; i < 4; ++i) a4[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        const float x = threadIdx.x * 0.001f, y = 1.0f - x;
        uint4 u = {threadIdx.x * 3u + 1u, threadIdx.x * 5u + 7u, 0x3f803f80u, 0x3f803f80u};
        const bf16x8 bx = __builtin_bit_cast(bf16x8, u), by = bx;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (KIND == 0) a4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4[i], 0, 0, 0);
                else a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, a4[i], 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) s += a4[i][0] + a4[i][3];
    } else {
        if (!run_valu) return;
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
        const float m = 1.0001f, c = 0.5f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c));
        }
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
float run(float *o, int iters, int a, int b) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        k<KIND><<<dim3(256), dim3(512)>>>(o, iters, a, b);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    float *o;
    (void)hipMalloc(&o, 256 * 512 * 4);
    const int iters = 20000;
    const char *names[2] = {"v_mfma_f32_16x16x4_f32  ", "v_mfma_f32_16x16x32_bf16"};
    for (int kind = 0; kind < 2; ++kind) {
        float tm = kind == 0 ? run<0>(o, iters, 1, 0) : run<1>(o, iters, 1, 0);
        float tv = kind == 0 ? run<0>(o, iters, 0, 1) : run<1>(o, iters, 0, 1);
        float tb = kind == 0 ? run<0>(o, iters, 1, 1) : run<1>(o, iters, 1, 1);
        printf("%s  mfma alone %.3f ms (4 per iter)   16 v_fma_f32 per iter alone %.3f ms   both %.3f ms   sum %.3f  max %.3f  -> %s\n",
               names[kind], tm, tv, tb, tm + tv, tm > tv ? tm : tv, tb < 0.75f * (tm + tv) ? "overlap" : "no overlap");
    }
    return 0;
}
