// What does the shader clock run at under sustained matrix load on the whole chip (gfx950)?  s_memtime counts core clock cycles,
// s_memrealtime the constant 100 MHz reference; their ratio over a long kernel is the average engine clock of that CU.  The
// roofline peaks (157.3 TFLOP/s fp32 matrix, 2.5 PFLOP/s bf16) assume 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 core_clock.hip -o /tmp/cc && /tmp/cc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0: fp32 16x16x4 MFMA, 1: bf16 16x16x32 MFMA, 2: v_fma_f32 only, 3: MFMA fp32 + VALU mix (1 : 6 like the edge passes)
__global__ void k(float *o, unsigned long long *t, int iters) {
    floatx4 a4[4];
    for (int i = 0; i < 4; ++i) a4[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    const float x = threadIdx.x * 0.001f, y = 1.0f - x;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = x + i;
    uint4 u = {threadIdx.x * 3u + 1u, threadIdx.x * 5u + 7u, 0x3f803f80u, 0x3f803f80u};
    const bf16x8 bx = __builtin_bit_cast(bf16x8, u), by = bx;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (KIND == 0 || KIND == 3) a4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4[i], 0, 0, 0);
            if (KIND == 1) a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, a4[i], 0, 0, 0);
            if (KIND == 2 || KIND == 3) {
#pragma unroll
                for (int j = 0; j < (KIND == 2 ? 8 : 6); ++j) f[j] = __builtin_fmaf(f[j], y, x);
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += a4[i][0] + a4[i][3];
    for (int i = 0; i < 8; ++i) s += f[i];
    if (s == 12345.678f) o[0] = s;
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND> void run(const char *name, int iters) {
    const int G = 256 * 2, B = 512;           // 2 workgroups of 8 waves per CU: 4 waves per SIMD
    float *o; unsigned long long *t;
    hipMalloc(&o, 4); hipMalloc(&t, sizeof(unsigned long long) * 2 * G);
    k<KIND><<<G, B>>>(o, t, iters / 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<KIND><<<G, B>>>(o, t, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * G);
    hipMemcpy(h.data(), t, sizeof(unsigned long long) * 2 * G, hipMemcpyDeviceToHost);
    std::vector<double> mhz;
    for (int b = 0; b < G; ++b) if (h[2 * b + 1]) mhz.push_back(100.0 * (double)h[2 * b] / (double)h[2 * b + 1]);
    std::sort(mhz.begin(), mhz.end());
    printf("%-34s %8.2f ms   core clock MHz over the workgroups: min %7.1f  median %7.1f  max %7.1f\n", name, ms, mhz.front(),
           mhz[mhz.size() / 2], mhz.back());
    hipFree(o); hipFree(t);
}

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<2>("v_fma_f32 only", 400000);
        run<0>("fp32 MFMA 16x16x4", 400000);
        run<1>("bf16 MFMA 16x16x32", 800000);
        run<3>("fp32 MFMA + 6 v_fma per MFMA", 200000);
    }
    return 0;
}
