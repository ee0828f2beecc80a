// Issue rates of the matrix instructions the kernels use or could use (gfx950), one to eight waves per SIMD, and the cost of
// the exact 3-way bf16 split of an fp32 register pair.   hipcc --offload-arch=gfx950 -O3 mfma_rates.hip && ./a.out
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ void k(float *o, int iters) {
    floatx4 a4[4];
    floatx16 a16[2];
    for (int i = 0; i < 4; ++i) a4[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) a16[i][r] = 0.f;
    const float x = threadIdx.x * 0.001f, y = 1.0f - x;
    uint4 u = {threadIdx.x * 3u + 1u, threadIdx.x * 5u + 7u, 0x3f803f80u, 0x3f803f80u};
    const bf16x8 bx = __builtin_bit_cast(bf16x8, u), by = bx;
    float sp = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4[i], 0, 0, 0);
        } else if (KIND == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, a4[i], 0, 0, 0);
        } else if (KIND == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a16[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a16[i], 0, 0, 0);
        } else if (KIND == 3) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a16[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a16[i], 0, 0, 0);
        } else {
            // split of one fp32 pair into three packed bf16 pairs, 4 independent pairs per iteration
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float p = x + (float)(it + i), q = y - (float)(it + i);
                unsigned p1, p2, p3;
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p1) : "v"(p), "v"(q));
                float rp = p - __uint_as_float(p1 << 16), rq = q - __uint_as_float(p1 & 0xffff0000u);
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p2) : "v"(rp), "v"(rq));
                rp -= __uint_as_float(p2 << 16); rq -= __uint_as_float(p2 & 0xffff0000u);
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p3) : "v"(rp), "v"(rq));
                sp += __uint_as_float(p1 ^ p2 ^ p3);
            }
        }
    }
    float s = sp;
    for (int i = 0; i < 4; ++i) s += a4[i][0] + a4[i][3];
    for (int i = 0; i < 2; ++i) s += a16[i][0] + a16[i][15];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *o;
    (void)hipMalloc(&o, 256 * 4 * 512 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    const char *names[5] = {"v_mfma_f32_16x16x4_f32  ", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x2_f32  ", "v_mfma_f32_32x32x16_bf16", "3-way bf16 split (pair) "};
    const int per_iter[5] = {4, 4, 2, 2, 4};
    for (int waves = 1; waves <= 4; waves *= 2)
        for (int kind = 0; kind < 5; ++kind) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                dim3 g(256 * 4), b(64 * waves);
                switch (kind) {
                    case 0: k<0><<<g, b>>>(o, iters); break;
                    case 1: k<1><<<g, b>>>(o, iters); break;
                    case 2: k<2><<<g, b>>>(o, iters); break;
                    case 3: k<3><<<g, b>>>(o, iters); break;
                    default: k<4><<<g, b>>>(o, iters); break;
                }
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double cyc = best * 1e-3 * 2.4e9 / ((double)waves * iters * per_iter[kind]);
            printf("%s waves/SIMD=%d  %.3f ms  %.1f cycles (at 2.4 GHz) per instruction%s per SIMD\n", names[kind], waves, best, cyc,
                   kind == 4 ? " group (11 VALU)" : "");
        }
    return 0;
}
