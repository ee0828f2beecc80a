// Issue cost of single VALU instructions on gfx950 at the edge passes' occupancy (3 waves per SIMD = one 768-thread workgroup per CU) and at
// 1 / 2 / 4 waves: each kernel runs a long stream of ONE instruction on eight independent register sets (no dependent back-to-back pairs).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/valu_rates.hip -o tools/microbench/bin/valu_rates && tools/microbench/bin/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(NAME, BODY)                                                                                   \
    __global__ void NAME(float *out, int iters, unsigned long long *cyc) {                                  \
        extern __shared__ float lds[];                                                                       \
        float a[8], b[8];                                                                                    \
        unsigned u[8];                                                                                       \
        for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; b[i] = 0.5f + i * 1e-2f; u[i] = threadIdx.x + i; } \
        const float S = 32768.0f;                                                                            \
        (void)S;                                                                                             \
        __syncthreads();                                                                                     \
        const unsigned long long t0 = __builtin_readcyclecounter();                                          \
        for (int it = 0; it < iters; ++it) {                                                                 \
            _Pragma("unroll") for (int rr = 0; rr < 8; ++rr) { BODY }                                        \
        }                                                                                                    \
        const unsigned long long t1 = __builtin_readcyclecounter();                                          \
        float s = 0.f;                                                                                       \
        for (int i = 0; i < 8; ++i) s += a[i] + b[i] + __uint_as_float(u[i] & 0x3fffffu);                    \
        if (s == 12345.678f) out[0] = s;                                                                     \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                     \
    }
#define ASM8(STR, ...) REP8(ASM1)
// one instruction per register set, eight sets
#define I_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
#define I_FMA_CLAMP(i) asm volatile("v_fma_f32 %0, %0, %1, %1 clamp" : "+v"(a[i]) : "v"(b[i]));
#define I_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define I_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define I_MAX(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define I_LDEXP(i) asm volatile("v_ldexp_f32 %0, %0, 1" : "+v"(a[i]));
#define I_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define I_RSQ(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
#define I_PKRTZ(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b[i]));
#define I_PKRNE(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b[i]));
#define I_PKBF16(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(b[i]));
#define I_MIXF32(i) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(a[i]) : "v"(u[i]), "v"(b[i]));
#define I_MIXLO(i) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(b[i]));
#define I_MIXHI(i) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(b[i]));
#define I_MIXLO_S(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(u[i]) : "v"(b[i]), "s"(S));
#define I_PKMULH(i) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
#define I_PKMAXH(i) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
#define I_PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double *>(&a[i & 6])) : "v"(*reinterpret_cast<double *>(&b[i & 6])));
#define I_LSHL(i) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
#define I_AND(i) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
#define I_SUB(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define I_DPP(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
#define I_PERM(i) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]), "v"(u[(i + 3) & 7]));
#define I_CVTF16(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define I_DOT2(i) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]));
KERNEL(k_fma, REP8(I_FMA))
KERNEL(k_fma_clamp, REP8(I_FMA_CLAMP))
KERNEL(k_mul, REP8(I_MUL))
KERNEL(k_add, REP8(I_ADD))
KERNEL(k_max, REP8(I_MAX))
KERNEL(k_ldexp, REP8(I_LDEXP))
KERNEL(k_exp, REP8(I_EXP))
KERNEL(k_rsq, REP8(I_RSQ))
KERNEL(k_pkrtz, REP8(I_PKRTZ))
KERNEL(k_pkrne, REP8(I_PKRNE))
KERNEL(k_pkbf16, REP8(I_PKBF16))
KERNEL(k_mixf32, REP8(I_MIXF32))
KERNEL(k_mixlo, REP8(I_MIXLO))
KERNEL(k_mixhi, REP8(I_MIXHI))
KERNEL(k_mixlo_sgpr, REP8(I_MIXLO_S))
KERNEL(k_pkmul_f16, REP8(I_PKMULH))
KERNEL(k_pkmax_f16, REP8(I_PKMAXH))
KERNEL(k_pkfma_f32, REP8(I_PKFMA32))
KERNEL(k_lshl, REP8(I_LSHL))
KERNEL(k_and, REP8(I_AND))
KERNEL(k_sub, REP8(I_SUB))
KERNEL(k_dpp, REP8(I_DPP))
KERNEL(k_perm, REP8(I_PERM))
KERNEL(k_cvt_f32_f16, REP8(I_CVTF16))
KERNEL(k_dot2c_f16, REP8(I_DOT2))

typedef void (*kfn)(float *, int, unsigned long long *);
int main() {
    struct { const char *name; kfn fn; } ks[] = {
        {"v_fma_f32", k_fma}, {"v_fma_f32 clamp", k_fma_clamp}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_sub_f32", k_sub}, {"v_max_f32", k_max},
        {"v_ldexp_f32", k_ldexp}, {"v_exp_f32", k_exp}, {"v_rsq_f32", k_rsq}, {"v_cvt_pkrtz_f16_f32", k_pkrtz}, {"v_cvt_pk_f16_f32", k_pkrne},
        {"v_cvt_pk_bf16_f32", k_pkbf16}, {"v_cvt_f32_f16", k_cvt_f32_f16}, {"v_fma_mix_f32 (f16 src)", k_mixf32}, {"v_fma_mixlo_f16", k_mixlo}, {"v_fma_mixhi_f16", k_mixhi},
        {"v_fma_mixlo_f16 (sgpr src)", k_mixlo_sgpr}, {"v_pk_mul_f16", k_pkmul_f16}, {"v_pk_max_f16", k_pkmax_f16}, {"v_pk_fma_f32", k_pkfma_f32},
        {"v_dot2c_f32_f16", k_dot2c_f16}, {"v_lshlrev_b32", k_lshl}, {"v_and_b32 (literal)", k_and}, {"v_add_f32_dpp", k_dpp}, {"v_perm_b32", k_perm}};
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 1024); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    printf("cycles per wave64 instruction per SIMD (s_memtime over the loop, mean over 256 workgroups; 8 x 8 x %d instructions per wave)\n", iters);
    printf("%-28s %10s %10s %10s %10s\n", "instruction", "1 wave", "2 waves", "3 waves", "4 waves");
    for (auto &k : ks) {
        printf("%-28s", k.name);
        for (int w = 1; w <= 4; ++w) {
            const int threads = 256 * w;           // w waves per SIMD, one workgroup per CU (100 KiB of LDS keeps a second one away)
            hipFuncSetAttribute(reinterpret_cast<const void *>(k.fn), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
            k.fn<<<256, threads, 100 * 1024>>>(out, 10, cyc);
            k.fn<<<256, threads, 100 * 1024>>>(out, iters, cyc);
            std::vector<unsigned long long> h(256);
            hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
            double mean = 0;
            for (auto c : h) mean += (double)c;
            mean /= 256;
            // s_memtime counts shader cycles; instructions per SIMD in the loop = w waves x 64 x iters
            printf(" %10.2f", mean / (64.0 * iters * w));
        }
        printf("\n");
    }
    return 0;
}
