// Exact 3-way bf16 split of a pair of fp32 values, two ways: the residuals by shift / and / subtract (11 instructions per pair), or by
// v_dot2_f32_bf16 against (-1, 0) / (0, -1) with the value as the accumulator (7 per pair).  Checks the pieces bit for bit on
// Gaussian-like inputs (incl. tiny and denormal ones) and times both.   hipcc --offload-arch=gfx950 -O3 split_dot2.hip -o bin/split_dot2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void split_ref(float x, float y, unsigned &p1, unsigned &p2, unsigned &p3) {
    p1 = cvt_pk(x, y);
    float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xffff0000u);
    p2 = cvt_pk(rx, ry);
    rx -= __uint_as_float(p2 << 16);
    ry -= __uint_as_float(p2 & 0xffff0000u);
    p3 = cvt_pk(rx, ry);
}
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
// builtins, not inline asm: a DOT result read by the next VALU instruction needs wait states that the compiler only inserts (and fills with
// independent work) when it sees both instructions -- with the dot in an asm statement the dependent conversion read a stale register
__device__ __forceinline__ bf2 cvt2(float x, float y) { f2 v = {x, y}; return __builtin_convertvector(v, bf2); }
__device__ __forceinline__ void split_dot(float x, float y, unsigned &p1, unsigned &p2, unsigned &p3) {
    // (-1, 0) and (0, -1), kept opaque in registers: as a literal, clang (ROCm 7.2) encodes 0x0000bf80 as the inline constant -1.0, which
    // the instruction reads as 0xbf800000 = (0, -1)
    unsigned ulo = 0x0000bf80u, uhi = 0xbf800000u;
    asm volatile("" : "+v"(ulo), "+v"(uhi));
    const bf2 mlo = __builtin_bit_cast(bf2, ulo), mhi = __builtin_bit_cast(bf2, uhi);
    const bf2 a = cvt2(x, y);
    float rx = __builtin_amdgcn_fdot2_f32_bf16(a, mlo, x, false), ry = __builtin_amdgcn_fdot2_f32_bf16(a, mhi, y, false);
    const bf2 b = cvt2(rx, ry);
    rx = __builtin_amdgcn_fdot2_f32_bf16(b, mlo, rx, false);
    ry = __builtin_amdgcn_fdot2_f32_bf16(b, mhi, ry, false);
    const bf2 c = cvt2(rx, ry);
    p1 = __builtin_bit_cast(unsigned, a); p2 = __builtin_bit_cast(unsigned, b); p3 = __builtin_bit_cast(unsigned, c);
}
__global__ void check_k(const float *in, unsigned *out, int n, int which) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * t + 1 >= n) return;
    unsigned a, b, c;
    if (which) split_dot(in[2 * t], in[2 * t + 1], a, b, c); else split_ref(in[2 * t], in[2 * t + 1], a, b, c);
    out[3 * t] = a; out[3 * t + 1] = b; out[3 * t + 2] = c;
}
template <int W>
__global__ void time_k(float *o, int iters) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.37f + threadIdx.x * 0.001f + i * 0.01f;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            unsigned a, b, c;
            if (W) split_dot(x[i], x[i + 1], a, b, c); else split_ref(x[i], x[i + 1], a, b, c);
            acc ^= a ^ b ^ c;
            x[i] = x[i] * 1.0001f; x[i + 1] = x[i + 1] * 0.9999f;
        }
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(acc) + x[0];
}
int main() {
    const int n = 1 << 20;
    std::vector<float> h(n);
    unsigned s = 12345;
    for (int i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        const float u = (s >> 8) * (1.0f / 16777216.0f);
        const int kind = i & 7;
        h[i] = kind == 0 ? 0.f : kind == 1 ? 1.f : kind == 2 ? std::exp2f(-140.f * u) : kind == 3 ? std::exp2f(-30.f * u) : kind == 4 ? -u : u;
    }
    float *din; unsigned *o0, *o1;
    hipMalloc(&din, n * 4); hipMalloc(&o0, n / 2 * 12); hipMalloc(&o1, n / 2 * 12);
    hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
    check_k<<<n / 2 / 256, 256>>>(din, o0, n, 0);
    check_k<<<n / 2 / 256, 256>>>(din, o1, n, 1);
    std::vector<unsigned> r0(n / 2 * 3), r1(n / 2 * 3);
    hipMemcpy(r0.data(), o0, r0.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r1.data(), o1, r1.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0, bad_big = 0;
    for (size_t i = 0; i < r0.size(); ++i)
        if (r0[i] != r1[i]) {
            ++bad;
            const size_t t = i / 3;
            if (std::fabs(h[2 * t]) > 1e-30f || std::fabs(h[2 * t + 1]) > 1e-30f) { if (bad_big < 5) printf("diff at pair %zu (%g, %g): piece %zu %08x vs %08x\n", t, h[2 * t], h[2 * t + 1], i % 3, r0[i], r1[i]); ++bad_big; }
        }
    printf("pieces differing: %ld of %zu (%ld of them with an input above 1e-30)\n", bad, r0.size(), bad_big);
    // sums: do the three pieces add up to the input exactly?
    long inexact = 0;
    for (size_t t = 0; t < (size_t)n / 2; ++t)
        for (int half = 0; half < 2; ++half) {
            auto piece = [&](unsigned w) { unsigned u = half ? (w & 0xffff0000u) : (w << 16); float f; std::memcpy(&f, &u, 4); return (double)f; };
            const double sum = piece(r1[3 * t]) + piece(r1[3 * t + 1]) + piece(r1[3 * t + 2]);
            if (sum != (double)h[2 * t + half] && std::fabs(h[2 * t + half]) > 1e-30f) ++inexact;
        }
    printf("dot2 split: inputs above 1e-30 not reproduced exactly by their three pieces: %ld\n", inexact);
    float *o; hipMalloc(&o, 256 * 4 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int waves = 1; waves <= 4; waves *= 2)
        for (int which = 0; which < 2; ++which) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (which) time_k<1><<<256 * 4, 64 * waves>>>(o, iters); else time_k<0><<<256 * 4, 64 * waves>>>(o, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("%s waves/SIMD=%d  %.3f ms  %.1f cycles per pair per SIMD (incl. 2 mul + 3 xor)\n", which ? "dot2 split" : "ref split ", waves, best,
                   best * 1e-3 * 2.4e9 / ((double)waves * iters * 4));
        }
    return 0;
}
