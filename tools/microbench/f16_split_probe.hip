// Semantics probe for the f16-pair second layer (round 6): v_cvt_pkrtz_f16_f32 / v_fma_mix_f32 / v_cvt_pk_f16_f32 split of a value in
// [0, 1] into two f16 pieces, f16 subnormals through v_mfma_f32_16x16x32_f16, and the instruction's K-slot map (lane (lo, g) holds
// k = 8g + j).  Build + run:  hipcc -O3 --offload-arch=gfx950 tools/microbench/f16_split_probe.hip -o /tmp/f16probe && /tmp/f16probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pkrtz(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b)); }
__device__ __forceinline__ unsigned pkrne(float a, float b) { unsigned r; asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float res_lo(unsigned h, float z) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(z)); return r; }
__device__ __forceinline__ float res_hi(unsigned h, float z) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(z)); return r; }

__device__ __forceinline__ void split_scaled(float x, float y, unsigned &h1, unsigned &h2) {          // td_split_h2_scaled (edge16.hip)
    const float S = 32768.0f;
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h1), "=&v"(h2) : "v"(x), "v"(y), "s"(S));
}
__global__ void split_scaled_kernel(const float *in, unsigned *h1o, unsigned *h2o, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * t + 1 >= n) return;
    unsigned h1, h2;
    split_scaled(in[2 * t], in[2 * t + 1], h1, h2);
    h1o[t] = h1; h2o[t] = h2;
}
__global__ void split_kernel(const float *in, unsigned *h1o, unsigned *h2o, float *ro, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * t + 1 >= n) return;
    const float a = in[2 * t], b = in[2 * t + 1];
    const unsigned h1 = pkrtz(a, b);
    const float ra = res_lo(h1, a), rb = res_hi(h1, b);
    h1o[t] = h1; h2o[t] = pkrne(ra, rb); ro[2 * t] = ra; ro[2 * t + 1] = rb;
}
// D[16][16] = A[16][32] B[32][16] with A, B given as f16 bit patterns, row-major; one wave
__global__ void mfma_kernel(const unsigned short *A, const unsigned short *B, float *D) {
    const int lane = threadIdx.x, lo = lane & 15, g = lane >> 4;
    unsigned short a[8], b[8];
    for (int j = 0; j < 8; ++j) { a[j] = A[lo * 32 + 8 * g + j]; b[j] = B[(8 * g + j) * 16 + lo]; }
    uint4 aq = make_uint4(a[0] | (a[1] << 16), a[2] | (a[3] << 16), a[4] | (a[5] << 16), a[6] | (a[7] << 16));
    uint4 bq = make_uint4(b[0] | (b[1] << 16), b[2] | (b[3] << 16), b[4] | (b[5] << 16), b[6] | (b[7] << 16));
    floatx4_t c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, aq), __builtin_bit_cast(half8, bq), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + lo] = c[r];
}

static float h2f(unsigned short h) {
    const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    float v = e == 0 ? ldexpf((float)m, -24) : ldexpf((float)(m | 1024), e - 25);
    return s ? -v : v;
}
static unsigned short f2h_trunc(float x) {      // positive values below 65504, truncation (enough for the probe's inputs)
    if (x <= 0.f) return 0;
    int e; float m = frexpf(x, &e);           // x = m 2^e, m in [0.5, 1)
    int E = e - 1 + 15;
    if (E <= 0) return (unsigned short)floorf(ldexpf(x, 24));
    return (unsigned short)((E << 10) | ((int)floorf(ldexpf(m, 11)) & 1023));
}

int main() {
    const int n = 1 << 16;
    std::vector<float> in(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const float u = (float)rand() / RAND_MAX;
        in[i] = i % 4 == 0 ? u : (i % 4 == 1 ? u * 0.03125f : (i % 4 == 2 ? ldexpf(u, -(rand() % 30)) : (i % 8 == 3 ? 0.f : 1.0f - ldexpf(u, -12))));
    }
    float *din, *dr; unsigned *d1, *d2;
    hipMalloc(&din, n * 4); hipMalloc(&dr, n * 4); hipMalloc(&d1, n * 2); hipMalloc(&d2, n * 2);
    hipMemcpy(din, in.data(), n * 4, hipMemcpyHostToDevice);
    split_kernel<<<n / 2 / 256, 256>>>(din, d1, d2, dr, n);
    std::vector<unsigned> h1(n / 2), h2(n / 2); std::vector<float> r(n);
    hipMemcpy(h1.data(), d1, n * 2, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), d2, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost);
    double worst_abs = 0, worst_rel = 0; int bad_res = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned short p1 = (h1[i / 2] >> (16 * (i & 1))) & 0xffff, p2 = (h2[i / 2] >> (16 * (i & 1))) & 0xffff;
        const double z = in[i], rec = (double)h2f(p1) + (double)h2f(p2);
        if ((double)r[i] != z - (double)h2f(p1)) ++bad_res;
        worst_abs = fmax(worst_abs, fabs(rec - z));
        if (z > ldexp(1.0, -10)) worst_rel = fmax(worst_rel, fabs(rec - z) / z);
    }
    printf("split: residual exact for %d of %d, max |z - (h1 + h2)| = %.3e (2^-24 = %.3e), max relative (z > 2^-10) = %.3e (2^-22 = %.3e)\n",
           n - bad_res, n, worst_abs, ldexp(1.0, -24), worst_rel, ldexp(1.0, -22));
    split_scaled_kernel<<<n / 2 / 256, 256>>>(din, d1, d2, n);
    hipMemcpy(h1.data(), d1, n * 2, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), d2, n * 2, hipMemcpyDeviceToHost);
    double sw_abs = 0, sw_rel = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned short p1 = (h1[i / 2] >> (16 * (i & 1))) & 0xffff, p2 = (h2[i / 2] >> (16 * (i & 1))) & 0xffff;
        const double z = in[i], rec = ((double)h2f(p1) + (double)h2f(p2)) / 32768.0;
        sw_abs = fmax(sw_abs, fabs(rec - z));
        if (z > ldexp(1.0, -18)) sw_rel = fmax(sw_rel, fabs(rec - z) / z);
    }
    printf("scaled split (S = 2^15): max |z - (h1 + h2) / S| = %.3e, max relative (z > 2^-18) = %.3e (2^-22 = %.3e)\n", sw_abs, sw_rel, ldexp(1.0, -22));
    // MFMA: random f16 incl. subnormals, against a double product of the decoded values
    std::vector<unsigned short> A(16 * 32), B(32 * 16);
    for (auto &x : A) x = f2h_trunc(ldexpf((float)rand() / RAND_MAX, -(rand() % 26)));
    for (auto &x : B) x = f2h_trunc(ldexpf((float)rand() / RAND_MAX, 8 - (rand() % 12)));
    for (int k = 0; k < 32; ++k) A[3 * 32 + k] = 1;            // row 3: the smallest subnormal, 2^-24
    unsigned short *dA, *dB; float *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    mfma_kernel<<<1, 64>>>(dA, dB, dD);
    std::vector<float> D(256);
    hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
    double wrel = 0; double sub_row = 0, sub_ref = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double ref = 0;
        for (int k = 0; k < 32; ++k) ref += (double)h2f(A[i * 32 + k]) * (double)h2f(B[k * 16 + j]);
        wrel = fmax(wrel, fabs(D[i * 16 + j] - ref) / fmax(fabs(ref), 1e-30));
        if (i == 3) { sub_row += D[i * 16 + j]; sub_ref += ref; }
    }
    printf("mfma_f32_16x16x32_f16: max relative error vs double = %.3e; subnormal row: sum %.6e (expected %.6e)\n", wrel, sub_row, sub_ref);
    // known answers: (a) A = 1.0, B = 2^-24 (subnormal); (b) A = 2^-24, B = 1.0; (c) normal operands only, random; (d) A = 2^-14 (smallest normal), B = 2^-14
    auto run = [&](const char *what) {
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        mfma_kernel<<<1, 64>>>(dA, dB, dD);
        hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
        double w = 0, ref00 = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double ref = 0;
            for (int k = 0; k < 32; ++k) ref += (double)h2f(A[i * 32 + k]) * (double)h2f(B[k * 16 + j]);
            if (i == 0 && j == 0) ref00 = ref;
            w = fmax(w, fabs(D[i * 16 + j] - ref) / fmax(fabs(ref), 1e-300));
        }
        printf("%s: D[0][0] = %.9e (expected %.9e), max relative error %.3e\n", what, D[0], ref00, w);
    };
    for (auto &x : A) x = 0x3C00; for (auto &x : B) x = 1; run("(a) 1.0 x 2^-24");
    for (auto &x : A) x = 1; for (auto &x : B) x = 0x3C00; run("(b) 2^-24 x 1.0");
    for (auto &x : A) x = f2h_trunc(0.25f + 0.5f * rand() / RAND_MAX); for (auto &x : B) x = f2h_trunc(1.0f + 100.0f * rand() / RAND_MAX); run("(c) normal operands");
    for (auto &x : A) x = 0x0400; for (auto &x : B) x = 0x0400; run("(d) 2^-14 x 2^-14");
    for (auto &x : A) x = f2h_trunc(ldexpf(0.5f + 0.5f * rand() / RAND_MAX, -(rand() % 24))); for (auto &x : B) x = 0x3C00; run("(e) wide-range A x 1.0");
    for (auto &x : A) x = 0x03ff; for (auto &x : B) x = 0x3C00; run("(f) largest subnormal x 1.0");
    for (auto &x : A) x = 0x3C00; for (auto &x : B) x = 0x03ff; run("(g) 1.0 x largest subnormal");
    for (auto &x : A) x = 1; for (auto &x : B) x = f2h_trunc(1.0f + 100.0f * rand() / RAND_MAX); run("(h) 2^-24 x random normal");
    for (auto &x : A) x = 0x0155; for (auto &x : B) x = f2h_trunc(1.0f + 100.0f * rand() / RAND_MAX); run("(i) subnormal 0x155 x random normal");
    for (auto &x : A) x = 0x0155; for (auto &x : B) x = 0x3E00; run("(j) subnormal 0x155 x 1.5");
    for (auto &x : A) x = 0x0001; for (auto &x : B) x = 0x3E00; run("(k) 2^-24 x 1.5");
    for (auto &x : A) x = 0x0001; for (auto &x : B) x = 0x3FFF; run("(l) 2^-24 x 1.999");
    for (int k = 0; k < 32 * 16; ++k) { A[k] = 0; B[k] = 0; }
    A[0] = 0x0001; B[0] = 0x3FFF; run("(m) single product 2^-24 x 1.999");
    A[0] = 0x0155; B[0] = 0x3FFF; run("(n) single product 0x155 x 1.999");
    A[0] = 0x0400; B[0] = 0x3FFF; run("(o) single product 2^-14 x 1.999");
    return (bad_res == 0 && wrel < 1e-6 && fabs(sub_row - sub_ref) <= 1e-6 * fabs(sub_ref)) ? 0 : 1;
}
