// Device-side helpers for the gfx950 kernels (wave64, MFMA 32x32x2 f32 fragment maps).
#pragma once

#include <hip/hip_runtime.h>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32 fragment maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[row = l & 31][k = l >> 5];  B: lane l holds B[k = l >> 5][col = l & 31]
//   C/D: lane l, reg r holds D[row = erow(r, l >> 5)][col = l & 31]
__device__ __forceinline__ int td_erow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// K permutation used by every 128-deep contraction here: k-step s (0..63) pairs
//   hi = 0 -> k = 8*(s/4) + (s%4),  hi = 1 -> k = 8*(s/4) + 4 + (s%4)
// so that a lane's A operands for 4 consecutive k-steps are 4 consecutive floats (one 16-byte load).
__host__ __device__ __forceinline__ int td_kmap(int s, int hi) { return 8 * (s >> 2) + 4 * hi + (s & 3); }

__device__ __forceinline__ floatx16 td_mfma(float a, float b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// clamp to [0, 1] in the form the backend folds into the producing instruction's output modifier (v_fma_f32 ... clamp): HIP's __saturatef
// becomes two compare + select pairs once the operands are elements of a vector
__device__ __forceinline__ float td_clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }

// sqrt of a squared distance.  sqrtf() is 14 instructions here (IEEE rounding with denormal scaling); this is v_rsq_f32 and one Heron step in
// FMAs -- six instructions, correctly rounded except for rare half-ulp ties of the intermediate (every margin of the GPU suite, the 1000-step
// trajectories included, stayed the same to all printed digits; x2h key / value pass -1 %).  The clamp keeps rsq finite for a pad's d^2 = 0 (the
// row itself stands in for a missing neighbour): 1e-15 instead of 0, the same Gaussians to the last bit.
__device__ __forceinline__ float td_sqrt_d2(float x) {
    x = fmaxf(x, 1e-30f);
    const float rs = __builtin_amdgcn_rsqf(x);
    const float y = x * rs, h = 0.5f * rs;
    return fmaf(fmaf(-y, y, x), h, y);
}

// ---- cross-lane reductions without LDS round trips (DPP modifiers + gfx950 v_permlane32_swap) -------------
template <int CTRL>
__device__ __forceinline__ float td_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141; // lane i <-> 7 - i inside each group of 8
constexpr int DPP_ROW_MIRROR = 0x140;      // lane i <-> 15 - i inside each row of 16
constexpr int DPP_ROW_ROR8 = 0x128;        // rotate right by 8 inside each row of 16

// Sum over the 8 consecutive lanes of a head group (lanes 8g .. 8g+7), result in every lane of the group.
__device__ __forceinline__ float td_sum8(float v) {
    v += td_dpp<DPP_QUAD_XOR1>(v);
    v += td_dpp<DPP_QUAD_XOR2>(v);
    v += td_dpp<DPP_ROW_HALF_MIRROR>(v);
    return v;
}

// v_permlane32_swap: lanes 32-63 of the first operand trade places with lanes 0-31 of the second.  Fed with two
// copies of v it yields lo = [v_lo, v_lo] and hi = [v_hi, v_hi] in one instruction.
// Inline asm on purpose: clang (ROCm 7.2) lowers BOTH elements of __builtin_amdgcn_permlane32_swap's result to
// extractvalue 0, i.e. r[1] silently aliases r[0].  The leading s_nop 1 covers the "VALU write -> permlane read"
// hazard (2 wait states), which the compiler does not pad inside an asm statement.
__device__ __forceinline__ void td_split_halves(float v, float &lo, float &hi) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    lo = a;
    hi = b;
}
// other half-wave's value (lane l <-> l ^ 32)
__device__ __forceinline__ float td_swap32(float v) {
    float lo, hi;
    td_split_halves(v, lo, hi);
    return (threadIdx.x & 32) ? lo : hi;
}
__device__ __forceinline__ float td_sum_halves(float v) {
    float lo, hi;
    td_split_halves(v, lo, hi);
    return lo + hi;
}
__device__ __forceinline__ float td_max_halves(float v) {
    float lo, hi;
    td_split_halves(v, lo, hi);
    return fmaxf(lo, hi);
}

// v_permlane16_swap: rows (16 lanes) 1 and 3 of the first operand trade places with rows 0 and 2 of the second.
// Fed with two copies of v: a = [r0, r0, r2, r2], b = [r1, r1, r3, r3], so a + b is the xor-16 butterfly sum.
__device__ __forceinline__ float td_sum_rows16(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

// maximum over the four rows of 16 lanes (lanes lo, lo + 16, lo + 32, lo + 48), result in all four
__device__ __forceinline__ float td_max_groups(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return td_max_halves(fmaxf(a, b));
}

// Sum over the 32 lanes of a half-wave (lanes with equal l >> 5), result in every lane of the half.  All VALU.
__device__ __forceinline__ float td_sum32(float v) {
    v = td_sum8(v);
    v += td_dpp<DPP_ROW_ROR8>(v);          // the two groups of 8 inside a row of 16
    return td_sum_rows16(v);               // the two rows of a half-wave
}

__device__ __forceinline__ float td_sum64(float v) {
    v = td_sum32(v);
    return td_sum_halves(v);
}

// Minimum of a 64-bit key over the 64 lanes, result in every lane: 4 DPP steps inside each row of 16, then the two
// cross-row exchanges (v_permlane16_swap / v_permlane32_swap) -- all VALU, no LDS crossbar (a __shfl_xor butterfly on a
// 64-bit value is 12 ds_bpermute per reduction and dominated the k-NN extraction loops).
template <int CTRL>
__device__ __forceinline__ unsigned long long td_dpp_u64(unsigned long long v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, false);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ unsigned long long td_min_u64(unsigned long long a, unsigned long long b) { return b < a ? b : a; }
__device__ __forceinline__ unsigned long long td_wave_min_u64(unsigned long long v) {
    v = td_min_u64(v, td_dpp_u64<DPP_QUAD_XOR1>(v));
    v = td_min_u64(v, td_dpp_u64<DPP_QUAD_XOR2>(v));
    v = td_min_u64(v, td_dpp_u64<DPP_ROW_HALF_MIRROR>(v));
    v = td_min_u64(v, td_dpp_u64<DPP_ROW_MIRROR>(v));
    {
        unsigned alo = (unsigned)v, blo = (unsigned)v, ahi = (unsigned)(v >> 32), bhi = (unsigned)(v >> 32);
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(alo), "+v"(blo));
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(ahi), "+v"(bhi));
        v = td_min_u64(((unsigned long long)ahi << 32) | alo, ((unsigned long long)bhi << 32) | blo);
    }
    {
        unsigned alo = (unsigned)v, blo = (unsigned)v, ahi = (unsigned)(v >> 32), bhi = (unsigned)(v >> 32);
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(alo), "+v"(blo));
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ahi), "+v"(bhi));
        v = td_min_u64(((unsigned long long)ahi << 32) | alo, ((unsigned long long)bhi << 32) | blo);
    }
    return v;
}

// Products and sums rounded one by one, whatever the surrounding expression: HIP's __fmul_rn / __fadd_rn are plain `x * y` / `x + y`
// (unless OCML_BASIC_ROUNDED_OPERATIONS is defined) and -ffp-contract=fast-honor-pragmas, hipcc's default, may fuse them into FMAs --
// differently in every kernel the expression is inlined into.  The pragma takes contraction off for these two operations only.
__device__ __forceinline__ float td_mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float td_add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}

// distance^2 with the project's fixed association and no FMA contraction (oracle/shims.py)
__device__ __forceinline__ float td_dist2(float dx, float dy, float dz) {
    return td_add_rn(td_add_rn(td_mul_rn(dx, dx), td_mul_rn(dy, dy)), td_mul_rn(dz, dz));
}

// 16-byte-per-lane async global -> LDS copy (global_load_lds_dwordx4): LDS address = wave-uniform base + lane * 16.
__device__ __forceinline__ void td_glds16(const float4 *gsrc_lane, float4 *lds_wave_base) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc_lane,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
#endif
}


// Stage `n4` float4 (a multiple of 64) from global memory into LDS with the async copy, all waves of the workgroup
// cooperating; the caller's next __syncthreads() waits for the copies (vmcnt(0)) and publishes them.
__device__ __forceinline__ void td_stage_lds16(const float4 *__restrict__ src, float4 *__restrict__ dst, int n4, int tid,
                                               int nthreads) {
    const int lane = tid & 63;
    for (int idx = tid; idx < n4; idx += nthreads) td_glds16(src + idx, dst + (idx - lane));
}
