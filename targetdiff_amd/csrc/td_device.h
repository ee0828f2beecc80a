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

// Sum over the 8 consecutive lanes of a head group (lanes 8g .. 8g+7), result in every lane of the group.
__device__ __forceinline__ float td_sum8(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}

// Sum over the 32 lanes of a half-wave (lanes with equal l >> 5), result in every lane of the half.
__device__ __forceinline__ float td_sum32(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    return v;
}

__device__ __forceinline__ float td_sum64(float v) {
    v = td_sum32(v);
    v += __shfl_xor(v, 32);
    return v;
}

// distance^2 with the project's fixed association and no FMA contraction (oracle/shims.py)
__device__ __forceinline__ float td_dist2(float dx, float dy, float dz) {
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}
