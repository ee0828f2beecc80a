// Operand layout of v_mfma_f32_16x16x32_bf16 (gfx950), determined by experiment: D = A (16 x 32) . B (32 x 16) with one-hot
// operands.  Prints, for every lane and input slot j (0..7), which k index the slot feeds (A and B sides).
//   hipcc --offload-arch=gfx950 -O3 mfma_bf16_layout.hip -o /tmp/lay && /tmp/lay
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// A[row][k] = (k == ka && row == 0) ? 1 : 0 through lane/slot (la, ja); B[k][col] = code(k) for col 0 through every lane/slot
__global__ void probe(float *out) {
    const int lane = threadIdx.x;
    // for each lane group g and slot j of A: set that single A element to 1 (row = lane & 15 = 0, i.e. lane = 16 g), and load B
    // with B[k][col 0] = k + 1 encoded per (lane group, slot) hypothesis-free: B lane (col 0 -> lanes 0, 16, 32, 48) slot j gets
    // value 1 + 8 * (lane >> 4) + j.  D[0][0] then tells which B (group, slot) pairs with the A (group, slot).
    for (int g = 0; g < 4; ++g)
        for (int j = 0; j < 8; ++j) {
            unsigned short a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8];
            if (lane == 16 * g) a[j] = 0x3f80;                                   // 1.0 in bf16
            for (int t = 0; t < 8; ++t) {
                const float v = (lane & 15) == 0 ? (float)(1 + 8 * (lane >> 4) + t) : 0.f;
                unsigned u; memcpy(&u, &v, 4);
                b[t] = (unsigned short)(u >> 16);                                // small integers are exact in bf16
            }
            bf16x8 av, bv;
            memcpy(&av, a, 16); memcpy(&bv, b, 16);
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0);
            if (lane == 0) out[g * 8 + j] = acc[0];                              // D[row 0][col 0]
        }
}

int main() {
    float *d, h[32];
    (void)hipMalloc(&d, sizeof(h));
    probe<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("A element (lane group g, slot j) pairs with B element (group, slot):\n");
    for (int g = 0; g < 4; ++g) {
        for (int j = 0; j < 8; ++j) {
            const int code = (int)h[g * 8 + j] - 1;
            printf("  A(g=%d,j=%d)->B(g=%d,j=%d)", g, j, code / 8, code % 8);
        }
        printf("\n");
    }
    return 0;
}
