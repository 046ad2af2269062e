// Probe: verifies the lane maps assumed for v_mfma_f32_32x32x2_f32 / 16x16x4_f32 with exact integers.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k32(const float* A, const float* B, float* D) {   // A[32][2], B[2][32]
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * 2 + h], B[h * 32 + r], c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) { int row = (i & 3) + 8 * (i >> 2) + 4 * h; D[row * 32 + r] = c[i]; }
}
__global__ void k16(const float* A, const float* B, float* D) {   // A[16][4], B[4][16]
    int l = threadIdx.x, r = l & 15, q = l >> 4;
    f32x4 c = {0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * 4 + q], B[q * 16 + r], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * q + i) * 16 + r] = c[i];
}
int main() {
    float hA[64], hB[64], hD[1024], *A, *B, *D;
    for (int i = 0; i < 64; ++i) { hA[i] = (float)(i % 7 + 1); hB[i] = (float)((i * 5) % 11 - 3); }
    hipMalloc(&A, 256); hipMalloc(&B, 256); hipMalloc(&D, 4096);
    hipMemcpy(A, hA, 256, hipMemcpyHostToDevice); hipMemcpy(B, hB, 256, hipMemcpyHostToDevice);
    k32<<<1, 64>>>(A, B, D); hipMemcpy(hD, D, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float w = hA[i * 2] * hB[j] + hA[i * 2 + 1] * hB[32 + j];
        if (hD[i * 32 + j] != w) ++bad;
    }
    printf("32x32x2: %d mismatches\n", bad);
    k16<<<1, 64>>>(A, B, D); hipMemcpy(hD, D, 1024, hipMemcpyDeviceToHost);
    bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        float w = 0; for (int k = 0; k < 4; ++k) w += hA[i * 4 + k] * hB[k * 16 + j];
        if (hD[i * 16 + j] != w) ++bad;
    }
    printf("16x16x4: %d mismatches\n", bad);
    return 0;
}
