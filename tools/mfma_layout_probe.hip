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
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k16d(const double* A, const double* B, double* D) {   // A[16][4], B[4][16], row = 4i + q
    int l = threadIdx.x, r = l & 15, q = l >> 4;
    f64x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[r * 4 + q], B[q * 16 + r], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * i + q) * 16 + r] = c[i];
}
__global__ void k16d_rate(double* D, long long* cyc, int n) {          // back-to-back f64 MFMAs, one wave
    f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = 1.0 + threadIdx.x, b = 2.0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    D[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
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
    {
        double hAd[64], hBd[64], hDd[256], *Ad, *Bd, *Dd; long long* cyc; long long hc;
        for (int i = 0; i < 64; ++i) { hAd[i] = (double)(i % 7 + 1); hBd[i] = (double)((i * 5) % 11 - 3); }
        hipMalloc(&Ad, 512); hipMalloc(&Bd, 512); hipMalloc(&Dd, 2048); hipMalloc(&cyc, 8);
        hipMemcpy(Ad, hAd, 512, hipMemcpyHostToDevice); hipMemcpy(Bd, hBd, 512, hipMemcpyHostToDevice);
        k16d<<<1, 64>>>(Ad, Bd, Dd); hipMemcpy(hDd, Dd, 2048, hipMemcpyDeviceToHost);
        bad = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double w = 0; for (int k = 0; k < 4; ++k) w += hAd[i * 4 + k] * hBd[k * 16 + j];
            if (hDd[i * 16 + j] != w) ++bad;
        }
        printf("f64 16x16x4 (row = 4*i + lane/16, col = lane%%16): %d mismatches\n", bad);
        k16d_rate<<<1, 64>>>(Dd, cyc, 1000); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        printf("f64 16x16x4 back-to-back: %.1f ticks of the cycle counter per MFMA (counter may run at 100 MHz)\n", hc / 4000.0);
    }
    return 0;
}
