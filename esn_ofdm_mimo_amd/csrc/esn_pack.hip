// Weight packing: float64 row-major reference matrices (pyESN.py:93-109,191-192)
// -> device images in the exact order the recurrence kernels consume them.
#include "esn_common.h"

namespace esn {

// Wext[row][k] of the padded GEMM (see Geometry): W | W_in | W_fb with zero padding.
__device__ __forceinline__ double wext(const double* W, const double* Win, const double* Wfb,
                                       int n_res, int n_in, int n_out, int tf,
                                       int kin, int kfb, int row, int k) {
    if (row >= n_res) return 0.0;
    if (k < n_res) return W[(size_t)row * n_res + k];
    if (k >= kin && k < kin + n_in) return Win[(size_t)row * n_in + (k - kin)];
    if (tf && k >= kfb && k < kfb + n_out) return Wfb[(size_t)row * n_out + (k - kfb)];
    return 0.0;
}

// float64 image: K-major Wk[k][n_res], k over [W | W_in | W_fb] without padding.
__global__ void pack_w_f64_kernel(const double* W, const double* Win, const double* Wfb,
                                  int n_res, int n_in, int n_out, int tf, int n_wsets,
                                  size_t set_stride_bytes, char* out_base) {
    const size_t K = (size_t)n_res + n_in + n_out;
    const size_t per = K * n_res;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per * n_wsets;
         i += (size_t)gridDim.x * blockDim.x) {
        size_t ws = i / per, j = i % per;
        int k = (int)(j / n_res), r = (int)(j % n_res);
        const double* w = W + ws * (size_t)n_res * n_res;
        const double* wi = Win + ws * (size_t)n_res * n_in;
        const double* wf = Wfb + ws * (size_t)n_res * n_out;
        double v;
        if (k < n_res) v = w[(size_t)r * n_res + k];
        else if (k < n_res + n_in) v = wi[(size_t)r * n_in + (k - n_res)];
        else v = tf ? wf[(size_t)r * n_out + (k - n_res - n_in)] : 0.0;
        reinterpret_cast<double*>(out_base + ws * set_stride_bytes)[j] = v;
    }
}

// float64 MFMA image (esn_recur_f64_mfma.hip): [row tile of 16][64-byte k-group][lane][2 doubles];
// lane (r = lane & 15, q = lane >> 4) holds elements k = 8 kg + 2 q + {0, 1} of row 16 rt + r.
__global__ void pack_w_f64_mfma_kernel(const double* W, const double* Win, const double* Wfb,
                                       int n_res, int n_in, int n_out, int tf, int n_wsets,
                                       Geometry g, size_t set_stride_bytes, size_t off_bytes, char* out_base) {
    const int nkg = g.Kp / 8;
    const size_t per = (size_t)g.Mp * g.Kp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per * n_wsets;
         i += (size_t)gridDim.x * blockDim.x) {
        size_t ws = i / per, j = i % per;
        const int e = (int)(j & 1); j >>= 1;
        const int lane = (int)(j % 64); j /= 64;
        const int kg = (int)(j % nkg);
        const int rt = (int)(j / nkg);
        const int row = rt * 16 + (lane & 15);
        const int k = kg * 8 + 2 * (lane >> 4) + e;
        double* out = reinterpret_cast<double*>(out_base + ws * set_stride_bytes + off_bytes);
        out[i % per] = wext(W + ws * (size_t)n_res * n_res, Win + ws * (size_t)n_res * n_in,
                            Wfb + ws * (size_t)n_res * n_out, n_res, n_in, n_out, tf, g.kin, g.kfb, row, k);
    }
}

// float64 MFMA readout image per group: [64-byte k-group][lane][2 doubles], lane (o = lane & 15, q):
// W_out[o][k], k = 8 kg + 2 q + {0, 1} in the kernel's k layout (state | inputs at kin | zeros).
__global__ void pack_wout_f64_mfma_kernel(const double* Wout, int n_res, int n_in, int n_out, int n_groups,
                                          Geometry g, size_t stride_bytes, size_t off_bytes, char* out_base) {
    const int ncols = n_res + n_in;
    const size_t per = (size_t)16 * g.Kp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per * n_groups;
         i += (size_t)gridDim.x * blockDim.x) {
        size_t grp = i / per, j = i % per;
        const int e = (int)(j & 1); j >>= 1;
        const int lane = (int)(j % 64);
        const int kg = (int)(j / 64);
        const int o = lane & 15;
        const int k = kg * 8 + 2 * (lane >> 4) + e;
        double v = 0.0;
        if (o < n_out) {
            const double* wo = Wout + (grp * n_out + o) * ncols;
            if (k < n_res) v = wo[k];
            else if (k >= g.kin && k < g.kin + n_in) v = wo[n_res + (k - g.kin)];
        }
        reinterpret_cast<double*>(out_base + grp * stride_bytes + off_bytes)[i % per] = v;
    }
}

// MFMA image: [row tile rt][32-byte k-group kg][lane][16 bytes]; lane (r,h) holds
// elements k = (32 kg + 16 h)/ES + j of row 32 rt + r.
template <typename T>
__global__ void pack_w_mfma_kernel(const double* W, const double* Win, const double* Wfb,
                                   int n_res, int n_in, int n_out, int tf, int n_wsets,
                                   Geometry g, size_t set_stride_bytes, char* out_base) {
    constexpr int ES = sizeof(T);
    constexpr int EPL = 16 / ES;                       // elements per lane per group
    const int nkg = g.Kp * ES / 32;
    const size_t per = (size_t)(g.Mp / 32) * nkg * 64 * EPL;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per * n_wsets;
         i += (size_t)gridDim.x * blockDim.x) {
        size_t ws = i / per, j = i % per;
        int e = (int)(j % EPL); j /= EPL;
        int lane = (int)(j % 64); j /= 64;
        int kg = (int)(j % nkg);
        int rt = (int)(j / nkg);
        int row = rt * 32 + (lane & 31);
        int k = (kg * 32 + 16 * (lane >> 5)) / ES + e;
        double v = wext(W + ws * (size_t)n_res * n_res, Win + ws * (size_t)n_res * n_in,
                        Wfb + ws * (size_t)n_res * n_out, n_res, n_in, n_out, tf, g.kin, g.kfb, row, k);
        // fp16/bf16 kernels evaluate tanh from 2^z: fold 2 log2(e) into the weights (esn_common.h)
        if (ES == 2) v *= ACT_PRESCALE;
        reinterpret_cast<T*>(out_base + ws * set_stride_bytes)[i % per] = (T)(float)v;
    }
}

// Image of the 16x16x32 skewed predict kernel (esn_recur_skew16_impl.h), behind the 32x32x16 image of each set:
// fragment (wave w, 32-k group kk, row tile m) at ((w NKK + kk) 4 + m) KB -- the 4 KB a wave needs per group are
// contiguous; lane (r = lane & 15, q = lane >> 4) holds W[64 w + 16 m + r][s16_nat(32 kk + 8 q + e)], e = 0..7.
template <typename T>
__global__ void pack_w_s16_kernel(const double* W, const double* Win, const double* Wfb,
                                  int n_res, int n_in, int n_out, int tf, int n_wsets,
                                  Geometry g, size_t set_stride_bytes, size_t off_bytes, char* out_base) {
    const int nkk = g.Kp / 32;
    const size_t per = (size_t)g.Mp * g.Kp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per * n_wsets;
         i += (size_t)gridDim.x * blockDim.x) {
        size_t ws = i / per, j = i % per;
        const int e = (int)(j & 7); j >>= 3;
        const int lane = (int)(j & 63); j >>= 6;
        const int m = (int)(j & 3); j >>= 2;
        const int kk = (int)(j % nkk);
        const int w = (int)(j / nkk);
        const int row = 64 * w + 16 * m + (lane & 15);
        const int k = s16_nat(32 * kk + 8 * (lane >> 4) + e);
        double v = wext(W + ws * (size_t)n_res * n_res, Win + ws * (size_t)n_res * n_in,
                        Wfb + ws * (size_t)n_res * n_out, n_res, n_in, n_out, tf, g.kin, g.kfb, row, k);
        v *= ACT_PRESCALE;
        reinterpret_cast<T*>(out_base + ws * set_stride_bytes + off_bytes)[i % per] = (T)(float)v;
    }
}

// bytes of the persistent kernels' read-out image (the big path's image, when present, follows 16-byte aligned)
__host__ __device__ inline size_t packed_wout_persistent_bytes(int es, int n_out, const Geometry& g) {
    const int n_ot = (n_out + 15) / 16;
    return (size_t)g.ro_parts * n_ot * 16 * g.Kp * es + 16;
}
size_t big_wout_image_bytes(int Mp);
// (defined in esn_recur_rs.hip in ESN_WITH_RS=1 builds; g.rs is 0 otherwise, so this is never called)
static inline size_t rs_wout_image_bytes(int Kp) { return (size_t)(Kp / 16) * 1024 + 16; }

// Readout image for the 16x16 MFMA: [part][ot][64-byte k-group][lane][16 B] then a
// 16-byte trailer {1/gain, gain, 0, 0} (float).  gain is a power of two that brings
// max|W_out| of the group to ~2^10 so fp16 images keep full precision; part 1 holds
// the rounding residual of part 0 (fp16/bf16 only).
template <typename T>
__global__ __launch_bounds__(256) void pack_wout_mfma_kernel(const double* Wout, int n_res, int n_in,
                                                              int n_out, Geometry g, size_t stride_bytes,
                                                              char* out_base) {
    constexpr int ES = sizeof(T);
    constexpr int EPL = 16 / ES;
    __shared__ double red[256];
    const int grp = blockIdx.x;
    const int ncols = n_res + n_in;
    const double* wo = Wout + (size_t)grp * n_out * ncols;
    double m = 0.0;
    for (int i = threadIdx.x; i < n_out * ncols; i += blockDim.x) m = fmax(m, fabs(wo[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    m = red[0];
    double gain = 1.0;
    if (ES == 2 && m > 0.0 && isfinite(m)) {
        int ex;
        frexp(m, &ex);                 // m = f * 2^ex, f in [0.5,1)
        gain = ldexp(1.0, 10 - ex);    // m*gain in [2^9, 2^10)
    }
    const int n_ot = (n_out + 15) / 16;
    const int nkg64 = g.Kp * ES / 64;
    char* out = out_base + (size_t)grp * stride_bytes;
    T* img = reinterpret_cast<T*>(out);
    const size_t per_part = (size_t)n_ot * nkg64 * 64 * EPL;
    for (size_t i = threadIdx.x; i < per_part; i += blockDim.x) {
        size_t j = i;
        int e = (int)(j % EPL); j /= EPL;
        int lane = (int)(j % 64); j /= 64;
        int kg = (int)(j % nkg64);
        int ot = (int)(j / nkg64);
        int o = ot * 16 + (lane & 15);
        int k = (kg * 64 + 16 * (lane >> 4)) / ES + e;
        const bool lo_row = g.ro_fold && o >= 8;        // folded image: rows 8..15 carry the residual
        if (lo_row) o -= 8;
        double v = 0.0;
        if (o < n_out) {
            if (k < n_res) v = wo[(size_t)o * ncols + k];
            else if (k >= g.kin && k < g.kin + n_in) v = wo[(size_t)o * ncols + n_res + (k - g.kin)];
        }
        v *= gain;
        T hi = (T)(float)v;
        T lo = (T)(float)(v - (double)(float)hi);
        img[i] = lo_row ? lo : hi;
        if (g.ro_parts == 2) img[per_part + i] = lo;
    }
    if (threadIdx.x == 0) {
        float* tr = reinterpret_cast<float*>(out + g.ro_parts * per_part * ES);
        tr[0] = (float)(1.0 / gain); tr[1] = (float)gain; tr[2] = 0.f; tr[3] = 0.f;
    }
    if (ES == 2 && g.s16) {
        // image of the 16x16x32 skewed kernel: [32-k group kk][lane (o = lane & 15, q)][8 elements] in ITS k order
        // (s16_nat), rows 0-7 hi, 8-15 the rounding residual; trailer {1/gain, gain, 0, 0}
        char* sb = out + (packed_wout_persistent_bytes(ES, n_out, g) + 15) / 16 * 16;
        T* simg = reinterpret_cast<T*>(sb);
        const int nkk = g.Kp / 32;
        for (int i = threadIdx.x; i < nkk * 512; i += blockDim.x) {
            const int e = i & 7, lane = (i >> 3) & 63, kk = i >> 9;
            const int o = lane & 15, oo = o & 7;
            const int k = s16_nat(32 * kk + 8 * (lane >> 4) + e);
            double v = 0.0;
            if (oo < n_out) {
                if (k < n_res) v = wo[(size_t)oo * ncols + k] * gain;
                else if (k >= g.kin && k < g.kin + n_in) v = wo[(size_t)oo * ncols + n_res + (k - g.kin)] * gain;
            }
            const T hi = (T)(float)v;
            const T lo = (T)(float)(v - (double)(float)hi);
            simg[i] = o < 8 ? hi : lo;
        }
        if (threadIdx.x == 0) {
            float* tr = reinterpret_cast<float*>(sb + (size_t)nkk * 1024);
            tr[0] = (float)(1.0 / gain); tr[1] = (float)gain; tr[2] = 0.f; tr[3] = 0.f;
        }
    }
    if (ES == 2 && g.rs) {
        // image of the register-resident-state kernel (esn_recur_rs.hip): A operand of the 32x32x16 MFMA,
        // [k-group of 16][lane (row = lane & 31, h)][8 elements], k = 16 kg + 8 h + e natural; rows 0-7 hi, 8-15 lo
        char* rsb = out + (packed_wout_persistent_bytes(ES, n_out, g) + 15) / 16 * 16;
        T* rimg = reinterpret_cast<T*>(rsb);
        const int nkg16 = g.Kp / 16;
        for (int i = threadIdx.x; i < nkg16 * 512; i += blockDim.x) {
            const int e = i & 7, lane = (i >> 3) & 63, kg = i >> 9;
            const int o = lane & 31, k = 16 * kg + 8 * (lane >> 5) + e;
            const int oo = o & 7;
            double v = 0.0;
            if (o < 16 && oo < n_out) {
                if (k < n_res) v = wo[(size_t)oo * ncols + k] * gain;
                else if (k >= g.kin && k < g.kin + n_in) v = wo[(size_t)oo * ncols + n_res + (k - g.kin)] * gain;
            }
            const T hi = (T)(float)v;
            const T lo = (T)(float)(v - (double)(float)hi);
            rimg[i] = o < 8 ? hi : lo;
        }
        if (threadIdx.x == 0) {
            float* tr = reinterpret_cast<float*>(rsb + (size_t)nkg16 * 1024);
            tr[0] = (float)(1.0 / gain); tr[1] = (float)gain; tr[2] = 0.f; tr[3] = 0.f;
        }
    }
    if (ES == 2 && g.big) {
        // image of the launch-per-step GEMM path (esn_recur_big.hip): A operand of a 32x32x16 MFMA whose B
        // operand is an accumulator tile -- [row tile of 32 k][k-step s2][lane (o = lane & 31, h)][8 elements],
        // element e <-> k = 32 rt + 16 s2 + 8 (e >> 2) + 4 h + (e & 3); rows 0-7 hi, 8-15 rounding residual
        char* big = out + (packed_wout_persistent_bytes(ES, n_out, g) + 15) / 16 * 16;
        T* bimg = reinterpret_cast<T*>(big);
        const size_t nb = (size_t)g.Mp * 32;
        for (size_t i = threadIdx.x; i < nb; i += blockDim.x) {
            const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), s2 = (int)((i >> 9) & 1), rt = (int)(i >> 10);
            const int o = lane & 31, hA = lane >> 5;
            const int k = 32 * rt + 16 * s2 + 8 * (e >> 2) + 4 * hA + (e & 3);
            const int oo = o & 7;
            double v = (o < 16 && oo < n_out && k < n_res) ? wo[(size_t)oo * ncols + k] * gain : 0.0;
            const T hi = (T)(float)v;
            const T lo = (T)(float)(v - (double)(float)hi);
            bimg[i] = o < 8 ? hi : lo;
        }
        float* wu = reinterpret_cast<float*>(big + (size_t)g.Mp * 64);
        for (int i = threadIdx.x; i < 128; i += blockDim.x) {
            const int o = i >> 4, ii = i & 15;
            wu[i] = (o < n_out && ii < n_in) ? (float)(wo[(size_t)o * ncols + n_res + ii] * gain) : 0.f;
        }
        if (threadIdx.x == 0) { wu[128] = (float)(1.0 / gain); wu[129] = (float)gain; wu[130] = 0.f; wu[131] = 0.f; }
    }
}

// ESN_F64 images hold the vector-ALU kernel's copy first and, when the matrix-pipe kernel fits the
// shape (g.m64), its fragment-ordered copy behind it (16-byte aligned offset).
size_t f64_w_offset(int n_res, int n_in, int n_out) {
    return (sizeof(double) * (size_t)(n_res + n_in + n_out) * n_res + 15) / 16 * 16;
}
size_t f64_wout_offset(int n_res, int n_in, int n_out) {
    return (sizeof(double) * (size_t)n_out * (n_res + n_in) + 15) / 16 * 16;
}

size_t packed_w_bytes(int precision, int n_res, int n_in, int n_out, const Geometry& g) {
    if (precision == ESN_F64)
        return f64_w_offset(n_res, n_in, n_out) + (g.m64 ? sizeof(double) * (size_t)g.Mp * g.Kp : 0);
    const int es = (precision == ESN_F32) ? 4 : 2;
    return (size_t)g.Mp * g.Kp * es * (g.s16 ? 2 : 1);        // (+ the 16x16x32 kernel's copy)
}

size_t packed_wout_bytes(int precision, int n_res, int n_in, int n_out, const Geometry& g) {
    if (precision == ESN_F64)
        return f64_wout_offset(n_res, n_in, n_out) + (g.m64 ? sizeof(double) * (size_t)16 * g.Kp : 0);
    const int es = (precision == ESN_F32) ? 4 : 2;
    const size_t base = packed_wout_persistent_bytes(es, n_out, g);
    if (g.rs) return (base + 15) / 16 * 16 + rs_wout_image_bytes(g.Kp);
    if (g.s16) return (base + 15) / 16 * 16 + (size_t)(g.Kp / 32) * 1024 + 16;
    return g.big ? (base + 15) / 16 * 16 + big_wout_image_bytes(g.Mp) : base;
}
size_t wout_big_offset(int precision, int n_out, const Geometry& g) {
    return (packed_wout_persistent_bytes(precision == ESN_F32 ? 4 : 2, n_out, g) + 15) / 16 * 16;
}

int launch_pack_weights(int precision, const esn_shape_t* sh, const Geometry& g, const double* W,
                        const double* Win, const double* Wfb, void* packed, hipStream_t stream) {
    const int blocks = 1024, threads = 256;
    const size_t mstride = packed_w_bytes(precision, sh->n_res, sh->n_in, sh->n_out, g);
    if (precision == ESN_F64) {
        const size_t stride = packed_w_bytes(precision, sh->n_res, sh->n_in, sh->n_out, g);
        hipLaunchKernelGGL(pack_w_f64_kernel, dim3(blocks), dim3(threads), 0, stream, W, Win, Wfb,
                           sh->n_res, sh->n_in, sh->n_out, sh->teacher_forcing, sh->n_wsets,
                           stride, reinterpret_cast<char*>(packed));
        if (g.m64)
            hipLaunchKernelGGL(pack_w_f64_mfma_kernel, dim3(blocks), dim3(threads), 0, stream, W, Win, Wfb,
                               sh->n_res, sh->n_in, sh->n_out, sh->teacher_forcing, sh->n_wsets, g, stride,
                               f64_w_offset(sh->n_res, sh->n_in, sh->n_out), reinterpret_cast<char*>(packed));
    } else if (precision == ESN_F32) {
        hipLaunchKernelGGL(pack_w_mfma_kernel<float>, dim3(blocks), dim3(threads), 0, stream, W, Win, Wfb,
                           sh->n_res, sh->n_in, sh->n_out, sh->teacher_forcing, sh->n_wsets, g, mstride,
                           reinterpret_cast<char*>(packed));
    } else if (precision == ESN_F16) {
        hipLaunchKernelGGL(pack_w_mfma_kernel<_Float16>, dim3(blocks), dim3(threads), 0, stream, W, Win, Wfb,
                           sh->n_res, sh->n_in, sh->n_out, sh->teacher_forcing, sh->n_wsets, g, mstride,
                           reinterpret_cast<char*>(packed));
        if (g.s16)
            hipLaunchKernelGGL(pack_w_s16_kernel<_Float16>, dim3(blocks), dim3(threads), 0, stream, W, Win, Wfb,
                               sh->n_res, sh->n_in, sh->n_out, sh->teacher_forcing, sh->n_wsets, g, mstride,
                               (size_t)g.Mp * g.Kp * 2, reinterpret_cast<char*>(packed));
    } else if (precision == ESN_BF16) {
        hipLaunchKernelGGL(pack_w_mfma_kernel<__bf16>, dim3(blocks), dim3(threads), 0, stream, W, Win, Wfb,
                           sh->n_res, sh->n_in, sh->n_out, sh->teacher_forcing, sh->n_wsets, g, mstride,
                           reinterpret_cast<char*>(packed));
        if (g.s16)
            hipLaunchKernelGGL(pack_w_s16_kernel<__bf16>, dim3(blocks), dim3(threads), 0, stream, W, Win, Wfb,
                               sh->n_res, sh->n_in, sh->n_out, sh->teacher_forcing, sh->n_wsets, g, mstride,
                               (size_t)g.Mp * g.Kp * 2, reinterpret_cast<char*>(packed));
    } else {
        return -1;
    }
    return (int)hipGetLastError();
}

int launch_pack_readout(int precision, const esn_shape_t* sh, const Geometry& g, int n_groups,
                        const double* Wout, void* packed, hipStream_t stream) {
    const size_t stride = packed_wout_bytes(precision, sh->n_res, sh->n_in, sh->n_out, g);
    if (precision == ESN_F64) {
        const size_t plain = sizeof(double) * (size_t)sh->n_out * (sh->n_res + sh->n_in);
        hipError_t e = hipMemcpy2DAsync(packed, stride, Wout, plain, plain, n_groups, hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return (int)e;
        if (g.m64)
            hipLaunchKernelGGL(pack_wout_f64_mfma_kernel, dim3(n_groups < 1024 ? n_groups : 1024), dim3(256), 0, stream,
                               Wout, sh->n_res, sh->n_in, sh->n_out, n_groups, g, stride,
                               f64_wout_offset(sh->n_res, sh->n_in, sh->n_out), reinterpret_cast<char*>(packed));
        return (int)hipGetLastError();
    }
    char* out = reinterpret_cast<char*>(packed);
    if (precision == ESN_F32)
        hipLaunchKernelGGL((pack_wout_mfma_kernel<float>), dim3(n_groups), dim3(256), 0, stream, Wout,
                           sh->n_res, sh->n_in, sh->n_out, g, stride, out);
    else if (precision == ESN_F16)
        hipLaunchKernelGGL((pack_wout_mfma_kernel<_Float16>), dim3(n_groups), dim3(256), 0, stream, Wout,
                           sh->n_res, sh->n_in, sh->n_out, g, stride, out);
    else if (precision == ESN_BF16)
        hipLaunchKernelGGL((pack_wout_mfma_kernel<__bf16>), dim3(n_groups), dim3(256), 0, stream, Wout,
                           sh->n_res, sh->n_in, sh->n_out, g, stride, out);
    else
        return -1;
    return (int)hipGetLastError();
}

}  // namespace esn
