// float64 recurrence kernel: reference arithmetic (pyESN.py:111-125,176-182,243-255)
// on the vector ALU.  One workgroup advances FB sequences for all S steps with
// the state double-buffered in LDS; thread t owns reservoir rows t, t+blockDim, ...
// Weights are read K-major (Wk[k][n_res]) so a wave's loads are coalesced and
// the LDS operand Z[f][k] is a wave-wide broadcast.
#include "esn_common.h"

namespace esn {

template <int FB>
__global__ __launch_bounds__(1024) void recur_f64_kernel(RecurParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* smem = reinterpret_cast<double*>(smem_raw);
    const int n_res = p.n_res, n_in = p.n_in, n_out = p.n_out;
    const int nio = n_in + n_out;
    double* X0 = smem;                       // [FB][n_res]
    double* X1 = X0 + FB * n_res;            // [FB][n_res]
    double* Uc = X1 + FB * n_res;            // [FB][n_in]   scaled inputs of this step
    double* Fc = Uc + FB * n_in;             // [FB][n_out]  fed-back output / teacher
    double* Yn = Fc + FB * n_out;            // [FB][n_out]  readout of this step

    const int tid = threadIdx.x, nth = blockDim.x;
    const int slot0 = blockIdx.x * FB;
    // per-slot frame / group tables (LDS, FB entries)
    __shared__ int s_fr[FB], s_grp[FB];
    if (tid < FB) {
        int grp;
        s_fr[tid] = slot_frame(p, slot0 + tid, grp);
        s_grp[tid] = grp;
    }
    __syncthreads();
    int grp0 = slot_group(p, slot0);
    if (grp0 >= p.n_groups) return;
    const int wset = slot_wset(p, slot0);
    const double* Wk = reinterpret_cast<const double*>(
        reinterpret_cast<const char*>(p.packed_w) + (size_t)wset * p.wset_stride);
    const int ncols = n_res + n_in;
    const int out_rows = p.S - p.transient;

    // ---- initial state, first input row, initial feedback -------------------
    for (int i = tid; i < FB * n_res; i += nth) {
        int f = i / n_res, r = i % n_res;
        double v = 0.0;
        if (p.x0 && s_fr[f] >= 0) v = p.x0[(size_t)s_grp[f] * n_res + r];
        X0[i] = v;
    }
    auto stage_io = [&](int s) {
        // inputs for step s (row s + in_row_off), feedback for step s when harvesting
        const int row = s + p.in_row_off;
        for (int i = tid; i < FB * n_in; i += nth) {
            int f = i / n_in, c = i % n_in;
            double v = 0.0;
            const int fr = s_fr[f];
            if (fr >= 0) {
                const int pg = s_grp[f];
                double raw = (row < p.T_in) ? p.U[((size_t)fr * p.T_in + row) * n_in + c] : 0.0;
                double sc = p.in_scale ? p.in_scale[(size_t)pg * n_in + c] : 1.0;
                double sh = p.in_shift ? p.in_shift[(size_t)pg * n_in + c] : 0.0;
                v = raw * sc + sh;
            }
            Uc[i] = v;
        }
        if (p.harvest) {
            for (int i = tid; i < FB * n_out; i += nth) {
                int f = i / n_out, c = i % n_out;
                double v = 0.0;
                const int fr = s_fr[f];
                if (fr >= 0) {
                    const int pg = s_grp[f];
                    double raw = p.D[((size_t)fr * (p.S + 1) + s) * n_out + c];
                    double sc = p.t_scale ? p.t_scale[(size_t)pg * n_out + c] : 1.0;
                    double sh = p.t_shift ? p.t_shift[(size_t)pg * n_out + c] : 0.0;
                    v = raw * sc + sh;
                }
                Fc[i] = v;
            }
        }
    };
    if (!p.harvest) {
        for (int i = tid; i < FB * n_out; i += nth) {
            int f = i / n_out, c = i % n_out;
            double v = 0.0;
            if (p.y0 && s_fr[f] >= 0) v = p.y0[(size_t)s_grp[f] * n_out + c];
            Fc[i] = v;
        }
    } else {
        // E row 0 = [0, scale(u[0])]  (pyESN.py:179,189)
        for (int i = tid; i < FB * ncols; i += nth) {
            int f = i / ncols, c = i % ncols;
            const int fr = s_fr[f];
            if (fr < 0) continue;
            const int pg = s_grp[f];
            double v = 0.0;
            if (c >= n_res) {
                int ci = c - n_res;
                double raw = p.U[((size_t)fr * p.T_in) * n_in + ci];
                double sc = p.in_scale ? p.in_scale[(size_t)pg * n_in + ci] : 1.0;
                double sh = p.in_shift ? p.in_shift[(size_t)pg * n_in + ci] : 0.0;
                v = raw * sc + sh;
            }
            p.E[((size_t)fr * (p.S + 1)) * ncols + c] = v;
        }
    }
    stage_io(0);
    __syncthreads();

    double* Xc = X0;
    double* Xn = X1;
    const int kfb_end = p.teacher_forcing ? nio : n_in;

    for (int s = 0; s < p.S; ++s) {
        // ---- x_{s+1} = tanh(W x_s + W_in u + W_fb f) + noise ------------------
        for (int r0 = tid; r0 < n_res; r0 += 2 * nth) {
            const int r1 = r0 + nth;
            const bool has1 = r1 < n_res;
            double a0[FB], a1[FB];
#pragma unroll
            for (int f = 0; f < FB; ++f) { a0[f] = 0.0; a1[f] = 0.0; }
            const double* w = Wk;
#pragma unroll 8
            for (int k = 0; k < n_res; ++k, w += n_res) {
                double w0 = w[r0];
                double w1 = has1 ? w[r1] : 0.0;
#pragma unroll
                for (int f = 0; f < FB; ++f) {
                    double z = Xc[f * n_res + k];
                    a0[f] = fma(w0, z, a0[f]);
                    a1[f] = fma(w1, z, a1[f]);
                }
            }
            for (int k = 0; k < kfb_end; ++k, w += n_res) {
                double w0 = w[r0];
                double w1 = has1 ? w[r1] : 0.0;
#pragma unroll
                for (int f = 0; f < FB; ++f) {
                    double z = (k < n_in) ? Uc[f * n_in + k] : Fc[f * n_out + (k - n_in)];
                    a0[f] = fma(w0, z, a0[f]);
                    a1[f] = fma(w1, z, a1[f]);
                }
            }
#pragma unroll
            for (int f = 0; f < FB; ++f) {
                double x0v = tanh(a0[f]);
                double x1v = tanh(a1[f]);
                if (p.leak != 1.0) {                       // leaky integration (extension; the reference has a == 1)
                    const double xo0 = Xc[f * n_res + r0], xo1 = has1 ? Xc[f * n_res + r1] : 0.0;
                    x0v = fma(p.leak, x0v - xo0, xo0);
                    x1v = fma(p.leak, x1v - xo1, xo1);
                }
                if (p.noise_mode != ESN_NOISE_NONE && s_fr[f] >= 0) {
                    const uint32_t fr = (uint32_t)s_fr[f];
                    double u0, u1 = 0.0;
                    if (p.noise_mode == ESN_NOISE_TENSOR) {
                        const double* nz = p.noise_u + ((size_t)fr * p.S + s) * n_res;
                        u0 = nz[r0];
                        if (has1) u1 = nz[r1];
                    } else {
                        uint32_t key = noise_key(p.seed, (uint32_t)fr + p.frame_off, (uint32_t)s);
                        u0 = noise_uniform(key, r0);
                        u1 = noise_uniform(key, r1);
                    }
                    x0v += p.noise * (u0 - 0.5);
                    x1v += p.noise * (u1 - 0.5);
                }
                Xn[f * n_res + r0] = x0v;
                if (has1) Xn[f * n_res + r1] = x1v;
            }
        }
        __syncthreads();

        if (p.harvest) {
            // E row s+1 = [x_{s+1}, u_scaled[s+1]]
            for (int i = tid; i < FB * ncols; i += nth) {
                int f = i / ncols, c = i % ncols;
                if (s_fr[f] < 0) continue;
                double v = (c < n_res) ? Xn[f * n_res + c] : Uc[f * n_in + (c - n_res)];
                p.E[((size_t)s_fr[f] * (p.S + 1) + (s + 1)) * ncols + c] = v;
            }
        } else {
            // ---- y = W_out [x_{s+1}; u]  (one wave per (output, frame) pair) ---
            const int lane = tid & 63, wv = tid >> 6, nwv = nth >> 6;
            for (int pr = wv; pr < FB * n_out; pr += nwv) {
                int f = pr / n_out, o = pr % n_out;
                double acc = 0.0;
                if (s_fr[f] >= 0) {
                    int pg = s_grp[f];
                    const double* wo = reinterpret_cast<const double*>(
                        reinterpret_cast<const char*>(p.packed_wout) + (size_t)pg * p.wout_stride)
                        + (size_t)o * ncols;
                    for (int k = lane; k < ncols; k += 64) {
                        double z = (k < n_res) ? Xn[f * n_res + k] : Uc[f * n_in + (k - n_res)];
                        acc = fma(wo[k], z, acc);
                    }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
                if (lane == 0) Yn[pr] = acc;
            }
        }
        __syncthreads();

        if (!p.harvest) {
            for (int i = tid; i < FB * n_out; i += nth) {
                int f = i / n_out, c = i % n_out;
                double y = Yn[i];
                Fc[i] = y;
                if (s_fr[f] >= 0 && s >= p.transient) {
                    int fr = s_fr[f], pg = s_grp[f];
                    double sc = p.t_scale ? p.t_scale[(size_t)pg * n_out + c] : 1.0;
                    double sh = p.t_shift ? p.t_shift[(size_t)pg * n_out + c] : 0.0;
                    p.Y[((size_t)fr * out_rows + (s - p.transient)) * n_out + c] = (y - sh) / sc;
                }
            }
        }
        if (s + 1 < p.S) stage_io(s + 1);
        __syncthreads();
        double* t = Xc; Xc = Xn; Xn = t;
    }
}

size_t recur_f64_lds_bytes(int FB, int n_res, int n_in, int n_out) {
    return sizeof(double) * (size_t)FB * (2 * n_res + n_in + 2 * n_out);
}

int launch_recur_f64(const RecurParams& p, hipStream_t stream) {
    // frames per tile: 8 while the double-buffered state fits in 150 KB, else 4/2/1
    const int FB = p.g.Bt;
    size_t lds = recur_f64_lds_bytes(FB, p.n_res, p.n_in, p.n_out);
    // two rows per thread; one row per thread for the small tiles (latency-bound: more waves, more loads in flight)
    int threads = round_up(FB <= 2 ? p.n_res : (p.n_res + 1) / 2, 64);
    if (threads > 1024) threads = 1024;
    if (threads < 64) threads = 64;
    dim3 grid(p.n_tiles), block(threads);
    hipError_t e;
#define ESN_LAUNCH_F64(FBV)                                                                  \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(recur_f64_kernel<FBV>),           \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
    if (e != hipSuccess) return (int)e;                                                      \
    hipLaunchKernelGGL(recur_f64_kernel<FBV>, grid, block, lds, stream, p);
    switch (FB) {
        case 8: ESN_LAUNCH_F64(8); break;
        case 4: ESN_LAUNCH_F64(4); break;
        case 2: ESN_LAUNCH_F64(2); break;
        case 1: ESN_LAUNCH_F64(1); break;
        default: return -1;
    }
#undef ESN_LAUNCH_F64
    return (int)hipGetLastError();
}

}  // namespace esn
