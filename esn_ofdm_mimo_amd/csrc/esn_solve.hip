// Readout training: W_out = (pinv(E[transient:]) @ teacher[transient:]).T  (pyESN.py:191-192)
// as a float64 Householder QR, one workgroup per trained ESN.
//
//   rows <  cols (4x8, N=128: 128 x 528): QR of A^T, minimum-norm solution
//                X = Q R^-T B   -- what pinv returns for an under-determined system
//   rows >= cols (SISO / 2x2: 512 x 104):           QR of [A | B], X = R^-1 (Q^T B)
//
// The working matrix is column-major in the caller's workspace (L2 resident);
// reflector j is applied to the trailing columns one wave per column.
#include "esn_common.h"

namespace esn {

struct SolveParams {
    const double* E; const double* D;
    int n_groups, T, transient, cols, n_out;
    const double* t_scale; const double* t_shift;
    double* W_out; int* status;
    double* work; size_t work_stride;   // doubles per group
    int m, n, wide;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return __shfl(v, 0);
}

__global__ __launch_bounds__(1024) void readout_qr_kernel(SolveParams sp) {
    __shared__ double red[16];
    __shared__ double bc[4];
    const int g = blockIdx.x;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int lane = tid & 63, wv = tid >> 6, nwv = nth >> 6;
    const int m = sp.m, n = sp.n, nrhs = sp.n_out;
    const int rows = sp.T - sp.transient, cols = sp.cols;
    double* M = sp.work + (size_t)g * sp.work_stride;       // [n + (wide?0:nrhs)][m] column-major
    const int ncol_tot = sp.wide ? n : n + nrhs;
    double* R = M + (size_t)ncol_tot * m;                    // rhs / solution block [nrhs][m]
    double* rdiag = R + (size_t)nrhs * m;                    // [n]
    double* beta = rdiag + n;                                // [n]
    const double* Eg = sp.E + ((size_t)g * sp.T + sp.transient) * cols;
    const double* Dg = sp.D + ((size_t)g * sp.T + sp.transient) * nrhs;

    // ---- load -----------------------------------------------------------------
    if (sp.wide) {
        // M = A^T: column j = row j of A (contiguous in E)
        for (size_t i = tid; i < (size_t)n * m; i += nth) M[i] = Eg[i];
        for (int i = tid; i < nrhs * m; i += nth) {
            int o = i / m, j = i % m;
            double v = 0.0;
            if (j < n) {
                double sc = sp.t_scale ? sp.t_scale[(size_t)g * nrhs + o] : 1.0;
                double sh = sp.t_shift ? sp.t_shift[(size_t)g * nrhs + o] : 0.0;
                v = Dg[(size_t)j * nrhs + o] * sc + sh;
            }
            R[i] = v;
        }
    } else {
        // M = [A | B]: column c of A is strided in E
        for (size_t i = tid; i < (size_t)n * m; i += nth) {
            int c = (int)(i / m), rr = (int)(i % m);
            M[i] = Eg[(size_t)rr * cols + c];
        }
        for (int i = tid; i < nrhs * m; i += nth) {
            int o = i / m, rr = i % m;
            double sc = sp.t_scale ? sp.t_scale[(size_t)g * nrhs + o] : 1.0;
            double sh = sp.t_shift ? sp.t_shift[(size_t)g * nrhs + o] : 0.0;
            M[(size_t)(n + o) * m + rr] = Dg[(size_t)rr * nrhs + o] * sc + sh;
        }
    }
    __syncthreads();

    // ---- Householder QR of the first n columns ----------------------------------
    for (int j = 0; j < n; ++j) {
        double* cj = M + (size_t)j * m;
        double part = 0.0;
        for (int i = j + tid; i < m; i += nth) { double x = cj[i]; part = fma(x, x, part); }
        part = wave_sum(part);
        if (lane == 0) red[wv] = part;
        __syncthreads();
        if (tid == 0) {
            double sigma = 0.0;
            for (int w = 0; w < nwv; ++w) sigma += red[w];
            double x0 = cj[j];
            double nx = sqrt(sigma);
            double alpha = (x0 >= 0.0) ? -nx : nx;
            double v0 = x0 - alpha;
            double vtv = sigma - x0 * x0 + v0 * v0;
            double b = (vtv > 0.0) ? 2.0 / vtv : 0.0;
            cj[j] = v0;
            rdiag[j] = alpha;
            beta[j] = b;
            bc[0] = b;
        }
        __syncthreads();
        const double b = bc[0];
        if (b != 0.0) {
            for (int c = j + 1 + wv; c < ncol_tot; c += nwv) {
                double* cc = M + (size_t)c * m;
                double dot = 0.0;
                for (int i = j + lane; i < m; i += 64) dot = fma(cj[i], cc[i], dot);
                dot = wave_sum(dot) * b;
                for (int i = j + lane; i < m; i += 64) cc[i] = fma(-dot, cj[i], cc[i]);
            }
        }
        __syncthreads();
    }

    // ---- rank check ---------------------------------------------------------------
    double rmax = 0.0;
    for (int j = tid; j < n; j += nth) rmax = fmax(rmax, fabs(rdiag[j]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rmax = fmax(rmax, __shfl_down(rmax, off));
    if (lane == 0) red[wv] = rmax;
    __syncthreads();
    if (tid == 0) {
        double r = 0.0;
        for (int w = 0; w < nwv; ++w) r = fmax(r, red[w]);
        bc[1] = r * 1e-13;
        int bad = 0;
        for (int j = 0; j < n; ++j) bad |= (fabs(rdiag[j]) <= r * 1e-13);
        sp.status[g] = bad ? 1 : 0;
    }
    __syncthreads();
    const double tol = bc[1];

    if (sp.wide) {
        // forward substitution R^T z = b, one wave per right-hand side (z overwrites b)
        for (int o = wv; o < nrhs; o += nwv) {
            double* z = R + (size_t)o * m;
            for (int j = 0; j < n; ++j) {
                const double* cj = M + (size_t)j * m;    // R[k][j], k<j, is cj[k]
                double acc = 0.0;
                for (int k = lane; k < j; k += 64) acc = fma(cj[k], z[k], acc);
                acc = wave_sum(acc);
                if (lane == 0) {
                    double d = rdiag[j];
                    z[j] = (fabs(d) > tol) ? (z[j] - acc) / d : 0.0;
                }
                __builtin_amdgcn_s_waitcnt(0);  // z[j] visible to this wave's later loads
                __builtin_amdgcn_wave_barrier();
            }
            // x = H_0 ... H_{n-1} [z; 0]
            for (int j = n - 1; j >= 0; --j) {
                const double* cj = M + (size_t)j * m;
                const double b = beta[j];
                if (b == 0.0) continue;
                double dot = 0.0;
                for (int i = j + lane; i < m; i += 64) dot = fma(cj[i], z[i], dot);
                dot = wave_sum(dot) * b;
                for (int i = j + lane; i < m; i += 64) z[i] = fma(-dot, cj[i], z[i]);
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
            }
            double* wo = sp.W_out + ((size_t)g * nrhs + o) * cols;
            for (int i = lane; i < m; i += 64) wo[i] = z[i];
        }
    } else {
        // back substitution R x = (Q^T b)[0:n], one wave per right-hand side
        for (int o = wv; o < nrhs; o += nwv) {
            double* c = M + (size_t)(n + o) * m;     // transformed rhs; x overwrites c[0:n]
            for (int j = n - 1; j >= 0; --j) {
                double acc = 0.0;
                for (int k = j + 1 + lane; k < n; k += 64) acc = fma(M[(size_t)k * m + j], c[k], acc);
                acc = wave_sum(acc);
                if (lane == 0) {
                    double d = rdiag[j];
                    c[j] = (fabs(d) > tol) ? (c[j] - acc) / d : 0.0;
                }
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
            }
            double* wo = sp.W_out + ((size_t)g * nrhs + o) * cols;
            for (int i = lane; i < n; i += 64) wo[i] = c[i];
        }
    }
}

size_t solve_work_doubles(int rows, int cols, int n_out) {
    const bool wide = rows < cols;
    const size_t m = wide ? cols : rows, n = wide ? rows : cols;
    // matrix (+ rhs columns when tall) + rhs block + rdiag + beta, rounded to 16 B
    size_t d = (n + (wide ? 0 : n_out)) * m + (size_t)n_out * m + 2 * n;
    return (d + 1) & ~(size_t)1;
}

int launch_readout_solve(const double* E, const double* D, int n_groups, int T, int transient,
                         int cols, int n_out, const double* t_scale, const double* t_shift,
                         double* W_out, int* status, void* workspace, hipStream_t stream) {
    SolveParams sp;
    const int rows = T - transient;
    sp.E = E; sp.D = D; sp.n_groups = n_groups; sp.T = T; sp.transient = transient;
    sp.cols = cols; sp.n_out = n_out; sp.t_scale = t_scale; sp.t_shift = t_shift;
    sp.W_out = W_out; sp.status = status;
    sp.work = reinterpret_cast<double*>(workspace);
    sp.work_stride = solve_work_doubles(rows, cols, n_out);
    sp.wide = rows < cols;
    sp.m = sp.wide ? cols : rows;
    sp.n = sp.wide ? rows : cols;
    hipLaunchKernelGGL(readout_qr_kernel, dim3(n_groups), dim3(1024), 0, stream, sp);
    return (int)hipGetLastError();
}

}  // namespace esn
