// Coded leg of the north-star driver, float64:
//   ldpc_encode_kernel   c = [u ; P u mod 2] per (frame, tx), written into the TxBits layout
//                        (ldpc_encode_bits / G.dot(u) % 2, Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py:90-93, :399-404)
//   qam_llr_kernel       decision-directed sigma^2 per frame (:108-112, :459, :465) and max-log LLRs
//                        LLR = (d1 - d0) / sigma^2, positive = bit 0 (:66-88)
//   ldpc_decode_kernel   pyldpc.decode restated: flooding log-domain sum-product, var = 10^(-snr/10),
//                        Lc = 2 y / var, stop on zero syndrome or maxiter; systematic message = first k
//                        bits (get_message); info-bit errors accumulated per group (:495-511)
#include "esn_common.h"

namespace esn {

__global__ __launch_bounds__(256) void ldpc_encode_kernel(LdpcEncodeParams ep) {
    extern __shared__ __attribute__((aligned(16))) char esm[];
    uint8_t* u = reinterpret_cast<uint8_t*>(esm);              // [k]
    const int cw = blockIdx.x, frame = cw / ep.n_t, tx = cw % ep.n_t;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int k = ep.k, n = ep.n, r = n - k;
    for (int j = tid; j < k; j += nth) {
        const uint8_t b = ep.u[(size_t)cw * k + j] & 1;
        u[j] = b;
        ep.bits[((size_t)frame * n + j) * ep.n_t + tx] = b;
    }
    __syncthreads();
    for (int i = tid; i < r; i += nth) {
        const uint8_t* prow = ep.P + (size_t)i * k;
        unsigned acc = 0;
        for (int j = 0; j < k; ++j) acc ^= (unsigned)(prow[j] & u[j]);
        ep.bits[((size_t)frame * n + k + i) * ep.n_t + tx] = (uint8_t)(acc & 1);
    }
}

__global__ __launch_bounds__(256) void qam_llr_kernel(LlrParams lp) {
    __shared__ double red[4];
    __shared__ double s_sigma2;
    const int N = lp.n_sub, n_t = lp.n_t, m = lp.m, half = m / 2;
    const int side = 1 << half;
    const double norm = sqrt(2.0 * (double)(side * side - 1) / 3.0);
    const int frame = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    const double* x = lp.X_hat + (size_t)frame * N * n_t * 2;
    // decision-directed noise estimate: mean over (subcarrier, tx) of |x - nearest point|^2
    double part = 0.0;
    for (int e = tid; e < N * n_t; e += nth) {
        const double re = x[2 * e], im = x[2 * e + 1];
        int i = (int)rint((re * norm + (double)(side - 1)) * 0.5);
        int j = (int)rint((im * norm + (double)(side - 1)) * 0.5);
        i = min(max(i, 0), side - 1); j = min(max(j, 0), side - 1);
        const double dr = re - (2.0 * i - (side - 1)) / norm, di = im - (2.0 * j - (side - 1)) / norm;
        part += dr * dr + di * di;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < (int)(nth >> 6); ++w) s += red[w];
        s_sigma2 = s / (double)(N * n_t) + 1e-12;
        if (lp.sigma2) lp.sigma2[frame] = s_sigma2;
    }
    __syncthreads();
    const double inv = 1.0 / fmax(s_sigma2, 1e-12);
    // index = i*side + j: bits [0, half) are the bits of j (Im axis), bits [half, m) those of i (Re axis);
    // the max-log LLR of a bit only involves its own axis (the other axis' nearest term cancels)
    for (int e = tid; e < N * n_t; e += nth) {
        const int sc = e / n_t, tx = e % n_t;
        const double re = x[2 * e], im = x[2 * e + 1];
        double* out = lp.llr + ((size_t)(frame * n_t + tx) * N + sc) * m;
        for (int b = 0; b < m; ++b) {
            const double t = (b < half) ? im : re;
            const int bb = (b < half) ? b : b - half;
            double d0 = 1e300, d1 = 1e300;
            for (int a = 0; a < side; ++a) {
                const double d = t - (2.0 * a - (side - 1)) / norm;
                const double dd = d * d;
                if ((a >> bb) & 1) d1 = fmin(d1, dd); else d0 = fmin(d0, dd);
            }
            out[b] = (d1 - d0) * inv;
        }
    }
}

__global__ __launch_bounds__(256) void ldpc_decode_kernel(LdpcDecodeParams dp) {
    extern __shared__ __attribute__((aligned(16))) char dsm2[];
    const int n = dp.n, mchk = dp.m_checks, E = dp.n_edges;
    double* Lq = reinterpret_cast<double*>(dsm2);      // [E] bit -> check, check-major edge order
    double* Lr = Lq + E;                                // [E] check -> bit
    double* Lc = Lr + E;                                // [n]
    uint8_t* xb = reinterpret_cast<uint8_t*>(Lc + n);   // [n]
    __shared__ int s_bad;
    const int cw = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    const double scale = 2.0 / dp.var;                  // Lc = 2 y / var
    for (int v = tid; v < n; v += nth) {
        const double l = dp.y[(size_t)cw * n + v] * scale;
        Lc[v] = l;
        xb[v] = (uint8_t)(l <= 0.0);
    }
    __syncthreads();
    for (int e = tid; e < E; e += nth) Lq[e] = Lc[dp.edge_var[e]];
    __syncthreads();
    for (int it = 0; it < dp.maxiter; ++it) {
        // horizontal step: Lr[e] = log((1 + X) / (1 - X)), X = prod_{e' != e} tanh(Lq[e'] / 2)
        for (int c = tid; c < mchk; c += nth) {
            const int s = dp.chk_ptr[c], t = dp.chk_ptr[c + 1];
            for (int e = s; e < t; ++e) {
                double X = 1.0;
                for (int q = s; q < t; ++q)
                    if (q != e) X *= tanh(0.5 * Lq[q]);
                const double num = 1.0 + X, den = 1.0 - X;
                Lr[e] = (num == 0.0) ? -1.0 : ((den == 0.0) ? 1.0 : log(num / den));
            }
        }
        __syncthreads();
        // vertical step + a-posteriori decision
        for (int v = tid; v < n; v += nth) {
            const int s = dp.var_ptr[v], t = dp.var_ptr[v + 1];
            double tot = Lc[v];
            for (int q = s; q < t; ++q) tot += Lr[dp.var_edge[q]];
            for (int q = s; q < t; ++q) { const int e = dp.var_edge[q]; Lq[e] = tot - Lr[e]; }
            xb[v] = (uint8_t)(tot <= 0.0);
        }
        if (tid == 0) s_bad = 0;
        __syncthreads();
        // syndrome
        int bad = 0;
        for (int c = tid; c < mchk; c += nth) {
            unsigned par = 0;
            for (int e = dp.chk_ptr[c]; e < dp.chk_ptr[c + 1]; ++e) par ^= xb[dp.edge_var[e]];
            bad |= (int)(par & 1);
        }
        if (bad) s_bad = 1;
        __syncthreads();
        const int stop = !s_bad;
        __syncthreads();
        if (stop) break;
    }
    int errs = 0;
    for (int v = tid; v < n; v += nth) {
        if (dp.x_out) dp.x_out[(size_t)cw * n + v] = xb[v];
        if (dp.u_true && v < dp.k) errs += (int)(xb[v] != (dp.u_true[(size_t)cw * dp.k + v] & 1));
    }
    if (dp.u_true) {
        __shared__ int red[4];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) errs += __shfl_down(errs, off);
        if ((tid & 63) == 0) red[tid >> 6] = errs;
        __syncthreads();
        if (tid == 0) {
            int e = 0;
            for (int w = 0; w < (int)(nth >> 6); ++w) e += red[w];
            const int grp = cw / dp.cw_per_group;
            atomicAdd(reinterpret_cast<unsigned long long*>(dp.err + grp), (unsigned long long)e);
            atomicAdd(reinterpret_cast<unsigned long long*>(dp.bits + grp), (unsigned long long)dp.k);
        }
    }
}

int launch_ldpc_encode(const LdpcEncodeParams& ep, hipStream_t stream) {
    hipLaunchKernelGGL(ldpc_encode_kernel, dim3(ep.n_frames * ep.n_t), dim3(256), (size_t)ep.k, stream, ep);
    return (int)hipGetLastError();
}

int launch_qam_llr(const LlrParams& lp, hipStream_t stream) {
    hipLaunchKernelGGL(qam_llr_kernel, dim3(lp.n_frames), dim3(256), 0, stream, lp);
    return (int)hipGetLastError();
}

int launch_ldpc_decode(const LdpcDecodeParams& dp, hipStream_t stream) {
    const size_t lds = sizeof(double) * (2 * (size_t)dp.n_edges + dp.n) + dp.n;
    if (lds > 150 * 1024) return -1;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ldpc_decode_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(ldpc_decode_kernel, dim3(dp.n_cw), dim3(256), lds, stream, dp);
    return (int)hipGetLastError();
}

}  // namespace esn
