// Baseline equaliser the reference plots the ESN against, float64 / complex128:
//
//   channel_estimate_kernel   per (block, rx): Y_LS = (1/N) FFT(y_LS[cp:]); per tx: LS at the pilot
//       subcarriers sc = tx, tx+n_t, ... (X_LS sparse pattern), linear inter/extrapolation to all N,
//       IFFT truncated to isi taps, diagonal MMSE shrinkage c/(1 + scaler/r_h), DFT back
//       (Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py:358-382, prior :212-214,279)
//   mmse_detect_kernel        per frame: Y = (1/N) FFT(y[cp:]) per rx; per subcarrier
//       X = (H^H H + No/Pi I)^-1 H^H Y / sqrt(Pi)  (:40-45, :444-448); nearest QAM point, natural
//       binary LSB-first bits, error count vs TxBits (:95-103, :451-456)
#include "esn_common.h"

namespace esn {

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cdiv(double2 a, double2 b) {
    const double d = b.x * b.x + b.y * b.y;
    return make_double2((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}

// in-place radix-2 DIT FFT of `count` length-N sequences stored back to back in LDS, input already in
// bit-reversed order; sign = -1 forward, +1 inverse (un-normalised).  All threads of the block call it.
__device__ __forceinline__ void fft_lds(double2* buf, int count, int N, int log2n, double sign) {
    const int half = N >> 1;
    for (int s = 1; s <= log2n; ++s) {
        const int hm = 1 << (s - 1);
        for (int e = threadIdx.x; e < count * half; e += blockDim.x) {
            const int q = e / half, b = e % half;
            const int j = b & (hm - 1);
            const int base = ((b >> (s - 1)) << s) + j;
            double sn, cs;
            sincospi(sign * (double)j / (double)hm, &sn, &cs);
            double2* xx = buf + (size_t)q * N;
            const double2 a = xx[base], c = xx[base + hm];
            const double tr = c.x * cs - c.y * sn, ti = c.x * sn + c.y * cs;
            xx[base] = make_double2(a.x + tr, a.y + ti);
            xx[base + hm] = make_double2(a.x - tr, a.y - ti);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void channel_estimate_kernel(ChanEstParams cp) {
    extern __shared__ __attribute__((aligned(16))) char csm[];
    const int N = cp.n_sub, T = N + cp.cp, n_t = cp.n_t, n_r = cp.n_r, isi = cp.isi, m = cp.m;
    double2* Yf = reinterpret_cast<double2*>(csm);        // [N]  (1/N) FFT of the LS pilot at this rx
    double2* full = Yf + N;                                // [N]  interpolated LS estimate
    double2* hls = full + N;                               // [N / n_t + 1] LS at the pilot subcarriers
    double2* ctd = hls + (N / n_t + 1);                    // [isi] time-domain taps
    const int tid = threadIdx.x, nth = blockDim.x;
    const int blk = blockIdx.x / n_r, rx = blockIdx.x % n_r;
    const double p_i = cp.p_i[blk], sp = sqrt(p_i);
    const int side = 1 << (m / 2);
    const double norm = sqrt(2.0 * (double)(side * side - 1) / 3.0);
    const double* y = cp.y_ls_cp + ((size_t)blk * T + cp.cp) * n_r * 2 + 2 * rx;
    for (int i = tid; i < N; i += nth) {
        const int rv = (int)(__brev((unsigned)i) >> (32 - cp.log2n));
        Yf[rv] = make_double2(y[(size_t)i * n_r * 2] / N, y[(size_t)i * n_r * 2 + 1] / N);
    }
    __syncthreads();
    fft_lds(Yf, 1, N, cp.log2n, -1.0);
    const int n_p = (N + n_t - 1) / n_t;                   // pilots per tx (sc = tx + n_t i < N)
    // prior: r_h[k] = exp(-k / (cp/9)) / sum_{j<=cp} exp(-j / (cp/9)), k < isi   (driver:212-214,279)
    const double tc = fmax((double)cp.cp / 9.0, 1e-12);
    double rsum = 0.0;
    for (int j = 0; j <= cp.cp; ++j) rsum += exp(-(double)j / tc);
    const double scaler = (cp.no / p_i) / ((double)N / 2.0);
    for (int tx = 0; tx < n_t; ++tx) {
        const int np_tx = (N - tx + n_t - 1) / n_t;
        for (int i = tid; i < np_tx; i += nth) {
            const int sc = tx + n_t * i;
            uint32_t idx = 0;
            for (int b = 0; b < m; ++b)
                idx |= (uint32_t)(cp.pilot_bits[((size_t)blk * N * m + (size_t)sc * m + b) * n_t + tx] & 1) << b;
            const double2 xp = make_double2((2.0 * (int)(idx / side) - (side - 1)) / norm * sp + 1e-12,
                                            (2.0 * (int)(idx % side) - (side - 1)) / norm * sp);
            hls[i] = cdiv(Yf[sc], xp);
        }
        __syncthreads();
        // linear interpolation with extrapolation beyond the first / last pilot (scipy interp1d)
        for (int k = tid; k < N; k += nth) {
            int i = (k - tx >= 0) ? (k - tx) / n_t : 0;
            if (i > np_tx - 2) i = np_tx - 2;
            if (i < 0) i = 0;
            const double w = ((double)k - (double)(tx + n_t * i)) / (double)n_t;
            const double2 lo = hls[i], hi = (np_tx > 1) ? hls[i + 1] : hls[i];
            full[k] = make_double2(lo.x + w * (hi.x - lo.x), lo.y + w * (hi.y - lo.y));
        }
        __syncthreads();
        if (cp.ls_only) {      // H_LS of the block-fading drivers: the interpolated LS estimate as is (:321-333)
            for (int k = tid; k < N; k += nth) {
                double* ho = cp.H + ((((size_t)blk * N + k) * n_r + rx) * n_t + tx) * 2;
                ho[0] = full[k].x; ho[1] = full[k].y;
            }
            __syncthreads();
            continue;
        }
        // c_LS[j] = (1/N) sum_k full[k] e^{+2 pi i jk/N}, j < isi; then shrink
        if (tid < isi) {
            double ar = 0.0, ai = 0.0;
            for (int k = 0; k < N; ++k) {
                double sn, cs;
                sincospi(2.0 * (double)((tid * k) % N) / (double)N, &sn, &cs);
                ar += full[k].x * cs - full[k].y * sn;
                ai += full[k].x * sn + full[k].y * cs;
            }
            const double rh = exp(-(double)tid / tc) / rsum;
            const double g = 1.0 / (scaler / rh + 1.0);
            ctd[tid] = make_double2(ar / N * g, ai / N * g);
        }
        __syncthreads();
        // H[k] = sum_{j<isi} c[j] e^{-2 pi i jk/N}
        for (int k = tid; k < N; k += nth) {
            double hr = 0.0, hi = 0.0;
            for (int j = 0; j < isi; ++j) {
                double sn, cs;
                sincospi(-2.0 * (double)((j * k) % N) / (double)N, &sn, &cs);
                hr += ctd[j].x * cs - ctd[j].y * sn;
                hi += ctd[j].x * sn + ctd[j].y * cs;
            }
            double* ho = cp.H + ((((size_t)blk * N + k) * n_r + rx) * n_t + tx) * 2;
            ho[0] = hr; ho[1] = hi;
        }
        __syncthreads();
    }
    (void)n_p;
}

constexpr int MMSE_NT = 4;     // transmit antennas handled in registers

__global__ __launch_bounds__(256) void mmse_detect_kernel(MmseParams mp) {
    extern __shared__ __attribute__((aligned(16))) char msm[];
    __shared__ int red[4];
    const int N = mp.n_sub, T = N + mp.cp, n_t = mp.n_t, n_r = mp.n_r, m = mp.m;
    double2* Y = reinterpret_cast<double2*>(msm);          // [n_r][N]
    const int tid = threadIdx.x, nth = blockDim.x;
    const int frame = blockIdx.x, blk = frame / mp.frames_per_group;
    const double* y = mp.y_cp + ((size_t)frame * T + mp.cp) * n_r * 2;
    for (int e = tid; e < N * n_r; e += nth) {
        const int i = e / n_r, rx = e % n_r;
        const int rv = (int)(__brev((unsigned)i) >> (32 - mp.log2n));
        Y[(size_t)rx * N + rv] = make_double2(y[(size_t)e * 2] / N, y[(size_t)e * 2 + 1] / N);
    }
    __syncthreads();
    fft_lds(Y, n_r, N, mp.log2n, -1.0);
    const double p_i = mp.p_i[blk], lam = mp.zf ? 1e-12 : mp.no / p_i, isp = 1.0 / sqrt(p_i);
    const int side = 1 << (m / 2);
    const double norm = sqrt(2.0 * (double)(side * side - 1) / 3.0);
    int errs = 0;
    for (int k = tid; k < N; k += nth) {
        // G = H^H H + lam I (Hermitian, n_t x n_t), r = H^H Y
        double2 G[MMSE_NT][MMSE_NT], rhs[MMSE_NT];
#pragma unroll
        for (int a = 0; a < MMSE_NT; ++a) {
            rhs[a] = make_double2(0.0, 0.0);
#pragma unroll
            for (int b = 0; b < MMSE_NT; ++b) G[a][b] = make_double2(a == b ? lam : 0.0, 0.0);
        }
        const double* hk = mp.H + (((size_t)blk * N + k) * n_r) * n_t * 2;
        for (int rx = 0; rx < n_r; ++rx) {
            double2 h[MMSE_NT];
#pragma unroll
            for (int a = 0; a < MMSE_NT; ++a)
                h[a] = (a < n_t) ? make_double2(hk[((size_t)rx * n_t + a) * 2], hk[((size_t)rx * n_t + a) * 2 + 1])
                                 : make_double2(0.0, 0.0);
            const double2 yv = Y[(size_t)rx * N + k];
#pragma unroll
            for (int a = 0; a < MMSE_NT; ++a) {
                const double2 hc = make_double2(h[a].x, -h[a].y);
                const double2 t = cmul(hc, yv);
                rhs[a].x += t.x; rhs[a].y += t.y;
#pragma unroll
                for (int b = 0; b < MMSE_NT; ++b) {
                    const double2 u = cmul(hc, h[b]);
                    G[a][b].x += u.x; G[a][b].y += u.y;
                }
            }
        }
#pragma unroll
        for (int a = 0; a < MMSE_NT; ++a)
            if (a >= n_t) G[a][a] = make_double2(1.0, 0.0);       // padding antennas: identity rows
        // Gaussian elimination (G is Hermitian positive definite: no pivoting needed)
#pragma unroll
        for (int c = 0; c < MMSE_NT; ++c) {
            const double2 piv = G[c][c];
#pragma unroll
            for (int rr = c + 1; rr < MMSE_NT; ++rr) {
                const double2 f = cdiv(G[rr][c], piv);
#pragma unroll
                for (int cc = c; cc < MMSE_NT; ++cc) {
                    const double2 t = cmul(f, G[c][cc]);
                    G[rr][cc].x -= t.x; G[rr][cc].y -= t.y;
                }
                const double2 t = cmul(f, rhs[c]);
                rhs[rr].x -= t.x; rhs[rr].y -= t.y;
            }
        }
        double2 x[MMSE_NT];
#pragma unroll
        for (int c = MMSE_NT - 1; c >= 0; --c) {
            double2 acc = rhs[c];
#pragma unroll
            for (int cc = c + 1; cc < MMSE_NT; ++cc) {
                const double2 t = cmul(G[c][cc], x[cc]);
                acc.x -= t.x; acc.y -= t.y;
            }
            x[c] = cdiv(acc, G[c][c]);
        }
#pragma unroll
        for (int tx = 0; tx < MMSE_NT; ++tx) {
            if (tx >= n_t) continue;
            const double re = x[tx].x * isp, im = x[tx].y * isp;
            if (mp.X_hat) {
                double* xo = mp.X_hat + (((size_t)frame * N + k) * n_t + tx) * 2;
                xo[0] = re; xo[1] = im;
            }
            int i = (int)rint((re * norm + (double)(side - 1)) * 0.5);
            int j = (int)rint((im * norm + (double)(side - 1)) * 0.5);
            i = min(max(i, 0), side - 1);
            j = min(max(j, 0), side - 1);
            const int idx = i * side + j;
            const uint8_t* tb = mp.tx_bits + ((size_t)frame * N * m + (size_t)k * m) * n_t + tx;
            for (int b = 0; b < m; ++b) errs += (((idx >> b) & 1) != (int)tb[(size_t)b * n_t]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) errs += __shfl_down(errs, off);
    if ((tid & 63) == 0) red[tid >> 6] = errs;
    __syncthreads();
    if (tid == 0) {
        int e = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) e += red[w];
        atomicAdd(reinterpret_cast<unsigned long long*>(mp.err + blk), (unsigned long long)e);
        atomicAdd(reinterpret_cast<unsigned long long*>(mp.bits + blk), (unsigned long long)(N * m * n_t));
    }
}

// Perfect-CSI channel: H[k] = sum_j c[j] e^{-2 pi i jk/N} per link (H_true of OFDM_MIMO_2-2_NBF_LDPC.py:273-279)
__global__ __launch_bounds__(256) void taps_to_freq_kernel(TapsFreqParams tp) {
    const int N = tp.n_sub, n_t = tp.n_t, n_r = tp.n_r, isi = tp.isi;
    const size_t total = (size_t)tp.n_blocks * N * n_r * n_t;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int tx = (int)(e % n_t), rx = (int)((e / n_t) % n_r), k = (int)((e / ((size_t)n_t * n_r)) % N);
        const size_t blk = e / ((size_t)n_t * n_r * N);
        const double* c = tp.taps + (((blk * n_r + rx) * n_t + tx) * isi) * 2;
        double hr = 0.0, hi = 0.0;
        for (int j = 0; j < isi; ++j) {
            double sn, cs;
            sincospi(-2.0 * (double)((j * k) % N) / (double)N, &sn, &cs);
            hr += c[2 * j] * cs - c[2 * j + 1] * sn;
            hi += c[2 * j] * sn + c[2 * j + 1] * cs;
        }
        tp.H[2 * e] = hr; tp.H[2 * e + 1] = hi;
    }
}

int launch_taps_to_freq(const TapsFreqParams& tp, hipStream_t stream) {
    const size_t total = (size_t)tp.n_blocks * tp.n_sub * tp.n_r * tp.n_t;
    const int blocks = (int)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
    hipLaunchKernelGGL(taps_to_freq_kernel, dim3(blocks), dim3(256), 0, stream, tp);
    return (int)hipGetLastError();
}

int launch_channel_estimate(const ChanEstParams& cp, hipStream_t stream) {
    const size_t lds = sizeof(double2) * ((size_t)2 * cp.n_sub + cp.n_sub / cp.n_t + 1 + cp.isi);
    hipLaunchKernelGGL(channel_estimate_kernel, dim3(cp.n_blocks * cp.n_r), dim3(256), lds, stream, cp);
    return (int)hipGetLastError();
}

int launch_mmse_detect(const MmseParams& mp, hipStream_t stream) {
    if (mp.n_t > MMSE_NT) return -1;
    const size_t lds = sizeof(double2) * (size_t)mp.n_r * mp.n_sub;
    if (lds > 150 * 1024) return -1;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mmse_detect_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mmse_detect_kernel, dim3(mp.n_frames), dim3(256), lds, stream, mp);
    return (int)hipGetLastError();
}

}  // namespace esn
