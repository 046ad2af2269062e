// Readout training: W_out = (pinv(E[transient:]) @ teacher[transient:]).T  (pyESN.py:191-192)
// as a float64 Householder QR, one workgroup per trained ESN.
//
//   rows <  cols (4x8, N=128: 128 x 528): QR of A^T, minimum-norm solution
//                X = Q R^-T B   -- what pinv returns for an under-determined system
//   rows >= cols (SISO / 2x2: 512 x 104):           QR of [A | B], X = R^-1 (Q^T B)
//
// The working matrix is column-major in the caller's workspace (L2 resident);
// reflector j is applied to the trailing columns one wave per column.
#include <stdlib.h>
#include "esn_common.h"

namespace esn {

struct SolveParams {
    const double* E; const double* D;
    const float* E32;       // Cholesky path: extended states as float32 (E unused) -- esn_harvest_batch_f32
    int n_groups, T, transient, cols, n_out;
    const double* t_scale; const double* t_shift;
    double* W_out; int* status;
    double* work; size_t work_stride;   // doubles per group
    int m, n, wide;
    int skip;   // diagnostic only (ESN_CHOL_SKIP env): bit0 Gram, bit1 Cholesky, bit2 solves, bit3 W_out
    int part_ok;   // big kernel: the three-partial-sums W_out pass fits the LDS the launcher allocated
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
    sp.E = E; sp.E32 = nullptr; sp.D = D; sp.n_groups = n_groups; sp.T = T; sp.transient = transient;
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

// ---------------------------------------------------------------------------------
// Fast path for well-conditioned batched fits: normal equations in float64 with the
// Gram matrix (<= 128 x 128) and its Cholesky factor resident in LDS.
//   rows <  cols:  G = A A^T,  G alpha = B,      W_out^T = A^T alpha   (minimum norm)
//   rows >= cols:  G = A^T A,  G W_out^T = A^T B
// Error ~ cond(A)^2 eps: with the model's state noise cond(A) ~ 1e3, i.e.
// ~1e-10 -- far below the float32 harvest.  A non-positive / tiny pivot sets status=1 and
// the caller re-solves that group with the QR kernel.
// ---------------------------------------------------------------------------------
namespace esn {

constexpr int CH_NP = 128;        // padded Gram dimension
constexpr int CH_LD = CH_NP + 1;  // LDS row stride (doubles): conflict-free row-strided reads
constexpr int CH_KC = 32;         // k-chunk staged per pass

__global__ __launch_bounds__(1024) void readout_chol_kernel(SolveParams sp) {
    extern __shared__ __attribute__((aligned(16))) char chol_smem[];
    double* Gs = reinterpret_cast<double*>(chol_smem);            // [CH_NP][CH_LD]   (phase 2+)
    double* As = Gs;                                               // [CH_KC][CH_NP+4] (phase 1, aliased)
    double* Bs = Gs + CH_NP * CH_LD;                               // [nrhs][CH_NP] rhs / solution
    __shared__ int sh_bad;
    const int g = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int rows = sp.T - sp.transient, cols = sp.cols, nrhs = sp.n_out;
    const bool wide = rows < cols;
    const int n = wide ? rows : cols;      // Gram dimension (<= CH_NP)
    const int m = wide ? cols : rows;      // contraction length
    const size_t a_off = ((size_t)g * sp.T + sp.transient) * cols;
    const double* A = sp.E ? sp.E + a_off : nullptr;                     // [rows][cols]
    const float* A32 = sp.E32 ? sp.E32 + a_off : nullptr;                //   ... or as float32
    const double* Dg = sp.D + ((size_t)g * sp.T + sp.transient) * nrhs;
    constexpr int AS_LD = CH_NP + 4;

    // ---- phase 1: G = sum_k a_k a_k^T on the float64 matrix pipe ------------------------------
    // v_mfma_f64_16x16x4_f64 (A[l%16][l/16], B[l/16][l%16], C reg i: row 4i + l/16, col l%16) runs at
    // the vector-FMA rate on gfx950 (64 cycles, probe in tools/mfma_layout_probe.hip) but takes
    // one LDS read per operand and 1024 FMAs, where a 4x4 register tile takes 8 reads per 16 FMAs
    // per lane -- the old loop was LDS-bound at a tenth of the FMA rate.  Only the 36 lower 16x16
    // tiles are formed (the factorisation reads nothing above the diagonal), dealt to the 16
    // waves so that every SIMD (wave % 4) carries 9: waves 0-3 three tiles, the others two.
    // lower tile t = ti (ti + 1) / 2 + tj goes to SIMD t % 4, slot t / 4 of its nine
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const int g_cnt = (wvu >> 2) == 0 ? 3 : 2;
    int g_ti[3], g_tj[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int m = wvu >> 2;
        const int idx = m == 0 ? q : 2 * m + 1 + (q < 2 ? q : 0);
        const int t = 4 * idx + (wvu & 3);
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        g_ti[q] = ti;
        g_tj[q] = t - ti * (ti + 1) / 2;
    }
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    f64x4 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = f64x4{0.0, 0.0, 0.0, 0.0};
    const int fr_off = (lane >> 4) * (CH_NP + 4) + (lane & 15);      // operand element of this lane in a 4-row slab
    // tall case also needs A^T B: thread (o, i) partial sums, o < nrhs, i < n  -> first nrhs*128 threads
    double atb = 0.0;
    const int ao = tid / CH_NP, ai = tid % CH_NP;
    // chunk staging is register-prefetched one chunk ahead and double-buffered in LDS, so the
    // global-load latency of chunk c+1 hides under the FMAs of chunk c (one barrier per chunk)
    constexpr int EPT = CH_KC * CH_NP / 1024;          // staged elements per thread (4)
    // two chunks in flight in registers (an HBM round trip is longer than the MFMAs of one chunk)
    double stgA[EPT], stgB[EPT];
    auto fetch = [&](double (&stg)[EPT], int k0) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = tid + 1024 * q;
            int kk, i;
            if (wide) { i = e / CH_KC; kk = e % CH_KC; } else { kk = e / CH_NP; i = e % CH_NP; }
            const int k = k0 + kk;
            const size_t ai_ = wide ? (size_t)i * cols + k : (size_t)k * cols + i;
            stg[q] = (i < n && k < m) ? (A32 ? (double)A32[ai_] : A[ai_]) : 0.0;
        }
    };
    auto commit = [&](const double (&stg)[EPT], double* dst) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = tid + 1024 * q;
            int kk, i;
            if (wide) { i = e / CH_KC; kk = e % CH_KC; } else { kk = e / CH_NP; i = e % CH_NP; }
            dst[kk * AS_LD + i] = stg[q];
        }
    };
    double* Abuf[2] = {As, As + CH_KC * AS_LD};
    const int m_run = (sp.skip & 1) ? CH_KC : m;
    // chunk c is multiplied out of Abuf[c & 1]; `stg_next` holds chunk c+1, `stg_free` receives chunk c+2
    auto chunk = [&](int k0, int cur, const double (&stg_next)[EPT], double (&stg_free)[EPT]) {
        if (k0 + 2 * CH_KC < m_run) fetch(stg_free, k0 + 2 * CH_KC);
        const double* Ac = Abuf[cur];
        const int kmax = (m - k0 < CH_KC) ? m - k0 : CH_KC;
        // (rows past m and columns past n of the chunk are zero-filled by fetch)
#pragma unroll 2
        for (int k4 = 0; k4 < CH_KC; k4 += 4) {
            const double* slab = Ac + k4 * AS_LD + fr_off;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (q < g_cnt)
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(slab[g_ti[q] * 16], slab[g_tj[q] * 16], acc[q], 0, 0, 0);
        }
        if (!wide && ao < nrhs) {
            const double sc = sp.t_scale ? sp.t_scale[(size_t)g * nrhs + ao] : 1.0;
            const double sh = sp.t_shift ? sp.t_shift[(size_t)g * nrhs + ao] : 0.0;
            for (int kk = 0; kk < kmax; ++kk)
                atb = fma(Ac[kk * AS_LD + ai], Dg[(size_t)(k0 + kk) * nrhs + ao] * sc + sh, atb);
        }
        if (k0 + CH_KC < m_run) commit(stg_next, Abuf[cur ^ 1]);
        __syncthreads();
    };
    fetch(stgA, 0);
    commit(stgA, Abuf[0]);
    if (CH_KC < m_run) fetch(stgA, CH_KC);
    __syncthreads();
    for (int k0 = 0; k0 < m_run; k0 += 2 * CH_KC) {
        chunk(k0, 0, stgA, stgB);
        if (k0 + CH_KC < m_run) chunk(k0 + CH_KC, 1, stgB, stgA);
    }
    __syncthreads();
    // ---- phase 2: G and the right-hand sides into LDS ----------------------------------
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (q < g_cnt) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                Gs[(g_ti[q] * 16 + 4 * i + (lane >> 4)) * CH_LD + g_tj[q] * 16 + (lane & 15)] = acc[q][i];
        }
    if (tid == 0) sh_bad = 0;
    for (int e = tid; e < nrhs * CH_NP; e += 1024) {
        const int o = e / CH_NP, i = e % CH_NP;
        double v = 0.0;
        if (wide && i < n) {
            const double sc = sp.t_scale ? sp.t_scale[(size_t)g * nrhs + o] : 1.0;
            const double sh = sp.t_shift ? sp.t_shift[(size_t)g * nrhs + o] : 0.0;
            v = Dg[(size_t)i * nrhs + o] * sc + sh;
        }
        if (wide) Bs[e] = v;
    }
    __syncthreads();
    if (!wide && ao < nrhs) Bs[ao * CH_NP + ai] = (ai < n) ? atb : 0.0;
    // largest diagonal entry (pivot tolerance)
    double dmax = 0.0;
    if (tid < n) dmax = Gs[tid * CH_LD + tid];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dmax = fmax(dmax, __shfl_down(dmax, off));
    __shared__ double sh_max[2];
    if (tid < 128 && lane == 0) sh_max[wv] = dmax;
    __syncthreads();
    const double piv_tol = fmax(sh_max[0], sh_max[1]) * 1e-14;

    // ---- phase 3: blocked right-looking Cholesky, 16-column blocks --------------------------
    // per block: (a) wave 0 factorises the 16x16 diagonal block in registers (lane r = row r, the
    // pivots travel by v_readlane) and inverts it (lane c = column c of L11^-1); (b) the panel
    // L21 = A21 L11^-T and (c) the trailing update A22 -= L21 L21^T are 16x16x4 float64 MFMAs out
    // of / into the LDS image.  Three barriers per 16 columns instead of two per column.
    // A rejected pivot (v <= tol) drops its direction, as pinv would: unit diagonal, zero column
    // (row c of L11^-1 is zeroed so the panel column vanishes too) and the group is flagged.
    __shared__ double sh_invd[CH_NP];
    __shared__ double sh_linv[16][17];
    auto bcast = [](double x, int src) -> double {       // wave-uniform copy of lane `src` (constant)
        const uint64_t u = __builtin_bit_cast(uint64_t, x);
        const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)u, src);
        const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(u >> 32), src);
        return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
    };
    const int nblk = (sp.skip & 2) ? 1 : (n + 15) / 16;
    const int lr = lane & 15, lq = lane >> 4;
    for (int kb = 0; kb < nblk; ++kb) {
        const int j0 = 16 * kb;
        if (wv == 0) {
            const int r = lane & 15;                       // lanes 16..63 mirror lanes 0..15 (no divergence)
            double a[16], x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = Gs[(j0 + r) * CH_LD + j0 + c];
            double my_invd = 1.0;
            unsigned rejected = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double v = bcast(a[j], j);
                const bool live = j0 + j < n;
                const bool ok = live && v > piv_tol;
                if (live && !ok) rejected |= 1u << j;
                const double d = ok ? sqrt(v) : 1.0, inv_d = ok ? 1.0 / d : 0.0;
                if (r == j) my_invd = ok ? inv_d : 1.0;
                a[j] = (r == j) ? d : ((r > j) ? a[j] * inv_d : 0.0);          // column j of L11
#pragma unroll
                for (int k = j + 1; k < 16; ++k) {
                    const double lkj = bcast(a[j], k);
                    if (r >= k) a[k] = fma(-a[j], lkj, a[k]);
                }
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) Gs[(j0 + r) * CH_LD + j0 + c] = a[c];
                sh_invd[j0 + r] = my_invd;
            }
            if (rejected && lane == 0) sh_bad = 1;
            // lane c: column c of L11^-1 by forward substitution, L11[i][k] = a[k] of lane i
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                double sacc = (i == r) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < i; ++k) sacc = fma(-bcast(a[k], i), x[k], sacc);   // x[k] = 0 for k < c
                const double idi = bcast(my_invd, i);
                x[i] = (i >= r && !((rejected >> i) & 1u)) ? sacc * idi : 0.0;
            }
            if (lane < 16) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sh_linv[i][r] = x[i];
            }
        }
        __syncthreads();
        // (b) panel: row tile rt of L21 = A21[rt] * L11^-T  (B operand [k][n] = L11^-1[n][k])
        const int ntile = (n + 15) / 16;
        {
            const int rt = kb + 1 + wv;
            if (rt < ntile) {
                f64x4 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k0 = 0; k0 < 16; k0 += 4)
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(Gs[(rt * 16 + lr) * CH_LD + j0 + k0 + lq],
                                                             sh_linv[lr][k0 + lq], c, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) Gs[(rt * 16 + 4 * i + lq) * CH_LD + j0 + lr] = c[i];
            }
        }
        __syncthreads();
        // (c) trailing update of the lower tiles (ti >= tj > kb): A22[ti][tj] -= L21[ti] L21[tj]^T
        {
            const int mt = ntile - kb - 1, cnt = mt * (mt + 1) / 2;
            for (int t = wv; t < cnt; t += 16) {
                int di = 0;
                while ((di + 1) * (di + 2) / 2 <= t) ++di;
                const int ti = kb + 1 + di, tj = kb + 1 + t - di * (di + 1) / 2;
                f64x4 c;
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = Gs[(ti * 16 + 4 * i + lq) * CH_LD + tj * 16 + lr];
#pragma unroll
                for (int k0 = 0; k0 < 16; k0 += 4)
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(-Gs[(ti * 16 + lr) * CH_LD + j0 + k0 + lq],
                                                             Gs[(tj * 16 + lr) * CH_LD + j0 + k0 + lq], c, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) Gs[(ti * 16 + 4 * i + lq) * CH_LD + tj * 16 + lr] = c[i];
            }
        }
        __syncthreads();
    }

    // ---- phase 4: L L^T x = b, column-oriented (no reductions), one wave per right-hand side --
    for (int o = wv; o < ((sp.skip & 4) ? 0 : nrhs); o += 16) {
        double* x = Bs + o * CH_NP;
        double b0 = x[lane], b1 = x[lane + 64];       // rows lane and lane+64 of this rhs
        for (int j = 0; j < n; ++j) {                 // forward: L z = b
            const double own = (j < 64) ? b0 : b1;
            const double zj = __shfl(own, j & 63) * sh_invd[j];
            if (lane == (j & 63)) { if (j < 64) b0 = zj; else b1 = zj; }
            const double l0 = (lane > j) ? Gs[lane * CH_LD + j] : 0.0;
            const double l1 = (lane + 64 > j && lane + 64 < n) ? Gs[(lane + 64) * CH_LD + j] : 0.0;
            b0 = fma(-l0, zj, b0);
            b1 = fma(-l1, zj, b1);
        }
        for (int j = n - 1; j >= 0; --j) {            // backward: L^T x = z
            const double own = (j < 64) ? b0 : b1;
            const double xj = __shfl(own, j & 63) * sh_invd[j];
            if (lane == (j & 63)) { if (j < 64) b0 = xj; else b1 = xj; }
            const double* Lj = Gs + j * CH_LD;        // row j: L[j][i], i < j
            const double l0 = (lane < j) ? Lj[lane] : 0.0;
            const double l1 = (lane + 64 < j) ? Lj[lane + 64] : 0.0;
            b0 = fma(-l0, xj, b0);
            b1 = fma(-l1, xj, b1);
        }
        x[lane] = b0; x[lane + 64] = b1;
    }
    __syncthreads();

    // ---- phase 5: W_out ---------------------------------------------------------------------
    if (wide) {
        // W_out[o][c] = sum_i A[i][c] alpha[i][o].  Every element of A is fetched once, by 16-byte
        // loads: thread (third, pair) sums a third of the rows for columns 2 pair, 2 pair + 1 and all
        // nrhs outputs; the three partial sums meet in LDS (the factor is no longer needed).  HBM
        // latency-bound: ~12 waves x 8 loads x 1 KB in flight per CU.
        const int npair = cols / 2;
        if ((cols & 1) == 0 && 3 * npair <= 1024 && 3 * npair * 16 <= CH_NP * CH_LD && !(sp.skip & 8)) {
            double* part = Gs;                               // [3][nrhs<=8][cols]
            const int third = tid / npair, pr = tid - third * npair;
            if (third < 3) {
                const int per = (n + 2) / 3;
                const int i0 = third * per, i1 = (i0 + per < n) ? i0 + per : n;
                double w0[8], w1[8];
#pragma unroll
                for (int o = 0; o < 8; ++o) { w0[o] = 0.0; w1[o] = 0.0; }
                const size_t ac = 2 * (size_t)pr;
#pragma unroll 8
                for (int i = i0; i < i1; ++i) {
                    double2 a;
                    if (A32) {
                        const float2 af = *reinterpret_cast<const float2*>(A32 + ac + (size_t)i * cols);
                        a = double2{(double)af.x, (double)af.y};
                    } else {
                        a = *reinterpret_cast<const double2*>(A + ac + (size_t)i * cols);
                    }
#pragma unroll
                    for (int o = 0; o < 8; ++o)
                        if (o < nrhs) {
                            const double al = Bs[o * CH_NP + i];
                            w0[o] = fma(a.x, al, w0[o]);
                            w1[o] = fma(a.y, al, w1[o]);
                        }
                }
#pragma unroll
                for (int o = 0; o < 8; ++o)
                    if (o < nrhs) {
                        part[(third * 8 + o) * cols + 2 * pr] = w0[o];
                        part[(third * 8 + o) * cols + 2 * pr + 1] = w1[o];
                    }
            }
            __syncthreads();
            for (int e = tid; e < nrhs * cols; e += 1024) {
                const int o = e / cols, c = e - o * cols;
                sp.W_out[((size_t)g * nrhs + o) * cols + c] =
                    (part[o * cols + c] + part[(8 + o) * cols + c]) + part[(16 + o) * cols + c];
            }
        } else {
            for (int c = tid; c < ((sp.skip & 8) ? 0 : cols); c += 1024) {
                double w[8];
#pragma unroll
                for (int o = 0; o < 8; ++o) w[o] = 0.0;
#pragma unroll 8
                for (int i = 0; i < n; ++i) {
                    const double a = A32 ? (double)A32[(size_t)i * cols + c] : A[(size_t)i * cols + c];
#pragma unroll
                    for (int o = 0; o < 8; ++o)
                        if (o < nrhs) w[o] = fma(a, Bs[o * CH_NP + i], w[o]);
                }
#pragma unroll
                for (int o = 0; o < 8; ++o)
                    if (o < nrhs) sp.W_out[((size_t)g * nrhs + o) * cols + c] = w[o];
            }
        }
    } else {
        for (int e = tid; e < nrhs * cols; e += 1024) {
            const int o = e / cols, c = e % cols;
            sp.W_out[((size_t)g * nrhs + o) * cols + c] = Bs[o * CH_NP + c];
        }
    }
    if (tid == 0) sp.status[g] = sh_bad;
}

int launch_readout_chol(const double* E, const float* E32, const double* D, int n_groups, int T, int transient,
                        int cols, int n_out, const double* t_scale, const double* t_shift,
                        double* W_out, int* status, hipStream_t stream) {
    SolveParams sp;
    sp.E32 = E32;
    const int rows = T - transient;
    const int n = rows < cols ? rows : cols;
    if (n > CH_NP || n_out > 8) return -1;     // tall case stages A^T B with nrhs*128 <= 1024 threads
    sp.E = E; sp.D = D; sp.n_groups = n_groups; sp.T = T; sp.transient = transient;
    sp.cols = cols; sp.n_out = n_out; sp.t_scale = t_scale; sp.t_shift = t_shift;
    sp.W_out = W_out; sp.status = status; sp.work = nullptr; sp.work_stride = 0;
    sp.wide = rows < cols; sp.m = sp.wide ? cols : rows; sp.n = n;
    sp.skip = knobs().chol_skip;
    const size_t lds = sizeof(double) * ((size_t)CH_NP * CH_LD + (size_t)n_out * CH_NP);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(readout_chol_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(readout_chol_kernel, dim3(n_groups), dim3(1024), lds, stream, sp);
    return (int)hipGetLastError();
}

}  // namespace esn

// ---------------------------------------------------------------------------------
// The same normal-equations solve for Gram dimensions 129..512 (4x8 at N = 512: 512 x 528;
// N_res = 300 at N = 512: 512 x 316), where neither the Gram matrix (2 MB) nor its factor fits LDS.
// One workgroup per trained ESN, the Gram matrix / factor column-major in a caller workspace
// (L2 / Infinity Cache resident), every O(n^3) part on the float64 matrix pipe:
//   phase 1  G (lower 16x16 tiles) in passes of four tile columns: the k-chunks of A stream through
//            LDS once per pass (double-buffered, register-prefetched), <= 8 accumulator tiles per wave
//   phase 3  left-looking Cholesky by 16-column panels: panel tile (rt, j) = G(rt, j) - L(rt, :16j) L(j, :16j)^T
//            with both operands read straight from the workspace in MFMA layout (column-major storage makes a
//            lane group's 16 rows one 128-byte segment), then the 16x16 diagonal factorisation / inversion in
//            registers and the panel solve by MFMA exactly as in the LDS kernel; the finished panel goes back
//            to the workspace
//   phase 4  column-oriented substitutions, one wave per right-hand side, the next column prefetched
//   phase 5  W_out^T = A^T alpha (wide) -- the LDS kernel's pass over A
// Same pivot rule (v <= 1e-14 max diag: direction dropped, group flagged) and the same arithmetic order per
// tile as the LDS kernel, so the two agree to round-off where both apply.
// ---------------------------------------------------------------------------------
namespace esn {

constexpr int CB_NMAX = 512;      // largest Gram dimension
constexpr int CB_KC = 8;          // k-chunk staged per pass
constexpr int CB_PLD = 17;        // LDS row stride of the panel (doubles)
constexpr int CB_NT = 512;        // threads: 8 waves, two per SIMD -> 256 registers per lane (16 accumulator tiles)
constexpr int CB_NW = CB_NT / 64;
constexpr int CB_EPT = CB_NMAX * CB_KC / CB_NT;     // staged elements per thread
constexpr int CB_TPW = 16;        // Gram tiles per wave and pass: ceil((32 + 31 + 30 + 29) / 8)

size_t chol_big_work_doubles(int n) {
    const size_t np = (size_t)round_up(n, 16);
    return np * np;
}

__global__ __launch_bounds__(CB_NT) void readout_chol_big_kernel(SolveParams sp) {
    extern __shared__ __attribute__((aligned(16))) char cb_smem[];
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    __shared__ int sh_bad;
    __shared__ double sh_invd[CB_NMAX];
    __shared__ double sh_linv[16][17];
    __shared__ double sh_red[CB_NW];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const int rows = sp.T - sp.transient, cols = sp.cols, nrhs = sp.n_out;
    const bool wide = rows < cols;
    const int n = wide ? rows : cols;      // Gram dimension
    const int m = wide ? cols : rows;      // contraction length
    const int np = round_up(n, 16), ntile = np / 16;
    const int ld = np;
    double* Gw = sp.work + (size_t)g * sp.work_stride;                  // [np][np] column-major, lower part
    const size_t a_off = ((size_t)g * sp.T + sp.transient) * cols;
    const double* A = sp.E ? sp.E + a_off : nullptr;
    const float* A32 = sp.E32 ? sp.E32 + a_off : nullptr;
    const double* Dg = sp.D + ((size_t)g * sp.T + sp.transient) * nrhs;
    const int lr = lane & 15, lq = lane >> 4;
    if (tid == 0) sh_bad = 0;

    // ---- phase 1: Gram ----------------------------------------------------------------------------
    {
        const int AS_LD = np + 4;
        double* Abuf[2] = {reinterpret_cast<double*>(cb_smem), reinterpret_cast<double*>(cb_smem) + (size_t)CB_KC * AS_LD};
        const int ept = (np * CB_KC + CB_NT - 1) / CB_NT;             // staged elements per thread (<= CB_EPT)
        double stg[CB_EPT];
        auto fetch = [&](int k0) {
#pragma unroll
            for (int q = 0; q < CB_EPT; ++q) {
                const int e = tid + CB_NT * q;
                int kk, i;
                if (wide) { i = e / CB_KC; kk = e % CB_KC; } else { kk = e / np; i = e % np; }
                const int k = k0 + kk;
                const size_t ai_ = wide ? (size_t)i * cols + k : (size_t)k * cols + i;
                stg[q] = (q < ept && e < np * CB_KC && i < n && k < m) ? (A32 ? (double)A32[ai_] : A[ai_]) : 0.0;
            }
        };
        auto commit = [&](double* dst) {
#pragma unroll
            for (int q = 0; q < CB_EPT; ++q) {
                const int e = tid + CB_NT * q;
                if (q < ept && e < np * CB_KC) {
                    int kk, i;
                    if (wide) { i = e / CB_KC; kk = e % CB_KC; } else { kk = e / np; i = e % np; }
                    dst[kk * AS_LD + i] = stg[q];
                }
            }
        };
        const int fr_off = lq * AS_LD + lr;
        const int n_pass = (ntile + 3) / 4;
        for (int jb = 0; jb < n_pass; ++jb) {
            // lower tiles of tile columns [4 jb, 4 jb + 4): t-th of them -> (ti, tj); wave w takes t = w, w + CB_NW, ...
            const int tj0 = 4 * jb, tj1 = (tj0 + 4 < ntile) ? tj0 + 4 : ntile;
            int cnt = 0;
            for (int tj = tj0; tj < tj1; ++tj) cnt += ntile - tj;
            int my_ti[CB_TPW], my_tj[CB_TPW];
#pragma unroll
            for (int q = 0; q < CB_TPW; ++q) {
                int t = wvu + CB_NW * q, tj = tj0;
                bool ok = t < cnt;
                while (ok && t >= ntile - tj) { t -= ntile - tj; ++tj; }
                my_tj[q] = ok ? tj : -1;
                my_ti[q] = ok ? tj + t : 0;
            }
            f64x4 acc[CB_TPW];
#pragma unroll
            for (int q = 0; q < CB_TPW; ++q) acc[q] = f64x4{0.0, 0.0, 0.0, 0.0};
            fetch(0);
            commit(Abuf[0]);
            __syncthreads();
            int cur = 0;
            for (int k0 = 0; k0 < m; k0 += CB_KC) {
                const bool more = k0 + CB_KC < m;
                if (more) fetch(k0 + CB_KC);
                const double* Ac = Abuf[cur];
#pragma unroll
                for (int k4 = 0; k4 < CB_KC; k4 += 4) {
                    const double* slab = Ac + k4 * AS_LD + fr_off;
#pragma unroll
                    for (int q = 0; q < CB_TPW; ++q)
                        if (my_tj[q] >= 0)
                            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(slab[my_ti[q] * 16], slab[my_tj[q] * 16], acc[q], 0, 0, 0);
                }
                if (more) commit(Abuf[cur ^ 1]);
                __syncthreads();
                cur ^= 1;
            }
#pragma unroll
            for (int q = 0; q < CB_TPW; ++q)
                if (my_tj[q] >= 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        Gw[(size_t)(my_tj[q] * 16 + lr) * ld + my_ti[q] * 16 + 4 * i + lq] = acc[q][i];
                }
            __syncthreads();
        }
        // right-hand sides into LDS: Bs[o][np]
        double* Bs = reinterpret_cast<double*>(cb_smem);
        __syncthreads();
        for (int e = tid; e < nrhs * np; e += CB_NT) {
            const int o = e / np, i = e % np;
            double v = 0.0;
            if (wide && i < n) {
                const double sc = sp.t_scale ? sp.t_scale[(size_t)g * nrhs + o] : 1.0;
                const double sh = sp.t_shift ? sp.t_shift[(size_t)g * nrhs + o] : 0.0;
                v = Dg[(size_t)i * nrhs + o] * sc + sh;
            }
            if (wide) Bs[e] = v;
        }
        if (!wide) {
            // tall case: right-hand side A^T B, thread -> (o, i), one pass over A (consecutive lanes = consecutive columns)
            for (int e = tid; e < nrhs * np; e += CB_NT) {
                const int o = e / np, i = e % np;
                double acc_b = 0.0;
                if (i < n) {
                    const double sc = sp.t_scale ? sp.t_scale[(size_t)g * nrhs + o] : 1.0;
                    const double sh = sp.t_shift ? sp.t_shift[(size_t)g * nrhs + o] : 0.0;
#pragma unroll 8
                    for (int k = 0; k < m; ++k) {
                        const double a = A32 ? (double)A32[(size_t)k * cols + i] : A[(size_t)k * cols + i];
                        acc_b = fma(a, Dg[(size_t)k * nrhs + o] * sc + sh, acc_b);
                    }
                }
                Bs[e] = acc_b;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    double* Bs = reinterpret_cast<double*>(cb_smem);                    // [nrhs][np]          (32 KB)
    double* P = Bs + (size_t)8 * CB_NMAX;                                // [np][CB_PLD] panel  (68 KB)
    // pivot tolerance from the largest diagonal entry
    double dmax = 0.0;
    for (int i = tid; i < n; i += CB_NT) dmax = fmax(dmax, Gw[(size_t)i * ld + i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dmax = fmax(dmax, __shfl_down(dmax, off));
    if (lane == 0) sh_red[wv] = dmax;
    __syncthreads();
    double piv_tol = 0.0;
    for (int w = 0; w < CB_NW; ++w) piv_tol = fmax(piv_tol, sh_red[w]);
    piv_tol *= 1e-14;

    auto bcast = [](double x, int src) -> double {
        const uint64_t u = __builtin_bit_cast(uint64_t, x);
        const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)u, src);
        const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(u >> 32), src);
        return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
    };
    // ---- phase 3: left-looking panel Cholesky ---------------------------------------------------------
    for (int j = 0; j < ntile; ++j) {
        const int j0 = 16 * j;
        // (1) panel tiles: G(rt, j) - L(rt, :j0) L(j, :j0)^T
        for (int rt = j + wvu; rt < ntile; rt += CB_NW) {
            f64x4 c;
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = Gw[(size_t)(j0 + lr) * ld + rt * 16 + 4 * i + lq];
            const double* la = Gw + (size_t)lq * ld + rt * 16 + lr;       // L[rt*16 + lr][k + lq]
            const double* lb = Gw + (size_t)lq * ld + j0 + lr;            // L[j0 + lr][k + lq]
#pragma unroll 4
            for (int k = 0; k < j0; k += 4)
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(-la[(size_t)k * ld], lb[(size_t)k * ld], c, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) P[(rt * 16 + 4 * i + lq) * CB_PLD + lr] = c[i];
        }
        __syncthreads();
        // (2) diagonal block: factorise and invert in registers (wave 0), as in the LDS kernel
        if (wv == 0) {
            const int r = lane & 15;
            double a[16], x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = P[(j0 + r) * CB_PLD + c];
            double my_invd = 1.0;
            unsigned rejected = 0;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const double v = bcast(a[jj], jj);
                const bool live = j0 + jj < n;
                const bool ok = live && v > piv_tol;
                if (live && !ok) rejected |= 1u << jj;
                const double d = ok ? sqrt(v) : 1.0, inv_d = ok ? 1.0 / d : 0.0;
                if (r == jj) my_invd = ok ? inv_d : 1.0;
                a[jj] = (r == jj) ? d : ((r > jj) ? a[jj] * inv_d : 0.0);
#pragma unroll
                for (int k = jj + 1; k < 16; ++k) {
                    const double lkj = bcast(a[jj], k);
                    if (r >= k) a[k] = fma(-a[jj], lkj, a[k]);
                }
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) P[(j0 + r) * CB_PLD + c] = a[c];
                sh_invd[j0 + r] = my_invd;
            }
            if (rejected && lane == 0) sh_bad = 1;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                double sacc = (i == r) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < i; ++k) sacc = fma(-bcast(a[k], i), x[k], sacc);
                const double idi = bcast(my_invd, i);
                x[i] = (i >= r && !((rejected >> i) & 1u)) ? sacc * idi : 0.0;
            }
            if (lane < 16) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sh_linv[i][r] = x[i];
            }
        }
        __syncthreads();
        // (3) panel solve: L(rt, j) = P(rt) L11^-T
        for (int rt = j + 1 + wvu; rt < ntile; rt += CB_NW) {
            f64x4 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k0 = 0; k0 < 16; k0 += 4)
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(P[(rt * 16 + lr) * CB_PLD + k0 + lq], sh_linv[lr][k0 + lq], c, 0, 0, 0);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) P[(rt * 16 + 4 * i + lq) * CB_PLD + lr] = c[i];
        }
        __syncthreads();
        // (4) the finished panel (rows j0 .. np-1, 16 columns) back to the workspace, column-major
        for (int e = tid; e < (np - j0) * 16; e += CB_NT) {
            const int c = e / (np - j0), rr = j0 + e % (np - j0);
            Gw[(size_t)(j0 + c) * ld + rr] = P[rr * CB_PLD + c];
        }
        __threadfence_block();
        __syncthreads();
    }

    // ---- phase 4: L L^T x = b, one wave per right-hand side; lane holds rows lane + 64 s ---------------
    if (wv < nrhs) {
        double* xb = Bs + (size_t)wv * np;
        constexpr int NS = CB_NMAX / 64;
        const int ns = (np + 63) / 64;
        double b[NS], l[NS], lnext[NS];
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) b[s2] = (s2 < ns && lane + 64 * s2 < np) ? xb[lane + 64 * s2] : 0.0;
        auto load_col = [&](double (&dst)[NS], int jc) {
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                const int i = lane + 64 * s2;
                dst[s2] = (s2 < ns && jc >= 0 && jc < n && i < n) ? Gw[(size_t)jc * ld + i] : 0.0;
            }
        };
        auto pick = [&](const double (&v)[NS], int jc) -> double {        // element jc of the distributed vector
            double own = 0.0;
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) own = ((jc >> 6) == s2) ? v[s2] : own;
            return __shfl(own, jc & 63);
        };
        load_col(l, 0);
        for (int jc = 0; jc < n; ++jc) {                                   // forward: L z = b
            load_col(lnext, jc + 1);
            const double zj = pick(b, jc) * sh_invd[jc];
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                const int i = lane + 64 * s2;
                b[s2] = (i == jc) ? zj : ((i > jc) ? fma(-l[s2], zj, b[s2]) : b[s2]);
            }
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) l[s2] = lnext[s2];
        }
        load_col(l, n - 1);
        for (int jc = n - 1; jc >= 0; --jc) {                              // backward: L^T x = z
            load_col(lnext, jc - 1);
            double part = 0.0;
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                const int i = lane + 64 * s2;
                part = (i > jc && i < n) ? fma(l[s2], b[s2], part) : part;
            }
            part = wave_sum(part);
            const double xj = (pick(b, jc) - part) * sh_invd[jc];
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) b[s2] = (lane + 64 * s2 == jc) ? xj : b[s2];
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) l[s2] = lnext[s2];
        }
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2)
            if (s2 < ns && lane + 64 * s2 < np) xb[lane + 64 * s2] = b[s2];
    }
    __syncthreads();

    // ---- phase 5: W_out ----------------------------------------------------------------------------------
    if (wide) {
        // W_out[o][c] = sum_i A[i][c] alpha[i][o]: every element of A fetched once by 16-byte loads; thread
        // (part, unit) sums a share of the rows for the unit's 2 (float64) or 4 (float32) columns and all
        // right-hand sides, the partial sums meet in LDS
        double* part = P;                                                // [parts][8][cols] behind Bs
        if (sp.part_ok) {
            const int cpt = A32 ? 4 : 2, nunit = cols / cpt;
            int parts = CB_NT / nunit;
            if (parts > 3) parts = 3;
            const int pt = tid / nunit, un = tid - pt * nunit;
            if (pt < parts) {
                const int per = (n + parts - 1) / parts;
                const int i0 = pt * per, i1 = (i0 + per < n) ? i0 + per : n;
                double w[4][8];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int o = 0; o < 8; ++o) w[c][o] = 0.0;
                const size_t ac = (size_t)cpt * un;
#pragma unroll 4
                for (int i = i0; i < i1; ++i) {
                    double a[4];
                    if (A32) {
                        const float4 af = *reinterpret_cast<const float4*>(A32 + ac + (size_t)i * cols);
                        a[0] = (double)af.x; a[1] = (double)af.y; a[2] = (double)af.z; a[3] = (double)af.w;
                    } else {
                        const double2 ad = *reinterpret_cast<const double2*>(A + ac + (size_t)i * cols);
                        a[0] = ad.x; a[1] = ad.y; a[2] = 0.0; a[3] = 0.0;
                    }
#pragma unroll
                    for (int o = 0; o < 8; ++o)
                        if (o < nrhs) {
                            const double al = Bs[o * np + i];
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                if (c < cpt) w[c][o] = fma(a[c], al, w[c][o]);
                        }
                }
#pragma unroll
                for (int o = 0; o < 8; ++o)
                    if (o < nrhs) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (c < cpt) part[(pt * 8 + o) * cols + cpt * un + c] = w[c][o];
                    }
            }
            __syncthreads();
            for (int e = tid; e < nrhs * cols; e += CB_NT) {
                const int o = e / cols, c = e - o * cols;
                double v = part[o * cols + c];
                for (int q = 1; q < parts; ++q) v += part[(q * 8 + o) * cols + c];
                sp.W_out[((size_t)g * nrhs + o) * cols + c] = v;
            }
        } else {
            for (int c = tid; c < cols; c += CB_NT) {
                double w[8];
#pragma unroll
                for (int o = 0; o < 8; ++o) w[o] = 0.0;
                for (int i = 0; i < n; ++i) {
                    const double a = A32 ? (double)A32[(size_t)i * cols + c] : A[(size_t)i * cols + c];
#pragma unroll
                    for (int o = 0; o < 8; ++o)
                        if (o < nrhs) w[o] = fma(a, Bs[o * np + i], w[o]);
                }
#pragma unroll
                for (int o = 0; o < 8; ++o)
                    if (o < nrhs) sp.W_out[((size_t)g * nrhs + o) * cols + c] = w[o];
            }
        }
    } else {
        for (int e = tid; e < nrhs * cols; e += CB_NT) {
            const int o = e / cols, c = e % cols;
            sp.W_out[((size_t)g * nrhs + o) * cols + c] = Bs[o * np + c];
        }
    }
    if (tid == 0) sp.status[g] = sh_bad;
}

static int chol_big_parts(int cols, bool f32) {          // 0 = the partial-sums pass does not apply
    const int cpt = f32 ? 4 : 2;
    if (cols % cpt) return 0;
    const int nunit = cols / cpt;
    if (nunit > CB_NT) return 0;
    const int parts = CB_NT / nunit > 3 ? 3 : CB_NT / nunit;
    return ((size_t)8 * CB_NMAX * 8 + (size_t)parts * 8 * cols * 8 <= 140 * 1024) ? parts : 0;
}
static size_t chol_big_lds_bytes(int cols, bool f32) {
    const size_t gram = 2 * (size_t)CB_KC * (CB_NMAX + 4) * 8;
    size_t tail = (size_t)CB_NMAX * CB_PLD * 8;
    const size_t pp = (size_t)chol_big_parts(cols, f32) * 8 * cols * 8;
    if (pp > tail) tail = pp;
    tail += (size_t)8 * CB_NMAX * 8;
    return gram > tail ? gram : tail;
}

int launch_readout_chol_big(const double* E, const float* E32, const double* D, int n_groups, int T, int transient,
                            int cols, int n_out, const double* t_scale, const double* t_shift,
                            double* W_out, int* status, void* workspace, hipStream_t stream) {
    SolveParams sp;
    const int rows = T - transient;
    const int n = rows < cols ? rows : cols;
    if (n > CB_NMAX || n_out > 8) return -1;
    const size_t lds = chol_big_lds_bytes(cols, E32 != nullptr);
    sp.part_ok = chol_big_parts(cols, E32 != nullptr) > 0 ? 1 : 0;
    sp.E = E; sp.E32 = E32; sp.D = D; sp.n_groups = n_groups; sp.T = T; sp.transient = transient;
    sp.cols = cols; sp.n_out = n_out; sp.t_scale = t_scale; sp.t_shift = t_shift;
    sp.W_out = W_out; sp.status = status;
    sp.work = reinterpret_cast<double*>(workspace); sp.work_stride = chol_big_work_doubles(n);
    sp.wide = rows < cols; sp.m = sp.wide ? cols : rows; sp.n = n; sp.skip = 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(readout_chol_big_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(readout_chol_big_kernel, dim3(n_groups), dim3(CB_NT), lds, stream, sp);
    return (int)hipGetLastError();
}

}  // namespace esn
