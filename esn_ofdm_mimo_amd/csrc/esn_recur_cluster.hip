// float64 recurrence of ONE sequence -- the reference's own call pattern (one `fit` / `predict` per OFDM
// frame, pyESN.py:176-182,243-255) -- with the reservoir matrix RESIDENT IN LDS.
//
// A lone sequence is a chain of T dependent matrix-vector products; streaming the 2 MB float64 matrix
// from L2 into one CU every step (esn_recur_f64.hip: ~36 us per step) is bound by that CU's L2 port.
// Here a cluster of C co-resident workgroups holds the matrix for all T steps: workgroup c keeps rows
// [cR, cR + R) of Wext = [W | W_in | W_feedb] in its LDS (K-major, R = 32 rows = 137 KB at N_res = 512,
// C = 16) and per step
//     gathers   x_{t-1} (and the read-out partials) of all C workgroups from L2,
//     computes  its R rows of  x_t = tanh(Wext [x_{t-1}; u_t; y_{t-1}]) + noise,
//               its partial    W_out[:, rows] x_t                       (predict),
//     publishes both.
// The hand-off is the guide's data-tagged granule form (MI355X_MICROARCH.md, Persistent kernels:
// handoff-1to1 / allgather): every double travels as TWO naturally aligned 8-byte words {32 data bits,
// step tag}, each written by one agent-scope (`sc1`) store and polled by agent-scope loads until the tag
// reads the awaited step -- no flag, no fence, no grid barrier, no relaunch; the buffer is double-buffered
// by step parity (a workgroup can only overwrite parity p after it has consumed every slice of the step in
// between, which every other workgroup produces only after consuming the previous parity-p slices).
// Every spin is bounded: a workgroup that waits longer than ~2 s of polls raises the error word and leaves,
// and so does everybody else (the host call then reports the failure instead of hanging).
// N_res small enough for one workgroup (C = 1) keeps the state in LDS and exchanges nothing.
#include "esn_common.h"

namespace esn {

struct ClusterGeom {
    int R;        // rows per workgroup (power of two, <= 256)
    int C;        // workgroups
    int K;        // n_res + n_in + n_out
    int per;      // doubles a workgroup publishes per step: R (+ n_out when predicting)
};

constexpr int CL_NT = 256;
constexpr int CL_NG = 12;                         // doubles a thread may own in a gather: C * per <= CL_NG * CL_NT
constexpr uint32_t CL_SPIN_LIMIT = 1u << 22;     // polls per wait before giving up (each poll >= ~0.5 us)

static bool cluster_geometry(int n_res, int n_in, int n_out, bool harvest, ClusterGeom* cg) {
    const int K = n_res + n_in + n_out;
    int R = 256;
    while (R > 1 && (size_t)K * R * 8 > 138 * 1024) R >>= 1;
    if ((size_t)K * R * 8 > 138 * 1024) return false;
    while (R / 2 >= n_res && R > 1) R >>= 1;                 // do not hold more rows than there are
    const int C = (n_res + R - 1) / R;
    if (C > 64 || n_out > 16 || n_in > 64) return false;
    cg->R = R; cg->C = C; cg->K = K; cg->per = R + (harvest ? 0 : n_out);
    if (C * cg->per > CL_NG * CL_NT) return false;
    // LDS: matrix + operand vector + gathered partials + small tables; checked by the launcher
    return true;
}
static size_t cluster_lds_bytes(const ClusterGeom& cg, int n_in, int n_out) {
    return sizeof(double) * ((size_t)cg.K * cg.R            // Ws[k][R]
                             + cg.K + 8                       // v = [x ; u ; fb]
                             + (size_t)CL_NT                  // matvec partial sums
                             + (size_t)cg.C * 16              // gathered read-out partials [C][n_out]
                             + (size_t)16 * cg.R              // W_out[:, my rows]
                             + (size_t)16 * n_in + 2 * n_in + cg.R + 64);      // + x slice of the step
}
size_t cluster_workspace_bytes(int n_res, int n_in, int n_out, bool harvest) {
    ClusterGeom cg;
    if (!cluster_geometry(n_res, n_in, n_out, harvest, &cg)) return 0;
    if (cluster_lds_bytes(cg, n_in, n_out) > 158 * 1024) return 0;
    return 2 * (size_t)cg.C * cg.per * 2 * 8 + 64;           // two parities of granules + the error word
}

template <bool HARVEST>
__global__ __launch_bounds__(CL_NT) void recur_cluster_kernel(RecurParams p, ClusterGeom cg, unsigned long long* xch) {
    extern __shared__ __attribute__((aligned(16))) char csm[];
    const int n_res = p.n_res, n_in = p.n_in, n_out = p.n_out;
    const int R = cg.R, C = cg.C, K = cg.K, per = cg.per;
    double* Ws = reinterpret_cast<double*>(csm);              // [K][R]
    double* v = Ws + (size_t)K * R;                           // [K]: x (n_res) | u (n_in) | fb (n_out)
    double* psum = v + K + 8;                                 // [CL_NT]
    double* pbuf = psum + CL_NT;                              // [C][16] read-out partials of the last step
    double* wo_rows = pbuf + (size_t)C * 16;                  // [16][R]   W_out[o][cR + r]
    double* wo_in = wo_rows + (size_t)16 * R;                 // [16][n_in] W_out[o][n_res + i]
    double* u_prev = wo_in + (size_t)16 * n_in;               // [n_in] scaled inputs of the previous step
    double* xs = u_prev + n_in;                               // [R] my rows of x_{s+1}
    const int tid = threadIdx.x;
    const int c = blockIdx.x, row0 = c * R;
    const int ncols = n_res + n_in;
    const int out_rows = p.S - p.transient;
    unsigned int* err = reinterpret_cast<unsigned int*>(xch + (size_t)2 * C * per * 2);
    __shared__ int sh_dead;
    if (tid == 0) sh_dead = 0;

    // ---- resident operands ---------------------------------------------------------------------------
    const double* Wk = reinterpret_cast<const double*>(p.packed_w);           // [K][n_res], K-major
    for (int i = tid; i < K * R; i += CL_NT) {
        const int k = i / R, r = i - k * R;
        Ws[i] = (row0 + r < n_res) ? Wk[(size_t)k * n_res + row0 + r] : 0.0;
    }
    if (!HARVEST) {
        const double* wo = reinterpret_cast<const double*>(p.packed_wout);    // plain W_out [n_out][ncols]
        for (int i = tid; i < 16 * R; i += CL_NT) {
            const int o = i / R, r = i - o * R;
            wo_rows[i] = (o < n_out && row0 + r < n_res) ? wo[(size_t)o * ncols + row0 + r] : 0.0;
        }
        for (int i = tid; i < 16 * n_in; i += CL_NT) {
            const int o = i / n_in, ci = i - o * n_in;
            wo_in[i] = (o < n_out) ? wo[(size_t)o * ncols + n_res + ci] : 0.0;
        }
    }
    for (int i = tid; i < K + 8; i += CL_NT) v[i] = 0.0;
    for (int i = tid; i < n_in; i += CL_NT) u_prev[i] = 0.0;
    __syncthreads();
    if (C == 1) {                                             // the whole state is ours: start from x0 locally
        for (int i = tid; i < n_res; i += CL_NT) v[i] = p.x0 ? p.x0[i] : 0.0;
    }
    auto scaled_input = [&](int row, int ci) -> double {
        const double raw = (row < p.T_in) ? p.U[(size_t)row * n_in + ci] : 0.0;
        const double sc = p.in_scale ? p.in_scale[ci] : 1.0, sh = p.in_shift ? p.in_shift[ci] : 0.0;
        return raw * sc + sh;
    };
    if (HARVEST && c == 0) {                                  // E row 0 = [0, scale(u[0])]  (pyESN.py:179,189)
        for (int i = tid; i < ncols; i += CL_NT)
            p.E[i] = (i >= n_res) ? scaled_input(0, i - n_res) : 0.0;
    }

    // granule helpers: a double = two 8-byte words {tag << 32 | half}
    auto publish = [&](int step_tag, int idx, double val) {   // idx < per
        const unsigned long long bits = __builtin_bit_cast(unsigned long long, val);
        unsigned long long* dst = xch + (((size_t)(step_tag & 1) * C + c) * per + idx) * 2;
        const unsigned long long tag = (unsigned long long)(unsigned)step_tag << 32;
        __hip_atomic_store(dst, tag | (bits & 0xffffffffULL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // gather every workgroup's slice of step `step_tag` into v[0:n_res] and pbuf; false on time-out.  A thread owns up
    // to CL_NG doubles (two granules each): it issues ALL its loads, then checks the tags, and re-polls only what has
    // not arrived -- one L2 round trip per step when the producers are on time, not one per granule.
    auto gather = [&](int step_tag) -> bool {
        const int total = C * per;
        unsigned pending = 0;
#pragma unroll
        for (int j = 0; j < CL_NG; ++j)
            if (tid + j * CL_NT < total) pending |= 1u << j;
        uint32_t spins = 0;
        bool ok = true;
        while (pending) {
            unsigned long long lo[CL_NG], hi[CL_NG];
#pragma unroll
            for (int j = 0; j < CL_NG; ++j)
                if (pending & (1u << j)) {
                    const unsigned long long* src = xch + ((size_t)(step_tag & 1) * total + tid + j * CL_NT) * 2;
                    lo[j] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hi[j] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
            for (int j = 0; j < CL_NG; ++j)
                if ((pending & (1u << j)) && (unsigned)(lo[j] >> 32) == (unsigned)step_tag &&
                    (unsigned)(hi[j] >> 32) == (unsigned)step_tag) {
                    const double val = __builtin_bit_cast(double, (hi[j] << 32) | (lo[j] & 0xffffffffULL));
                    const int i = tid + j * CL_NT;
                    const int cc = i / per, jj = i - cc * per;
                    if (jj < R) { if (cc * R + jj < n_res) v[cc * R + jj] = val; }
                    else pbuf[cc * 16 + (jj - R)] = val;
                    pending &= ~(1u << j);
                }
            if (pending) {
                if (++spins > CL_SPIN_LIMIT ||
                    ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (!ok) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_dead = 1;
        }
        __syncthreads();
        return sh_dead == 0;
    };

    // step tag t+1 carries x_t (t = 0: the start state) and, when predicting, the partials of y_t
    if (C > 1) {
        for (int r = tid; r < R; r += CL_NT)
            publish(1, r, (p.x0 && row0 + r < n_res) ? p.x0[row0 + r] : 0.0);
        if (!HARVEST)                                         // y_0 = lastoutput (continuation) or 0: workgroup 0 carries it whole
            for (int o = tid; o < n_out; o += CL_NT) publish(1, R + o, (c == 0 && p.y0) ? p.y0[o] : 0.0);
    } else if (!HARVEST) {
        for (int o = tid; o < n_out; o += CL_NT) pbuf[o] = p.y0 ? p.y0[o] : 0.0;
    }

    const int kpart = tid / R, rr = tid - kpart * R;          // matvec: thread (k-part, row)
    const int nkp = CL_NT / R;
    // per-step operands from global memory (input row, teacher row, noise row) are fetched ONE STEP AHEAD into
    // registers: their ~1 us of load latency then runs under the previous step's gather instead of in front of
    // the matrix-vector product
    auto fetch_u = [&](int s) -> double { return (tid < n_in && s < p.S) ? scaled_input(s + p.in_row_off, tid) : 0.0; };
    auto fetch_d = [&](int s) -> double {
        const int o = tid - 64;
        if (!HARVEST || o < 0 || o >= n_out || s >= p.S || !p.teacher_forcing) return 0.0;
        const double sc = p.t_scale ? p.t_scale[o] : 1.0, sh = p.t_shift ? p.t_shift[o] : 0.0;
        return p.D[(size_t)s * n_out + o] * sc + sh;
    };
    auto fetch_nz = [&](int s) -> double {
        const int row = row0 + tid;
        return (p.noise_mode == ESN_NOISE_TENSOR && tid < R && row < n_res && s < p.S) ? p.noise_u[(size_t)s * n_res + row] : 0.5;
    };
    double u_cur = fetch_u(0), d_cur = fetch_d(0), nz_cur = fetch_nz(0);
    // loop-invariant global operands of the output row (no load in front of a barrier inside the loop)
    const double y_sc = (tid < n_out && p.t_scale) ? p.t_scale[tid] : 1.0;
    const double y_sh = (tid < n_out && p.t_shift) ? p.t_shift[tid] : 0.0;
    const int ro_o = tid / R;                                 // partial read-out: thread (output o, row r), R <= 64
    const bool ro_par = R <= 64;
    for (int s = 0; s <= p.S; ++s) {
        if (HARVEST && s == p.S) break;                       // (the last state is in E; nobody reads it back)
        const double u_nx = fetch_u(s + 1), d_nx = fetch_d(s + 1), nz_nx = fetch_nz(s + 1);
        // ---- x_s and the partials of y_s ---------------------------------------------------------------
        if (C > 1) { if (!gather(s + 1)) return; }
        else __syncthreads();
        // y_s = sum of partials + W_out[:, inputs] u_s   (pyESN.py:252; s = 0: the start feedback as is)
        if (!HARVEST && tid < n_out) {
            double y = 0.0, y2 = 0.0;
#pragma unroll 8
            for (int cc = 0; cc < C; ++cc) y += pbuf[cc * 16 + tid];
            if (s > 0) {
#pragma unroll 8
                for (int ci = 0; ci < n_in; ++ci) y2 = fma(wo_in[tid * n_in + ci], u_prev[ci], y2);
            }
            y += y2;
            v[n_res + n_in + tid] = y;
            if (c == 0 && s > 0 && s - 1 >= p.transient)                   // output row s-1, unscaled (pyESN.py:255)
                p.Y[(size_t)(s - 1 - p.transient) * n_out + tid] = (y - y_sh) / y_sc;
        }
        if (s == p.S) break;
        // inputs of this step (row s + in_row_off); harvest: the teacher row s as feedback
        if (tid < n_in) {
            const double u = u_cur;
            v[n_res + tid] = u;
            u_prev[tid] = u;
            if (HARVEST && c == 0) p.E[(size_t)(s + 1) * ncols + n_res + tid] = u;
        }
        if (HARVEST && tid >= 64 && tid < 64 + n_out) v[n_res + n_in + (tid - 64)] = d_cur;
        __syncthreads();
        // ---- my R rows of Wext v ------------------------------------------------------------------------
        // (eight operand pairs in flight and four accumulators: left to itself the compiler emits one dependent
        //  LDS-read / FMA pair per trip -- 67 LDS latencies = 4 us of the step)
        double acc;
        {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            const double* wp = Ws + (size_t)kpart * R + rr;
            const double* vp = v + kpart;
            const int sw = nkp * R, sv = nkp;
            const int n_it = (K - kpart + nkp - 1) / nkp;
            int i = 0;
            for (; i + 8 <= n_it; i += 8) {
                double w[8], x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { w[j] = wp[j * sw]; x[j] = vp[j * sv]; }
                a0 = fma(w[0], x[0], a0); a1 = fma(w[1], x[1], a1); a2 = fma(w[2], x[2], a2); a3 = fma(w[3], x[3], a3);
                a0 = fma(w[4], x[4], a0); a1 = fma(w[5], x[5], a1); a2 = fma(w[6], x[6], a2); a3 = fma(w[7], x[7], a3);
                wp += 8 * sw; vp += 8 * sv;
            }
            for (; i < n_it; ++i) { a0 = fma(wp[0], vp[0], a0); wp += sw; vp += sv; }
            acc = (a0 + a1) + (a2 + a3);
        }
        psum[tid] = acc;
        __syncthreads();
        double xn = 0.0;
        if (tid < R) {
            double z = 0.0;
#pragma unroll 8
            for (int q = 0; q < nkp; ++q) z += psum[q * R + tid];
            const int row = row0 + tid;
            // (wave-uniform vote: the OFDM workload keeps |z| < 0.15, where the 12-instruction series is exact to 2e-18)
            xn = __all(fabs(z) <= TANH64_SERIES_MAX) ? tanh_f64_series(z) : tanh(z);
            if (p.leak != 1.0 && row < n_res) { const double xo = v[row]; xn = fma(p.leak, xn - xo, xo); }   // (extension)
            if (p.noise_mode == ESN_NOISE_TENSOR && row < n_res)
                xn += p.noise * (nz_cur - 0.5);
            else if (p.noise_mode == ESN_NOISE_COUNTER)
                xn += p.noise * ((double)noise_uniform(noise_key(p.seed, p.frame_off, (uint32_t)s), (uint32_t)row) - 0.5);
            if (row >= n_res) xn = 0.0;
            if (HARVEST && row < n_res) p.E[(size_t)(s + 1) * ncols + row] = xn;
            if (C > 1) publish(s + 2, tid, xn);
        }
        if (C == 1) {
            __syncthreads();                                  // everybody has read v: the state may be replaced
            if (tid < R && tid < n_res) v[tid] = xn;
        }
        if (!HARVEST) {
            // partial read-out of my rows: W_out[:, rows] x_{s+1}
            if (tid < R) xs[tid] = xn;
            __syncthreads();
            if (ro_par) {
                // thread (o, r) holds one product; the R lanes of an output sum by shuffles (R <= 64: inside a wave)
                for (int o0 = 0; o0 < n_out; o0 += CL_NT / R) {
                    const int o = o0 + ro_o;
                    double part = (o < n_out) ? wo_rows[o * R + rr] * xs[rr] : 0.0;
                    for (int off = R >> 1; off > 0; off >>= 1) part += __shfl_down(part, off, R);
                    if (rr == 0 && o < n_out) {
                        if (C > 1) publish(s + 2, R + o, part);
                        else pbuf[o] = part;
                    }
                }
            } else if (tid < n_out) {
                double part = 0.0;
                for (int r = 0; r < R; ++r) part = fma(wo_rows[tid * R + r], xs[r], part);
                if (C > 1) publish(s + 2, R + tid, part);
                else pbuf[tid] = part;
            }
        }
        u_cur = u_nx; d_cur = d_nx; nz_cur = nz_nx;
    }
}

bool cluster_applies(int precision, const RecurParams& p) {
    if (precision != ESN_F64 || p.n_frames != 1 || p.n_groups != 1 || p.n_wsets != 1) return false;
    return cluster_workspace_bytes(p.n_res, p.n_in, p.n_out, p.harvest != 0) > 0;
}

int launch_recur_cluster(const RecurParams& p, void* workspace, hipStream_t stream) {
    ClusterGeom cg;
    if (!cluster_geometry(p.n_res, p.n_in, p.n_out, p.harvest != 0, &cg)) return -1;
    const size_t lds = cluster_lds_bytes(cg, p.n_in, p.n_out);
    const size_t ws = cluster_workspace_bytes(p.n_res, p.n_in, p.n_out, p.harvest != 0);
    hipError_t e = hipMemsetAsync(workspace, 0, ws, stream);           // step tags start at 1: zero = nothing published
    if (e != hipSuccess) return (int)e;
    unsigned long long* xch = reinterpret_cast<unsigned long long*>(workspace);
    if (p.harvest) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(recur_cluster_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(recur_cluster_kernel<true>, dim3(cg.C), dim3(CL_NT), lds, stream, p, cg, xch);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(recur_cluster_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(recur_cluster_kernel<false>, dim3(cg.C), dim3(CL_NT), lds, stream, p, cg, xch);
    }
    return (int)hipGetLastError();
}

}  // namespace esn
