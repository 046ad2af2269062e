// float64 recurrence on the float64 matrix pipe: the reference's own arithmetic (pyESN.py:111-125,
// 176-182, 243-255 are float64 throughout) batched over frames, v_mfma_f64_16x16x4_f64.
//
// One workgroup (8 waves) owns Bt = 16*NT sequences for all S timesteps.  LDS holds the B operand
// Zt[frame][k] = [X ; U ; F] in float64 (k contiguous, row stride an odd number of 16-byte slots);
// wave w owns rows [16*MT*w, 16*MT*(w+1)) x all Bt columns: MT x NT accumulator tiles of 16 x 16
// (4 float64 per lane each).  K is consumed in 64-byte groups of 8 elements: lane (c = lane & 15,
// q = lane >> 4) takes bytes [64 g + 16 q, +16) of row c of BOTH operands -- one ds_read_b128 / one
// 16-byte buffer load feed two MFMA k-steps (k is a summation index: every element of the group is
// used exactly once, by the same lane quarter on both sides).  The weights Wext = [W | W_in | W_fb]
// are pre-packed in that fragment order ([row tile][k-group][lane][16 B], esn_pack.hip) and streamed
// from L2 through a fixed two-deep register pipeline; at N_res = 512 a step is 2.2 MB of weights per
// 32 frames against 1088 MFMAs of 64 cycles per SIMD -- matrix-pipe bound, not L2 bound.
//
// Per step (in-step schedule; the float64 pipe leaves the barriers in the noise):
//   G1  P = Wext[:, state k] X_s; readout partials Wout[:, k-slice] X_s, the state k-groups of a
//       16-frame column tile split over the 8/NT waves that share it             [predict]
//   --  partials -> LDS, barrier, 512 threads: Y = yU + sum of partials -> F_s into Zt, unscaled
//       output row to HBM, barrier
//   G2  P += Wext[:, input+feedback k] [U_s ; F_s];  yU_s = Wout[:, input k] U_s
//   --  barrier (all reads of Zt done)
//   E   X_{s+1} = tanh(P) + noise (u - 1/2) -> Zt;  U_{s+1} (, teacher F_{s+1}) -> Zt;  barrier
//       harvest: E row s+1 -> HBM
#include <type_traits>
#include "esn_common.h"

namespace esn {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

// Geometry of the float64 MFMA path (fields of Geometry it uses: Mp, kin, kfb, Kp, Ks, MT, NT, Bt64).
bool f64_mfma_geometry(int n_res, int n_in, int n_out, bool harvest, Geometry* g) {
    if (n_out > 16 || n_res > 1024) return false;
    const int MT = (n_res + 127) / 128;                  // 8 waves x 16 rows per row round
    // accumulators: MT*NT*8 VGPRs <= 64.  The harvest takes 16-frame tiles whatever MT is: a fit has one pilot per
    // trained ESN, a few hundred to a few thousand sequences, and the launch is 137 serial steps long -- twice the
    // workgroups at half the MFMAs per step (512 pilots: 9.9 -> 5.x ms).
    const int NT = (MT <= 4 && !harvest) ? 2 : 1;
    g->NW = 8; g->MT = MT; g->NT = NT;
    g->Mp = 128 * MT;
    g->kin = g->Mp;
    g->kfb = g->kin + round_up(n_in, 4);
    g->Kp = round_up(g->kfb + round_up(n_out, 4), 8);
    g->Ks = g->Kp + (((g->Kp / 2) % 2 == 0) ? 2 : 0);    // odd number of 16-byte slots per row
    g->Bt64 = 16 * NT;
    // one thread per (frame, input column) / (frame, output column) element stages the per-step rows
    if (round_up(n_in, 4) * g->Bt64 > 512 || round_up(n_out, 4) * g->Bt64 > 512) { g->m64 = 0; return false; }
    const size_t lds = (size_t)g->Bt64 * g->Ks * 8 + (harvest ? 0 : (size_t)8 * 256 * 8) + 8 * (size_t)g->Bt64 + 64;
    g->m64 = lds <= 160 * 1024 ? 1 : 0;
    return g->m64 != 0;
}

template <int MT, int NT, bool HARVEST>
__global__ __launch_bounds__(512) void recur_f64_mfma_kernel(RecurParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 8, BT = 16 * NT, NTH = 512;
    constexpr int WPT = NW / NT;                         // waves sharing one 16-frame column tile (readout K split)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const Geometry& g = p.g;
    const int n_res = p.n_res, n_in = p.n_in, n_out = p.n_out;
    const int nkg = g.Kp / 8, nkgS = g.Mp / 8;
    const int kin_p = g.kfb - g.kin, kfb_p = round_up(n_out, 4);
    const int ncols = n_res + n_in;
    const int out_rows = p.S - p.transient;
    const size_t row_bytes = (size_t)g.Ks * 8;

    double* Zt = reinterpret_cast<double*>(smem);                                  // [BT][Ks]
    double* ro_part = Zt + (size_t)BT * g.Ks;                                      // [NT][WPT][16][16] (predict)
    int* tab_fr = reinterpret_cast<int*>(ro_part + (HARVEST ? 0 : 8 * 256));       // [BT]
    int* tab_grp = tab_fr + BT;                                                    // [BT]

    const int slot0 = blockIdx.x * BT;
    const int grp0 = slot_group(p, slot0);
    if (grp0 >= p.n_groups) return;
    const int wset = slot_wset(p, slot0);
    for (int i = tid; i < BT; i += NTH) { int gg; tab_fr[i] = slot_frame(p, slot0 + i, gg); tab_grp[i] = gg; }
    __syncthreads();
    for (int i = tid; i < BT * g.Ks; i += NTH) {
        const int f = i / g.Ks, k = i % g.Ks;
        double v = 0.0;
        if (tab_fr[f] >= 0) {
            const int pg = tab_grp[f];
            if (k < n_res) { if (p.x0) v = p.x0[(size_t)pg * n_res + k]; }
            else if (!HARVEST && k >= g.kfb && k < g.kfb + n_out) { if (p.y0) v = p.y0[(size_t)pg * n_out + (k - g.kfb)]; }
        }
        Zt[i] = v;
    }
    __syncthreads();

    // ---- per-thread staging of the next step's input / teacher element ----------------------
    // thread t < BT*kin_p: frame t / kin_p, input column t % kin_p; t < BT*kfb_p likewise for the teacher
    const int in_f = tid / kin_p, in_c = tid - in_f * kin_p;
    const bool in_mine = tid < BT * kin_p;
    const int in_fr = in_mine ? tab_fr[in_f] : -1;
    const bool in_live = in_fr >= 0 && in_c < n_in;
    double in_sc = 1.0, in_sh = 0.0;
    if (in_live) {
        const int pg = tab_grp[in_f];
        if (p.in_scale) in_sc = p.in_scale[(size_t)pg * n_in + in_c];
        if (p.in_shift) in_sh = p.in_shift[(size_t)pg * n_in + in_c];
    }
    const int t_f = tid / kfb_p, t_c = tid - t_f * kfb_p;
    const bool t_mine = HARVEST && tid < BT * kfb_p;
    const int t_fr = t_mine ? tab_fr[t_f] : -1;
    const bool t_live = t_fr >= 0 && t_c < n_out;
    double t_sc = 1.0, t_sh = 0.0;
    if (t_live) {
        const int pg = tab_grp[t_f];
        if (p.t_scale) t_sc = p.t_scale[(size_t)pg * n_out + t_c];
        if (p.t_shift) t_sh = p.t_shift[(size_t)pg * n_out + t_c];
    }
    auto fetch_in = [&](int s) -> double {           // raw input row of recurrence step s
        const int row = s + p.in_row_off;
        return (in_live && row < p.T_in) ? p.U[((size_t)in_fr * p.T_in + row) * n_in + in_c] : 0.0;
    };
    auto commit_in = [&](int s, double raw) {
        if (!in_mine) return;
        double sv = 0.0;
        if (in_live) {
            sv = raw * in_sc + in_sh;                // (rows past T_in are zeros BEFORE scaling: raw = 0)
            if (HARVEST) p.E[((size_t)in_fr * (p.S + 1) + (s + p.in_row_off)) * ncols + n_res + in_c] = sv;
        }
        Zt[(size_t)in_f * g.Ks + g.kin + in_c] = sv;
    };
    auto fetch_t = [&](int s) -> double { return t_live ? p.D[((size_t)t_fr * (p.S + 1) + s) * n_out + t_c] : 0.0; };
    auto commit_t = [&](double raw) {
        if (t_mine) Zt[(size_t)t_f * g.Ks + g.kfb + t_c] = t_live ? raw * t_sc + t_sh : 0.0;
    };
    commit_in(0, fetch_in(0));
    if (HARVEST) {
        commit_t(fetch_t(0));
        // E row 0 = [0, scale(u[0])] (pyESN.py:179,189); commit_in(0) above wrote row in_row_off = 1's inputs
        for (int f = wave; f < BT; f += NW) {
            const int fr = tab_fr[f];
            if (fr < 0) continue;
            const int pg = tab_grp[f];
            double* e0 = p.E + ((size_t)fr * (p.S + 1)) * ncols;
            for (int cc = lane; cc < ncols; cc += 64) {
                double v = 0.0;
                if (cc >= n_res) {
                    const int ci = cc - n_res;
                    const double sc = p.in_scale ? p.in_scale[(size_t)pg * n_in + ci] : 1.0;
                    const double sh = p.in_shift ? p.in_shift[(size_t)pg * n_in + ci] : 0.0;
                    v = p.U[((size_t)fr * p.T_in) * n_in + ci] * sc + sh;
                }
                e0[cc] = v;
            }
        }
    }
    __syncthreads();

    // ---- operand streams ----------------------------------------------------------------------
    // A: packed [row tile][k-group][lane][16 B]; wave w owns row tiles w*MT .. w*MT+MT-1
    const char* w_img = reinterpret_cast<const char*>(p.packed_w) + (size_t)wset * p.wset_stride + p.w64_off;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(w_img), 0, (int)((size_t)g.Mp * g.Kp * 8), 0x00020000);
    const int lane16 = lane * 16;
    constexpr int OOB = 0x7ffffff0;
    const int w_rt0 = wave * MT;
    auto loadA = [&](u32x4v (&a)[MT], int kg) {          // kg >= nkg: zeros, no traffic
        const bool live = kg < nkg;
        const int voff = live ? lane16 : OOB;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            a[mt] = __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(
                w_rsrc, voff, live ? ((w_rt0 + mt) * nkg + kg) * 1024 : 0, 0));
    };
    const char* bbase = smem + (size_t)c * row_bytes + 16 * q;
    auto loadB = [&](u32x4v (&b)[NT], int kg) {
        kg = kg < nkg ? kg : nkg - 1;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            b[nt] = *reinterpret_cast<const u32x4v*>(bbase + (size_t)nt * 16 * row_bytes + (size_t)kg * 64);
    };
    // readout: this wave's column tile and K slice
    const int ro_ct = wave % NT, ro_part_id = wave / NT;
    const int ro_grp = __builtin_amdgcn_readfirstlane(slot_group(p, slot0 + ro_ct * 16));
    const bool ro_on = !HARVEST && ro_grp < p.n_groups;
    const char* wo_img = reinterpret_cast<const char*>(p.packed_wout) + (size_t)(ro_on ? ro_grp : 0) * p.wout_stride + p.wo64_off;
    const __amdgpu_buffer_rsrc_t wo_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(wo_img), 0, (int)((size_t)16 * g.Kp * 8), 0x00020000);
    const int ro_len = nkgS / WPT;                        // state k-groups per readout slice (even: nkgS = 16 MT)
    const int ro_k0 = ro_len * ro_part_id, ro_k1 = ro_k0 + ro_len;
    const int npos = (nkg + 1) & ~1;                      // k positions per step, padded to even
    auto loadRA = [&](int kg, bool on) -> u32x4v {
        on = on && kg < nkg;
        return __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(wo_rsrc, on ? lane16 : OOB, on ? kg * 1024 : 0, 0));
    };
    const char* robase = smem + (size_t)(ro_ct * 16 + c) * row_bytes + 16 * q;

    auto mma2 = [&](f64x4& acc, const u32x4v& a, const u32x4v& b) {
        const f64x2 av = __builtin_bit_cast(f64x2, a), bv = __builtin_bit_cast(f64x2, b);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc, 0, 0, 0);
    };

    f64x4 yU = {0.0, 0.0, 0.0, 0.0};                     // Wout[:, input k] U_{s-1} of the owned column tile (part 0 only)
    u32x4v abuf[2][MT];
    loadA(abuf[0], 0); loadA(abuf[1], 1);

    for (int s = 0; s < p.S; ++s) {
        const bool have_next = s + 1 < p.S;
        double pre_in = 0.0, pre_t = 0.0;
        if (have_next) { pre_in = fetch_in(s + 1); if (HARVEST) pre_t = fetch_t(s + 1); }

        f64x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f64x4{0.0, 0.0, 0.0, 0.0};
        u32x4v bfr[2][NT];

        // positions [k_lo, k_hi) of the k sequence (both even) out of the two-deep A pipeline: abuf[j]
        // holds position k_lo + j on entry and is refilled with the position two ahead right after its
        // last MFMA.  The sequence is padded to an even number of positions per step (npos; a padding
        // position loads zeros: out-of-range buffer offset, no traffic), so the buffers keep their
        // parity across steps and the look-ahead runs through phase E into the next step's first groups.
        // Nothing in the loop is conditional (a branch around a load costs s_waitcnt vmcnt(0) at the
        // loop head): the readout rides only in the first call, over this wave's K slice.
        auto gemm_groups = [&](int k_lo, int k_hi, auto ro_tag, f64x4& racc, int rk0, bool ro_s) {
            constexpr bool WITH_RO = decltype(ro_tag)::value;
            loadB(bfr[0], k_lo);
            u32x4v ra = {0, 0, 0, 0}, rb;
            if (WITH_RO) ra = loadRA(rk0, ro_s);
            for (int kg = k_lo; kg < k_hi; kg += 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int k = kg + j;
                    loadB(bfr[j ^ 1], k + 1);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) mma2(acc[mt][nt], abuf[j][mt], bfr[j][nt]);
                    if (WITH_RO) {
                        const int rk = rk0 + (k - k_lo);
                        rb = *reinterpret_cast<const u32x4v*>(robase + (size_t)rk * 64);
                        mma2(racc, ra, rb);
                        ra = loadRA(rk + 1, ro_s && k + 1 < k_hi);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const int pn = k + 2;
                    loadA(abuf[j], pn < npos ? pn : (have_next ? pn - npos : nkg));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        const std::integral_constant<bool, true> with_ro;
        const std::integral_constant<bool, false> no_ro;
        f64x4 dummy = {0.0, 0.0, 0.0, 0.0};

        if constexpr (HARVEST) {
            gemm_groups(0, npos, no_ro, dummy, 0, false);
        } else {
            // ===== G1: state k-groups + readout partial of Y for X_s ============================
            f64x4 racc = (ro_part_id == 0) ? yU : f64x4{0.0, 0.0, 0.0, 0.0};
            gemm_groups(0, ro_len, with_ro, racc, ro_k0, ro_on && s > 0);
            gemm_groups(ro_len, nkgS, no_ro, dummy, 0, false);
            if (s > 0) {
                double* rp = ro_part + ((size_t)(ro_ct * WPT + ro_part_id) * 16) * 16;    // [row][col]
#pragma unroll
                for (int i = 0; i < 4; ++i) rp[(q + 4 * i) * 16 + c] = racc[i];
            }
            __syncthreads();
            if (s > 0) {
                // thread (frame f, output o): Y = sum of partials; F_s -> Zt; unscaled row s-1 -> HBM
                const int f = tid >> 4, o = tid & 15;
                if (f < BT && o < kfb_p) {
                    const int ct = f >> 4, fc = f & 15;
                    double y = 0.0;
#pragma unroll
                    for (int pp = 0; pp < WPT; ++pp) y += ro_part[((size_t)(ct * WPT + pp) * 16 + o) * 16 + fc];
                    Zt[(size_t)f * g.Ks + g.kfb + o] = (o < n_out) ? y : 0.0;
                    const int fr = tab_fr[f], orow = s - 1 - p.transient;
                    if (fr >= 0 && o < n_out && orow >= 0) {
                        const int pg = tab_grp[f];
                        const double sc = p.t_scale ? p.t_scale[(size_t)pg * n_out + o] : 1.0;
                        const double sh = p.t_shift ? p.t_shift[(size_t)pg * n_out + o] : 0.0;
                        p.Y[((size_t)fr * out_rows + orow) * n_out + o] = (y - sh) / sc;
                    }
                }
                __syncthreads();
            }
            // ===== G2: input + feedback k-groups; yU_s for the next readout =========================
            gemm_groups(nkgS, npos, no_ro, dummy, 0, false);
            if (ro_part_id == 0) {
                yU = f64x4{0.0, 0.0, 0.0, 0.0};
                for (int kg = nkgS; kg < nkg; ++kg) {           // feedback columns of Wout carry zeros
                    const u32x4v ra = loadRA(kg, ro_on);
                    const u32x4v rb = *reinterpret_cast<const u32x4v*>(robase + (size_t)kg * 64);
                    mma2(yU, ra, rb);
                }
            }
        }
        __syncthreads();                                         // every wave has finished reading Z_s

        // ===== E: X_{s+1} = tanh(P) + noise (u - 1/2) ===============================================
        if (have_next) { commit_in(s + 1, pre_in); if (HARVEST) commit_t(pre_t); }
        // short series when every pre-activation of the wave is small (the OFDM workload stays below 0.15),
        // library tanh otherwise -- a wave-uniform choice, so neither path is predicated
        double amax = 0.0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) amax = fmax(amax, fabs(acc[mt][nt][i]));
        const bool small = __all(amax <= TANH64_SERIES_MAX) != 0;   // NaN compares false -> library path
        const bool leaky = p.leak != 1.0;
        auto activate = [&](auto series_tag) {
            constexpr bool SERIES = decltype(series_tag)::value;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int fcol = nt * 16 + c;
                const int fr = tab_fr[fcol];
                uint32_t key = 0;
                const double* nz = nullptr;
                if (p.noise_mode == ESN_NOISE_COUNTER && fr >= 0) key = noise_key(p.seed, (uint32_t)fr + p.frame_off, (uint32_t)s);
                if (p.noise_mode == ESN_NOISE_TENSOR && fr >= 0) nz = p.noise_u + ((size_t)fr * p.S + s) * n_res;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = (w_rt0 + mt) * 16 + q + 4 * i;
                        double x = SERIES ? tanh_f64_series(acc[mt][nt][i]) : tanh(acc[mt][nt][i]);
                        if (leaky) {                       // (extension; wave-uniform)
                            const double xo = Zt[(size_t)fcol * g.Ks + row];
                            x = fma(p.leak, x - xo, xo);
                        }
                        if (fr >= 0 && row < n_res) {
                            if (p.noise_mode == ESN_NOISE_COUNTER) x += p.noise * ((double)noise_uniform(key, row) - 0.5);
                            else if (p.noise_mode == ESN_NOISE_TENSOR) x += p.noise * (nz[row] - 0.5);
                        }
                        Zt[(size_t)fcol * g.Ks + row] = x;
                    }
                }
            }
        };
        if (small) activate(std::true_type{}); else activate(std::false_type{});
        __syncthreads();                                         // X_{s+1}, U_{s+1} (, F_{s+1}) complete
        if (HARVEST) {
            // E row s+1, state columns: wave w copies frames w, w+8, ...
            for (int f = wave; f < BT; f += NW) {
                const int fr = tab_fr[f];
                if (fr < 0) continue;
                double* er = p.E + ((size_t)fr * (p.S + 1) + (s + 1)) * ncols;
                const double* zr = Zt + (size_t)f * g.Ks;
                if ((ncols & 1) == 0) {
                    for (int cc = 2 * lane; cc < n_res; cc += 128) {
                        if (cc + 1 < n_res) *reinterpret_cast<f64x2*>(er + cc) = f64x2{zr[cc], zr[cc + 1]};
                        else er[cc] = zr[cc];
                    }
                } else {
                    for (int cc = lane; cc < n_res; cc += 64) er[cc] = zr[cc];
                }
            }
        }
    }
    if (!HARVEST) {
        // final readout: Y for X_S = yU_{S-1} + Wout_x X_S  -> output row S-1
        f64x4 racc = (ro_part_id == 0) ? yU : f64x4{0.0, 0.0, 0.0, 0.0};
        for (int kg = ro_k0; kg < ro_k1; ++kg) {
            const u32x4v ra = loadRA(kg, ro_on);
            const u32x4v rb = *reinterpret_cast<const u32x4v*>(robase + (size_t)kg * 64);
            mma2(racc, ra, rb);
        }
        double* rp = ro_part + ((size_t)(ro_ct * WPT + ro_part_id) * 16) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) rp[(q + 4 * i) * 16 + c] = racc[i];
        __syncthreads();
        const int f = tid >> 4, o = tid & 15;
        const int orow = p.S - 1 - p.transient;
        if (f < BT && o < n_out && orow >= 0) {
            const int ct = f >> 4, fc = f & 15, fr = tab_fr[f];
            if (fr >= 0) {
                double y = 0.0;
#pragma unroll
                for (int pp = 0; pp < WPT; ++pp) y += ro_part[((size_t)(ct * WPT + pp) * 16 + o) * 16 + fc];
                const int pg = tab_grp[f];
                const double sc = p.t_scale ? p.t_scale[(size_t)pg * n_out + o] : 1.0;
                const double sh = p.t_shift ? p.t_shift[(size_t)pg * n_out + o] : 0.0;
                p.Y[((size_t)fr * out_rows + orow) * n_out + o] = (y - sh) / sc;
            }
        }
    }
}

template <int MT, int NT>
static int launch_mt(const RecurParams& p, hipStream_t stream) {
    const size_t lds = (size_t)p.g.Bt64 * p.g.Ks * 8 + (p.harvest ? 0 : (size_t)8 * 256 * 8) + 8 * (size_t)p.g.Bt64 + 64;
    hipError_t e;
    if (p.harvest) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(recur_f64_mfma_kernel<MT, NT, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((recur_f64_mfma_kernel<MT, NT, true>), dim3(p.n_tiles), dim3(512), lds, stream, p);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(recur_f64_mfma_kernel<MT, NT, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((recur_f64_mfma_kernel<MT, NT, false>), dim3(p.n_tiles), dim3(512), lds, stream, p);
    }
    return (int)hipGetLastError();
}

int launch_recur_f64_mfma(const RecurParams& p, hipStream_t stream) {
    const Geometry& g = p.g;
    if (g.NT == 2) {
        switch (g.MT) {
            case 1: return launch_mt<1, 2>(p, stream);
            case 2: return launch_mt<2, 2>(p, stream);
            case 3: return launch_mt<3, 2>(p, stream);
            case 4: return launch_mt<4, 2>(p, stream);
        }
    } else if (g.NT == 1) {
        switch (g.MT) {
            case 1: return launch_mt<1, 1>(p, stream);
            case 2: return launch_mt<2, 1>(p, stream);
            case 3: return launch_mt<3, 1>(p, stream);
            case 4: return launch_mt<4, 1>(p, stream);
            case 5: return launch_mt<5, 1>(p, stream);
            case 6: return launch_mt<6, 1>(p, stream);
            case 7: return launch_mt<7, 1>(p, stream);
            case 8: return launch_mt<8, 1>(p, stream);
        }
    }
    return -1;
}

}  // namespace esn
