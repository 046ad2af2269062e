// Batched state harvest of ESN.fit (pyESN.py:176-182) at 257..512 reservoir units, fp16 / bf16, with the reservoir
// matrix RESIDENT ON CHIP.
//
// A fit has ONE training sequence per trained ESN (2048 pilots at the benchmark size where predict has 153 600
// frames).  The persistent harvest kernel (esn_recur_mfma_impl.h, HARVEST) gives a tile of 32 pilots to one workgroup,
// which re-streams the whole 557 KB weight image from L2 every timestep: 64 workgroups on 256 CUs, each bound by its
// CU's L2 port (19 k cycles per step, 5 % of the MFMA peak); more, smaller tiles only multiply the aggregate stream
// (DESIGN 3.1).  Here a CLUSTER of C co-resident workgroups owns 8 C pilots for all T - 1 steps: member c keeps rows
// [512 c / C, 512 (c + 1) / C) of Wext = [W | W_in | W_feedb] -- fragments of the 16x16x32 weight image of
// esn_pack_weights, as they are -- in REGISTERS, the fragment-major state image of the cluster's pilots in LDS (the
// layout of esn_recur_skew16_impl.h), and per step
//     multiplies  its rows x the pilots x 544 k on v_mfma_f32_16x16x32 (4 waves, 68 MFMAs each),
//     activates   (tanh + state noise -> operand type), stores its rows of E[:, s + 1],
//     publishes   its 8 KB slice of X_{s+1} and gathers the other C - 1 -- nothing else moves: no weight traffic.
// The hand-off is the data-tagged granule form of esn_recur_cluster.hip: an 8-byte granule carries four state values
// AND the step tag, in the one bit of every half that |x| < 2 leaves free (bit 14: tanh + noise never reaches 2), is
// written by one agent-scope store and polled by agent-scope loads until the tag reads the awaited step; two
// buffers by step parity; every spin is bounded (a timed-out workgroup raises the error word in the last 64 bytes of
// the workspace and leaves, and so does everybody else).  Clusters are laid out so that their members are consecutive
// blocks of one XCD (block id mod 8).
// Arithmetic, noise stream (counter noise keyed by (seed, global pilot, step, row)) and the rounding of the states to
// the operand type are those of the persistent harvest kernel; only the summation order inside a dot product differs.
#include "esn_recur_mfma_impl.h"

namespace esn {

// Cluster shapes: C workgroups (row slices of 512 / C rows) own P = 8 C pilots -- every member publishes 8 KB per step
// whatever C is, and gathers (C - 1) x 8 KB.  C = 2 (256 rows and 16 pilots per workgroup, ONE peer, 8 KB gathered) is
// the default: the step is the hand-off, and the hand-off is its bytes.  C = 8 (64 rows x 64 pilots, 56 KB gathered:
// the first version) and C = 4 stay instantiated for A/B runs (debug knob hcluster = 8 / 4).
constexpr int HC_NT = 256;                        // 4 waves: NWR row groups x NWC pilot-tile groups
constexpr int HC_NKK = 17;
constexpr uint32_t HC_SPIN_LIMIT = 1u << 22;
constexpr unsigned long long HC_TAGMASK = 0x4000400040004000ULL;

// clusters: the pilots of one weight set (groups g with (g + rot) % n_wsets == w; all of them when the reservoir is
// shared) are cut into runs of 8 C; cluster k serves set k % n_wsets, run k / n_wsets
static inline int hc_clusters(int n_pilots, int C, int n_wsets) {
    const int per_set = (n_pilots + n_wsets - 1) / n_wsets;
    return n_wsets * ((per_set + 8 * C - 1) / (8 * C));
}
size_t harvest_cluster_workspace_bytes(int n_pilots, int C, int n_wsets) {
    return (size_t)hc_clusters(n_pilots, C, n_wsets) * 2 * C * 8 * 1024 + 64;   // two parities x C members x 8 KB, + error word
}

__device__ __forceinline__ unsigned long long hc_tag_bits(int tag) {         // tag 1..15 -> bit 14 of each of four halves
    return ((unsigned long long)(tag & 1) << 14) | ((unsigned long long)((tag >> 1) & 1) << 30) |
           ((unsigned long long)((tag >> 2) & 1) << 46) | ((unsigned long long)((tag >> 3) & 1) << 62);
}

template <typename TR, int NOISE, int C>
__global__ __launch_bounds__(HC_NT) void harvest_cluster_kernel(RecurParams p, int n_clusters, unsigned long long* xch) {
    constexpr int ROWS = 512 / C;                  // rows of Wext per member
    constexpr int RT = ROWS / 16, CT = C / 2;      // 16-row tiles per member, 16-pilot tiles per cluster
    constexpr int P = 16 * CT;                     // pilots per cluster
    constexpr int NWC = CT < 2 ? CT : 2, NWR = 4 / NWC;
    constexpr int WR = RT / NWR, WC = CT / NWC;    // row tiles x pilot tiles per wave
    constexpr int KPM = ROWS / 32;                 // 32-k groups a member produces (blocks per pilot tile it publishes)
    constexpr int NG = (C - 1) * 2;                // 16-byte chunks a thread gathers per step
    constexpr int ZF_BYTES = CT * HC_NKK * 1024;
    // weight fragments of groups [0, KREG) live in registers, the last HC_NKK - KREG in a per-wave LDS slab: with four row
    // tiles per wave all 272 registers do not fit the 256 architectural ones, the compiler parks the overflow in the
    // accumulator half and copies every such fragment back in front of its MFMA (four dependent v_accvgpr_read + a
    // hazard nop per MFMA)
    constexpr int KREG = WR > 2 ? 12 : HC_NKK;
    constexpr int WL_WAVE = (HC_NKK - KREG) * WR * 1024;                   // bytes of LDS weight slab per wave
    static_assert(RT % NWR == 0 && WR % 2 == 0 && CT * KPM == 8, "cluster shape");
    extern __shared__ __attribute__((aligned(16))) char hsm[];
    char* Zf = hsm;                                // [tile][kk][lane][16 B]
    __shared__ int sh_dead;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int vr = wave / NWC, vc = wave % NWC;
    const int g4 = lane >> 4, col = lane & 15;
    const int lane16 = lane * 16;
    const Geometry& g = p.g;
    const int n_res = p.n_res, n_in = p.n_in, n_out = p.n_out;
    const int kin_p = g.kfb - g.kin;
    const int ncols = n_res + n_in;
    // block -> (cluster, member): the members of a cluster are consecutive blocks of ONE XCD (block id mod 8)
    const int cpx = (n_clusters + 7) / 8;                                   // clusters per XCD
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int cluster = xcd * cpx + li / C, c = li % C;
    if (cluster >= n_clusters) return;                                      // (a whole cluster leaves together)
    const int nws = p.n_wsets;
    const int ws = cluster % nws, kc = cluster / nws;                       // weight set of this cluster, run inside the set
    const int gbase = (ws - p.wset_rot % nws + nws) % nws;                  // first group that uses set ws (group g: set (g + rot) % nws)
    unsigned int* err = reinterpret_cast<unsigned int*>(xch + (size_t)n_clusters * 2 * C * 1024);
    unsigned long long* xc = xch + (size_t)cluster * 2 * C * 1024;           // [parity][member][block][lane][2]
    if (tid == 0) sh_dead = 0;

    // ---- resident operands: the wave's WR row tiles x 17 groups of weight fragments live in REGISTERS for the whole
    // launch (136 or 272 of the 512 a lone wave per SIMD may hold): global row tile grt = c RT + vr WR + i of the 16x16x32
    // image, fragment (kk, grt) at (((grt / 4) NKK + kk) 4 + grt % 4) KB.  Only the state image is in LDS.
    u32x4 areg[KREG][WR];
    char* wl = hsm + ZF_BYTES + (size_t)wave * WL_WAVE + lane16;             // this wave's slab: [kk - KREG][i][lane][16 B]
    {
        const char* src = reinterpret_cast<const char*>(p.packed_w) + (size_t)ws * p.wset_stride + p.w16_off + lane16;
#pragma unroll
        for (int kk = 0; kk < HC_NKK; ++kk)
#pragma unroll
            for (int i = 0; i < WR; ++i) {
                const int grt = c * RT + vr * WR + i;
                const u32x4 f = *reinterpret_cast<const u32x4*>(src + ((size_t)((grt >> 2) * HC_NKK + kk) * 4 + (grt & 3)) * 1024);
                if (kk < KREG) areg[kk < KREG ? kk : 0][i] = f;
                else *reinterpret_cast<u32x4*>(wl + (size_t)((kk - KREG) * WR + i) * 1024) = f;
            }
        for (int i = tid; i < ZF_BYTES / 16; i += HC_NT) reinterpret_cast<u32x4*>(Zf)[i] = u32x4{0, 0, 0, 0};   // X_0 = 0
    }
    // pilot (= group = frame) index of the cluster's f-th pilot: the run's f-th group of weight set ws (-1 past the end)
    auto pilot_of = [&](int f) -> int {
        const int gi = gbase + (kc * P + f) * nws;
        return gi < p.n_groups ? gi : -1;
    };
    auto store_E4 = [&](size_t idx, float v0, float v1, float v2, float v3) {      // idx multiple of 4
        if (p.E32) {
            *reinterpret_cast<f32x4*>(p.E32 + idx) = f32x4{v0, v1, v2, v3};
        } else {
            typedef double f64x2s __attribute__((ext_vector_type(2)));
            *reinterpret_cast<f64x2s*>(p.E + idx) = f64x2s{(double)v0, (double)v1};
            *reinterpret_cast<f64x2s*>(p.E + idx + 2) = f64x2s{(double)v2, (double)v3};
        }
    };
    // ---- [U ; F] staging: ALL 256 threads share the 32 positions x P pilots of the group: thread (pilot sf = tid % P,
    // slice sq = tid / P) builds positions EPT sq .. EPT sq + EPT - 1 of its pilot (EPT = P / 8 = 2, 4 or 8 values: with
    // 64 threads doing eight each, as the first version had it, the staging state alone took 96 registers of every lane):
    // inputs row s + 1 scaled (pyESN.py:180-182), teacher row s scaled
    constexpr int EPT = P / 8;
    const int sf = tid & (P - 1), sq = tid / P;
    const int s_pil = pilot_of(sf);
    const bool s_ok = s_pil >= 0;
    double uf_sc[EPT], uf_sh[EPT];
    int uf_kind[EPT];                                                        // 0 = zero, 1 = input, 2 = teacher
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int pos = EPT * sq + e;
        uf_kind[e] = 0; uf_sc[e] = 0.0; uf_sh[e] = 0.0;
        if (!s_ok) continue;
        if (pos < n_in) {
            uf_kind[e] = 1;
            uf_sc[e] = p.in_scale ? p.in_scale[(size_t)s_pil * n_in + pos] : 1.0;
            uf_sh[e] = p.in_shift ? p.in_shift[(size_t)s_pil * n_in + pos] : 0.0;
        } else if (pos >= kin_p && pos < kin_p + n_out) {
            uf_kind[e] = 2;
            uf_sc[e] = p.t_scale ? p.t_scale[(size_t)s_pil * n_out + (pos - kin_p)] : 1.0;
            uf_sh[e] = p.t_shift ? p.t_shift[(size_t)s_pil * n_out + (pos - kin_p)] : 0.0;
        }
    }
    // raw operands of step s: input row s + 1, teacher row s.  Branch-free: every position has a base pointer and a
    // per-step stride (positions that hold nothing point at a valid address with stride 0 and are masked when staged)
    const double* uf_ptr[EPT];
    int uf_stride[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int pos = EPT * sq + e;
        uf_ptr[e] = p.U; uf_stride[e] = 0;
        if (uf_kind[e] == 1) { uf_ptr[e] = p.U + ((size_t)s_pil * p.T_in + 1) * n_in + pos; uf_stride[e] = n_in; }
        if (uf_kind[e] == 2) { uf_ptr[e] = p.D + (size_t)s_pil * (p.S + 1) * n_out + (pos - kin_p); uf_stride[e] = n_out; }
    }
    auto fetch_uf = [&](int s, double (&raw)[EPT]) {
        const int sc = s < p.S ? s : p.S - 1;                                // (past the end: any valid row, never staged)
#pragma unroll
        for (int e = 0; e < EPT; ++e) raw[e] = uf_ptr[e][(size_t)sc * uf_stride[e]];
    };
    // scaled values -> the [U ; F] group of the pilot's tile; member 0 also writes the input columns of E row s + 1
    auto stage_uf = [&](int s, const double (&raw)[EPT]) {
        uint32_t out[EPT / 2];
        double sv[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) sv[e] = uf_kind[e] ? raw[e] * uf_sc[e] + uf_sh[e] : 0.0;    // as the persistent kernel
#pragma unroll
        for (int e = 0; e < EPT / 2; ++e) out[e] = TR::pack2((float)sv[2 * e], (float)sv[2 * e + 1]);
        const int pos0 = EPT * sq;                                           // EPT divides 8: the slice stays inside one chunk
        char* d = Zf + ((size_t)((sf >> 4) * HC_NKK + 16) * 64 + (pos0 >> 3) * 16 + (sf & 15)) * 16 + 2 * (pos0 & 7);
#pragma unroll
        for (int e = 0; e < EPT / 2; ++e) reinterpret_cast<uint32_t*>(d)[e] = out[e];
        if (c == 0 && s_ok && pos0 < n_in) {
            const size_t e0 = ((size_t)s_pil * (p.S + 1) + (s + 1)) * ncols + n_res + pos0;
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                if (pos0 + e < n_in) { if (p.E32) p.E32[e0 + e] = (float)sv[e]; else p.E[e0 + e] = sv[e]; }
        }
    };
    if (c == 0 && s_ok && sq == 0) {                                         // E row 0 = [0, scale(u[0])]  (pyESN.py:179,189)
        const size_t e0 = (size_t)s_pil * (p.S + 1) * ncols;
        for (int k = 0; k < ncols; ++k) {
            double v = 0.0;
            if (k >= n_res) {
                const int ci = k - n_res;
                const double sc = p.in_scale ? p.in_scale[(size_t)s_pil * n_in + ci] : 1.0;
                const double sh = p.in_shift ? p.in_shift[(size_t)s_pil * n_in + ci] : 0.0;
                v = p.U[(size_t)s_pil * p.T_in * n_in + ci] * sc + sh;
            }
            if (p.E32) p.E32[e0 + k] = (float)v; else p.E[e0 + k] = v;
        }
    }
    __syncthreads();
    double raw_cur[EPT], raw_nx[EPT];
    fetch_uf(0, raw_cur);
    stage_uf(0, raw_cur);
    fetch_uf(1, raw_cur);

    const float noise = (float)p.noise;
    const float n_c1 = noise * (1.0f / 256.0f), n_c0 = noise * (0.5f / 256.0f - 0.5f);
    // the WC pilots of this lane (tiles vc WC + n, column col): frame index and step-independent key half
    int fr2[WC];
    uint32_t key1[WC];
#pragma unroll
    for (int n = 0; n < WC; ++n) {
        const int pl = pilot_of((vc * WC + n) * 16 + col);
        fr2[n] = pl;
        key1[n] = mix32((uint32_t)p.seed ^ (((uint32_t)pl + p.frame_off) * 0x9E3779B9U));
    }
    const uint32_t seed_hi = (uint32_t)(p.seed >> 32);
    __syncthreads();

#ifdef ESN_STAMPS
    unsigned long long hc_polls = 0;
#endif
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(xc), 0, 2 * C * 8192, 0x00020000);
    // gather the other members' slices of X_{s+1} (tag of step s) into the state image; false on time-out
    auto gather = [&](int s) -> bool {
        const int par = s & 1;
        const unsigned long long want = hc_tag_bits(s % 15 + 1);
        unsigned pending = (1u << NG) - 1;
        uint32_t spins = 0;
#ifdef ESN_STAMPS
        hc_polls = 0;
#endif
        bool ok = true;
        while (pending) {
            unsigned long long lo[NG], hi[NG];
#ifdef ESN_STAMPS
            ++hc_polls;
#endif
            // one 16-byte agent-scope (sc1) load per chunk: each 8-byte half carries its own tag, so a torn pair is
            // simply not accepted yet
#pragma unroll
            for (int j = 0; j < NG; ++j)
                if (pending & (1u << j)) {
                    const int ch = tid + j * HC_NT;                          // chunk: peer (ch >> 9), block, lane
                    const int cp = (c + 1 + (ch >> 9)) & (C - 1);
                    const int off = (((par * C + cp) * 8 + ((ch >> 6) & 7)) * 64 + (ch & 63)) * 16;
                    const u32x4 w = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, off, 0, 16));
                    lo[j] = (unsigned long long)w[0] | ((unsigned long long)w[1] << 32);
                    hi[j] = (unsigned long long)w[2] | ((unsigned long long)w[3] << 32);
                }
#pragma unroll
            for (int j = 0; j < NG; ++j)
                if ((pending & (1u << j)) && (lo[j] & HC_TAGMASK) == want && (hi[j] & HC_TAGMASK) == want) {
                    const int ch = tid + j * HC_NT;
                    const int cp = (c + 1 + (ch >> 9)) & (C - 1), blk = (ch >> 6) & 7, ln = ch & 63;
                    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<u64x2*>(Zf + ((size_t)((blk / KPM) * HC_NKK + KPM * cp + blk % KPM) * 64 + ln) * 16) =
                        u64x2{lo[j] & ~HC_TAGMASK, hi[j] & ~HC_TAGMASK};
                    pending &= ~(1u << j);
                }
            if (pending) {
                if (++spins > HC_SPIN_LIMIT ||
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

    const char* zf_w = Zf + (size_t)(vc * WC) * HC_NKK * 1024 + lane16;        // the wave's pilot tiles
#ifdef ESN_STAMPS
    unsigned long long hst[6] = {0, 0, 0, 0, 0, 0}, ht_prev = 0;
#define HC_T(v) const unsigned long long v = __builtin_amdgcn_s_memrealtime();
#else
#define HC_T(v)
#endif
    for (int s = 0; s < p.S; ++s) {
        HC_T(ht0)
        fetch_uf(s + 2, raw_nx);                                             // two steps ahead of its use (see stage_uf below)
        // ---- P = Wext[rows of c] * [X_s ; U ; F]: wave (vr, vc) takes WR row tiles x WC pilot tiles ----
        f32x4 acc[WR][WC];
#pragma unroll
        for (int m = 0; m < WR; ++m)
#pragma unroll
            for (int n = 0; n < WC; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        // state fragments four groups ahead in a ring of registers (left to itself the compiler reads one group at a
        // time into the same registers and waits for it: 17 exposed LDS latencies per step)
        {
            u32x4 bq[4][WC];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int n = 0; n < WC; ++n) bq[j][n] = *reinterpret_cast<const u32x4*>(zf_w + (size_t)(n * HC_NKK + j) * 1024);
#pragma unroll
            for (int kk = 0; kk < HC_NKK; ++kk) {
                u32x4 af[WR];
#pragma unroll
                for (int m = 0; m < WR; ++m)
                    af[m] = kk < KREG ? areg[kk < KREG ? kk : 0][m]
                                      : *reinterpret_cast<const u32x4*>(wl + (size_t)((kk - KREG) * WR + m) * 1024);
#pragma unroll
                for (int n = 0; n < WC; ++n)
#pragma unroll
                    for (int m = 0; m < WR; ++m) TR::mma16(acc[m][n], af[m], bq[kk & 3][n]);
                if (kk + 4 < HC_NKK) {
#pragma unroll
                    for (int n = 0; n < WC; ++n)
                        bq[kk & 3][n] = *reinterpret_cast<const u32x4*>(zf_w + (size_t)(n * HC_NKK + kk + 4) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        HC_T(ht1)
        __syncthreads();                                                     // every wave has read the image of step s
        // ---- activation + noise -> operand type; own slice into the image, to the peers, and into E row s + 1 ----
        const uint32_t step_mix = seed_hi ^ ((uint32_t)s * 0x85EBCA6BU + 0x27d4eb2fU);
        const unsigned long long tagb = hc_tag_bits(s % 15 + 1);
        // (order: image + publish first, THEN everything else that enters this CU's memory queue -- the E rows leave after
        //  the gather)
        uint32_t outw[WC][WR / 2][4];
#pragma unroll
        for (int n = 0; n < WC; ++n) {
            const int t = vc * WC + n;
            uint32_t key = 0;
            const double* nz = nullptr;
            if (NOISE == ESN_NOISE_COUNTER) key = mix32(key1[n] ^ step_mix) + (uint32_t)((ROWS * c + 16 * WR * vr) / 4 + g4) * 0x9E3779B9U;
            if (NOISE == ESN_NOISE_TENSOR && fr2[n] >= 0) nz = p.noise_u + ((size_t)fr2[n] * p.S + s) * n_res;
#pragma unroll
            for (int pr = 0; pr < WR / 2; ++pr) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int m = 2 * pr + tt;
                    const int row = ROWS * c + 16 * (WR * vr + m) + 4 * g4;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = TR::act(acc[m][n][j]);
                    if (NOISE == ESN_NOISE_COUNTER) {
                        const uint32_t sq4 = noise_mix(key + (uint32_t)(4 * m) * 0x9E3779B9U);
                        v[0] = fmaf((float)(sq4 & 0xffU), n_c1, v[0] + n_c0);
                        v[1] = fmaf((float)((sq4 >> 8) & 0xffU), n_c1, v[1] + n_c0);
                        v[2] = fmaf((float)((sq4 >> 16) & 0xffU), n_c1, v[2] + n_c0);
                        v[3] = fmaf((float)(sq4 >> 24), n_c1, v[3] + n_c0);
                    } else if (NOISE == ESN_NOISE_TENSOR) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (nz && row + j < n_res) v[j] += noise * ((float)nz[row + j] - 0.5f);
                    }
                    outw[n][pr][2 * tt] = TR::pack2(v[0], v[1]);
                    outw[n][pr][2 * tt + 1] = TR::pack2(v[2], v[3]);
                }
                const int kkl = (WR * vr) / 2 + pr;                          // the member's own group index, < KPM
                *reinterpret_cast<u32x4*>(Zf + ((size_t)(t * HC_NKK + KPM * c + kkl) * 64 + lane) * 16) =
                    u32x4{outw[n][pr][0], outw[n][pr][1], outw[n][pr][2], outw[n][pr][3]};
                if (s + 1 < p.S) {
                    unsigned long long* dst = xc + ((size_t)((s & 1) * C + c) * 8 + (t * KPM + kkl)) * 128 + lane * 2;
                    __hip_atomic_store(dst, ((unsigned long long)outw[n][pr][0] | ((unsigned long long)outw[n][pr][1] << 32)) | tagb,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 1, ((unsigned long long)outw[n][pr][2] | ((unsigned long long)outw[n][pr][3] << 32)) | tagb,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        // E row s + 1 holds the ROUNDED state (what the recurrence continues from)
        auto store_rows = [&]() {
#pragma unroll
            for (int n = 0; n < WC; ++n)
#pragma unroll
                for (int m = 0; m < WR; ++m) {
                    const int row = ROWS * c + 16 * (WR * vr + m) + 4 * g4;
                    if (fr2[n] >= 0 && row < n_res) {
                        float r[4];
                        TR::unpack2(outw[n][m >> 1][2 * (m & 1)], r[0], r[1]);
                        TR::unpack2(outw[n][m >> 1][2 * (m & 1) + 1], r[2], r[3]);
                        store_E4(((size_t)fr2[n] * (p.S + 1) + (s + 1)) * ncols + row, r[0], r[1], r[2], r[3]);
                    }
                }
        };
        HC_T(ht2)
        if (s + 1 == p.S) { store_rows(); break; }                           // (the last state is in E; nobody reads it back)
        stage_uf(s + 1, raw_cur);                                            // [U ; F] of step s + 1 (fetched one step ago)
#pragma unroll
        for (int e = 0; e < EPT; ++e) raw_cur[e] = raw_nx[e];
        if (!gather(s)) return;                                              // (ends with a workgroup barrier)
        HC_T(ht3)
        store_rows();
#ifdef ESN_STAMPS
        {
            const unsigned long long ht4 = __builtin_amdgcn_s_memrealtime();
            hst[0] += ht1 - ht0; hst[1] += ht2 - ht1; hst[2] += ht3 - ht2; hst[3] += hc_polls; hst[4] += ht4 - ht3;
            if (s > 0) hst[5] += ht0 - ht_prev;                    // loop back-edge: end of the previous step -> top of this one
            ht_prev = ht4;
        }
#endif
    }
#ifdef ESN_STAMPS
    if (p.stamps && blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 6; ++i) p.stamps[wave * 8 + i] = hst[i];
#endif
}

bool harvest_cluster_applies(int precision, const RecurParams& p) {
    return (precision == ESN_F16 || precision == ESN_BF16) && p.harvest && p.g.s16 && p.F == 1 &&
           p.n_groups >= 1 && (p.n_res % 4) == 0 && (p.n_res + p.n_in) % 4 == 0;
}

template <typename TR, int C>
static int launch_hc(const RecurParams& p, int n_clusters, unsigned long long* xch, hipStream_t stream) {
    // state image + (four row tiles per wave) the LDS slab of the last weight groups: see KREG in the kernel
    const size_t lds = (size_t)(C / 2) * HC_NKK * 1024 + (C <= 4 ? (size_t)4 * (HC_NKK - 12) * 4 * 1024 : 0);
    const int grid = 8 * ((n_clusters + 7) / 8) * C;
    auto go = [&](auto kern) -> int {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(HC_NT), lds, stream, p, n_clusters, xch);
        return (int)hipGetLastError();
    };
    switch (p.noise_mode) {
        case ESN_NOISE_NONE: return go(harvest_cluster_kernel<TR, ESN_NOISE_NONE, C>);
        case ESN_NOISE_TENSOR: return go(harvest_cluster_kernel<TR, ESN_NOISE_TENSOR, C>);
        default: return go(harvest_cluster_kernel<TR, ESN_NOISE_COUNTER, C>);
    }
}

// C = members per cluster: 2 (default), 4 or 8
int launch_harvest_cluster(int precision, const RecurParams& p, int C, void* workspace, hipStream_t stream) {
    const int n_clusters = hc_clusters(p.n_groups, C, p.n_wsets);
    hipError_t e = hipMemsetAsync(workspace, 0, harvest_cluster_workspace_bytes(p.n_groups, C, p.n_wsets), stream);   // tags start at 1
    if (e != hipSuccess) return (int)e;
    unsigned long long* xch = reinterpret_cast<unsigned long long*>(workspace);
#define HC_CASE(TRv, Cv) if (C == Cv) return launch_hc<TRv, Cv>(p, n_clusters, xch, stream);
    if (precision == ESN_F16) { HC_CASE(TraitsF16, 2) HC_CASE(TraitsF16, 4) HC_CASE(TraitsF16, 8) }
    if (precision == ESN_BF16) { HC_CASE(TraitsBF16, 2) HC_CASE(TraitsBF16, 4) HC_CASE(TraitsBF16, 8) }
#undef HC_CASE
    return -1;
}

}  // namespace esn
