// Register-resident-state recurrence (fp16 / bf16 predict, N_res <= 512, n_in 13..16, n_out <= 8).
//
// The skewed LDS-state kernel (esn_recur_mfma_impl.h) keeps Zt = [X;U;F] of 128 frames in LDS, lets
// eight waves each own 64 rows x 128 frames and pays three barriers, an LDS round trip of the whole
// state and a serial activation phase per step (MFMA pipe busy 56 % of the time).  Here the roles flip:
//
//   * 4 waves per workgroup, ONE per SIMD, 512 registers each.  A wave owns 32 frames x ALL rows: the
//     fp16 state of its frames (N_res x 32) lives in its registers as the B operand of every MFMA, and
//     the new state comes out of the accumulators in the same lane (frame) -- no exchange, no LDS
//     write, no barrier for the state.  Two v_permlane32_swap per k-group turn an accumulator tile into
//     the next step's B fragments.
//   * the weights (A operand) stream once per step per CU through an LDS ring: one 34 KB chunk per
//     32-row tile, LDS-DMA three chunks ahead (~100 KB in flight per CU), one raw s_barrier per chunk;
//     every wave reads every fragment (ds_read_b128, conflict-free fragment order).
//   * row-tile-major order: tile rt is complete after its NKG MFMAs, so its activation (tanh, noise,
//     pack) runs as VALU filler between the MFMAs of tile rt+1: the matrix pipe never waits for it.
//   * read-out: per step up to three more chunks -- the W_out images (rows 0-7 hi, 8-15 rounding
//     residual, x gain) of the groups this 128-frame tile touches -- ride in the same ring; a wave runs
//     the chunk(s) of its own two 16-frame halves against the NEW state, which gives output row s and
//     the fed-back F_{s+1} directly (no separate final pass).
//
// Arithmetic, weight image and noise stream are those of the other fp16 / bf16 kernels.
#include <utility>
#include "esn_recur_mfma_impl.h"

namespace esn {

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>) -- the chunk loop must be
// unrolled for real (register arrays indexed by the chunk number), which `#pragma unroll` does not guarantee
template <class F, size_t... I>
__device__ __forceinline__ void rs_for_seq(F&& f, std::index_sequence<I...>) {
    (f(std::integral_constant<int, (int)I>{}), ...);
}

template <typename E>
__device__ __forceinline__ u32x2 rs_pack4(float a, float b, float c, float d) {
    typedef E vec4 __attribute__((ext_vector_type(4)));
    const vec4 v = {(E)a, (E)b, (E)c, (E)d};
    return __builtin_bit_cast(u32x2, v);
}

constexpr int RS_RING = 4;                    // ring slots
constexpr int RS_GC = 3;                      // read-out chunks per step (groups a 128-slot tile can touch at Fpad >= 64)

// per-group read-out image of this kernel: [NKG k-groups][64 lanes][16 B] (row = lane & 31: 0-7 hi, 8-15 lo,
// 16-31 zero; k natural), then {1/gain, gain, 0, 0}
static inline size_t rs_wout_image_bytes(int Kp) { return (size_t)(Kp / 16) * 1024 + 16; }

bool rs_path_applies(int precision, const RecurParams& p) {
    const int kin_p = p.g.kfb - p.g.kin;
    return (precision == ESN_F16 || precision == ESN_BF16) && !p.harvest && p.n_wsets == 1 && p.g.rs &&
           p.g.Mp == 512 && p.g.Kp == 544 && kin_p == 16 && p.n_out <= 8 && round_up(p.F, 16) >= 64 &&
           p.noise_mode != ESN_NOISE_TENSOR;
}

template <typename TR, int NRT, int NKG, int NOISE>
__global__ __launch_bounds__(256) void recur_rs_kernel(RecurParams p, size_t wo_rs_off) {
    extern __shared__ __attribute__((aligned(16))) char ring[];
    constexpr int CH_STRIDE = 36 * 1024;                 // 36 pieces per slot: 4 waves x 9 DMAs (NKG = 34 used)
    constexpr int NCH = NRT + RS_GC;                      // chunks per step
    static_assert(NKG == 2 * NRT + 2 && NKG <= 36, "k-groups: two per row tile + [U] + [F]");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int n_in = p.n_in, n_out = p.n_out, n_res = p.n_res;
    const int tile = xcd_remap(blockIdx.x, p.n_tiles);
    const int slot0 = tile * 128;
    const int g0 = slot0 / p.Fpad;
    if (g0 >= p.n_groups) return;

    // ---- this lane's frame ---------------------------------------------------------------------------
    int grp;
    const int fr = slot_frame(p, slot0 + wave * 32 + r, grp);
    const bool live = fr >= 0;
    const size_t fr_c = live ? (size_t)fr : 0, grp_c = live ? (size_t)grp : 0;
    // the wave's two 16-frame halves and their groups relative to the tile's first group (wave-uniform)
    const int gl = __builtin_amdgcn_readfirstlane((slot0 + wave * 32) / p.Fpad - g0);
    const int gh = __builtin_amdgcn_readfirstlane((slot0 + wave * 32 + 16) / p.Fpad - g0);
    float in_sc[8], in_sh[8];                             // inputs 8h .. 8h+7 of this frame's group
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int i = 8 * h + e, ic = i < n_in ? i : 0;
        const double sc = p.in_scale ? p.in_scale[grp_c * n_in + ic] : 1.0;
        const double sh = p.in_shift ? p.in_shift[grp_c * n_in + ic] : 0.0;
        in_sc[e] = (live && i < n_in) ? (float)sc : 0.f;
        in_sh[e] = (live && i < n_in) ? (float)sh : 0.f;
    }
    float un_inv[4], un_sh[4];                            // outputs 4h .. 4h+3: {1/t_scale, t_shift}
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int o = 4 * h + j, oc = o < n_out ? o : 0;
        un_inv[j] = p.t_scale ? (float)(1.0 / p.t_scale[grp_c * n_out + oc]) : 1.f;
        un_sh[j] = p.t_shift ? (float)p.t_shift[grp_c * n_out + oc] : 0.f;
    }
    const char* wo_base = reinterpret_cast<const char*>(p.packed_wout) + wo_rs_off;
    const float inv_gain = *reinterpret_cast<const float*>(wo_base + grp_c * p.wout_stride + (size_t)NKG * 1024);

    // ---- state: B fragments of all k-groups, this lane = (frame r, k-half h) ------------------------
    u32x4 st[NKG];                                        // [0, 2 NRT): state; NKG-2: U_s; NKG-1: F_s
    u32x4 sn[2 * NRT];                                    // new state, built tile by tile
#pragma unroll
    for (int kg = 0; kg < 2 * NRT; ++kg) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * kg + 8 * h + e;
            v[e] = (live && p.x0 && k < n_res) ? (float)p.x0[grp_c * n_res + k] : 0.f;
        }
        const u32x2 a = rs_pack4<typename TR::elem>(v[0], v[1], v[2], v[3]), b = rs_pack4<typename TR::elem>(v[4], v[5], v[6], v[7]);
        st[kg] = u32x4{a[0], a[1], b[0], b[1]};
    }
    const double* u_row = p.U + fr_c * p.T_in * n_in + (8 * h < n_in ? 8 * h : 0);
    auto load_in = [&](int s, double (&raw)[8]) {         // raw inputs 8h..8h+7 of recurrence step s (clamped, masked later)
        const int row = s < p.T_in ? s : 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) raw[e] = u_row[(size_t)row * n_in + (8 * h + e < n_in ? e : 0)];
    };
    auto make_u = [&](int s, const double (&raw)[8]) -> u32x4 {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaf(s < p.T_in ? (float)raw[e] : 0.f, in_sc[e], in_sh[e]);
        const u32x2 a = rs_pack4<typename TR::elem>(v[0], v[1], v[2], v[3]), b = rs_pack4<typename TR::elem>(v[4], v[5], v[6], v[7]);
        return u32x4{a[0], a[1], b[0], b[1]};
    };
    double in_next[8];
    load_in(0, in_next);
    st[NKG - 2] = make_u(0, in_next);
    load_in(1, in_next);
    {   // F_0 = y0 of the group (continuation) or zeros: k-local 0..7 of the [F] group in lane half 0
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = (live && h == 0 && e < n_out && p.y0 && p.teacher_forcing) ? (float)p.y0[grp_c * n_out + e] : 0.f;
        const u32x2 a = rs_pack4<typename TR::elem>(v[0], v[1], v[2], v[3]), b = rs_pack4<typename TR::elem>(v[4], v[5], v[6], v[7]);
        st[NKG - 1] = u32x4{a[0], a[1], b[0], b[1]};
    }

    // ---- weight stream: chunk ci of a step = row tile ci (ci < NRT) or read-out image of group g0 + ci - NRT ----
    const int lane16 = lane * 16;
    constexpr int OOB = 0x7ffffff0;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.packed_w)), 0, (int)p.wset_stride, 0x00020000);
    const size_t wo_total = (size_t)p.n_groups * p.wout_stride;
    const __amdgpu_buffer_rsrc_t wo_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.packed_wout)), 0, (int)(wo_total < 0x7fffffffu ? wo_total : 0x7fffffffu), 0x00020000);
    // wave w moves pieces w, w+4, .. (9 DMAs per chunk per wave; pieces >= NKG: out-of-range offset, no traffic)
    auto issue = [&](auto ci_tag, bool on, int slot) {
        constexpr int ci = decltype(ci_tag)::value;
        char* dst = ring + (size_t)slot * CH_STRIDE;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int piece = wave + 4 * t;
            const bool ok = on && piece < NKG;
            if constexpr (ci < NRT) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16,
                                                         ok ? lane16 : OOB, ok ? (ci * NKG + piece) * 1024 : 0, 0, 0);
            } else {
                const int g = g0 + (ci - NRT);
                const bool okg = ok && g < p.n_groups;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wo_rsrc, (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16,
                                                         okg ? lane16 : OOB,
                                                         okg ? (int)((size_t)g * p.wout_stride + wo_rs_off) + piece * 1024 : 0, 0, 0);
            }
        }
    };
    const float noise = (float)p.noise;
    const float n_c1 = noise * (1.0f / 256.0f), n_c0 = noise * (0.5f / 256.0f - 0.5f);
    const int out_rows = p.S - p.transient;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.Y, 0, (int)(((size_t)p.n_frames * out_rows * n_out * 8) < 0x7fffffffu ? ((size_t)p.n_frames * out_rows * n_out * 8) : 0x7fffffffu), 0x00020000);

    // Activation of a finished accumulator tile -> two B fragments of the NEW state (k-groups 2 rt, 2 rt + 1), cut
    // into 32 slices of 4-6 vector instructions: slice `pos` is issued right behind MFMA `pos` of the NEXT tile and
    // pinned there (sched_barrier), so every MFMA gap carries its share of the tanh / noise / pack work.  Left to
    // itself the scheduler emits the activation in blocks of ~50 instructions and the matrix pipe idles meanwhile.
    //   quad q = pos / 8 holds rows 8 q + 4 h + j (j = 0..3) of the tile; phases pos % 8:
    //   0-1 exp2, 1-2 +1, 2-3 rcp, 3-4 tanh = 1 - 2 / (1 + 2^z) (+ noise bias), 4 noise hash, 5-6 noise, 7 pack;
    //   phase 7 of quads 1 and 3 also runs the two v_permlane32_swap that complete a fragment.
    constexpr bool PK_NOISE = NOISE == ESN_NOISE_COUNTER && std::is_same<TR, TraitsF16>::value;
    typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const _Float16 c1s = (_Float16)(1024.0f * n_c1);
    const h16x2 c1h = {c1s, c1s};
    const float t_bias = PK_NOISE ? 1.0f + n_c0 - (float)c1s : 1.0f;
    float az[4];                 // per-quad temporaries of the sliced activation
    uint32_t asq = 0;
    h16x2 aw01, aw23, at01, at23;
    u32x2 axq[4];
    auto act_slice = [&](auto pos_tag, const f32x16& ap, int rt, uint32_t key, u32x4& f0, u32x4& f1) {
        constexpr int pos = decltype(pos_tag)::value;
        if constexpr (pos < 32) {
            constexpr int q = pos / 8, ph = pos % 8;
            if constexpr (ph == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) az[j] = ap[4 * q + j];
                az[0] = __builtin_amdgcn_exp2f(az[0]); az[1] = __builtin_amdgcn_exp2f(az[1]);
            } else if constexpr (ph == 1) {
                az[2] = __builtin_amdgcn_exp2f(az[2]); az[3] = __builtin_amdgcn_exp2f(az[3]);
                az[0] += 1.0f; az[1] += 1.0f;
            } else if constexpr (ph == 2) {
                az[2] += 1.0f; az[3] += 1.0f;
                az[0] = __builtin_amdgcn_rcpf(az[0]); az[1] = __builtin_amdgcn_rcpf(az[1]);
            } else if constexpr (ph == 3) {
                az[2] = __builtin_amdgcn_rcpf(az[2]); az[3] = __builtin_amdgcn_rcpf(az[3]);
                az[0] = fmaf(-2.0f, az[0], t_bias); az[1] = fmaf(-2.0f, az[1], t_bias);
            } else if constexpr (ph == 4) {
                az[2] = fmaf(-2.0f, az[2], t_bias); az[3] = fmaf(-2.0f, az[3], t_bias);
                if (NOISE == ESN_NOISE_COUNTER) asq = noise_quad(key, (uint32_t)(rt * 8 + 2 * q + h));
            } else if constexpr (ph == 5) {
                if constexpr (PK_NOISE) {
                    aw01 = __builtin_bit_cast(h16x2, __builtin_amdgcn_perm(0x3C3C3C3Cu, asq, 0x04010400u));
                    aw23 = __builtin_bit_cast(h16x2, __builtin_amdgcn_perm(0x3C3C3C3Cu, asq, 0x04030402u));
                    at01 = __builtin_convertvector(f32x2{az[0], az[1]}, h16x2);
                    at23 = __builtin_convertvector(f32x2{az[2], az[3]}, h16x2);
                } else if (NOISE == ESN_NOISE_COUNTER) {
                    az[0] = fmaf((float)(asq & 0xffU), n_c1, az[0] + n_c0);
                    az[1] = fmaf((float)((asq >> 8) & 0xffU), n_c1, az[1] + n_c0);
                }
            } else if constexpr (ph == 6) {
                if constexpr (PK_NOISE) {
                    const h16x2 x01 = __builtin_elementwise_fma(aw01, c1h, at01);
                    const h16x2 x23 = __builtin_elementwise_fma(aw23, c1h, at23);
                    axq[q] = u32x2{__builtin_bit_cast(uint32_t, x01), __builtin_bit_cast(uint32_t, x23)};
                } else if (NOISE == ESN_NOISE_COUNTER) {
                    az[2] = fmaf((float)((asq >> 16) & 0xffU), n_c1, az[2] + n_c0);
                    az[3] = fmaf((float)(asq >> 24), n_c1, az[3] + n_c0);
                }
            } else {
                if constexpr (!PK_NOISE) axq[q] = rs_pack4<typename TR::elem>(az[0], az[1], az[2], az[3]);
                // lane (r, h) holds rows 8 q + 4 h + j; fragment kg = 2 rt + (q >> 1) wants k-local 8 h' .. 8 h' + 7 in
                // lane (r, h'): half 0 keeps its even quad and receives the partner's, half 1 likewise with the odd one
                if constexpr (q == 1 || q == 3) {
                    u32x2 lo = axq[q - 1], hi = axq[q];
#pragma unroll
                    for (int w2 = 0; w2 < 2; ++w2) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(lo[w2], hi[w2], false, false);
                        lo[w2] = sw[0]; hi[w2] = sw[1];
                    }
                    (q == 1 ? f0 : f1) = u32x4{lo[0], lo[1], hi[0], hi[1]};
                }
            }
        }
    };

    // prologue of the stream: chunks 0 .. RS_RING-2
    issue(std::integral_constant<int, 0>{}, true, 0);
    issue(std::integral_constant<int, 1>{}, true, 1);
    issue(std::integral_constant<int, 2>{}, true, 2);
    int cnt = 0;                                          // chunk number of the step's first chunk (ring slot = chunk & 3)
    f32x16 acc_a, acc_b;                                  // even / odd row tiles
    asm volatile("s_waitcnt vmcnt(18)" ::: "memory");     // chunk 0 has landed (chunks 1 and 2 may still fly)
    __builtin_amdgcn_s_barrier();
#ifdef ESN_STAMPS
    unsigned long long rs_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
    for (int s = 0; s < p.S; ++s) {
        ESN_STAMP(t_s0)
        uint32_t key = 0;
        if (NOISE == ESN_NOISE_COUNTER) key = noise_key(p.seed, (uint32_t)fr + p.frame_off, (uint32_t)s);
        f32x16 racc_lo, racc_hi;
#pragma unroll
        for (int i = 0; i < 16; ++i) { racc_lo[i] = 0.f; racc_hi[i] = 0.f; }
        // Event E(c), once per chunk c: chunk c+1 has landed (this wave's 9 pieces: all but the 9 youngest VMEM
        // ops -- those of chunk c+2 --, then everybody's: barrier) and, everybody being inside chunk c, the slot of
        // chunk c-1 is refilled with chunk c+3.  Tile chunks run it in the MIDDLE of their MFMAs, so the A-fragment
        // pipeline flows across chunk boundaries without draining; read-out chunks (conditional per wave) at their start.
        auto event = [&](auto ci_tag) {
            constexpr int ci = decltype(ci_tag)::value;
            ESN_STAMP(e0)
            asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            ESN_STAMP(e1)
            __builtin_amdgcn_s_barrier();
            ESN_STAMP(e2)
#ifdef ESN_STAMPS
            rs_acc[0] += e1 - e0; rs_acc[1] += e2 - e1;
#endif
            constexpr int ahead = RS_RING - 1;
            constexpr int cj = (ci + ahead) % NCH;
            const bool on = s + (ci + ahead) / NCH < p.S;
            issue(std::integral_constant<int, cj>{}, on, (cnt + ci + ahead) & (RS_RING - 1));
        };
        // ---- the NRT tile chunks as ONE stream of NRT * NKG MFMA positions, A fragments RS_D ahead (static rotation
        // of 8 buffers), activation slice of the previous tile pinned behind every MFMA ----
        constexpr int RS_D = 6, MID = NKG / 2;
        {
            u32x4 a[8];
            const char* slot_ptr[RS_RING];
#pragma unroll
            for (int i = 0; i < RS_RING; ++i) slot_ptr[i] = ring + (size_t)((cnt + i) & (RS_RING - 1)) * CH_STRIDE + lane16;
#pragma unroll
            for (int i = 0; i < RS_D; ++i) a[i] = *reinterpret_cast<const u32x4*>(slot_ptr[0] + i * 1024);
            rs_for_seq([&](auto ci_tag) {
                constexpr int ci = decltype(ci_tag)::value;
                f32x16& acc = (ci & 1) ? acc_b : acc_a;
                const f32x16& ap = (ci & 1) ? acc_a : acc_b;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                rs_for_seq([&](auto pos_tag) {
                    constexpr int pos = decltype(pos_tag)::value;
                    constexpr int pq = ci * NKG + pos;
                    if constexpr (pos == MID) event(ci_tag);
#ifndef RS_DIAG_NOLDS            // (diagnostic builds only: wrong results, timing of one ingredient)
                    if constexpr (pq + RS_D < NRT * NKG) {
                        constexpr int cj = (pq + RS_D) / NKG, kj = (pq + RS_D) % NKG;
                        a[(pq + RS_D) & 7] = *reinterpret_cast<const u32x4*>(slot_ptr[cj & (RS_RING - 1)] + kj * 1024);
                    }
#endif
                    TR::mma32(acc, a[pq & 7], st[pos]);
#ifndef RS_DIAG_NOACT
                    if constexpr (ci > 0) act_slice(pos_tag, ap, ci - 1, key, sn[2 * (ci > 0 ? ci - 1 : 0)], sn[2 * (ci > 0 ? ci - 1 : 0) + 1]);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }, std::make_index_sequence<NKG>{});
            }, std::make_index_sequence<NRT>{});
        }
        ESN_STAMP(t_s1)
        // ---- read-out chunks: NKG MFMAs against the NEW state, position `pos` takes k-group korder(pos) ----
        auto run_chunk = [&](const char* ab, f32x16& acc, auto korder, auto act_tag, const f32x16& ap) {
            constexpr bool ACT = decltype(act_tag)::value;
            u32x4 a[4];
#pragma unroll
            for (int i = 0; i < 3; ++i) a[i] = *reinterpret_cast<const u32x4*>(ab + korder(i) * 1024);
            rs_for_seq([&](auto pos_tag) {
                constexpr int pos = decltype(pos_tag)::value;
                if constexpr (pos + 3 < NKG) a[(pos + 3) & 3] = *reinterpret_cast<const u32x4*>(ab + korder(pos + 3) * 1024);
                constexpr int kg = korder(pos);
                TR::mma32(acc, a[pos & 3], kg < 2 * NRT ? sn[kg] : st[kg]);
                if constexpr (ACT) act_slice(pos_tag, ap, NRT - 1, key, sn[2 * NRT - 2], sn[2 * NRT - 1]);
                __builtin_amdgcn_sched_barrier(0);
            }, std::make_index_sequence<NKG>{});
        };
        auto k_natural = [](int pos) constexpr { return pos; };
        // behind the last tile: k-groups 2 NRT - 2 and 2 NRT - 1 (that tile's own fragments) go last
        auto k_readout = [](int pos) constexpr { return pos < 2 * NRT - 2 ? pos : (pos < 2 * NRT ? pos + 2 : pos - 2); };
        const std::integral_constant<bool, true> with_act;
        const std::integral_constant<bool, false> no_act;
        rs_for_seq([&](auto j_tag) {
            constexpr int j = decltype(j_tag)::value;
            constexpr int ci = NRT + j;
            event(std::integral_constant<int, ci>{});
            const char* ab = ring + (size_t)((cnt + ci) & (RS_RING - 1)) * CH_STRIDE + lane16;
            if constexpr (j == 0) {
                // the activation of the last tile rides under the first read-out chunk (or runs alone in a wave that
                // has no use for this group's image)
                const f32x16& ap = ((NRT - 1) & 1) ? acc_b : acc_a;
                if (gl == 0) {
                    run_chunk(ab, racc_lo, k_readout, with_act, ap);
                } else {
                    rs_for_seq([&](auto pos_tag) { act_slice(pos_tag, ap, NRT - 1, key, sn[2 * NRT - 2], sn[2 * NRT - 1]); },
                               std::make_index_sequence<32>{});
                }
            } else {
                if (j == gl) run_chunk(ab, racc_lo, k_natural, no_act, racc_lo);
                if (j == gh && gh != gl) run_chunk(ab, racc_hi, k_natural, no_act, racc_hi);
            }
        }, std::make_index_sequence<RS_GC>{});
        cnt += NCH;
        ESN_STAMP(t_s2)
        // ---- step boundary: Y_{s+1} = W_out [X_{s+1} ; U_s] -> output row s, F_{s+1}; next inputs; state swap ----
        float y4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = racc_lo[j] + racc_lo[4 + j], hi = racc_hi[j] + racc_hi[4 + j];
            y4[j] = ((gh != gl && r >= 16) ? hi : lo) * inv_gain;
        }
        {
            const int orow = s - p.transient;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = 4 * h + j;
                const bool wr = live && o < n_out && orow >= 0;
                const double yo = (double)((y4[j] - un_sh[j]) * un_inv[j]);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, yo), y_rsrc,
                                                      wr ? (int)((((size_t)fr_c * out_rows + (orow >= 0 ? orow : 0)) * n_out + o) * 8) : OOB, 0, 0);
            }
        }
        {   // F_{s+1}: lane (r, 0) = outputs 0..7 of its frame (own 0..3, partner's 4..7), lane (r, 1) = zeros
            u32x2 own = rs_pack4<typename TR::elem>(y4[0], y4[1], y4[2], y4[3]);
            u32x2 oth = own;
#pragma unroll
            for (int w2 = 0; w2 < 2; ++w2) {
                const auto sw = __builtin_amdgcn_permlane32_swap(own[w2], oth[w2], false, false);
                own[w2] = sw[0]; oth[w2] = sw[1];
            }
            const bool keep = h == 0 && live && p.teacher_forcing;
            st[NKG - 1] = keep ? u32x4{own[0], own[1], oth[0], oth[1]} : u32x4{0, 0, 0, 0};
        }
        st[NKG - 2] = make_u(s + 1, in_next);
        load_in(s + 2, in_next);
#pragma unroll
        for (int kg = 0; kg < 2 * NRT; ++kg) st[kg] = sn[kg];
        ESN_STAMP(t_s3)
#ifdef ESN_STAMPS
        rs_acc[2] += t_s1 - t_s0; rs_acc[3] += t_s2 - t_s1; rs_acc[4] += t_s3 - t_s2;
#endif
    }
#ifdef ESN_STAMPS
    if (p.stamps && blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 5; ++i) p.stamps[wave * 8 + i] = rs_acc[i];
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (drain the out-of-range tail DMAs before the LDS is released)
}

template <typename TR>
static int launch_rs_t(const RecurParams& p, size_t wo_rs_off, hipStream_t stream) {
    const size_t lds = (size_t)RS_RING * 36 * 1024;
    const void* k = nullptr;
    if (p.noise_mode == ESN_NOISE_NONE) k = reinterpret_cast<const void*>(recur_rs_kernel<TR, 16, 34, ESN_NOISE_NONE>);
    else k = reinterpret_cast<const void*>(recur_rs_kernel<TR, 16, 34, ESN_NOISE_COUNTER>);
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    if (p.noise_mode == ESN_NOISE_NONE)
        hipLaunchKernelGGL((recur_rs_kernel<TR, 16, 34, ESN_NOISE_NONE>), dim3(p.n_tiles), dim3(256), lds, stream, p, wo_rs_off);
    else
        hipLaunchKernelGGL((recur_rs_kernel<TR, 16, 34, ESN_NOISE_COUNTER>), dim3(p.n_tiles), dim3(256), lds, stream, p, wo_rs_off);
    return (int)hipGetLastError();
}

int launch_recur_rs(int precision, const RecurParams& p, size_t wo_rs_off, hipStream_t stream) {
    if (precision == ESN_F16) return launch_rs_t<TraitsF16>(p, wo_rs_off, stream);
    if (precision == ESN_BF16) return launch_rs_t<TraitsBF16>(p, wo_rs_off, stream);
    return -1;
}

}  // namespace esn
