// fp16 / bf16 predict at 257..512 reservoir units on v_mfma_f32_16x16x32: the skewed schedule of
// esn_recur_mfma_impl.h (two wave sets one third of a step apart, see there) with every image re-cut for
// the 16 x 16 x 32 shape.  Why: the chip lowers its clock under this kernel's load, and holds a 13-17 %
// higher one on the small shape (tools/probes/mfma_shape_probe.hip: the kernel's own slots rebuilt with
// both shapes -- 12 % more cycles, 4-10 % less wall time per slot).  What changes with the shape:
//
//  * State image, fragment-major:  Zf[column tile c of 16 frames][32-k group kk][lane][16 B].  Lane
//    (q = lane >> 4, col = lane & 15) of block (c, kk) holds the eight halves of k-chunk q of frame 16 c + col --
//    exactly the B operand of one MFMA, read by ONE linear, conflict-free ds_read_b128 (address = block + 16 lane).
//  * k order.  The accumulator tile of a 16x16x32 MFMA leaves rows 4 g .. 4 g + 3 (g = lane >> 4) of column
//    lane & 15 in a lane; two row tiles give that lane eight values = one 16-byte chunk of the NEXT step's B
//    operand.  So state row R = 32 kk + 16 t + 4 g + e (t = 0, 1) lives at position 32 kk + 8 g + 4 t + e of
//    its 32-k group (s16_pos below); the weight image (esn_pack.hip) is cut in the same order, and phase E
//    stores 16 bytes per lane and tile pair at block + 16 lane -- linear again, half the stores.
//  * Read-out.  The B fragment of column tile c and group kk IS the operand of the read-out MFMA of that tile
//    (the 32x32x16 kernel reads a second, differently cut view of the rows, with bank conflicts).
//  * [U ; F] group kk = 16 in natural order: inputs at k = 512 + ci, feedback at kfb + o.
// Serves Mp = 512, Kp = 544 (n_in <= 16, n_out <= 8), 8 waves of 64 rows x 128 frames, 128 accumulator
// registers.  Everything else -- slot program, fixed-buffer weight prefetch, OOB loads instead of branches,
// LDS-DMA input staging, packed-half noise tail, the counter noise as a function of (seed, frame, step, row) --
// is the 32x32x16 kernel's, and the two are compared bit-for-bit-tolerance in tests/test_gpu_parity.py.
#pragma once
#include "esn_recur_mfma_impl.h"

namespace esn {

constexpr int S16_NKK = 17;            // 32-k groups: 16 of state rows + the [U ; F] group
constexpr int S16_NKH = 8;             // state groups per half
constexpr int S16_MP = 512;

template <typename TR, int NOISE>
__global__ __launch_bounds__(512) void recur_skew16_kernel(RecurParams p) {
    extern __shared__ __attribute__((aligned(16))) char zf[];
    constexpr int NW = 8, BT = 128, NOWN = 8, NKK = S16_NKK, NKH = S16_NKH;
    constexpr int NTHREADS = NW * 64;
    constexpr int TILE_B = NKK * 1024;                    // bytes of one column tile's image
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Geometry& g = p.g;
    const int n_res = p.n_res, n_in = p.n_in, n_out = p.n_out;
    const int kin_p = g.kfb - g.kin;
    const int kfb_p = round_up(n_out, 4);
    const int out_rows = p.S - p.transient;

    const int tile = xcd_remap(blockIdx.x, p.n_tiles);
    const int slot0 = tile * BT;
    const int grp0 = slot_group(p, slot0);
    if (grp0 >= p.n_groups) return;
    const int wset = slot_wset(p, slot0);

    // tables behind the state image (as in esn_recur_mfma_impl.h) + the step-independent half of the noise key
    int* tab_fr = reinterpret_cast<int*>(zf + (size_t)NOWN * TILE_B);
    uint32_t* tab_key = reinterpret_cast<uint32_t*>(tab_fr + BT);
    float2* tab_in = reinterpret_cast<float2*>(tab_key + BT);
    float2* tab_un = tab_in + NOWN * kin_p;
    char* in_slots = reinterpret_cast<char*>(tab_un + NOWN * 16);       // [NOWN][2][1 KB] raw float64 input rows
    int* tab_off = reinterpret_cast<int*>(in_slots + (size_t)NOWN * 2048);
    // counter noise: the (frame, step) keys of the 128 frames, written once per step by the tile's own wave (sixteen
    // lanes, one mix32) instead of being re-derived by every lane of every wave for each of its eight tiles; two
    // buffers by step parity (set B reads step s while set A already writes step s + 1)
    uint32_t* tab_ks = reinterpret_cast<uint32_t*>(tab_off + BT);
    for (int i = tid; i < BT; i += NTHREADS) {
        int gtmp;
        const int fr = slot_frame(p, slot0 + i, gtmp);
        tab_fr[i] = fr;
        tab_key[i] = mix32((uint32_t)p.seed ^ (((uint32_t)fr + p.frame_off) * 0x9E3779B9U));   // noise_key, stage 1
    }
    for (int i = tid; i < NOWN * kin_p; i += NTHREADS) {
        const int c16 = i / kin_p, c = i % kin_p;
        const int cg = slot_group(p, slot0 + c16 * 16);
        float2 v = make_float2(0.f, 0.f);
        if (cg < p.n_groups && c < n_in) {
            v.x = p.in_scale ? (float)p.in_scale[(size_t)cg * n_in + c] : 1.f;
            v.y = p.in_shift ? (float)p.in_shift[(size_t)cg * n_in + c] : 0.f;
        }
        tab_in[i] = v;
    }
    for (int i = tid; i < NOWN * 16; i += NTHREADS) {
        const int c16 = i / 16, o = i % 16;
        const int cg = slot_group(p, slot0 + c16 * 16);
        float2 v = make_float2(1.f, 0.f);
        if (cg < p.n_groups && o < n_out) {
            if (p.t_scale) v.x = (float)(1.0 / p.t_scale[(size_t)cg * n_out + o]);
            if (p.t_shift) v.y = (float)p.t_shift[(size_t)cg * n_out + o];
        }
        tab_un[i] = v;
    }
    // byte offset of natural column k of frame f in the state image
    auto zoff = [&](int f, int k) -> size_t {
        const int pos = s16_pos(k);
        return ((size_t)((f >> 4) * NKK + (pos >> 5)) * 64 + ((pos >> 3) & 3) * 16 + (f & 15)) * 16 + 2 * (pos & 7);
    };
    // ---- LDS init: x0 in the state rows, y0 in the feedback columns, zeros elsewhere --------------
    for (int i = tid; i < BT * g.Kp; i += NTHREADS) {
        const int f = i / g.Kp, k = i % g.Kp;
        float v = 0.f;
        int pg;
        const int fr = slot_frame(p, slot0 + f, pg);
        if (fr >= 0) {
            if (k < n_res) {
                if (p.x0) v = (float)p.x0[(size_t)pg * n_res + k];
            } else if (k >= g.kfb && k < g.kfb + n_out) {
                if (p.y0) v = (float)p.y0[(size_t)pg * n_out + (k - g.kfb)];
            }
        }
        TR::store1(zf + zoff(f, k), v);
    }
    __syncthreads();

    // ---- the wave's own column tile (tile `wave`): read-out lane view = output rows 4 oq .. 4 oq + 3 of frame ofc
    const int oq_w = lane >> 4, ofc_w = lane & 15;
    const int own_grp = __builtin_amdgcn_readfirstlane(slot_group(p, slot0 + wave * 16));
    const bool has_ro = own_grp < p.n_groups;                 // wave-uniform
    float wo_inv = 1.f;
    if (has_ro)
        wo_inv = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.packed_wout)
                                                 + (size_t)own_grp * p.wout_stride + p.wo16_off + (size_t)NKK * 1024);
    const uint32_t in_stride = (uint32_t)(p.T_in * n_in);
    // inputs of step 0, straight from HBM (every later step arrives by LDS-DMA)
    for (int e = tid; e < BT * kin_p; e += NTHREADS) {
        const int f = e / kin_p, ci = e % kin_p;
        const int fr = tab_fr[f];
        float v = 0.f;
        if (fr >= 0 && ci < n_in) {
            const int row = p.in_row_off;
            const double raw = (row < p.T_in) ? p.U[(size_t)fr * in_stride + (size_t)row * n_in + ci] : 0.0;
            const float2 ss = tab_in[(f >> 4) * kin_p + ci];
            v = fmaf((float)raw, ss.x, ss.y);
        }
        TR::store1(zf + zoff(f, g.kin + ci), v);
    }

    const float noise = (float)p.noise;
    const float n_c1 = noise * (1.0f / 256.0f), n_c0 = noise * (0.5f / 256.0f - 0.5f);
    f32x4 yacc = {0.f, 0.f, 0.f, 0.f};
    const int lane16 = lane * 16;
    constexpr int OOB = 0x7ffffff0;

    // Y complete: feedback columns into the image, unscaled output row `orow` to HBM, reset yacc
    auto finish_readout = [&](int orow, bool write_fb) {
        int ofc = ofc_w, oq = oq_w;
        asm volatile("" : "+v"(ofc), "+v"(oq));              // derived per step, not kept across the GEMM phases
        if (!has_ro) return;
        const int of = wave * 16 + ofc;
        f32x4 y = yacc;
        const int o0 = 4 * oq;
        const int fr = tab_fr[of];
        const float4* un4 = reinterpret_cast<const float4*>(tab_un + wave * 16 + o0);           // {1/scale, shift} x 4
        const float4 u01 = un4[0], u23 = un4[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] += __shfl_xor(y[j], 32);        // rows 8..15: the residual image's product
        y *= wo_inv;
        if (o0 < kfb_p) {
            if (write_fb) {
                const int pos = g.kfb - S16_MP + o0;                     // position inside the [U ; F] group
                TR::store4(zf + ((size_t)(wave * NKK + 16) * 64 + (pos >> 3) * 16 + ofc) * 16 + 2 * (pos & 7),
                           y[0], y[1], y[2], y[3]);
            }
            if (orow >= 0 && fr >= 0) {
                double* yo = p.Y + ((size_t)fr * out_rows + orow) * n_out;
                if ((n_out & 3) == 0) {
                    typedef double f64x2s __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<f64x2s*>(yo + o0) =
                        f64x2s{(double)((y[0] - u01.y) * u01.x), (double)((y[1] - u01.w) * u01.z)};
                    *reinterpret_cast<f64x2s*>(yo + o0 + 2) =
                        f64x2s{(double)((y[2] - u23.y) * u23.x), (double)((y[3] - u23.w) * u23.z)};
                } else {
                    const float2* un = tab_un + wave * 16 + o0;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (o0 + j < n_out) yo[o0 + j] = (double)((y[j] - un[j].y) * un[j].x);
                }
            }
        }
        yacc = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ================= skewed schedule =============================================================
    // slot 3s   : A P0(s) reads X_A(s)            B P2(s-1) writes X_B(s)
    // slot 3s+1 : A P1(s) reads X_B(s)            B P0(s)   reads X_A(s)     + read-out Y_s -> F_s (both sets)
    // slot 3s+2 : A P2(s) writes X_A(s+1)         B P1(s)   reads X_B(s)     + yU_s
    const int lag = wave >= NW / 2 ? 1 : 0;
    f32x4 acc[4][8];
    u32x4 abuf[2][4], b[4], ra[4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.packed_w) + (size_t)wset * p.wset_stride + p.w16_off), 0,
        S16_MP * NKK * 64, 0x00020000);
    const int w_frag0 = wave * NKK * 4;                    // fragment (wave, kk, m) at ((wave NKK + kk) 4 + m) KB
    auto loadA = [&](u32x4 (&a)[4], int kk) {             // kk >= NKK: zeros, no traffic
        const bool live = kk < NKK;
        const int voff = live ? lane16 : OOB;
#pragma unroll
        for (int m = 0; m < 4; ++m)
            a[m] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                w_rsrc, voff, live ? (w_frag0 + kk * 4 + m) * 1024 : 0, 0));
    };
    const __amdgpu_buffer_rsrc_t wo_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.packed_wout)
                          + (size_t)(has_ro ? own_grp : 0) * p.wout_stride + p.wo16_off),
        0, NKK * 1024 + 16, 0x00020000);
    // W_out fragments of one trip (two groups): ra[2j] = group kk0 + i + j (the half being multiplied),
    // ra[2j+1] = group (kk0 + i + j) ^ 8 (the other half); `on` false: zeros, no traffic
    auto load_ra = [&](int j, int kk, bool on) {
        on = on && kk < 2 * NKH;
        const int voff = on ? lane16 : OOB;
        ra[2 * j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wo_rsrc, voff, on ? kk * 1024 : 0, 0));
        ra[2 * j + 1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wo_rsrc, voff, on ? (kk ^ 8) * 1024 : 0, 0));
    };
    auto ro_prefetch = [&](int kk0, bool on) { load_ra(0, kk0, on); load_ra(1, kk0 + 1, on); };
    typedef const __attribute__((address_space(3))) u32x4* lds_frag_t;
    // State groups [kk0, kk0 + 8) out of abuf, two per trip; abuf[j] holds group kk0 + j on entry and kk0 + 8 + j on
    // exit.  B ring: slot n & 3 holds column tile n's fragment and is refilled behind that tile's four MFMAs with
    // tile n + 4's (n < 4: same group; else tile n - 4 of the next group) -- twelve MFMAs of look-ahead, 16 registers.
    // RO: the read-out of the own tile rides along, both halves of k per group position.
    auto gemm_half = [&](int kk0, auto ro_tag, bool ro_on) {
        constexpr bool RO = decltype(ro_tag)::value;
        const int voff_ro = ro_on ? lane16 : OOB;      // (step 0 and tiles of padding: zero fragments, no traffic)
        // (addresses re-derived from the lane id per call, opaque to the optimiser: kept live across the step loop they
        //  are spilled, and a scratch reload in front of the loop costs an s_waitcnt vmcnt(0) INSIDE it -- the reload is
        //  the youngest vector-memory operation, so waiting for it drains the whole weight prefetch every trip)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const uint32_t zb = (uint32_t)(uintptr_t)zf + (uint32_t)(ln << 4);
        uint32_t bp0 = zb + (uint32_t)kk0 * 1024;                               // tiles 0-3 / 4-7 of group kk0 + i
        uint32_t bp1 = bp0 + 4 * TILE_B;
#pragma unroll
        for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<lds_frag_t>((uintptr_t)(bp0 + n * TILE_B));
        uint32_t rp = zb + (uint32_t)(wave * TILE_B) + (uint32_t)kk0 * 1024;      // own tile, this half / other half
        uint32_t xp = zb + (uint32_t)(wave * TILE_B) + (uint32_t)(kk0 ^ 8) * 1024;
        int sA = (w_frag0 + (kk0 + 2) * 4) * 1024;        // weight fragments two groups ahead
        int sR = (kk0 + 2) * 1024;                        // W_out fragments one trip ahead
        auto trip = [&](int i, auto tail_tag) {
            constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kk = kk0 + i + j;
                u32x4 rb0, rbx;
#pragma unroll
                for (int n = 0; n < 8; ++n) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        TR::mma16(acc[m][n], abuf[j][m], b[n & 3]);
                        const int u = n * 4 + m;
                        if (RO) {      // own tile's fragments of this group and of its twin in the other half; the W_out
                                       // fragments of the next trip go out right behind the read-out MFMA that used them
                            if (u == 0) rb0 = *reinterpret_cast<lds_frag_t>((uintptr_t)(rp + j * 1024));
                            if (u == 1) rbx = *reinterpret_cast<lds_frag_t>((uintptr_t)(xp + j * 1024));
                            if (u == 8) {
                                TR::mma16(yacc, ra[2 * j], rb0);
                                ra[2 * j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                    wo_rsrc, TAIL ? OOB : voff_ro, TAIL ? 0 : sR + j * 1024, 0));
                            }
                            if (u == 10) {
                                TR::mma16(yacc, ra[2 * j + 1], rbx);
                                ra[2 * j + 1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                    wo_rsrc, TAIL ? OOB : voff_ro, TAIL ? 0 : (sR ^ 8192) + j * 1024, 0));
                            }
                        }
                        if (n == 7) {                     // weight fragment m of the group two ahead
                            if (TAIL) {
                                const bool live = kk + 2 < NKK;
                                abuf[j][m] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                    w_rsrc, live ? lane16 : OOB, live ? (w_frag0 + (kk + 2) * 4 + m) * 1024 : 0, 0));
                            } else {
                                abuf[j][m] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                    w_rsrc, lane16, sA + (j * 4 + m) * 1024, 0));
                            }
                        }
                        if (m == 3) {
                            if (n < 4) b[n] = *reinterpret_cast<lds_frag_t>((uintptr_t)(bp1 + n * TILE_B + j * 1024));
                            else b[n - 4] = *reinterpret_cast<lds_frag_t>((uintptr_t)(bp0 + (n - 4) * TILE_B + (j + 1) * 1024));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            asm volatile("v_add_u32 %0, 2048, %0" : "+v"(bp0));
            asm volatile("v_add_u32 %0, 2048, %0" : "+v"(bp1));
            if (RO) { rp += 2048; xp += 2048; }
            sA += 8192;
            sR += 2048;
        };
        // (Tried: a peeled first trip whose MFMAs start from C = 0 instead of the 128 v_mov of zero_acc -- the third copy of
        //  the trip body costs 20-35 spilled registers in the hot path: 14.5 vs 11.9 ms.)
        for (int i = 0; i < NKH - 2; i += 2) trip(i, std::false_type{});
        trip(NKH - 2, std::true_type{});
    };
    // [U ; F] group: abuf[0] holds group 16 on entry
    auto uf_group = [&](u32x4 ra_u) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const uint32_t up = (uint32_t)(uintptr_t)zf + (uint32_t)(ln << 4) + 16 * 1024;
#pragma unroll
        for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<lds_frag_t>((uintptr_t)(up + n * TILE_B));
        {   // yU_s = Wout[:, inputs] U_s (the feedback columns of that group carry zero weights)
            const u32x4 rb_u = *reinterpret_cast<lds_frag_t>((uintptr_t)(up + wave * TILE_B));
            TR::mma16(yacc, ra_u, rb_u);
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) {
#pragma unroll
            for (int m = 0; m < 4; ++m) TR::mma16(acc[m][n], abuf[0][m], b[n & 3]);
            if (n < 4) b[n] = *reinterpret_cast<lds_frag_t>((uintptr_t)(up + (n + 4) * TILE_B));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto next_step_A = [&]() { loadA(abuf[0], 0); loadA(abuf[1], 1); };

    // phase E for tile positions [N0, N1): eight values per lane and row-tile pair -> one 16-byte store
    constexpr bool PK_NOISE = NOISE == ESN_NOISE_COUNTER && std::is_same<TR, TraitsF16>::value;
    typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const _Float16 c1s = (_Float16)(1024.0f * n_c1);
    const h16x2 c1h = {c1s, c1s};
    const float t_bias = 1.0f + n_c0 - (float)c1s;
    const uint32_t seed_hi = (uint32_t)(p.seed >> 32);
    auto activate = [&](int s, auto n0_tag, auto n1_tag, int ln) {
        constexpr int N0 = decltype(n0_tag)::value, N1 = decltype(n1_tag)::value;
        const int g4 = ln >> 4, col = ln & 15;
        uint32_t k1[N1 - N0];
        int frs[N1 - N0];
#pragma unroll
        for (int n = N0; n < N1; ++n) {
            const int t = n;
            if (NOISE == ESN_NOISE_COUNTER) k1[n - N0] = tab_ks[(s & 1) * BT + t * 16 + col];
            if (NOISE == ESN_NOISE_TENSOR) frs[n - N0] = tab_fr[t * 16 + col];
        }
#pragma unroll
        for (int n = N0; n < N1; ++n) {
            const int t = n;
            uint32_t key = 0;
            const double* nz = nullptr;
            if (NOISE == ESN_NOISE_COUNTER)        // noise_key(seed, frame, step) + row4 stride: row4 = 16 wave + 4 m + g4
                key = k1[n - N0] + (uint32_t)(16 * wave + g4) * 0x9E3779B9U;
            if (NOISE == ESN_NOISE_TENSOR && frs[n - N0] >= 0)
                nz = p.noise_u + ((size_t)frs[n - N0] * p.S + s) * n_res;
            char* dst = zf + (size_t)t * TILE_B + (size_t)(2 * wave) * 1024 + (size_t)ln * 16;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                uint32_t out[4];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int m = 2 * pr + tt;
                    const f32x4 a4 = acc[m][n];
                    if constexpr (PK_NOISE) {
                        float tv[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float e = __builtin_amdgcn_exp2f(a4[j]);
                            tv[j] = fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), t_bias);
                        }
                        const uint32_t sq = noise_mix(key + (uint32_t)(4 * m) * 0x9E3779B9U);
                        const h16x2 w01 = __builtin_bit_cast(h16x2, __builtin_amdgcn_perm(0x3C3C3C3Cu, sq, 0x04010400u));
                        const h16x2 w23 = __builtin_bit_cast(h16x2, __builtin_amdgcn_perm(0x3C3C3C3Cu, sq, 0x04030402u));
                        const h16x2 t01 = __builtin_convertvector(f32x2{tv[0], tv[1]}, h16x2);
                        const h16x2 t23 = __builtin_convertvector(f32x2{tv[2], tv[3]}, h16x2);
                        out[2 * tt] = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(w01, c1h, t01));
                        out[2 * tt + 1] = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(w23, c1h, t23));
                    } else {
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = TR::act(a4[j]);
                        const int row = 64 * wave + 16 * m + 4 * g4;
                        if (NOISE == ESN_NOISE_COUNTER) {
                            const uint32_t sq = noise_mix(key + (uint32_t)(4 * m) * 0x9E3779B9U);
                            v[0] = fmaf((float)(sq & 0xffU), n_c1, v[0] + n_c0);
                            v[1] = fmaf((float)((sq >> 8) & 0xffU), n_c1, v[1] + n_c0);
                            v[2] = fmaf((float)((sq >> 16) & 0xffU), n_c1, v[2] + n_c0);
                            v[3] = fmaf((float)(sq >> 24), n_c1, v[3] + n_c0);
                        } else if (NOISE == ESN_NOISE_TENSOR) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (nz && row + j < n_res) v[j] += noise * ((float)nz[row + j] - 0.5f);
                        }
                        out[2 * tt] = TR::pack2(v[0], v[1]);
                        out[2 * tt + 1] = TR::pack2(v[2], v[3]);
                    }
                }
                *reinterpret_cast<u32x4*>(dst + pr * 1024) = u32x4{out[0], out[1], out[2], out[3]};
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    const std::integral_constant<int, 0> n_lo;
    const std::integral_constant<int, 4> n_mid;
    const std::integral_constant<int, 8> n_hi;

    // ---- inputs: set B stages them for all column tiles, two tiles per wave (as in the 32x32x16 kernel) ----------
    const int cpf = n_in / 2;                                   // 16-byte chunks per frame row
    constexpr int IN_TILES = NOWN / (NW / 2);
    const int in_c0 = IN_TILES * (wave - NW / 2);
    const int lcpf = __builtin_ctz(cpf), lkin = __builtin_ctz(kin_p);
    const size_t in_frame_bytes = (size_t)in_stride * 8;
    int j0;
    slot_group(p, slot0, j0);
    size_t u_base_frame = j0 < p.F ? (size_t)grp0 * p.F + j0 : ((size_t)grp0 + (p.spw ? p.n_wsets : 1)) * p.F;
    if (u_base_frame >= (size_t)p.n_frames) u_base_frame = (size_t)p.n_frames - 1;
    for (int i = tid; i < BT; i += NTHREADS) {
        const int fr = tab_fr[i];
        tab_off[i] = fr >= 0 ? (int)(((size_t)fr - u_base_frame) * in_frame_bytes) : -1;
    }
    __syncthreads();
    const size_t u_left = ((size_t)p.n_frames - u_base_frame) * in_frame_bytes;
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.U) + u_base_frame * in_frame_bytes), 0,
        (int)(u_left < 0x7fffffffu ? u_left : 0x7fffffffu), 0x00020000);
    auto dma_inputs_b = [&](int s) {
        const int row = s + p.in_row_off;
        const bool row_ok = row < p.T_in;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        int off[IN_TILES][2];
#pragma unroll
        for (int ti = 0; ti < IN_TILES; ++ti)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = 64 * i + ln;
                off[ti][i] = tab_off[(in_c0 + ti) * 16 + ((e >> lcpf) & 15)];
            }
#pragma unroll
        for (int ti = 0; ti < IN_TILES; ++ti) {
            const int c = in_c0 + ti;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = 64 * i + ln;
                const bool on = off[ti][i] >= 0 && row_ok && e < 16 * cpf;
                const int voff = on ? off[ti][i] + ((e & (cpf - 1)) << 4) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    u_rsrc, (__attribute__((address_space(3))) void*)(in_slots + (size_t)(c * 2 + i) * 1024), 16,
                    voff, row_ok ? row * n_in * 8 : 0, 0, 0);
            }
        }
    };
    auto commit_inputs_b = [&](int s) {
        const bool row_ok = s + p.in_row_off < p.T_in;
        const int lk2 = lkin - 1;
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int ti = 0; ti < IN_TILES; ++ti) {
            const int c = in_c0 + ti;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e2 = ln + 64 * k;
                const int f = (e2 >> lk2) & 15, c2 = e2 & ((kin_p >> 1) - 1), ci = 2 * c2;
                const bool live = tab_fr[c * 16 + f] >= 0 && ci < n_in;
                const float4 ss = *reinterpret_cast<const float4*>(tab_in + c * kin_p + ci);
                const int ch = (f << lcpf) + c2;
                const double2 raw = *reinterpret_cast<const double2*>(
                    in_slots + (size_t)(c * 2 + (ch >> 6)) * 1024 + (size_t)(ch & 63) * 16);
                const float v0 = live ? fmaf((float)(row_ok ? raw.x : 0.0), ss.x, ss.y) : 0.f;
                const float v1 = live ? fmaf((float)(row_ok ? raw.y : 0.0), ss.z, ss.w) : 0.f;
                if (e2 < 8 * kin_p)
                    TR::store2(zf + ((size_t)(c * NKK + 16) * 64 + (ci >> 3) * 16 + f) * 16 + 2 * (ci & 7), v0, v1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

#ifdef ESN_STAMPS
    unsigned long long sk_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define ESN_SK_ADD(i, a, b) sk_acc[i] += (b) - (a);
#else
#define ESN_SK_ADD(i, a, b)
#endif
    const uint32_t key1_own = tab_key[wave * 16 + (lane & 15)];
    next_step_A();
    ro_prefetch(0, false);
    if (lag) __syncthreads();                                      // slot 0: set A alone
    for (int s = 0; s < p.S; ++s) {
        const bool ro = has_ro && s > 0;
        ESN_STAMP(t0)
        zero_acc();
        // ---- P0 (set A: slot 3s; set B: slot 3s+1 with the read-out of Y_s).  ONE code path for both sets: a
        // branch around the MFMAs makes the compiler shuffle the accumulators between its arms (hundreds of spills);
        // where a wave does not read out, its W_out fragments are zero (no traffic) and the MFMAs add nothing
        if (lag) __builtin_amdgcn_s_setprio(1);
        if (NOISE == ESN_NOISE_COUNTER) {      // keys of step s for the own tile's frames: read from slot 3s+2 on
            const uint32_t k = mix32(key1_own ^ seed_hi ^ ((uint32_t)s * 0x85EBCA6BU + 0x27d4eb2fU));
            if (lane < 16) tab_ks[(s & 1) * BT + wave * 16 + lane] = k;
        }
        if (lag && s > 0) commit_inputs_b(s);
        gemm_half(0, std::true_type{}, ro && lag);
        if (lag) { if (ro) finish_readout(s - 1 - p.transient, true); }
        if (lag) __builtin_amdgcn_s_setprio(0);
        ro_prefetch(NKH, ro && !lag);                              // set A: for P1, in flight over the barrier
        ESN_STAMP(t1)
        __syncthreads();
        ESN_STAMP(t2)
        // ---- P1 (set A: slot 3s+1 with the read-out; set B: slot 3s+2)
        gemm_half(NKH, std::true_type{}, ro && !lag);
        if (!lag) { if (ro) finish_readout(s - 1 - p.transient, true); }
        ESN_STAMP(t3)
        __syncthreads();
        ESN_STAMP(t4)
        // ---- P2: [U ; F] group + phase E
        __builtin_amdgcn_s_setprio(2);
        const u32x4 ra_u = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
            wo_rsrc, has_ro ? lane16 : OOB, has_ro ? 16 * 1024 : 0, 0));
        int ln_e = lane;
        asm volatile("" : "+v"(ln_e));
        uf_group(ra_u);
        ESN_STAMP(u1)
        activate(s, n_lo, n_mid, ln_e);
        ESN_STAMP(u2)
        next_step_A();
        ro_prefetch(0, lag && has_ro && s + 1 < p.S);              // set B: for P0(s+1)
        ESN_STAMP(u3)
        if (lag && s + 1 < p.S) dma_inputs_b(s + 1);               // last LDS read of the phase is behind us
        ESN_STAMP(u4)
        activate(s, n_mid, n_hi, ln_e);
        ESN_STAMP(u5)
        __builtin_amdgcn_s_setprio(0);
        ESN_STAMP(t5)
        __syncthreads();
        ESN_STAMP(t6)
        ESN_SK_ADD(0, t0, t1) ESN_SK_ADD(1, t1, t2) ESN_SK_ADD(2, t2, t3)
        ESN_SK_ADD(3, t3, t4) ESN_SK_ADD(4, t4, t5) ESN_SK_ADD(5, t5, t6)
        ESN_SK_ADD(6, t4, u1) ESN_SK_ADD(7, u1, u2) ESN_SK_ADD(8, u2, u3) ESN_SK_ADD(9, u3, u4)
        ESN_SK_ADD(10, u4, u5) ESN_SK_ADD(11, u5, t5)
    }
#undef ESN_SK_ADD
#ifdef ESN_STAMPS
    if (p.stamps && blockIdx.x == 0 && lane == 0) {
        for (int i = 0; i < 6; ++i) p.stamps[wave * 8 + i] = sk_acc[i];
        for (int i = 0; i < 6; ++i) p.stamps[(8 + wave) * 8 + i] = sk_acc[6 + i];    // inside P2
        p.stamps[wave * 8 + 6] = (unsigned long long)__builtin_amdgcn_s_getreg(63492);   // HW_ID
    }
#endif
    if (!lag) __syncthreads();                                     // slot 3S: set B finishes X_B(S)
    if (has_ro) {                                                  // Y_S = yU_{S-1} + Wout_x X_S
        for (int kk = 0; kk < 2 * NKH; ++kk) {
            const u32x4 rb = *reinterpret_cast<const u32x4*>(zf + (size_t)(wave * NKK + kk) * 1024 + lane16);
            const u32x4 rw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wo_rsrc, lane16, kk * 1024, 0));
            TR::mma16(yacc, rw, rb);
        }
        finish_readout(p.S - 1 - p.transient, false);
    }
}

template <typename TR>
static int launch_skew16(const RecurParams& p, hipStream_t stream) {
    const int kin_p = p.g.kfb - p.g.kin;
    const size_t lds = (size_t)8 * S16_NKK * 1024 + 4 * 128 + 4 * 128 + 8 * (size_t)8 * (kin_p + 16) + 8 * 2048 + 4 * 128 + 8 * 128;
    auto go = [&](auto kern) -> int {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(p.n_tiles), dim3(512), lds, stream, p);
        return (int)hipGetLastError();
    };
    switch (p.noise_mode) {
        case ESN_NOISE_NONE: return go(recur_skew16_kernel<TR, ESN_NOISE_NONE>);
        case ESN_NOISE_TENSOR: return go(recur_skew16_kernel<TR, ESN_NOISE_TENSOR>);
        default: return go(recur_skew16_kernel<TR, ESN_NOISE_COUNTER>);
    }
}

}  // namespace esn
