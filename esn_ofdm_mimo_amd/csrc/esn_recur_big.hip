// Large reservoirs (N_res > 1024, BASELINE configs[4]: 2048): the recurrence as ONE GEMM LAUNCH PER
// TIMESTEP.  The persistent kernels keep the state of a frame tile in LDS and stream the weights; at
// N_res = 2048 the fp16 weight image is 8.5 MB -- beyond one XCD's L2 -- and 160 KB of LDS hold the
// state of only 32 frames, so every CU re-streamed 8.5 MB from the Infinity Cache per step for 32
// frames (6 % of the MFMA peak, round 1).  Here the state lives in HBM / Infinity Cache in MFMA
// fragment order and step s is a tiled GEMM with a fused epilogue:
//
//   X_{s+1}[Mp x F] = act( Wext[Mp x Kp] * [X_s ; U_s ; F_s] )          256 x 256 tile per workgroup
//
//   * both operands arrive by LDS-DMA in 1 KB fragments (the weight image of esn_pack_weights and
//     the state image are already in the order a wave's ds_read_b128 consumes them: no swizzle, no
//     conflicts), double-buffered 64-deep k-chunks, counted vmcnt + raw s_barrier;
//   * 8 waves = 2 (rows) x 4 (frames), 4 x 2 accumulator tiles of 32 x 32 each;
//   * a small "prep" launch ahead of every GEMM launch turns the previous launch's read-out partials
//     into Y_s (output row s-1 -> HBM, unscaled) and writes the [U_s ; F_s] k-groups of the state
//     image (scaled inputs, fed-back output), so the GEMM sees ONE uniform operand: Kp/64 chunks;
//   * epilogue: tanh + state noise -> fp16 -> next state image (coalesced 8-byte pieces), and the
//     read-out partial Wout[:, these 256 rows] x_{s+1} with the accumulator tile used directly as the
//     B operand of a 32x32x16 MFMA (guide: "an accumulator tile as the next MFMA's operand");
//   * workgroup -> tile map: the Mp/256 row tiles of one frame tile run side by side on one XCD (they
//     share the state fragments through that XCD's L2), frame tiles are dealt over the 8 XCDs.
//
// The launch boundaries are the per-step dependencies (X_{s+1} needs every row of X_s, Y_s every row
// tile's partial): two boundaries of ~1.5 us per step against ~0.3-0.5 ms of GEMM at the benchmark size.  Arithmetic and noise stream are those of the
// persistent fp16/bf16 kernels (same packed weights incl. the 2 log2 e pre-scale, same counter noise).
#include "esn_recur_mfma_impl.h"

namespace esn {

struct BigParams {
    RecurParams r;
    int n_slots;          // padded slot axis, multiple of 256
    int n_mt;             // row tiles of 256: Mp / 256
    int nkgS, nkg;        // 32-byte k-groups: state (Mp/16) and total (Kp/16, a multiple of 4)
    int step;
    const char* x_in;     // state image of step s   [n_slots/32][nkg][64 lanes][16 B]: state groups, then [U;F], zeros
    char* x_out;          // ... of step s+1
    const float* yp_in;   // [n_mt][n_slots][8] read-out partials (x gain) of X_s, written by the previous GEMM launch
    float* yp_out;
    size_t wo_big_off;    // byte offset of the big-path read-out image inside a group's packed read-out
};

template <typename E>
__device__ __forceinline__ u32x2 pack4(float a, float b, float c, float d) {
    typedef E vec4 __attribute__((ext_vector_type(4)));
    const vec4 v = {(E)a, (E)b, (E)c, (E)d};
    return __builtin_bit_cast(u32x2, v);
}

constexpr int BIG_STAGE = 65536;                      // one k-chunk: A 32 KB + B 32 KB
constexpr int BIG_LDS = 2 * BIG_STAGE;                // (the epilogue's 8 KB scratch aliases stage 0)

size_t big_workspace_bytes(int n_slots, int Mp, int Kp) {
    return 2 * ((size_t)n_slots * Kp * 2) + 2 * ((size_t)(Mp / 256) * n_slots * 8 * 4);
}
// per-group read-out image of this path: [Mp/32 row tiles][2 k-steps][64 lanes][16 B] (rows 0-7 hi, 8-15 lo
// of W_out gain, k in accumulator order), then (W_out gain)[:, inputs] as float [8][16], then {1/gain, gain, 0, 0}
size_t big_wout_image_bytes(int Mp) { return (size_t)Mp * 64 + 512 + 16; }

// both state images: X_0 = x0 of the frame's group (or zeros) in the first, zeros in the second; the
// [U;F] and padding k-groups zero in both (prep writes [U;F] every step, the padding groups stay zero)
template <typename TR>
__global__ void big_init_kernel(BigParams bp, char* x0_img, char* x1_img) {
    const RecurParams& p = bp.r;
    const size_t n = (size_t)(bp.n_slots / 32) * bp.nkg * 64;        // 16-byte pieces: [ct][kg][lane]
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t j = i >> 6;
        const int kg = (int)(j % bp.nkg);
        const int ct = (int)(j / bp.nkg);
        const int slot = ct * 32 + (lane & 31);
        int grp;
        const int fr = slot_frame(p, slot, grp);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = kg * 16 + 8 * (lane >> 5) + e;
            v[e] = (fr >= 0 && p.x0 && k < p.n_res) ? (float)p.x0[(size_t)grp * p.n_res + k] : 0.f;
        }
        *reinterpret_cast<u32x2*>(x0_img + i * 16) = pack4<typename TR::elem>(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<u32x2*>(x0_img + i * 16 + 8) = pack4<typename TR::elem>(v[4], v[5], v[6], v[7]);
        *reinterpret_cast<u32x4*>(x1_img + i * 16) = u32x4{0, 0, 0, 0};
    }
}

// prep(s), one thread per slot, ahead of GEMM launch s (and once more after the last one):
//   s >= 1:  Y_s = (1/gain) sum_m partials(X_s) + W_out[:, inputs] U_{s-1}    -> output row s-1 (unscaled)
//   s == 0:  Y_0 = y0 of the group (continuation) or 0
//   s <  S:  [U_s ; F_s = Y_s] -> k-groups nkgS, nkgS+1 of the image GEMM launch s reads (bp.x_out here)
// U_{s-1} is read back from the image of the previous step (bp.x_in), in the operand type -- the same
// rounded inputs the persistent kernels multiply with W_out.
template <typename TR>
__global__ __launch_bounds__(256) void big_prep_kernel(BigParams bp) {
    __shared__ float wou[17 * 128];                       // (W_out gain)[:, inputs] of the block's groups
    __shared__ float inv_gain[17];
    const RecurParams& p = bp.r;
    const int n_in = p.n_in, n_out = p.n_out;
    const int slot = blockIdx.x * 256 + threadIdx.x;
    const int grp_first = (blockIdx.x * 256) / p.Fpad;
    for (int i = threadIdx.x; i < 17 * 128; i += 256) {
        const int g = grp_first + (i >> 7);
        const char* gimg = reinterpret_cast<const char*>(p.packed_wout) + (size_t)(g < p.n_groups ? g : 0) * p.wout_stride + bp.wo_big_off;
        wou[i] = *reinterpret_cast<const float*>(gimg + (size_t)p.g.Mp * 64 + (size_t)(i & 127) * 4);
        if ((i & 127) == 0) inv_gain[i >> 7] = *reinterpret_cast<const float*>(gimg + (size_t)p.g.Mp * 64 + 512);
    }
    __syncthreads();
    int grp;
    const int fr = slot_frame(p, slot, grp);
    const bool live = fr >= 0;
    const size_t fr_c = live ? (size_t)fr : 0, grp_c = live ? (size_t)grp : 0;
    const int gi = live ? grp - grp_first : 0;
    const int s = bp.step;
    const int ct = slot >> 5, lr = slot & 31;
    const int kin_p = p.g.kfb - p.g.kin;
    // every load is unconditional (clamped indices, masked results): no branch, one wait
    float part[8][8];
#pragma unroll
    for (int mm = 0; mm < 8; ++mm) {
        const f32x4* src = reinterpret_cast<const f32x4*>(bp.yp_in + ((size_t)(mm < bp.n_mt ? mm : 0) * bp.n_slots + slot) * 8);
        const f32x4 a = src[0], b = src[1];
#pragma unroll
        for (int o = 0; o < 4; ++o) { part[mm][o] = a[o]; part[mm][4 + o] = b[o]; }
    }
    float uprev[16];                                       // k - kin = 0..15 of the previous step's [U;F] group
    {
        const char* g0 = bp.x_in + ((size_t)ct * bp.nkg + bp.nkgS) * 1024;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            float t4[4];
            TR::load4(g0 + (size_t)(lr + 32 * (c4 >> 1)) * 16 + 8 * (c4 & 1), t4);
            uprev[4 * c4] = t4[0]; uprev[4 * c4 + 1] = t4[1]; uprev[4 * c4 + 2] = t4[2]; uprev[4 * c4 + 3] = t4[3];
        }
    }
    const int row = s + p.in_row_off;
    const int row_c = row < p.T_in ? row : 0;
    const bool has_isc = p.in_scale != nullptr, has_ish = p.in_shift != nullptr, has_y0 = p.y0 != nullptr,
               has_tsc = p.t_scale != nullptr, has_tsh = p.t_shift != nullptr;
    const double* sc_in = p.in_scale ? p.in_scale : p.U;       // (any valid address; the value is masked below)
    const double* sh_in = p.in_shift ? p.in_shift : p.U;
    float uu[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ic = i < n_in ? i : 0;
        const double raw = p.U[(fr_c * p.T_in + row_c) * n_in + ic];
        const float scv = (float)sc_in[grp_c * n_in + ic], shv = (float)sh_in[grp_c * n_in + ic];   // loaded, then masked
        const float sc = has_isc ? scv : 1.f, sh = has_ish ? shv : 0.f;
        const float v = fmaf(row < p.T_in ? (float)raw : 0.f, sc, sh);       // rows past T_in: zeros BEFORE scaling
        uu[i] = (live && i < n_in) ? v : 0.f;
    }
    const double* y0p = p.y0 ? p.y0 : p.U;
    const double* tsc = p.t_scale ? p.t_scale : p.U;
    const double* tsh = p.t_shift ? p.t_shift : p.U;
    const int orow = s - 1 - p.transient;
    const int out_rows = p.S - p.transient;
    float yy[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        const int oc = o < n_out ? o : 0;
        const float y0l = (float)y0p[grp_c * n_out + oc];
        const double tscl = tsc[grp_c * n_out + oc], tshl = tsh[grp_c * n_out + oc];
        const float y0v = has_y0 ? y0l : 0.f;
        float y = 0.f;
#pragma unroll
        for (int mm = 0; mm < 8; ++mm) y += (mm < bp.n_mt) ? part[mm][o] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) y = fmaf(wou[gi * 128 + o * 16 + i], uprev[i], y);
        y *= inv_gain[gi];
        y = (s == 0) ? y0v : y;
        const double sc = has_tsc ? tscl : 1.0;
        const double sh = has_tsh ? tshl : 0.0;
        if (live && o < n_out && s > 0 && orow >= 0)
            p.Y[((size_t)fr * out_rows + orow) * n_out + o] = ((double)y - sh) / sc;
        yy[o] = (live && o < n_out) ? y : 0.f;
    }
    if (s < p.S) {
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                float e8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int kk = 16 * g2 + 8 * hh + e;              // k - kin
                    float v = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) v = (kk == i) ? uu[i] : v;
#pragma unroll
                    for (int o = 0; o < 8; ++o) v = (kk == kin_p + o) ? yy[o] : v;
                    e8[e] = v;
                }
                char* d = bp.x_out + ((size_t)ct * bp.nkg + bp.nkgS + g2) * 1024 + (size_t)(lr + 32 * hh) * 16;
                *reinterpret_cast<u32x2*>(d) = pack4<typename TR::elem>(e8[0], e8[1], e8[2], e8[3]);
                *reinterpret_cast<u32x2*>(d + 8) = pack4<typename TR::elem>(e8[4], e8[5], e8[6], e8[7]);
            }
    }
}

// NT = column tiles (of 32 frames) per wave: 2 -> 8 waves of 128 x 64 (two per SIMD, 128 accumulator registers), the round-2
// shape; 4 -> 4 waves of 128 x 128 (ONE per SIMD, 256 accumulator registers of its 512).  At NT = 2 a k-group costs a wave 6
// fragment reads for 8 MFMAs: 8 waves x 24 KB per 64-deep chunk + the 64 KB the LDS-DMA writes = 256 KB per 1024 MFMA cycles,
// i.e. the whole LDS bandwidth (256 B/clk) at full matrix rate -- the matrix pipe sat at 53 %.  At NT = 4 it is 8 reads for
// 16 MFMAs: 192 KB per 2048 cycles = 94 B/clk.
template <typename TR, int NOISE, int NT, bool PIPE>
__global__ __launch_bounds__(1024 / NT) void big_step_kernel(BigParams bp) {
    constexpr int NW = 16 / NT;                          // waves: 2 (rows) x NW/2 (frames)
    constexpr int WNC = NW / 2;                          // wave columns
    constexpr int DT = 8 / WNC;                          // 32-row / 32-frame operand tiles a wave fetches per chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const RecurParams& p = bp.r;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNC, wn = wave % WNC;          // 2 x WNC waves: rows 128 wm.., frames 32 NT wn..
    const int r = lane & 31, h = lane >> 5;
    const int n_res = p.n_res;
    const int nkg = bp.nkg;

    // workgroup -> (row tile m, frame tile n): the n_mt row tiles of a frame tile are consecutive ids on one XCD
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int m = local % bp.n_mt;
    const int n = (local / bp.n_mt) * 8 + xcd;
    const int slot0 = n * 256;
    if (slot0 >= bp.n_slots) return;

    char* stage = smem;                                               // [2][A 32 KB | B 32 KB]
    float* red = reinterpret_cast<float*>(stage);                     // epilogue: [256 frames][8] (aliases stage 0)

    // ---- main loop: Kp/64 chunks of four k-groups, LDS-DMA double-buffered ---------------------------
    const int lane16 = lane * 16;
    const size_t x_bytes = (size_t)bp.n_slots * p.g.Kp * 2;
    // the first half of the waves fetches DT row tiles of A each, the second half DT column tiles of B: 4 DT pieces each
    const bool is_a = wave < WNC;
    const int pair = (wave % WNC) * DT;
    const int src_t0 = (is_a ? (m * 8 + pair) : ((slot0 >> 5) + pair)) * nkg;
    const int dst_off = (is_a ? 0 : 32768) + pair * 4096;
    // one descriptor per wave, built from values the compiler can PROVE wave-uniform (readfirstlane of the
    // pointer halves and the size): otherwise every DMA is wrapped in a waterfall loop (guide T20)
    const uint64_t dma_ptr = is_a ? (uint64_t)reinterpret_cast<uintptr_t>(p.packed_w) : (uint64_t)reinterpret_cast<uintptr_t>(bp.x_in);
    const uint32_t dma_lo = __builtin_amdgcn_readfirstlane((uint32_t)dma_ptr);
    const uint32_t dma_hi = __builtin_amdgcn_readfirstlane((uint32_t)(dma_ptr >> 32));
    const int dma_size = __builtin_amdgcn_readfirstlane(is_a ? (int)p.wset_stride : (int)x_bytes);
    const __amdgpu_buffer_rsrc_t dma_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>((uintptr_t)(((uint64_t)dma_hi << 32) | dma_lo)), 0, dma_size, 0x00020000);
    auto issue = [&](int c, int buf) {
        char* dst = stage + (size_t)buf * BIG_STAGE + dst_off;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int kg = 0; kg < 4; ++kg)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(dma_rsrc, (__attribute__((address_space(3))) void*)(dst + (t * 4 + kg) * 1024),
                                                         16, lane16, (src_t0 + t * nkg + 4 * c + kg) * 1024, 0, 0);
    };
    f32x16 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
    auto compute = [&](int buf) {
        const char* ab = stage + (size_t)buf * BIG_STAGE + (size_t)(wm * 4) * 4096 + lane16;
        const char* bb = stage + (size_t)buf * BIG_STAGE + 32768 + (size_t)(wn * NT) * 4096 + lane16;
        if constexpr (NT == 4) {
            // one wave per SIMD: nobody else covers an LDS latency, so the fragments of k-group kg+1 are requested
            // BEFORE the 16 MFMAs of k-group kg (two register sets); left to itself the compiler reads one B fragment
            // at a time and waits for it in front of every four MFMAs
            u32x4 a[2][4], b[2][NT];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a[0][mt] = *reinterpret_cast<const u32x4*>(ab + mt * 4096);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[0][nt] = *reinterpret_cast<const u32x4*>(bb + nt * 4096);
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                const int cur = kg & 1, nxt = cur ^ 1;
                if (kg + 1 < 4) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) a[nxt][mt] = *reinterpret_cast<const u32x4*>(ab + mt * 4096 + (kg + 1) * 1024);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) b[nxt][nt] = *reinterpret_cast<const u32x4*>(bb + nt * 4096 + (kg + 1) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) TR::mma32(acc[mt][nt], a[cur][mt], b[cur][nt]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                u32x4 a[4], b[NT];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const u32x4*>(ab + mt * 4096 + kg * 1024);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const u32x4*>(bb + nt * 4096 + kg * 1024);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) TR::mma32(acc[mt][nt], a[mt], b[nt]);
            }
        }
    };
    if constexpr (PIPE) {
        // Round-3 main loop (NT = 2).  The round-2 loop below puts every wave's eight DMA launches of chunk c+1 in
        // front of the wait and the barrier of chunk c: all eight waves launch (8 x ~100 cycles each: the guide's
        // LDS-DMA issue price inside a busy phase), THEN all eight compute (1024 MFMA cycles per wave) -- the matrix
        // pipe idles through every launch burst (53 % busy).  Here a stage is 32 deep (32 KB), FOUR stages are in
        // flight, a stage's four launches per wave are spread between the MFMAs of the stage being multiplied (the
        // pipe keeps running while the wave issues them), a stage has a whole stage time to land before it is waited
        // for, and there is ONE barrier per stage: a buffer is refilled three stages after it was read, behind the
        // barrier every wave passes after reading it.
        static_assert(NT == 2, "pipelined main loop: 8 waves of 128 x 64");
        constexpr int STG = 32768;
        const int half_off = is_a ? 0 : 16384;
        auto issue_piece = [&](int st, int j) {               // j < 4: operand tile pair + (j >> 1), k-group 2 st + (j & 1)
            char* dst = stage + (size_t)(st & 3) * STG + half_off + (pair + (j >> 1)) * 2048 + (j & 1) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dma_rsrc, (__attribute__((address_space(3))) void*)dst, 16, lane16,
                                                     (src_t0 + (j >> 1) * nkg + 2 * st + (j & 1)) * 1024, 0, 0);
        };
        const int nst = nkg / 2;
#pragma unroll
        for (int st = 0; st < 3; ++st)
            if (st < nst) {
#pragma unroll
                for (int j = 0; j < 4; ++j) issue_piece(st, j);
            }
        for (int st = 0; st < nst; ++st) {
            if (st + 2 < nst) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // this wave's pieces of stage st have landed
            else if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // ... everybody's; and stage st-1 has been read by all
            const bool more = st + 3 < nst;
            const char* ab = stage + (size_t)(st & 3) * STG + (size_t)(wm * 4) * 2048 + lane16;
            const char* bb = stage + (size_t)(st & 3) * STG + 16384 + (size_t)(wn * 2) * 2048 + lane16;
#pragma unroll
            for (int kg = 0; kg < 2; ++kg) {
                u32x4 a[4], b[2];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const u32x4*>(ab + mt * 2048 + kg * 1024);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) b[nt] = *reinterpret_cast<const u32x4*>(bb + nt * 2048 + kg * 1024);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        TR::mma32(acc[mt][nt], a[mt], b[nt]);
                        if (mt == 1 && more) issue_piece(st + 3, kg * 2 + nt);      // in the issue shadow of the MFMAs
                    }
                }
            }
        }
        __syncthreads();                                           // stage buffers free (the epilogue's scratch aliases them)
    } else {
    const int nch = nkg / 4;
    issue(0, 0);
    for (int c = 0; c + 1 < nch; ++c) {
        issue(c + 1, (c + 1) & 1);
        if constexpr (DT == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // this wave's pieces of chunk c have landed
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // ... and everybody else's
        compute(c & 1);
        __builtin_amdgcn_s_barrier();                          // chunk c read: its buffer may be refilled
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    compute((nch - 1) & 1);
    __syncthreads();                                           // stage buffers free (the epilogue's scratch aliases them)
    }

    // ---- epilogue: activation + noise -> next state image; read-out partial ---------------------------
    const float noise = (float)p.noise;
    const float n_c1 = noise * (1.0f / 256.0f), n_c0 = noise * (0.5f / 256.0f - 0.5f);
    const char* wo_base = reinterpret_cast<const char*>(p.packed_wout) + bp.wo_big_off;
    const __amdgpu_buffer_rsrc_t xo_rsrc = __builtin_amdgcn_make_buffer_rsrc(bp.x_out, 0, (int)x_bytes, 0x00020000);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int fcol = wn * 32 * NT + nt * 32 + r;
        int fgrp;
        const int fr = slot_frame(p, slot0 + fcol, fgrp);
        const int ct_g = (slot0 >> 5) + wn * NT + nt;
        // the column tile's two 16-frame halves may belong to different groups (different W_out)
        const int g_lo = (ct_g * 32) / p.Fpad, g_hi = (ct_g * 32 + 16) / p.Fpad;       // wave-uniform
        uint32_t key = 0;
        const double* nz = nullptr;
        if (NOISE == ESN_NOISE_COUNTER) key = noise_key(p.seed, (uint32_t)fr + p.frame_off, (uint32_t)bp.step);
        if (NOISE == ESN_NOISE_TENSOR && fr >= 0) nz = p.noise_u + ((size_t)fr * p.S + bp.step) * n_res;
        f32x16 racc_lo, racc_hi;
#pragma unroll
        for (int i = 0; i < 16; ++i) { racc_lo[i] = 0.f; racc_hi[i] = 0.f; }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int rt_g = m * 8 + wm * 4 + mt;
            // W_out fragments of this row tile (two k-steps), issued ahead of the activation arithmetic
            u32x4 wa_lo[2], wa_hi[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                wa_lo[s2] = *reinterpret_cast<const u32x4*>(wo_base + (size_t)(g_lo < p.n_groups ? g_lo : 0) * p.wout_stride
                                                            + (size_t)(rt_g * 2 + s2) * 1024 + lane16);
                wa_hi[s2] = *reinterpret_cast<const u32x4*>(wo_base + (size_t)(g_hi < p.n_groups ? g_hi : 0) * p.wout_stride
                                                            + (size_t)(rt_g * 2 + s2) * 1024 + lane16);
            }
            u32x2 xh[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = rt_g * 32 + 8 * q + 4 * h;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = TR::act(acc[mt][nt][4 * q + j]);
                if (NOISE == ESN_NOISE_COUNTER) {
                    const uint32_t sq = noise_quad(key, (uint32_t)(row >> 2));
                    v[0] = fmaf((float)(sq & 0xffU), n_c1, v[0] + n_c0);
                    v[1] = fmaf((float)((sq >> 8) & 0xffU), n_c1, v[1] + n_c0);
                    v[2] = fmaf((float)((sq >> 16) & 0xffU), n_c1, v[2] + n_c0);
                    v[3] = fmaf((float)(sq >> 24), n_c1, v[3] + n_c0);
                } else if (NOISE == ESN_NOISE_TENSOR) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (nz && row + j < n_res) v[j] += noise * ((float)nz[row + j] - 0.5f);
                }
                xh[q] = pack4<typename TR::elem>(v[0], v[1], v[2], v[3]);
                // state image: k-group 2 rt + (q >> 1), lane r + 32 (q & 1), bytes 8 h .. 8 h + 7
                const int soff = ((ct_g * nkg + 2 * rt_g + (q >> 1)) * 64 + 32 * (q & 1)) * 16;
                __builtin_amdgcn_raw_buffer_store_b64(xh[q], xo_rsrc, r * 16 + 8 * h, soff, 0);
            }
            // read-out partial: the converted tile is the B operand (rows of X = k), k-step s2 = registers 8 s2 .. 8 s2 + 7
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const u32x4 xb = {xh[2 * s2][0], xh[2 * s2][1], xh[2 * s2 + 1][0], xh[2 * s2 + 1][1]};
                TR::mma32(racc_lo, wa_lo[s2], xb);
                TR::mma32(racc_hi, wa_hi[s2], xb);
            }
        }
        // rows 0-7 of the image hold hi(W_out gain), rows 8-15 the rounding residual: lane (r, h) sums
        // registers j and 4 + j into output o = 4 h + j of its frame
        float y4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) y4[j] = (r >= 16) ? racc_hi[j] + racc_hi[4 + j] : racc_lo[j] + racc_lo[4 + j];
        // sum over the two row halves of the workgroup (wm = 1 -> LDS -> wm = 0), then one 16-byte store
        if (wm == 1) *reinterpret_cast<f32x4*>(red + (size_t)fcol * 8 + 4 * h) = f32x4{y4[0], y4[1], y4[2], y4[3]};
        __syncthreads();
        if (wm == 0) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(red + (size_t)fcol * 8 + 4 * h);
            *reinterpret_cast<f32x4*>(bp.yp_out + ((size_t)m * bp.n_slots + slot0 + fcol) * 8 + 4 * h) =
                f32x4{y4[0] + o[0], y4[1] + o[1], y4[2] + o[2], y4[3] + o[3]};
        }
        __syncthreads();
    }
}

template <typename TR>
static int launch_big_t(const RecurParams& rp, size_t wo_big_off, void* workspace, hipStream_t stream) {
    BigParams bp;
    bp.r = rp;
    const int Mp = rp.g.Mp, Kp = rp.g.Kp;
    bp.n_slots = round_up(rp.n_groups * rp.Fpad, 256);
    bp.n_mt = Mp / 256;
    bp.nkgS = Mp / 16;
    bp.nkg = Kp / 16;
    bp.wo_big_off = wo_big_off;
    char* ws = reinterpret_cast<char*>(workspace);
    const size_t xb = (size_t)bp.n_slots * Kp * 2, yb = (size_t)bp.n_mt * bp.n_slots * 8 * 4;
    char* X[2] = {ws, ws + xb};
    float* YP[2] = {reinterpret_cast<float*>(ws + 2 * xb), reinterpret_cast<float*>(ws + 2 * xb + yb)};
    bp.step = 0; bp.x_in = X[1]; bp.x_out = X[0]; bp.yp_in = YP[0]; bp.yp_out = YP[1];
    hipLaunchKernelGGL(big_init_kernel<TR>, dim3(2048), dim3(256), 0, stream, bp, X[0], X[1]);
    hipError_t e = hipMemsetAsync(YP[0], 0, yb, stream);          // prep(0) reads (and discards) partials
    if (e != hipSuccess) return (int)e;
    const int n_nt8 = (bp.n_slots / 256 + 7) / 8;
    const dim3 grid(8 * n_nt8 * bp.n_mt);
    // main loop: 0 = round-2 (two 64-deep buffers, two barriers per chunk), 1 = round-3 pipeline (four 32-deep stages, DMA
    // launches between the MFMAs, one barrier per stage; default), 2 = 4 waves of 128 x 128 (a kept negative result)
    const int variant = knobs().big_nt == 4 ? 2 : (knobs().big_pipe ? 1 : 0);
    const int ni = rp.noise_mode == ESN_NOISE_NONE ? 0 : rp.noise_mode == ESN_NOISE_TENSOR ? 1 : 2;
#define ESN_BIG_K(NZ) {reinterpret_cast<const void*>(big_step_kernel<TR, NZ, 2, false>), \
                       reinterpret_cast<const void*>(big_step_kernel<TR, NZ, 2, true>),  \
                       reinterpret_cast<const void*>(big_step_kernel<TR, NZ, 4, false>)}
    const void* kern[3][3] = {ESN_BIG_K(ESN_NOISE_NONE), ESN_BIG_K(ESN_NOISE_TENSOR), ESN_BIG_K(ESN_NOISE_COUNTER)};
#undef ESN_BIG_K
    e = hipFuncSetAttribute(kern[ni][variant], hipFuncAttributeMaxDynamicSharedMemorySize, BIG_LDS);
    if (e != hipSuccess) return (int)e;
    const dim3 block(variant == 2 ? 256 : 512);
    for (int s = 0; s <= rp.S; ++s) {
        // prep(s): partials of X_s (from GEMM s-1, in YP[s & 1]) -> Y row s-1; [U_s ; F_s] -> image X[s & 1]
        bp.step = s;
        bp.x_in = X[(s + 1) & 1];          // previous step's image: its [U;F] group holds U_{s-1}
        bp.x_out = X[s & 1];
        bp.yp_in = YP[s & 1];
        hipLaunchKernelGGL(big_prep_kernel<TR>, dim3(bp.n_slots / 256), dim3(256), 0, stream, bp);
        if (s == rp.S) break;
        bp.x_in = X[s & 1]; bp.x_out = X[(s + 1) & 1];
        bp.yp_out = YP[(s + 1) & 1];
        void* args[] = {&bp};
        e = hipLaunchKernel(kern[ni][variant], grid, block, args, BIG_LDS, stream);
        if (e != hipSuccess) return (int)e;
    }
    return (int)hipGetLastError();
}

// shapes this path serves: fp16/bf16 predict, shared reservoir, n_in <= 16, n_out <= 8, Mp a multiple of 256
bool big_path_applies(int precision, const RecurParams& p) {
    return (precision == ESN_F16 || precision == ESN_BF16) && !p.harvest && p.n_wsets == 1 && p.n_res > 1024 &&
           p.n_in <= 16 && p.n_out <= 8 && p.g.Mp % 256 == 0 && p.g.Mp <= 2048 && p.g.Kp % 64 == 0 &&
           p.g.Kp - p.g.Mp >= 32 && p.g.kfb - p.g.kin + round_up(p.n_out, 4) <= 32 &&
           (size_t)round_up(p.n_groups * round_up(p.F, 16), 256) * p.g.Kp * 2 < 0x7fffffffu;
}
int big_slots(const RecurParams& p) { return round_up(p.n_groups * round_up(p.F, 16), 256); }

// =====================================================================================================
// Harvest (teacher-forced state collection of ESN.fit, pyESN.py:176-182) on the same launch-per-step
// GEMM.  A fit has ONE pilot per trained ESN, i.e. few columns (512 at the benchmark size), so the tile
// is cut for workgroup count instead of reuse: 128 rows x 64 pilots, 4 waves (one per SIMD) of 64 x 32
// each -- Mp/128 x n_slots/64 workgroups (128 at 2048 x 512) where the predict tile would give 16 and the
// persistent kernel ran 16 workgroups of 16 waves at 1.9 % of the MFMA peak.  Teacher forcing makes the
// step simpler than predict: no read-out, and [U_{s+1} ; F_{s+1}] depend on the inputs only, so the row
// tile m = 0 of every pilot tile writes them in its epilogue and there is ONE launch per step.
//   * operands by LDS-DMA in 1 KB fragments, FOUR 64-deep chunks in flight (24 KB each: 4 row tiles +
//     2 pilot tiles x 4 k-groups), counted vmcnt, ONE barrier per chunk (a buffer is refilled three
//     chunks after it was read, behind the barrier every wave passes after reading it);
//   * epilogue: tanh + state noise -> operand type -> next state image; the same values, as float32 /
//     float64, go through an LDS transpose so that the extended-state rows leave in 512-byte runs
//     (E[pilot][s+1][128 rows]) instead of 16-byte pieces 1 MB apart.
struct BigHarvestParams {
    RecurParams r;
    int n_slots;          // padded pilot axis, multiple of 64
    int n_mt;             // row tiles of 128: Mp / 128
    int nkgS, nkg;
    int step;
    const char* x_in;
    char* x_out;
};
constexpr int BH_NST = 6;                              // stage buffers: BH_NST - 1 chunks in flight (the stream is bound by
                                                       // bytes in flight x L2 latency: 4 buffers gave 38 GB/s per CU)
constexpr int BH_STAGE = 24576;                        // A 16 KB + B 8 KB
constexpr int BH_LDS = BH_NST * BH_STAGE;              // 144 KB (the epilogue's 33 KB transpose scratch aliases it)
constexpr int BH_ELD = 132;                            // floats per pilot row of the transpose scratch (128 + pad)

size_t big_harvest_workspace_bytes(int n_groups, int Kp) { return 2 * (size_t)round_up(n_groups, 64) * Kp * 2; }

// [U_s ; F_s] of pilot slot `slot` -> the two k-groups behind the state groups of `img` (inputs row s + 1
// scaled, teacher row s scaled: pyESN.py:180-182; the same float conversions as the persistent harvest)
template <typename TR>
__device__ __forceinline__ void bigh_write_uf(const BigHarvestParams& hp, char* img, int slot, int s) {
    const RecurParams& p = hp.r;
    const int n_in = p.n_in, n_out = p.n_out;
    const int kin_p = p.g.kfb - p.g.kin;
    const bool live = slot < p.n_groups;
    const size_t fr = live ? (size_t)slot : 0;
    const int T = p.S + 1;
    const int row = s + 1;
    float uu[16], ff[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ic = i < n_in ? i : 0;
        const double raw = (row < T) ? p.U[(fr * T + (row < T ? row : 0)) * n_in + ic] : 0.0;
        const double sc = p.in_scale ? p.in_scale[fr * n_in + ic] : 1.0;
        const double sh = p.in_shift ? p.in_shift[fr * n_in + ic] : 0.0;
        uu[i] = (live && i < n_in) ? (float)(raw * sc + sh) : 0.f;
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        const int oc = o < n_out ? o : 0;
        const double raw = p.D[(fr * T + s) * n_out + oc];
        const double sc = p.t_scale ? p.t_scale[fr * n_out + oc] : 1.0;
        const double sh = p.t_shift ? p.t_shift[fr * n_out + oc] : 0.0;
        ff[o] = (live && o < n_out && p.teacher_forcing) ? (float)(raw * sc + sh) : 0.f;
    }
    const int ct = slot >> 5, lr = slot & 31;
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            float e8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kk = 16 * g2 + 8 * hh + e;              // k - kin
                float v = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) v = (kk == i) ? uu[i] : v;
#pragma unroll
                for (int o = 0; o < 8; ++o) v = (kk == kin_p + o) ? ff[o] : v;
                e8[e] = v;
            }
            char* d = img + ((size_t)ct * hp.nkg + hp.nkgS + g2) * 1024 + (size_t)(lr + 32 * hh) * 16;
            *reinterpret_cast<u32x2*>(d) = pack4<typename TR::elem>(e8[0], e8[1], e8[2], e8[3]);
            *reinterpret_cast<u32x2*>(d + 8) = pack4<typename TR::elem>(e8[4], e8[5], e8[6], e8[7]);
        }
}

// both state images zero (X_0 = 0: a fit starts from the zero state, pyESN.py:177), then [U_0 ; F_0]
template <typename TR>
__global__ void bigh_init_kernel(BigHarvestParams hp, char* x0_img, char* x1_img) {
    const size_t n = (size_t)(hp.n_slots / 32) * hp.nkg * 64;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        *reinterpret_cast<u32x4*>(x0_img + i * 16) = u32x4{0, 0, 0, 0};
        *reinterpret_cast<u32x4*>(x1_img + i * 16) = u32x4{0, 0, 0, 0};
    }
}
template <typename TR>
__global__ void bigh_uf0_kernel(BigHarvestParams hp, char* x0_img) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot < hp.n_slots) bigh_write_uf<TR>(hp, x0_img, slot, 0);
}
// E[:, 0, :n_res] = 0 and the scaled-input columns of every row (pyESN.py:179,189)
__global__ void bigh_fill_e_kernel(BigHarvestParams hp) {
    const RecurParams& p = hp.r;
    const int T = p.S + 1, ncols = p.n_res + p.n_in;
    const size_t n_in_el = (size_t)p.n_groups * T * p.n_in, n_z = (size_t)p.n_groups * p.n_res;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_in_el + n_z; i += (size_t)gridDim.x * blockDim.x) {
        size_t idx; double v = 0.0;
        if (i < n_in_el) {
            const int ci = (int)(i % p.n_in);
            const size_t ft = i / p.n_in;                       // fr * T + t
            const size_t fr = ft / T;
            const double sc = p.in_scale ? p.in_scale[fr * p.n_in + ci] : 1.0;
            const double sh = p.in_shift ? p.in_shift[fr * p.n_in + ci] : 0.0;
            v = p.U[i] * sc + sh;
            idx = ft * ncols + p.n_res + ci;
        } else {
            const size_t j = i - n_in_el;
            idx = (j / p.n_res) * T * ncols + (j % p.n_res);
        }
        if (p.E32) p.E32[idx] = (float)v; else p.E[idx] = v;
    }
}

// KGC = k-groups per chunk: 4 (64 deep) where Kp is a multiple of 64 (reservoirs beyond 1024 units), 2 (32 deep) for the
// persistent kernels' images (Kp a multiple of 32: N_res = 512 has 34 k-groups)
template <typename TR, int NOISE, int KGC>
__global__ __launch_bounds__(256) void bigh_step_kernel(BigHarvestParams hp) {
    constexpr int TILE_B = KGC * 1024;                   // one 32-row / 32-pilot operand tile of a chunk
    constexpr int A_B = 4 * TILE_B, STAGE_B = 6 * TILE_B;
    constexpr int PPC = KGC + KGC / 2;                   // DMA pieces per wave and chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const RecurParams& p = hp.r;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;             // 2 x 2 waves: rows 64 wm.., pilots 32 wn..
    const int r = lane & 31, h = lane >> 5;
    const int n_res = p.n_res, nkg = hp.nkg;
    const int n_ft = hp.n_slots >> 6;
    // workgroup -> (row tile m, pilot tile n): an XCD keeps a contiguous band of row tiles (its slice of the
    // weights stays in that L2) and sees every pilot tile
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    int m, n;
    if (hp.n_mt % 8 == 0) { const int per = hp.n_mt / 8; m = xcd * per + local % per; n = local / per; }
    else { m = local % hp.n_mt; n = (local / hp.n_mt) * 8 + xcd; }
    if (n >= n_ft) return;
    const int slot0 = n * 64;

    const int lane16 = lane * 16;
    const size_t x_bytes = (size_t)hp.n_slots * p.g.Kp * 2;
    auto uniform_rsrc = [](const void* ptr, int bytes) {
        const uint64_t a = (uint64_t)reinterpret_cast<uintptr_t>(ptr);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>((uintptr_t)(((uint64_t)hi << 32) | lo)), 0,
                                                 __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t a_rsrc = uniform_rsrc(p.packed_w, (int)p.wset_stride);
    const __amdgpu_buffer_rsrc_t b_rsrc = uniform_rsrc(hp.x_in, (int)x_bytes);
    // per chunk: wave w fetches row tile w of A (KGC k-groups) and half a pilot tile of B (KGC / 2 k-groups)
    const int a_src = (m * 4 + wave) * nkg;
    const int b_src = ((slot0 >> 5) + wn) * nkg + (KGC / 2) * wm;
    auto issue = [&](int c, int buf) {
        char* st = smem + (size_t)buf * STAGE_B;
#pragma unroll
        for (int kg = 0; kg < KGC; ++kg)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(st + (wave * KGC + kg) * 1024),
                                                     16, lane16, (a_src + KGC * c + kg) * 1024, 0, 0);
#pragma unroll
        for (int kg = 0; kg < KGC / 2; ++kg)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (__attribute__((address_space(3))) void*)(st + A_B + (wn * KGC + (KGC / 2) * wm + kg) * 1024),
                                                     16, lane16, (b_src + KGC * c + kg) * 1024, 0, 0);
    };
    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
    auto compute = [&](int buf) {
        const char* ab = smem + (size_t)buf * STAGE_B + (size_t)(wm * 2) * TILE_B + lane16;
        const char* bb = smem + (size_t)buf * STAGE_B + A_B + (size_t)wn * TILE_B + lane16;
#pragma unroll
        for (int kg = 0; kg < KGC; ++kg) {
            const u32x4 b = *reinterpret_cast<const u32x4*>(bb + kg * 1024);
            const u32x4 a0 = *reinterpret_cast<const u32x4*>(ab + kg * 1024);
            const u32x4 a1 = *reinterpret_cast<const u32x4*>(ab + TILE_B + kg * 1024);
            TR::mma32(acc[0], a0, b);
            TR::mma32(acc[1], a1, b);
        }
    };
    const int nch = nkg / KGC;
#pragma unroll
    for (int c = 0; c < BH_NST - 1; ++c)
        if (c < nch) issue(c, c);
    int buf = 0, buf_issue = BH_NST - 1;                       // c % BH_NST and (c + BH_NST - 1) % BH_NST
    for (int c = 0; c < nch; ++c) {
        // this wave's six pieces of chunk c have landed when at most the later chunks' are outstanding
        const int later = nch - 1 - c < BH_NST - 2 ? nch - 1 - c : BH_NST - 2;
        static_assert(PPC == 6 || PPC == 3, "DMA pieces per wave and chunk");
        if constexpr (PPC == 6) {
            switch (later) {
                case 4: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        } else {
            switch (later) {
                case 4: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        }
        __builtin_amdgcn_s_barrier();                          // everybody's pieces of chunk c; chunk c-1 read by all
        if (c + BH_NST - 1 < nch) issue(c + BH_NST - 1, buf_issue);
        compute(buf);
        buf = buf + 1 == BH_NST ? 0 : buf + 1;
        buf_issue = buf_issue + 1 == BH_NST ? 0 : buf_issue + 1;
    }
    __syncthreads();                                           // stage buffers free: the transpose scratch aliases them

    // ---- epilogue ------------------------------------------------------------------------------------
    const float noise = (float)p.noise;
    const float n_c1 = noise * (1.0f / 256.0f), n_c0 = noise * (0.5f / 256.0f - 0.5f);
    const __amdgpu_buffer_rsrc_t xo_rsrc = __builtin_amdgcn_make_buffer_rsrc(hp.x_out, 0, (int)x_bytes, 0x00020000);
    float* es = reinterpret_cast<float*>(smem);                // [64 pilots][BH_ELD]
    const int fcol = wn * 32 + r;
    const int slot = slot0 + fcol;
    const int fr = slot < p.n_groups ? slot : -1;
    const int ct_g = (slot0 >> 5) + wn;
    uint32_t key = 0;
    const double* nz = nullptr;
    if (NOISE == ESN_NOISE_COUNTER) key = noise_key(p.seed, (uint32_t)fr + p.frame_off, (uint32_t)hp.step);
    if (NOISE == ESN_NOISE_TENSOR && fr >= 0) nz = p.noise_u + ((size_t)fr * p.S + hp.step) * n_res;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int rt_l = wm * 2 + mt, rt_g = m * 4 + rt_l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = rt_g * 32 + 8 * q + 4 * h;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = TR::act(acc[mt][4 * q + j]);
            if (NOISE == ESN_NOISE_COUNTER) {
                const uint32_t sq = noise_quad(key, (uint32_t)(row >> 2));
                v[0] = fmaf((float)(sq & 0xffU), n_c1, v[0] + n_c0);
                v[1] = fmaf((float)((sq >> 8) & 0xffU), n_c1, v[1] + n_c0);
                v[2] = fmaf((float)((sq >> 16) & 0xffU), n_c1, v[2] + n_c0);
                v[3] = fmaf((float)(sq >> 24), n_c1, v[3] + n_c0);
            } else if (NOISE == ESN_NOISE_TENSOR) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (nz && row + j < n_res) v[j] += noise * ((float)nz[row + j] - 0.5f);
            }
            typedef typename TR::elem EL;
            const EL e0 = (EL)v[0], e1 = (EL)v[1], e2 = (EL)v[2], e3 = (EL)v[3];     // the state IS the rounded value
            typedef EL vec4 __attribute__((ext_vector_type(4)));
            const u32x2 xh = __builtin_bit_cast(u32x2, (vec4){e0, e1, e2, e3});
            const int soff = ((ct_g * nkg + 2 * rt_g + (q >> 1)) * 64 + 32 * (q & 1)) * 16;
            __builtin_amdgcn_raw_buffer_store_b64(xh, xo_rsrc, r * 16 + 8 * h, soff, 0);
            *reinterpret_cast<f32x4*>(es + (size_t)fcol * BH_ELD + rt_l * 32 + 8 * q + 4 * h) =
                f32x4{(float)e0, (float)e1, (float)e2, (float)e3};
        }
    }
    __syncthreads();
    // extended-state row s+1, rows [128 m, 128 m + 128) of every pilot of the tile: a wave writes two pilots per
    // instruction, 512 (float32) or 1024 (float64) contiguous bytes each
    {
        const int ncols = n_res + p.n_in;
        const int T = p.S + 1;
        const int rr = 4 * (lane & 31);
        const int row0 = m * 128 + rr;
        const bool vec_ok = (n_res % 4 == 0) && (ncols % 4 == 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = wave * 16 + 2 * i + (lane >> 5);
            const int sl = slot0 + f;
            if (sl >= p.n_groups) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(es + (size_t)f * BH_ELD + rr);
            const size_t base = ((size_t)sl * T + hp.step + 1) * ncols + row0;
            if (vec_ok && row0 + 3 < n_res) {
                if (p.E32) {
                    *reinterpret_cast<f32x4*>(p.E32 + base) = v;
                } else {
                    typedef double f64x2 __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<f64x2*>(p.E + base) = f64x2{(double)v[0], (double)v[1]};
                    *reinterpret_cast<f64x2*>(p.E + base + 2) = f64x2{(double)v[2], (double)v[3]};
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (row0 + j < n_res) { if (p.E32) p.E32[base + j] = v[j]; else p.E[base + j] = (double)v[j]; }
            }
        }
    }
    // the next step's [U ; F] k-groups of this pilot tile
    if (m == 0 && hp.step + 1 < p.S && tid < 64) bigh_write_uf<TR>(hp, hp.x_out, slot0 + tid, hp.step + 1);
}

template <typename TR>
static int launch_bigh_t(const RecurParams& rp, void* workspace, hipStream_t stream) {
    BigHarvestParams hp;
    hp.r = rp;
    const int Mp = rp.g.Mp, Kp = rp.g.Kp;
    hp.n_slots = round_up(rp.n_groups, 64);
    hp.n_mt = Mp / 128;
    hp.nkgS = Mp / 16;
    hp.nkg = Kp / 16;
    char* ws = reinterpret_cast<char*>(workspace);
    const size_t xb = (size_t)hp.n_slots * Kp * 2;
    char* X[2] = {ws, ws + xb};
    hp.step = 0; hp.x_in = X[0]; hp.x_out = X[1];
    hipLaunchKernelGGL(bigh_init_kernel<TR>, dim3(1024), dim3(256), 0, stream, hp, X[0], X[1]);
    hipLaunchKernelGGL(bigh_uf0_kernel<TR>, dim3((hp.n_slots + 255) / 256), dim3(256), 0, stream, hp, X[0]);
    hipLaunchKernelGGL(bigh_fill_e_kernel, dim3(1024), dim3(256), 0, stream, hp);
    const int n_ft = hp.n_slots / 64;
    const dim3 grid(hp.n_mt % 8 == 0 ? hp.n_mt * n_ft : 8 * ((n_ft + 7) / 8) * hp.n_mt);
    const bool deep = (Kp % 64) == 0;                      // 64-deep chunks where the image allows, else 32-deep
    const int ni = rp.noise_mode == ESN_NOISE_NONE ? 0 : rp.noise_mode == ESN_NOISE_TENSOR ? 1 : 2;
    const void* kern[2][3] = {
        {reinterpret_cast<const void*>(bigh_step_kernel<TR, ESN_NOISE_NONE, 2>),
         reinterpret_cast<const void*>(bigh_step_kernel<TR, ESN_NOISE_TENSOR, 2>),
         reinterpret_cast<const void*>(bigh_step_kernel<TR, ESN_NOISE_COUNTER, 2>)},
        {reinterpret_cast<const void*>(bigh_step_kernel<TR, ESN_NOISE_NONE, 4>),
         reinterpret_cast<const void*>(bigh_step_kernel<TR, ESN_NOISE_TENSOR, 4>),
         reinterpret_cast<const void*>(bigh_step_kernel<TR, ESN_NOISE_COUNTER, 4>)}};
    const void* kfn = kern[deep ? 1 : 0][ni];
    const int lds = BH_LDS > 36 * 1024 ? BH_LDS : 36 * 1024;     // (the transpose scratch needs 33 KB whatever the stage size)
    hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    for (int s = 0; s < rp.S; ++s) {
        hp.step = s; hp.x_in = X[s & 1]; hp.x_out = X[(s + 1) & 1];
        void* args[] = {&hp};
        e = hipLaunchKernel(kfn, grid, dim3(256), args, lds, stream);
        if (e != hipSuccess) return (int)e;
    }
    return (int)hipGetLastError();
}

// shapes the harvest GEMM serves: fp16/bf16, shared reservoir, reservoirs beyond 1024 units
bool big_harvest_applies(int precision, const RecurParams& p) {
    // (Reservoirs of 257..1024 units can run here too -- 32-deep chunks, KGC = 2; debug knob "harvest_gemm" -- and were
    //  measured at N_res = 512, 2048 pilots: 137 launches of 128 workgroups took ~1.8 ms against 1.16 ms for the persistent
    //  kernel's 64 workgroups; the launch chain, not the tile, is the cost.  Off by default.)
    return (precision == ESN_F16 || precision == ESN_BF16) && p.harvest && p.n_wsets == 1 &&
           (p.n_res > 1024 || (knobs().harvest_gemm && p.n_res > 256 && p.n_groups >= 64)) &&
           p.n_in <= 16 && p.n_out <= 8 && p.g.Mp % 128 == 0 && p.g.Kp % 32 == 0 && p.g.Kp / 64 >= BH_NST &&
           p.g.Kp - p.g.Mp >= 32 && p.g.kfb - p.g.kin + round_up(p.n_out, 4) <= 32 &&
           (size_t)round_up(p.n_groups, 64) * p.g.Kp * 2 < 0x7fffffffu && p.wset_stride < 0x7fffffffu;
}

int launch_harvest_big(int precision, const RecurParams& p, void* workspace, hipStream_t stream) {
    if (precision == ESN_F16) return launch_bigh_t<TraitsF16>(p, workspace, stream);
    if (precision == ESN_BF16) return launch_bigh_t<TraitsBF16>(p, workspace, stream);
    return -1;
}

int launch_recur_big(int precision, const RecurParams& p, size_t wo_big_off, void* workspace, hipStream_t stream) {
    if (precision == ESN_F16) return launch_big_t<TraitsF16>(p, wo_big_off, workspace, stream);
    if (precision == ESN_BF16) return launch_big_t<TraitsBF16>(p, wo_big_off, workspace, stream);
    return -1;
}

}  // namespace esn
