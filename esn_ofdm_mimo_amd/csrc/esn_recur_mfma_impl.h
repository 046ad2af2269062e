// MFMA recurrence kernels (float32 / fp16 / bf16 operands, float32 accumulate).
//
// One workgroup owns a tile of Bt = 32*NT sequences for all S timesteps; nothing
// leaves the CU between steps.  Per step:
//
//   phase G1 P  = Wext[:, state k] * X_s                 Z = [X_s ; U_s ; F_s] lives in LDS as
//            Y_s = yU_{s-1} + Wout[:, state k] * X_s    Zt[frame][k] (k contiguous) = B operand
//            A operands (Wext, and Wout of the column's group) are pre-packed in MFMA
//            fragment order and streamed from L2 inside the same k-loop; wave w owns rows
//            [32*MT*w, 32*MT*(w+1)) x all Bt columns of P, and -- as "column owner" c = w <
//            Bt/16 -- the 16x16 readout tile of frames 16c..16c+15.
//            owners: F_s <- Y_s into Zt, unscaled Y_s -> HBM (row s-1, if past the transient)
//   barrier  (F_s visible)                                    [skipped when harvesting]
//   phase G2 P += Wext[:, input+feedback k] * [U_s ; F_s];  owners: yU_s = Wout[:, input k] * U_s
//   barrier  (all reads of Zt done)
//   phase E  X_{s+1} = tanh(P) + noise*(u-0.5) -> Zt state rows;  owners: U_{s+1} (prefetched at
//            the top of the step) and, when harvesting, the teacher F_{s+1} -> Zt
//   barrier
// after the last step the owners run one more readout pass for Y_S.  The readout therefore
// costs 16x16 MFMAs inside the GEMM loop instead of a serial chain of L2 loads per step.
//
// K is consumed in 32-byte groups of each LDS row: lane (r = lane&31, h = lane>>5)
// takes bytes [32*kg + 16*h, +16) of row r of both operands, so one ds_read_b128 /
// one 16-byte global load feeds 4 x v_mfma_f32_32x32x2_f32 (float32) or
// 1 x v_mfma_f32_32x32x16_{f16,bf16}.  (The k order inside a group is a fixed
// permutation applied to both operands, which a dot product does not see.)
#pragma once
#include <type_traits>
#include "esn_common.h"

namespace esn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));

struct TraitsF32 {
    typedef float elem;
    static constexpr int ES = 4;
    static constexpr int PARTS = 1;   // readout images (1 = W_out as is)
    static __device__ __forceinline__ void mma32(f32x16& c, u32x4 a, u32x4 b) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const uint32_t ai = a[i], bi = b[i];   // copy out: bit_cast of a vector element lvalue reads lane 0
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ai), __uint_as_float(bi), c, 0, 0, 0);
        }
    }
    // 64-byte row group, lane quarter q takes 16 B: 4 x (16x16x4)
    static __device__ __forceinline__ void mma16(f32x4& c, u32x4 a, u32x4 b) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const uint32_t ai = a[i], bi = b[i];
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ai), __uint_as_float(bi), c, 0, 0, 0);
        }
    }
    static __device__ __forceinline__ void store4(char* dst, float v0, float v1, float v2, float v3) {
        f32x4 v = {v0, v1, v2, v3};
        *reinterpret_cast<f32x4*>(dst) = v;
    }
    static __device__ __forceinline__ void store1(char* dst, float v) { *reinterpret_cast<float*>(dst) = v; }
    static __device__ __forceinline__ void store2(char* dst, float v0, float v1) {
        *reinterpret_cast<float2*>(dst) = make_float2(v0, v1);
    }
    static __device__ __forceinline__ float load1(const char* src) { return *reinterpret_cast<const float*>(src); }
    static __device__ __forceinline__ void load4(const char* src, float (&v)[4]) {     // 16-byte aligned
        const f32x4 t = *reinterpret_cast<const f32x4*>(src);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
    static __device__ __forceinline__ float act(float x) { return tanh_f32(x); }
    // when every pre-activation of a wave is below TANH32_SERIES_MAX the select in tanh_f32 always takes the
    // series: evaluate only that (bit-identical result, half the instructions)
    static constexpr bool HAS_SMALL = true;
    static __device__ __forceinline__ float act_small(float x) { return tanh_f32_series(x); }
};

struct TraitsF16 {
    typedef _Float16 elem;
    static constexpr int ES = 2;
    static constexpr int PARTS = 2;   // W_out = hi + lo (two fp16 images)
    static __device__ __forceinline__ void mma32(f32x16& c, u32x4 a, u32x4 b) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a),
                                                   __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma16(f32x4& c, u32x4 a, u32x4 b) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a),
                                                   __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void store4(char* dst, float v0, float v1, float v2, float v3) {
        h16x4 v = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
        *reinterpret_cast<h16x4*>(dst) = v;
    }
    static __device__ __forceinline__ void store1(char* dst, float v) { *reinterpret_cast<_Float16*>(dst) = (_Float16)v; }
    static __device__ __forceinline__ void store2(char* dst, float v0, float v1) {      // 4-byte aligned
        typedef _Float16 h16x2v __attribute__((ext_vector_type(2)));
        *reinterpret_cast<h16x2v*>(dst) = h16x2v{(_Float16)v0, (_Float16)v1};
    }
    static __device__ __forceinline__ uint32_t pack2(float v0, float v1) {
        typedef _Float16 h16x2v __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, h16x2v{(_Float16)v0, (_Float16)v1});
    }
    static __device__ __forceinline__ void unpack2(uint32_t w, float& v0, float& v1) {
        typedef _Float16 h16x2v __attribute__((ext_vector_type(2)));
        const h16x2v h = __builtin_bit_cast(h16x2v, w);
        v0 = (float)h[0]; v1 = (float)h[1];
    }
    static __device__ __forceinline__ float load1(const char* src) { return (float)*reinterpret_cast<const _Float16*>(src); }
    static __device__ __forceinline__ void load4(const char* src, float (&v)[4]) {     // 8-byte aligned
        const h16x4 t = *reinterpret_cast<const h16x4*>(src);
        v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
    }
    static __device__ __forceinline__ float act(float z) { return tanh_prescaled(z); }   // weights carry 2 log2 e
    static constexpr bool HAS_SMALL = false;
    static __device__ __forceinline__ float act_small(float z) { return tanh_prescaled(z); }
};

struct TraitsBF16 {
    typedef __bf16 elem;
    static constexpr int ES = 2;
    static constexpr int PARTS = 2;
    static __device__ __forceinline__ void mma32(f32x16& c, u32x4 a, u32x4 b) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b16x8, a),
                                                    __builtin_bit_cast(b16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma16(f32x4& c, u32x4 a, u32x4 b) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b16x8, a),
                                                    __builtin_bit_cast(b16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void store4(char* dst, float v0, float v1, float v2, float v3) {
        b16x4 v = {(__bf16)v0, (__bf16)v1, (__bf16)v2, (__bf16)v3};
        *reinterpret_cast<b16x4*>(dst) = v;
    }
    static __device__ __forceinline__ void store1(char* dst, float v) { *reinterpret_cast<__bf16*>(dst) = (__bf16)v; }
    static __device__ __forceinline__ void store2(char* dst, float v0, float v1) {      // 4-byte aligned
        typedef __bf16 b16x2v __attribute__((ext_vector_type(2)));
        *reinterpret_cast<b16x2v*>(dst) = b16x2v{(__bf16)v0, (__bf16)v1};
    }
    static __device__ __forceinline__ uint32_t pack2(float v0, float v1) {
        typedef __bf16 b16x2v __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, b16x2v{(__bf16)v0, (__bf16)v1});
    }
    static __device__ __forceinline__ void unpack2(uint32_t w, float& v0, float& v1) {
        v0 = __uint_as_float(w << 16); v1 = __uint_as_float(w & 0xffff0000u);
    }
    static __device__ __forceinline__ float load1(const char* src) { return (float)*reinterpret_cast<const __bf16*>(src); }
    static __device__ __forceinline__ void load4(const char* src, float (&v)[4]) {     // 8-byte aligned
        const b16x4 t = *reinterpret_cast<const b16x4*>(src);
        v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
    }
    static __device__ __forceinline__ float act(float z) { return tanh_prescaled(z); }   // weights carry 2 log2 e
    static constexpr bool HAS_SMALL = false;
    static __device__ __forceinline__ float act_small(float z) { return tanh_prescaled(z); }
};

template <typename TR, int NW, int MT, int NT, bool HARVEST, int NOISE, bool SKEW>
__global__ __launch_bounds__(NW * 64) void recur_mfma_kernel(RecurParams p) {
    extern __shared__ __attribute__((aligned(16))) char zt[];   // Zt[Bt][Ks] elements, then tables
    constexpr int ES = TR::ES;
    constexpr int BT = 32 * NT;
    constexpr int NTHREADS = NW * 64;
    constexpr int NOWN = BT / 16;                       // 16-frame column tiles of the workgroup
    constexpr int OC = (NOWN + NW - 1) / NW;            // column tiles owned per wave: c = wave + i*NW
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const Geometry& g = p.g;
    const int n_res = p.n_res, n_in = p.n_in, n_out = p.n_out;
    const int row_bytes = g.Ks * ES;
    const int nkg = g.Kp * ES / 32;          // 32-byte k-groups per row
    const int nkgS = g.Mp * ES / 32;         // ... of which state rows
    const int nk64 = nkg / 2;                // 64-byte groups (readout MFMA granularity)
    const int nk64S = nkgS / 2;
    const int kin_p = g.kfb - g.kin;
    const int kfb_p = round_up(n_out, 4);
    const int n_ot = (n_out + 15) / 16;      // == 1 (mfma_geometry rejects n_out > 16)
    const int out_rows = p.S - p.transient;
    const int ncols = n_res + n_in;
    // harvest output: float64 (the reference's dtype) or float32 (exact for the state columns of the
    // MFMA kernels, 6e-8 relative on the input columns; half the store tail and half the solve's reads)
    auto store_E = [&](size_t idx, double v) {
        if (p.E32) p.E32[idx] = (float)v; else p.E[idx] = v;
    };

    const int tile = xcd_remap(blockIdx.x, p.n_tiles);
    const int slot0 = tile * BT;
    const int grp0 = slot_group(p, slot0);
    if (grp0 >= p.n_groups) return;
    const int wset = slot_wset(p, slot0);
    const char* wp = reinterpret_cast<const char*>(p.packed_w) + (size_t)wset * p.wset_stride
                     + ((size_t)(wave * MT) * nkg * 64 + lane) * 16;

    // small per-tile tables behind the Zt image (instead of per-lane registers):
    //   tab_fr[BT]            frame index of each tile-local frame (-1 = padding slot)
    //   tab_in[NOWN][kin_p]   {scale, shift} of the column tile's group, per input column
    //   tab_un[NOWN][16]      {1/t_scale, t_shift} of the column tile's group, per output
    int* tab_fr = reinterpret_cast<int*>(zt + (size_t)BT * row_bytes);
    float2* tab_in = reinterpret_cast<float2*>(tab_fr + BT);
    float2* tab_un = tab_in + NOWN * kin_p;
    // raw float64 input rows of the NEXT step, filled by LDS-DMA (no VGPRs): [NOWN][16][n_in]
    double* in_raw = reinterpret_cast<double*>(tab_un + NOWN * 16);
    // skewed schedule: byte offset of each frame's input block from the tile's base frame (-1 = padding slot)
    int* tab_off = reinterpret_cast<int*>(in_raw + (size_t)(SKEW ? NOWN * 256 : BT * n_in));
    for (int i = tid; i < BT; i += NTHREADS) { int gtmp; tab_fr[i] = slot_frame(p, slot0 + i, gtmp); }
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
    // ---- LDS init: state rows, padding, initial feedback -----------------------
    for (int i = tid; i < BT * g.Ks; i += NTHREADS) {
        int f = i / g.Ks, k = i % g.Ks;
        float v = 0.f;
        int pg;
        const int fr = slot_frame(p, slot0 + f, pg);
        if (fr >= 0) {
            if (k < n_res) {
                if (p.x0) v = (float)p.x0[(size_t)pg * n_res + k];
            } else if (k >= g.kfb && k < g.kfb + n_out && !HARVEST) {
                if (p.y0) v = (float)p.y0[(size_t)pg * n_out + (k - g.kfb)];
            }
        }
        TR::store1(zt + (size_t)i * ES, v);
    }
    __syncthreads();

    // ---- column ownership: wave owns tiles c_i = wave + i*NW (i < OC) of 16 frames each ------
    // readout view of a lane inside a 16x16 tile: frame column ofc, output rows 4*oq .. 4*oq+3
    const int oq = lane >> 4, ofc = lane & 15;
    const int oq_w = oq, ofc_w = ofc;
    const char* wop[OC];          // this group's packed W_out (+ lane offset), nullptr = no readout
    float wo_inv[OC];
    bool own[OC];
#pragma unroll
    for (int i = 0; i < OC; ++i) {
        const int c = wave + i * NW;
        own[i] = c < NOWN;
        wop[i] = nullptr; wo_inv[i] = 1.f;
        if (!HARVEST && own[i]) {
            const int cg = slot_group(p, slot0 + c * 16);
            if (cg < p.n_groups) {
                const char* base = reinterpret_cast<const char*>(p.packed_wout) + (size_t)cg * p.wout_stride;
                wop[i] = base + (size_t)lane * 16;
                wo_inv[i] = *reinterpret_cast<const float*>(base + (size_t)g.ro_parts * n_ot * nk64 * 1024);
            }
        }
    }

    // inputs of recurrence step s (input row s + in_row_off) for owned tile c, element e of its
    // 16 x kin_p block; when harvesting also writes the input columns of E
    const uint32_t in_stride = (uint32_t)(p.T_in * n_in);
    auto load_in = [&](int s, int c, int e) -> float {
        const int f = c * 16 + e / kin_p, ci = e % kin_p;
        const int fr = tab_fr[f];
        float v = 0.f;
        if (fr >= 0 && ci < n_in) {
            const int row = s + p.in_row_off;
            double raw = (row < p.T_in) ? p.U[(size_t)fr * in_stride + (size_t)row * n_in + ci] : 0.0;
            const float2 ss = tab_in[c * kin_p + ci];
            if (HARVEST) {
                int pg;
                slot_frame(p, slot0 + f, pg);
                double sc = p.in_scale ? p.in_scale[(size_t)pg * n_in + ci] : 1.0;
                double sh = p.in_shift ? p.in_shift[(size_t)pg * n_in + ci] : 0.0;
                double sv = raw * sc + sh;
                store_E(((size_t)fr * (p.S + 1) + row) * ncols + n_res + ci, sv);
                v = (float)sv;
            } else {
                v = fmaf((float)raw, ss.x, ss.y);
            }
        }
        return v;
    };
    auto stage_inputs = [&](int s) {
#pragma unroll
        for (int i = 0; i < OC; ++i) {
            if (!own[i]) continue;
            const int c = wave + i * NW;
            for (int e = lane; e < 16 * kin_p; e += 64)
                TR::store1(zt + (size_t)(c * 16 + e / kin_p) * row_bytes + (size_t)(g.kin + e % kin_p) * ES,
                           load_in(s, c, e));
        }
    };
    auto stage_teacher = [&](int s) {
#pragma unroll
        for (int i = 0; i < OC; ++i) {
            if (!own[i]) continue;
            const int c = wave + i * NW;
            for (int e = lane; e < 16 * kfb_p; e += 64) {
                const int f = c * 16 + e / kfb_p, co = e % kfb_p;
                int pg;
                const int fr = slot_frame(p, slot0 + f, pg);
                float v = 0.f;
                if (fr >= 0 && co < n_out) {
                    double raw = p.D[((size_t)fr * (p.S + 1) + s) * n_out + co];
                    double sc = p.t_scale ? p.t_scale[(size_t)pg * n_out + co] : 1.0;
                    double sh = p.t_shift ? p.t_shift[(size_t)pg * n_out + co] : 0.0;
                    v = (float)(raw * sc + sh);
                }
                TR::store1(zt + (size_t)f * row_bytes + (size_t)(g.kfb + co) * ES, v);
            }
        }
    };
    // LDS-DMA path for the per-step inputs (predict): every 16-byte chunk of the owned frames' input
    // row goes global -> LDS with no register staging; the owner wave converts it in phase E.
    const int cpf = n_in / 2;                                   // 16-byte chunks per frame row
    const bool in_dma = !HARVEST && (n_in % 2 == 0) && ((size_t)p.T_in * n_in % 2 == 0);
    auto dma_inputs = [&](int s) {
        const int row = s + p.in_row_off;
#pragma unroll
        for (int i = 0; i < OC; ++i) {
            if (!own[i]) continue;
            const int c = wave + i * NW;
            for (int e0 = 0; e0 < 16 * cpf; e0 += 64) {          // wave-uniform trip count
                const int e = e0 + lane;
                const int f = e / cpf, ch = e % cpf;
                const int fr = (e < 16 * cpf) ? tab_fr[c * 16 + f] : -1;
                if (fr >= 0 && row < p.T_in) {
                    const double* src = p.U + (size_t)fr * in_stride + (size_t)row * n_in + 2 * ch;
                    // destination = wave-uniform base + lane*16 (hardware): chunk e of this tile
                    char* dst = reinterpret_cast<char*>(in_raw + (size_t)c * 16 * n_in) + (size_t)e0 * 16;
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)src,
                        (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                }
            }
        }
    };
    auto commit_inputs = [&](int s) {                             // after the issuing wave's vmcnt(0)
        const int row = s + p.in_row_off;
#pragma unroll
        for (int i = 0; i < OC; ++i) {
            if (!own[i]) continue;
            const int c = wave + i * NW;
            for (int e = lane; e < 16 * kin_p; e += 64) {
                const int f = e / kin_p, ci = e % kin_p;
                float v = 0.f;
                if (tab_fr[c * 16 + f] >= 0 && ci < n_in) {
                    const float2 ss = tab_in[c * kin_p + ci];
                    const double raw = (row < p.T_in) ? in_raw[((size_t)c * 16 + f) * n_in + ci] : 0.0;
                    v = fmaf((float)raw, ss.x, ss.y);
                }
                TR::store1(zt + (size_t)(c * 16 + f) * row_bytes + (size_t)(g.kin + ci) * ES, v);
            }
        }
    };
    stage_inputs(0);
    if (HARVEST) {
        stage_teacher(0);
        // E row 0 = [0, scale(u[0])]  (pyESN.py:179,189): wave w writes frames w, w+NW, ...
        for (int f = wave; f < BT; f += NW) {
            int pg;
            const int fr = slot_frame(p, slot0 + f, pg);
            if (fr < 0) continue;
            const size_t er0 = ((size_t)fr * (p.S + 1)) * ncols;
            for (int c = lane; c < ncols; c += 64) {
                double v = 0.0;
                if (c >= n_res) {
                    const int ci = c - n_res;
                    double raw = p.U[((size_t)fr * p.T_in) * n_in + ci];
                    double sc = p.in_scale ? p.in_scale[(size_t)pg * n_in + ci] : 1.0;
                    double sh = p.in_shift ? p.in_shift[(size_t)pg * n_in + ci] : 0.0;
                    v = raw * sc + sh;
                }
                store_E(er0 + c, v);
            }
        }
    }
    __syncthreads();

    const float noise = (float)p.noise;
    const float n_c1 = noise * (1.0f / 256.0f), n_c0 = noise * (0.5f / 256.0f - 0.5f);
    const char* bbase = zt + (size_t)r * row_bytes + 16 * h;
    // readout B rows of the owned tiles: frame c*16 + ofc, bytes 64 kk + 16 oq
    const char* zrow0 = zt + (size_t)(wave * 16 + ofc) * row_bytes + 16 * oq;
    const size_t zrow_step = (size_t)NW * 16 * row_bytes;
    f32x4 yacc[OC];
#pragma unroll
    for (int i = 0; i < OC; ++i) yacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool ro_simple = (g.ro_parts == 1);
    // (a per-tile rotation of the k walk was measured SLOWER than every CU streaming the same
    //  weight lines at the same time: 27.9k vs 23.1k cycles in G1 -- keep it at 0)
    constexpr int krot = 0;

    // readout MFMAs over 64-byte groups [k0, k1) of the owned tiles, accumulating into yacc
    auto readout_groups = [&](int k0, int k1) {
        for (int kk = k0; kk < k1; ++kk) {
#pragma unroll
            for (int i = 0; i < OC; ++i) {
                if (wop[i]) {
                    const u32x4 rb = *reinterpret_cast<const u32x4*>(zrow0 + i * zrow_step + kk * 64);
                    for (int part = 0; part < g.ro_parts; ++part) {
                        const u32x4 ra = *reinterpret_cast<const u32x4*>(wop[i] + ((size_t)part * nk64 + kk) * 1024);
                        TR::mma16(yacc[i], ra, rb);
                    }
                }
            }
        }
    };
    // Y complete: feedback rows into Zt, unscaled output row `orow` to HBM, reset yacc
    auto finish_readout = [&](int orow, bool write_fb) {
        // (opaque copies of the lane coordinates when skewed: the addresses below are then derived here, once
        //  per step, instead of living in registers across the GEMM phases of a kernel that has none to spare)
        int ofc = ofc_w, oq = oq_w;
        if (SKEW) asm volatile("" : "+v"(ofc), "+v"(oq));
#pragma unroll
        for (int i = 0; i < OC; ++i) {
            if (!wop[i]) continue;
            const int c = wave + i * NW;
            const int of = c * 16 + ofc;
            f32x4 y = yacc[i];
            const int o0 = 4 * oq;
            // the table reads first: their LDS latency runs under the fold below
            const int fr = tab_fr[of];
            const float4* un4 = reinterpret_cast<const float4*>(tab_un + c * 16 + o0);         // {1/scale, shift} x 4
            const float4 u01 = un4[0], u23 = un4[1];
            if (g.ro_fold) {          // rows 8..15 (lanes 32..63) hold the residual image's product
                // (a v_permlane32_swap of the value with itself would give both halves to every lane without the
                //  trip through the LDS crossbar; tried here, it produced wrong sums -- not understood, not kept)
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] += __shfl_xor(y[j], 32);
            }
            y *= wo_inv[i];
            if (o0 < kfb_p) {
                if (write_fb)
                    TR::store4(zt + (size_t)of * row_bytes + (size_t)(g.kfb + o0) * ES, y[0], y[1], y[2], y[3]);
                if (orow >= 0 && fr >= 0) {
                    double* yo = p.Y + ((size_t)fr * out_rows + orow) * n_out;
                    if ((n_out & 3) == 0) {                                // whole quads: two 16-byte stores
                        typedef double f64x2s __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<f64x2s*>(yo + o0) =
                            f64x2s{(double)((y[0] - u01.y) * u01.x), (double)((y[1] - u01.w) * u01.z)};
                        *reinterpret_cast<f64x2s*>(yo + o0 + 2) =
                            f64x2s{(double)((y[2] - u23.y) * u23.x), (double)((y[3] - u23.w) * u23.z)};
                    } else {
                        const float2* un = tab_un + c * 16 + o0;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (o0 + j < n_out) yo[o0 + j] = (double)((y[j] - un[j].y) * un[j].x);
                    }
                }
            }
            yacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    bool any_ro = false;
#pragma unroll
    for (int i = 0; i < OC; ++i) any_ro = any_ro || (wop[i] != nullptr);

    if constexpr (SKEW) {
        // ================= skewed schedule (predict) ==================================
        // The in-step schedule below serialises MFMA work (G) and VALU work (E) behind barriers.
        // Here the waves form two sets, A = waves [0, NW/2) owning rows / state k-groups of the
        // lower half and B = the upper half, one wave of each per SIMD.  Every wave walks the same
        // cyclic program  P0: k-groups of half A | P1: k-groups of half B | P2: [U;F] groups + E
        // with a barrier after each phase, set B one phase behind set A:
        //     slot 3s   : A P0(s) reads X_A(s)            B P2(s-1) writes X_B(s)
        //     slot 3s+1 : A P1(s) reads X_B(s)            B P0(s)   reads X_A(s)     + readout Y_s -> F_s
        //     slot 3s+2 : A P2(s) writes X_A(s+1)         B P1(s)   reads X_B(s)     + yU_s
        // so the activation of one wave runs beside the MFMAs of the other wave of its SIMD, the
        // state stays single-buffered and the weights are still streamed once per step.  Every
        // accumulator sums the k-groups in the same order as the in-step schedule.
        static_assert(!HARVEST && OC == 1 && NW % 2 == 0, "skewed schedule: predict, one column tile per wave");
        const int lag = wave >= NW / 2 ? 1 : 0;
        const int nkgH = nkgS / 2;               // k-groups per half (multiple of 4: mfma_geometry)
        const int nk64H = nk64S / 2;             // == trips per half == readout groups per half
        const int n_uf = nkg - nkgS;             // [U;F] k-groups, <= 4
        f32x16 acc[MT][NT];
        u32x4 abuf[4][MT], bA[NT], ra[4];
        auto zero_acc = [&]() {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
        };
        // Weight streams by buffer loads: one descriptor per stream, the per-lane part (lane*16) in
        // one VGPR, the fragment offset in an SGPR -- no per-load 64-bit address arithmetic.
        // A fragments: FOUR k-groups in flight per wave.  Buffer j holds k-groups = j (mod 4) and is
        // refilled with the group four ahead right after its last MFMA.  (A rotating register
        // pipeline does not survive the compiler: it re-times the loads to "when the register
        // frees", about one k-group of look-ahead, and an L2 hit costs 1-2k cycles under this load.)
        // Nothing in the GEMM code is conditional: a branch around a load makes the compiler fall
        // back to s_waitcnt vmcnt(0) at the loop head, and a branch around MFMAs makes it shuffle
        // the accumulators between the arms.  A load that must not happen gets the per-lane offset
        // OOB (>= the descriptor's num_records): the buffer unit returns zeros without touching
        // memory, and zero fragments leave the accumulators unchanged.
        const int lane16 = lane * 16;
        constexpr int OOB = 0x7ffffff0;
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(p.packed_w) + (size_t)wset * p.wset_stride), 0,
            (int)p.wset_stride, 0x00020000);
        const int w_row0 = wave * MT * nkg;
#ifdef ESN_KO_A      // knock-out build (timing only, wrong results): no weight traffic, MFMAs on zeros
        constexpr bool KO_A = true;
#else
        constexpr bool KO_A = false;
#endif
        auto loadA = [&](u32x4 (&a)[MT], int kg) {       // kg >= nkg: no load, zeros
            const bool live = !KO_A && kg < nkg;
            const int voff = live ? lane16 : OOB;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                a[mt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    w_rsrc, voff, live ? (w_row0 + mt * nkg + kg) * 1024 : 0, 0));
        };
        auto loadA1 = [&](u32x4& a, int mt, int kg) {    // one row tile's fragment of k-group kg
            const bool live = !KO_A && kg < nkg;
            a = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                w_rsrc, live ? lane16 : OOB, live ? (w_row0 + mt * nkg + kg) * 1024 : 0, 0));
        };
        auto loadB = [&](u32x4 (&b)[NT], int kg) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                b[nt] = *reinterpret_cast<const u32x4*>(bbase + (size_t)nt * 32 * row_bytes + kg * 32);
        };
        auto mma_all = [&](const u32x4 (&a)[MT], const u32x4 (&b)[NT]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) TR::mma32(acc[mt][nt], a[mt], b[nt]);
        };
        // W_out fragments of the owned column tile, two trips ahead: ra[2i], ra[2i+1] = 64-byte
        // groups t and t + nk64H of the trips t = i (mod 2); `on` false: zeros, no traffic
        const int own_grp = __builtin_amdgcn_readfirstlane(slot_group(p, slot0 + wave * 16));
        const __amdgpu_buffer_rsrc_t wo_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(p.packed_wout)
                              + (size_t)(own_grp < p.n_groups ? own_grp : 0) * p.wout_stride),
            0, (int)p.wout_stride, 0x00020000);
        auto load_ra = [&](int i, int t, bool on) {
            on = on && t < nk64H;
            const int voff = on ? lane16 : OOB;
            ra[2 * i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                wo_rsrc, voff, on ? t * 1024 : 0, 0));
            ra[2 * i + 1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                wo_rsrc, voff, on ? (t + nk64H) * 1024 : 0, 0));
        };
        auto load_ra1 = [&](u32x4& dst, int t, int part_off, bool on) {   // one half of load_ra
            on = on && t < nk64H;
            dst = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                wo_rsrc, on ? lane16 : OOB, on ? (t + part_off) * 1024 : 0, 0));
        };
        auto ro_prefetch = [&](bool on) { load_ra(0, 0, on); load_ra(1, 1, on); };
        // State k-groups [kg0, kg0 + nkgH) out of abuf, four per trip of the loop; abuf[j] holds group
        // kg0 + j on entry and kg0 + nkgH + j on exit (past the last [U;F] group: zeros -- the next
        // step's first groups are fetched half-way through phase E, when half of the accumulators
        // have been retired).  The readout of the owned column tile rides along, 64-byte groups t
        // and t + nk64H in trip t; in the slots where this wave does not read out, ra[] is zero.
        auto gemm_half = [&](int kg0, bool ro_on) {
            loadB(bA, kg0);
            u32x4 rb0, rb1;
#ifdef ESN_GEMM_BLOCKED
            // (round-1 placement, kept for A/B runs: all MFMAs of the k-group, then all of its loads)
            for (int i = 0; i < nkgH; i += 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kg = kg0 + i + j;
                    const int t = (i + j) >> 1;
                    mma_all(abuf[j], bA);
                    if (j % 2 == 1) { TR::mma16(yacc[0], ra[j - 1], rb0); TR::mma16(yacc[0], ra[j], rb1); }
                    __builtin_amdgcn_sched_barrier(0);
                    loadA(abuf[j], kg + 4);
                    loadB(bA, kg + 1);
                    if (j % 2 == 0) {
                        rb0 = *reinterpret_cast<const u32x4*>(zrow0 + t * 64);
                        rb1 = *reinterpret_cast<const u32x4*>(zrow0 + (t + nk64H) * 64);
                    } else {
                        load_ra(j >> 1, t + 2, ro_on);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#else
            // Hand-placed k-groups.  A wave issues in order and the SIMD issues about one instruction per four
            // cycles for BOTH of its waves, so every instruction of this loop is paid twice: as matrix-pipe idle
            // time when it sits between two MFMA bursts, and as an issue slot the partner wave's phase E does not
            // get (knock-out builds, DESIGN 3.1b: with NO weight and NO state traffic the blocked loop still
            // spent 440 cycles per k-group on 256 cycles of MFMAs).  Hence
            //  * placement: the MFMAs run column-tile-major; B fragment nt is dead after its MT MFMAs and is
            //    re-read for the next k-group right there, 6-7 MFMAs before its next use (single buffering is
            //    enough); the A fragments are four k-groups ahead anyway and are refilled behind the last
            //    column tile.  Every reload is pinned into the 32-cycle issue shadow of the MFMA that frees its
            //    register -- no load above an MFMA that still reads its destination.  The readout pair of an
            //    odd k-group goes first, so that the W_out fragments it read are refilled in the first shadows;
            //  * count: addresses are running values stepped once per trip of four k-groups, the k-group's own
            //    piece sits in the instruction's offset field, and nothing is conditional except in the last
            //    trip of a half (the only one whose look-ahead can leave the image).
            const int voff_ro = ro_on ? lane16 : OOB;
            // LDS address of B fragment nt of k-group kg0 + i + 1
            uint32_t bp[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                bp[nt] = (uint32_t)(uintptr_t)(bbase + (size_t)nt * 32 * row_bytes + (kg0 + 1) * 32);
            int sA[MT];                                      // A fragment of row tile mt, k-group kg0 + i + 4
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) sA[mt] = (w_row0 + mt * nkg + kg0 + 4) * 1024;
            int sR0 = 2 * 1024, sR1 = (2 + nk64H) * 1024;   // W_out fragments two trips of the readout ahead
            auto trip = [&](int i, auto tail_tag) {
                constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kg = kg0 + i + j;
                    const int t = (i + j) >> 1;
                    if (j % 2 == 1) {
                        TR::mma16(yacc[0], ra[j - 1], rb0);
                        TR::mma16(yacc[0], ra[j], rb1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            TR::mma32(acc[mt][nt], abuf[j][mt], bA[nt]);
                            const int m = nt * MT + mt;
#ifndef ESN_KO_PURE  // knock-out build (timing only, wrong results): the k-group is its MFMAs and nothing else
                            if (j % 2 == 0) {
                                if (m == 0) rb0 = *reinterpret_cast<const u32x4*>(zrow0 + t * 64);
                                if (m == 1) rb1 = *reinterpret_cast<const u32x4*>(zrow0 + (t + nk64H) * 64);
                            } else if (TAIL) {
                                if (m == 0) load_ra1(ra[j - 1], t + 2, 0, ro_on);
                                if (m == 1) load_ra1(ra[j], t + 2, nk64H, ro_on);
                            } else {
                                if (m == 0) ra[j - 1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                wo_rsrc, voff_ro, sR0 + (j >> 1) * 1024, 0));
                                if (m == 1) ra[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                wo_rsrc, voff_ro, sR1 + (j >> 1) * 1024, 0));
                            }
                            if (nt == NT - 1) {
                                if (TAIL) loadA1(abuf[j][mt], mt, kg + 4);
                                else abuf[j][mt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                         w_rsrc, KO_A ? OOB : lane16, sA[mt] + j * 1024, 0));
                            }
                            if (mt == MT - 1) bA[nt] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(
                                                  (uintptr_t)(bp[nt] + j * 32));
#endif
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                // one step of the running addresses per trip
                // (in place: left to itself the compiler keeps the base AND a stepped copy per fragment alive)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) asm volatile("v_add_u32 %0, 128, %0" : "+v"(bp[nt]));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) sA[mt] += 4096;
                sR0 += 2048; sR1 += 2048;
            };
            for (int i = 0; i < nkgH - 4; i += 4) trip(i, std::false_type{});
            trip(nkgH - 4, std::true_type{});
#endif
        };
        auto uf_groups = [&]() {
            loadB(bA, nkgS);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < n_uf) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {                // same placement as gemm_half
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) TR::mma32(acc[mt][nt], abuf[j][mt], bA[nt]);
                        if (j + 1 < n_uf)
                            bA[nt] = *reinterpret_cast<const u32x4*>(bbase + (size_t)nt * 32 * row_bytes + (nkgS + j + 1) * 32);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        };
        auto next_step_A = [&]() {
#pragma unroll
            for (int j = 0; j < 4; ++j) loadA(abuf[j], j);
        };
        // E for column tiles [NT0, NT1).  fp16 + counter noise takes a packed-half tail: the four noise
        // bytes of a quad become two pairs of halves 1 + byte/1024 (one v_perm_b32 each, high byte
        // 0x3C), and   x = tanh + noise ((byte + 1/2)/256 - 1/2)   is one v_pk_fma_f16 per pair,
        //   x = (1 + byte/1024) c1 + t,   t = half(1 - 2/(1 + 2^z) + c0 - c1),  c1 = half(4 noise)
        // instead of four conversions, four adds and four fmas in float32.  (c1 is a normal half --
        // noise/256 itself would be subnormal -- and t is offset by the ROUNDED c1, so the rounding
        // of c1 changes the noise width by 2^-11 relative and adds no bias.)
        constexpr bool PK_NOISE = NOISE == ESN_NOISE_COUNTER && std::is_same<TR, TraitsF16>::value;
        typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const _Float16 c1s = (_Float16)(1024.0f * n_c1);
        const h16x2 c1h = {c1s, c1s};
        const float t_bias = 1.0f + n_c0 - (float)c1s;
        auto activate = [&](int s, auto nt0_tag, auto nt1_tag, const int (&frs)[NT / 2], int r, int h) {
            // (r, h: the lane coordinates, re-derived per step by the caller so that the store addresses
            //  below are computed here and do not occupy registers across the GEMM phases)
            constexpr int NT0 = decltype(nt0_tag)::value, NT1 = decltype(nt1_tag)::value;
#pragma unroll
            for (int nt = NT0; nt < NT1; ++nt) {
                const int col = nt * 32 + r;
                const int fr = frs[nt - NT0];
                uint32_t key = 0;
                const double* nz = nullptr;
                if (NOISE == ESN_NOISE_COUNTER)
                    key = noise_key(p.seed, (uint32_t)fr + p.frame_off, (uint32_t)s) + (uint32_t)(wave * MT * 8 + h) * 0x9E3779B9U;
                if (NOISE == ESN_NOISE_TENSOR && fr >= 0)
                    nz = p.noise_u + ((size_t)fr * p.S + s) * n_res;
                // (Tried: a packed-half odd Taylor series for tanh on small pre-activations -- 5 v_pk instructions per
                //  PAIR instead of exp2/add/rcp/fma per value -- forced on: 13.24 -> 13.00 ms and 3e-3 instead of 7e-4
                //  output error; and a float32 series z (a1 + a3 z^2 + a5 z^4) behind a wave-uniform vote: the same
                //  four instructions per value as exp2/add/rcp/fma, and the transcendental unit runs beside the
                //  plain ones, so the vote made it 2.5 % slower.  What phase E pays for is its instruction COUNT.)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = (wave * MT + mt) * 32 + 8 * q + 4 * h;
                        char* dst = zt + (size_t)col * row_bytes + (size_t)row * ES;
                        if constexpr (PK_NOISE) {
                            float t[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
#ifdef ESN_KO_E      // knock-out build (timing only, wrong results): no transcendental work in phase E
                                t[j] = acc[mt][nt][4 * q + j] + t_bias;
#else
                                const float e = __builtin_amdgcn_exp2f(acc[mt][nt][4 * q + j]);
                                t[j] = fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), t_bias);
#endif
                            }
                            const uint32_t sq = noise_mix(key + (uint32_t)(mt * 8 + 2 * q) * 0x9E3779B9U);
                            const h16x2 w01 = __builtin_bit_cast(h16x2, __builtin_amdgcn_perm(0x3C3C3C3Cu, sq, 0x04010400u));
                            const h16x2 w23 = __builtin_bit_cast(h16x2, __builtin_amdgcn_perm(0x3C3C3C3Cu, sq, 0x04030402u));
                            const h16x2 t01 = __builtin_convertvector(f32x2{t[0], t[1]}, h16x2);
                            const h16x2 t23 = __builtin_convertvector(f32x2{t[2], t[3]}, h16x2);
                            const h16x2 x01 = __builtin_elementwise_fma(w01, c1h, t01);
                            const h16x2 x23 = __builtin_elementwise_fma(w23, c1h, t23);
                            *reinterpret_cast<u32x2*>(dst) =
                                u32x2{__builtin_bit_cast(uint32_t, x01), __builtin_bit_cast(uint32_t, x23)};
                        } else {
                            float v[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = TR::act(acc[mt][nt][4 * q + j]);
                            if (NOISE == ESN_NOISE_COUNTER) {
                                const uint32_t sq = noise_mix(key + (uint32_t)(mt * 8 + 2 * q) * 0x9E3779B9U);
                                v[0] = fmaf((float)(sq & 0xffU), n_c1, v[0] + n_c0);
                                v[1] = fmaf((float)((sq >> 8) & 0xffU), n_c1, v[1] + n_c0);
                                v[2] = fmaf((float)((sq >> 16) & 0xffU), n_c1, v[2] + n_c0);
                                v[3] = fmaf((float)(sq >> 24), n_c1, v[3] + n_c0);
                            } else if (NOISE == ESN_NOISE_TENSOR) {
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    if (nz && row + j < n_res) v[j] += noise * ((float)nz[row + j] - 0.5f);
                            }
                            TR::store4(dst, v[0], v[1], v[2], v[3]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        const std::integral_constant<int, 0> nt_lo;
        const std::integral_constant<int, NT / 2> nt_mid;
        const std::integral_constant<int, NT> nt_hi;
        // Inputs: set B stages them for all column tiles (IN_TILES consecutive tiles per wave).  The raw
        // float64 rows of step s+1 go HBM -> LDS staging area by LDS-DMA at the very end of phase
        // E of step s (slot 3s+3) and are converted into the U columns of Zt at the start of
        // P0(s+1) (slot 3s+4), the one slot in which nobody reads U.  Placement matters twice:
        // with a DMA pending the compiler puts s_waitcnt vmcnt(0) in front of the next LDS read
        // (so the DMA must not precede a GEMM loop), and staging through VGPRs instead ends in
        // scratch spills that wait for every HBM load in turn.
        static_assert(NOWN % (NW / 2) == 0, "skewed schedule: set B shares the column tiles evenly");
        constexpr int IN_TILES = NOWN / (NW / 2);          // column tiles staged per wave of set B
        const int in_c0 = IN_TILES * (wave - NW / 2);
        const int lcpf = __builtin_ctz(cpf), lkin = __builtin_ctz(kin_p);     // powers of two (mfma_geometry)
        const size_t in_frame_bytes = (size_t)in_stride * 8;
        // first frame of the tile (frames grow with the slot index): offsets stay below Bt frames' worth
        // of bytes however many frames a group has
        int j0;
        slot_group(p, slot0, j0);
        // (a padding slot at the head of the tile: the next group on this slot axis -- the next group of the same set)
        size_t u_base_frame = j0 < p.F ? (size_t)grp0 * p.F + j0 : ((size_t)grp0 + (p.spw ? p.n_wsets : 1)) * p.F;
        if (u_base_frame >= (size_t)p.n_frames) u_base_frame = (size_t)p.n_frames - 1;   // tile of padding only
        for (int i = tid; i < BT; i += NTHREADS) {
            const int fr = tab_fr[i];
            tab_off[i] = fr >= 0 ? (int)(((size_t)fr - u_base_frame) * in_frame_bytes) : -1;
        }
        __syncthreads();
        const size_t u_left = ((size_t)p.n_frames - u_base_frame) * in_frame_bytes;
        const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(p.U) + u_base_frame * in_frame_bytes), 0,
            (int)(u_left < 0x7fffffffu ? u_left : 0x7fffffffu), 0x00020000);
        // staging layout of the skewed schedule: one 1 KB slot per DMA instruction (tile c, trip i),
        // chunk e = 64 i + lane at byte 16 lane of slot (c, i).  Lanes that must not load (padding
        // frames, lanes past the tile when n_in < 16, rows past T_in) get an out-of-range offset:
        // nothing is fetched, and whatever the DMA does with such a lane stays inside its own slot
        // (with tiles packed back to back those lanes landed in the neighbouring tile's rows).
        // (Tried instead: carrying the rows in registers -- four 16-byte loads in the middle of phase E, converted
        //  at the start of the next P0.  The four DMA launches below cost the issuing wave ~1.05 k cycles of its
        //  phase E (stamps), but the 16 registers are not there: parked in a retired accumulator or not, the
        //  kernel spills 19-36 VGPRs.)
        char* in_slots = reinterpret_cast<char*>(in_raw);
        auto dma_inputs_b = [&](int s) {
            const int row = s + p.in_row_off;
            const bool row_ok = row < p.T_in;
            // straight-line: all frame offsets first (one LDS latency for the lot), then the launches; a trip
            // past the tile (n_in < 8) has every lane out of range and fetches nothing
            int ln = lane;
            asm volatile("" : "+v"(ln));                       // opaque: derived per step, not kept across the GEMM phases
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
            // one lane converts one staged 16-byte chunk (two inputs of one frame): a frame-table read, one 16-byte
            // read each of the scale table and the staging slot, one 4-byte store -- straight-line, no loop
            const bool row_ok = s + p.in_row_off < p.T_in;     // (an out-of-range DMA leaves its slot undefined)
            const int lk2 = lkin - 1;                          // log2 of the pairs per frame (kin_p / 2)
            int ln = lane;
            asm volatile("" : "+v"(ln));                       // opaque: the addresses below are derived per step, not kept
#pragma unroll
            for (int ti = 0; ti < IN_TILES; ++ti) {
                const int c = in_c0 + ti;
#pragma unroll
                for (int k = 0; k < 2; ++k) {                  // 8 kin_p <= 128 pairs per 16-frame tile
                    const int e2 = ln + 64 * k;
                    const int f = (e2 >> lk2) & 15, c2 = e2 & ((kin_p >> 1) - 1), ci = 2 * c2;
                    const bool live = tab_fr[c * 16 + f] >= 0 && ci < n_in;          // (n_in is even: whole pairs)
                    const float4 ss = *reinterpret_cast<const float4*>(tab_in + c * kin_p + ci);
                    const int ch = (f << lcpf) + c2;                                  // < 128 for every lane
                    const double2 raw = *reinterpret_cast<const double2*>(
                        in_slots + (size_t)(c * 2 + (ch >> 6)) * 1024 + (size_t)(ch & 63) * 16);
                    const float v0 = live ? fmaf((float)(row_ok ? raw.x : 0.0), ss.x, ss.y) : 0.f;
                    const float v1 = live ? fmaf((float)(row_ok ? raw.y : 0.0), ss.z, ss.w) : 0.f;
                    if (e2 < 8 * kin_p)
                        TR::store2(zt + (size_t)(c * 16 + f) * row_bytes + (size_t)(g.kin + ci) * ES, v0, v1);
                    __builtin_amdgcn_sched_barrier(0);         // one chunk at a time: the registers are not there for four
                }
            }
        };
#ifdef ESN_STAMPS
        unsigned long long sk_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define ESN_SK_ADD(i, a, b) sk_acc[i] += (b) - (a);
#else
#define ESN_SK_ADD(i, a, b)
#endif
        // One program for both sets.  The readout runs in slot 3s+1 (P1 of set A, P0 of set B) and
        // yU_s in slot 3s+2 (P2 of set A, P1 of set B).
        const bool has_ro = __builtin_amdgcn_readfirstlane(wop[0] != nullptr ? 1 : 0) != 0;   // provably wave-uniform
        // (Tried on the weight stream, all within +-1 % of 13.3 ms: cache-policy bits sc0 / nt / sc1 / sc1+sc0 on
        //  the A loads (nt: +3 %), and starting the workgroups of an XCD up to one timestep apart so that the CUs
        //  do not walk the image in lockstep.  The stream is not an L1 or an L2 hot-spot problem.)
        next_step_A();
        ro_prefetch(false);
        if (lag) __syncthreads();                                      // slot 0: set A alone
        for (int s = 0; s < p.S; ++s) {
            const bool ro = has_ro && s > 0;
            ESN_STAMP(t0)
            zero_acc();
            // ---- P0.  Set B (the younger wave of each SIMD) would lose every issue arbitration of
            // slot 3s+1 to set A's P1 and finish its half alone, latency-bound: even them out
            if (lag) __builtin_amdgcn_s_setprio(1);
            if (lag && s > 0) commit_inputs_b(s);
            gemm_half(0, ro && lag);
            if (lag) { if (ro) finish_readout(s - 1 - p.transient, true); }
            if (lag) __builtin_amdgcn_s_setprio(0);
            ro_prefetch(ro && !lag);                                   // set A: for P1, in flight over the barrier
            ESN_STAMP(t1)
            __syncthreads();
            ESN_STAMP(t2)
            gemm_half(nkgH, ro && !lag);                               // ---- P1
            if (!lag) { if (ro) finish_readout(s - 1 - p.transient, true); }
            ESN_STAMP(t3)
            __syncthreads();
            ESN_STAMP(t4)
            // ---- P2.  The VALU-bound wave gets the issue slots: its SIMD partner is in an MFMA
            // phase with slack (and, for set B, would otherwise win every arbitration by age)
            __builtin_amdgcn_s_setprio(2);
            const u32x4 ra_u = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                wo_rsrc, has_ro ? lane16 : OOB, has_ro ? nk64S * 1024 : 0, 0));
            int r_e = r, h_e = h;
            asm volatile("" : "+v"(r_e), "+v"(h_e));               // opaque copies: see activate
            int frs[NT / 2];
#pragma unroll
            for (int i = 0; i < NT / 2; ++i) frs[i] = tab_fr[i * 32 + r_e];
            uf_groups();
            ESN_STAMP(u1)
            activate(s, nt_lo, nt_mid, frs, r_e, h_e);
            ESN_STAMP(u2)
            {   // yU_s = Wout[:, inputs] U_s (the feedback columns of that group carry zero weights);
                // U_s stays in Zt until set B commits U_{s+1} at the start of slot 3s+4
                const u32x4 rb_u = *reinterpret_cast<const u32x4*>(zrow0 + nk64S * 64);
                TR::mma16(yacc[0], ra_u, rb_u);
            }
            next_step_A();
            ro_prefetch(lag && has_ro && s + 1 < p.S);                 // set B: for P0(s+1)
#pragma unroll
            for (int i = 0; i < NT / 2; ++i) frs[i] = tab_fr[(NT / 2 + i) * 32 + r_e];
            ESN_STAMP(u3)
            // last LDS read of the phase is behind us: the input DMA of step s+1 flies during the rest of E
            if (lag && s + 1 < p.S) dma_inputs_b(s + 1);
            ESN_STAMP(u4)
            activate(s, nt_mid, nt_hi, frs, r_e, h_e);
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
        if (!lag) __syncthreads();                                     // slot 3S: set B finishes X_B(S)
#undef ESN_SK_ADD
        if (any_ro) {                                                  // Y_S = yU_{S-1} + Wout_x X_S
            readout_groups(0, nk64S);
            finish_readout(p.S - 1 - p.transient, false);
        }
#ifdef ESN_STAMPS
        if (p.stamps && blockIdx.x == 0 && lane == 0) {
            for (int i = 0; i < 6; ++i) p.stamps[wave * 8 + i] = sk_acc[i];
            for (int i = 0; i < 6; ++i) p.stamps[(8 + wave) * 8 + i] = sk_acc[6 + i];    // inside P2
            p.stamps[wave * 8 + 6] = (unsigned long long)__builtin_amdgcn_s_getreg(63492);   // HW_ID
        }
#endif
        return;
    }

#ifdef ESN_STAMPS
    unsigned long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    // Harvest: few sequences (one pilot per trained ESN), so a launch is a handful of workgroups
    // whose critical path is the weight stream of every step.  Same recipe as the skewed
    // schedule: HD k-groups in flight per wave in fixed buffers, refilled by buffer loads right
    // after their last MFMA, nothing conditional.  The k sequence is padded to a multiple of HD
    // positions per step (positions >= nkg: out-of-range offset, zero fragments, no traffic) so
    // the buffers keep their phase across steps and the look-ahead runs through phase E.
    // buffers: 2 at 16 waves x 64 accumulators (128-VGPR budget), else 4 (8 measured no faster)
    constexpr int HD = (NW * MT * NT >= 64) ? 2 : 4;
    u32x4 hbuf[HD][MT];
    const int h_npos = (nkg + HD - 1) / HD * HD;
    const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.packed_w) + (size_t)wset * p.wset_stride), 0,
        (int)p.wset_stride, 0x00020000);
    const int h_row0 = wave * MT * nkg;
    auto hload = [&](u32x4 (&a)[MT], int pos) {
        const int kg = pos >= h_npos ? pos - h_npos : pos;
        const bool live = kg < nkg;
        const int voff = live ? lane * 16 : 0x7ffffff0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            a[mt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                h_rsrc, voff, live ? (h_row0 + mt * nkg + kg) * 1024 : 0, 0));
    };
    if constexpr (HARVEST) {
#pragma unroll
        for (int j = 0; j < HD; ++j) hload(hbuf[j], j);
    }
    // E row `erow`, state columns, straight from the LDS image: wave w copies frames w, w + NW, ...;
    // (Tried: issuing these stores inside the GEMM loop -- they share vmcnt with the weight loads
    // and doubled the loop time.)
    auto copy_row = [&](int erow) {
        for (int f = wave; f < BT; f += NW) {
            const int fr = tab_fr[f];
            if (fr < 0) continue;
            const size_t e0 = ((size_t)fr * (p.S + 1) + erow) * ncols;
            const char* zr = zt + (size_t)f * row_bytes;
            if (p.E32 && (ncols & 3) == 0) {        // float32: 16-byte stores of four columns per lane
                // (the LDS row holds Kp >= n_res elements: reading 4 columns at c < n_res stays inside it)
                for (int c0 = 0; c0 < n_res; c0 += 512) {
                    float v[2][4];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int c = c0 + 256 * u + 4 * lane;
                        if (c < n_res) TR::load4(zr + (size_t)c * ES, v[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int c = c0 + 256 * u + 4 * lane;
                        if (c + 3 < n_res) *reinterpret_cast<f32x4*>(p.E32 + e0 + c) = f32x4{v[u][0], v[u][1], v[u][2], v[u][3]};
                        else
                            for (int q = 0; q < 4; ++q) if (c + q < n_res) p.E32[e0 + c + q] = v[u][q];
                    }
                }
            } else if (!p.E32 && (ncols & 1) == 0) {   // float64: 16-byte stores, two columns per lane
                double* er = p.E + e0;
                for (int c0 = 0; c0 < n_res; c0 += 512) {
                    float v[4][2];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = c0 + 128 * u + 2 * lane;
                        v[u][0] = c < n_res ? TR::load1(zr + (size_t)c * ES) : 0.f;
                        v[u][1] = c + 1 < n_res ? TR::load1(zr + (size_t)(c + 1) * ES) : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = c0 + 128 * u + 2 * lane;
                        if (c + 1 < n_res) *reinterpret_cast<double2*>(er + c) = double2{(double)v[u][0], (double)v[u][1]};
                        else if (c < n_res) er[c] = (double)v[u][0];
                    }
                }
            } else {
                for (int c = lane; c < n_res; c += 64) store_E(e0 + c, (double)TR::load1(zr + (size_t)c * ES));
            }
        }
    };
    for (int s = 0; s < p.S; ++s) {
        ESN_STAMP(t0)
        const bool have_next = s + 1 < p.S;
        if (in_dma && have_next) dma_inputs(s + 1);          // lands in in_raw while the GEMM runs
        // harvest: thread t fetches input element t and teacher element t of step s+1 now, into two
        // registers, and converts them in phase E -- the HBM latency hides under the GEMM (staging
        // by the two column-owner waves alone cost 9k of the 26k cycles of a harvest step)
        double pre_in = 0.0, pre_t = 0.0;
        int pre_fr_in = -1, pre_fr_t = -1;
        const bool pre_ok = HARVEST && BT * kin_p <= NTHREADS && BT * kfb_p <= NTHREADS;
        if (HARVEST && pre_ok && have_next) {
            if (tid < BT * kin_p) {
                const int f = tid / kin_p, ci = tid - f * kin_p;
                const int fr = tab_fr[f], row = s + 1 + p.in_row_off;
                if (fr >= 0 && ci < n_in) {
                    pre_fr_in = fr;
                    if (row < p.T_in) pre_in = p.U[(size_t)fr * in_stride + (size_t)row * n_in + ci];
                }
            }
            if (tid < BT * kfb_p) {
                const int f = tid / kfb_p, co = tid - f * kfb_p;
                const int fr = tab_fr[f];
                if (fr >= 0 && co < n_out) {
                    pre_fr_t = fr;
                    pre_t = p.D[((size_t)fr * (p.S + 1) + (s + 1)) * n_out + co];
                }
            }
        }
        // ================= phase G1: state k-groups (+ readout of Y_s) ===========
        f32x16 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
#ifdef ESN_STAMPS
        unsigned long long t1 = 0, t2 = 0;
#endif
        if constexpr (HARVEST) {
            // B fragments double-buffered: with two MFMAs per k-group the LDS latency of a
            // single-buffered B operand was the critical path (285 cycles per k-group)
            constexpr int HB = NW < 16 ? 2 : 1;              // (one buffer at 16 waves: 128-VGPR budget)
            u32x4 hb[HB][NT];
            auto hloadB = [&](u32x4 (&b)[NT], int kg) {
                kg = kg < nkg ? kg : nkg - 1;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    b[nt] = *reinterpret_cast<const u32x4*>(bbase + (size_t)nt * 32 * row_bytes + kg * 32);
            };
            hloadB(hb[0], 0);
            for (int i = 0; i < h_npos; i += HD) {
#pragma unroll
                for (int j = 0; j < HD; ++j) {
                    if constexpr (HB == 2) hloadB(hb[(j + 1) & 1], i + j + 1);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) TR::mma32(acc[mt][nt], hbuf[j][mt], hb[j & (HB - 1)][nt]);
                    __builtin_amdgcn_sched_barrier(0);
                    hload(hbuf[j], i + j + HD);
                    if constexpr (HB == 1) hloadB(hb[0], i + j + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            ESN_STAMP_SET(t1)
            ESN_STAMP_SET(t2)
        } else {
        // software pipeline: A fragments two k-groups ahead (L2 latency), B fragments one ahead (LDS)
        u32x4 aA[MT], aB[MT], aC[MT], bA[NT], bB[NT];
        // position i of the k sequence -> k-group: the state groups are walked from a per-tile
        // rotation so the CUs of an XCD do not all hit the same L2 channel at the same time
        auto kg_of = [&](int i) -> int {
            if (i >= nkgS) return i < nkg ? i : nkg - 1;
            int kr = (i >> 1) + krot;
            kr = kr >= nk64S ? kr - nk64S : kr;
            return 2 * kr + (i & 1);
        };
        auto loadA = [&](u32x4 (&a)[MT], int pos) {
            const int kg = kg_of(pos);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#ifdef ESN_NT_WEIGHTS   // measured: non-temporal weight loads are 1.5x SLOWER in G1 (every CU re-reads W from L2)
                a[mt] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + ((size_t)mt * nkg + kg) * 1024));
#else
                a[mt] = *reinterpret_cast<const u32x4*>(wp + ((size_t)mt * nkg + kg) * 1024);
#endif
        };
        auto loadB = [&](u32x4 (&b)[NT], int pos) {
            const int kg = kg_of(pos);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                b[nt] = *reinterpret_cast<const u32x4*>(bbase + (size_t)nt * 32 * row_bytes + kg * 32);
        };
        auto mma_all = [&](const u32x4 (&b)[NT]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) TR::mma32(acc[mt][nt], aA[mt], b[nt]);
        };
        auto rotateA = [&]() {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { aA[mt] = aB[mt]; aB[mt] = aC[mt]; }
        };
        const bool ro_now = any_ro && s > 0;
        loadA(aA, 0); loadA(aB, 1); loadB(bA, 0);
        // Two k-groups per trip, B fragments ping-pong bA / bB; every trip is branch-free.  (The
        // last trip prefetches the first [U|F] group early: it is re-read after the barrier.)
        if (ro_now && ro_simple) {
            // readout operands: W_out fragment one 64-byte group ahead, state fragment a k-group early
            u32x4 ra_cur[OC], ra_nxt[OC], rb[OC];
#pragma unroll
            for (int i = 0; i < OC; ++i)
                ra_cur[i] = wop[i] ? *reinterpret_cast<const u32x4*>(wop[i] + (size_t)krot * 1024) : u32x4{0, 0, 0, 0};
            for (int kk = 0; kk < nk64S; ++kk) {
                loadA(aC, 2 * kk + 2); loadB(bB, 2 * kk + 1);
                const int kc = kg_of(2 * kk) >> 1;                               // this trip's 64-byte group
                const int kn = kk + 1 < nk64S ? (kg_of(2 * kk + 2) >> 1) : kc;   // next trip's
#pragma unroll
                for (int i = 0; i < OC; ++i) {
                    rb[i] = *reinterpret_cast<const u32x4*>(zrow0 + i * zrow_step + kc * 64);
                    ra_nxt[i] = wop[i] ? *reinterpret_cast<const u32x4*>(wop[i] + (size_t)kn * 1024) : u32x4{0, 0, 0, 0};
                }
                mma_all(bA); rotateA();
                loadA(aC, 2 * kk + 3); loadB(bA, 2 * kk + 2);
                mma_all(bB);
#pragma unroll
                for (int i = 0; i < OC; ++i) { TR::mma16(yacc[i], ra_cur[i], rb[i]); ra_cur[i] = ra_nxt[i]; }
                rotateA();
            }
        } else {
            for (int kk = 0; kk < nk64S; ++kk) {
                loadA(aC, 2 * kk + 2); loadB(bB, 2 * kk + 1);
                mma_all(bA); rotateA();
                loadA(aC, 2 * kk + 3); loadB(bA, 2 * kk + 2);
                mma_all(bB); rotateA();
                if (ro_now) { const int kc = kg_of(2 * kk) >> 1; readout_groups(kc, kc + 1); }
            }
        }
        if (ro_now) finish_readout(s - 1 - p.transient, true);
        ESN_STAMP_SET(t1)
        if (!HARVEST) __syncthreads();            // F_s visible to every wave
        ESN_STAMP_SET(t2)
        // ================= phase G2: input + feedback k-groups =====================
        loadB(bA, nkgS);
        for (int kg = nkgS; kg < nkg; ++kg) {
            loadA(aC, kg + 2);
            if (kg + 1 < nkg) loadB(bB, kg + 1);
            mma_all(bA); rotateA();
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bA[nt] = bB[nt];
        }
        if (any_ro) readout_groups(nk64S, nk64);  // yU_s (feedback columns carry zero weights)
        }   // !HARVEST
        ESN_STAMP(t3)
        __syncthreads();                          // every wave has finished reading Z_s
        ESN_STAMP(t4)

        // ================= phase E: X_{s+1} = tanh(P) + noise =====================
        // next step's inputs first: the HBM latency hides under the activation arithmetic
        if (HARVEST && pre_ok) {
            if (have_next) {
                if (tid < BT * kin_p) {                       // same arithmetic as load_in / stage_teacher
                    const int f = tid / kin_p, ci = tid - f * kin_p;
                    float v = 0.f;
                    if (pre_fr_in >= 0) {
                        int pg;
                        slot_frame(p, slot0 + f, pg);
                        const double sc = p.in_scale ? p.in_scale[(size_t)pg * n_in + ci] : 1.0;
                        const double sh = p.in_shift ? p.in_shift[(size_t)pg * n_in + ci] : 0.0;
                        const double sv = pre_in * sc + sh;
                        store_E(((size_t)pre_fr_in * (p.S + 1) + (s + 1 + p.in_row_off)) * ncols + n_res + ci, sv);
                        v = (float)sv;
                    }
                    TR::store1(zt + (size_t)f * row_bytes + (size_t)(g.kin + ci) * ES, v);
                }
                if (tid < BT * kfb_p) {
                    const int f = tid / kfb_p, co = tid - f * kfb_p;
                    float v = 0.f;
                    if (pre_fr_t >= 0) {
                        int pg;
                        slot_frame(p, slot0 + f, pg);
                        const double sc = p.t_scale ? p.t_scale[(size_t)pg * n_out + co] : 1.0;
                        const double sh = p.t_shift ? p.t_shift[(size_t)pg * n_out + co] : 0.0;
                        v = (float)(pre_t * sc + sh);
                    }
                    TR::store1(zt + (size_t)f * row_bytes + (size_t)(g.kfb + co) * ES, v);
                }
            }
        } else if (have_next) {
            if (in_dma) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's own DMA (issued a GEMM ago)
                commit_inputs(s + 1);
            } else {
                stage_inputs(s + 1);
            }
            if (HARVEST) stage_teacher(s + 1);
        }
        bool small = false;
        if constexpr (TR::HAS_SMALL) {
            float amax = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(acc[mt][nt][i]));
            small = __all(amax < TANH32_SERIES_MAX) != 0;          // NaN compares false -> full routine
        }
        auto activate = [&](auto small_tag) {
            constexpr bool SMALL = decltype(small_tag)::value;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = nt * 32 + r;
                const int fr = tab_fr[col];
                uint32_t key = 0;
                const double* nz = nullptr;
                if (NOISE == ESN_NOISE_COUNTER)
                    key = noise_key(p.seed, (uint32_t)fr + p.frame_off, (uint32_t)s) + (uint32_t)(wave * MT * 8 + h) * 0x9E3779B9U;
                if (NOISE == ESN_NOISE_TENSOR && fr >= 0)
                    nz = p.noise_u + ((size_t)fr * p.S + s) * n_res;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = (wave * MT + mt) * 32 + 8 * q + 4 * h;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = SMALL ? TR::act_small(acc[mt][nt][4 * q + j]) : TR::act(acc[mt][nt][4 * q + j]);
                        // (state rows >= n_res are padding: their columns of Wext / W_out are zero, so the
                        //  noise they pick up is never read -- no masking needed)
                        if (NOISE == ESN_NOISE_COUNTER) {
                            // row/4 = (wave*MT*8 + h) + (mt*8 + 2q): the second term folds at compile time
                            const uint32_t sq = noise_mix(key + (uint32_t)(mt * 8 + 2 * q) * 0x9E3779B9U);
                            // + noise*((byte+0.5)/256 - 0.5) = byte*n_c1 + n_c0
                            v[0] = fmaf((float)(sq & 0xffU), n_c1, v[0] + n_c0);
                            v[1] = fmaf((float)((sq >> 8) & 0xffU), n_c1, v[1] + n_c0);
                            v[2] = fmaf((float)((sq >> 16) & 0xffU), n_c1, v[2] + n_c0);
                            v[3] = fmaf((float)(sq >> 24), n_c1, v[3] + n_c0);
                        } else if (NOISE == ESN_NOISE_TENSOR) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (nz && row + j < n_res) v[j] += noise * ((float)nz[row + j] - 0.5f);
                        }
                        TR::store4(zt + (size_t)col * row_bytes + (size_t)row * ES, v[0], v[1], v[2], v[3]);
                    }
                    // keep the scheduler from interleaving all 16*MT*NT activations (register pressure)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if constexpr (TR::HAS_SMALL) {
            if (small) activate(std::true_type{}); else activate(std::false_type{});
        } else {
            activate(std::false_type{});
        }
        ESN_STAMP(t5)
        __syncthreads();                          // X_{s+1}, U_{s+1} (, F_{s+1}) complete
        ESN_STAMP(t6)
        if (HARVEST) copy_row(s + 1);
        ESN_STAMP(t7)
#ifdef ESN_STAMPS
        st_acc[0] += t1 - t0; st_acc[1] += t2 - t1; st_acc[2] += t3 - t2;
        st_acc[3] += t4 - t3; st_acc[4] += t5 - t4; st_acc[5] += t6 - t5; st_acc[6] += t7 - t6;
#endif
    }
    // final readout Y_S = yU_{S-1} + Wout_x X_S
    if (any_ro) {
        readout_groups(0, nk64S);
        finish_readout(p.S - 1 - p.transient, false);
    }
#ifdef ESN_STAMPS
    if (p.stamps && blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 7; ++i) p.stamps[wave * 8 + i] = st_acc[i];
#endif
}

// ---- host side: geometry choice and launch -------------------------------------

template <typename TR, int NW, int MT, int NT, bool HARVEST, int NOISE, bool SKEW = false>
static int launch_k(const RecurParams& p, hipStream_t stream) {
    const int kin_p = p.g.kfb - p.g.kin, nown = p.g.Bt / 16;
    size_t lds = (size_t)p.g.Bt * p.g.Ks * TR::ES + 4 * (size_t)p.g.Bt + 8 * (size_t)nown * (kin_p + 16)
                 + (SKEW ? (size_t)nown * 2048 : 8 * (size_t)p.g.Bt * p.n_in) + 4 * (size_t)p.g.Bt;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(recur_mfma_kernel<TR, NW, MT, NT, HARVEST, NOISE, SKEW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((recur_mfma_kernel<TR, NW, MT, NT, HARVEST, NOISE, SKEW>), dim3(p.n_tiles), dim3(NW * 64), lds,
                       stream, p);
    return (int)hipGetLastError();
}

template <typename TR, int NW, int MT, int NT, bool HARVEST, bool SKEW = false>
static int launch_n(const RecurParams& p, hipStream_t stream) {
    switch (p.noise_mode) {
        case ESN_NOISE_NONE: return launch_k<TR, NW, MT, NT, HARVEST, ESN_NOISE_NONE, SKEW>(p, stream);
        case ESN_NOISE_TENSOR: return launch_k<TR, NW, MT, NT, HARVEST, ESN_NOISE_TENSOR, SKEW>(p, stream);
        default: return launch_k<TR, NW, MT, NT, HARVEST, ESN_NOISE_COUNTER, SKEW>(p, stream);
    }
}

template <typename TR, int NW, int MT, int NT>
static int launch_one(const RecurParams& p, hipStream_t stream) {
    // skewed schedule: instantiated for the 8-wave fp16 / bf16 predict tilings (mfma_geometry sets g.skew)
    if constexpr (NW == 8 && TR::ES == 2 && (NT == 2 || NT == 4)) {
        if (!p.harvest && p.g.skew) return launch_n<TR, NW, MT, NT, false, true>(p, stream);
    }
    return p.harvest ? launch_n<TR, NW, MT, NT, true>(p, stream) : launch_n<TR, NW, MT, NT, false>(p, stream);
}

}  // namespace esn
