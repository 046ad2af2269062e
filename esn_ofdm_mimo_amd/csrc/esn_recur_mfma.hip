// MFMA recurrence kernels (float32 / fp16 / bf16 operands, float32 accumulate).
//
// One workgroup owns a tile of Bt = 32*NT sequences for all S timesteps; nothing
// leaves the CU between steps.  Per step:
//
//   phase G  P[Mp x Bt] = Wext[Mp x Kp] * Z[Kp x Bt]     Z = [X_s ; U_s ; F_s]
//            A operand  = Wext, pre-packed in MFMA fragment order, streamed from
//                         L2 (shared by every workgroup of the weight set)
//            B operand  = Z, resident in LDS as Zt[frame][k] (k contiguous)
//            wave w owns rows [32*MT*w, 32*MT*(w+1)) and all Bt columns
//   barrier
//   phase E  X_{s+1} = tanh(P) + noise*(u-0.5) written back into Zt (state rows)
//   barrier
//   phase R  wave c < Bt/16 (the "column owner" of frames 16c..16c+15):
//            Y = Wout * [X_{s+1}; U_s] by 16x16 MFMA, store unscaled Y (s >= transient),
//            F_{s+1} = Y (or the teacher when harvesting), U_{s+1} from HBM
//   barrier
//
// K is consumed in 32-byte groups of each LDS row: lane (r = lane&31, h = lane>>5)
// takes bytes [32*kg + 16*h, +16) of row r of both operands, so one ds_read_b128 /
// one 16-byte global load feeds 4 x v_mfma_f32_32x32x2_f32 (float32) or
// 1 x v_mfma_f32_32x32x16_{f16,bf16}.  (The k order inside a group is a fixed
// permutation applied to both operands, which a dot product does not see.)
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
    static __device__ __forceinline__ float load1(const char* src) { return *reinterpret_cast<const float*>(src); }
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
    static __device__ __forceinline__ float load1(const char* src) { return (float)*reinterpret_cast<const _Float16*>(src); }
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
    static __device__ __forceinline__ float load1(const char* src) { return (float)*reinterpret_cast<const __bf16*>(src); }
};

template <typename TR, int NW, int MT, int NT>
__global__ __launch_bounds__(NW * 64) void recur_mfma_kernel(RecurParams p) {
    extern __shared__ __attribute__((aligned(16))) char zt[];   // Zt[Bt][Ks] elements
    constexpr int ES = TR::ES;
    constexpr int BT = 32 * NT;
    constexpr int NTHREADS = NW * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const Geometry& g = p.g;
    const int n_res = p.n_res, n_in = p.n_in, n_out = p.n_out;
    const int row_bytes = g.Ks * ES;
    const int nkg = g.Kp * ES / 32;          // 32-byte k-groups per row (GEMM)
    const int nkg64 = g.Kp * ES / 64;        // 64-byte k-groups per row (readout)

    const int tile = xcd_remap(blockIdx.x, p.n_tiles);
    const int slot0 = tile * BT;
    const int grp0 = slot0 / p.Fpad;
    if (grp0 >= p.n_groups) return;
    const int wset = grp0 % p.n_wsets;
    const char* wp = reinterpret_cast<const char*>(p.packed_w) + (size_t)wset * p.wset_stride
                     + ((size_t)(wave * MT) * nkg * 64 + lane) * 16;
    const int n_ot = (n_out + 15) / 16;
    const int out_rows = p.S - p.transient;
    const int ncols = n_res + n_in;
    const float in_gain = 1.0f, fb_gain = 1.0f;   // reserved: power-of-two operand gains

    // frame / group of the column this lane owns in each 32-wide column tile (GEMM layout)
    int col_fr[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { int gtmp; col_fr[nt] = slot_frame(p, slot0 + nt * 32 + r, gtmp); }
    // column-owner view: wave c16 < BT/16 owns frames 16*c16 .. 16*c16+15 (one group: Fpad % 16 == 0)
    int ro_grp = 0;
    const int ro_fr = slot_frame(p, slot0 + (wave < BT / 16 ? wave : 0) * 16 + (lane & 15), ro_grp);
    const char* wop = nullptr;
    float wo_inv = 1.f;
    if (!p.harvest && wave < BT / 16) {
        const int cg = (slot0 + wave * 16) / p.Fpad;
        if (cg < p.n_groups) {
            const char* base = reinterpret_cast<const char*>(p.packed_wout) + (size_t)cg * p.wout_stride;
            wop = base + (size_t)lane * 16;
            // readout image trailer: {1/gain, gain} of this group's power-of-two W_out scaling
            wo_inv = *reinterpret_cast<const float*>(base + (size_t)TR::PARTS * n_ot * nkg64 * 1024);
        }
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
            } else if (k >= g.kfb && k < g.kfb + n_out && !p.harvest) {
                if (p.y0) v = (float)p.y0[(size_t)pg * n_out + (k - g.kfb)] * fb_gain;
            }
        }
        TR::store1(zt + (size_t)i * ES, v);
    }
    if (p.harvest) {
        // E row 0 = [0, scale(u[0])]  (pyESN.py:179,189)
        for (int i = tid; i < BT * ncols; i += NTHREADS) {
            int f = i / ncols, c = i % ncols;
            int pg;
            const int fr = slot_frame(p, slot0 + f, pg);
            if (fr < 0) continue;
            double v = 0.0;
            if (c >= n_res) {
                int ci = c - n_res;
                double raw = p.U[((size_t)fr * p.T_in) * n_in + ci];
                double sc = p.in_scale ? p.in_scale[(size_t)pg * n_in + ci] : 1.0;
                double sh = p.in_shift ? p.in_shift[(size_t)pg * n_in + ci] : 0.0;
                v = raw * sc + sh;
            }
            p.E[((size_t)fr * (p.S + 1)) * ncols + c] = v;
        }
    }
    __syncthreads();

    // column-owner staging of inputs (and the teacher when harvesting) for step s
    const int kin_p = g.kfb - g.kin;
    auto stage_io = [&](int s, int c16) {
        const int row = s + p.in_row_off;
        for (int i = lane; i < 16 * kin_p; i += 64) {
            int f = c16 * 16 + i / kin_p, c = i % kin_p;
            float v = 0.f;
            int pg;
            const int fr = slot_frame(p, slot0 + f, pg);
            if (fr >= 0 && c < n_in) {
                double raw = (row < p.T_in) ? p.U[((size_t)fr * p.T_in + row) * n_in + c] : 0.0;
                double sc = p.in_scale ? p.in_scale[(size_t)pg * n_in + c] : 1.0;
                double sh = p.in_shift ? p.in_shift[(size_t)pg * n_in + c] : 0.0;
                v = (float)(raw * sc + sh) * in_gain;
            }
            TR::store1(zt + (size_t)f * row_bytes + (size_t)(g.kin + c) * ES, v);
        }
        if (p.harvest) {
            const int kfb_p = round_up(n_out, 4);
            for (int i = lane; i < 16 * kfb_p; i += 64) {
                int f = c16 * 16 + i / kfb_p, c = i % kfb_p;
                float v = 0.f;
                int pg;
                const int fr = slot_frame(p, slot0 + f, pg);
                if (fr >= 0 && c < n_out) {
                    double raw = p.D[((size_t)fr * (p.S + 1) + s) * n_out + c];
                    double sc = p.t_scale ? p.t_scale[(size_t)pg * n_out + c] : 1.0;
                    double sh = p.t_shift ? p.t_shift[(size_t)pg * n_out + c] : 0.0;
                    v = (float)(raw * sc + sh) * fb_gain;
                }
                TR::store1(zt + (size_t)f * row_bytes + (size_t)(g.kfb + c) * ES, v);
            }
        }
    };
    if (wave < BT / 16) stage_io(0, wave);
    __syncthreads();

    const float noise = (float)p.noise;

    for (int s = 0; s < p.S; ++s) {
        // ================= phase G: P = Wext * Z ================================
        f32x16 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

        const char* bbase = zt + (size_t)r * row_bytes + 16 * h;
        u32x4 a_cur[MT], a_nxt[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            a_cur[mt] = *reinterpret_cast<const u32x4*>(wp + (size_t)mt * nkg * 1024);
        for (int kg = 0; kg < nkg; ++kg) {
            const int kn = (kg + 1 < nkg) ? kg + 1 : kg;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                a_nxt[mt] = *reinterpret_cast<const u32x4*>(wp + ((size_t)mt * nkg + kn) * 1024);
            u32x4 b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                b[nt] = *reinterpret_cast<const u32x4*>(bbase + (size_t)nt * 32 * row_bytes + kg * 32);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) TR::mma32(acc[mt][nt], a_cur[mt], b[nt]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a_cur[mt] = a_nxt[mt];
        }
        __syncthreads();   // every wave has finished reading X_s

        // ================= phase E: X_{s+1} = tanh(P) + noise ===================
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = nt * 32 + r;
            const int fr = col_fr[nt];
            uint32_t key = 0;
            const double* nz = nullptr;
            if (p.noise_mode == ESN_NOISE_COUNTER) key = noise_key(p.seed, (uint32_t)fr, (uint32_t)s);
            if (p.noise_mode == ESN_NOISE_TENSOR && fr >= 0)
                nz = p.noise_u + ((size_t)fr * p.S + s) * n_res;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = (wave * MT + mt) * 32 + 8 * q + 4 * h;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x = tanh_f32(acc[mt][nt][4 * q + j]);
                        if (p.noise_mode == ESN_NOISE_COUNTER) {
                            if (row + j < n_res) x += noise * (noise_uniform(key, row + j) - 0.5f);
                        } else if (p.noise_mode == ESN_NOISE_TENSOR) {
                            if (nz && row + j < n_res) x += noise * ((float)nz[row + j] - 0.5f);
                        }
                        v[j] = x;
                    }
                    TR::store4(zt + (size_t)col * row_bytes + (size_t)row * ES, v[0], v[1], v[2], v[3]);
                }
            }
        }
        __syncthreads();   // X_{s+1} complete

        if (p.harvest) {
            // E row s+1 = [x_{s+1}, u_scaled[s+1]] straight from the LDS image
            for (int i = tid; i < BT * ncols; i += NTHREADS) {
                int f = i / ncols, c = i % ncols;
                int pg;
                const int fr = slot_frame(p, slot0 + f, pg);
                if (fr < 0) continue;
                int k = (c < n_res) ? c : g.kin + (c - n_res);
                float v = TR::load1(zt + (size_t)f * row_bytes + (size_t)k * ES);
                p.E[((size_t)fr * (p.S + 1) + (s + 1)) * ncols + c] = (double)v;
            }
            __syncthreads();
            if (wave < BT / 16 && s + 1 < p.S) stage_io(s + 1, wave);
        } else if (wave < BT / 16 && wop) {
            // ================= phase R: readout + feedback + next inputs ==========
            const int c16 = wave;
            const int q = lane >> 4, fc = lane & 15;
            const int f = c16 * 16 + fc;
            const char* zrow = zt + (size_t)f * row_bytes + 16 * q;
            for (int ot = 0; ot < n_ot; ++ot) {
                f32x4 y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int part = 0; part < TR::PARTS; ++part) {
                    const char* wa = wop + ((size_t)(part * n_ot + ot) * nkg64) * 1024;
                    for (int kg = 0; kg < nkg64; ++kg) {
                        u32x4 a = *reinterpret_cast<const u32x4*>(wa + (size_t)kg * 1024);
                        u32x4 b = *reinterpret_cast<const u32x4*>(zrow + kg * 64);
                        TR::mma16(y, a, b);
                    }
                }
                y *= wo_inv;
                // lane holds outputs o = 16*ot + 4*q + j of frame f
                const int o0 = ot * 16 + 4 * q;
                if (o0 < round_up(n_out, 4)) {
                    TR::store4(zt + (size_t)f * row_bytes + (size_t)(g.kfb + o0) * ES,
                               y[0] * fb_gain, y[1] * fb_gain, y[2] * fb_gain, y[3] * fb_gain);
                    if (s >= p.transient && ro_fr >= 0) {
                        const int fr = ro_fr, pg = ro_grp;
                        double* yo = p.Y + ((size_t)fr * out_rows + (s - p.transient)) * n_out;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            int o = o0 + j;
                            if (o < n_out) {
                                double sc = p.t_scale ? p.t_scale[(size_t)pg * n_out + o] : 1.0;
                                double sh = p.t_shift ? p.t_shift[(size_t)pg * n_out + o] : 0.0;
                                yo[o] = ((double)y[j] - sh) / sc;
                            }
                        }
                    }
                }
            }
            if (s + 1 < p.S) stage_io(s + 1, c16);
        }
        __syncthreads();
    }
}

// ---- host side: geometry choice and launch -------------------------------------

template <typename TR, int NW, int MT, int NT>
static int launch_one(const RecurParams& p, hipStream_t stream) {
    size_t lds = (size_t)p.g.Bt * p.g.Ks * TR::ES;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(recur_mfma_kernel<TR, NW, MT, NT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((recur_mfma_kernel<TR, NW, MT, NT>), dim3(p.n_tiles), dim3(NW * 64), lds, stream, p);
    return (int)hipGetLastError();
}

// Tiling table: (NW, MT, NT) per precision and reservoir size.  Constraints:
// Mp = 32*MT*NW >= n_res; LDS = 32*NT*Ks*ES <= 160 KiB; accumulators 16*MT*NT
// VGPRs/lane within the 512/(waves per SIMD) budget; NT*2 <= NW (column owners).
bool mfma_geometry(int precision, int n_res, int n_in, int n_out, Geometry* g) {
    const int es = (precision == ESN_F32) ? 4 : 2;
    int NW, MT, NT;
    if (precision == ESN_F32) {
        if (n_res <= 128)       { NW = 4;  MT = 1; NT = 2; }
        else if (n_res <= 256)  { NW = 4;  MT = 2; NT = 2; }
        else if (n_res <= 512)  { NW = 8;  MT = 2; NT = 2; }
        else if (n_res <= 1024) { NW = 8;  MT = 4; NT = 1; }
        else return false;
    } else {
        if (n_res <= 128)       { NW = 4;  MT = 1; NT = 2; }
        else if (n_res <= 256)  { NW = 8;  MT = 1; NT = 4; }
        else if (n_res <= 512)  { NW = 8;  MT = 2; NT = 4; }
        else if (n_res <= 1024) { NW = 8;  MT = 4; NT = 2; }
        else if (n_res <= 2048) { NW = 16; MT = 4; NT = 1; }
        else return false;
    }
    if (n_out > 32) return false;
    g->NW = NW; g->MT = MT; g->NT = NT;
    g->Mp = 32 * MT * NW;
    g->kin = g->Mp;
    g->kfb = g->kin + round_up(n_in, 4);
    g->Kp = round_up(g->kfb + round_up(n_out, 4), 32);
    // row stride: an odd number of 16-byte slots -> conflict-free b128 column reads
    int slots = g->Kp * es / 16;
    g->Ks = g->Kp + ((slots % 2 == 0) ? 16 / es : 0);
    g->Bt = 32 * NT;
    return (size_t)g->Bt * g->Ks * es <= 160 * 1024;
}

int launch_recur_mfma(int precision, const RecurParams& p, hipStream_t stream) {
    const Geometry& g = p.g;
#define ESN_CASE(TR, NWv, MTv, NTv) \
    if (g.NW == NWv && g.MT == MTv && g.NT == NTv) return launch_one<TR, NWv, MTv, NTv>(p, stream);
    if (precision == ESN_F32) {
        ESN_CASE(TraitsF32, 4, 1, 2)
        ESN_CASE(TraitsF32, 4, 2, 2)
        ESN_CASE(TraitsF32, 8, 2, 2)
        ESN_CASE(TraitsF32, 8, 4, 1)
    } else if (precision == ESN_F16) {
        ESN_CASE(TraitsF16, 4, 1, 2)
        ESN_CASE(TraitsF16, 8, 1, 4)
        ESN_CASE(TraitsF16, 8, 2, 4)
        ESN_CASE(TraitsF16, 8, 4, 2)
        ESN_CASE(TraitsF16, 16, 4, 1)
    } else if (precision == ESN_BF16) {
        ESN_CASE(TraitsBF16, 4, 1, 2)
        ESN_CASE(TraitsBF16, 8, 1, 4)
        ESN_CASE(TraitsBF16, 8, 2, 4)
        ESN_CASE(TraitsBF16, 8, 4, 2)
        ESN_CASE(TraitsBF16, 16, 4, 1)
    }
#undef ESN_CASE
    return -1;
}

}  // namespace esn
