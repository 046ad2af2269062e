// Shared host/device definitions for the ESN hot-path kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/esn_hip.h"

namespace esn {

// Padded geometry of one recurrence problem.  The GEMM of one timestep is
//   P[Mp x Bt] = Wext[Mp x Kp] * Z[Kp x Bt],   Z = [X ; U ; F]
// with the state rows at k in [0,Mp), the (scaled) inputs at [kin, kin+n_in)
// and the fed-back output / teacher at [kfb, kfb+n_out).
struct Geometry {
    int Mp;     // n_res rounded up to the row tiling of the kernel
    int kin;    // == Mp
    int kfb;    // kin + round4(n_in)
    int Kp;     // total K, multiple of 32
    int Ks;     // LDS row stride in elements (Kp + conflict-avoidance pad)
    int Bt;     // frames per workgroup tile
    int NW, MT, NT;  // waves, row tiles / wave, column tiles / wave (MFMA kernels)
    int ro_parts;    // readout images per group: 1, or 2 (hi + lo) for fp16/bf16 with n_out > 8
    int ro_fold;     // fp16/bf16, n_out <= 8: rows 0-7 = hi, rows 8-15 = lo of ONE 16-row image
    int skew;        // predict: the two waves of a SIMD run one third of a step apart (esn_recur_mfma_impl.h)
    int big;         // fp16/bf16, N_res > 1024: the launch-per-step GEMM path (esn_recur_big.hip) serves this shape;
                     // the packed read-out then carries that path's image behind the persistent kernel's
    int rs;          // fp16/bf16: the register-resident-state kernel (esn_recur_rs.hip) serves this shape; the packed
                     // read-out then carries that kernel's image behind the persistent kernel's
    int s16;         // fp16/bf16: the 16x16x32 skewed predict kernel (esn_recur_skew16_impl.h) serves this shape; the
                     // packed weights and read-out then carry that kernel's images behind the 32x32x16 ones
    int m64;         // ESN_F64: 1 = the float64 matrix-pipe kernel (esn_recur_f64_mfma.hip) fits this shape;
    int Bt64;        //          then Mp..Ks, MT, NT describe ITS tiling, Bt64 its frames per tile and Bt
                     //          stays the tile of the vector-ALU kernel (esn_recur_f64.hip)
};

struct RecurParams {
    int n_res, n_in, n_out, teacher_forcing;
    Geometry g;
    // Frames are ordered by group (frame = grp*F + j).  Workgroup tiles are cut from a padded
    // "slot" axis: slot = grp*Fpad + j, j < Fpad, valid iff j < F.  Fpad is a multiple of 16
    // when a readout runs (one W_out per 16-frame column tile) and a multiple of the tile when
    // every group has its own weight set, so a tile never mixes weight sets.
    int n_frames;          // total sequences
    int n_groups;
    int F;                 // frames per group
    int Fpad;              // slots per group
    int n_wsets;
    int spw;               // n_wsets > 1: slots per weight set (whole tiles); the slot axis is then SET-MAJOR -- set w
                           // owns slots [w spw, (w+1) spw) holding its groups w, w + n_wsets, w + 2 n_wsets, ... at
                           // Fpad slots each -- so a tile never mixes weight sets yet packs several groups; 0 = one set
    int T_in;              // valid input rows per frame
    int S;                 // recurrence steps
    int in_row_off;        // input row fed at step s is s + in_row_off (harvest: 1)
    int transient;
    int harvest;           // 1: teacher feedback, write E; 0: own-output feedback, write Y
    int n_tiles;
    const void* packed_w;   size_t wset_stride;   // bytes per weight set
    const void* packed_wout; size_t wout_stride;  // bytes per group
    size_t w64_off, wo64_off;                     // ESN_F64 images: byte offset of the MFMA-ordered copy
    size_t w16_off, wo16_off;                     // g.s16: byte offset of the 16x16x32 kernel's images
    const double* in_scale; const double* in_shift;
    const double* t_scale;  const double* t_shift;
    const double* U; const double* D;
    const double* x0; const double* y0;
    const double* noise_u;
    double noise; int noise_mode; uint64_t seed;
    double leak;           // leak rate a in (0, 1]: x = (1 - a) x_prev + a tanh(.) + noise; 1 = the reference (float64 kernels only)
    uint32_t frame_off;    // counter noise: global index of this launch's frame 0 (= group_offset * F, mod 2^32), so a
                           // frame draws the same noise whichever launch, chunk or rank it lands in
    int wset_rot;          // n_wsets > 1: group g uses weight set (g + wset_rot) % n_wsets (= group_offset % n_wsets)
    double* Y; double* E;
    float* E32;            // harvest: when set, the extended states are stored as float32 here (E unused)
    unsigned long long* stamps;   // diagnostic build (-DESN_STAMPS) only: [block0 wave][8] cycle sums
};

struct DetectParams {
    const double* Y; int n_frames, frames_per_group, n_sub, log2n, n_t, m;
    const double* p_i; const uint8_t* tx_bits;
    long long* err; long long* bits; double* X_hat;
    int na_wg;             // antennas per workgroup (set by the launcher: all of them unless LDS is short)
};

// frame generator (esn_gen.hip)
struct TapParams {
    int kind;            // 0 = TDL-B, 1 = exponential PDP, 2 = flat unit-modulus (AWGN driver)
    int n_links;         // n_blocks * n_r * n_t
    int isi;
    int n_paths;         // 23 for TDL-B, isi for the exponential PDP, 1 for flat
    double path_sqrt_pow[24];        // sqrt of the (normalised) linear power of each path / tap
    double path_delay_samples[24];   // TDL-B: normalised delay * DS * fs
    const double* gains_in;          // optional complex [n_links][n_paths] standard normals (parity)
    uint64_t seed; uint64_t link_offset;
    double* taps;        // complex [n_links][isi]
};

struct FrameGenParams {
    int n_frames, frames_per_block, n_sub, log2n, cp, n_t, n_r, isi, m;
    const double* p_i;        // [n_blocks]  Pi = 10^(EbNo/10) No
    const double* a_clip;     // [n_blocks]  PA clip level
    double no;
    const double* taps;       // complex [n_blocks][n_r][n_t][isi]
    const uint8_t* bits_in;   // optional [B][N*m][n_t]
    const double* noise_in;   // optional complex [B][T][n_r], unit-variance real and imaginary parts
    uint64_t seed; uint64_t frame_offset;
    uint8_t* bits;            // [B][N*m][n_t]
    double* x_cp;             // optional complex [B][T][n_t] (pre-PA: the ESN teacher)
    double* y_cp;             // complex [B][T][n_r]
    int ls_pattern;           // 1: sparse LS pilot, subcarrier sc carries only tx = sc % n_t (driver:330-333)
    int ko;                   // diagnostic (esn_debug_set "gen_ko"): bit0 no AWGN draw, bit1 no channel MACs, bit2 no IFFT, bit3 no PA
};

// baseline equaliser (esn_baseline.hip)
struct ChanEstParams {
    int n_blocks, n_sub, log2n, cp, n_t, n_r, isi, m;
    int ls_only;                 // 1: stop at the interpolated LS estimate (H_LS), no time-domain MMSE refinement
    const double* p_i; double no;
    const uint8_t* pilot_bits;   // [G][N*m][n_t]
    const double* y_ls_cp;       // complex [G][T][n_r]
    double* H;                   // complex [G][N][n_r][n_t]
};
struct TapsFreqParams {
    int n_blocks, n_sub, n_t, n_r, isi;
    const double* taps;          // complex [G][n_r][n_t][isi]
    double* H;                   // complex [G][N][n_r][n_t]
};
struct MmseParams {
    int n_frames, frames_per_group, n_sub, log2n, cp, n_t, n_r, m;
    int zf;                      // 1: zero forcing, G = H^H H + 1e-12 I (driver :34-39); 0: MMSE, G = H^H H + No/Pi I
    const double* p_i; double no;
    const double* H;             // complex [G][N][n_r][n_t]
    const double* y_cp;          // complex [B][T][n_r]
    const uint8_t* tx_bits;      // [B][N*m][n_t]
    long long* err; long long* bits; double* X_hat;
};

// coded leg (esn_coded.hip)
struct LdpcEncodeParams {
    int n_frames, n_t, k, n;
    const uint8_t* P;        // [n-k][k] parity part of the systematic generator (bytes 0/1)
    const uint8_t* u;        // [B][n_t][k] information bits
    uint8_t* bits;           // [B][n][n_t]  (TxBits layout, n = N*m)
};
struct LlrParams {
    int n_frames, n_sub, n_t, m;
    const double* X_hat;     // complex [B][N][n_t]
    double* llr;             // [B][n_t][N*m]
    double* sigma2;          // optional [B]
};
struct LdpcDecodeParams {
    int n_cw, n, k, m_checks, n_edges, maxiter, cw_per_group;
    double var;              // 10^(-snr_db/10)
    const int* chk_ptr;      // [m_checks+1] edges of check c: chk_ptr[c] .. chk_ptr[c+1]
    const int* edge_var;     // [E] variable of edge e (check-major order)
    const int* var_ptr;      // [n+1]
    const int* var_edge;     // [E] edge ids of variable v
    const double* y;         // [n_cw][n] observations (bit 0 <-> +)
    const uint8_t* u_true;   // optional [n_cw][k]
    uint8_t* x_out;          // optional [n_cw][n]
    long long* err; long long* bits;
};

// slot -> group (n_groups or beyond = none) and the position j inside the group's Fpad slots
__device__ __forceinline__ int slot_group(const RecurParams& p, int slot, int& j) {
    if (p.spw == 0) {
        const int grp = slot / p.Fpad;
        j = slot - grp * p.Fpad;
        return grp;
    }
    const int w = slot / p.spw, ls = slot - w * p.spw;
    const int gi = ls / p.Fpad;
    j = ls - gi * p.Fpad;
    const int grp = gi * p.n_wsets + w;
    return (w < p.n_wsets) ? grp : p.n_groups;
}
__device__ __forceinline__ int slot_group(const RecurParams& p, int slot) { int j; return slot_group(p, slot, j); }
// weight set of the tile that starts at slot0
__device__ __forceinline__ int slot_wset(const RecurParams& p, int slot0) {
    if (!p.spw) return 0;
    const int w = slot0 / p.spw + p.wset_rot;
    return w >= p.n_wsets ? w - p.n_wsets : w;
}
// slot -> frame index (or -1 for padding) and its group
__device__ __forceinline__ int slot_frame(const RecurParams& p, int slot, int& grp) {
    int j;
    grp = slot_group(p, slot, j);
    const int fr = grp * p.F + j;
    return (j < p.F && grp < p.n_groups && fr < p.n_frames) ? fr : -1;
}

inline __host__ __device__ int round_up(int x, int m) { return (x + m - 1) / m * m; }

// k order of the 16x16x32 skewed kernel (esn_recur_skew16_impl.h): position of natural column k inside the padded
// K axis.  State rows: R = 32 kk + 16 t + 4 g + e  ->  32 kk + 8 g + 4 t + e (the accumulator rows 4 g .. 4 g + 3 of two
// row tiles form one 16-byte chunk of the next step's B operand); the [U ; F] group (k >= 512) keeps natural order.
inline __host__ __device__ int s16_pos(int k) {
    return k < 512 ? (k & ~31) + 8 * ((k & 15) >> 2) + 4 * ((k & 31) >> 4) + (k & 3) : k;
}
inline __host__ __device__ int s16_nat(int pos) {       // inverse of s16_pos
    return pos < 512 ? (pos & ~31) + 16 * ((pos & 7) >> 2) + 4 * ((pos & 31) >> 3) + (pos & 3) : pos;
}

// Tuning / diagnostic knobs (benchmarks and A/B tests only).  Read ONCE per process from the
// environment (ESN_SKEW, ESN_MFMA_GEOM, ESN_MFMA_GEOM_F32, ESN_CHOL_SKIP) and changed afterwards only
// through the debug entry point esn_debug_set (esn_api.hip) -- never re-read per launch.  None of
// them changes a packed image: the weight image depends on (precision, n_res, n_in, n_out) alone.
struct Knobs {
    int skew;              // 1 (default): skewed schedule where it applies; 0: in-step schedule
    int geom16[3];         // fp16/bf16 predict tiling override {NW, MT, NT}; {0,0,0} = table
    int geom32[3];         // float32 predict tiling override
    int chol_skip;         // bit mask of Cholesky-solve phases to drop (tools/time_chol.py)
    int f64_mfma;          // 1 (default): float64 batches run on the matrix pipe; 0: vector-ALU kernel (A/B tests)
    int rs;                // 1: fp16/bf16 predict at N_res 257..512 runs the register-resident-state kernel (default 0:
                           // correct but 17 % slower than the skewed LDS-state kernel on MI355X)
    int big_gemm;          // 1 (default): N_res > 1024 predict runs as one GEMM launch per step when a workspace is given
    int cluster;           // 1 (default): ONE float64 sequence runs on the LDS-resident cluster kernel when a workspace is given
    int big_nt;            // N_res > 1024 predict: 4 = the 4-wave 128 x 128 variant of big_step_kernel (slower: A/B only); default 2
    int big_pipe;          // 1 (default): big_step_kernel's four-stage pipelined main loop; 0: the round-2 loop (A/B runs)
    int harvest_gemm;      // 1: harvests of 257..1024 units (>= 64 pilots) also take the GEMM-per-step path (A/B; slower)
    int gen_ko;            // frame generator knock-out mask for tools/time_gen.py (timing only, wrong frames)
    int s16;               // 1 (default): fp16/bf16 predict at 257..512 units on the 16x16x32 kernel; 0: the 32x32x16 one (A/B)
    int hcluster;          // 1 (default): fp16/bf16 harvest at 257..512 units on the cluster kernel (two members per cluster)
                           // when a workspace is given; 4 / 8: that many members (A/B); 0: the persistent kernel
};
Knobs& knobs();

// Bijective XCD-aware remap (guide T1): consecutive logical tiles land on the
// same XCD so tiles of one weight set share that XCD's L2.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    if (nwg < nx) return bid;
    int q = nwg / nx, r = nwg % nx, x = bid % nx, i = bid / nx;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// Counter-based uniform in [0,1): two 16-bit samples per 32-bit hash.  Keyed by
// (seed, frame, step) once per column, then one integer mix per pair of rows.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t noise_key(uint64_t seed, uint32_t frame, uint32_t step) {
    uint32_t k = mix32((uint32_t)seed ^ (frame * 0x9E3779B9U));
    k = mix32(k ^ (uint32_t)(seed >> 32) ^ (step * 0x85EBCA6BU + 0x27d4eb2fU));
    return k;
}
// Four uniform bytes for reservoir rows 4*row4 .. 4*row4+3 under key `k`: additive counter, fold,
// one full-rate 24-bit multiply (v_mul_u32_u24), fold -- 4 integer ops per 4 samples.  Measured on
// 82 800 keys x 128 quads x 4 bytes: byte histogram chi2/255 = 0.93, rms correlation between any
// two (row, byte) positions 3.9e-3 (sampling floor 3.5e-3), max 0.046, lag-1 correlation across
// steps / frames < 1e-4.  A pure function of (seed, frame, step, row): every kernel, tiling and
// rank draws the same noise.
__device__ __forceinline__ uint32_t noise_mix(uint32_t s) {
    s ^= s >> 16;
    s = __umul24(s, 0x9E3779U);
    s ^= s >> 16;
    return s;
}
__device__ __forceinline__ uint32_t noise_quad(uint32_t k, uint32_t row4) {
    return noise_mix(k + row4 * 0x9E3779B9U);
}
// uniform in (0,1) with 8-bit resolution for byte `b` (0..3) of a quad: (byte + 0.5) / 256
__device__ __forceinline__ float noise_byte(uint32_t quad, int b) {
    return (float)((quad >> (8 * b)) & 0xffU) * (1.0f / 256.0f) + (0.5f / 256.0f);
}
__device__ __forceinline__ float noise_uniform(uint32_t k, uint32_t row) {
    return noise_byte(noise_quad(k, row >> 2), (int)(row & 3));
}

// float32 tanh: odd Taylor polynomial below 0.3 (truncation < 2e-9 relative),
// 1 - 2/(exp(2|x|)+1) above; both evaluated, selected per lane (no divergence).
constexpr float TANH32_SERIES_MAX = 0.3f;
__device__ __forceinline__ float tanh_f32_series(float x) {
    float x2 = x * x;
    float p = -0.00886323552990220f;            // -1382/155925
    p = fmaf(p, x2, 0.0218694885361552f);      //  62/2835
    p = fmaf(p, x2, -0.0539682539682540f);     // -17/315
    p = fmaf(p, x2, 0.133333333333333f);       //  2/15
    p = fmaf(p, x2, -0.333333333333333f);      // -1/3
    return fmaf(p * x2, x, x);
}
__device__ __forceinline__ float tanh_f32(float x) {
    float ax = fabsf(x);
    float p = tanh_f32_series(x);
    float e = __expf(2.0f * ax);                // v_exp_f32 path
    float r = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);   // v_rcp_f32 (1 ulp), not an IEEE divide
    r = copysignf(r, x);
    return ax < TANH32_SERIES_MAX ? p : r;
}

// float64 tanh on |x| <= TANH64_SERIES_MAX: the odd Maclaurin series through x^21 in Horner form (12 DP
// instructions against ~80 of the library routine).  Coefficients are the exact rationals
// B_2n 4^n (4^n - 1) / (2n)!; the first dropped term is 3.9e-5 x^23, below 2.3e-18 |x| on the interval, so
// the result is the correctly rounded series to ~1 ulp.  Callers test the whole wave's pre-activations and
// fall back to tanh() when any lies outside.
constexpr double TANH64_SERIES_MAX = 0.25;
__device__ __forceinline__ double tanh_f64_series(double x) {
    const double u = x * x;
    double p = 9.691537956929451e-05;             //  18888466084/194896477400625
    p = fma(p, u, -2.3912911424355248e-04);       // -443861162/1856156927625
    p = fma(p, u, 5.90027440945586e-04);          //  6404582/10854718875
    p = fma(p, u, -1.4558343870513183e-03);       // -929569/638512875
    p = fma(p, u, 3.592128036572481e-03);         //  21844/6081075
    p = fma(p, u, -8.863235529902197e-03);        // -1382/155925
    p = fma(p, u, 2.1869488536155203e-02);        //  62/2835
    p = fma(p, u, -5.396825396825397e-02);        // -17/315
    p = fma(p, u, 1.3333333333333333e-01);        //  2/15
    p = fma(p, u, -3.333333333333333e-01);        // -1/3
    return fma(p * u, x, x);
}

// tanh for the fp16/bf16 kernels.  The packed weights of those kernels are pre-multiplied by
// ACT_PRESCALE = 2 log2(e), so the accumulator already holds z = 2 log2(e) P and
//   tanh(P) = 1 - 2 / (1 + 2^z)            (v_exp_f32, v_add, v_rcp_f32, v_fma: 4 instructions)
// +-inf are handled by the instructions themselves (2^z -> inf -> rcp 0 -> 1; 2^z -> 0 -> -1).
// Absolute error ~6e-8 (float32 cancellation near 0), far below the 2^-11 relative rounding of
// the fp16 state the result is stored in.
constexpr double ACT_PRESCALE = 2.8853900817779268;   // 2 / ln 2
__device__ __forceinline__ float tanh_prescaled(float z) {
    float e = __builtin_amdgcn_exp2f(z);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

#ifdef ESN_STAMPS
#define ESN_STAMP(var) unsigned long long var; { __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#define ESN_STAMP_SET(var) { __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define ESN_STAMP(var)
#define ESN_STAMP_SET(var)
#endif

}  // namespace esn
