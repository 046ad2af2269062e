/* esn_hip.h -- C ABI of the MI355X-native ESN OFDM/MIMO detector hot path.
 *
 * Shared library: esn_ofdm_mimo_amd/libesn_hip.so (built by __graft_entry__.build()).
 * Every entry point is extern "C", takes plain pointers and sizes, returns an int
 * status (0 = ok, <0 = error; text via esn_last_error()) and launches its work
 * on the caller's hipStream_t (passed as void*; NULL = default stream).  No
 * hidden global state besides a thread-local error string and the tuning knobs
 * of esn_debug_set (below; never needed by a user); no exceptions cross
 * the boundary.  All array arguments are DEVICE pointers, except in the `esn_*_mem` entry points
 * called with ESN_MEM_HOST (below).  Arrays keep the reference's own row-major float64 layouts so a
 * binding needs no repacking:
 *
 *   inputs   U  [B][T_in][n_in]     (== complex128 [B][T_in][N_r] viewed as float64
 *                                    with Re/Im interleaved: the packing of
 *                                    Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py:433-436 and
 *                                    helper_mimo_esn_generic.py:30-33 is a VIEW; the
 *                                    d trailing zero rows are synthesised: T > T_in)
 *   outputs  Y  [B][T-transient][n_out]  (== complex128 [B][N][N_t] view, :47-58)
 *
 * Reference interface each entry point replaces (paths relative to the
 * reference tree, libs/pyESN.py unless noted):
 *
 *   esn_pack_weights        ESN.initweights result W/W_in/W_feedb (:93-109) ->
 *                           device-resident, MFMA-fragment-ordered copy.
 *   esn_pack_readout        W_out of ESN.fit (:191-192) -> fragment-ordered copy.
 *   esn_predict_batch       ESN.predict (:218-255) incl. _scale_inputs (:127-135),
 *                           _update (:111-125), readout (:252), _unscale_teacher
 *                           (:146-152); batched over B frames in G groups.
 *   esn_harvest_batch       state-harvest loop of ESN.fit (:176-182,189) ->
 *                           extended states [states, inputs_scaled].
 *   esn_readout_solve_batch, esn_readout_solve_chol_batch
 *                           pinv solve of ESN.fit (:191-192).
 *   esn_gen_taps, esn_gen_frames   transmitter + channel + noise of the drivers (:127-177, :397-427).
 *   esn_channel_estimate, esn_mmse_detect_count   LS/MMSE baseline (:358-382, :40-45, :444-448).
 *   esn_detect_count        driver tail: reconstruct (:47-58 of the 4x8 driver),
 *                           (1/N) FFT / sqrt(Pi) (:439-441), hard decision
 *                           (:95-103), bit-error count (:451-456).
 */
#ifndef ESN_HIP_H
#define ESN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Arithmetic of the recurrence. */
enum esn_precision {
    ESN_F64 = 0,   /* float64, the reference's arithmetic: v_mfma_f64_16x16x4_f64 for batches (N_res <= 1024),
                      float64 FMA on the vector ALU for a single sequence and larger reservoirs */
    ESN_F32 = 1,   /* v_mfma_f32_32x32x2_f32: exact float32 products + accumulate */
    ESN_F16 = 2,   /* v_mfma_f32_16x16x32_f16 (predict at 257..512 units) / v_mfma_f32_32x32x16_f16: fp16 operands,
                      float32 accumulate */
    ESN_BF16 = 3   /* the same with bf16 operands */
};

/* State-noise source (pyESN.py:124-125: + noise * (U[0,1) - 0.5)). */
enum esn_noise_mode {
    ESN_NOISE_NONE = 0,     /* noise == 0                                             */
    ESN_NOISE_TENSOR = 1,   /* caller supplies the uniforms [B][S][n_res] (parity)     */
    ESN_NOISE_COUNTER = 2   /* counter-based generator keyed (seed, frame, step, row)  */
};

/* Geometry shared by the recurrence entry points. */
typedef struct esn_shape {
    int n_res;            /* reservoir size                                          */
    int n_in;             /* input units  (2 N_r)                                    */
    int n_out;            /* output units (2 N_t)                                    */
    int teacher_forcing;  /* 1: W_feedb term active (pyESN.py:117-120)               */
    int n_wsets;          /* number of (W, W_in, W_feedb) sets: 1 = shared reservoir */
    double leak_rate;     /* EXTENSION (the reference has none): x[t] = (1-a) x[t-1] + a tanh(...) + noise.
                             0 (what `{n_res, n_in, n_out, tf, n_wsets}` initialises it to) or 1 = the reference's
                             update.  Values in (0, 1) are served by ESN_F64 only (other precisions answer -2). */
} esn_shape_t;

const char* esn_last_error(void);

/* ABI version (bumped on any signature change). */
int esn_abi_version(void);

/* Tuning / diagnostic knobs for benchmarks and A/B tests (no counterpart in the reference).  The
 * library reads ESN_SKEW, ESN_MFMA_GEOM, ESN_MFMA_GEOM_F32, ESN_CHOL_SKIP and ESN_F64_MFMA from the environment
 * ONCE, at its first call, as initial values; afterwards only this call changes them:
 *   "skew"          "0" = in-step schedule for the fp16/bf16 predict kernel, else skewed (default)
 *   "mfma_geom"     "NW,MT,NT" re-cuts the fp16/bf16 predict tiling; ignored unless 32*NW*MT equals
 *   "mfma_geom_f32" the table's padded row count, so a packed image never goes stale; NULL = table
 *   "chol_skip"     bit mask of Cholesky-solve phases to drop (timing only, wrong results)
 *   "f64_mfma"      "0" = ESN_F64 batches on the vector-ALU kernel instead of the float64 matrix pipe
 *   "rs"            "1" = fp16/bf16 predict at N_res 257..512 on the register-resident-state kernel
 *                   (esn_recur_rs.hip; an experiment kept for A/B runs, compiled only into ESN_WITH_RS=1 builds:
 *                   the product library answers -3) instead of the skewed LDS-state kernel
 *   "s16"           "0" = fp16/bf16 predict at 257..512 units on the 32x32x16 skewed kernel instead of the 16x16x32 one
 *                   (esn_recur_skew16_impl.h, the default: same schedule, 4 % less wall time at a higher clock; A/B runs)
 *   "big_gemm"      "0" = N_res > 1024 predict on the persistent kernel even when a workspace is given
 *   "cluster"       "0" = a single float64 sequence on the vector-ALU kernel even when a workspace is given
 *   "big_pipe"      "0" = N_res > 1024 predict with the round-2 main loop (two buffers, two barriers per chunk)
 *   "big_nt"        "4" = N_res > 1024 predict on the 4-wave 128 x 128 variant (slower; A/B runs)
 *   "harvest_gemm"  "1" = harvests of 257..1024 units (>= 64 pilots) on the GEMM-per-step path (slower; A/B runs)
 *   "hcluster"      "0" = fp16/bf16 harvest at 257..512 units on the persistent kernel even when a workspace is given;
 *                   "4" / "8" = clusters of that many workgroups instead of pairs (esn_harvest_cluster.hip; A/B runs).
 *                   esn_harvest_workspace_bytes follows the knob: ask for the size after setting it
 *   "gen_ko"        frame-generator knock-out mask for tools/time_gen.py (timing only, wrong frames)
 * Returns 0, or -1 for an unknown key. */
int esn_debug_set(const char* key, const char* value);

/* Device facts used for roofline reporting (any pointer may be NULL). */
int esn_device_info(int* cu_count, int* lds_bytes_per_cu, int* clock_khz,
                    char* arch_name, int arch_name_len);

/* Frames per workgroup tile the recurrence kernel uses for this shape and
 * precision (>= 1).  Padding happens inside the kernel, never in the caller's
 * arrays: a group's frames are padded to a multiple of 16 (the MFMA column
 * tile; no padding on the float64 vector-ALU path), and with n_wsets > 1 the
 * groups of one weight set are laid end to end and that span is rounded up to
 * whole tiles, so a tile streams one weight image and may serve several groups. */
int esn_tile_frames(int precision, const esn_shape_t* shape);

/* Bytes of the packed weight image for one weight set / one readout. */
size_t esn_packed_weights_bytes(int precision, const esn_shape_t* shape);
size_t esn_packed_readout_bytes(int precision, const esn_shape_t* shape);

/* W [n_wsets][n_res][n_res], W_in [n_wsets][n_res][n_in], W_fb [n_wsets][n_res][n_out]
 * (float64, row-major) -> packed [n_wsets][esn_packed_weights_bytes]. */
int esn_pack_weights(int precision, const esn_shape_t* shape,
                     const double* W, const double* W_in, const double* W_fb,
                     void* packed, void* stream);

/* W_out [n_groups][n_out][n_res+n_in] (float64) -> packed [n_groups][esn_packed_readout_bytes]. */
int esn_pack_readout(int precision, const esn_shape_t* shape, int n_groups,
                     const double* W_out, void* packed, void* stream);

/* Batched ESN.predict.
 *
 * Frames are ordered by group: frame b belongs to group b / frames_per_group
 * (its W_out, scalings, initial state) and, when shape->n_wsets > 1, to weight
 * set (group_offset + b / frames_per_group) % n_wsets.
 *
 *   group_offset  global index of this call's group 0 in the caller's sweep (0 for a stand-alone call).  The
 *                 counter noise of frame b is keyed by (seed, group_offset * frames_per_group + b, step, row) and the
 *                 weight set by the global group, so cutting a sweep into chunks, launches or ranks
 *                 changes neither: a frame's result is a function of its global index only.  (The frame index
 *                 enters the key modulo 2^32.)
 *
 *   packed_w      from esn_pack_weights            packed_wout  from esn_pack_readout
 *   in_scale/in_shift   [n_groups][n_in]  or NULL (=1 / =0)        (pyESN.py:131-134)
 *   t_scale/t_shift     [n_groups][n_out] or NULL                  (pyESN.py:140-151)
 *   U             [B][T_in][n_in]; rows T_in..T-1 are zeros before scaling
 *   x0, y0        [n_groups][n_res], [n_groups][n_out] start state and fed-back
 *                 output (continuation=True: laststate/lastoutput, :234-237) or NULL (zeros)
 *   noise_u       [B][T][n_res] uniforms when noise_mode == ESN_NOISE_TENSOR
 *   Y             [B][T-transient][n_out], unscaled (:255); 16-byte aligned (rows are written as 16-byte pairs)
 *   workspace     device scratch of esn_predict_workspace_bytes(...) bytes, or NULL.  Two shapes use it:
 *                 reservoirs beyond 1024 units in fp16/bf16 (the recurrence runs as one tiled GEMM launch
 *                 per timestep with the state images in the workspace), and ONE float64 sequence (n_frames = 1:
 *                 the reference's own call pattern) -- the matrix then stays resident in the LDS of a cluster of
 *                 co-resident workgroups that exchange the state through the workspace every step
 *                 (esn_recur_cluster.hip); its last 64 bytes hold an error word that is non-zero if a workgroup
 *                 timed out waiting for the others (the outputs are then invalid).  With NULL every shape runs on
 *                 the persistent kernels.  No allocation happens inside the call either way.
 */
size_t esn_predict_workspace_bytes(int precision, const esn_shape_t* shape, int n_frames, int frames_per_group);
int esn_predict_batch(int precision, const esn_shape_t* shape,
                      const void* packed_w, const void* packed_wout,
                      const double* in_scale, const double* in_shift,
                      const double* t_scale, const double* t_shift,
                      const double* U, int n_frames, int frames_per_group,
                      int T_in, int T, int transient,
                      const double* x0, const double* y0,
                      double noise, int noise_mode, const double* noise_u,
                      uint64_t seed, uint64_t group_offset,
                      double* Y, void* workspace, size_t workspace_bytes, void* stream);

/* Batched state harvest of ESN.fit: one training sequence per group.
 *
 *   U [n_groups][T][n_in], D [n_groups][T][n_out] (teacher, unscaled)
 *   E [n_groups][T][n_res+n_in] = hstack(states, inputs_scaled) (:189); row 0 of
 *   the states is zero and input row 0 is never fed (:179-182).
 *   noise_u [n_groups][T-1][n_res] when noise_mode == ESN_NOISE_TENSOR.
 *   group_offset: as in esn_predict_batch (noise key and weight set follow the GLOBAL group index);
 *   E is 16-byte aligned.
 *   workspace: device scratch of esn_harvest_workspace_bytes(...) bytes, or NULL.  Three shapes use it: reservoirs
 *   beyond 1024 units in fp16/bf16 (one tiled GEMM launch per timestep, 128 x 64 tiles: a fit has one sequence per
 *   trained ESN, so the tile is cut for workgroup count); fp16/bf16 at 257..512 units (pairs
 *   of co-resident workgroups keep the weight matrix in registers and exchange their state slices through the
 *   workspace every step, esn_harvest_cluster.hip; its last 64 bytes hold an error word that is non-zero if a
 *   workgroup timed out waiting for its peer: the states are then invalid); and ONE float64 sequence (the cluster
 *   kernel of esn_predict_batch).  With NULL the persistent kernels run.
 *   precision: ESN_F64 / ESN_F32 keep the states at (better than) float32; ESN_F16 / ESN_BF16
 *   harvest states rounded to the operand type (round-off ~6e-6 abs, far below the model's own
 *   state noise 2.9e-4 rms) -- statistically equivalent, not bit-comparable.
 */
size_t esn_harvest_workspace_bytes(int precision, const esn_shape_t* shape, int n_groups);
int esn_harvest_batch(int precision, const esn_shape_t* shape,
                      const void* packed_w,
                      const double* in_scale, const double* in_shift,
                      const double* t_scale, const double* t_shift,
                      const double* U, const double* D, int n_groups, int T,
                      double noise, int noise_mode, const double* noise_u,
                      uint64_t seed, uint64_t group_offset, double* E,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Same harvest, extended states stored as float32 (MFMA precisions only: their state columns are
 * exactly representable, the scaled-input columns round at 6e-8 relative).  Halves the harvest's
 * store tail and the two passes esn_readout_solve_chol_batch_f32 makes over E.  No counterpart in
 * the reference (its states are float64, :176); an internal fast path of the batched fit. */
int esn_harvest_batch_f32(int precision, const esn_shape_t* shape,
                          const void* packed_w,
                          const double* in_scale, const double* in_shift,
                          const double* t_scale, const double* t_shift,
                          const double* U, const double* D, int n_groups, int T,
                          double noise, int noise_mode, const double* noise_u,
                          uint64_t seed, uint64_t group_offset, float* E,
                          void* workspace, size_t workspace_bytes, void* stream);

/* W_out[g] = (pinv(E[g][transient:]) @ (D[g][transient:]*t_scale + t_shift)).T  (:191-192)
 *
 * float64 Householder QR: of E^T when rows < cols (minimum-norm solution, what
 * pinv returns for the under-determined 4x8 case) and of E otherwise.
 *   workspace      esn_readout_solve_workspace_bytes(...) bytes of device memory
 *   status [n_groups]  0 = ok, 1 = numerically rank deficient (|r_jj| <= 1e-13 max|r|;
 *                  the dependent direction is dropped), written on device
 */
size_t esn_readout_solve_workspace_bytes(int n_groups, int rows, int cols, int n_out);
int esn_readout_solve_batch(const double* E, const double* D, int n_groups, int T,
                            int transient, int cols, int n_out,
                            const double* t_scale, const double* t_shift,
                            double* W_out, int* status, void* workspace, void* stream);

/* Same contract, normal equations in float64 on the float64 matrix pipe -- the fast path for well-conditioned
 * batched fits (with the model's state noise cond(E) ~ 1e3, error ~ cond^2 eps ~ 1e-10).  min(rows, cols) <= 128:
 * Gram matrix and Cholesky factor resident in LDS, no workspace (NULL).  129 .. 512 (4x8 at N = 512: 512 x 528;
 * N_res = 300: 512 x 316): Gram matrix / factor column-major in the caller's workspace of
 * esn_readout_chol_workspace_bytes(...) bytes (16-byte aligned, as E), left-looking blocked factorisation.
 * status[g] = 1 when a pivot was rejected: re-solve that group with esn_readout_solve_batch.  Needs n_out <= 8;
 * returns -2 when the shape is not served. */
size_t esn_readout_chol_workspace_bytes(int n_groups, int rows, int cols);
int esn_readout_solve_chol_batch(const double* E, const double* D, int n_groups, int T,
                                 int transient, int cols, int n_out,
                                 const double* t_scale, const double* t_shift,
                                 double* W_out, int* status, void* workspace, size_t workspace_bytes, void* stream);
/* ... with E as written by esn_harvest_batch_f32 (arithmetic still float64). */
int esn_readout_solve_chol_batch_f32(const float* E, const double* D, int n_groups, int T,
                                     int transient, int cols, int n_out,
                                     const double* t_scale, const double* t_shift,
                                     double* W_out, int* status, void* workspace, size_t workspace_bytes, void* stream);

/* Fused detector tail: Y [B][N][2 N_t] time-domain ESN outputs
 * -> (1/N) FFT_N / sqrt(Pi[group]) -> nearest unit-power square-QAM point ->
 * natural-binary LSB-first bits -> compare with tx_bits [B][N*m][N_t] (uint8) ->
 * err_count[group] += mismatches, bit_count[group] += N*m*N_t  (int64, device).
 * X_hat (complex128 [B][N][N_t] as float64 pairs) is optional (NULL to skip). */
int esn_detect_count(const double* Y, int n_frames, int frames_per_group,
                     int n_sub, int n_t, int bits_per_sym,
                     const double* p_i, const uint8_t* tx_bits,
                     long long* err_count, long long* bit_count,
                     double* X_hat, void* stream);

/* ---- Host-memory front ends ("caller-owned device or host pointers flagged by an enum").
 * The reference's callers hold C-contiguous float64 NumPy arrays on the host (pyESN.py:154,218); a binding that
 * lives there passes them as they are with ESN_MEM_HOST.  Each `esn_X_mem(mem_kind, ...)` takes the arguments
 * of `esn_X(...)`:
 *   ESN_MEM_DEVICE  forwards to esn_X unchanged (asynchronous on `stream`, no allocation).
 *   ESN_MEM_HOST    every ARRAY argument (weights, scalings, U, D, x0, y0, noise_u, Y, E, W_out, status, p_i,
 *                   tx_bits, counters, X_hat) is host memory: the call stages it in device memory from the
 *                   stream-ordered pool, runs esn_X on `stream`, copies the results back and returns after
 *                   the stream has drained, so the host arrays are complete on return.  The counters of
 *                   esn_detect_count_mem are read, added to and written back.
 * `packed` images and `workspace` are device memory in both kinds -- esn_device_alloc / esn_device_free hand
 * them to a caller that has no HIP toolchain of its own (sizes from esn_packed_*_bytes / esn_*_workspace_bytes;
 * esn_readout_solve_batch_mem takes its scratch from the pool when `workspace` is NULL).
 * Returns as esn_X; -1 for an unknown mem_kind. */
enum esn_mem_kind {
    ESN_MEM_DEVICE = 0,
    ESN_MEM_HOST = 1
};
void* esn_device_alloc(size_t bytes);   /* NULL on failure (esn_last_error()) */
int esn_device_free(void* p);
int esn_pack_weights_mem(int mem_kind, int precision, const esn_shape_t* shape,
                         const double* W, const double* W_in, const double* W_fb,
                         void* packed, void* stream);
int esn_pack_readout_mem(int mem_kind, int precision, const esn_shape_t* shape, int n_groups,
                         const double* W_out, void* packed, void* stream);
int esn_predict_batch_mem(int mem_kind, int precision, const esn_shape_t* shape,
                          const void* packed_w, const void* packed_wout,
                          const double* in_scale, const double* in_shift,
                          const double* t_scale, const double* t_shift,
                          const double* U, int n_frames, int frames_per_group,
                          int T_in, int T, int transient,
                          const double* x0, const double* y0,
                          double noise, int noise_mode, const double* noise_u,
                          uint64_t seed, uint64_t group_offset,
                          double* Y, void* workspace, size_t workspace_bytes, void* stream);
int esn_harvest_batch_mem(int mem_kind, int precision, const esn_shape_t* shape,
                          const void* packed_w,
                          const double* in_scale, const double* in_shift,
                          const double* t_scale, const double* t_shift,
                          const double* U, const double* D, int n_groups, int T,
                          double noise, int noise_mode, const double* noise_u,
                          uint64_t seed, uint64_t group_offset, double* E,
                          void* workspace, size_t workspace_bytes, void* stream);
int esn_readout_solve_batch_mem(int mem_kind, const double* E, const double* D, int n_groups, int T,
                                int transient, int cols, int n_out,
                                const double* t_scale, const double* t_shift,
                                double* W_out, int* status, void* workspace, void* stream);
int esn_detect_count_mem(int mem_kind, const double* Y, int n_frames, int frames_per_group,
                         int n_sub, int n_t, int bits_per_sym,
                         const double* p_i, const uint8_t* tx_bits,
                         long long* err_count, long long* bit_count,
                         double* X_hat, void* stream);

/* ---- Monte-Carlo frame generator: the producer directly upstream of the detector.
 * float64 / complex128 like the reference; counter-based random streams keyed by
 * (seed, global frame / link index), so a frame is identical on any rank and launch shape.
 * Each random input may instead be supplied (bits_in, noise_in, gains_in = standard normals):
 * the deterministic mode the parity tests use.
 *
 * esn_gen_taps   per-link impulse responses [n_blocks][n_r][n_t][isi] complex128:
 *                kind 0 TDL-B (Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py:127-177: 23 paths, delays
 *                tau*DS*fs split linearly between floor/ceil taps, CN(0,p) gains, unit energy),
 *                kind 1 exponential-PDP Rayleigh (OFDM_MIMO_2-2_NBF_LDPC.py:162-164,272-279),
 *                kind 2 flat unit-modulus (Demo_SISO_QPSK_AWGN_LDPC_ESN_with_ZF_LS.py:205-206).
 * esn_gen_frames bits -> 2^m-QAM (:406-411) -> N*ifft (:416) -> CP (:417) -> sqrt(Pi) -> PA (:419)
 *                -> per-link FIR with zero initial state (:422-425) -> AWGN sqrt(T No/2) (:426).
 *                p_i / a_clip are per block [n_blocks]; x_cp (pre-PA teacher) may be NULL.
 *                ls_pattern = 1 keeps only tx = sc % n_t on subcarrier sc (the sparse LS pilot of
 *                :330-333); with the pilot's seed / frame index it shares the pilot's bits AND noise
 *                (:354-356), as the reference does. */
int esn_gen_taps(int kind, int n_blocks, int n_r, int n_t, int isi, double fs_hz, double ds_ns,
                 const double* gains_in, uint64_t seed, uint64_t link_offset,
                 double* taps, void* stream);
int esn_gen_frames(int n_frames, int frames_per_block, int n_sub, int cp, int n_t, int n_r, int isi,
                   int bits_per_sym, int ls_pattern, const double* p_i, const double* a_clip, double no,
                   const double* taps, const uint8_t* bits_in, const double* noise_in,
                   uint64_t seed, uint64_t frame_offset,
                   uint8_t* bits, double* x_cp, double* y_cp, void* stream);

/* ---- Baseline equaliser the reference compares the ESN with, float64.
 * esn_channel_estimate   pilot_bits [G][N*m][n_t], y_ls_cp complex [G][T][n_r] (received sparse LS
 *                        pilot) -> H complex [G][N][n_r][n_t]: LS at sc = tx + n_t i, linear
 *                        inter/extrapolation, IFFT -> isi taps, diagonal MMSE shrinkage, DFT
 *                        (Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py:358-382); ls_only = 1 stops at the interpolated LS
 *                        estimate, the H_LS the block-fading drivers feed their LS-ZF detector
 *                        (OFDM_MIMO_2-2_NBF_LDPC.py:321-333,457).
 * esn_mmse_detect_count  y_cp complex [B][T][n_r] -> X = (H^H H + No/Pi I)^-1 H^H Y / sqrt(Pi) per
 *                        subcarrier (:40-45, :444-448), hard decision + error count as in
 *                        esn_detect_count; n_t <= 4.  X_hat complex [B][N][n_t] optional.
 * esn_zf_detect_count    the same with G = H^H H + 1e-12 I: equalize_zf (:34-39; OFDM_MIMO_2-2_NBF_LDPC.py:41-47),
 *                        "LS-ZF" with an estimated H and "Perfect-ZF" with the true one (:450-460).
 * esn_taps_to_freq       taps complex [G][n_r][n_t][isi] -> the true channel H complex [G][N][n_r][n_t]
 *                        = FFT_N of the zero-padded impulse response (H_true, OFDM_MIMO_2-2_NBF_LDPC.py:273-279). */
int esn_channel_estimate(int n_blocks, int n_sub, int cp, int n_t, int n_r, int isi, int bits_per_sym,
                         const double* p_i, double no, const uint8_t* pilot_bits,
                         const double* y_ls_cp, int ls_only, double* H, void* stream);
int esn_mmse_detect_count(int n_frames, int frames_per_group, int n_sub, int cp, int n_t, int n_r,
                          int bits_per_sym, const double* p_i, double no, const double* H,
                          const double* y_cp, const uint8_t* tx_bits,
                          long long* err_count, long long* bit_count, double* X_hat, void* stream);
int esn_zf_detect_count(int n_frames, int frames_per_group, int n_sub, int cp, int n_t, int n_r,
                        int bits_per_sym, const double* p_i, const double* H,
                        const double* y_cp, const uint8_t* tx_bits,
                        long long* err_count, long long* bit_count, double* X_hat, void* stream);
int esn_taps_to_freq(int n_blocks, int n_sub, int n_t, int n_r, int isi, const double* taps, double* H, void* stream);

/* ---- Coded leg of the north-star driver, float64.  The reference delegates the code
 * to the un-vendored package pyldpc (requirements-sm2.txt:5); these entry points restate its
 * published algorithms at the reference's call sites (parity unpinned, see oracle/ldpc_oracle.py).
 * esn_ldpc_encode        c = [u ; P u mod 2] per (frame, tx) into TxBits [B][n][n_t], n = N*m
 *                        (Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py:90-93, :399-404); P [n-k][k] bytes.
 * esn_qam_llr            X_hat complex [B][N][n_t] -> max-log LLRs [B][n_t][N*m] (positive = bit 0)
 *                        scaled by the decision-directed sigma^2 of the frame (:66-88, :108-112, :459-469).
 * esn_ldpc_decode_count  pyldpc.decode: flooding log-domain sum-product on y [n_cw][n] with
 *                        var = 10^(-snr_db/10), Lc = 2 y / var, at most maxiter sweeps, early stop on a
 *                        zero syndrome (:495-496); message = first k bits (get_message); info-bit
 *                        errors vs u_true accumulated per group of cw_per_group codewords (:508-511).
 *                        Graph in CSR form: chk_ptr [m+1], edge_var [E] (check-major), var_ptr [n+1],
 *                        var_edge [E]. */
int esn_ldpc_encode(int n_frames, int n_t, int k, int n, const uint8_t* P, const uint8_t* u,
                    uint8_t* bits, void* stream);
int esn_qam_llr(int n_frames, int n_sub, int n_t, int bits_per_sym, const double* X_hat,
                double* llr, double* sigma2, void* stream);
int esn_ldpc_decode_count(int n_cw, int n, int k, int m_checks, int n_edges,
                          const int* chk_ptr, const int* edge_var, const int* var_ptr, const int* var_edge,
                          const double* y, double snr_db, int maxiter, const uint8_t* u_true, int cw_per_group,
                          uint8_t* x_out, long long* err_count, long long* bit_count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ESN_HIP_H */
