// C ABI entry points (include/esn_hip.h): argument validation, geometry choice,
// kernel launches.  No allocation, no synchronisation, no global state besides
// the thread-local error string -- every call is capturable into a hipGraph.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include "esn_common.h"

namespace esn {
// esn_recur_f64.hip
int launch_recur_f64(const RecurParams& p, hipStream_t stream);
size_t recur_f64_lds_bytes(int FB, int n_res, int n_in, int n_out);
// esn_recur_f64_mfma.hip
bool f64_mfma_geometry(int n_res, int n_in, int n_out, bool harvest, Geometry* g);
int launch_recur_f64_mfma(const RecurParams& p, hipStream_t stream);
#ifdef ESN_WITH_RS
// esn_recur_rs.hip (a kept negative result, DESIGN.md 3.1b: only in builds made with ESN_WITH_RS=1)
bool rs_path_applies(int precision, const RecurParams& p);
int launch_recur_rs(int precision, const RecurParams& p, size_t wo_rs_off, hipStream_t stream);
#endif
// esn_recur_big.hip
bool big_path_applies(int precision, const RecurParams& p);
int big_slots(const RecurParams& p);
size_t big_workspace_bytes(int n_slots, int Mp, int Kp);
int launch_recur_big(int precision, const RecurParams& p, size_t wo_big_off, void* workspace, hipStream_t stream);
bool big_harvest_applies(int precision, const RecurParams& p);
size_t big_harvest_workspace_bytes(int n_groups, int Kp);
int launch_harvest_big(int precision, const RecurParams& p, void* workspace, hipStream_t stream);
// esn_recur_cluster.hip
bool cluster_applies(int precision, const RecurParams& p);
size_t cluster_workspace_bytes(int n_res, int n_in, int n_out, bool harvest);
int launch_recur_cluster(const RecurParams& p, void* workspace, hipStream_t stream);
// esn_recur_mfma.hip
bool mfma_geometry(int precision, int n_res, int n_in, int n_out, bool harvest, Geometry* g);
int launch_recur_mfma(int precision, const RecurParams& p, hipStream_t stream);
// esn_harvest_cluster.hip
bool harvest_cluster_applies(int precision, const RecurParams& p);
size_t harvest_cluster_workspace_bytes(int n_pilots, int C, int n_wsets);
int launch_harvest_cluster(int precision, const RecurParams& p, int C, void* workspace, hipStream_t stream);
// esn_recur_skew16.hip
int launch_recur_skew16(int precision, const RecurParams& p, hipStream_t stream);
// esn_pack.hip
size_t packed_w_bytes(int precision, int n_res, int n_in, int n_out, const Geometry& g);
size_t packed_wout_bytes(int precision, int n_res, int n_in, int n_out, const Geometry& g);
size_t wout_big_offset(int precision, int n_out, const Geometry& g);
size_t f64_w_offset(int n_res, int n_in, int n_out);
size_t f64_wout_offset(int n_res, int n_in, int n_out);
int launch_pack_weights(int precision, const esn_shape_t* sh, const Geometry& g, const double* W,
                        const double* Win, const double* Wfb, void* packed, hipStream_t stream);
int launch_pack_readout(int precision, const esn_shape_t* sh, const Geometry& g, int n_groups,
                        const double* Wout, void* packed, hipStream_t stream);
// esn_solve.hip
size_t solve_work_doubles(int rows, int cols, int n_out);
int launch_readout_solve(const double* E, const double* D, int n_groups, int T, int transient,
                         int cols, int n_out, const double* t_scale, const double* t_shift,
                         double* W_out, int* status, void* workspace, hipStream_t stream);
int launch_readout_chol(const double* E, const float* E32, const double* D, int n_groups, int T, int transient,
                        int cols, int n_out, const double* t_scale, const double* t_shift,
                        double* W_out, int* status, hipStream_t stream);
size_t chol_big_work_doubles(int n);
int launch_readout_chol_big(const double* E, const float* E32, const double* D, int n_groups, int T, int transient,
                            int cols, int n_out, const double* t_scale, const double* t_shift,
                            double* W_out, int* status, void* workspace, hipStream_t stream);
// esn_gen.hip
int launch_gen_taps(const TapParams& tp, hipStream_t stream);
int launch_gen_frames(const FrameGenParams& fp, hipStream_t stream);
// esn_baseline.hip
int launch_channel_estimate(const ChanEstParams& cp, hipStream_t stream);
int launch_mmse_detect(const MmseParams& mp, hipStream_t stream);
int launch_taps_to_freq(const TapsFreqParams& tp, hipStream_t stream);
// esn_coded.hip
int launch_ldpc_encode(const LdpcEncodeParams& ep, hipStream_t stream);
int launch_qam_llr(const LlrParams& lp, hipStream_t stream);
int launch_ldpc_decode(const LdpcDecodeParams& dp, hipStream_t stream);
// esn_detect.hip
int launch_detect_count(const DetectParams& dp, hipStream_t stream);
}  // namespace esn

using namespace esn;

static thread_local char g_err[512] = "";

namespace esn {
static bool parse3(const char* v, int (&out)[3]) {
    int a, b, c;
    if (v && sscanf(v, "%d,%d,%d", &a, &b, &c) == 3 && a > 0 && b > 0 && c > 0) { out[0] = a; out[1] = b; out[2] = c; return true; }
    out[0] = out[1] = out[2] = 0;
    return false;
}
Knobs& knobs() {
    static Knobs k = [] {
        Knobs x;
        const char* v = getenv("ESN_SKEW");
        x.skew = (v && v[0] == '0') ? 0 : 1;
        parse3(getenv("ESN_MFMA_GEOM"), x.geom16);
        parse3(getenv("ESN_MFMA_GEOM_F32"), x.geom32);
        v = getenv("ESN_CHOL_SKIP");
        x.chol_skip = v ? atoi(v) : 0;
        v = getenv("ESN_F64_MFMA");
        x.f64_mfma = (v && v[0] == '0') ? 0 : 1;
        v = getenv("ESN_RS");
        x.rs = (v && v[0] == '1') ? 1 : 0;          // opt-in: measured slower than the skewed LDS-state kernel
        v = getenv("ESN_BIG_GEMM");
        x.big_gemm = (v && v[0] == '0') ? 0 : 1;
        v = getenv("ESN_CLUSTER");
        x.cluster = (v && v[0] == '0') ? 0 : 1;
        x.gen_ko = 0;
        v = getenv("ESN_HARVEST_GEMM");
        x.harvest_gemm = (v && v[0] == '1') ? 1 : 0;
        v = getenv("ESN_BIG_NT");
        x.big_nt = (v && v[0] == '4') ? 4 : 2;
        v = getenv("ESN_HCLUSTER");
        x.hcluster = (v && (v[0] == '0' || v[0] == '4' || v[0] == '8')) ? v[0] - '0' : 1;
        v = getenv("ESN_S16");
        x.s16 = (v && v[0] == '0') ? 0 : 1;
        v = getenv("ESN_BIG_PIPE");
        x.big_pipe = (v && v[0] == '0') ? 0 : 1;
        return x;
    }();
    return k;
}
}  // namespace esn

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace esn {
// for esn_host.hip: same error string, same conventions
int api_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace esn

static int hip_fail(int e, const char* what) {
    if (e == 0) return 0;
    if (e > 0) return fail(-1000 - e, "%s: HIP error %d (%s)", what, e, hipGetErrorString((hipError_t)e));
    return fail(-2, "%s: no kernel instance for this shape", what);
}

static bool check_shape(const esn_shape_t* s) {
    return s && s->n_res > 0 && s->n_in > 0 && s->n_out > 0 && s->n_wsets > 0;
}

// Geometry for a precision; for ESN_F64 only Bt (frames per tile) is meaningful.
static bool geometry_for(int precision, const esn_shape_t* s, Geometry* g, bool harvest = false) {
    memset(g, 0, sizeof(*g));
    if (precision == ESN_F64) {
        int fb = 8;
        while (fb > 1 && recur_f64_lds_bytes(fb, s->n_res, s->n_in, s->n_out) > 150 * 1024) fb >>= 1;
        if (recur_f64_lds_bytes(fb, s->n_res, s->n_in, s->n_out) > 160 * 1024) return false;
        // the matrix-pipe kernel where it fits (predict AND harvest must fit: one packed image serves both)
        Geometry gp, gh;
        memset(&gp, 0, sizeof(gp)); memset(&gh, 0, sizeof(gh));
        if (f64_mfma_geometry(s->n_res, s->n_in, s->n_out, false, &gp) &&
            f64_mfma_geometry(s->n_res, s->n_in, s->n_out, true, &gh))
            *g = harvest ? gh : gp;
        g->Bt = fb;
        return true;
    }
    if (precision < ESN_F64 || precision > ESN_BF16) return false;
    return mfma_geometry(precision, s->n_res, s->n_in, s->n_out, harvest, g);
}

#ifdef ESN_STAMPS
static unsigned long long* g_stamp_buf = nullptr;
extern "C" void esn_debug_set_stamp_buffer(void* dev) { g_stamp_buf = (unsigned long long*)dev; }
#define ESN_SET_STAMPS(p) (p).stamps = g_stamp_buf
#else
#define ESN_SET_STAMPS(p) (p).stamps = nullptr
#endif

extern "C" {

const char* esn_last_error(void) { return g_err; }

int esn_abi_version(void) { return 9; }

int esn_debug_set(const char* key, const char* value) {
    if (!key) return fail(-1, "esn_debug_set: null key");
    Knobs& k = knobs();
    if (!strcmp(key, "skew")) { k.skew = (value && value[0] == '0') ? 0 : 1; return 0; }
    if (!strcmp(key, "mfma_geom")) { parse3(value, k.geom16); return 0; }
    if (!strcmp(key, "mfma_geom_f32")) { parse3(value, k.geom32); return 0; }
    if (!strcmp(key, "chol_skip")) { k.chol_skip = value ? atoi(value) : 0; return 0; }
    if (!strcmp(key, "f64_mfma")) { k.f64_mfma = (value && value[0] == '0') ? 0 : 1; return 0; }
    if (!strcmp(key, "rs")) {
#ifdef ESN_WITH_RS
        k.rs = (value && value[0] == '1') ? 1 : 0; return 0;
#else
        if (value && value[0] == '1') return fail(-3, "esn_debug_set: the register-state kernel is not in this build (ESN_WITH_RS=1)");
        return 0;
#endif
    }
    if (!strcmp(key, "big_gemm")) { k.big_gemm = (value && value[0] == '0') ? 0 : 1; return 0; }
    if (!strcmp(key, "cluster")) { k.cluster = (value && value[0] == '0') ? 0 : 1; return 0; }
    if (!strcmp(key, "gen_ko")) { k.gen_ko = value ? atoi(value) : 0; return 0; }
    if (!strcmp(key, "harvest_gemm")) { k.harvest_gemm = (value && value[0] == '1') ? 1 : 0; return 0; }
    if (!strcmp(key, "big_nt")) { k.big_nt = (value && value[0] == '4') ? 4 : 2; return 0; }
    if (!strcmp(key, "hcluster")) {       // "0" off, "4" / "8" members per cluster (A/B), anything else: the default (2)
        k.hcluster = (value && (value[0] == '0' || value[0] == '4' || value[0] == '8')) ? value[0] - '0' : 1;
        return 0;
    }
    if (!strcmp(key, "s16")) { k.s16 = (value && value[0] == '0') ? 0 : 1; return 0; }
    if (!strcmp(key, "big_pipe")) { k.big_pipe = (value && value[0] == '0') ? 0 : 1; return 0; }
    return fail(-1, "esn_debug_set: unknown key '%s'", key);
}

int esn_device_info(int* cu_count, int* lds_bytes_per_cu, int* clock_khz, char* arch_name, int arch_name_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail((int)e, "hipGetDevice");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return hip_fail((int)e, "hipGetDeviceProperties");
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (clock_khz) *clock_khz = prop.clockRate;
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, prop.gcnArchName, arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    return 0;
}

int esn_tile_frames(int precision, const esn_shape_t* shape) {
    Geometry g;
    if (!check_shape(shape)) return fail(-1, "esn_tile_frames: invalid shape");
    if (!geometry_for(precision, shape, &g)) return fail(-2, "esn_tile_frames: unsupported shape/precision");
    return g.Bt;
}

size_t esn_packed_weights_bytes(int precision, const esn_shape_t* shape) {
    Geometry g;
    if (!check_shape(shape) || !geometry_for(precision, shape, &g)) return 0;
    return packed_w_bytes(precision, shape->n_res, shape->n_in, shape->n_out, g);
}

size_t esn_packed_readout_bytes(int precision, const esn_shape_t* shape) {
    Geometry g;
    if (!check_shape(shape) || !geometry_for(precision, shape, &g)) return 0;
    return packed_wout_bytes(precision, shape->n_res, shape->n_in, shape->n_out, g);
}

int esn_pack_weights(int precision, const esn_shape_t* shape, const double* W, const double* W_in,
                     const double* W_fb, void* packed, void* stream) {
    Geometry g;
    if (!check_shape(shape)) return fail(-1, "esn_pack_weights: invalid shape");
    if (!W || !W_in || !W_fb || !packed) return fail(-1, "esn_pack_weights: null pointer");
    if (!geometry_for(precision, shape, &g)) return fail(-2, "esn_pack_weights: unsupported shape/precision");
    return hip_fail(launch_pack_weights(precision, shape, g, W, W_in, W_fb, packed, (hipStream_t)stream),
                    "esn_pack_weights");
}

int esn_pack_readout(int precision, const esn_shape_t* shape, int n_groups, const double* W_out,
                     void* packed, void* stream) {
    Geometry g;
    if (!check_shape(shape) || n_groups <= 0) return fail(-1, "esn_pack_readout: invalid shape");
    if (!W_out || !packed) return fail(-1, "esn_pack_readout: null pointer");
    if (!geometry_for(precision, shape, &g)) return fail(-2, "esn_pack_readout: unsupported shape/precision");
    return hip_fail(launch_pack_readout(precision, shape, g, n_groups, W_out, packed, (hipStream_t)stream),
                    "esn_pack_readout");
}

// float64 kernel: every frame slot of a tile costs its share of FMAs whether it holds a frame or not,
// so a batch smaller than the tile (the 2-D drop-in is ONE sequence) gets a smaller tile
static void shrink_f64_tile(int precision, RecurParams& p) {
    if (precision != ESN_F64) return;
    const long long slots = (long long)p.n_groups * p.Fpad;
    while (p.g.Bt > 1 && p.g.Bt / 2 >= slots) p.g.Bt >>= 1;
}

// float64: the matrix-pipe kernel for batches (more slots than one vector-ALU tile holds), the
// vector-ALU kernel for the 2-D drop-in's single sequence and for shapes the MFMA tiling does not cover
static bool use_f64_mfma(int precision, const RecurParams& p, long long sequences) {
    return precision == ESN_F64 && p.g.m64 && sequences > 8 && knobs().f64_mfma;
}

static int fill_common(RecurParams& p, int precision, const esn_shape_t* shape, const char* who,
                       bool harvest = false) {
    memset(&p, 0, sizeof(p));
    if (!check_shape(shape)) return fail(-1, "%s: invalid shape", who);
    if (!geometry_for(precision, shape, &p.g, harvest)) return fail(-2, "%s: unsupported shape/precision", who);
    p.n_res = shape->n_res; p.n_in = shape->n_in; p.n_out = shape->n_out;
    p.teacher_forcing = shape->teacher_forcing ? 1 : 0;
    p.n_wsets = shape->n_wsets;
    p.wset_stride = packed_w_bytes(precision, p.n_res, p.n_in, p.n_out, p.g);
    p.wout_stride = packed_wout_bytes(precision, p.n_res, p.n_in, p.n_out, p.g);
    p.leak = (shape->leak_rate == 0.0) ? 1.0 : shape->leak_rate;
    if (!(p.leak > 0.0 && p.leak <= 1.0)) return fail(-1, "%s: leak_rate %g outside (0, 1]", who, shape->leak_rate);
    if (p.leak != 1.0 && precision != ESN_F64)
        return fail(-2, "%s: leak_rate != 1 is an extension of the float64 kernels (precision ESN_F64)", who);
    p.w16_off = (size_t)p.g.Mp * p.g.Kp * 2;
    p.wo16_off = wout_big_offset(precision, p.n_out, p.g);
    p.w64_off = f64_w_offset(p.n_res, p.n_in, p.n_out);
    p.wo64_off = f64_wout_offset(p.n_res, p.n_in, p.n_out);
    return 0;
}

int esn_predict_batch(int precision, const esn_shape_t* shape, const void* packed_w, const void* packed_wout,
                      const double* in_scale, const double* in_shift, const double* t_scale,
                      const double* t_shift, const double* U, int n_frames, int frames_per_group, int T_in,
                      int T, int transient, const double* x0, const double* y0, double noise, int noise_mode,
                      const double* noise_u, uint64_t seed, uint64_t group_offset, double* Y, void* workspace,
                      size_t workspace_bytes, void* stream) {
    RecurParams p;
    int rc = fill_common(p, precision, shape, "esn_predict_batch");
    if (rc) return rc;
    if (!packed_w || !packed_wout || !U || !Y) return fail(-1, "esn_predict_batch: null pointer");
    if (n_frames <= 0 || frames_per_group <= 0 || T <= 0 || T_in < 0 || T_in > T || transient < 0 || transient >= T)
        return fail(-1, "esn_predict_batch: invalid sizes (n_frames=%d F=%d T_in=%d T=%d transient=%d)",
                    n_frames, frames_per_group, T_in, T, transient);
    if (noise_mode == ESN_NOISE_TENSOR && !noise_u) return fail(-1, "esn_predict_batch: noise tensor missing");
    if (noise_mode < ESN_NOISE_NONE || noise_mode > ESN_NOISE_COUNTER) return fail(-1, "esn_predict_batch: bad noise mode");
    p.n_frames = n_frames;
    p.F = frames_per_group;
    p.n_groups = (n_frames + frames_per_group - 1) / frames_per_group;
    // slots per group: whole tiles when each group has its own weight set, else the readout's
    // 16-frame column granularity (MFMA kernels) or no padding at all (float64 kernel)
    const bool m64 = use_f64_mfma(precision, p, n_frames);
    const int tile = m64 ? p.g.Bt64 : p.g.Bt;
    // slots per group: the readout's 16-frame column granularity (MFMA kernels) or no padding at all (float64
    // vector-ALU kernel); with several weight sets the slot axis is set-major (esn_common.h: spw), so a tile
    // packs the groups of ONE set back to back instead of padding every group to a whole tile
    p.Fpad = (precision == ESN_F64 && !m64) ? p.F : round_up(p.F, 16);
    if (!m64) shrink_f64_tile(precision, p);
    const int tslots = m64 ? tile : p.g.Bt;
    if (p.n_wsets > 1) {
        const int gpw = (p.n_groups + p.n_wsets - 1) / p.n_wsets;
        p.spw = round_up(gpw * p.Fpad, tslots);
        p.n_tiles = p.n_wsets * (p.spw / tslots);
    } else {
        p.n_tiles = (int)(((long long)p.n_groups * p.Fpad + tslots - 1) / tslots);
    }
    p.T_in = T_in; p.S = T; p.in_row_off = 0; p.transient = transient; p.harvest = 0;
    p.packed_w = packed_w; p.packed_wout = packed_wout;
    p.in_scale = in_scale; p.in_shift = in_shift; p.t_scale = t_scale; p.t_shift = t_shift;
    p.U = U; p.x0 = x0; p.y0 = y0; p.noise_u = noise_u;
    p.noise = noise; p.noise_mode = (noise == 0.0) ? ESN_NOISE_NONE : noise_mode; p.seed = seed;
    p.frame_off = (uint32_t)(group_offset * (uint64_t)frames_per_group);
    p.wset_rot = (int)(group_offset % (uint64_t)p.n_wsets);
    if (((uintptr_t)Y & 15) != 0) return fail(-1, "esn_predict_batch: Y must be 16-byte aligned");
    p.Y = Y;
    ESN_SET_STAMPS(p);
    // ONE float64 sequence (the reference's own call pattern): the matrix resident in the LDS of a cluster of
    // workgroups that exchange the state through L2 every step (esn_recur_cluster.hip), when a workspace is lent
    if (workspace && knobs().cluster && cluster_applies(precision, p)) {
        const size_t need = cluster_workspace_bytes(p.n_res, p.n_in, p.n_out, false);
        if (workspace_bytes < need)
            return fail(-1, "esn_predict_batch: workspace holds %zu bytes, esn_predict_workspace_bytes says %zu",
                        workspace_bytes, need);
        return hip_fail(launch_recur_cluster(p, workspace, (hipStream_t)stream), "esn_predict_batch");
    }
    // large reservoirs: one GEMM launch per step when the caller lends a workspace (else the persistent kernel)
    if (workspace && knobs().big_gemm && p.g.big && big_path_applies(precision, p)) {
        const size_t need = big_workspace_bytes(big_slots(p), p.g.Mp, p.g.Kp);
        if (workspace_bytes < need)
            return fail(-1, "esn_predict_batch: workspace holds %zu bytes, esn_predict_workspace_bytes says %zu",
                        workspace_bytes, need);
        p.Fpad = round_up(p.F, 16);
        return hip_fail(launch_recur_big(precision, p, wout_big_offset(precision, p.n_out, p.g), workspace,
                                         (hipStream_t)stream), "esn_predict_batch");
    }
    // N_res 257..512, fp16/bf16: state in registers, one wave per SIMD (tiles of 128 slots like the skewed kernel)
#ifdef ESN_WITH_RS
    if (knobs().rs && rs_path_applies(precision, p) &&
        (size_t)n_frames * (T - transient) * p.n_out * 8 < 0x7fffffffu &&
        (size_t)p.n_groups * p.wout_stride < 0x7fffffffu)         // its buffer descriptors are 31-bit
        return hip_fail(launch_recur_rs(precision, p, wout_big_offset(precision, p.n_out, p.g), (hipStream_t)stream),
                        "esn_predict_batch");
#endif
    // N_res 257..512, fp16/bf16: the skewed schedule on 16x16x32 MFMAs (the chip holds a higher clock on that shape)
    if (p.g.s16 && p.g.skew && knobs().s16 && (precision == ESN_F16 || precision == ESN_BF16))
        return hip_fail(launch_recur_skew16(precision, p, (hipStream_t)stream), "esn_predict_batch");
    int e = m64 ? launch_recur_f64_mfma(p, (hipStream_t)stream)
            : (precision == ESN_F64) ? launch_recur_f64(p, (hipStream_t)stream)
                                     : launch_recur_mfma(precision, p, (hipStream_t)stream);
    return hip_fail(e, "esn_predict_batch");
}

size_t esn_predict_workspace_bytes(int precision, const esn_shape_t* shape, int n_frames, int frames_per_group) {
    RecurParams p;
    if (n_frames <= 0 || frames_per_group <= 0) return 0;
    if (fill_common(p, precision, shape, "esn_predict_workspace_bytes")) return 0;
    p.harvest = 0; p.F = frames_per_group; p.n_frames = n_frames;
    p.n_groups = (n_frames + frames_per_group - 1) / frames_per_group;
    if (knobs().cluster && cluster_applies(precision, p)) return cluster_workspace_bytes(p.n_res, p.n_in, p.n_out, false);
    if (!p.g.big || !big_path_applies(precision, p)) return 0;
    return big_workspace_bytes(big_slots(p), p.g.Mp, p.g.Kp);
}

size_t esn_harvest_workspace_bytes(int precision, const esn_shape_t* shape, int n_groups) {
    RecurParams p;
    if (n_groups <= 0) return 0;
    if (fill_common(p, precision, shape, "esn_harvest_workspace_bytes", true)) return 0;
    p.harvest = 1; p.n_groups = n_groups; p.n_frames = n_groups; p.F = 1;
    if (knobs().cluster && cluster_applies(precision, p)) return cluster_workspace_bytes(p.n_res, p.n_in, p.n_out, true);
    if (knobs().hcluster && harvest_cluster_applies(precision, p))
        return harvest_cluster_workspace_bytes(n_groups, knobs().hcluster == 1 ? 2 : knobs().hcluster, p.n_wsets);
    if (!big_harvest_applies(precision, p)) return 0;
    return big_harvest_workspace_bytes(n_groups, p.g.Kp);
}

static int harvest_common(int precision, const esn_shape_t* shape, const void* packed_w, const double* in_scale,
                      const double* in_shift, const double* t_scale, const double* t_shift, const double* U,
                      const double* D, int n_groups, int T, double noise, int noise_mode, const double* noise_u,
                      uint64_t seed, uint64_t group_offset, double* E, float* E32, void* workspace,
                      size_t workspace_bytes, void* stream) {
    RecurParams p;
    int rc = fill_common(p, precision, shape, "esn_harvest_batch", true);
    if (rc) return rc;
    if (!packed_w || !U || !D || (!E && !E32)) return fail(-1, "esn_harvest_batch: null pointer");
    if (E32 && precision == ESN_F64) return fail(-2, "esn_harvest_batch_f32: float32 states are an MFMA-kernel option (precision f32/f16/bf16)");
    if (n_groups <= 0 || T < 2) return fail(-1, "esn_harvest_batch: invalid sizes (n_groups=%d T=%d)", n_groups, T);
    if (noise_mode == ESN_NOISE_TENSOR && !noise_u) return fail(-1, "esn_harvest_batch: noise tensor missing");
    if (noise_mode < ESN_NOISE_NONE || noise_mode > ESN_NOISE_COUNTER) return fail(-1, "esn_harvest_batch: bad noise mode");
    p.n_frames = n_groups;
    p.n_groups = n_groups;
    p.F = 1;
    const bool m64 = use_f64_mfma(precision, p, n_groups);
    const int tile = m64 ? p.g.Bt64 : p.g.Bt;
    p.Fpad = 1;                             // one pilot per group; tiles span groups (of one weight set: spw)
    // (Tried: leaving slots empty so that 2048 pilots spread over 256 tiles of 8 instead of 64 tiles of 32 -- the
    //  harvest is bound by the AGGREGATE L2 weight stream, 4x the workgroups stream 4x the bytes: 1.14 -> 1.23 ms.)
    if (p.n_wsets == 1 && !m64) shrink_f64_tile(precision, p);
    const int tslots = m64 ? tile : p.g.Bt;
    if (p.n_wsets > 1) {
        const int gpw = (n_groups + p.n_wsets - 1) / p.n_wsets;
        p.spw = round_up(gpw, tslots);
        p.n_tiles = p.n_wsets * (p.spw / tslots);
    } else {
        p.n_tiles = (int)(((long long)n_groups + tslots - 1) / tslots);
    }
    p.T_in = T; p.S = T - 1; p.in_row_off = 1; p.transient = 0; p.harvest = 1;
    p.packed_w = packed_w;
    p.in_scale = in_scale; p.in_shift = in_shift; p.t_scale = t_scale; p.t_shift = t_shift;
    p.U = U; p.D = D; p.noise_u = noise_u;
    p.noise = noise; p.noise_mode = (noise == 0.0) ? ESN_NOISE_NONE : noise_mode; p.seed = seed;
    p.frame_off = (uint32_t)group_offset;
    p.wset_rot = (int)(group_offset % (uint64_t)p.n_wsets);
    if ((((uintptr_t)E) | ((uintptr_t)E32)) & 15) return fail(-1, "esn_harvest_batch: E must be 16-byte aligned");
    p.E = E; p.E32 = E32;
    ESN_SET_STAMPS(p);
    if (workspace && knobs().cluster && !E32 && cluster_applies(precision, p)) {          // one float64 sequence
        const size_t need = cluster_workspace_bytes(p.n_res, p.n_in, p.n_out, true);
        if (workspace_bytes < need)
            return fail(-1, "esn_harvest_batch: workspace holds %zu bytes, esn_harvest_workspace_bytes says %zu",
                        workspace_bytes, need);
        return hip_fail(launch_recur_cluster(p, workspace, (hipStream_t)stream), "esn_harvest_batch");
    }
    // 257..512 units, fp16/bf16, shared reservoir: clusters of eight workgroups with the matrix resident in LDS
    if (workspace && knobs().hcluster && harvest_cluster_applies(precision, p)) {
        const int hc_c = knobs().hcluster == 1 ? 2 : knobs().hcluster;           // members per cluster
        const size_t need = harvest_cluster_workspace_bytes(n_groups, hc_c, p.n_wsets);
        if (workspace_bytes < need)
            return fail(-1, "esn_harvest_batch: workspace holds %zu bytes, esn_harvest_workspace_bytes says %zu",
                        workspace_bytes, need);
        return hip_fail(launch_harvest_cluster(precision, p, hc_c, workspace, (hipStream_t)stream), "esn_harvest_batch");
    }
    // large reservoirs: one GEMM launch per step when the caller lends a workspace (else the persistent kernel)
    if (workspace && knobs().big_gemm && big_harvest_applies(precision, p)) {
        const size_t need = big_harvest_workspace_bytes(n_groups, p.g.Kp);
        if (workspace_bytes < need)
            return fail(-1, "esn_harvest_batch: workspace holds %zu bytes, esn_harvest_workspace_bytes says %zu",
                        workspace_bytes, need);
        return hip_fail(launch_harvest_big(precision, p, workspace, (hipStream_t)stream), "esn_harvest_batch");
    }
    int e = m64 ? launch_recur_f64_mfma(p, (hipStream_t)stream)
            : (precision == ESN_F64) ? launch_recur_f64(p, (hipStream_t)stream)
                                     : launch_recur_mfma(precision, p, (hipStream_t)stream);
    return hip_fail(e, "esn_harvest_batch");
}

int esn_harvest_batch(int precision, const esn_shape_t* shape, const void* packed_w, const double* in_scale,
                      const double* in_shift, const double* t_scale, const double* t_shift, const double* U,
                      const double* D, int n_groups, int T, double noise, int noise_mode, const double* noise_u,
                      uint64_t seed, uint64_t group_offset, double* E, void* workspace, size_t workspace_bytes,
                      void* stream) {
    if (!E) return fail(-1, "esn_harvest_batch: null pointer");
    return harvest_common(precision, shape, packed_w, in_scale, in_shift, t_scale, t_shift, U, D, n_groups, T, noise,
                          noise_mode, noise_u, seed, group_offset, E, nullptr, workspace, workspace_bytes, stream);
}

int esn_harvest_batch_f32(int precision, const esn_shape_t* shape, const void* packed_w, const double* in_scale,
                          const double* in_shift, const double* t_scale, const double* t_shift, const double* U,
                          const double* D, int n_groups, int T, double noise, int noise_mode,
                          const double* noise_u, uint64_t seed, uint64_t group_offset, float* E, void* workspace,
                          size_t workspace_bytes, void* stream) {
    if (!E) return fail(-1, "esn_harvest_batch_f32: null pointer");
    return harvest_common(precision, shape, packed_w, in_scale, in_shift, t_scale, t_shift, U, D, n_groups, T, noise,
                          noise_mode, noise_u, seed, group_offset, nullptr, E, workspace, workspace_bytes, stream);
}

size_t esn_readout_solve_workspace_bytes(int n_groups, int rows, int cols, int n_out) {
    if (n_groups <= 0 || rows <= 0 || cols <= 0 || n_out <= 0) return 0;
    return sizeof(double) * solve_work_doubles(rows, cols, n_out) * (size_t)n_groups;
}

int esn_readout_solve_batch(const double* E, const double* D, int n_groups, int T, int transient, int cols,
                            int n_out, const double* t_scale, const double* t_shift, double* W_out,
                            int* status, void* workspace, void* stream) {
    if (!E || !D || !W_out || !status || !workspace) return fail(-1, "esn_readout_solve_batch: null pointer");
    if (n_groups <= 0 || T <= 0 || transient < 0 || transient >= T || cols <= 0 || n_out <= 0)
        return fail(-1, "esn_readout_solve_batch: invalid sizes");
    return hip_fail(launch_readout_solve(E, D, n_groups, T, transient, cols, n_out, t_scale, t_shift, W_out,
                                         status, workspace, (hipStream_t)stream),
                    "esn_readout_solve_batch");
}

size_t esn_readout_chol_workspace_bytes(int n_groups, int rows, int cols) {
    if (n_groups <= 0 || rows <= 0 || cols <= 0) return 0;
    const int n = rows < cols ? rows : cols;
    if (n <= 128 || n > 512) return 0;             // LDS-resident kernel / not served
    return sizeof(double) * chol_big_work_doubles(n) * (size_t)n_groups;
}

static int chol_common(const char* who, const double* E, const float* E32, const double* D, int n_groups, int T,
                       int transient, int cols, int n_out, const double* t_scale, const double* t_shift, double* W_out,
                       int* status, void* workspace, size_t workspace_bytes, void* stream) {
    if ((!E && !E32) || !D || !W_out || !status) return fail(-1, "%s: null pointer", who);
    if (n_groups <= 0 || T <= 0 || transient < 0 || transient >= T || cols <= 0 || n_out <= 0)
        return fail(-1, "%s: invalid sizes", who);
    const int rows = T - transient, n = rows < cols ? rows : cols;
    if (n > 128) {            // Gram matrix and factor in the caller's workspace (esn_solve.hip, readout_chol_big_kernel)
        const size_t need = esn_readout_chol_workspace_bytes(n_groups, rows, cols);
        if (need == 0 || n_out > 8) return fail(-2, "%s: no kernel instance for this shape", who);
        if (!workspace || workspace_bytes < need)
            return fail(-1, "%s: workspace holds %zu bytes, esn_readout_chol_workspace_bytes says %zu", who,
                        workspace ? workspace_bytes : (size_t)0, need);
        if (((uintptr_t)(E ? (const void*)E : (const void*)E32) & 15) || ((uintptr_t)workspace & 15))
            return fail(-1, "%s: E and the workspace must be 16-byte aligned", who);
        return hip_fail(launch_readout_chol_big(E, E32, D, n_groups, T, transient, cols, n_out, t_scale, t_shift, W_out,
                                                status, workspace, (hipStream_t)stream), who);
    }
    return hip_fail(launch_readout_chol(E, E32, D, n_groups, T, transient, cols, n_out, t_scale, t_shift, W_out,
                                        status, (hipStream_t)stream), who);
}

int esn_readout_solve_chol_batch(const double* E, const double* D, int n_groups, int T, int transient, int cols,
                                 int n_out, const double* t_scale, const double* t_shift, double* W_out,
                                 int* status, void* workspace, size_t workspace_bytes, void* stream) {
    return chol_common("esn_readout_solve_chol_batch", E, nullptr, D, n_groups, T, transient, cols, n_out, t_scale,
                       t_shift, W_out, status, workspace, workspace_bytes, stream);
}

int esn_readout_solve_chol_batch_f32(const float* E, const double* D, int n_groups, int T, int transient, int cols,
                                     int n_out, const double* t_scale, const double* t_shift, double* W_out,
                                     int* status, void* workspace, size_t workspace_bytes, void* stream) {
    return chol_common("esn_readout_solve_chol_batch_f32", nullptr, E, D, n_groups, T, transient, cols, n_out, t_scale,
                       t_shift, W_out, status, workspace, workspace_bytes, stream);
}

int esn_detect_count(const double* Y, int n_frames, int frames_per_group, int n_sub, int n_t, int bits_per_sym,
                     const double* p_i, const uint8_t* tx_bits, long long* err_count, long long* bit_count,
                     double* X_hat, void* stream) {
    if (!Y || !p_i || !tx_bits || !err_count || !bit_count) return fail(-1, "esn_detect_count: null pointer");
    if (n_frames <= 0 || frames_per_group <= 0 || n_t <= 0) return fail(-1, "esn_detect_count: invalid sizes");
    int log2n = 0;
    while ((1 << log2n) < n_sub) ++log2n;
    if ((1 << log2n) != n_sub || n_sub < 2 || n_sub > 2048)
        return fail(-1, "esn_detect_count: N=%d must be a power of two in [2, 2048]", n_sub);
    if (bits_per_sym < 2 || bits_per_sym > 10 || (bits_per_sym & 1))
        return fail(-1, "esn_detect_count: bits_per_sym=%d must be even (square QAM)", bits_per_sym);
    DetectParams dp;
    dp.Y = Y; dp.n_frames = n_frames; dp.frames_per_group = frames_per_group; dp.n_sub = n_sub;
    dp.log2n = log2n; dp.n_t = n_t; dp.m = bits_per_sym; dp.p_i = p_i; dp.tx_bits = tx_bits;
    dp.err = err_count; dp.bits = bit_count; dp.X_hat = X_hat;
    return hip_fail(launch_detect_count(dp, (hipStream_t)stream), "esn_detect_count");
}

static const double kTdlbDelay[23] = {0.0000, 0.1072, 0.2155, 0.2095, 0.2870, 0.2986, 0.3752, 0.5055, 0.3681,
                                      0.3697, 0.5700, 0.5283, 1.1021, 1.2756, 1.5474, 1.7842, 2.0169, 2.8294,
                                      3.0219, 3.6187, 4.1067, 4.2790, 4.7834};
static const double kTdlbPowDb[23] = {0.0, -2.2, -4.0, -3.2, -9.8, -1.2, -3.4, -5.2, -7.6, -3.0, -8.9, -9.0,
                                      -4.8, -5.7, -7.5, -1.9, -7.6, -12.2, -9.8, -11.4, -14.9, -9.2, -11.3};

int esn_gen_taps(int kind, int n_blocks, int n_r, int n_t, int isi, double fs_hz, double ds_ns,
                 const double* gains_in, uint64_t seed, uint64_t link_offset, double* taps, void* stream) {
    if (!taps) return fail(-1, "esn_gen_taps: null pointer");
    if (kind < 0 || kind > 2 || n_blocks <= 0 || n_r <= 0 || n_t <= 0 || isi <= 0 || isi > 16)
        return fail(-1, "esn_gen_taps: invalid arguments (kind=%d isi=%d)", kind, isi);
    TapParams tp;
    memset(&tp, 0, sizeof(tp));
    tp.kind = kind; tp.n_links = n_blocks * n_r * n_t; tp.isi = isi;
    if (kind == 0) {
        tp.n_paths = 23;
        double sum = 0.0;
        for (int i = 0; i < 23; ++i) sum += pow(10.0, kTdlbPowDb[i] / 10.0);
        for (int i = 0; i < 23; ++i) {
            tp.path_sqrt_pow[i] = sqrt(pow(10.0, kTdlbPowDb[i] / 10.0) / sum);
            tp.path_delay_samples[i] = kTdlbDelay[i] * ds_ns * 1e-9 * fs_hz;
        }
    } else if (kind == 1) {
        tp.n_paths = isi;
        const int cp = isi - 1;
        const double tc = (cp / 9.0 > 1e-12) ? cp / 9.0 : 1e-12;
        double sum = 0.0;
        for (int i = 0; i < isi; ++i) sum += exp(-(double)i / tc);
        for (int i = 0; i < isi; ++i) tp.path_sqrt_pow[i] = sqrt(exp(-(double)i / tc) / sum);
    } else {
        tp.n_paths = 1;
        tp.path_sqrt_pow[0] = 1.0;
    }
    tp.gains_in = gains_in; tp.seed = seed; tp.link_offset = link_offset; tp.taps = taps;
    return hip_fail(launch_gen_taps(tp, (hipStream_t)stream), "esn_gen_taps");
}

int esn_gen_frames(int n_frames, int frames_per_block, int n_sub, int cp, int n_t, int n_r, int isi,
                   int bits_per_sym, int ls_pattern, const double* p_i, const double* a_clip, double no, const double* taps,
                   const uint8_t* bits_in, const double* noise_in, uint64_t seed, uint64_t frame_offset,
                   uint8_t* bits, double* x_cp, double* y_cp, void* stream) {
    if (!p_i || !a_clip || !taps || !bits || !y_cp) return fail(-1, "esn_gen_frames: null pointer");
    int log2n = 0;
    while ((1 << log2n) < n_sub) ++log2n;
    if ((1 << log2n) != n_sub || n_sub < 2 || n_sub > 2048)
        return fail(-1, "esn_gen_frames: N=%d must be a power of two in [2, 2048]", n_sub);
    if (bits_per_sym < 2 || bits_per_sym > 10 || (bits_per_sym & 1))
        return fail(-1, "esn_gen_frames: bits_per_sym=%d must be even (square QAM)", bits_per_sym);
    if (n_frames <= 0 || frames_per_block <= 0 || cp < 0 || cp >= n_sub || n_t <= 0 || n_r <= 0 || isi <= 0)
        return fail(-1, "esn_gen_frames: invalid sizes");
    FrameGenParams fp;
    fp.n_frames = n_frames; fp.frames_per_block = frames_per_block; fp.n_sub = n_sub; fp.log2n = log2n;
    fp.cp = cp; fp.n_t = n_t; fp.n_r = n_r; fp.isi = isi; fp.m = bits_per_sym;
    fp.p_i = p_i; fp.a_clip = a_clip; fp.no = no; fp.taps = taps; fp.bits_in = bits_in; fp.noise_in = noise_in;
    fp.seed = seed; fp.frame_offset = frame_offset; fp.bits = bits; fp.x_cp = x_cp; fp.y_cp = y_cp;
    fp.ls_pattern = ls_pattern ? 1 : 0;
    fp.ko = knobs().gen_ko;
    int e = launch_gen_frames(fp, (hipStream_t)stream);
    if (e == -1) return fail(-2, "esn_gen_frames: frame does not fit LDS");
    return hip_fail(e, "esn_gen_frames");
}

static int pow2_log(int n) { int l = 0; while ((1 << l) < n) ++l; return ((1 << l) == n) ? l : -1; }

int esn_channel_estimate(int n_blocks, int n_sub, int cp, int n_t, int n_r, int isi, int bits_per_sym,
                         const double* p_i, double no, const uint8_t* pilot_bits, const double* y_ls_cp,
                         int ls_only, double* H, void* stream) {
    if (!p_i || !pilot_bits || !y_ls_cp || !H) return fail(-1, "esn_channel_estimate: null pointer");
    const int l2 = pow2_log(n_sub);
    if (l2 < 1 || n_sub > 2048) return fail(-1, "esn_channel_estimate: N=%d must be a power of two in [2, 2048]", n_sub);
    if (n_blocks <= 0 || cp < 0 || cp >= n_sub || n_t <= 0 || n_r <= 0 || isi <= 0 || isi > 64 || n_sub / n_t < 2 ||
        bits_per_sym < 2 || (bits_per_sym & 1))
        return fail(-1, "esn_channel_estimate: invalid sizes");
    ChanEstParams c;
    c.n_blocks = n_blocks; c.n_sub = n_sub; c.log2n = l2; c.cp = cp; c.n_t = n_t; c.n_r = n_r; c.isi = isi;
    c.m = bits_per_sym; c.p_i = p_i; c.no = no; c.pilot_bits = pilot_bits; c.y_ls_cp = y_ls_cp; c.H = H;
    c.ls_only = ls_only ? 1 : 0;
    return hip_fail(launch_channel_estimate(c, (hipStream_t)stream), "esn_channel_estimate");
}

static int linear_detect(const char* who, int zf, int n_frames, int frames_per_group, int n_sub, int cp, int n_t, int n_r,
                         int bits_per_sym, const double* p_i, double no, const double* H, const double* y_cp,
                         const uint8_t* tx_bits, long long* err_count, long long* bit_count, double* X_hat,
                         void* stream) {
    if (!p_i || !H || !y_cp || !tx_bits || !err_count || !bit_count)
        return fail(-1, "%s: null pointer", who);
    const int l2 = pow2_log(n_sub);
    if (l2 < 1 || n_sub > 2048) return fail(-1, "%s: N=%d must be a power of two in [2, 2048]", who, n_sub);
    if (n_frames <= 0 || frames_per_group <= 0 || cp < 0 || cp >= n_sub || n_t <= 0 || n_r <= 0 ||
        bits_per_sym < 2 || (bits_per_sym & 1))
        return fail(-1, "%s: invalid sizes", who);
    MmseParams m;
    m.n_frames = n_frames; m.frames_per_group = frames_per_group; m.n_sub = n_sub; m.log2n = l2; m.cp = cp;
    m.n_t = n_t; m.n_r = n_r; m.m = bits_per_sym; m.zf = zf; m.p_i = p_i; m.no = no; m.H = H; m.y_cp = y_cp;
    m.tx_bits = tx_bits; m.err = err_count; m.bits = bit_count; m.X_hat = X_hat;
    int e = launch_mmse_detect(m, (hipStream_t)stream);
    if (e == -1) return fail(-2, "%s: needs n_t <= 4 and n_r * N * 16 bytes of LDS", who);
    return hip_fail(e, who);
}

int esn_mmse_detect_count(int n_frames, int frames_per_group, int n_sub, int cp, int n_t, int n_r, int bits_per_sym,
                          const double* p_i, double no, const double* H, const double* y_cp,
                          const uint8_t* tx_bits, long long* err_count, long long* bit_count, double* X_hat,
                          void* stream) {
    return linear_detect("esn_mmse_detect_count", 0, n_frames, frames_per_group, n_sub, cp, n_t, n_r, bits_per_sym, p_i,
                         no, H, y_cp, tx_bits, err_count, bit_count, X_hat, stream);
}

int esn_zf_detect_count(int n_frames, int frames_per_group, int n_sub, int cp, int n_t, int n_r, int bits_per_sym,
                        const double* p_i, const double* H, const double* y_cp, const uint8_t* tx_bits,
                        long long* err_count, long long* bit_count, double* X_hat, void* stream) {
    return linear_detect("esn_zf_detect_count", 1, n_frames, frames_per_group, n_sub, cp, n_t, n_r, bits_per_sym, p_i,
                         0.0, H, y_cp, tx_bits, err_count, bit_count, X_hat, stream);
}

int esn_taps_to_freq(int n_blocks, int n_sub, int n_t, int n_r, int isi, const double* taps, double* H, void* stream) {
    if (!taps || !H) return fail(-1, "esn_taps_to_freq: null pointer");
    if (n_blocks <= 0 || n_sub <= 0 || n_t <= 0 || n_r <= 0 || isi <= 0 || isi > n_sub)
        return fail(-1, "esn_taps_to_freq: invalid sizes");
    TapsFreqParams tp;
    tp.n_blocks = n_blocks; tp.n_sub = n_sub; tp.n_t = n_t; tp.n_r = n_r; tp.isi = isi; tp.taps = taps; tp.H = H;
    return hip_fail(launch_taps_to_freq(tp, (hipStream_t)stream), "esn_taps_to_freq");
}

int esn_ldpc_encode(int n_frames, int n_t, int k, int n, const uint8_t* P, const uint8_t* u, uint8_t* bits,
                    void* stream) {
    if (!P || !u || !bits) return fail(-1, "esn_ldpc_encode: null pointer");
    if (n_frames <= 0 || n_t <= 0 || k <= 0 || k >= n || k > 60000) return fail(-1, "esn_ldpc_encode: invalid sizes");
    LdpcEncodeParams ep;
    ep.n_frames = n_frames; ep.n_t = n_t; ep.k = k; ep.n = n; ep.P = P; ep.u = u; ep.bits = bits;
    return hip_fail(launch_ldpc_encode(ep, (hipStream_t)stream), "esn_ldpc_encode");
}

int esn_qam_llr(int n_frames, int n_sub, int n_t, int bits_per_sym, const double* X_hat, double* llr,
                double* sigma2, void* stream) {
    if (!X_hat || !llr) return fail(-1, "esn_qam_llr: null pointer");
    if (n_frames <= 0 || n_sub <= 0 || n_t <= 0 || bits_per_sym < 2 || bits_per_sym > 10 || (bits_per_sym & 1))
        return fail(-1, "esn_qam_llr: invalid sizes");
    LlrParams lp;
    lp.n_frames = n_frames; lp.n_sub = n_sub; lp.n_t = n_t; lp.m = bits_per_sym; lp.X_hat = X_hat; lp.llr = llr;
    lp.sigma2 = sigma2;
    return hip_fail(launch_qam_llr(lp, (hipStream_t)stream), "esn_qam_llr");
}

int esn_ldpc_decode_count(int n_cw, int n, int k, int m_checks, int n_edges, const int* chk_ptr, const int* edge_var,
                          const int* var_ptr, const int* var_edge, const double* y, double snr_db, int maxiter,
                          const uint8_t* u_true, int cw_per_group, uint8_t* x_out, long long* err_count,
                          long long* bit_count, void* stream) {
    if (!chk_ptr || !edge_var || !var_ptr || !var_edge || !y) return fail(-1, "esn_ldpc_decode_count: null pointer");
    if (u_true && (!err_count || !bit_count)) return fail(-1, "esn_ldpc_decode_count: counters missing");
    if (n_cw <= 0 || n <= 0 || k <= 0 || k > n || m_checks <= 0 || n_edges <= 0 || maxiter <= 0 || cw_per_group <= 0)
        return fail(-1, "esn_ldpc_decode_count: invalid sizes");
    LdpcDecodeParams dp;
    dp.n_cw = n_cw; dp.n = n; dp.k = k; dp.m_checks = m_checks; dp.n_edges = n_edges; dp.maxiter = maxiter;
    dp.cw_per_group = cw_per_group; dp.var = pow(10.0, -snr_db / 10.0);
    dp.chk_ptr = chk_ptr; dp.edge_var = edge_var; dp.var_ptr = var_ptr; dp.var_edge = var_edge; dp.y = y;
    dp.u_true = u_true; dp.x_out = x_out; dp.err = err_count; dp.bits = bit_count;
    int e = launch_ldpc_decode(dp, (hipStream_t)stream);
    if (e == -1) return fail(-2, "esn_ldpc_decode_count: graph does not fit LDS (16 B per edge)");
    return hip_fail(e, "esn_ldpc_decode_count");
}

}  // extern "C"
