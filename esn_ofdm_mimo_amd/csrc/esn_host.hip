// Host-memory front ends of the four SURVEY 8(b) entry points (+ the two pack calls they need):
// `esn_*_mem(mem_kind, ...)`.  ESN_MEM_DEVICE forwards to the device-pointer entry point unchanged.
// ESN_MEM_HOST treats every ARRAY argument as caller-owned host memory (the reference's own calling
// convention: C-contiguous float64 NumPy arrays, pyESN.py:154,218): the call stages each array in device
// memory from the stream-ordered pool (hipMallocAsync), copies in, runs the device entry point on the same
// stream, copies the results out, frees, and returns after the stream has drained -- the host arrays are
// complete on return.  Packed weight / readout images and workspaces stay device memory in both kinds
// (esn_device_alloc / esn_device_free give a caller without a HIP toolchain the means to hold them).
// This is a convenience for callers that live on the host; the measured path is the device one (the
// PCIe-inclusive rate is in DESIGN.md 1).
#include <stdarg.h>
#include <stdio.h>
#include <vector>
#include "esn_common.h"

namespace esn {
int api_fail(int code, const char* fmt, ...);      // esn_api.hip: sets esn_last_error()
}
using namespace esn;

namespace {

// Device copies of the host arrays of one call; everything is released (stream-ordered) by the destructor.
struct Staging {
    hipStream_t stream;
    int err = 0;
    struct Out { void* host; void* dev; size_t bytes; };
    std::vector<void*> owned;
    std::vector<Out> outs;
    explicit Staging(void* s) : stream((hipStream_t)s) {}
    void* alloc(size_t bytes) {
        void* d = nullptr;
        if (err) return nullptr;
        hipError_t e = hipMallocAsync(&d, bytes ? bytes : 16, stream);
        if (e != hipSuccess) { err = (int)e; return nullptr; }
        owned.push_back(d);
        return d;
    }
    // host array -> device copy (NULL stays NULL)
    template <typename T> const T* in(const T* host, size_t count) {
        if (!host || err) return nullptr;
        void* d = alloc(count * sizeof(T));
        if (!d) return nullptr;
        hipError_t e = hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) err = (int)e;
        return static_cast<const T*>(d);
    }
    // device buffer whose contents go back to `host` in finish(); `preload` copies the host values in first
    // (accumulators such as the error counters)
    template <typename T> T* out(T* host, size_t count, bool preload = false) {
        if (!host || err) return nullptr;
        void* d = alloc(count * sizeof(T));
        if (!d) return nullptr;
        if (preload) {
            hipError_t e = hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, stream);
            if (e != hipSuccess) err = (int)e;
        }
        outs.push_back({host, d, count * sizeof(T)});
        return static_cast<T*>(d);
    }
    // copy the outputs back (only when the device call succeeded) and wait for the stream
    int finish(int rc, const char* what) {
        if (!err && rc == 0)
            for (const Out& o : outs) {
                hipError_t e = hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, stream);
                if (e != hipSuccess) { err = (int)e; break; }
            }
        for (void* d : owned) hipFreeAsync(d, stream);
        owned.clear();
        hipError_t e = hipStreamSynchronize(stream);
        if (!err && e != hipSuccess) err = (int)e;
        if (rc) return rc;                                  // the device entry point has set the error text
        if (err) return api_fail(-1000 - err, "%s: HIP error %d (%s) while staging host arrays", what, err,
                                 hipGetErrorString((hipError_t)err));
        return 0;
    }
    ~Staging() { for (void* d : owned) hipFreeAsync(d, stream); }
};

bool kind_ok(int k) { return k == ESN_MEM_DEVICE || k == ESN_MEM_HOST; }
size_t groups_of(int n_frames, int frames_per_group) {
    return frames_per_group > 0 ? ((size_t)n_frames + frames_per_group - 1) / frames_per_group : 0;
}

}  // namespace

extern "C" {

void* esn_device_alloc(size_t bytes) {
    void* d = nullptr;
    hipError_t e = hipMalloc(&d, bytes ? bytes : 16);
    if (e != hipSuccess) {
        api_fail(-1000 - (int)e, "esn_device_alloc(%zu): HIP error %d (%s)", bytes, (int)e, hipGetErrorString(e));
        return nullptr;
    }
    return d;
}

int esn_device_free(void* p) {
    if (!p) return 0;
    hipError_t e = hipFree(p);
    return e == hipSuccess ? 0 : api_fail(-1000 - (int)e, "esn_device_free: HIP error %d (%s)", (int)e, hipGetErrorString(e));
}

int esn_pack_weights_mem(int mem_kind, int precision, const esn_shape_t* shape, const double* W, const double* W_in,
                         const double* W_fb, void* packed, void* stream) {
    if (!kind_ok(mem_kind)) return api_fail(-1, "esn_pack_weights_mem: unknown memory kind %d", mem_kind);
    if (mem_kind == ESN_MEM_DEVICE) return esn_pack_weights(precision, shape, W, W_in, W_fb, packed, stream);
    if (!shape || shape->n_res <= 0 || shape->n_in <= 0 || shape->n_out <= 0 || shape->n_wsets <= 0 || !W || !W_in)
        return api_fail(-1, "esn_pack_weights_mem: invalid shape or null pointer");
    const size_t ws = (size_t)shape->n_wsets, n = (size_t)shape->n_res;
    Staging st(stream);
    const double* dW = st.in(W, ws * n * n);
    const double* dWin = st.in(W_in, ws * n * shape->n_in);
    const double* dWfb = st.in(W_fb, ws * n * shape->n_out);
    int rc = st.err ? 0 : esn_pack_weights(precision, shape, dW, dWin, dWfb, packed, stream);
    return st.finish(rc, "esn_pack_weights_mem");
}

int esn_pack_readout_mem(int mem_kind, int precision, const esn_shape_t* shape, int n_groups, const double* W_out,
                         void* packed, void* stream) {
    if (!kind_ok(mem_kind)) return api_fail(-1, "esn_pack_readout_mem: unknown memory kind %d", mem_kind);
    if (mem_kind == ESN_MEM_DEVICE) return esn_pack_readout(precision, shape, n_groups, W_out, packed, stream);
    if (!shape || shape->n_res <= 0 || shape->n_in <= 0 || shape->n_out <= 0 || n_groups <= 0 || !W_out)
        return api_fail(-1, "esn_pack_readout_mem: invalid shape or null pointer");
    Staging st(stream);
    const double* dWo = st.in(W_out, (size_t)n_groups * shape->n_out * ((size_t)shape->n_res + shape->n_in));
    int rc = st.err ? 0 : esn_pack_readout(precision, shape, n_groups, dWo, packed, stream);
    return st.finish(rc, "esn_pack_readout_mem");
}

int esn_predict_batch_mem(int mem_kind, int precision, const esn_shape_t* shape, const void* packed_w,
                          const void* packed_wout, const double* in_scale, const double* in_shift,
                          const double* t_scale, const double* t_shift, const double* U, int n_frames,
                          int frames_per_group, int T_in, int T, int transient, const double* x0, const double* y0,
                          double noise, int noise_mode, const double* noise_u, uint64_t seed, uint64_t group_offset,
                          double* Y, void* workspace, size_t workspace_bytes, void* stream) {
    if (!kind_ok(mem_kind)) return api_fail(-1, "esn_predict_batch_mem: unknown memory kind %d", mem_kind);
    if (mem_kind == ESN_MEM_DEVICE)
        return esn_predict_batch(precision, shape, packed_w, packed_wout, in_scale, in_shift, t_scale, t_shift, U,
                                 n_frames, frames_per_group, T_in, T, transient, x0, y0, noise, noise_mode, noise_u,
                                 seed, group_offset, Y, workspace, workspace_bytes, stream);
    if (!shape || shape->n_res <= 0 || shape->n_in <= 0 || shape->n_out <= 0 || !U || !Y || n_frames <= 0 ||
        frames_per_group <= 0 || T_in <= 0 || T < T_in || transient < 0 || transient >= T)
        return api_fail(-1, "esn_predict_batch_mem: invalid sizes or null pointer");
    const size_t G = groups_of(n_frames, frames_per_group), B = (size_t)n_frames;
    const size_t n_res = shape->n_res, n_in = shape->n_in, n_out = shape->n_out;
    Staging st(stream);
    const double* d_is = st.in(in_scale, G * n_in);
    const double* d_ih = st.in(in_shift, G * n_in);
    const double* d_ts = st.in(t_scale, G * n_out);
    const double* d_th = st.in(t_shift, G * n_out);
    const double* d_U = st.in(U, B * T_in * n_in);
    const double* d_x0 = st.in(x0, G * n_res);
    const double* d_y0 = st.in(y0, G * n_out);
    const double* d_nz = (noise_mode == ESN_NOISE_TENSOR) ? st.in(noise_u, B * T * n_res) : nullptr;
    double* d_Y = st.out(Y, B * (size_t)(T - transient) * n_out);
    int rc = st.err ? 0
                    : esn_predict_batch(precision, shape, packed_w, packed_wout, d_is, d_ih, d_ts, d_th, d_U, n_frames,
                                        frames_per_group, T_in, T, transient, d_x0, d_y0, noise, noise_mode, d_nz, seed,
                                        group_offset, d_Y, workspace, workspace_bytes, stream);
    return st.finish(rc, "esn_predict_batch_mem");
}

int esn_harvest_batch_mem(int mem_kind, int precision, const esn_shape_t* shape, const void* packed_w,
                          const double* in_scale, const double* in_shift, const double* t_scale,
                          const double* t_shift, const double* U, const double* D, int n_groups, int T, double noise,
                          int noise_mode, const double* noise_u, uint64_t seed, uint64_t group_offset, double* E,
                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!kind_ok(mem_kind)) return api_fail(-1, "esn_harvest_batch_mem: unknown memory kind %d", mem_kind);
    if (mem_kind == ESN_MEM_DEVICE)
        return esn_harvest_batch(precision, shape, packed_w, in_scale, in_shift, t_scale, t_shift, U, D, n_groups, T,
                                 noise, noise_mode, noise_u, seed, group_offset, E, workspace, workspace_bytes, stream);
    if (!shape || shape->n_res <= 0 || shape->n_in <= 0 || shape->n_out <= 0 || !U || !D || !E || n_groups <= 0 || T < 2)
        return api_fail(-1, "esn_harvest_batch_mem: invalid sizes or null pointer");
    const size_t G = (size_t)n_groups, n_res = shape->n_res, n_in = shape->n_in, n_out = shape->n_out;
    Staging st(stream);
    const double* d_is = st.in(in_scale, G * n_in);
    const double* d_ih = st.in(in_shift, G * n_in);
    const double* d_ts = st.in(t_scale, G * n_out);
    const double* d_th = st.in(t_shift, G * n_out);
    const double* d_U = st.in(U, G * T * n_in);
    const double* d_D = st.in(D, G * T * n_out);
    const double* d_nz = (noise_mode == ESN_NOISE_TENSOR) ? st.in(noise_u, G * (size_t)(T - 1) * n_res) : nullptr;
    double* d_E = st.out(E, G * T * (n_res + n_in));
    int rc = st.err ? 0
                    : esn_harvest_batch(precision, shape, packed_w, d_is, d_ih, d_ts, d_th, d_U, d_D, n_groups, T, noise,
                                        noise_mode, d_nz, seed, group_offset, d_E, workspace, workspace_bytes, stream);
    return st.finish(rc, "esn_harvest_batch_mem");
}

int esn_readout_solve_batch_mem(int mem_kind, const double* E, const double* D, int n_groups, int T, int transient,
                                int cols, int n_out, const double* t_scale, const double* t_shift, double* W_out,
                                int* status, void* workspace, void* stream) {
    if (!kind_ok(mem_kind)) return api_fail(-1, "esn_readout_solve_batch_mem: unknown memory kind %d", mem_kind);
    if (mem_kind == ESN_MEM_DEVICE)
        return esn_readout_solve_batch(E, D, n_groups, T, transient, cols, n_out, t_scale, t_shift, W_out, status,
                                       workspace, stream);
    if (!E || !D || !W_out || !status || n_groups <= 0 || T <= 0 || transient < 0 || transient >= T || cols <= 0 || n_out <= 0)
        return api_fail(-1, "esn_readout_solve_batch_mem: invalid sizes or null pointer");
    const size_t G = (size_t)n_groups;
    Staging st(stream);
    const double* d_E = st.in(E, G * T * cols);
    const double* d_D = st.in(D, G * T * n_out);
    const double* d_ts = st.in(t_scale, G * n_out);
    const double* d_th = st.in(t_shift, G * n_out);
    double* d_W = st.out(W_out, G * n_out * cols);
    int* d_st = st.out(status, G);
    // the QR workspace is scratch: taken from the pool when the caller (who may have no device allocator) passes NULL
    void* ws = workspace ? workspace : st.alloc(esn_readout_solve_workspace_bytes(n_groups, T - transient, cols, n_out));
    int rc = st.err ? 0
                    : esn_readout_solve_batch(d_E, d_D, n_groups, T, transient, cols, n_out, d_ts, d_th, d_W, d_st, ws,
                                              stream);
    return st.finish(rc, "esn_readout_solve_batch_mem");
}

int esn_detect_count_mem(int mem_kind, const double* Y, int n_frames, int frames_per_group, int n_sub, int n_t,
                         int bits_per_sym, const double* p_i, const uint8_t* tx_bits, long long* err_count,
                         long long* bit_count, double* X_hat, void* stream) {
    if (!kind_ok(mem_kind)) return api_fail(-1, "esn_detect_count_mem: unknown memory kind %d", mem_kind);
    if (mem_kind == ESN_MEM_DEVICE)
        return esn_detect_count(Y, n_frames, frames_per_group, n_sub, n_t, bits_per_sym, p_i, tx_bits, err_count,
                                bit_count, X_hat, stream);
    if (!Y || !p_i || !tx_bits || !err_count || !bit_count || n_frames <= 0 || frames_per_group <= 0 || n_sub <= 0 ||
        n_t <= 0 || bits_per_sym <= 0)
        return api_fail(-1, "esn_detect_count_mem: invalid sizes or null pointer");
    const size_t B = (size_t)n_frames, G = groups_of(n_frames, frames_per_group), N = (size_t)n_sub;
    Staging st(stream);
    const double* d_Y = st.in(Y, B * N * 2 * n_t);
    const double* d_pi = st.in(p_i, G);
    const uint8_t* d_tx = st.in(tx_bits, B * N * bits_per_sym * n_t);
    long long* d_err = st.out(err_count, G, true);          // the kernel ADDS to the counters
    long long* d_bits = st.out(bit_count, G, true);
    double* d_X = st.out(X_hat, B * N * n_t * 2);
    int rc = st.err ? 0
                    : esn_detect_count(d_Y, n_frames, frames_per_group, n_sub, n_t, bits_per_sym, d_pi, d_tx, d_err,
                                       d_bits, d_X, stream);
    return st.finish(rc, "esn_detect_count_mem");
}

}  // extern "C"
