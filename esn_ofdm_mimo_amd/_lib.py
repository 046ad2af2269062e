"""ctypes binding of libesn_hip.so (include/esn_hip.h).  Fails loudly: a missing
library or a missing GPU is an error, never a silent CPU path."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ESN_HIP_LIB") or os.path.join(_PKG, "libesn_hip.so")   # env: tuning builds only

F64, F32, F16, BF16 = 0, 1, 2, 3
PRECISIONS = {"f64": F64, "f32": F32, "f16": F16, "bf16": BF16}
NOISE_NONE, NOISE_TENSOR, NOISE_COUNTER = 0, 1, 2
ABI_VERSION = 9
MEM_DEVICE, MEM_HOST = 0, 1


class Shape(C.Structure):
    _fields_ = [("n_res", C.c_int), ("n_in", C.c_int), ("n_out", C.c_int),
                ("teacher_forcing", C.c_int), ("n_wsets", C.c_int),
                ("leak_rate", C.c_double)]       # extension: 0 or 1 = the reference's update (include/esn_hip.h)


class EsnHipError(RuntimeError):
    pass


_vp, _dp, _ip = C.c_void_p, C.c_void_p, C.c_void_p   # device pointers travel as integers
SIGNATURES = {
    "esn_last_error": (C.c_char_p, []),
    "esn_abi_version": (C.c_int, []),
    "esn_debug_set": (C.c_int, [C.c_char_p, C.c_char_p]),
    "esn_device_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.c_char_p, C.c_int]),
    "esn_tile_frames": (C.c_int, [C.c_int, C.POINTER(Shape)]),
    "esn_packed_weights_bytes": (C.c_size_t, [C.c_int, C.POINTER(Shape)]),
    "esn_packed_readout_bytes": (C.c_size_t, [C.c_int, C.POINTER(Shape)]),
    "esn_pack_weights": (C.c_int, [C.c_int, C.POINTER(Shape), _dp, _dp, _dp, _vp, _vp]),
    "esn_pack_readout": (C.c_int, [C.c_int, C.POINTER(Shape), C.c_int, _dp, _vp, _vp]),
    "esn_predict_batch": (C.c_int, [C.c_int, C.POINTER(Shape), _vp, _vp, _dp, _dp, _dp, _dp, _dp,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp,
                                    C.c_double, C.c_int, _dp, C.c_uint64, C.c_uint64, _dp, _vp, C.c_size_t, _vp]),
    "esn_predict_workspace_bytes": (C.c_size_t, [C.c_int, C.POINTER(Shape), C.c_int, C.c_int]),
    "esn_harvest_batch": (C.c_int, [C.c_int, C.POINTER(Shape), _vp, _dp, _dp, _dp, _dp, _dp, _dp,
                                    C.c_int, C.c_int, C.c_double, C.c_int, _dp, C.c_uint64, C.c_uint64, _dp,
                                    _vp, C.c_size_t, _vp]),
    "esn_harvest_workspace_bytes": (C.c_size_t, [C.c_int, C.POINTER(Shape), C.c_int]),
    "esn_harvest_batch_f32": (C.c_int, [C.c_int, C.POINTER(Shape), _vp, _dp, _dp, _dp, _dp, _dp, _dp,
                                        C.c_int, C.c_int, C.c_double, C.c_int, _dp, C.c_uint64, C.c_uint64, _vp,
                                        _vp, C.c_size_t, _vp]),
    "esn_readout_solve_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "esn_readout_solve_batch": (C.c_int, [_dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          _dp, _dp, _dp, _ip, _vp, _vp]),
    "esn_readout_chol_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "esn_readout_solve_chol_batch": (C.c_int, [_dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                               _dp, _dp, _dp, _ip, _vp, C.c_size_t, _vp]),
    "esn_readout_solve_chol_batch_f32": (C.c_int, [_vp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                   _dp, _dp, _dp, _ip, _vp, C.c_size_t, _vp]),
    "esn_gen_taps": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _dp,
                               C.c_uint64, C.c_uint64, _dp, _vp]),
    "esn_gen_frames": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 _dp, _dp, C.c_double, _dp, _vp, _dp, C.c_uint64, C.c_uint64, _vp, _dp, _dp, _vp]),
    "esn_channel_estimate": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp,
                                       C.c_double, _vp, _dp, C.c_int, _dp, _vp]),
    "esn_mmse_detect_count": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp,
                                        C.c_double, _dp, _dp, _vp, _vp, _vp, _dp, _vp]),
    "esn_zf_detect_count": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp,
                                      _dp, _dp, _vp, _vp, _vp, _dp, _vp]),
    "esn_taps_to_freq": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _vp]),
    "esn_ldpc_encode": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "esn_qam_llr": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _vp]),
    "esn_ldpc_decode_count": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, _ip, _ip, _dp,
                                        C.c_double, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "esn_detect_count": (C.c_int, [_dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _vp,
                                   _vp, _vp, _dp, _vp]),
}
# host-memory front ends: the arguments of esn_X behind a leading esn_mem_kind (include/esn_hip.h)
for _name in ("esn_pack_weights", "esn_pack_readout", "esn_predict_batch", "esn_harvest_batch",
              "esn_readout_solve_batch", "esn_detect_count"):
    SIGNATURES[_name + "_mem"] = (C.c_int, [C.c_int] + SIGNATURES[_name][1])
SIGNATURES["esn_device_alloc"] = (C.c_void_p, [C.c_size_t])
SIGNATURES["esn_device_free"] = (C.c_int, [C.c_void_p])

_lib = None


def load():
    """Load (once) and type the library.  Raises EsnHipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EsnHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the ESN hot path.")
    # torch first: libesn_hip.so must resolve libamdhip64 to the copy torch has already loaded -- loaded before torch it
    # binds the system ROCm runtime instead, the process ends up with two HIP runtimes and every launch of this
    # library fails with hipErrorNoDevice (seen with build() + smoke() in one process)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.esn_abi_version() != ABI_VERSION:
        raise EsnHipError(f"libesn_hip.so ABI {lib.esn_abi_version()} != binding {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().esn_last_error().decode(errors="replace")
        raise EsnHipError(f"{what} failed ({rc}): {msg}")


def ptr(t):
    """data_ptr of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise EsnHipError("no HIP device visible: the ESN hot path runs on the GPU only "
                          "(no CPU fallback is provided)")
    return torch


def stream_handle():
    import torch
    return torch.cuda.current_stream().cuda_stream


def debug_set(key, value):
    """Tuning knob of the library (benchmarks / A-B tests): see esn_debug_set in include/esn_hip.h."""
    check(load().esn_debug_set(key.encode(), None if value is None else str(value).encode()), "esn_debug_set")


def device_info():
    lib = load()
    cu, lds, clk = C.c_int(), C.c_int(), C.c_int()
    name = C.create_string_buffer(64)
    check(lib.esn_device_info(C.byref(cu), C.byref(lds), C.byref(clk), name, 64), "esn_device_info")
    return dict(cu_count=cu.value, lds_bytes_per_cu=lds.value, clock_khz=clk.value,
                arch=name.value.decode())
