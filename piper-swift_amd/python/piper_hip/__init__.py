"""ctypes shim over lib/libpiper_hip.so (include/piper_hip.h).

Host-side mirror of the reference's operator interface for this path: `HipBackend` exposes the
`MetalBackend` per-op methods (Sources/PiperMetal/Execution/MetalBackend.swift) with the same names
and argument meaning — buffer + shape in, (buffer, shape) out, optional stream standing in for
`commandBuffer:` — and raises `ExecutionError` subclasses where the Swift code `throws`.

There is NO CPU fallback here: if the shared library is missing, or no gfx950 device is present,
construction fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libpiper_hip.so"))
LIB_NORT_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libpiper_hip_nort.so"))  # linked with -no-hip-rt

# ---- errors: ExecutionError (CPUBackend.swift:3-17) ----


class ExecutionError(RuntimeError):
    code = None


class ShapeMismatch(ExecutionError):
    code = -1


class TypeMismatch(ExecutionError):
    code = -2


class UnsupportedOp(ExecutionError):
    code = -3


class DeviceUnavailable(ExecutionError):  # ExecutionError.metalUnavailable
    code = -4


class AllocationFailed(ExecutionError):
    code = -5


class LaunchFailed(ExecutionError):
    code = -6


class InvalidArgument(ExecutionError):
    code = -7


_ERRORS = {c.code: c for c in (ShapeMismatch, TypeMismatch, UnsupportedOp, DeviceUnavailable, AllocationFailed,
                               LaunchFailed, InvalidArgument)}

RELU, LEAKYRELU, TANH, SIGMOID, EXP, NEG, SQRT, SOFTPLUS, CEIL, ERF = range(10)
ADD, SUB, MUL, DIV, POW = range(5)

c_f32p = C.POINTER(C.c_float)
c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)
c_vp = C.c_void_p


class Conv1dParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("stride", "dilation", "pad_l", "pad_r", "groups")]


class ConvTranspose1dParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("stride", "dilation", "pad_l", "pad_r", "output_padding", "groups")]


class VoiceConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_vocab", "hidden", "n_heads", "n_layers", "ffn", "ffn_kernel", "window",
                                         "inter", "n_flows", "wn_layers", "wn_kernel", "up_initial", "n_ups")] + [
        ("up_rates", C.c_int32 * 4), ("up_kernels", C.c_int32 * 4), ("resblock_type", C.c_int32), ("n_rb", C.c_int32),
        ("rb_kernels", C.c_int32 * 3), ("rb_n_dil", C.c_int32), ("rb_dilations", (C.c_int32 * 3) * 3),
        ("sample_rate", C.c_int32), ("dp_present", C.c_int32), ("dp_kernel", C.c_int32), ("dp_dds_layers", C.c_int32),
        ("dp_n_flows", C.c_int32), ("dp_bins", C.c_int32), ("dp_tail_bound", C.c_float)]

    @property
    def hop(self):
        h = 1
        for i in range(self.n_ups):
            h *= self.up_rates[i]
        return h


class TensorInfo(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("kind", C.c_int32), ("rank", C.c_int32), ("shape", C.c_int64 * 3),
                ("fan_in", C.c_int64), ("offset", C.c_uint64), ("count", C.c_uint64)]


class OnnxTensorInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("data_type", C.c_int32), ("rank", C.c_int32), ("dims", C.c_int64 * 8),
                ("count", C.c_int64)]


class PiperJsonInfo(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("num_symbols", C.c_int32), ("num_speakers", C.c_int32),
                ("noise_scale", C.c_float), ("length_scale", C.c_float), ("noise_w", C.c_float)]


class Utterance(C.Structure):
    _fields_ = [("phoneme_ids", c_i64p), ("t", C.c_int32), ("durations", c_i32p), ("noise", c_f32p),
                ("noise_scale", C.c_float), ("noise_mode", C.c_int32), ("seed", C.c_uint32), ("length_scale", C.c_float),
                ("noise_w", C.c_float), ("dp_noise", c_f32p)]


NOISE_MODES = {"injected": 0, "device": 1}


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("avg_us", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


# every symbol include/piper_hip.h declares: (restype, argtypes)
_PROTOS = {
    "piper_hip_last_error": (C.c_char_p, []),
    "piper_hip_abi_version": (C.c_int, []),
    "piper_hip_config_string": (C.c_int, [C.c_char_p, C.c_size_t]),
    "piper_hip_device_count": (C.c_int, []),
    "piper_hip_create": (C.c_int, [C.c_int, C.POINTER(c_vp)]),
    "piper_hip_destroy": (None, [c_vp]),
    "piper_hip_alloc": (C.c_int, [c_vp, C.c_size_t, C.POINTER(c_vp)]),
    "piper_hip_free": (C.c_int, [c_vp, c_vp]),
    "piper_hip_upload_f32": (C.c_int, [c_vp, c_f32p, C.c_size_t, C.POINTER(c_vp)]),
    "piper_hip_upload_i64": (C.c_int, [c_vp, c_i64p, C.c_size_t, C.POINTER(c_vp)]),
    "piper_hip_download_f32": (C.c_int, [c_vp, c_vp, c_f32p, C.c_size_t]),
    "piper_hip_stream_create": (C.c_int, [c_vp, C.POINTER(c_vp)]),
    "piper_hip_stream_destroy": (C.c_int, [c_vp, c_vp]),
    "piper_hip_stream_sync": (C.c_int, [c_vp, c_vp]),
    "piper_hip_timer_begin": (C.c_int, [c_vp, c_vp]),
    "piper_hip_timer_end": (C.c_int, [c_vp, c_vp, C.POINTER(C.c_double)]),
    "piper_hip_conv1d_f32": (C.c_int, [c_vp, c_vp, c_i64p, c_vp, c_i64p, c_vp, C.POINTER(Conv1dParams),
                                       C.POINTER(c_vp), c_i64p, c_vp]),
    "piper_hip_convtranspose1d_f32": (C.c_int, [c_vp, c_vp, c_i64p, c_vp, c_i64p, c_vp,
                                                C.POINTER(ConvTranspose1dParams), C.POINTER(c_vp), c_i64p, c_vp]),
    "piper_hip_conv1d_bf16": (C.c_int, [c_vp, c_vp, c_i64p, c_vp, c_i64p, c_vp, C.POINTER(Conv1dParams),
                                        C.POINTER(c_vp), c_i64p, c_vp]),
    "piper_hip_convtranspose1d_bf16": (C.c_int, [c_vp, c_vp, c_i64p, c_vp, c_i64p, c_vp,
                                                 C.POINTER(ConvTranspose1dParams), C.POINTER(c_vp), c_i64p, c_vp]),
    "piper_hip_matmul_f32": (C.c_int, [c_vp, c_vp, c_i64p, c_vp, c_i64p, C.c_int, C.POINTER(c_vp), c_i64p, c_vp]),
    "piper_hip_softmax_lastdim_f32": (C.c_int, [c_vp, c_vp, c_i64p, C.c_int, C.POINTER(c_vp), c_vp]),
    "piper_hip_unary_f32": (C.c_int, [c_vp, C.c_int, c_vp, C.c_size_t, C.c_float, C.POINTER(c_vp), c_vp]),
    "piper_hip_binary_broadcast_f32": (C.c_int, [c_vp, C.c_int, c_vp, c_i64p, C.c_int, c_vp, c_i64p, C.c_int,
                                                 C.POINTER(c_vp), c_i64p, C.POINTER(C.c_int), c_vp]),
    "piper_hip_pad_constant_f32": (C.c_int, [c_vp, c_vp, c_i64p, C.c_int, c_i64p, C.c_float, C.POINTER(c_vp), c_i64p,
                                             c_vp]),
    "piper_hip_slice_f32": (C.c_int, [c_vp, c_vp, c_i64p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64,
                                      C.POINTER(c_vp), c_i64p, c_vp]),
    "piper_hip_transpose_f32": (C.c_int, [c_vp, c_vp, c_i64p, C.c_int, c_i32p, C.POINTER(c_vp), c_i64p, c_vp]),
    "piper_hip_concat2_axis1_f32": (C.c_int, [c_vp, c_vp, c_i64p, c_vp, c_i64p, C.POINTER(c_vp), c_i64p, c_vp]),
    "piper_hip_split2_axis1_f32": (C.c_int, [c_vp, c_vp, c_i64p, C.c_int64, C.POINTER(c_vp), C.POINTER(c_vp), c_vp]),
    "piper_hip_expand_f32": (C.c_int, [c_vp, c_vp, c_i64p, c_i64p, C.c_int, C.POINTER(c_vp), c_vp]),
    "piper_hip_reduce_mean_lastdim_f32": (C.c_int, [c_vp, c_vp, c_i64p, C.c_int, C.POINTER(c_vp), c_vp]),
    "piper_hip_random_normal_like_f32": (C.c_int, [c_vp, C.c_size_t, C.c_uint64, C.POINTER(c_vp), c_vp]),
    "piper_hip_random_draws_u32": (C.c_int, [c_vp, C.c_size_t, C.c_uint64, C.POINTER(c_vp), c_vp]),
    "piper_hip_rel_attention_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_int64,
                                              C.c_int64, C.c_int64, C.POINTER(c_vp), c_vp]),
    "piper_hip_attention_block_f32": (C.c_int, [c_vp] * 11 + [C.c_int64] * 5 + [C.c_float, C.POINTER(c_vp), c_vp]),
    "piper_hip_add_layernorm_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_int64, C.c_float,
                                              C.POINTER(c_vp), c_vp]),
    "piper_hip_wavenet_layer_f32": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int64, C.c_int64, C.c_int64,
                                              C.c_int64, C.c_int64, C.c_int, C.POINTER(c_vp), C.POINTER(c_vp), c_vp]),
    "piper_hip_hifigan_resblock_f32": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, c_i32p,
                                                 C.c_int, C.POINTER(c_vp), C.POINTER(c_vp), C.c_float, C.POINTER(c_vp),
                                                 c_vp]),
    "piper_hip_voice_config_preset": (C.c_int, [C.c_int, C.POINTER(VoiceConfig)]),
    "piper_hip_voice_blob_floats": (C.c_int, [C.POINTER(VoiceConfig), C.POINTER(C.c_size_t)]),
    "piper_hip_voice_blob_layout": (C.c_int, [C.POINTER(VoiceConfig), C.POINTER(TensorInfo), C.c_int,
                                              C.POINTER(C.c_int)]),
    "piper_hip_voice_synthetic_blob": (C.c_int, [C.POINTER(VoiceConfig), C.c_uint64, c_f32p, C.c_size_t]),
    "piper_hip_voice_create": (C.c_int, [c_vp, C.POINTER(VoiceConfig), c_vp, C.c_int, C.POINTER(c_vp)]),
    "piper_hip_onnx_open": (C.c_int, [C.c_char_p, C.POINTER(c_vp)]),
    "piper_hip_onnx_open_memory": (C.c_int, [c_vp, C.c_size_t, C.POINTER(c_vp)]),
    "piper_hip_onnx_close": (None, [c_vp]),
    "piper_hip_onnx_counts": (C.c_int, [c_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int)]),
    "piper_hip_onnx_initializer": (C.c_int, [c_vp, C.c_int, C.POINTER(OnnxTensorInfo)]),
    "piper_hip_onnx_find": (C.c_int, [c_vp, C.c_char_p]),
    "piper_hip_onnx_read_f32": (C.c_int, [c_vp, C.c_int, c_f32p, C.c_size_t]),
    "piper_hip_onnx_infer_config": (C.c_int, [c_vp, C.POINTER(VoiceConfig)]),
    "piper_hip_onnx_build_blob": (C.c_int, [c_vp, C.POINTER(VoiceConfig), c_f32p, C.c_size_t]),
    "piper_hip_onnx_build_blob_unchecked": (C.c_int, [c_vp, C.POINTER(VoiceConfig), c_f32p, C.c_size_t]),
    "piper_hip_onnx_verify_graph": (C.c_int, [c_vp, C.POINTER(VoiceConfig)]),
    "piper_hip_voice_last_build_breakdown": (C.c_int, [c_vp, C.POINTER(C.c_double)]),
    "piper_hip_voice_set_plan_cache": (C.c_int, [c_vp, C.c_int, C.c_size_t]),
    "piper_hip_piper_json": (C.c_int, [C.c_char_p, C.POINTER(PiperJsonInfo)]),
    "piper_hip_voice_check_json": (C.c_int, [C.POINTER(VoiceConfig), C.POINTER(PiperJsonInfo)]),
    "piper_hip_pcm16_from_f32": (C.c_int, [c_f32p, C.c_size_t, C.POINTER(C.c_int16)]),
    "piper_hip_wav_write": (C.c_int, [C.c_char_p, c_f32p, C.c_size_t, C.c_int32]),
    "piper_hip_voice_receptive_field": (C.c_int, [c_vp]),
    "piper_hip_voice_stream_begin": (C.c_int, [c_vp, C.POINTER(Utterance), C.c_int, C.c_int]),
    "piper_hip_voice_stream_next": (C.c_int, [c_vp, C.c_int, c_f32p, C.c_int64, C.POINTER(C.c_int64)]),
    "piper_hip_memory_stats": (C.c_int, [c_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "piper_hip_memory_trim": (C.c_int, [c_vp]),
    "piper_hip_memory_reserve": (C.c_int, [c_vp, C.c_size_t]),
    "piper_hip_voice_set_precision": (C.c_int, [c_vp, C.c_int]),
    "piper_hip_voice_precision": (C.c_int, [c_vp]),
    "piper_hip_voice_destroy": (None, [c_vp]),
    "piper_hip_voice_num_samples": (C.c_int64, [c_vp, C.POINTER(Utterance)]),
    "piper_hip_host_alloc": (C.c_int, [c_vp, C.c_size_t, C.POINTER(C.c_void_p)]),
    "piper_hip_host_free": (C.c_int, [c_vp, C.c_void_p]),
    "piper_hip_comm_unique_id": (C.c_int, [C.c_void_p]),
    "piper_hip_comm_create": (C.c_int, [c_vp, C.c_void_p, C.c_int, C.c_int, C.POINTER(c_vp)]),
    "piper_hip_comm_destroy": (None, [c_vp]),
    "piper_hip_comm_rank": (C.c_int, [c_vp]),
    "piper_hip_comm_world": (C.c_int, [c_vp]),
    "piper_hip_comm_broadcast_f32": (C.c_int, [c_vp, C.c_void_p, C.c_size_t, C.c_int]),
    "piper_hip_comm_max_f64": (C.c_int, [c_vp, C.POINTER(C.c_double)]),
    "piper_hip_comm_barrier": (C.c_int, [c_vp]),
    "piper_hip_voice_prepare": (C.c_int, [c_vp, C.POINTER(Utterance), C.c_int]),
    "piper_hip_voice_predict_durations": (C.c_int, [c_vp, C.POINTER(Utterance), C.c_int, c_i32p, c_f32p, C.c_int]),
    "piper_hip_voice_prepared_samples": (C.c_int, [c_vp, C.c_int, c_i64p, C.c_int, c_i64p]),
    "piper_hip_voice_durations": (C.c_int, [c_vp, C.c_int, c_i32p, C.c_int, C.POINTER(C.c_int)]),
    "piper_hip_voice_prepare_batch": (C.c_int, [c_vp, C.POINTER(Utterance), C.c_int, C.c_int]),
    "piper_hip_voice_prepare_batch_bounded": (C.c_int, [c_vp, C.POINTER(Utterance), C.c_int, C.c_int, C.c_int]),
    "piper_hip_voice_batch_size": (C.c_int, [c_vp, C.c_int]),
    "piper_hip_voice_plan_info": (C.c_int, [c_vp, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                            C.POINTER(C.c_size_t)]),
    "piper_hip_voice_launch": (C.c_int, [c_vp, C.c_int]),
    "piper_hip_voice_collect": (C.c_int, [c_vp, C.c_int, c_f32p, C.c_int64]),
    "piper_hip_voice_synthesize": (C.c_int, [c_vp, C.POINTER(Utterance), c_f32p, C.c_int64, C.POINTER(C.c_int64)]),
    "piper_hip_voice_tap": (C.c_int, [c_vp, C.c_int, C.c_char_p, c_f32p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "piper_hip_voice_last_gpu_ms": (C.c_int, [c_vp, C.c_int, C.POINTER(C.c_double)]),
    "piper_hip_voice_slot_stream": (c_vp, [c_vp, C.c_int]),
    "piper_hip_voice_profile": (C.c_int, [c_vp, C.c_int, C.c_int, C.POINTER(KernelStat), C.c_int, C.POINTER(C.c_int)]),
    "piper_hip_voice_time_subset": (C.c_int, [c_vp, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int),
                                              C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None
_runtime = os.environ.get("PIPER_HIP_RUNTIME", "system")  # "system" (ROCm install) or "torch" (PyTorch's bundled copy)


def set_runtime(which):
    """Choose the HIP runtime the library runs on; must be called before the first load_library().

    A process may hold only ONE libamdhip64 that touches the GPU.  "system" loads libpiper_hip.so, which depends on the
    ROCm install's runtime.  "torch" is for processes that also drive the GPU through PyTorch (torch.cuda, RCCL): PyTorch's
    bundled libamdhip64 is made global and libpiper_hip_nort.so (linked with -no-hip-rt) binds to it."""
    global _runtime
    if which not in ("system", "torch"):
        raise ValueError("runtime must be 'system' or 'torch'")
    if _lib is not None and which != _runtime:
        raise RuntimeError("piper_hip: the HIP runtime is already bound to %r" % _runtime)
    _runtime = which


def _library_path():
    if _runtime == "torch":
        import torch
        rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if not os.path.exists(rt):
            raise DeviceUnavailable("PyTorch's HIP runtime not found at %s" % rt)
        C.CDLL(rt, mode=C.RTLD_GLOBAL)  # already loaded by torch; this only promotes it to the global symbol scope
        return LIB_NORT_PATH
    return LIB_PATH


def load_library(path=None):
    """dlopen the C-ABI library and bind every declared symbol (raises if one is missing)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or _library_path()
    if not os.path.exists(p):
        raise DeviceUnavailable(f"{p} not found: build it with `make -C piper-swift_amd` "
                                "(python __graft_entry__.py build). There is no CPU fallback.")
    lib = C.CDLL(p)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def exported_symbols():
    return sorted(_PROTOS)


def _check(rc):
    if rc != 0:
        msg = load_library().piper_hip_last_error().decode("utf-8", "replace")
        raise _ERRORS.get(rc, ExecutionError)(msg or f"piper_hip error {rc}")


def _i64(a):
    arr = (C.c_int64 * len(a))(*[int(v) for v in a])
    return arr


def device_count():
    return load_library().piper_hip_device_count()


class DeviceBuffer:
    """An MTLBuffer stand-in: device pointer + element count, owned by a backend's pool."""

    def __init__(self, backend, ptr, count, owned=True):
        self.backend, self.ptr, self.count, self.owned = backend, ptr, int(count), owned

    def free(self):
        if self.owned and self.ptr:
            _check(self.backend.lib.piper_hip_free(self.backend.ctx, self.ptr))
        self.ptr = None


def _ptr(b):
    if b is None:
        return None
    return b.ptr if isinstance(b, DeviceBuffer) else b


class HipBackend:
    """MetalBackend's hot-path surface (MetalBackend.swift:8-3427) over the C-ABI."""

    def __init__(self, device=0):
        self.lib = load_library()
        ctx = c_vp()
        _check(self.lib.piper_hip_create(device, C.byref(ctx)))
        self.ctx = ctx

    def close(self):
        if self.ctx:
            self.lib.piper_hip_destroy(self.ctx)
            self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- buffers (MetalBackend.swift:34-39, 963-993)
    def memory_stats(self):
        r, l = C.c_size_t(), C.c_size_t()
        _check(self.lib.piper_hip_memory_stats(self.ctx, C.byref(r), C.byref(l)))
        return dict(reserved=r.value, live=l.value)

    def memory_reserve(self, nbytes):
        """One slab of device memory up front; later allocations are carved from it (no first-touch stalls under requests)."""
        _check(self.lib.piper_hip_memory_reserve(self.ctx, int(nbytes)))

    def memory_trim(self):
        _check(self.lib.piper_hip_memory_trim(self.ctx))

    def allocateBuffer(self, length):
        p = c_vp()
        _check(self.lib.piper_hip_alloc(self.ctx, length, C.byref(p)))
        return DeviceBuffer(self, p.value, length // 4)

    def uploadFloat32(self, data):
        a = np.ascontiguousarray(data, dtype=np.float32)
        p = c_vp()
        _check(self.lib.piper_hip_upload_f32(self.ctx, a.ctypes.data_as(c_f32p), a.size, C.byref(p)))
        return DeviceBuffer(self, p.value, a.size)

    def downloadFloat32(self, buf, count=None):
        n = buf.count if count is None else int(count)
        out = np.empty(n, np.float32)
        _check(self.lib.piper_hip_download_f32(self.ctx, _ptr(buf), out.ctypes.data_as(c_f32p), n))
        return out

    def makeCommandBuffer(self):
        s = c_vp()
        _check(self.lib.piper_hip_stream_create(self.ctx, C.byref(s)))
        return s

    def flush(self, stream):
        _check(self.lib.piper_hip_stream_sync(self.ctx, stream))

    def _out(self, p, shape):
        return DeviceBuffer(self, p.value, int(np.prod(shape)) if len(shape) else 1), [int(s) for s in shape]

    # -- ops
    def conv1dF32(self, input, inputShape, weight, weightShape, bias, stride=1, dilation=1, padL=0, padR=0, groups=1,
                  commandBuffer=None):
        if len(inputShape) != 3:
            raise ShapeMismatch("conv1dF32 input must be [N,C,L]")
        if len(weightShape) != 3:
            raise ShapeMismatch("conv1dF32 weight must be [C_out,C_in,K]")
        prm = Conv1dParams(stride, dilation, padL, padR, groups)
        p, osh = c_vp(), (C.c_int64 * 3)()
        _check(self.lib.piper_hip_conv1d_f32(self.ctx, _ptr(input), _i64(inputShape), _ptr(weight), _i64(weightShape),
                                             _ptr(bias), C.byref(prm), C.byref(p), osh, commandBuffer))
        return self._out(p, list(osh))

    def convTranspose1dF32(self, input, inputShape, weight, weightShape, bias, stride=1, dilation=1, padL=0, padR=0,
                           outputPadding=0, groups=1, commandBuffer=None):
        if len(inputShape) != 3:
            raise ShapeMismatch("convTranspose1dF32 input must be [N,C,L]")
        if len(weightShape) != 3:
            raise ShapeMismatch("convTranspose1dF32 weight must be [C_in,C_out_per_group,K]")
        prm = ConvTranspose1dParams(stride, dilation, padL, padR, outputPadding, groups)
        p, osh = c_vp(), (C.c_int64 * 3)()
        _check(self.lib.piper_hip_convtranspose1d_f32(self.ctx, _ptr(input), _i64(inputShape), _ptr(weight),
                                                      _i64(weightShape), _ptr(bias), C.byref(prm), C.byref(p), osh,
                                                      commandBuffer))
        return self._out(p, list(osh))

    def conv1dBF16(self, input, inputShape, weight, weightShape, bias, stride=1, dilation=1, padL=0, padR=0, groups=1,
                   commandBuffer=None):
        """conv1dF32's contract with bf16 operands / fp32 accumulation (build extension; UnsupportedOp outside its geometry)."""
        if len(inputShape) != 3:
            raise ShapeMismatch("conv1dBF16 input must be [N,C,L]")
        if len(weightShape) != 3:
            raise ShapeMismatch("conv1dBF16 weight must be [C_out,C_in,K]")
        prm = Conv1dParams(stride, dilation, padL, padR, groups)
        p, osh = c_vp(), (C.c_int64 * 3)()
        _check(self.lib.piper_hip_conv1d_bf16(self.ctx, _ptr(input), _i64(inputShape), _ptr(weight), _i64(weightShape),
                                              _ptr(bias), C.byref(prm), C.byref(p), osh, commandBuffer))
        return self._out(p, list(osh))

    def convTranspose1dBF16(self, input, inputShape, weight, weightShape, bias, stride=1, dilation=1, padL=0, padR=0,
                            outputPadding=0, groups=1, commandBuffer=None):
        if len(inputShape) != 3:
            raise ShapeMismatch("convTranspose1dBF16 input must be [N,C,L]")
        if len(weightShape) != 3:
            raise ShapeMismatch("convTranspose1dBF16 weight must be [C_in,C_out,K]")
        prm = ConvTranspose1dParams(stride, dilation, padL, padR, outputPadding, groups)
        p, osh = c_vp(), (C.c_int64 * 3)()
        _check(self.lib.piper_hip_convtranspose1d_bf16(self.ctx, _ptr(input), _i64(inputShape), _ptr(weight),
                                                       _i64(weightShape), _ptr(bias), C.byref(prm), C.byref(p), osh,
                                                       commandBuffer))
        return self._out(p, list(osh))

    def matmulF32(self, a, aShape, b, bShape, commandBuffer=None):
        if len(aShape) < 2 or len(bShape) < 2:
            raise ShapeMismatch(f"matmulF32 requires rank>=2 (got {aShape} x {bShape})")
        if len(aShape) != len(bShape):
            raise ShapeMismatch(f"matmulF32 rank mismatch (got {len(aShape)} vs {len(bShape)})")
        r = len(aShape)
        p, osh = c_vp(), (C.c_int64 * r)()
        _check(self.lib.piper_hip_matmul_f32(self.ctx, _ptr(a), _i64(aShape), _ptr(b), _i64(bShape), r, C.byref(p), osh,
                                             commandBuffer))
        return self._out(p, list(osh))

    def softmaxLastDimF32(self, input, shape, commandBuffer=None):
        if len(shape) < 1:
            raise ShapeMismatch("softmaxLastDimF32 requires non-empty last dim")
        p = c_vp()
        _check(self.lib.piper_hip_softmax_lastdim_f32(self.ctx, _ptr(input), _i64(shape), len(shape), C.byref(p),
                                                      commandBuffer))
        return self._out(p, shape)

    def unaryF32(self, op, input, count, alpha=0.0, commandBuffer=None):
        p = c_vp()
        _check(self.lib.piper_hip_unary_f32(self.ctx, op, _ptr(input), count, alpha, C.byref(p), commandBuffer))
        return DeviceBuffer(self, p.value, count)

    def reluF32(self, input, count, commandBuffer=None):
        return self.unaryF32(RELU, input, count, 0.0, commandBuffer)

    def leakyReluF32(self, input, count, alpha=0.01, commandBuffer=None):
        return self.unaryF32(LEAKYRELU, input, count, alpha, commandBuffer)

    def tanhF32(self, input, count, commandBuffer=None):
        return self.unaryF32(TANH, input, count, 0.0, commandBuffer)

    def sigmoidF32(self, input, count, commandBuffer=None):
        return self.unaryF32(SIGMOID, input, count, 0.0, commandBuffer)

    def binaryBroadcastF32(self, op, a, aShape, b, bShape, commandBuffer=None):
        r = max(len(aShape), len(bShape))
        p, osh, orank = c_vp(), (C.c_int64 * max(r, 1))(), C.c_int()
        _check(self.lib.piper_hip_binary_broadcast_f32(self.ctx, op, _ptr(a), _i64(aShape), len(aShape), _ptr(b),
                                                       _i64(bShape), len(bShape), C.byref(p), osh, C.byref(orank),
                                                       commandBuffer))
        return self._out(p, list(osh)[:orank.value])

    def addF32(self, a, aShape, b, bShape, commandBuffer=None):
        return self.binaryBroadcastF32(ADD, a, aShape, b, bShape, commandBuffer)

    def subF32(self, a, aShape, b, bShape, commandBuffer=None):
        return self.binaryBroadcastF32(SUB, a, aShape, b, bShape, commandBuffer)

    def mulF32(self, a, aShape, b, bShape, commandBuffer=None):
        return self.binaryBroadcastF32(MUL, a, aShape, b, bShape, commandBuffer)

    def divF32(self, a, aShape, b, bShape, commandBuffer=None):
        return self.binaryBroadcastF32(DIV, a, aShape, b, bShape, commandBuffer)

    def padConstantF32(self, input, shape, pads, value=0.0, commandBuffer=None):
        r = len(shape)
        p, osh = c_vp(), (C.c_int64 * r)()
        _check(self.lib.piper_hip_pad_constant_f32(self.ctx, _ptr(input), _i64(shape), r, _i64(pads), value, C.byref(p),
                                                   osh, commandBuffer))
        return self._out(p, list(osh))

    def sliceF32(self, input, shape, axis, start, end, step=1, commandBuffer=None):
        r = len(shape)
        p, osh = c_vp(), (C.c_int64 * r)()
        _check(self.lib.piper_hip_slice_f32(self.ctx, _ptr(input), _i64(shape), r, axis, start, end, step, C.byref(p), osh,
                                            commandBuffer))
        return self._out(p, list(osh))

    def transposeF32(self, input, shape, perm, commandBuffer=None):
        r = len(shape)
        if len(perm) != r:
            raise ShapeMismatch(f"Transpose perm rank mismatch: perm={perm} shape={shape}")
        p, osh = c_vp(), (C.c_int64 * r)()
        pm = (C.c_int32 * r)(*perm)
        _check(self.lib.piper_hip_transpose_f32(self.ctx, _ptr(input), _i64(shape), r, pm, C.byref(p), osh, commandBuffer))
        return self._out(p, list(osh))

    def concat2Axis1F32(self, a, aShape, b, bShape, commandBuffer=None):
        p, osh = c_vp(), (C.c_int64 * 3)()
        _check(self.lib.piper_hip_concat2_axis1_f32(self.ctx, _ptr(a), _i64(aShape), _ptr(b), _i64(bShape), C.byref(p),
                                                    osh, commandBuffer))
        return self._out(p, list(osh))

    def split2Axis1F32(self, input, shape, c0, commandBuffer=None):
        p0, p1 = c_vp(), c_vp()
        _check(self.lib.piper_hip_split2_axis1_f32(self.ctx, _ptr(input), _i64(shape), c0, C.byref(p0), C.byref(p1),
                                                   commandBuffer))
        n, c, l = shape
        return self._out(p0, [n, c0, l]), self._out(p1, [n, c - c0, l])

    def expandF32(self, input, inShape, outShape, commandBuffer=None):
        p = c_vp()
        _check(self.lib.piper_hip_expand_f32(self.ctx, _ptr(input), _i64(inShape), _i64(outShape), len(outShape),
                                             C.byref(p), commandBuffer))
        return self._out(p, outShape)[0]

    def reduceMeanLastDimF32(self, input, shape, commandBuffer=None):
        p = c_vp()
        _check(self.lib.piper_hip_reduce_mean_lastdim_f32(self.ctx, _ptr(input), _i64(shape), len(shape), C.byref(p),
                                                          commandBuffer))
        return self._out(p, list(shape[:-1]))

    def randomNormalLike(self, shape, seed=1234, commandBuffer=None):
        """MetalBackend.randomNormalLike(shape:seed:) — device buffer of prod(shape) floats."""
        n = int(np.prod(shape)) if len(shape) else 1
        p = c_vp()
        _check(self.lib.piper_hip_random_normal_like_f32(self.ctx, n, int(seed), C.byref(p), commandBuffer))
        return DeviceBuffer(self, p.value, n)

    def randomDraws(self, count, seed=1234):
        """The raw (u0, u1) 32-bit draws per element, as a host uint32 array [count, 2]."""
        p = c_vp()
        _check(self.lib.piper_hip_random_draws_u32(self.ctx, int(count), int(seed), C.byref(p), None))
        out = np.empty(2 * int(count), np.float32)
        _check(self.lib.piper_hip_download_f32(self.ctx, p, out.ctypes.data_as(c_f32p), 2 * int(count)))
        _check(self.lib.piper_hip_free(self.ctx, p))
        return out.view(np.uint32).reshape(-1, 2)

    # -- fused
    def relAttentionF32(self, q, k, v, embRelK, embRelV, n, heads, headDim, t, window, commandBuffer=None):
        p = c_vp()
        _check(self.lib.piper_hip_rel_attention_f32(self.ctx, _ptr(q), _ptr(k), _ptr(v), _ptr(embRelK), _ptr(embRelV), n,
                                                    heads, headDim, t, window, C.byref(p), commandBuffer))
        return self._out(p, [n, heads * headDim, t])

    def attentionBlockF32(self, q, k, v, embRelK, embRelV, wO, bO, x, gamma, beta, n, heads, headDim, t, window, eps=1e-5,
                          commandBuffer=None):
        """LN(x + conv_o(rel_attention(q, k, v)))·gamma + beta in one launch."""
        p = c_vp()
        _check(self.lib.piper_hip_attention_block_f32(self.ctx, _ptr(q), _ptr(k), _ptr(v), _ptr(embRelK), _ptr(embRelV), _ptr(wO), _ptr(bO),
                                                      _ptr(x), _ptr(gamma), _ptr(beta), n, heads, headDim, t, window, eps, C.byref(p),
                                                      commandBuffer))
        return self._out(p, [n, heads * headDim, t])

    def addLayerNormF32(self, x, y, gamma, beta, n, c, t, eps=1e-5, commandBuffer=None):
        p = c_vp()
        _check(self.lib.piper_hip_add_layernorm_f32(self.ctx, _ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), n, c, t, eps,
                                                    C.byref(p), commandBuffer))
        return self._out(p, [n, c, t])

    def wavenetLayerF32(self, x, skipIn, wIn, bIn, wRs, bRs, n, c, t, k, dilation, last, commandBuffer=None):
        px, ps = c_vp(), c_vp()
        _check(self.lib.piper_hip_wavenet_layer_f32(self.ctx, _ptr(x), _ptr(skipIn), _ptr(wIn), _ptr(bIn), _ptr(wRs),
                                                    _ptr(bRs), n, c, t, k, dilation, int(bool(last)), C.byref(px),
                                                    C.byref(ps), commandBuffer))
        xo = None if last else self._out(px, [n, c, t])[0]
        return xo, self._out(ps, [n, c, t])[0]

    def hifiganResblockF32(self, type, x, n, c, t, k, dilations, weights, biases, slope=0.1, commandBuffer=None):
        nd = len(dilations)
        d = (C.c_int32 * nd)(*dilations)
        w = (c_vp * len(weights))(*[_ptr(b) for b in weights])
        b = (c_vp * len(biases))(*[_ptr(x_) for x_ in biases])
        p = c_vp()
        _check(self.lib.piper_hip_hifigan_resblock_f32(self.ctx, type, _ptr(x), n, c, t, k, d, nd, w, b, slope, C.byref(p),
                                                       commandBuffer))
        return self._out(p, [n, c, t])[0]


# ---------------------------------------------------------------- voice level (PiperMetalRuntime)

def voice_config(quality="medium"):
    cfg = VoiceConfig()
    _check(load_library().piper_hip_voice_config_preset({"medium": 0, "high": 1}[quality], C.byref(cfg)))
    return cfg


def blob_floats(cfg):
    n = C.c_size_t()
    _check(load_library().piper_hip_voice_blob_floats(C.byref(cfg), C.byref(n)))
    return n.value


def blob_layout(cfg):
    lib = load_library()
    n = C.c_int()
    _check(lib.piper_hip_voice_blob_layout(C.byref(cfg), None, 0, C.byref(n)))
    arr = (TensorInfo * n.value)()
    _check(lib.piper_hip_voice_blob_layout(C.byref(cfg), arr, n.value, C.byref(n)))
    return [dict(name=t.name.decode(), kind=t.kind, shape=[int(t.shape[i]) for i in range(t.rank)], fan_in=int(t.fan_in),
                 offset=int(t.offset), count=int(t.count)) for t in arr]


def synthetic_blob(cfg, seed=1234):
    """Host-only (no GPU): the synthetic voice of SURVEY.md §8d."""
    n = blob_floats(cfg)
    blob = np.empty(n, np.float32)
    _check(load_library().piper_hip_voice_synthetic_blob(C.byref(cfg), seed, blob.ctypes.data_as(c_f32p), n))
    return blob


COMM_ID_BYTES = 128


def comm_unique_id():
    """128 opaque bytes (ncclUniqueId) made by ONE rank; every rank of the world passes the same bytes to Comm()."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(load_library().piper_hip_comm_unique_id(buf))
    return buf.raw


def config_string():
    """The PIPER_HIP_* tuning switches this process has honoured so far (needs PIPER_HIP_TUNING=1), '' when none."""
    buf = C.create_string_buffer(2048)
    _check(load_library().piper_hip_config_string(buf, len(buf)))
    return buf.value.decode()


def comm_available():
    """True when the library can reach RCCL here (librccl.so.1 loads and ncclGetUniqueId works). Local, not a collective — what
    every rank checks BEFORE the ranks agree to enter piper_hip_comm_* together (bench.py)."""
    try:
        comm_unique_id()
        return True
    except Exception:
        return False


class Comm:
    """piper_hip_comm_*: an RCCL communicator for the one-shot voice-blob broadcast (and the bench's MAX / barrier)."""

    def __init__(self, backend, unique_id, rank, world):
        assert len(unique_id) == COMM_ID_BYTES
        self.lib = backend.lib
        h = c_vp()
        _check(self.lib.piper_hip_comm_create(backend.ctx, C.c_char_p(unique_id), int(rank), int(world), C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.lib.piper_hip_comm_destroy(self.h)
            self.h = None

    @property
    def rank(self):
        return self.lib.piper_hip_comm_rank(self.h)

    @property
    def world(self):
        return self.lib.piper_hip_comm_world(self.h)

    def broadcast_f32(self, device_ptr, count, root=0):
        ptr = _ptr(device_ptr)
        if not isinstance(ptr, (C.c_void_p, C._Pointer)):
            ptr = C.c_void_p(int(ptr))  # a raw address (torch.Tensor.data_ptr())
        _check(self.lib.piper_hip_comm_broadcast_f32(self.h, ptr, int(count), int(root)))

    def max(self, value):
        v = C.c_double(float(value))
        _check(self.lib.piper_hip_comm_max_f64(self.h, C.byref(v)))
        return v.value

    def barrier(self):
        _check(self.lib.piper_hip_comm_barrier(self.h))


class OnnxModel:
    """A Piper `.onnx` opened by the library's own protobuf reader (host-only; the role of PiperONNX.ONNXLoader)."""

    def __init__(self, path=None, data=None):
        self.lib = load_library()
        h = c_vp()
        if path is not None:
            _check(self.lib.piper_hip_onnx_open(str(path).encode(), C.byref(h)))
        else:
            buf = (C.c_char * len(data)).from_buffer_copy(data)
            _check(self.lib.piper_hip_onnx_open_memory(buf, len(data), C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.lib.piper_hip_onnx_close(self.h)
            self.h = None

    def counts(self):
        ir, op, nn, ni = C.c_int64(), C.c_int64(), C.c_int(), C.c_int()
        _check(self.lib.piper_hip_onnx_counts(self.h, C.byref(ir), C.byref(op), C.byref(nn), C.byref(ni)))
        return dict(ir_version=ir.value, opset=op.value, nodes=nn.value, initializers=ni.value)

    def initializer(self, index):
        info = OnnxTensorInfo()
        _check(self.lib.piper_hip_onnx_initializer(self.h, index, C.byref(info)))
        return dict(name=info.name.decode(), data_type=info.data_type, dims=[int(d) for d in info.dims[:info.rank]],
                    count=int(info.count))

    def find(self, name):
        return int(self.lib.piper_hip_onnx_find(self.h, name.encode()))

    def read_f32(self, index):
        n = self.initializer(index)["count"]
        out = np.empty(n, np.float32)
        _check(self.lib.piper_hip_onnx_read_f32(self.h, index, out.ctypes.data_as(c_f32p), n))
        return out

    def infer_config(self):
        cfg = VoiceConfig()
        _check(self.lib.piper_hip_onnx_infer_config(self.h, C.byref(cfg)))
        return cfg

    def verify_graph(self, cfg):
        """Raises (UNSUPPORTED, naming the first differing node) unless the node graph is the computation the launch schedule performs."""
        _check(self.lib.piper_hip_onnx_verify_graph(self.h, C.byref(cfg)))

    def build_blob(self, cfg, verify=True):
        blob = np.empty(blob_floats(cfg), np.float32)
        fn = self.lib.piper_hip_onnx_build_blob if verify else self.lib.piper_hip_onnx_build_blob_unchecked
        _check(fn(self.h, C.byref(cfg), blob.ctypes.data_as(c_f32p), blob.size))
        return blob


def load_voice(onnx_path, json_path=None, verify=True):
    """(cfg, blob, json info) of a Piper voice: `<voice>.onnx` + `<voice>.onnx.json` (PiperVoices layout). The node graph is verified
    against the launch schedule (refused otherwise) unless verify=False."""
    m = OnnxModel(onnx_path)
    try:
        cfg = m.infer_config()
        info = None
        jp = json_path or (str(onnx_path) + ".json")
        if os.path.exists(jp):
            info = piper_json(open(jp, "r", encoding="utf-8").read())
            _check(load_library().piper_hip_voice_check_json(C.byref(cfg), C.byref(info)))  # multi-speaker / vocabulary mismatch
            cfg.sample_rate = info.sample_rate
        return cfg, m.build_blob(cfg, verify), info
    finally:
        m.close()


def pcm16(samples):
    a = np.ascontiguousarray(samples, np.float32)
    out = np.empty(a.size, np.int16)
    _check(load_library().piper_hip_pcm16_from_f32(a.ctypes.data_as(c_f32p), a.size, out.ctypes.data_as(C.POINTER(C.c_int16))))
    return out


def wav_write(path, samples, sample_rate=22050):
    a = np.ascontiguousarray(samples, np.float32)
    _check(load_library().piper_hip_wav_write(str(path).encode(), a.ctypes.data_as(c_f32p), a.size, int(sample_rate)))


def piper_json(text):
    info = PiperJsonInfo()
    _check(load_library().piper_hip_piper_json(text.encode("utf-8"), C.byref(info)))
    return info


class HipRuntime:
    """PiperMetalRuntime.synthesize (PiperMetalRuntime.swift:62-80) over the C-ABI, durations/noise injected."""

    def __init__(self, backend, cfg, blob, on_device=False):
        self.backend, self.cfg, self.lib = backend, cfg, backend.lib
        v = c_vp()
        if on_device:
            ptr = blob
        else:
            self._blob = np.ascontiguousarray(blob, np.float32)
            ptr = self._blob.ctypes.data_as(c_vp)
        _check(self.lib.piper_hip_voice_create(backend.ctx, C.byref(cfg), ptr, int(on_device), C.byref(v)))
        self.voice = v
        self._keep = {}
        self._pinned = []

    def close(self):
        if self.voice:
            self.lib.piper_hip_voice_destroy(self.voice)
            self.voice = None
        for p in self._pinned:
            self.lib.piper_hip_host_free(self.backend.ctx, p)
        self._pinned = []

    def set_precision(self, precision):
        """"f32" (default, the parity configuration) or "bf16" (generator convs on bf16 operands, fp32 accumulate)."""
        code = {"f32": 0, "fp32": 0, "bf16": 1}.get(precision, precision)
        _check(self.lib.piper_hip_voice_set_precision(self.voice, int(code)))
        self._keep.clear()  # prepared slots are dropped by the library

    def _utt(self, ids, durations, noise, noise_scale, noise_mode="injected", seed=1234, length_scale=1.0, noise_w=0.8, dp_noise=None):
        """durations None ⇒ predicted on the device (duration predictor) from length_scale / noise_w / dp_noise."""
        ids = np.ascontiguousarray(ids, np.int64)
        dur = None if durations is None else np.ascontiguousarray(durations, np.int32)
        nz = None if noise is None else np.ascontiguousarray(noise, np.float32)
        dpn = None if dp_noise is None else np.ascontiguousarray(dp_noise, np.float32)
        u = Utterance(ids.ctypes.data_as(c_i64p), len(ids), None if dur is None else dur.ctypes.data_as(c_i32p),
                      None if nz is None else nz.ctypes.data_as(c_f32p), float(noise_scale),
                      NOISE_MODES.get(noise_mode, noise_mode), int(seed) & 0xFFFFFFFF, float(length_scale), float(noise_w),
                      None if dpn is None else dpn.ctypes.data_as(c_f32p))
        return u, (ids, dur, nz, dpn)

    def num_samples(self, ids, durations):
        u, _k = self._utt(ids, durations, None, 0.0)
        return int(self.lib.piper_hip_voice_num_samples(self.voice, C.byref(u)))

    def synthesize(self, phonemeIDs, durations=None, noise=None, noiseScale=0.667, lengthScale=1.0, noiseW=0.8, **kw):
        """PiperMetalRuntime.synthesize(phonemeIDs:noiseScale:lengthScale:noiseW:); `durations` / `noise` / dp_noise are the
        reference's `overrides`. Without durations the frames per id come from the voice's duration predictor."""
        if durations is not None and not kw:  # the one-call C entry point
            u, _k = self._utt(phonemeIDs, durations, noise, noiseScale)
            n = int(self.lib.piper_hip_voice_num_samples(self.voice, C.byref(u)))
            out = np.empty(max(n, 1), np.float32)
            got = C.c_int64()
            _check(self.lib.piper_hip_voice_synthesize(self.voice, C.byref(u), out.ctypes.data_as(c_f32p), n, C.byref(got)))
            self._keep.pop(0, None)
            return out[:got.value]
        self.prepare(0, phonemeIDs, durations, noise, noiseScale, length_scale=lengthScale, noise_w=noiseW, **kw)
        self.launch(0)
        return self.collect(0)

    def prepare(self, slot, phonemeIDs, durations=None, noise=None, noiseScale=0.667, noise_mode="injected", seed=1234, length_scale=1.0,
                noise_w=0.8, dp_noise=None, max_frames=None):
        """max_frames (durations must be None): predicted durations WITHOUT the host round trip — the plan is the bucket of that bound and the
        frame count stays on the device until collect (piper_hip_voice_prepare_batch_bounded)."""
        u, k = self._utt(phonemeIDs, durations, noise, noiseScale, noise_mode, seed, length_scale, noise_w, dp_noise)
        if max_frames is not None:
            rc = self.lib.piper_hip_voice_prepare_batch_bounded(self.voice, C.byref(u), 1, slot, int(max_frames))
        else:
            rc = self.lib.piper_hip_voice_prepare(self.voice, C.byref(u), slot)
        if rc < 0:
            _check(rc)
        tot = C.c_int64()
        _check(self.lib.piper_hip_voice_prepared_samples(self.voice, slot, None, 0, C.byref(tot)))
        self._keep[slot] = (k, int(tot.value), max_frames is not None)
        return rc

    def prepare_batch_bounded(self, slot, utterances, max_frames, noiseScale=0.667, noise_mode="device", seed=1234, length_scale=1.0, noise_w=0.8):
        """utterances: list of (phonemeIDs, dp_noise-or-None); durations predicted on the device, at most max_frames frames per item."""
        n = len(utterances)
        arr = (Utterance * n)()
        keep = []
        for i, (ids, dpn) in enumerate(utterances):
            u, k = self._utt(ids, None, None, noiseScale, noise_mode, seed, length_scale, noise_w, dpn)
            arr[i] = u
            keep.append(k)
        rc = self.lib.piper_hip_voice_prepare_batch_bounded(self.voice, arr, n, slot, int(max_frames))
        if rc < 0:
            _check(rc)
        tot = C.c_int64()
        _check(self.lib.piper_hip_voice_prepared_samples(self.voice, slot, None, 0, C.byref(tot)))
        self._keep[slot] = (keep, int(tot.value), True)
        return rc

    def prepared_samples(self, slot):
        """samples per batch item of the slot (a bounded slot: its capacity before collect, the true lengths after)."""
        nb = self.lib.piper_hip_voice_batch_size(self.voice, slot)
        per = (C.c_int64 * max(nb, 1))()
        tot = C.c_int64()
        _check(self.lib.piper_hip_voice_prepared_samples(self.voice, slot, per, nb, C.byref(tot)))
        return [int(x) for x in per[:nb]], int(tot.value)

    def durations(self, slot):
        """Frames per id the prepared slot uses (supplied or predicted), items back to back."""
        n = C.c_int()
        _check(self.lib.piper_hip_voice_durations(self.voice, slot, None, 0, C.byref(n)))
        out = np.empty(n.value, np.int32)
        _check(self.lib.piper_hip_voice_durations(self.voice, slot, out.ctypes.data_as(c_i32p), n.value, C.byref(n)))
        return out

    def predict_durations(self, utterances, noise_w=0.8, length_scale=1.0, noise_mode="injected", seed=1234):
        """utterances: list of (phonemeIDs, dp_noise-or-None). Returns (durations, logw) lists per utterance."""
        n = len(utterances)
        arr = (Utterance * n)()
        keep = []
        for i, (ids, dpn) in enumerate(utterances):
            u, k = self._utt(ids, None, None, 0.0, noise_mode, seed, length_scale, noise_w, dpn)
            arr[i] = u
            keep.append(k)
        total = sum(len(u[0]) for u in utterances)
        dur = np.empty(total, np.int32)
        lw = np.empty(total, np.float32)
        _check(self.lib.piper_hip_voice_predict_durations(self.voice, arr, n, dur.ctypes.data_as(c_i32p), lw.ctypes.data_as(c_f32p), total))
        outs, off = [], 0
        for ids, _ in utterances:
            outs.append((dur[off:off + len(ids)].copy(), lw[off:off + len(ids)].copy()))
            off += len(ids)
        return outs

    def synthesize_stream(self, phonemeIDs, durations, noise=None, noiseScale=0.667, chunkFrames=64, slot=0):
        """Generator of waveform chunks (PiperMetalRuntime.synthesizeStream): encoder + flow once, generator per window."""
        u, keep = self._utt(phonemeIDs, durations, noise, noiseScale)
        n_chunks = self.lib.piper_hip_voice_stream_begin(self.voice, C.byref(u), slot, int(chunkFrames))
        if n_chunks < 0:
            _check(n_chunks)
        buf = np.empty(int(chunkFrames) * self.cfg.hop, np.float32)
        got = C.c_int64()
        while True:
            _check(self.lib.piper_hip_voice_stream_next(self.voice, slot, buf.ctypes.data_as(c_f32p), buf.size, C.byref(got)))
            if got.value == 0:
                return
            yield buf[:got.value].copy()

    def prepare_batch(self, slot, utterances, noiseScale=0.667):
        """utterances: list of (phonemeIDs, durations, noise-or-None); lengths may differ (ragged batch, one bucket)."""
        n = len(utterances)
        arr = (Utterance * n)()
        keep = []
        total = 0
        for i, (ids, dur, noise) in enumerate(utterances):
            u, k = self._utt(ids, dur, noise, noiseScale)
            arr[i] = u
            keep.append(k)
        rc = self.lib.piper_hip_voice_prepare_batch(self.voice, arr, n, slot)
        if rc < 0:
            _check(rc)
        tot = C.c_int64()
        _check(self.lib.piper_hip_voice_prepared_samples(self.voice, slot, None, 0, C.byref(tot)))
        self._keep[slot] = (keep, int(tot.value))
        return rc

    def plan_info(self, slot):
        bt, bf, n, by = C.c_int32(), C.c_int32(), C.c_int32(), C.c_size_t()
        _check(self.lib.piper_hip_voice_plan_info(self.voice, slot, C.byref(bt), C.byref(bf), C.byref(n), C.byref(by)))
        return dict(bucket_t=bt.value, bucket_f=bf.value, cached_plans=n.value, cached_bytes=by.value)

    def set_plan_cache(self, max_plans, max_bytes=0):
        _check(self.lib.piper_hip_voice_set_plan_cache(self.voice, int(max_plans), int(max_bytes)))

    def last_build_breakdown(self):
        """ms of the phases of the latest plan build (cold prepare): streams, schedule + arena, arena init, eager pass, capture, instantiate."""
        a = (C.c_double * 6)()
        _check(self.lib.piper_hip_voice_last_build_breakdown(self.voice, a))
        return dict(zip(("streams_events", "schedule_and_arena", "arena_init", "eager_pass", "capture", "instantiate"), (round(x, 3) for x in a)))

    def launch(self, slot):
        _check(self.lib.piper_hip_voice_launch(self.voice, slot))

    def pinned_empty(self, count):
        """float32 array in page-locked host memory (piper_hip_host_alloc): `collect(slot, out=…)` into it is one DMA. Freed by close()."""
        p = C.c_void_p()
        _check(self.lib.piper_hip_host_alloc(self.backend.ctx, int(count) * 4, C.byref(p)))
        self._pinned.append(p)
        return np.ctypeslib.as_array(C.cast(p, c_f32p), shape=(int(count),))

    def collect(self, slot, want_audio=True, out=None):
        n = self._keep[slot][1]
        if not want_audio:
            _check(self.lib.piper_hip_voice_collect(self.voice, slot, None, 0))
            return None
        if out is None:
            out = np.empty(max(n, 1), np.float32)
        assert out.dtype == np.float32 and out.size >= n and out.flags.c_contiguous
        _check(self.lib.piper_hip_voice_collect(self.voice, slot, out.ctypes.data_as(c_f32p), n))
        if len(self._keep[slot]) > 2 and self._keep[slot][2]:  # bounded: n was the capacity; the true lengths are known now
            n = self.prepared_samples(slot)[1]
            self._keep[slot] = (self._keep[slot][0], n, False)
        return out[:n]

    def tap(self, slot, name, max_floats):
        out = np.empty(max_floats, np.float32)
        n = C.c_size_t()
        _check(self.lib.piper_hip_voice_tap(self.voice, slot, name.encode(), out.ctypes.data_as(c_f32p), max_floats,
                                            C.byref(n)))
        return out[:n.value]

    def last_gpu_ms(self, slot):
        ms = C.c_double()
        _check(self.lib.piper_hip_voice_last_gpu_ms(self.voice, slot, C.byref(ms)))
        return ms.value

    def time_subset(self, slot, name_filter, iters=20):
        """Graph replay of the launches whose name contains `name_filter`: (avg µs per launch, launches, flops, bytes)."""
        us, n, fl, by = C.c_double(), C.c_int(), C.c_double(), C.c_double()
        _check(self.lib.piper_hip_voice_time_subset(self.voice, slot, name_filter.encode(), iters, C.byref(us), C.byref(n),
                                                    C.byref(fl), C.byref(by)))
        return us.value, n.value, fl.value, by.value

    def profile(self, slot, iters=5, max_entries=512):
        arr = (KernelStat * max_entries)()
        n = C.c_int()
        _check(self.lib.piper_hip_voice_profile(self.voice, slot, iters, arr, max_entries, C.byref(n)))
        return [dict(name=a.name.decode(), avg_us=a.avg_us, flops=a.flops, bytes=a.bytes) for a in arr[:n.value]]
