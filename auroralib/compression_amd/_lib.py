"""Loader of the in-tree C-ABI library (libauroralz.so).  Fails loudly: there is no CPU fallback."""
import ctypes as C
import os

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libauroralz.so")
_lib = None
# Set when this process first makes a call that initialises HIP (alz_create, alz_device_count ...).  From then on nothing here starts a compiler: on the GPU pool a process that has
# touched the GPU must not fork + exec (the lazy builders -- synth.py here, the checker library of the tests -- ask this and tell the caller to run `__graft_entry__.build()` instead).
GPU_TOUCHED = False


def refuse_build_after_gpu(what):
    if GPU_TOUCHED:
        raise RuntimeError("%s is missing or older than its source, and this process has already initialised the GPU: build first "
                           "(python -c 'import __graft_entry__ as g; g.build()'), a compiler is not started from here" % what)


class AlzError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("auroralz error %d: %s" % (code, msg))
        self.code = code


def _preload_hip_runtime():
    """One HIP runtime per process: if torch is installed its bundled libamdhip64.so (SONAME libamdhip64.so.7)
    is loaded first, so both torch and libauroralz.so bind to the same runtime instance."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(p):
                C.CDLL(p, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                          "auroralib.compression_amd has no CPU fallback." % SO_PATH)
    _preload_hip_runtime()
    lib = C.CDLL(SO_PATH)
    vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
    lib.alz_last_error.restype = C.c_char_p
    lib.alz_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.alz_destroy.argtypes = [vp]
    lib.alz_destroy.restype = None
    lib.alz_device_info.argtypes = [vp, C.c_char_p, sz, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    lib.alz_decode_batch.argtypes = [vp, vp, u32, vp, sz, vp, vp, sz, vp]
    lib.alz_decode.argtypes = [vp, u32, vp, vp, u32, u32, u32, u32, vp, u32, vp]
    lib.alz_encode_batch.argtypes = [vp, vp, vp, u32, vp, sz, vp, vp, sz, vp, vp]
    lib.alz_encode_batch_device.argtypes = [vp, vp, vp, u32, vp, sz, vp, vp, sz, vp, vp]
    lib.alz_encode_batch_multi.argtypes = [C.POINTER(vp), u32, vp, vp, u32, vp, sz, vp, vp, sz, vp, vp, vp]
    lib.alz_plan_create.argtypes = [vp, vp, u32, vp, C.POINTER(vp)]
    lib.alz_plan_execute.argtypes = [vp, vp, vp, vp, vp]
    lib.alz_plan_execute_timed.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(C.c_float)]
    lib.alz_plan_results.argtypes = [vp, vp, vp]
    lib.alz_plan_destroy.argtypes = [vp, vp]
    lib.alz_plan_destroy.restype = None
    lib.alz_plan_create_multi.argtypes = [vp, u32, vp, u32, vp, vp, C.POINTER(vp), vp]
    lib.alz_plan_execute_multi.argtypes = [vp, vp, vp]
    lib.alz_plan_results_multi.argtypes = [vp, vp]
    lib.alz_plan_destroy_multi.argtypes = [vp]
    lib.alz_plan_destroy_multi.restype = None
    lib.alz_device_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.alz_device_free.argtypes = [vp, vp]
    lib.alz_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    lib.alz_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    lib.alz_memset_d.argtypes = [vp, vp, C.c_int, sz]
    lib.alz_synchronize.argtypes = [vp]
    lib.alz_ctx_set_exact_kernels.argtypes = [vp, C.c_int]
    lib.alz_ctx_set_kernel_variant.argtypes = [vp, C.c_int]
    lib.alz_ctx_release_scratch.argtypes = [vp]
    lib.alz_ctx_big_stream.argtypes = [vp, u32, C.POINTER(C.c_uint64)]
    lib.alz_decode_batch_multi.argtypes = [C.POINTER(vp), u32, vp, u32, vp, sz, vp, vp, sz, vp, vp]
    lib.alz_partition_batch.argtypes = [u32, vp, u32, vp, vp]
    lib.alz_measure_copy_bandwidth.argtypes = [vp, sz, C.c_int, C.POINTER(C.c_double)]
    lib.alz_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    if lib.alz_abi_version() != A.ABI_VERSION:
        raise ImportError("libauroralz.so ABI %d != python mirror %d" % (lib.alz_abi_version(), A.ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise AlzError(rc, load().alz_last_error().decode("utf-8", "replace"))
