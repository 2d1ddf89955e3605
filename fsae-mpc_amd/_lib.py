"""ctypes binding of libfsaempc.so (include/fsaempc.h).  The library is HIP-only: there is no CPU
fallback anywhere in this package -- if the .so is missing or no gfx950 device is present the calls
raise."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FSAEMPC_LIB") or os.path.join(_HERE, "lib", "libfsaempc.so")

EXPORTS = [
    "fsaempc_qp_default_opts", "fsaempc_qp_workspace_bytes", "fsaempc_qp_solve_batch_device", "fsaempc_qp_solve_batch",
    "fsaempc_ltv_nx", "fsaempc_ltv_nV", "fsaempc_ltv_nC", "fsaempc_ltv_build_qp_batch_device",
    "fsaempc_ltv_workspace_bytes", "fsaempc_ltv_step_batch_device", "fsaempc_ltv_step_batch_device_aux", "fsaempc_last_error", "fsaempc_selftest_mfma",
    "fsaempc_debug_set_dump", "fsaempc_qp_solve_batch_device_aux", "fsaempc_qp_set_timing", "fsaempc_qp_get_timing", "fsaempc_ltv_get_timing",
    "fsaempc_seq_init", "fsaempc_seq_hotstart", "fsaempc_seq_hotstart_matrices", "fsaempc_seq_equality", "fsaempc_seq_cleanup",
    "fsaempc_obtain_reference_batch_device", "fsaempc_reference_live_batch_device",
    "fsaempc_cl_pre_batch_device", "fsaempc_cl_plant_batch_device", "fsaempc_cl_accept_batch_device",
    "fsaempc_track_from_csv", "fsaempc_track_from_points", "fsaempc_track_free", "fsaempc_track_save", "fsaempc_track_load", "fsaempc_track_last_error",
]


class QpOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("tol_loose", C.c_double), ("tol_x", C.c_double), ("inf_bound", C.c_double),
                ("max_iter", C.c_int), ("polish", C.c_int)]


class QpAux(C.Structure):
    _fields_ = [("kkt", C.c_void_p), ("polished", C.c_void_p), ("x_init", C.c_void_p), ("difficulty", C.c_void_p)]


class QpDesc(C.Structure):
    _fields_ = [("nV", C.c_int), ("nC", C.c_int), ("batch", C.c_int), ("shared_HA", C.c_int)]


class Spline(C.Structure):
    _fields_ = [("M", C.c_int), ("dl", C.c_double), ("xP", C.c_void_p), ("yP", C.c_void_p)]


class LtvDesc(C.Structure):
    _fields_ = [("model", C.c_int), ("N", C.c_int), ("batch", C.c_int), ("dt", C.c_double), ("integrator", C.c_int)]


class TrackTable(C.Structure):
    _fields_ = [("M", C.c_int), ("dl", C.c_double), ("L", C.c_double), ("xP", C.POINTER(C.c_double)), ("yP", C.POINTER(C.c_double))]


class FsaempcError(RuntimeError):
    pass


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise FsaempcError("libfsaempc.so not built (%s): run `make` or __graft_entry__.build(); "
                               "there is no CPU fallback" % LIB_PATH)
        try:                       # the library and torch must share ONE HIP runtime: torch ships its own libamdhip64, and a process
            import torch  # noqa: F401  that binds /opt/rocm's copy first and torch's afterwards ends up with a runtime that sees no device
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.fsaempc_last_error.restype = C.c_char_p
        L.fsaempc_qp_workspace_bytes.restype = C.c_longlong
        L.fsaempc_ltv_workspace_bytes.restype = C.c_longlong
        vp, ll = C.c_void_p, C.c_longlong
        L.fsaempc_qp_solve_batch_device.argtypes = [C.POINTER(QpDesc)] + [vp] * 7 + [C.POINTER(QpOpts)] + [vp] * 5 + [vp, ll, vp]
        L.fsaempc_qp_solve_batch_device_aux.argtypes = [C.POINTER(QpDesc)] + [vp] * 7 + [C.POINTER(QpOpts)] + [vp] * 5 + [C.POINTER(QpAux), vp, ll, vp]
        L.fsaempc_qp_solve_batch.argtypes = [C.POINTER(QpDesc)] + [vp] * 7 + [C.POINTER(QpOpts)] + [vp] * 5
        L.fsaempc_ltv_build_qp_batch_device.argtypes = [C.POINTER(LtvDesc), C.POINTER(Spline)] + [vp] * 4 + [vp] * 7 + [vp] * 3 + [vp]
        L.fsaempc_ltv_step_batch_device.argtypes = [C.POINTER(LtvDesc), C.POINTER(Spline)] + [vp] * 4 + [C.POINTER(QpOpts)] + [vp] * 6 + [vp, ll, vp]
        L.fsaempc_ltv_step_batch_device_aux.argtypes = [C.POINTER(LtvDesc), C.POINTER(Spline)] + [vp] * 4 + [C.POINTER(QpOpts)] + [vp] * 6 + [C.POINTER(QpAux), vp, ll, vp]
        L.fsaempc_obtain_reference_batch_device.argtypes = [vp, C.c_double, C.c_int, vp, vp, C.c_double, C.c_int, C.c_int, vp, vp]
        L.fsaempc_reference_live_batch_device.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, vp, vp, vp]
        L.fsaempc_cl_pre_batch_device.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.POINTER(Spline), vp, vp, C.c_int, vp, vp, vp, vp]
        L.fsaempc_cl_plant_batch_device.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, vp, vp, vp, vp, vp, vp, vp]
        L.fsaempc_cl_accept_batch_device.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
        L.fsaempc_debug_set_dump.argtypes = [vp, C.c_int]
        L.fsaempc_track_last_error.restype = C.c_char_p
        L.fsaempc_track_from_csv.argtypes = [C.c_char_p, C.c_int, C.POINTER(TrackTable)]
        L.fsaempc_track_from_points.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(TrackTable)]
        L.fsaempc_track_free.argtypes = [C.POINTER(TrackTable)]
        L.fsaempc_track_save.argtypes = [C.POINTER(TrackTable), C.c_char_p]
        L.fsaempc_track_load.argtypes = [C.c_char_p, C.POINTER(TrackTable)]
        _LIB = L
    return _LIB


def check(rc, what):
    if rc != 0:
        raise FsaempcError("%s failed (%d): %s" % (what, rc, lib().fsaempc_last_error().decode()))


def default_opts(**kw):
    o = QpOpts()
    lib().fsaempc_qp_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o
