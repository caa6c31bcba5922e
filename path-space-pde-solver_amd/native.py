"""ctypes binding of libpsp_hip.so (C ABI in include/psp.h).

The library is the product path for the HJB rollout; there is no CPU fallback behind it.
``load()`` raises ``NativeLibraryError`` if the shared object is missing or does not export
every symbol the header declares.  torch is imported first on purpose: libpsp_hip.so needs
``libamdhip64.so.7`` and must bind to the HIP runtime torch already loaded, because stream
handles and device pointers are shared between the two.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the dlopen below)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSP_LIB_PATH", os.path.join(HERE, "csrc", "libpsp_hip.so"))   # override: diagnostic builds

# enums (include/psp.h)
DRIFT_ZERO, DRIFT_DENSE, DRIFT_DIAG, DRIFT_DOUBLE_WELL = 0, 1, 2, 3
SIGMA_IDENTITY, SIGMA_DENSE, SIGMA_SCALED_IDENTITY = 0, 1, 2
RUNCOST_ZERO, RUNCOST_DIAG_QUAD = 0, 1
TERM_LINEAR, TERM_DIAG_QUAD, TERM_SHIFTED_QUAD = 0, 1, 2
LOSS_LOG_VARIANCE, LOSS_MOMENT, LOSS_WEIGHTS, LOSS_REL_ENTROPY = 0, 1, 2, 3
NOISE_SUPPLIED, NOISE_PHILOX = 0, 1
GH_ZERO, GH_QUAD, GH_ALLEN_CAHN, GH_EXPBALL_LIN, GH_EXPBALL_SQ, GH_EXPBALL_SIN = 0, 1, 2, 3, 4, 5
MLP_FP32, MLP_BF16_FWD, MLP_BF16, MLP_F16X3 = 0, 1, 2, 3
DT_F32, DT_F64 = 0, 1
COMM_ID_BYTES = 128
DOM_NONE, DOM_SPHERE, DOM_BOX, DOM_BOX_UPPER_ALL, DOM_BOX_UPPER_ANY, DOM_ANNULUS = 0, 1, 2, 3, 4, 5
ACT_RELU2, ACT_TANH2, ACT_TANH = 0, 1, 2


class NativeLibraryError(RuntimeError):
    pass


class NativeCallError(RuntimeError):
    pass


class HjbConfig(C.Structure):
    _fields_ = [
        ("d", C.c_int32), ("H", C.c_int32), ("K_local", C.c_int32), ("N", C.c_int32),
        ("K_global", C.c_int64), ("k_offset", C.c_int64),
        ("dt", C.c_float), ("sqrt_dt", C.c_float),
        ("drift_kind", C.c_int32), ("sigma_kind", C.c_int32), ("runcost_kind", C.c_int32),
        ("term_kind", C.c_int32), ("adaptive", C.c_int32), ("loss_kind", C.c_int32),
        ("noise_mode", C.c_int32), ("store_path", C.c_int32),
        ("sigma_scale", C.c_float), ("reserved", C.c_int32),
        ("drift", C.c_void_p), ("sigma", C.c_void_p), ("runcost", C.c_void_p), ("term", C.c_void_p),
        ("u_ref", C.c_void_p), ("u_l2_out", C.c_void_p),
        ("mlp_dtype", C.c_int32), ("reserved2", C.c_int32),
        ("iter_dev", C.c_void_p),
        ("range_flag", C.c_void_p),
    ]


class IterState(C.Structure):
    _fields_ = [("iter", C.c_uint32), ("step", C.c_uint32), ("beta1_pow", C.c_double), ("beta2_pow", C.c_double)]


class HjbSizes(C.Structure):
    _fields_ = [
        ("path_bytes", C.c_int64), ("fwd_partial_bytes", C.c_int64), ("grad_partial_bytes", C.c_int64),
        ("n_params", C.c_int32), ("fwd_workgroups", C.c_int32), ("bwd_workgroups", C.c_int32),
        ("fwd_coop_tiles", C.c_int32),
    ]


class GenConfig(C.Structure):
    _fields_ = [
        ("d", C.c_int32), ("H", C.c_int32), ("K_local", C.c_int32), ("N", C.c_int32),
        ("k_offset", C.c_int64),
        ("dt", C.c_float), ("sqrt_dt", C.c_float), ("T", C.c_float), ("sigma_scale", C.c_float),
        ("drift_kind", C.c_int32), ("h_kind", C.c_int32), ("adaptive", C.c_int32), ("noise_mode", C.c_int32),
        ("store_path", C.c_int32), ("domain_kind", C.c_int32),
        ("drift", C.c_void_p),
        ("dom_a", C.c_float), ("dom_b", C.c_float), ("h_par", C.c_float * 4),
        ("d_real", C.c_int32), ("mlp_dtype", C.c_int32),
        ("v_steps_out", C.c_void_p), ("y_steps_out", C.c_void_p),
        ("per_sample_weights", C.c_int32), ("reserved", C.c_int32),
        ("range_flag", C.c_void_p),
    ]


class DnetConfig(C.Structure):
    _fields_ = [("base", HjbConfig), ("d_real", C.c_int32), ("H_real", C.c_int32), ("time_input", C.c_int32),
                ("per_step", C.c_int32), ("r1_out", C.c_void_p), ("r2_out", C.c_void_p), ("images_out", C.c_void_p)]


class DnetSizes(C.Structure):
    _fields_ = [("table_bytes", C.c_int64), ("fwd_partial_bytes", C.c_int64), ("n_params_per_set", C.c_int64),
                ("fwd_workgroups", C.c_int32), ("reserved", C.c_int32),
                ("image_bytes", C.c_int64), ("partial_bytes", C.c_int64),
                ("bwd_supported", C.c_int32), ("slices", C.c_int32), ("padded_params", C.c_int32), ("bwd_workgroups", C.c_int32)]


class GenSizes(C.Structure):
    _fields_ = [
        ("path_bytes", C.c_int64), ("ahat_bytes", C.c_int64), ("grad_partial_bytes", C.c_int64),
        ("n_params", C.c_int32), ("fwd_workgroups", C.c_int32), ("bwd_workgroups", C.c_int32),
        ("fwd_coop_tiles", C.c_int32),
    ]


class GenlConfig(C.Structure):
    _fields_ = [("base", GenConfig), ("has_time", C.c_int32), ("n_hidden", C.c_int32), ("widths", C.c_int32 * 4),
                ("activation", C.c_int32), ("linear_layout", C.c_int32), ("time_first", C.c_int32), ("time_scale", C.c_float)]


class GenlSizes(C.Structure):
    _fields_ = [("table_bytes", C.c_int64), ("path_bytes", C.c_int64), ("ahat_bytes", C.c_int64), ("n_params", C.c_int64),
                ("grad_partial_bytes", C.c_int64), ("n_blocks", C.c_int32), ("fwd_workgroups", C.c_int32),
                ("bwd_workgroups", C.c_int32), ("waves_per_tile", C.c_int32), ("seg_block_offset", C.c_int32 * 5),
                ("reserved", C.c_int32)]


_P = C.c_void_p
SIGNATURES = {
    "psp_version": (C.c_int, []),
    "psp_abi_struct_sizes": (C.c_int, [C.POINTER(C.c_int32 * 6)]),
    "psp_abi_struct_sizes2": (C.c_int, [C.POINTER(C.c_int32 * 2)]),
    "psp_last_error": (C.c_char_p, []),
    "psp_genl_query": (C.c_int, [C.POINTER(GenlConfig), C.POINTER(GenlSizes)]),
    "psp_genl_rollout_fwd": (C.c_int, [C.POINTER(GenlConfig), _P, _P, _P, _P, C.c_uint64, C.c_uint32, _P, _P, _P, _P, _P, _P, _P,
                                       _P, _P]),
    "psp_genl_rollout_bwd": (C.c_int, [C.POINTER(GenlConfig), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "psp_hjb_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "psp_hjb_family": (C.c_int, [C.c_int32, C.c_int32]),
    "psp_hjb_adjoint_sweep": (C.c_int, [C.POINTER(HjbConfig), _P, _P, _P, _P, _P, _P, _P, _P]),
    "psp_dnet_instance_count": (C.c_int, []),
    "psp_dnet_instance_get": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "psp_dnet_query": (C.c_int, [C.POINTER(DnetConfig), C.POINTER(DnetSizes)]),
    "psp_dnet_terminal_reduce": (C.c_int, [C.POINTER(DnetConfig), _P, _P, _P]),
    "psp_dnet_rollout_bwd": (C.c_int, [C.POINTER(DnetConfig), _P, _P, _P, _P, _P]),
    "psp_dnet_adjoint_sweep": (C.c_int, [C.POINTER(DnetConfig), _P, _P, _P, _P, _P, _P, _P, _P]),
    "psp_dnet_rollout_fwd": (C.c_int, [C.POINTER(DnetConfig), _P, _P, C.c_int32, _P, _P, C.c_uint64, C.c_uint32, _P,
                                       _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "psp_gen_instance_count": (C.c_int, []),
    "psp_gen_instance_get": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "psp_hjb_instance_count": (C.c_int, []),
    "psp_hjb_instance_get": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "psp_hjb_query": (C.c_int, [C.POINTER(HjbConfig), C.POINTER(HjbSizes)]),
    "psp_hjb_rollout_fwd": (C.c_int, [C.POINTER(HjbConfig), _P, _P, C.c_int32, _P, _P, C.c_uint64, C.c_uint32,
                                      _P, _P, _P, _P, _P, _P]),
    "psp_hjb_rollout_eval": (C.c_int, [C.POINTER(HjbConfig), _P, _P, C.c_int32, _P, C.c_uint64, C.c_uint32, _P, _P, _P,
                                       _P, _P, _P]),
    "psp_hjb_terminal_reduce": (C.c_int, [C.POINTER(HjbConfig), _P, _P, _P]),
    "psp_hjb_rollout_bwd": (C.c_int, [C.POINTER(HjbConfig), _P, _P, C.c_uint64, C.c_uint32, _P, _P, _P, _P, _P, _P]),
    "psp_adam_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "psp_philox_normal_fill": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_uint64, C.c_uint32, _P]),
    "psp_hjb_control_eval": (C.c_int, [C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_float, _P, _P]),
    "psp_debug_set_stamp_buffer": (C.c_int, [_P, C.c_int64]),
    "psp_iter_state_init": (C.c_int, [C.POINTER(IterState), C.c_uint32, C.c_int32, C.c_float, C.c_float]),
    "psp_iter_state_advance": (C.c_int, [_P, C.c_float, C.c_float, _P]),
    "psp_hjb_terminal_reduce_loss": (C.c_int, [C.POINTER(HjbConfig), _P, _P, _P, _P, _P]),
    "psp_hjb_rollout_bwd_step": (C.c_int, [C.POINTER(HjbConfig), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float,
                                           C.c_float, C.c_float, _P]),
    "psp_adam_step_dev": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "psp_comm_unique_id": (C.c_int, [_P]),
    "psp_comm_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, _P]),
    "psp_comm_destroy": (C.c_int, [_P]),
    "psp_allreduce": (C.c_int, [_P, C.c_int64, C.c_int32, _P, _P]),
    "psp_gen_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "psp_gen_query": (C.c_int, [C.POINTER(GenConfig), C.POINTER(GenSizes)]),
    "psp_gen_rollout_fwd": (C.c_int, [C.POINTER(GenConfig), _P, _P, _P, _P, C.c_uint64, C.c_uint32, _P, _P, _P, _P,
                                      _P, _P, _P, _P]),
    "psp_gen_rollout_bwd": (C.c_int, [C.POINTER(GenConfig), _P, _P, _P, _P, _P, _P, _P, _P]),
}

_lib = None


def load():
    """dlopen libpsp_hip.so (once) and bind every symbol of include/psp.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            "%s not found: build it with `python path-space-pde-solver_amd/build.py` "
            "(or __graft_entry__.build()). The HJB rollout has no non-HIP fallback." % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise NativeLibraryError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise NativeLibraryError("%s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    sizes = (C.c_int32 * 6)()
    lib.psp_abi_struct_sizes(C.byref(sizes))
    mine = [C.sizeof(t) for t in (HjbConfig, HjbSizes, GenConfig, GenSizes, DnetConfig, DnetSizes)]
    if list(sizes) != mine:          # a stale build or a drifted struct declaration would corrupt kernel arguments silently
        raise NativeLibraryError("%s was built for other struct layouts (library %s, binding %s): rebuild it"
                                 % (LIB_PATH, list(sizes), mine))
    sizes2 = (C.c_int32 * 2)()
    lib.psp_abi_struct_sizes2(C.byref(sizes2))
    mine2 = [C.sizeof(GenlConfig), C.sizeof(GenlSizes)]
    if list(sizes2) != mine2:
        raise NativeLibraryError("%s was built for other struct layouts (library %s, binding %s): rebuild it"
                                 % (LIB_PATH, list(sizes2), mine2))
    _lib = lib
    return lib


def is_built():
    return os.path.exists(LIB_PATH)


def last_error():
    return load().psp_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc != 0:
        raise NativeCallError("%s failed (%d): %s" % (what, rc, last_error()))


def ptr(t, offset=0):
    """Raw device/host pointer of a tensor (or None), optionally `offset` ELEMENTS into it."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr() + int(offset) * t.element_size())


def stream_ptr(device):
    if device.type != "cuda":
        return None
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def supported(d, H):
    return bool(load().psp_hjb_supported(int(d), int(H)))


def family(d, H):
    """0 none, 1 narrow kernels, 2 wide kernels (large d)."""
    return int(load().psp_hjb_family(int(d), int(H)))


def instances():
    """[(d, H, family)] of the compiled HJB kernel instances (family 1 narrow, 2 wide)."""
    lib = load()
    out = []
    for i in range(lib.psp_hjb_instance_count()):
        d, H, f = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib.psp_hjb_instance_get(i, C.byref(d), C.byref(H), C.byref(f)), 'psp_hjb_instance_get')
        out.append((d.value, H.value, f.value))
    return out


def gen_instances():
    """[(d, H)] of the compiled GeneralSolver kernel instances."""
    lib = load()
    out = []
    for i in range(lib.psp_gen_instance_count()):
        d, H = C.c_int32(), C.c_int32()
        check(lib.psp_gen_instance_get(i, C.byref(d), C.byref(H)), 'psp_gen_instance_get')
        out.append((d.value, H.value))
    return out


def dnet_instances():
    """[(d, H)] of the compiled DenseNet-control forward kernels."""
    lib = load()
    out = []
    for i in range(lib.psp_dnet_instance_count()):
        d, H = C.c_int32(), C.c_int32()
        check(lib.psp_dnet_instance_get(i, C.byref(d), C.byref(H)), 'psp_dnet_instance_get')
        out.append((d.value, H.value))
    return out


def gen_query_rc(cfg):
    sizes = GenSizes()
    lib = load()
    rc = lib.psp_gen_query(C.byref(cfg), C.byref(sizes))
    return rc, sizes, (lib.psp_last_error().decode() if rc else '')


def query_rc(cfg):
    """psp_hjb_query without raising: (rc, sizes, message)."""
    sizes = HjbSizes()
    lib = load()
    rc = lib.psp_hjb_query(C.byref(cfg), C.byref(sizes))
    return rc, sizes, (lib.psp_last_error().decode() if rc else '')


def gen_supported(d, H):
    return bool(load().psp_gen_supported(int(d), int(H)))


def gen_query(cfg):
    sizes = GenSizes()
    check(load().psp_gen_query(C.byref(cfg), C.byref(sizes)), "psp_gen_query")
    return sizes


def query(cfg):
    sizes = HjbSizes()
    check(load().psp_hjb_query(C.byref(cfg), C.byref(sizes)), "psp_hjb_query")
    return sizes
