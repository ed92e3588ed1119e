"""ctypes binding of libavddpg_hip.so (C ABI: include/avddpg_hip.h).

There is NO fallback: if the shared library is missing or a call fails, an
exception is raised.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C avddpg_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libavddpg_hip.so")
if os.environ.get("AVDDPG_HIP_LIB"):  # diagnostics: A/B another build of the same library (tools/ab.sh @ tag r06-pre-prune)
    LIB_PATH = os.path.abspath(os.environ["AVDDPG_HIP_LIB"])
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "avddpg_hip.h")
AVD_MAX_L = 16


class AvdError(RuntimeError):
    pass


class EnvConsts(C.Structure):
    """avd_env_consts"""
    _fields_ = [("L", C.c_int32), ("model_a", C.c_int32), ("can_terminate", C.c_int32), ("uniform_reset", C.c_int32),
                ("max_ep", C.c_float), ("max_ev", C.c_float), ("abs_action_high", C.c_float),
                ("two_max_a", C.c_float), ("T", C.c_float),
                ("ca", C.c_float), ("cb", C.c_float), ("cc", C.c_float), ("cd", C.c_float),
                ("re_scalar", C.c_float), ("terminal_reward", C.c_float),
                ("stand_still", C.c_float), ("timegap", C.c_float),
                ("reset_ep_max", C.c_float), ("reset_max_ev", C.c_float), ("reset_max_a", C.c_float),
                ("reset_ep_eval", C.c_float), ("reset_ev_eval", C.c_float), ("reset_a_eval", C.c_float),
                ("leader_reset_a", C.c_float),
                ("A", (C.c_float * 16) * AVD_MAX_L), ("B", (C.c_float * 4) * AVD_MAX_L),
                ("C", (C.c_float * 4) * AVD_MAX_L)]


_LAYOUT_FIELDS = ["S", "A", "H1", "H2", "Ha", "B",
                  "aW1", "ab1", "ag1", "abe1", "aW2", "ab2", "ag2", "abe2", "aW3", "ab3", "actor_size",
                  "cWs", "cbs", "cgs", "cbes", "cWa", "cba", "cga", "cbea", "cW2", "cb2", "cg3", "cbe3", "cW3", "cb3",
                  "theta_size",
                  "amm1", "amv1", "amm2", "amv2", "cmms", "cmvs", "cmma", "cmva", "cmm3", "cmv3", "stats_size"]


class MlpLayout(C.Structure):
    """avd_mlp_layout"""
    _fields_ = [(n, C.c_int32) for n in _LAYOUT_FIELDS]


_P = C.c_void_p
_i, _f, _d, _u64, _i64 = C.c_int, C.c_float, C.c_double, C.c_uint64, C.c_int64
_LP = C.POINTER(MlpLayout)

# name -> argtypes (every one of these returns int)
_PROTOS = {
    "avd_mlp_layout_init": [_LP, _i, _i, _i, _i, _i, _i],
    "avd_env_step_f32": [_P, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "avd_env_reset_f32": [_P, _i, _i, _P, _P, _P, _P, _P, _i, _u64, _u64, _P, _P],
    "avd_episode_end_f32": [_P, _i, _i, _i, _P, _P, _P, _P, _P, _P, _i, _P, _P, _P, _P, _i, _u64, _u64, _P],
    "avd_ou_step_f32": [_i, _P, _P, _f, _f, _f, _f, _u64, _u64, _P],
    "avd_policy_f32": [_i, _P, _P, _f, _f, _P, _P],
    "avd_normal_f32": [_i, _P, _f, _u64, _u64, _P],
    "avd_uniform_f32": [_i, _P, _f, _u64, _u64, _P],
    "avd_step_fused_f32": [_P, _i, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _f, _f, _f, _f, _f, _f, _f, _i, _u64, _u64,
                           _u64, _P, _i, _i64, _P, _P],
    "avd_replay_sample_f32": [_i, _i, _i, _i, _i, _P, _i, _u64, _u64, _P, _P, _P, _P, _P, _P],
    "avd_actor_forward_set_f32": [_LP, _i, _i, _P, _P, _P, _i, _f, _P, _P, _P],
    "avd_replay_add_f32": [_i, _i, _i, _i, _P, _i64, _P, _P, _i, _P, _P, _P],
    "avd_replay_indices": [_i, _i, _i, _u64, _u64, _P, _P],
    "avd_replay_gather_f32": [_i, _i, _i, _i, _i, _P, _P, _P, _P, _P, _P, _P],
    "avd_actor_forward_f32": [_LP, _i, _i, _P, _P, _P, _i, _f, _P, _P],
    "avd_actor_forward_cond_f32": [_LP, _i, _i, _P, _P, _P, _i, _f, _P, _P, _P],
    "avd_critic_forward_f32": [_LP, _i, _i, _P, _P, _P, _i, _P, _P, _P],
    "avd_learn_f32": [_LP, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _f, _f, _P, _P, _P],
    "avd_learn_update_f32": [_LP, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _f, _f, _f, _f, _d, _P, _P, _P],
    "avd_learn_update_act_f32": [_LP, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _f, _f, _f, _f, _d, _P, _P, _P, _i, _P, _P],
    "avd_learn_update_plan": [_LP, _i, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "avd_adam_polyak_f32": [_LP, _i, _P, _P, _P, _P, _P, _P, _P, _P, _f, _f, _d, _P],
    "avd_adam_polyak_guarded_f32": [_LP, _i, _P, _P, _P, _P, _P, _P, _P, _P, _f, _f, _d, _P, _P],
    "avd_polyak_f32": [_i64, _P, _P, _d, _P],
    "avd_fed_sum_f32": [_i, _i, _i, _i, _i, _P, _P, _P, _P, _P],
    "avd_fed_finalize_f32": [_i, _i, _P, _f, _P, _P],
    "avd_fed_scatter_f32": [_i, _i, _i, _i, _i, _i, _P, _P, _P],
    "avd_adam_polyak_intra_f32": [_LP, _i, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _f, _f, _d, _P],
    "avd_fed_history_push_f32": [_i, _i, _i, _P, _P, _P, _i, _P, _i, _i, _P, _P, _P],
    "avd_fed_weights_f32": [_i, _i, _i, _P, _P, _i, _P, _P, _P, _P],
    "avd_learn_shared_workspace": [_LP, _i, _i, C.POINTER(C.c_size_t)],
    "avd_learn_shared_bf16": [_LP, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _f, _f, _P, _P, _P, C.c_size_t, _P],
    "avd_learn_set_fused_workspace": [_LP, _i, _i, C.POINTER(C.c_size_t)],
    "avd_learn_set_fused_bf16": [_LP, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _f, _f, _P, _P, _P, C.c_size_t, _P],
    "avd_learn_set_split_workspace": [_LP, _i, _i, C.POINTER(C.c_size_t)],
    "avd_learn_set_split_mfma_count": [_LP, _i, _i, C.POINTER(C.c_ulonglong)],
    "avd_learn_set_split_critic": [_LP, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _f, _f, _P, _P, _P, C.c_size_t, _P],
    "avd_learn_set_split_actor": [_LP, _i, _i, _P, _P, _P, _f, _P, _P, C.c_size_t, _P],
    "avd_learn_set_split_bf16x3": [_LP, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _f, _f, _P, _P, _P, C.c_size_t, _P],  # deprecated alias
    "avd_learn_set_split_f16x3": [_LP, _i, _i, _P, _P, _P, _P, _P, _P, _P, _P, _P, _f, _f, _P, _P, _P, C.c_size_t, _P],
    "avd_actor_forward_shared_workspace": [_LP, _i, _i, C.POINTER(C.c_size_t)],
    "avd_actor_forward_shared_bf16": [_LP, _i, _i, _P, _P, _P, _f, _P, _P, C.c_size_t, _P],
    "avd_gemm_bt_bf16": [_i, _i, _i, _P, C.c_long, _P, C.c_long, _P, C.c_long, _P],
}

_lib = None
DIAG_LIB_PATH = os.path.join(_HERE, "lib", "libavddpg_hip_diag.so")


def _load(path):
    if not os.path.exists(path):
        raise AvdError(f"{path} not found: build the HIP extension first (__graft_entry__.build()); "
                       "avddpg_amd has no CPU fallback")
    l = C.CDLL(path)
    missing = []
    for name, args in _PROTOS.items():
        fn = getattr(l, name, None)
        if fn is None:
            missing.append(name)
            continue
        fn.argtypes = args
        fn.restype = C.c_int
    if missing and not os.environ.get("AVDDPG_HIP_LIB"):
        # the product / diagnostic build of THIS tree must export every entry point (a stale or partial build, or a typo above);
        # only an older build loaded on purpose through AVDDPG_HIP_LIB (A/B tooling) may lack some: call() raises when one is used
        raise AvdError(f"{path} does not export {', '.join(missing)}: rebuild it (make -C avddpg_amd/csrc)")
    l.avd_last_error.restype = C.c_char_p
    l.avd_last_error.argtypes = []
    l.avd_version.restype, l.avd_version.argtypes = C.c_int, []
    if getattr(l, "avd_diagnostics_enabled", None) is None:  # pre-r04 library: no diagnostic build existed
        l.avd_diagnostics_enabled = lambda: 0
    else:
        l.avd_diagnostics_enabled.restype, l.avd_diagnostics_enabled.argtypes = C.c_int, []
    return l


def lib():
    """The loaded library; raises AvdError when it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


class diag_library:
    """Context manager: route every call of this process through the DIAGNOSTIC build (`make -C avddpg_amd/csrc diag`,
    -DAVD_DIAG) while it is active. Only that build reads the AVD_* environment switches (kernel variants for cross-checks
    and A/B runs); the shipped library reads none. Test / tool infrastructure: nothing in the package enters it."""
    _cached = None
    _stack = []

    def __enter__(self):
        global _lib, LIB_PATH
        if diag_library._cached is None:
            diag_library._cached = _load(DIAG_LIB_PATH)
            if not diag_library._cached.avd_diagnostics_enabled():
                raise AvdError(f"{DIAG_LIB_PATH} is not a diagnostic build")
        diag_library._stack.append((lib(), LIB_PATH))  # (nests: every exit restores what its own enter found)
        _lib, LIB_PATH = diag_library._cached, DIAG_LIB_PATH
        return _lib

    def __exit__(self, *exc):
        global _lib, LIB_PATH
        _lib, LIB_PATH = diag_library._stack.pop()
        return False


def call(name, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    fn = getattr(lib(), name, None)
    if fn is None:
        raise AvdError(f"{name} is not exported by the loaded library (an older build?)")
    rc = fn(*args)
    if rc != 0:
        raise AvdError(f"{name} failed ({rc}): {lib().avd_last_error().decode()}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL). The tensor must be contiguous."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise AvdError("non-contiguous tensor passed to the C ABI")
    return C.c_void_p(t.data_ptr())


def stream_handle():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_layout(S, A, H1, H2, Ha, B):
    lay = MlpLayout()
    call("avd_mlp_layout_init", C.byref(lay), S, A, H1, H2, Ha, B)
    return lay
