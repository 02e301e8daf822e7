"""ctypes binding of libeaz.so (include/eaz.h).  No fallback: if the HIP library is missing this raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EAZ_LIB", os.path.join(_HERE, "libeaz.so"))  # EAZ_LIB: diagnostic builds only

K_NAMES = ("tree", "gather", "hash", "net", "misc", "search")


class SearchOpts(C.Structure):
    _fields_ = [
        ("num_simulations", C.c_int), ("max_depth", C.c_int), ("max_num_considered_actions", C.c_int),
        ("gumbel_scale", C.c_float), ("rescale_values", C.c_int), ("use_mixed_value", C.c_int),
        ("value_scale", C.c_float), ("maxvisit_init", C.c_float), ("discount", C.c_float),
        ("two_players_game", C.c_int), ("exploration", C.c_int),
        ("epistemic_bonus_after_rescale", C.c_int), ("epistemic_variance_backup", C.c_int),
    ]


class ReanalyzeCfg(C.Structure):
    _fields_ = [("reanalyze_beta", C.c_float), ("exploration_beta", C.c_float), ("exploration_ube_target", C.c_int),
                ("exploration_policy_target_temperature", C.c_float)]


class SelfplayCfg(C.Structure):
    _fields_ = [
        ("selfplay_steps", C.c_int), ("directed_exploration", C.c_int), ("exploration_beta", C.c_float),
        ("uniform_search_policy", C.c_int), ("sample_actions", C.c_int), ("sample_from_improved_policy", C.c_int),
        ("use_graph", C.c_int),
    ]


_vp, _i, _f, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_uint64
_pp = C.POINTER(C.c_void_p)

# name -> argtypes; every function returns int except eaz_last_error
SIGNATURES = {
    "eaz_version": [],
    "eaz_ctx_create": [_i, _pp],
    "eaz_ctx_destroy": [_vp],
    "eaz_key_fold": [_u64, C.c_uint32, C.POINTER(_u64)],
    "eaz_env_create_deepsea": [_vp, _i, _vp, _pp],
    "eaz_env_create_subleq": [_vp, _i, _vp, _i, _i, _pp],
    "eaz_env_set_subleq_reward": [_vp, _i],
    "eaz_subleq16_test_vectors": [_i, _i, _vp, _vp],
    "eaz_subleq_test_vectors": [_i, _i, _i, _vp, _vp],
    "eaz_env_create_connect4": [_vp, _pp],
    "eaz_env_num_players": [_vp],
    "eaz_env_destroy": [_vp],
    "eaz_env_info": [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)],
    "eaz_env_init": [_vp, _i, _u64, _vp, _vp],
    "eaz_env_step": [_vp, _i, _vp, _vp, _u64, _i, _vp, _vp],
    "eaz_env_observe": [_vp, _i, _vp, _vp, _vp],
    "eaz_env_fields": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "eaz_net_create_fc": [_vp, _i, _i, _i, _vp, _vp, _f, _f, _i, _i, _i, _pp],
    "eaz_net_create_resnet": [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _pp],
    "eaz_net_destroy": [_vp],
    "eaz_net_set_weights": [_vp, _vp, _vp, _vp],
    "eaz_net_set_input_order": [_vp, _i, _vp, _vp, _vp],
    "eaz_net_set_resnet_weights": [_vp, _vp, _i, _vp],
    "eaz_net_set_simhash": [_vp, _vp, _vp],
    "eaz_net_set_bitset": [_vp, _vp, _vp],
    "eaz_net_get_bitset": [_vp, _vp, _vp],
    "eaz_net_forward": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "eaz_hash_lookup": [_vp, _i, _vp, _vp, _vp, _vp],
    "eaz_hash_update": [_vp, _i, _vp, _vp],
    "eaz_search_create": [_vp, _vp, _vp, _i, C.POINTER(SearchOpts), _pp],
    "eaz_search_destroy": [_vp],
    "eaz_search_begin": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "eaz_search_simulate": [_vp, _i, _vp],
    "eaz_search_run": [_vp, _vp],
    "eaz_search_finish": [_vp, _vp, _vp, _vp],
    "eaz_search_summary": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "eaz_search_select": [_vp, _i, _vp, _vp, _vp, _vp],
    "eaz_search_expand": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "eaz_search_tree_view": [_vp, _i, _pp, C.POINTER(C.c_int64), C.POINTER(_i)],
    "eaz_search_launch_info": [_vp, C.POINTER(_i), C.POINTER(_i)],
    "eaz_search_debug_poison": [_vp, _i, _i, _vp],
    "eaz_selfplay_run": [_vp, C.POINTER(SelfplayCfg), _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "eaz_random_play": [_vp, _i, _i, _u64, _vp, _vp, _vp],
    "eaz_evaluate_run": [_vp, _f, _i, _vp, _vp, C.POINTER(_i), _vp],
    "eaz_reanalyze_run": [_vp, C.POINTER(ReanalyzeCfg), _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "eaz_profile_enable": [_vp, _i],
    "eaz_profile_read": [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), _i],
    "eaz_profile_net_ksteps": [_vp, C.POINTER(C.c_uint64), _i],
    "eaz_tune_set": [_i, _i],
    "eaz_profile_replay": [_vp, _i, _i, C.POINTER(C.c_double), _vp],
}

_lib = None


class EazError(RuntimeError):
    pass


def load():
    """Load libeaz.so.  The product path has no CPU fallback: a missing library is a hard error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EazError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
            )
        try:
            # One HIP runtime per process: PyTorch ships its own libamdhip64.so (same SONAME as /opt/rocm's).  Loading it
            # first makes libeaz.so bind to that copy, so torch tensors and libeaz kernels share devices and streams.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.eaz_last_error.restype = C.c_char_p
        L.eaz_last_error.argtypes = []
        L.eaz_build_hash.restype = C.c_char_p
        L.eaz_build_hash.argtypes = []
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = C.c_int
            fn.argtypes = args
        _lib = L
    return _lib


def source_hash() -> str:
    """First 12 hex digits of the SHA-256 over csrc/*.hip, csrc/*.hpp (sorted by name) and include/eaz.h of this tree -- what
    csrc/Makefile bakes into the library as eaz_build_hash().  No GPU, no library needed."""
    import glob
    import hashlib

    csrc = os.path.join(_HERE, "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")), key=os.path.basename)
    files.append(os.path.join(os.path.dirname(_HERE), "include", "eaz.h"))
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def build_hash() -> str:
    """The source hash the loaded libeaz.so was built from (eaz_build_hash)."""
    return load().eaz_build_hash().decode()


def check(rc: int) -> None:
    if rc != 0:
        raise EazError(load().eaz_last_error().decode())
