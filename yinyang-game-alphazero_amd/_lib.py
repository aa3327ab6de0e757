"""ctypes binding of libyy_hip.so (include/yy_engine.h).  There is NO fallback: if the HIP library
is missing or a call fails, this raises."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO = os.path.join(CSRC, "libyy_hip.so")
HEADER = os.path.normpath(os.path.join(_HERE, "..", "include", "yy_engine.h"))

YY_OK = 0
YY_E_ARENA = -6
FLAG_ROWCOL = 1
FLAG_ALIASED = 2
FLAG_REUSE_PASS_VALUE = 4
FLAG_REUSE_TRANSPOSITIONS = 8
FLAG_KEEP_EVALUATIONS = 16

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


class YYError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libyy_hip: status {code}: {msg}")
        self.code = code


class MctsConfig(C.Structure):
    _fields_ = [("G", C.c_int32), ("R", C.c_int32), ("C", C.c_int32), ("max_sims", C.c_int32),
                ("cpuct", C.c_float), ("flags", C.c_uint32), ("edges_per_game", C.c_int64),
                ("nodes_per_game", C.c_int64)]


def build(force=False, verbose=False):
    """Compile csrc/yy_engine.hip for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in ("yy_engine.hip", "yy_tower.hip", "yy_towerq.hip", "yy_tower_f32.hip", "yy_tower_h3r.hip",
                                            "yy_tower_g.hip", "yy_fc_heads.hip", "yy_selfplay.hip", "yy_bitboard.h")] + [HEADER]
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(s) for s in srcs):
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = f"{SO}.{os.getpid()}.tmp"            # link to a private name, then rename: a reader never sees a half-written file
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", tmp] + [s for s in srcs if s.endswith(".hip")]
    if verbose:
        print(" ".join(cmd).replace(tmp, SO))
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, SO)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return SO


_LIB = None
_vp = C.c_void_p
_SIGS = {
    # name: argtypes   (restype is int unless noted)
    "yy_rules_valid_mask": [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_uint32, _vp, _vp],
    "yy_rules_step": [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_uint32, _vp, _vp],
    "yy_rules_game_ended": [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_uint32, _vp, _vp, _vp],
    "yy_encode_planes": [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp],
    "yy_rules_mask_terminal_bb": [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_uint32, _vp, _vp, _vp, _vp],
    "yy_pack_boards": [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp],
    "yy_unpack_boards": [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp],
    "yy_mcts_create": [C.POINTER(MctsConfig), C.POINTER(_vp)],
    "yy_mcts_destroy": [_vp],
    "yy_mcts_memory_bytes": [_vp, C.POINTER(C.c_uint64)],
    "yy_mcts_begin": [_vp, _vp, _vp, _vp, _vp, _vp],
    "yy_mcts_expand_root": [_vp, _vp, _vp, C.c_double, _vp],
    "yy_mcts_select": [_vp, _vp, _vp, _vp],
    "yy_mcts_expand_backup": [_vp, _vp, _vp, _vp],
    "yy_mcts_step": [_vp, _vp, _vp, _vp, _vp, _vp],
    "yy_mcts_root_counts": [_vp, _vp, _vp, _vp, _vp],
    "yy_mcts_root_policy": [_vp, C.c_int, _vp, _vp],
    "yy_mcts_root_stats": [_vp, _vp, _vp, _vp],
    "yy_mcts_get_boards": [_vp, _vp, _vp],
    "yy_mcts_status": [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)],
    "yy_mcts_reset_counters": [_vp, _vp],
    "yy_mcts_cache_clear": [_vp, _vp],
    "yy_book_insert": [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int64, _vp, _vp],
    "yy_mcts_set_book": [_vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int],
    "yy_nn_bias_act_bf16": [_vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, _vp],
    "yy_nn_tower_bf16": [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "yy_nn_tower_heads_bf16": [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "yy_nn_head_finish_bf16": [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp],
    "yy_nn_tower_f32": [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "yy_nn_head_finish_f32": [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "yy_compact_rows": [_vp, C.c_int, _vp, _vp, _vp],
    "yy_nn_tower_f16x3_regs": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, _vp],
    "yy_nn_tower_g_forms": [C.c_int, C.POINTER(C.c_int)],
    "yy_nn_tower_g": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "yy_nn_fc_heads_f16x3": [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "yy_selfplay_root_noise": [C.c_uint64, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_double, _vp, _vp],
    "yy_selfplay_sample_actions": [C.c_uint64, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp],
    "yy_version": [],
}


def exported_symbols():
    return sorted(list(_SIGS) + ["yy_last_error"])


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO):
            raise YYError(-100, f"{SO} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                                "there is no CPU fallback for the product path")
        L = C.CDLL(SO)
        for name, args in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        L.yy_last_error.restype = C.c_char_p
        L.yy_last_error.argtypes = []
        _LIB = L
    return _LIB


def check(status):
    if status != YY_OK:
        raise YYError(status, lib().yy_last_error().decode())
    return status
