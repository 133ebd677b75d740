"""ctypes binding of libtmpnn.so (the C-ABI in include/tmpnn.h). Fails loudly when the HIP library is
missing or does not export the declared surface — there is no CPU or PyTorch fallback."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libtmpnn.so")
DEBUG_LIB_PATH = os.path.join(HERE, "libtmpnn_debug.so")     # -DTMPNN_DEBUG_BUILD: kernel-form switches + phase timers (TMPNN_LIB selects it)

OK = 0
KS = 48
HID = 128
VOCAB = 21
N_MPNN_TENSORS = 118
N_TENSORS = 130

_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_sz = C.c_size_t

# name -> (restype, argtypes); mirrors include/tmpnn.h declaration by declaration
SIGNATURES = {
    "tmpnn_version": (_i, []),
    "tmpnn_last_error": (C.c_char_p, []),
    "tmpnn_matmul_mode": (C.c_char_p, []),
    "tmpnn_num_tensors": (_i, []),
    "tmpnn_tensor_name": (C.c_char_p, [_i]),
    "tmpnn_tensor_numel": (_i64, [_i]),
    "tmpnn_weights_packed_bytes": (_sz, []),
    "tmpnn_weights_packed_bytes_p": (_sz, [C.c_char_p]),
    "tmpnn_status_error": (_i, [C.c_int32]),
    "tmpnn_selftest": (_i, [_p, _p]),
    "tmpnn_weights_create": (_i, [C.POINTER(_p), C.POINTER(_p), _i, _p, _sz, _p]),
    "tmpnn_weights_create_p": (_i, [C.POINTER(_p), C.POINTER(_p), _i, _p, _sz, C.c_char_p, _p]),
    "tmpnn_weights_precision": (C.c_char_p, [_p]),
    "tmpnn_weights_destroy": (None, [_p]),
    "tmpnn_knn_topk": (_i, [_p, _p, _p, _i, _i64, _i, _i, _p, _p, _p, _p]),
    "tmpnn_centrality": (_i, [_p, _p, _p, _i, _i64, C.c_float, _p, _p]),
    "tmpnn_edge_featurize": (_i, [_p, _p, _p, _p, _p, _p, _i64, _p, _p, _p]),
    "tmpnn_gather_nodes": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "tmpnn_gather_rows_i32": (_i, [_p, _p, _i64, _i, _p, _p]),
    "tmpnn_gather_edges": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "tmpnn_layer_workspace_bytes": (_sz, [_i64]),
    "tmpnn_workspace_bytes": (_sz, [_i64]),
    "tmpnn_enc_layer": (_i, [_p, _i, _p, _p, _p, _p, _i64, _p, _sz, _p]),
    "tmpnn_dec_layer": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _i64, _p, _sz, _p]),
    "tmpnn_seq_embed": (_i, [_p, _p, _i64, _p, _p]),
    "tmpnn_log_probs": (_i, [_p, _p, _i64, _p, _p, _p]),
    "tmpnn_ddg_head": (_i, [_p, _p, _p, _p, _i64, _p, _p, _p, _p]),
    "tmpnn_head_generic_workspace_bytes": (_sz, [_i64, _i, _i, _p]),
    "tmpnn_ddg_head_generic": (_i, [_p, _i, _p, _p, _i64, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p, _p]),
    "tmpnn_pdb_parse": (_i, [C.c_char_p, C.c_char_p, C.POINTER(_p)]),
    "tmpnn_pdb_parse_batch": (_i, [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), _i, _i, C.POINTER(_p)]),
    "tmpnn_pdb_parse_batch_status": (_i, [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), _i, _i, C.POINTER(_p), _p]),
    "tmpnn_pdb_length": (_i64, [_p]),
    "tmpnn_pdb_num_chains": (_i, [_p]),
    "tmpnn_pdb_fill": (_i, [_p, _p, _p, _p, _p, _p, C.c_char_p, _p]),
    "tmpnn_pdb_free": (None, [_p]),
    "tmpnn_pdb_seq": (_p, [_p]),
    "tmpnn_pdb_pack_batch": (_i, [C.POINTER(_p), _i, _i, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "tmpnn_csv_open": (_i, [C.c_char_p, _i, C.POINTER(_p)]),
    "tmpnn_csv_open_ex": (_i, [C.c_char_p, _i, _i, C.POINTER(_p)]),
    "tmpnn_csv_header": (_i, [_i, _i, C.c_char_p, _i]),
    "tmpnn_csv_open_mem": (_i, [_i, _i, _i64, C.POINTER(_p)]),
    "tmpnn_csv_mem": (_p, [_p, C.POINTER(_i64)]),
    "tmpnn_csv_write_ssm_ex": (_i, [_p, _p, _i, _p, _i, _p, _p, _p, _p, C.c_char_p, C.c_char_p, _p, C.c_char_p, _i, _i, _p, _p]),
    "tmpnn_csv_write_ssm": (_i, [_p, _p, _i, _p, _i, _p, _p, _p, _p, C.c_char_p, C.c_char_p, _p, C.c_char_p, _i, _i]),
    "tmpnn_csv_write_listed": (_i, [_p, _p, _i, _p, _i, _p, _p, _p, C.c_char_p, C.c_char_p, _p, _i64]),
    "tmpnn_csv_close": (_i, [_p, C.POINTER(_i64), C.POINTER(_i64)]),
    "tmpnn_csv_format_double": (_i, [C.c_double, C.c_char_p]),
    "tmpnn_ssm_forward": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i64, _i, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
}
# include/tmpnn_debug.h: measurement / experiment hooks (bench.py's per-kernel timing, tools/); not the operator boundary
DEBUG_SIGNATURES = {
    "tmpnn_profile_enable": (_i, [_i]),
    "tmpnn_profile_fetch": (_i, [C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(_i64), _i]),
    "tmpnn_profile_select": (_i, [C.c_char_p]),
    "tmpnn_gemm_probe": (_i, [_i, _p, _p, _p, _i64, _i, _p]),
    "tmpnn_clock_probe": (_i, [_i, _i, _p, _p, _p]),
    "tmpnn_clock_monitor": (_i, [_i, _p, _p]),
    "tmpnn_launch_probe": (_i, [_i, _i, _i, _i, _p, _i, _p]),
}
PRECISIONS = ("f16x2", "bf16x3", "fp32")
STATUS_RANGE, STATUS_MAXLEN, STATUS_SELFTEST = 1, 2, 4
E_RANGE = -5


class TmpnnError(RuntimeError):
    pass


class TmpnnRangeError(TmpnnError):
    """TMPNN_E_RANGE: a result left the finite range (fp16 overflow of the f16x2 matrix-core path)."""


_lib = None


def load(path: str | None = None):
    """Load libtmpnn.so and bind every declared symbol. Raises TmpnnError if the library is absent
    (build it with ``python -m thermompnn_amd.build`` / ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("TMPNN_LIB") or LIB_PATH      # TMPNN_LIB: an A/B build (thermompnn_amd/build.py --variant)
    # torch ships its own copy of the ROCm runtime (libamdhip64); it must be the one already mapped when libtmpnn.so
    # resolves its HIP symbols, otherwise two runtimes coexist and the second one sees no device.
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise TmpnnError(f"{path} not found: the HIP engine is not built (run `python -m thermompnn_amd.build`); "
                         "there is no CPU fallback for the product path")
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise TmpnnError(f"cannot load {path}: {e}") from e
    for name, (res, args) in {**SIGNATURES, **DEBUG_SIGNATURES}.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise TmpnnError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != OK:
        msg = load().tmpnn_last_error()
        cls = TmpnnRangeError if rc == E_RANGE else TmpnnError
        raise cls(f"{what or 'tmpnn call'} failed with code {rc}: {msg.decode() if msg else ''}")


def tensor_names():
    lib = load()
    return [lib.tmpnn_tensor_name(i).decode() for i in range(lib.tmpnn_num_tensors())]
