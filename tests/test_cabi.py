"""The C-ABI library loads and exports every symbol include/tmpnn.h declares (no compute calls; CPU only)."""
import os
import re

import pytest

from conftest import REPO


def declared_symbols(header="tmpnn.h"):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tmpnn_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from thermompnn_amd import _lib, build
    build.build_library()
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libtmpnn.so does not export {n}"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names
    # the measurement hooks live in their own header, outside the operator boundary
    dbg = declared_symbols("tmpnn_debug.h")
    assert sorted(_lib.DEBUG_SIGNATURES) == dbg and not set(dbg) & set(names)
    for n in dbg:
        assert hasattr(lib, n), f"libtmpnn.so does not export {n}"
    assert not [n for n in names if "probe" in n or "ablate" in n or "profile" in n]


def test_tensor_table_matches_python_state_dict_order():
    from thermompnn_amd import _lib, weights
    lib = _lib.load()
    shapes = weights.transfer_param_shapes()
    want = [k[len("prot_mpnn."):] if k.startswith("prot_mpnn.") else k for k in shapes]
    assert _lib.tensor_names() == want
    for i, shape in enumerate(shapes.values()):
        n = 1
        for s in shape:
            n *= s
        assert lib.tmpnn_tensor_numel(i) == n
    assert lib.tmpnn_tensor_name(130) is None and lib.tmpnn_tensor_numel(-1) == -1
    assert lib.tmpnn_version() == 200
    assert lib.tmpnn_workspace_bytes(256) > 256 * 48 * 128 * 4
    tables = (66 * 128 + 3 * 21 * 128 + 384 * 384) * 4
    assert lib.tmpnn_weights_packed_bytes() == tables + 122 * 65536
    assert lib.tmpnn_weights_packed_bytes_p(b"f16x2") == tables + 122 * 65536            # fragment images: f16x2 handles only
    assert lib.tmpnn_weights_packed_bytes_p(b"bf16x3") == lib.tmpnn_weights_packed_bytes_p(b"fp32") == tables
    assert lib.tmpnn_weights_packed_bytes_p(b"fp64") == 0


def test_argument_errors_do_not_need_a_gpu():
    """Argument validation happens before any launch: bad calls return error codes + a message."""
    from thermompnn_amd import _lib
    lib = _lib.load()
    rc = lib.tmpnn_knn_topk(None, None, None, 1, 10, 10, 48, None, None, None, None)
    assert rc == -1 and b"null" in lib.tmpnn_last_error()
    rc = lib.tmpnn_gather_nodes(None, None, 1, -3, 2, 4, None, None)
    assert rc == -1 and b"bad shape" in lib.tmpnn_last_error()
    assert lib.tmpnn_gather_nodes(None, None, 0, 5, 2, 4, None, None) == 0     # empty input is OK
    with pytest.raises(_lib.TmpnnError):
        _lib.check(-2, "demo")
    # the device status word maps to error codes on the host (no GPU needed)
    assert lib.tmpnn_status_error(0) == 0
    assert lib.tmpnn_status_error(_lib.STATUS_RANGE) == _lib.E_RANGE and b"bf16x3" in lib.tmpnn_last_error()
    assert lib.tmpnn_status_error(_lib.STATUS_MAXLEN) == -1 and b"max_len" in lib.tmpnn_last_error()
    with pytest.raises(_lib.TmpnnRangeError):
        _lib.check(lib.tmpnn_status_error(_lib.STATUS_RANGE), "demo")
    # precision is an argument of the weight handle; unknown names are an error code, not an abort()
    import ctypes as C
    h = C.c_void_p()
    arr = (C.c_void_p * 118)(*([16] * 118))
    rc = lib.tmpnn_weights_create_p(C.byref(h), arr, 118, C.c_void_p(16), 1 << 30, b"fp64", None)
    assert rc == -1 and b"unknown precision" in lib.tmpnn_last_error()
    with pytest.raises(_lib.TmpnnError, match="not found"):
        _lib.load("/nonexistent/libtmpnn.so")


def test_product_refuses_cpu():
    import torch
    from thermompnn_amd.engine import Engine, gather_nodes
    from thermompnn_amd._lib import TmpnnError
    from thermompnn_amd.weights import synthetic_state_dict
    with pytest.raises(TmpnnError, match="CUDA"):
        Engine(synthetic_state_dict(0, "mpnn"), "cpu")
    with pytest.raises(TmpnnError, match="no CPU path"):
        gather_nodes(torch.zeros(1, 4, 8), torch.zeros(1, 4, 2, dtype=torch.long))


def test_shipped_library_has_no_environment_switches():
    """kernel-form switches and phase timers (getenv, hipMalloc + blocking hipMemcpy + fprintf inside
    launchers) live only in the -DTMPNN_DEBUG_BUILD variant; the shipped library's launchers pick forms from the launch size."""
    from thermompnn_amd import _lib, build
    build.build_library()
    build.build_debug_library()
    names = [b"TMPNN_KNN_REG", b"TMPNN_KNN_SEL", b"TMPNN_FEAT_IMG", b"TMPNN_FEAT_WAVES", b"TMPNN_FEAT_SPLIT", b"TMPNN_FEAT_PROF",
             b"TMPNN_HEAD_SPLIT", b"TMPNN_NODE_IMG", b"TMPNN_NODE_SPLIT", b"TMPNN_NODE_DEEP", b"TMPNN_NODE_PROF", b"TMPNN_EDGE_PROF",
             b"TMPNN_MSG_PROF", b"TMPNN_MSG_WAVE_MIN", b"TMPNN_FUSE_SMALL"]
    shipped, debug = open(_lib.LIB_PATH, "rb").read(), open(_lib.DEBUG_LIB_PATH, "rb").read()
    assert not [n for n in names if n in shipped]
    assert all(n in debug for n in names)
    assert b"phases (" not in shipped and b"phases (" in debug          # the timers' fprintf formats


def test_shipped_kernels_have_one_form_and_no_ablations():
    """the lab is out of the product sources. Timing ablations (TM_ABL_*: results wrong by construction) compile
    only with -DTMPNN_DEBUG_BUILD — the shipped flags refuse them with an #error —, the dead kernel-form switches of rounds 1-4 are
    gone from the sources, and no 'TM_ABL' string is in the shipped library."""
    import glob
    import subprocess
    from thermompnn_amd import _lib, build
    csrc = os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc")
    text = {f: open(f).read() for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))}
    for gone in ("TM_EDGE_UNROLL2", "TM_SETPRIO", "TM_SPLIT_MIX", "TM_MSG_PFD", "TM_EDGE_Y_ALIAS", "TM_GELU_FORM", "TM_GELU_ASM", "TM_MSG_TOUCH"):
        assert not [f for f, t in text.items() if gone in t], gone
    assert b"TM_ABL" not in open(_lib.LIB_PATH, "rb").read()
    assert set(os.path.basename(f) for f in text) >= {"tmpnn_edge.hip", "tmpnn_msg.hip", "tmpnn_edge_msg.hip", "tmpnn_node.hip"}
    assert max(t.count("\n") for t in text.values()) < 1300              # (round 4: one 1 735-line file held seven kernels)
    # an ablation without the debug define does not compile (preprocessor only: no GPU, a second)
    hipcc = build._hipcc()
    src = os.path.join(csrc, "tmpnn_msg.hip")
    flags = [*build.FLAGS, *build.FILE_FLAGS["tmpnn_msg.hip"], "--offload-device-only", "-E", "-o", os.devnull, src]
    bad = subprocess.run([hipcc, *flags, "-DTM_ABL_NOGELU=1"], capture_output=True, text=True)
    assert bad.returncode != 0 and "TMPNN_DEBUG_BUILD" in bad.stderr
    ok = subprocess.run([hipcc, *flags, "-DTM_ABL_NOGELU=1", "-DTMPNN_DEBUG_BUILD"], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr[-500:]
