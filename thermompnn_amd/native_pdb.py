"""ctypes front end of the native PDB reader/packer (csrc/tmpnn_pdb.cpp): PDB files -> the engine's packed host
arrays without going through Python dicts. Equivalent to ``tied_featurize(alt_parse_PDB(path, chains), ...)`` for the
tensors the hot path consumes (tests/test_host.py checks that equivalence)."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

from . import _lib


def _chains_arg(chains) -> Optional[bytes]:
    if not chains:
        return None
    return ("".join(chains) if not isinstance(chains, str) else chains).encode()


def _unpack(lib, handle) -> dict:
    L = lib.tmpnn_pdb_length(handle)
    out = dict(X=np.empty((L, 4, 3), np.float32), S=np.empty(L, np.int32), mask=np.empty(L, np.float32),
               residue_idx=np.empty(L, np.int32), chain_enc=np.empty(L, np.int32), ca_mask=np.empty(L, np.float32))
    seq = C.create_string_buffer(L + 1)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    _lib.check(lib.tmpnn_pdb_fill(handle, p(out["X"]), p(out["S"]), p(out["mask"]), p(out["residue_idx"]),
                                  p(out["chain_enc"]), seq, p(out["ca_mask"])), "tmpnn_pdb_fill")
    out["seq"] = seq.value.decode()
    out["num_of_chains"] = lib.tmpnn_pdb_num_chains(handle)
    return out


def parse_pdb(path: str, chains=None) -> dict:
    """One structure -> dict(X [L,4,3] f32, S, mask, residue_idx, chain_enc, ca_mask, seq, num_of_chains, name)."""
    lib = _lib.load()
    h = C.c_void_p()
    _lib.check(lib.tmpnn_pdb_parse(os.fsencode(path), _chains_arg(chains), C.byref(h)), "tmpnn_pdb_parse")
    try:
        out = _unpack(lib, h)
    finally:
        lib.tmpnn_pdb_free(h)
    out["name"] = path[path.rfind("/") + 1:-4]
    return out


def parse_pdbs(paths: Sequence[str], chains: Optional[Sequence] = None, n_threads: int = 0, skip_bad: bool = False) -> List[Optional[dict]]:
    """Many structures on several host threads (the many-PDB scan of analysis/SSM.py:105). ``skip_bad``: a file that fails to
    parse yields ``None`` in its place (``tmpnn_pdb_parse_batch_status``) instead of failing the whole batch; the library's last
    error message names the failing files (``_lib.last_error()``)."""
    lib = _lib.load()
    n = len(paths)
    if n == 0:
        return []
    cp = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    cc = (C.c_char_p * n)(*[_chains_arg(c) for c in (chains if chains is not None else [None] * n)])
    hs = (C.c_void_p * n)()
    nt = n_threads or min(32, os.cpu_count() or 1)
    if skip_bad:
        status = np.zeros(n, np.int32)
        _lib.check(lib.tmpnn_pdb_parse_batch_status(cp, cc, n, nt, hs, status.ctypes.data), "tmpnn_pdb_parse_batch_status")
    else:
        _lib.check(lib.tmpnn_pdb_parse_batch(cp, cc, n, nt, hs), "tmpnn_pdb_parse_batch")
    out = []
    try:
        for i in range(n):
            if not hs[i]:
                out.append(None)
                continue
            d = _unpack(lib, C.c_void_p(hs[i]))
            d["name"] = paths[i][paths[i].rfind("/") + 1:-4]
            out.append(d)
    finally:
        for i in range(n):
            if hs[i]:
                lib.tmpnn_pdb_free(C.c_void_p(hs[i]))
    return out
