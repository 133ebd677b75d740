"""ctypes front end of the native columnar result writer (csrc/tmpnn_csv.cpp): the gathered [T, 21] ddG table -> the
reference's CSV layouts (analysis/SSM.py:102-176, analysis/custom_inference.py:64,94-111) without a Python object per row.
``ssm_scan.rows_for_protein`` + ``ssm_scan.write_csv`` stay as the reference-shaped checker (tests compare the bytes)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from . import _lib

SCHEMA_SSM, SCHEMA_CUSTOM_INFERENCE = 0, 1
PICK_BEST, INCLUDE_CYS, NO_HEADER = 1, 2, 4


def _cstrs(items: Sequence) -> C.Array:
    """bytes / str / raw ``char *`` addresses (ints, e.g. tmpnn_pdb_seq) -> a ``const char *[n]``."""
    arr = (C.c_void_p * len(items))()
    keep = []
    for i, s in enumerate(items):
        if isinstance(s, int):
            arr[i] = s
        else:
            b = s if isinstance(s, bytes) else str(s).encode()
            keep.append(C.create_string_buffer(b))
            arr[i] = C.addressof(keep[-1])
    arr._keep = keep
    return arr


def header_text(schema: int = SCHEMA_SSM, pick_best: bool = False) -> bytes:
    buf = C.create_string_buffer(256)
    n = _lib.load().tmpnn_csv_header(int(schema), PICK_BEST if pick_best else 0, buf, 256)
    if n < 0:
        _lib.check(n, "tmpnn_csv_header")
    return buf.raw[:n]


def format_double(v: float) -> str:
    buf = C.create_string_buffer(40)
    _lib.load().tmpnn_csv_format_double(float(v), buf)
    return buf.value.decode()


class CsvWriter:
    """One output file; ``write_ssm`` / ``write_listed`` append chunks (the running index continues)."""

    def __init__(self, path: Optional[str], schema: int = SCHEMA_SSM, header: bool = True, pick_best: bool = False,
                 memory_capacity: int = 0):
        """``header=False``: a part file of a sharded scan (no header line); ``pick_best``: the header carries the reference's
        extra ``dupe_detector`` column at once (otherwise the first ``write_ssm`` decides). ``path=None`` + ``memory_capacity``: no
        file — the text is kept in an anonymous buffer of that many bytes of address space (``memory()``), header-less."""
        self.lib = _lib.load()
        self.h = C.c_void_p()
        flags = (0 if header else NO_HEADER) | (PICK_BEST if pick_best else 0)
        if path is None:
            _lib.check(self.lib.tmpnn_csv_open_mem(int(schema), flags, int(memory_capacity), C.byref(self.h)), "tmpnn_csv_open_mem")
        else:
            _lib.check(self.lib.tmpnn_csv_open_ex(os.fsencode(path), int(schema), flags, C.byref(self.h)), "tmpnn_csv_open")
        self.path, self.rows, self.bytes = path, 0, 0

    def memory(self):
        """-> a ctypes char array over the text written so far (memory writers only; valid until ``close``)."""
        n = C.c_int64()
        p = self.lib.tmpnn_csv_mem(self.h, C.byref(n))
        if not p:
            raise ValueError("not a memory writer")
        return (C.c_char * n.value).from_address(p)

    def write_ssm(self, table: np.ndarray, offsets: np.ndarray, seqs: Sequence, names: Sequence,
                  neighbors: Optional[np.ndarray] = None, model: str = "ThermoMPNN", dataset: str = "custom",
                  datasets: Optional[Sequence] = None, chain: str = "", pick_best: bool = False, include_cys: bool = False,
                  n_threads: int = 0, wt_cells: Optional[Sequence] = None, first_rows: Optional[Sequence[int]] = None,
                  want_bytes: bool = False) -> Optional[np.ndarray]:
        """table: host float32 [T, ld >= 20] (C-contiguous); offsets int32 [n+1]; seqs / names: str, bytes or char* addresses;
        wt_cells: per-protein 'WT Seq' cells when they are not the parsed sequences (a dataset's own wild-type strings);
        first_rows: the running index of each protein's first row when this writer holds one rank's share of a larger listing;
        want_bytes: -> int64 [n] bytes of text written per protein."""
        table = np.ascontiguousarray(table, dtype=np.float32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        n = offsets.size - 1
        assert table.ndim == 2 and len(seqs) == n and len(names) == n
        nb = None if neighbors is None else np.ascontiguousarray(neighbors, dtype=np.int32)
        cs, cn = _cstrs(seqs), _cstrs(names)
        cd = _cstrs(datasets) if datasets is not None else None
        cw = _cstrs(wt_cells) if wt_cells is not None else None
        assert wt_cells is None or len(wt_cells) == n
        flags = (PICK_BEST if pick_best else 0) | (INCLUDE_CYS if include_cys else 0)
        fr = None if first_rows is None else np.ascontiguousarray(first_rows, dtype=np.int64)
        assert fr is None or fr.size == n
        nbytes = np.zeros(n, np.int64) if want_bytes else None
        _lib.check(self.lib.tmpnn_csv_write_ssm_ex(self.h, table.ctypes.data, table.shape[1], offsets.ctypes.data, n, cs, cw, cn,
                                                   None if nb is None else nb.ctypes.data, model.encode(), dataset.encode(), cd,
                                                   chain.encode(), flags, n_threads or min(16, os.cpu_count() or 1),
                                                   None if fr is None else fr.ctypes.data, None if nbytes is None else nbytes.ctypes.data),
                   "tmpnn_csv_write_ssm")
        return nbytes

    def write_listed(self, table: np.ndarray, offsets: np.ndarray, seqs: Sequence, names: Sequence, triples: np.ndarray,
                     neighbors: Optional[np.ndarray] = None, model: str = "ThermoMPNN", dataset: str = "custom") -> None:
        table = np.ascontiguousarray(table, dtype=np.float32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        tri = np.ascontiguousarray(triples, dtype=np.int64).reshape(-1, 3)
        n = offsets.size - 1
        nb = None if neighbors is None else np.ascontiguousarray(neighbors, dtype=np.int32)
        cs, cn = _cstrs(seqs), _cstrs(names)
        _lib.check(self.lib.tmpnn_csv_write_listed(self.h, table.ctypes.data, table.shape[1], offsets.ctypes.data, n, cs, cn,
                                                   None if nb is None else nb.ctypes.data, model.encode(), dataset.encode(),
                                                   tri.ctypes.data, tri.shape[0]), "tmpnn_csv_write_listed")

    def close(self):
        if self.h is not None and self.h.value:
            r, b = C.c_int64(), C.c_int64()
            h, self.h = self.h, None
            _lib.check(self.lib.tmpnn_csv_close(h, C.byref(r), C.byref(b)), "tmpnn_csv_close")
            self.rows, self.bytes = r.value, b.value
        return self.rows

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
