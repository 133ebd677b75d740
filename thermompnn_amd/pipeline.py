"""PDB files -> ddG tables at engine speed: the many-protein scan of /root/reference/analysis/SSM.py:105-176 as a three-stage
host pipeline over chunks of files,

    parser thread(s)                 GPU stage (the calling thread)                      writer thread
    tmpnn_pdb_parse_batch  ->  ONE async H2D of the packed chunk (pinned staging)  ->  event wait, status word,
    tmpnn_pdb_pack_batch       tmpnn_ssm_forward (no status read-back, no sync)        sink(chunk): native CSV rows /
    into a pinned slot         [centrality]  ONE async D2H into a pinned slot          binary table

so parse(k+1) || forward(k) || write(k-1). Everything the GPU sees is enqueued on ONE stream (the caller's current stream):
the device input / output buffers and the engine workspace are reused chunk after chunk in stream order, only the HOST
buffers are slotted (a slot is recycled when its writer is done). The reference runs one protein per forward and reads
every value back with its own sync (SSM.py:126,139).

A chunk whose f16x2 forward left the fp16 range (TMPNN_STATUS_RANGE in the chunk's status word) is rerun by the writer
thread at the engine's retry precision from the chunk's still-held staging slot, exactly as Engine.ssm_forward does for a
single call; ``TMPNN_STATUS_MAXLEN`` cannot happen (max_len comes from the parsed lengths).
"""
from __future__ import annotations

import ctypes as C
import os
import queue
import threading
import time
import warnings
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import check


def usable_cpus(per_rank: bool = True) -> int:
    """Logical CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota and — one process per
    GPU under a launcher that exports LOCAL_WORLD_SIZE (torchrun) — divided among the ranks sharing this host, so that eight
    pipelines do not each start a thread per CPU (``per_rank=False``: the host's figure)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per)))
        except (OSError, ValueError):
            pass
    if per_rank:
        try:
            n = max(1, n // max(1, int(os.environ.get("LOCAL_WORLD_SIZE") or 1)))
        except ValueError:
            pass
    return n


def _align(x: int, a: int = 256) -> int:
    return (x + a - 1) // a * a


class _Staging:
    """One pinned host slot: the packed inputs of a chunk (one contiguous region -> one H2D copy) and its results."""

    def __init__(self):
        self.inp: Optional[torch.Tensor] = None      # uint8, pinned
        self.out: Optional[torch.Tensor] = None      # float32 [cap, 21], pinned: the ddG tables
        self.cen: Optional[torch.Tensor] = None      # int32 [cap], pinned: neighbour counts (centrality scans)
        self.status = torch.zeros(1, dtype=torch.int32).pin_memory()

    @staticmethod
    def layout(T: int, n: int):
        """Byte offsets of X, S, mask, residue_idx, chain_enc, ca_mask, offsets inside the input region."""
        o, pos = {}, 0
        for name, nbytes in (("X", T * 48), ("S", T * 4), ("mask", T * 4), ("ridx", T * 4), ("cenc", T * 4), ("ca", T * 4),
                             ("offsets", (n + 1) * 4)):
            o[name] = pos
            pos = _align(pos + nbytes)
        return o, pos

    def reserve(self, T: int, n: int):
        _, nbytes = self.layout(T, n)
        if self.inp is None or self.inp.numel() < nbytes:
            self.inp = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8).pin_memory()
        if self.out is None or self.out.shape[0] < T:
            self.out = torch.empty((int(T * 1.25) + 64, 21), dtype=torch.float32).pin_memory()
            self.cen = torch.empty(self.out.shape[0], dtype=torch.int32).pin_memory()


@dataclass
class Chunk:
    """What the sink receives (valid only during the call: the buffers are recycled afterwards)."""
    index: int                       # chunk number
    first: int                       # index of the chunk's first file in the scan's path list
    n: int                           # proteins
    T: int                           # residues
    offsets: np.ndarray              # int32 [n+1]
    table: Optional[np.ndarray]      # float32 [T, 21] view of the pinned slot: ddG of mutating to ALPHABET[a] (None when the scan
                                     # keeps the tables on the device: ``device_table``)
    neighbors: Optional[np.ndarray]  # int32 [T] (#CA within the radius) when the scan computes centrality
    seq_ptrs: List[int]              # char* of every parsed sequence (tmpnn_pdb_seq), for the native writer
    names: List[str]
    handles: C.Array = field(repr=False, default=None)
    row0: int = 0                    # first row of this chunk in the scan's packed residue axis (= its rows of ``device_table``)

    def seqs(self) -> List[str]:
        return [p if isinstance(p, str) else C.string_at(p).decode() for p in self.seq_ptrs]      # (tests hand the sink plain strings)


@dataclass
class ScanStats:
    files: int = 0
    residues: int = 0
    chunks: int = 0
    parse_s: float = 0.0             # busy seconds of each stage (they overlap: the sum exceeds the wall time)
    gpu_enqueue_s: float = 0.0
    gpu_wait_s: float = 0.0          # writer thread waiting for a chunk's event
    sink_s: float = 0.0
    wall_s: float = 0.0
    reruns: int = 0


def scan_files(engine, paths: Sequence[str], chains: Optional[Sequence] = None, sink: Callable[[Chunk], None] = None,
               centrality: bool = False, radius: float = 10.0, chunk_files: int = 96, chunk_residues: int = 1 << 18,
               parse_threads: int = 0, depth: int = 3, device_table: Optional[torch.Tensor] = None) -> ScanStats:
    """Run the scan; ``sink(chunk)`` is called in file order from the writer thread. ``chunk_files`` files are parsed per
    chunk (fewer when their residues would exceed ``chunk_residues``: the chunk is split before the forward).
    ``device_table`` (float32 [>= total residues, 21] on the engine's device): the tables stay THERE — chunk k's forward writes its
    rows ``[row0, row0 + T)`` of it and nothing is copied back (``chunk.table`` is None): what a multi-rank scan gathers over
    RCCL without a host round trip (dist.scan_files)."""
    lib = _lib.load()
    n_files = len(paths)
    chains = list(chains) if chains is not None else [None] * n_files
    assert len(chains) == n_files
    stats = ScanStats(files=n_files)
    if n_files == 0:
        return stats
    from .native_pdb import _chains_arg
    parse_threads = parse_threads or max(1, usable_cpus() - 2)
    dev = engine.device
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev)            # the CALLER's stream: every enqueue of this scan, the writer thread's rerun too
    # pinned staging slots are kept on the engine between scans (pinning a few MB costs milliseconds: hipHostMalloc)
    pool: List[_Staging] = getattr(engine, "_staging_pool", None) or []
    engine._staging_pool = []                              # (taken: a concurrent scan on the same engine gets fresh ones)
    free_slots: "queue.Queue[_Staging]" = queue.Queue()
    slots = [pool.pop() if pool else _Staging() for _ in range(depth)]
    for sl in slots:
        free_slots.put(sl)
    q_parsed: "queue.Queue" = queue.Queue(maxsize=depth)
    q_done: "queue.Queue" = queue.Queue(maxsize=depth)
    errors: List[BaseException] = []
    stop = threading.Event()
    enqueue_lock = threading.Lock()         # a whole forward is enqueued at a time (the rerun path shares stream + workspace)

    def parser():
        try:
            pos, index = 0, 0
            while pos < n_files and not stop.is_set():
                m = min(chunk_files, n_files - pos)
                t0 = time.perf_counter()
                cp = (C.c_char_p * m)(*[os.fsencode(p) for p in paths[pos:pos + m]])
                cc = (C.c_char_p * m)(*[_chains_arg(c) for c in chains[pos:pos + m]])
                hs = (C.c_void_p * m)()
                check(lib.tmpnn_pdb_parse_batch(cp, cc, m, parse_threads, hs), "tmpnn_pdb_parse_batch")
                lens = [lib.tmpnn_pdb_length(C.c_void_p(hs[i])) for i in range(m)]
                stats.parse_s += time.perf_counter() - t0
                # split so that no forward exceeds chunk_residues (a single longer protein still goes alone)
                a = 0
                while a < m:
                    b, tot = a, 0
                    while b < m and (b == a or tot + lens[b] <= chunk_residues):
                        tot += lens[b]
                        b += 1
                    t0 = time.perf_counter()
                    slot = free_slots.get()
                    slot.reserve(tot, b - a)
                    lay, nbytes = _Staging.layout(tot, b - a)
                    base = slot.inp.data_ptr()
                    sub = (C.c_void_p * (b - a))(*[hs[i] for i in range(a, b)])
                    check(lib.tmpnn_pdb_pack_batch(sub, b - a, parse_threads, tot, base + lay["X"], base + lay["S"],
                                                   base + lay["mask"], base + lay["ridx"], base + lay["cenc"], base + lay["ca"],
                                                   base + lay["offsets"]), "tmpnn_pdb_pack_batch")
                    names = [os.path.basename(p)[:-4] for p in paths[pos + a:pos + b]]
                    stats.parse_s += time.perf_counter() - t0
                    q_parsed.put((index, pos + a, slot, sub, b - a, tot, max(lens[a:b]) if b > a else 0, lay, nbytes, names))
                    index += 1
                    a = b
                pos += m
        except BaseException as e:          # noqa: BLE001 - handed to the caller's thread
            errors.append(e)
            stop.set()
        finally:
            q_parsed.put(None)

    def writer():
        try:
            while True:
                item = q_done.get()
                if item is None:
                    return
                index, first, slot, sub, n, T, max_len, lay, ev, row0 = item
                try:
                    if not stop.is_set():
                        t0 = time.perf_counter()
                        if ev is not None:
                            ev.synchronize()
                        stats.gpu_wait_s += time.perf_counter() - t0
                        st = int(slot.status[0]) if ev is not None else 0
                        if st & _lib.STATUS_MAXLEN:
                            check(lib.tmpnn_status_error(_lib.STATUS_MAXLEN), "tmpnn_ssm_forward")
                        if st & _lib.STATUS_RANGE:
                            _rerun(slot, n, T, max_len, lay, row0)
                        t0 = time.perf_counter()
                        off = np.frombuffer((C.c_int32 * (n + 1)).from_address(slot.inp.data_ptr() + lay["offsets"]), dtype=np.int32)
                        if sink is not None:
                            sink(Chunk(index=index, first=first, n=n, T=T, offsets=off, row0=row0,
                                       table=slot.out.numpy()[:T] if device_table is None else None,
                                       neighbors=slot.cen.numpy()[:T] if centrality else None,
                                       seq_ptrs=[lib.tmpnn_pdb_seq(C.c_void_p(sub[i])) for i in range(n)],
                                       names=item_names[index], handles=sub))
                        stats.sink_s += time.perf_counter() - t0
                finally:
                    for i in range(n):
                        lib.tmpnn_pdb_free(C.c_void_p(sub[i]))
                    item_names.pop(index, None)
                    free_slots.put(slot)
        except BaseException as e:          # noqa: BLE001
            errors.append(e)
            stop.set()
            while True:                      # keep draining so the GPU stage never blocks on a full queue
                item = q_done.get()
                if item is None:
                    return
                for i in range(item[4]):
                    lib.tmpnn_pdb_free(C.c_void_p(item[3][i]))
                free_slots.put(item[2])

    rerun_priv: dict = {}                      # the rerun's own workspace + status word (the engine's shared ones may be in use elsewhere)

    def _rerun(slot, n, T, max_len, lay, row0):
        retry = engine.retry_precision
        if engine.precision != "f16x2" or not retry or retry == engine.precision:
            check(lib.tmpnn_status_error(_lib.STATUS_RANGE), f"tmpnn_ssm_forward[{engine.precision}]")
        warnings.warn(f"ThermoMPNN HIP engine: non-finite result in {engine.precision} (an operand left the fp16 range); "
                      f"rerunning this chunk at precision {retry}", RuntimeWarning, stacklevel=2)
        stats.reruns += 1
        host = slot.inp
        view = lambda name, dt, cnt: host[lay[name]:lay[name] + cnt * 4].view(dt)
        # torch's current stream is thread-local: without this the retry would run on the device's DEFAULT stream, unordered with the
        # caller's; it also gets a private workspace, like the scan's own forwards
        with enqueue_lock, torch.cuda.device(dev), torch.cuda.stream(stream):
            if "status" not in rerun_priv:
                rerun_priv["status"] = torch.zeros(1, dtype=torch.int32, device=dev)
            res = engine.ssm_forward(view("X", torch.float32, T * 12).view(T, 4, 3), view("S", torch.int32, T),
                                     view("mask", torch.float32, T), view("ridx", torch.int32, T), view("cenc", torch.int32, T),
                                     view("offsets", torch.int32, n + 1), max_len=max_len, precision=retry, _private=rerun_priv)
            if device_table is None:
                slot.out[:T].copy_(res["ddg"])
            else:
                device_table[row0:row0 + T].copy_(res["ddg"])
            stream.synchronize()

    item_names: dict = {}
    tp = threading.Thread(target=parser, name="tmpnn-parse", daemon=True)
    tw = threading.Thread(target=writer, name="tmpnn-write", daemon=True)
    t_wall = time.perf_counter()
    tp.start()
    tw.start()
    dev_in: Optional[torch.Tensor] = None        # device copy of a chunk's packed inputs (reused in stream order)
    dev_out: Optional[torch.Tensor] = None       # [cap, 21] ddG
    dev_cen: Optional[torch.Tensor] = None       # [cap] neighbour counts
    priv = {"status": torch.zeros(1, dtype=torch.int32, device=dev)}      # this scan's workspace + status word
    cur_item = None                              # the chunk the GPU stage holds (its handles and slot go back if the stage fails)
    row = 0
    try:
        with torch.cuda.device(dev), torch.cuda.stream(stream):
            while True:
                item = cur_item = q_parsed.get()
                if item is None:
                    break
                index, first, slot, sub, n, T, max_len, lay, nbytes, names = item
                item_names[index] = names
                row0, row = row, row + T
                if stop.is_set() or T == 0:
                    q_done.put((index, first, slot, sub, n, T, max_len, lay, None, row0))
                    cur_item = None
                    continue
                t0 = time.perf_counter()
                ev = torch.cuda.Event()
                with enqueue_lock:
                    if dev_in is None or dev_in.numel() < nbytes:
                        dev_in = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=dev)
                    if device_table is not None:
                        if row > device_table.shape[0]:
                            raise ValueError(f"device_table holds {device_table.shape[0]} rows, the scan needs at least {row}")
                        out_rows = device_table[row0:row]
                    else:
                        if dev_out is None or dev_out.shape[0] < T:
                            dev_out = torch.empty((int(T * 1.25) + 64, 21), dtype=torch.float32, device=dev)
                        out_rows = dev_out[:T]
                    if centrality and (dev_cen is None or dev_cen.shape[0] < T):
                        dev_cen = torch.empty(int(T * 1.25) + 64, dtype=torch.int32, device=dev)
                    dev_in[:nbytes].copy_(slot.inp[:nbytes], non_blocking=True)
                    sec = lambda name, dt, cnt: dev_in[lay[name]:lay[name] + cnt * 4].view(dt)
                    X = sec("X", torch.float32, T * 12).view(T, 4, 3)
                    offs = sec("offsets", torch.int32, n + 1)
                    engine.ssm_forward(X, sec("S", torch.int32, T), sec("mask", torch.float32, T), sec("ridx", torch.int32, T),
                                       sec("cenc", torch.int32, T), offs, max_len=max_len, check_status=False, _private=priv,
                                       out={"ddg": out_rows})
                    if device_table is None:
                        slot.out[:T].copy_(out_rows, non_blocking=True)
                    if centrality:
                        engine.centrality(X, sec("ca", torch.float32, T), offs, radius, out=dev_cen[:T])
                        slot.cen[:T].copy_(dev_cen[:T], non_blocking=True)
                    slot.status.copy_(priv["status"], non_blocking=True)
                    ev.record(stream)
                stats.gpu_enqueue_s += time.perf_counter() - t0
                stats.residues += T
                stats.chunks += 1
                q_done.put((index, first, slot, sub, n, T, max_len, lay, ev, row0))
                cur_item = None
    except BaseException as e:                   # noqa: BLE001
        errors.append(e)
        stop.set()
        if cur_item is not None:                 # the chunk this stage was holding when it failed: handles freed, slot returned
            for i in range(cur_item[4]):
                lib.tmpnn_pdb_free(C.c_void_p(cur_item[3][i]))
            item_names.pop(cur_item[0], None)
            free_slots.put(cur_item[2])
        while True:                              # unblock the parser
            try:
                it = q_parsed.get(timeout=0.05)
            except queue.Empty:
                if not tp.is_alive():
                    break
                continue
            if it is None:
                break
            for i in range(it[4]):
                lib.tmpnn_pdb_free(C.c_void_p(it[3][i]))
            free_slots.put(it[2])
    finally:
        q_done.put(None)
        tw.join()
        tp.join()
        if errors:                               # a failed chunk's async H2D copy may still be reading its pinned slot: the next scan's
            try:                                 # parser must not overwrite it under the copy
                stream.synchronize()
            except Exception:                    # noqa: BLE001 - the original error is the one to report
                pass
        engine._staging_pool = slots                       # every slot is back in free_slots by now (both threads have ended)
    stats.wall_s = time.perf_counter() - t_wall
    if errors:
        raise errors[0]
    return stats
