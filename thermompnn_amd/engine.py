"""Thin Python glue over the C-ABI: torch owns device memory and streams (plumbing), libtmpnn.so does
all arithmetic. Every method takes/returns torch CUDA tensors and launches on the current stream."""
from __future__ import annotations

import ctypes as C
import warnings
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import HID, KS, VOCAB, TmpnnError, TmpnnRangeError, check


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise TmpnnError("the ThermoMPNN HIP engine needs CUDA (ROCm) tensors; there is no CPU path")


_selftested: set = set()


def _selftest(lib, device) -> None:
    """tmpnn_selftest once per (library, device) and process: a library built with flags that break the f16x2 overflow
    detection or the persistent tile loops must not produce numbers. One tiny launch + a 4-byte read-back."""
    key = (id(lib), device.index)
    if key in _selftested:
        return
    st = torch.zeros(1, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        check(lib.tmpnn_selftest(_ptr(st), _stream()), "tmpnn_selftest")
    check(lib.tmpnn_status_error(int(st.item())), "tmpnn_selftest")
    _selftested.add(key)


class Weights:
    """Device-resident weight set + the native handle (tmpnn_weights_create)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device, with_head: Optional[bool] = None,
                 precision: Optional[str] = None, tensors: Optional[List[torch.Tensor]] = None):
        """``precision``: "f16x2" | "bf16x3" | "fp32" (matrix-core path of every call through this handle) or None for
        the library default. ``tensors``: device tensors of another Weights of the same model to share (a second handle
        costs only its derived data: 0.8 MB of tables, plus 7.2 MB of f16 fragment images for an "f16x2" handle)."""
        lib = _lib.load()
        if precision is not None and precision not in _lib.PRECISIONS:
            raise TmpnnError(f"precision={precision!r}: expected one of {_lib.PRECISIONS}")
        device = torch.device(device)
        if device.type != "cuda":
            raise TmpnnError("weights must live on a CUDA (ROCm) device")
        names = _lib.tensor_names()
        sd = dict(state_dict)
        if with_head is None:
            with_head = all(n in sd for n in names[_lib.N_MPNN_TENSORS:])
        n = _lib.N_TENSORS if with_head else _lib.N_MPNN_TENSORS
        self.tensors: List[torch.Tensor] = list(tensors) if tensors is not None else []
        for i, name in enumerate(names[:n] if tensors is None else []):
            key = name if i >= _lib.N_MPNN_TENSORS else ("prot_mpnn." + name if ("prot_mpnn." + name) in sd else name)
            if key not in sd:
                raise KeyError(f"state dict lacks {key!r}")
            t = sd[key].detach().to(device=device, dtype=torch.float32).contiguous()
            if t.numel() != lib.tmpnn_tensor_numel(i):
                raise TmpnnError(f"{key}: {t.numel()} elements, engine expects {lib.tmpnn_tensor_numel(i)}")
            if t.data_ptr() % 16:
                t = t.clone()
            self.tensors.append(t)
        self.device = device
        self.with_head = with_head
        _selftest(lib, device)
        nbytes = lib.tmpnn_weights_packed_bytes_p(precision.encode() if precision else None) or lib.tmpnn_weights_packed_bytes()
        self.packed = torch.empty(nbytes, dtype=torch.uint8, device=device)      # (0 = unknown name: create_p reports it)
        arr = (C.c_void_p * n)(*[t.data_ptr() for t in self.tensors])
        self.handle = C.c_void_p()
        with torch.cuda.device(device):
            check(lib.tmpnn_weights_create_p(C.byref(self.handle), arr, n, _ptr(self.packed), self.packed.numel(),
                                             precision.encode() if precision else None, _stream()), "tmpnn_weights_create_p")
        self.precision = lib.tmpnn_weights_precision(self.handle).decode()

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            try:
                _lib.load().tmpnn_weights_destroy(h)
            except Exception:      # interpreter shutdown: the library may already be gone
                pass
            self.handle = None


class Engine:
    """One weight set on one GPU. All tensors are packed along the residue axis (T = sum of lengths)."""

    def __init__(self, state_dict, device="cuda", k_neighbors: int = 48, precision: Optional[str] = None,
                 retry_precision: Optional[str] = "bf16x3", with_head: Optional[bool] = None):
        """``precision``: matrix-core path of the per-edge GEMMs ("f16x2" default | "bf16x3" | "fp32").
        ``retry_precision``: when a forward in f16x2 leaves the finite range (an operand >= 65504 overflowed fp16), it is
        rerun once at this precision with a warning (None: raise TmpnnRangeError instead)."""
        self.lib = _lib.load()
        self.w = Weights(state_dict, device, with_head=with_head, precision=precision)
        self.device = self.w.device
        self.precision = self.w.precision
        self.retry_precision = retry_precision
        if not 1 <= int(k_neighbors) <= KS:
            raise TmpnnError(f"k_neighbors={k_neighbors} outside [1, {KS}]")
        self.K = int(k_neighbors)
        self._ws: Optional[torch.Tensor] = None
        self._alt: Dict[str, Weights] = {self.precision: self.w}
        self._status = torch.zeros(1, dtype=torch.int32, device=self.device)

    def weights_for(self, precision: str) -> Weights:
        """The handle for ``precision`` (created on first use; shares the raw tensors)."""
        if precision not in self._alt:
            self._alt[precision] = Weights({}, self.device, with_head=self.w.with_head, precision=precision,
                                           tensors=self.w.tensors)
        return self._alt[precision]

    def _raise_status(self, what: str, status: Optional[torch.Tensor] = None) -> int:
        """Reads the device status word (ONE 4-byte D2H copy = a stream sync) and raises on TMPNN_STATUS_MAXLEN;
        returns the remaining bits (TMPNN_STATUS_RANGE is the caller's to handle)."""
        st = int((self._status if status is None else status).item())
        if st & _lib.STATUS_MAXLEN:
            check(self.lib.tmpnn_status_error(_lib.STATUS_MAXLEN), what)
        return st

    @staticmethod
    def _max_len(offsets, given: Optional[int]) -> int:
        """Longest protein of the batch. Derived from ``offsets`` whenever that costs no device sync (host data);
        a caller-supplied value is only trusted for device-resident offsets (the kernel still guards it and reports
        TMPNN_STATUS_MAXLEN instead of overrunning its scratch)."""
        if not (isinstance(offsets, torch.Tensor) and offsets.is_cuda):
            o = np.asarray(offsets).astype(np.int64)
            true_len = int((o[1:] - o[:-1]).max()) if o.size > 1 else 0
            if given is not None and given < true_len:
                raise TmpnnError(f"max_len={given} is smaller than the longest protein ({true_len})")
            return max(true_len, 1) if given is None else int(given)
        if given is not None:
            return int(given)
        return int((offsets[1:] - offsets[:-1]).max().item()) if offsets.numel() > 1 else 0

    # -- buffers ---------------------------------------------------------------------------------
    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def _i32(self, t) -> torch.Tensor:
        return torch.as_tensor(t).to(device=self.device, dtype=torch.int32).contiguous()

    def _f32(self, t) -> torch.Tensor:
        return torch.as_tensor(t).to(device=self.device, dtype=torch.float32).contiguous()

    # -- individual operators (parity tests drive these) --------------------------------------------
    def knn_topk(self, X, mask, offsets, max_len: Optional[int] = None):
        max_len = self._max_len(offsets, max_len)
        X, mask, offsets = self._f32(X), self._f32(mask), self._i32(offsets)
        T, N = X.shape[0], offsets.numel() - 1
        E_idx = torch.empty((T, KS), dtype=torch.int32, device=self.device)
        D_nb = torch.empty((T, KS), dtype=torch.float32, device=self.device)
        self._status.zero_()
        check(self.lib.tmpnn_knn_topk(_ptr(X), _ptr(mask), _ptr(offsets), N, T, max_len, self.K, _ptr(E_idx), _ptr(D_nb),
                                      _ptr(self._status), _stream()), "tmpnn_knn_topk")
        self._raise_status("tmpnn_knn_topk")
        return E_idx, D_nb

    def centrality(self, X, mask, offsets, radius: float = 10.0, out: Optional[torch.Tensor] = None):
        X, mask, offsets = self._f32(X), self._f32(mask), self._i32(offsets)
        if out is None:
            out = torch.empty(X.shape[0], dtype=torch.int32, device=self.device)
        check(self.lib.tmpnn_centrality(_ptr(X), _ptr(mask), _ptr(offsets), offsets.numel() - 1, X.shape[0], float(radius),
                                        _ptr(out), _stream()), "tmpnn_centrality")
        return out

    def edge_featurize(self, X, residue_idx, chain_enc, E_idx, D_nb, want_E: bool = False):
        X, ridx, cenc = self._f32(X), self._i32(residue_idx), self._i32(chain_enc)
        T = X.shape[0]
        h_E = torch.empty((T, KS, HID), dtype=torch.float32, device=self.device)
        E = torch.empty_like(h_E) if want_E else None
        check(self.lib.tmpnn_edge_featurize(self.w.handle, _ptr(X), _ptr(ridx), _ptr(cenc), _ptr(E_idx), _ptr(D_nb), T,
                                            _ptr(h_E), _ptr(E), _stream()), "tmpnn_edge_featurize")
        return (h_E, E) if want_E else h_E

    def enc_layer(self, layer: int, h_V, h_E, E_idx, mask):
        T = h_V.shape[0]
        ws = self._workspace(self.lib.tmpnn_layer_workspace_bytes(T))
        check(self.lib.tmpnn_enc_layer(self.w.handle, layer, _ptr(h_V), _ptr(h_E), _ptr(E_idx), _ptr(mask), T, _ptr(ws),
                                       ws.numel(), _stream()), "tmpnn_enc_layer")
        return h_V, h_E

    def dec_layer(self, layer: int, h_V, h_E, E_idx, S, mask):
        T = h_V.shape[0]
        out = torch.empty_like(h_V)
        ws = self._workspace(self.lib.tmpnn_layer_workspace_bytes(T))
        check(self.lib.tmpnn_dec_layer(self.w.handle, layer, _ptr(h_V), _ptr(out), _ptr(h_E), _ptr(E_idx), _ptr(S),
                                       _ptr(mask), T, _ptr(ws), ws.numel(), _stream()), "tmpnn_dec_layer")
        return out

    def seq_embed(self, S):
        S = self._i32(S)
        out = torch.empty((S.numel(), HID), dtype=torch.float32, device=self.device)
        check(self.lib.tmpnn_seq_embed(self.w.handle, _ptr(S), S.numel(), _ptr(out), _stream()), "tmpnn_seq_embed")
        return out

    def log_probs(self, h_V):
        out = torch.empty((h_V.shape[0], VOCAB), dtype=torch.float32, device=self.device)
        check(self.lib.tmpnn_log_probs(self.w.handle, _ptr(h_V), h_V.shape[0], _ptr(out), None, _stream()), "tmpnn_log_probs")
        return out

    def ddg_head(self, hV_last, hV_prev, S, want_z: bool = False):
        S = self._i32(S)
        T = S.numel()
        ddg = torch.empty((T, VOCAB), dtype=torch.float32, device=self.device)
        z = torch.empty_like(ddg) if want_z else None
        check(self.lib.tmpnn_ddg_head(self.w.handle, _ptr(hV_last), _ptr(hV_prev), _ptr(S), T, _ptr(ddg), _ptr(z),
                                      None, _stream()), "tmpnn_ddg_head")
        return (ddg, z) if want_z else ddg

    def ddg_head_generic(self, hidden: Sequence[torch.Tensor], S, Ws: torch.Tensor, mlp_w: Sequence[torch.Tensor],
                         mlp_b: Sequence[torch.Tensor], ddg_w: torch.Tensor, ddg_b: torch.Tensor,
                         conv_w: Optional[torch.Tensor] = None, conv_b: Optional[torch.Tensor] = None, want_z: bool = False):
        """The head of ANY TransferModel configuration (tmpnn_ddg_head_generic): ``hidden`` = the decoder states that enter the
        input, LAST layer first (all_mpnn_hid[:num_final_layers], transfer_model.py:84-85); ``conv_w`` [D0, D0, 9] / ``conv_b``
        = LightAttention's feature convolution or None; ``mlp_w[l]`` / ``mlp_b[l]`` = both_out's Linear l. -> ddg [T,21] (, z)."""
        S = self._i32(S)
        T = S.numel()
        f = lambda t: None if t is None else t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        hidden = [f(h) for h in hidden]
        mlp_w, mlp_b = [f(w) for w in mlp_w], [f(b) for b in mlp_b]
        Ws, ddg_w, ddg_b, conv_w, conv_b = f(Ws), f(ddg_w), f(ddg_b), f(conv_w), f(conv_b)
        _need_cuda(S, Ws, *hidden)
        n_layers = len(mlp_w)
        dims = [HID * len(hidden) + HID] + [int(w.shape[0]) for w in mlp_w]
        for l, w in enumerate(mlp_w):
            if tuple(w.shape) != (dims[l + 1], dims[l]) or mlp_b[l].numel() != dims[l + 1]:
                raise TmpnnError(f"both_out Linear {l}: weight {tuple(w.shape)}, expected {(dims[l + 1], dims[l])}")
        if conv_w is not None and tuple(conv_w.shape) != (dims[0], dims[0], 9):
            raise TmpnnError(f"feature_convolution weight {tuple(conv_w.shape)}, expected {(dims[0], dims[0], 9)}")
        cdims = (C.c_int32 * len(dims))(*dims)
        arr = lambda ts: (C.c_void_p * max(1, len(ts)))(*[t.data_ptr() for t in ts])
        need = self.lib.tmpnn_head_generic_workspace_bytes(T, len(hidden), n_layers, cdims)
        if T > 0 and need == 0:
            raise TmpnnError(f"head dims {dims} (num_final_layers {len(hidden)}): not a TransferModel head layout")
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=self.device)
        ddg = torch.empty((T, VOCAB), dtype=torch.float32, device=self.device)
        z = torch.empty_like(ddg) if want_z else None
        self._status.zero_()
        check(self.lib.tmpnn_ddg_head_generic(arr(hidden), len(hidden), _ptr(Ws), _ptr(S), T, _ptr(conv_w), _ptr(conv_b), n_layers,
                                              arr(mlp_w), arr(mlp_b), cdims, _ptr(ddg_w), _ptr(ddg_b), _ptr(ddg), _ptr(z), _ptr(ws),
                                              ws.numel(), _ptr(self._status), _stream()), "tmpnn_ddg_head_generic")
        st = self._raise_status("tmpnn_ddg_head_generic")
        if st:
            check(self.lib.tmpnn_status_error(st), "tmpnn_ddg_head_generic")
        return (ddg, z) if want_z else ddg

    def gather_rows(self, nodes, idx_i32):
        nodes = nodes.contiguous()
        idx = idx_i32.contiguous()
        out = torch.empty((idx.numel(), nodes.shape[-1]), dtype=torch.float32, device=nodes.device)
        check(self.lib.tmpnn_gather_rows_i32(_ptr(nodes), _ptr(idx), idx.numel(), nodes.shape[-1], _ptr(out), _stream()),
              "tmpnn_gather_rows_i32")
        return out

    # -- the fused path -----------------------------------------------------------------------------
    def ssm_forward(self, X, S, mask, residue_idx, chain_enc, offsets, max_len: Optional[int] = None,
                    want_ddg: bool = True, want_hidden: bool = False, want_log_probs: bool = False,
                    want_E_idx: bool = False, out: Optional[dict] = None, check_status: bool = True,
                    precision: Optional[str] = None, _private: Optional[dict] = None):
        """Packed inputs ([T,4,3], [T], ...) -> dict(ddg [T,21], hidden [3,T,128], log_probs [T,21], E_idx [T,48]).

        ``check_status`` (default) reads the 4-byte device status word after the launches — one stream sync — and
          * raises if a protein is longer than ``max_len``,
          * on a non-finite ddG / log-probability (fp16 overflow of the f16x2 path) reruns the batch once at
            ``retry_precision`` with a warning, or raises TmpnnRangeError.
        Throughput loops pass ``check_status=False`` (nothing syncs; call ``check_last_status()`` when convenient).
        ``precision`` overrides the engine's precision for this call. (``_private``: workspace / status word owned by a
        captured graph instead of the engine's shared ones, see ``capture_graph``.)"""
        max_len = self._max_len(offsets, max_len)
        X, mask = self._f32(X), self._f32(mask)
        S, ridx, cenc, offsets = self._i32(S), self._i32(residue_idx), self._i32(chain_enc), self._i32(offsets)
        _need_cuda(X, S, mask)
        T, N = X.shape[0], offsets.numel() - 1
        res = out if out is not None else {}
        dev = self.device
        if want_ddg and "ddg" not in res:
            res["ddg"] = torch.empty((T, VOCAB), dtype=torch.float32, device=dev)
        if want_hidden and "hidden" not in res:
            res["hidden"] = torch.empty((3, T, HID), dtype=torch.float32, device=dev)
        if want_log_probs and "log_probs" not in res:
            res["log_probs"] = torch.empty((T, VOCAB), dtype=torch.float32, device=dev)
        if want_E_idx and "E_idx" not in res:
            res["E_idx"] = torch.empty((T, KS), dtype=torch.int32, device=dev)
        need = self.lib.tmpnn_workspace_bytes(T)
        if _private is not None:
            if _private.get("ws") is None or _private["ws"].numel() < need:
                _private["ws"] = torch.empty(need, dtype=torch.uint8, device=dev)
            if _private.get("status") is None:
                _private["status"] = torch.zeros(1, dtype=torch.int32, device=dev)
            ws, status = _private["ws"], _private["status"]
        else:
            ws, status = self._workspace(need), self._status

        def run(w: Weights):
            check(self.lib.tmpnn_ssm_forward(w.handle, _ptr(X), _ptr(S), _ptr(mask), _ptr(ridx), _ptr(cenc), _ptr(offsets),
                                             N, T, max_len, self.K, _ptr(res.get("ddg") if want_ddg else None),
                                             _ptr(res.get("hidden") if want_hidden else None),
                                             _ptr(res.get("log_probs") if want_log_probs else None),
                                             _ptr(res.get("E_idx") if want_E_idx else None), _ptr(status), _ptr(ws),
                                             ws.numel(), _stream()), "tmpnn_ssm_forward")

        used = precision or self.precision
        run(self.weights_for(used))
        if check_status and T > 0 and N > 0:
            st = self._raise_status("tmpnn_ssm_forward", status)
            if st & _lib.STATUS_RANGE:
                retry = self.retry_precision
                if used != "f16x2" or not retry or retry == used:
                    check(self.lib.tmpnn_status_error(st), f"tmpnn_ssm_forward[{used}]")
                warnings.warn(f"ThermoMPNN HIP engine: non-finite result in {used} (an operand left the fp16 range); "
                              f"rerunning this batch at precision {retry}", RuntimeWarning, stacklevel=2)
                run(self.weights_for(retry))
                st = self._raise_status("tmpnn_ssm_forward", status)
                if st:
                    check(self.lib.tmpnn_status_error(st), f"tmpnn_ssm_forward[{retry}]")
        return res

    def capture_graph(self, X, S, mask, residue_idx, chain_enc, offsets, max_len: int, out: Optional[dict] = None, **want):
        """Capture ONE fused forward (its ~20 launches + the status memset) into a hipGraph: -> (graph, out). ``graph.replay()``
        reruns it on the captured tensors — refill X / S / ... in place for a new protein of the same length. The C-ABI
        never allocates or synchronises, so the call sequence captures as is. Removes the per-launch CPU cost that
        dominates a single small protein (20 launches for ~0.2 ms of GPU work).

        Every buffer whose address is baked into the captured launches belongs to the GRAPH, not to the engine: the
        workspace and the status word are allocated here (the engine's shared workspace is reallocated when a later call
        needs more bytes — a graph that pointed into it would replay on freed memory), and the graph keeps them, the
        inputs, the outputs and the weight handle alive. ``check_graph_status(graph)`` reads the graph's status word."""
        X, mask = self._f32(X), self._f32(mask)
        S, ridx, cenc, offsets = self._i32(S), self._i32(residue_idx), self._i32(chain_enc), self._i32(offsets)
        priv: dict = {}
        kw = dict(max_len=int(max_len), check_status=False, _private=priv, **want)
        w = self.weights_for(want.get("precision") or self.precision)
        cur = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            out = self.ssm_forward(X, S, mask, ridx, cenc, offsets, out=out, **kw)        # warm-up: buffers exist afterwards
        keep = [X, S, mask, ridx, cenc, offsets, priv["ws"], priv["status"], *out.values()]
        for t in keep:                      # allocated / first used on the side stream, replayed from others later
            t.record_stream(cur)
        cur.wait_stream(side)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.ssm_forward(X, S, mask, ridx, cenc, offsets, out=out, **kw)
        graph._tmpnn_keepalive = (keep, out, priv, w)                                      # captured pointers stay valid
        graph._tmpnn_status = priv["status"]
        return graph, out

    def check_graph_status(self, graph) -> None:
        """Raise if the last replay of a graph from ``capture_graph`` flagged a range / max_len problem (syncs)."""
        st = int(graph._tmpnn_status.item())
        if st:
            check(self.lib.tmpnn_status_error(st), "tmpnn_ssm_forward[graph]")

    def check_last_status(self) -> None:
        """Raise if the most recent ``ssm_forward(check_status=False)`` flagged a range / max_len problem (syncs)."""
        st = int(self._status.item())
        if st:
            check(self.lib.tmpnn_status_error(st), "tmpnn_ssm_forward")


# module-level gathers with the reference's signatures (protein_mpnn_utils.py:763-791)
def gather_nodes(nodes: torch.Tensor, neighbor_idx: torch.Tensor) -> torch.Tensor:
    _need_cuda(nodes, neighbor_idx)
    lib = _lib.load()
    B, N, C_ = nodes.shape
    K = neighbor_idx.shape[2]
    src = nodes.to(torch.float32).contiguous()
    idx = neighbor_idx.to(torch.int64).contiguous()
    out = torch.empty((B, idx.shape[1], K, C_), dtype=torch.float32, device=nodes.device)
    with torch.cuda.device(nodes.device):
        check(lib.tmpnn_gather_nodes(_ptr(src), _ptr(idx), B, N, K, C_, _ptr(out), _stream()), "tmpnn_gather_nodes")
    return out.to(nodes.dtype)


def gather_edges(edges: torch.Tensor, neighbor_idx: torch.Tensor) -> torch.Tensor:
    _need_cuda(edges, neighbor_idx)
    lib = _lib.load()
    B, N, _, C_ = edges.shape
    K = neighbor_idx.shape[2]
    src = edges.to(torch.float32).contiguous()
    idx = neighbor_idx.to(torch.int64).contiguous()
    out = torch.empty((B, N, K, C_), dtype=torch.float32, device=edges.device)
    with torch.cuda.device(edges.device):
        check(lib.tmpnn_gather_edges(_ptr(src), _ptr(idx), B, N, K, C_, _ptr(out), _stream()), "tmpnn_gather_edges")
    return out.to(edges.dtype)


def cat_neighbors_nodes(h_nodes, h_neighbors, E_idx):
    return torch.cat([h_neighbors, gather_nodes(h_nodes, E_idx)], -1)
