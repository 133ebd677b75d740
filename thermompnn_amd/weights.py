"""Weight formats of the ThermoMPNN hot path: names/shapes, a deterministic synthetic generator,
and loaders for the two on-disk formats the reference reads.

Reference formats:
  * vanilla ProteinMPNN: ``torch.load`` -> ``{'num_edges': int, 'model_state_dict': {...}}``
    (/root/reference/transfer_model.py:25-29)
  * ThermoMPNN Lightning checkpoint: ``ckpt['state_dict']`` with ``model.`` prefix
    (/root/reference/analysis/thermompnn_benchmarking.py:78-84, train_thermompnn.py:28-40)

The real checkpoints are not shipped with the reference mount, so parity is pinned with the synthetic
weights produced here (same generator on the build box and on the GPU box: numpy PCG64 is
platform-deterministic).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

H = 128          # hidden / node / edge width (transfer_model.py:19)
VOCAB = 21       # protein_mpnn_utils.py:1186
N_POS = 16       # num_positional_embeddings (protein_mpnn_utils.py:1085)
N_RBF = 16
EDGE_IN = N_POS + 25 * N_RBF   # 416 (protein_mpnn_utils.py:1097)
HEAD_IN = 3 * H  # num_final_layers(2)*128 + 128 (transfer_model.py:57)


def mpnn_param_shapes(num_enc: int = 3, num_dec: int = 3) -> "OrderedDict[str, tuple]":
    """Names and shapes of ProteinMPNN's state dict, in module-registration order
    (protein_mpnn_utils.py:1184-1215)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["features.embeddings.linear.weight"] = (N_POS, 66)
    s["features.embeddings.linear.bias"] = (N_POS,)
    s["features.edge_embedding.weight"] = (H, EDGE_IN)
    s["features.norm_edges.weight"] = (H,)
    s["features.norm_edges.bias"] = (H,)
    s["W_e.weight"] = (H, H)
    s["W_e.bias"] = (H,)
    s["W_s.weight"] = (VOCAB, H)

    def layer(prefix: str, num_in: int, edge_update: bool):
        norms = ("norm1", "norm2", "norm3") if edge_update else ("norm1", "norm2")
        for n in norms:
            s[f"{prefix}.{n}.weight"] = (H,)
            s[f"{prefix}.{n}.bias"] = (H,)
        lin = [("W1", H + num_in), ("W2", H), ("W3", H)]
        if edge_update:
            lin += [("W11", H + num_in), ("W12", H), ("W13", H)]
        for n, fan_in in lin:
            s[f"{prefix}.{n}.weight"] = (H, fan_in)
            s[f"{prefix}.{n}.bias"] = (H,)
        s[f"{prefix}.dense.W_in.weight"] = (4 * H, H)
        s[f"{prefix}.dense.W_in.bias"] = (4 * H,)
        s[f"{prefix}.dense.W_out.weight"] = (H, 4 * H)
        s[f"{prefix}.dense.W_out.bias"] = (H,)

    for i in range(num_enc):
        layer(f"encoder_layers.{i}", 2 * H, True)
    for i in range(num_dec):
        layer(f"decoder_layers.{i}", 3 * H, False)
    s["W_out.weight"] = (VOCAB, H)
    s["W_out.bias"] = (VOCAB,)
    return s


def head_param_shapes(hidden_dims=(64, 32), num_final_layers: int = 2, lightattn: bool = True) -> "OrderedDict[str, tuple]":
    """TransferModel's own parameters (transfer_model.py:57-73, 131-134) for any configuration its constructor accepts:
    input width 128 * num_final_layers + 128, LightAttention's two convolutions only when ``lightattn``."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    head_in = H * int(num_final_layers) + H
    if lightattn:
        for conv in ("feature_convolution", "attention_convolution"):
            s[f"light_attention.{conv}.weight"] = (head_in, head_in, 9)
            s[f"light_attention.{conv}.bias"] = (head_in,)
    sizes = [head_in, *hidden_dims, VOCAB]
    for i, (a, b) in enumerate(zip(sizes, sizes[1:])):
        s[f"both_out.{2 * i + 1}.weight"] = (b, a)   # ReLU sits at even indices
        s[f"both_out.{2 * i + 1}.bias"] = (b,)
    s["ddg_out.weight"] = (1, 1)
    s["ddg_out.bias"] = (1,)
    return s


def transfer_param_shapes(hidden_dims=(64, 32), num_final_layers: int = 2, lightattn: bool = True) -> "OrderedDict[str, tuple]":
    s: "OrderedDict[str, tuple]" = OrderedDict()
    for k, v in mpnn_param_shapes().items():
        s["prot_mpnn." + k] = v
    s.update(head_param_shapes(hidden_dims, num_final_layers, lightattn))
    return s


STYLES = ("xavier", "hot", "wide")


def _draw(rng: np.random.Generator, name: str, shape: tuple, style: str = "xavier") -> np.ndarray:
    """``style`` 'xavier': Xavier-uniform matrices, N(0, 0.02) biases, LayerNorm gamma 1 +- 0.1 / beta +- 0.1.
    'hot': a deliberately heavy draw for the accuracy margin of the split-precision kernels — matrices x 3, biases x 5,
    LayerNorm gamma uniform in [-2, 2] and beta in [-0.5, 0.5]: activations one to two orders of magnitude above the
    Xavier draw's, still finite in fp32 through the reference.
    'wide': the same LayerNorm parameters with matrices x 8 and N(0, 0.5) biases — Linear outputs reach 1e3..1e4 through the
    reference (ddG of +-1.4e4): the upper end of what the f16x2 matrix-core path can carry (fp16 ends at 65504); tests its
    ACCURACY near the range limit, not just the overflow flag."""
    hot = style in ("hot", "wide")
    leaf = name.rsplit(".", 1)[-1]
    if name.startswith("ddg_out"):
        return np.full(shape, 1.7 if leaf == "weight" else -0.3, dtype=np.float32)
    if ".norm" in name or "norm_edges" in name:
        if leaf == "weight":
            return (rng.uniform(-2.0, 2.0, shape) if hot else 1.0 + rng.uniform(-0.1, 0.1, shape)).astype(np.float32)
        return rng.uniform(-0.5, 0.5, shape).astype(np.float32) if hot else rng.uniform(-0.1, 0.1, shape).astype(np.float32)
    if leaf == "bias":
        return rng.normal(0.0, 0.5 if style == "wide" else 0.1 if hot else 0.02, shape).astype(np.float32)
    bound = math.sqrt(6.0 / (shape[0] + shape[1]))        # Xavier-uniform (protein_mpnn_utils.py:1217-1219); conv taps: all nine non-zero
    if hot:
        bound *= 8.0 if style == "wide" else 3.0
    return rng.uniform(-bound, bound, shape).astype(np.float32)


def synthetic_state_dict(seed: int = 0, which: str = "transfer", style: str = "xavier", head=None) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic synthetic weights. ``which`` = 'transfer' (full TransferModel tree) or 'mpnn'; ``style`` see _draw;
    ``head`` = dict(hidden_dims=..., num_final_layers=..., lightattn=...) for a non-default TransferModel head (the ProteinMPNN
    tensors come first, so they are the same draws whatever the head)."""
    if style not in STYLES:
        raise ValueError(f"style={style!r}: expected one of {STYLES}")
    rng = np.random.default_rng(seed)
    shapes = transfer_param_shapes(**(head or {})) if which == "transfer" else mpnn_param_shapes()
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in shapes.items():
        out[name] = torch.from_numpy(_draw(rng, name, shape, style))
    return out


def split_transfer_state_dict(sd):
    """-> (mpnn_sd without prefix, head_sd)."""
    mp, hd = OrderedDict(), OrderedDict()
    for k, v in sd.items():
        if k.startswith("prot_mpnn."):
            mp[k[len("prot_mpnn."):]] = v
        else:
            hd[k] = v
    return mp, hd


def save_vanilla_checkpoint(path, mpnn_sd, num_edges: int = 48) -> None:
    torch.save({"num_edges": int(num_edges), "model_state_dict": OrderedDict(mpnn_sd)}, path)


def load_vanilla_checkpoint(path):
    """-> (num_edges, state_dict). Mirrors transfer_model.py:25-29."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    return int(ckpt["num_edges"]), ckpt["model_state_dict"]


def save_lightning_checkpoint(path, transfer_sd) -> None:
    torch.save({"state_dict": OrderedDict(("model." + k, v) for k, v in transfer_sd.items())}, path)


class _Opaque:
    """Stand-in for every pickled global that is not a tensor / plain-container constructor: constructing it, calling it,
    setting its state or filling it does nothing. An OmegaConf config, an optimizer object or a hostile ``__reduce__``
    all come out as inert ``_Opaque`` instances — nothing from the file is imported or executed."""

    def __new__(cls, *a, **k):
        return object.__new__(cls)

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __setstate__(self, state):
        pass

    def __setitem__(self, k, v):
        pass

    def append(self, v):
        pass

    def extend(self, vs):
        pass

    def add(self, v):
        pass

    def update(self, *a, **k):
        pass


_SAFE_GLOBALS = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"), ("builtins", "set"), ("builtins", "frozenset"),
    ("builtins", "list"), ("builtins", "dict"), ("builtins", "tuple"), ("builtins", "int"), ("builtins", "float"),
    ("builtins", "complex"), ("builtins", "bytearray"), ("builtins", "slice"), ("builtins", "range"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch", "Size"), ("torch", "device"), ("torch", "dtype"),
    ("torch.serialization", "_get_layout"), ("torch._tensor", "_rebuild_from_type_v2"),
    # numpy arrays of plain dtypes / of Python strings (the reference's dataset_splits/*.pkl)
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
}


def _safe_pickle_module():
    """A ``pickle_module`` for ``torch.load`` / a plain loader whose Unpickler resolves ONLY the constructors above (plus
    torch's dtypes and storage classes, which torch's own wrapper handles) and maps every other global to ``_Opaque``."""
    import pickle
    import types

    class SafeUnpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if (module, name) in _SAFE_GLOBALS:
                return super().find_class(module, name)
            if module == "torch" and (name.endswith("Storage") or name in {d for d in dir(torch) if isinstance(getattr(torch, d), torch.dtype)}):
                return super().find_class(module, name)
            return _Opaque

    mod = types.ModuleType("tmpnn_safe_pickle")
    mod.Unpickler = SafeUnpickler
    mod.load = lambda f, **kw: SafeUnpickler(f, **kw).load()
    mod.loads = lambda b, **kw: SafeUnpickler(__import__("io").BytesIO(b), **kw).load()
    mod.__name__ = "pickle"
    return mod


def safe_unpickle(path):
    """Plain-pickle files of containers / numpy arrays (the reference's split dictionaries) without executing their globals."""
    with open(path, "rb") as fh:
        return _safe_pickle_module().load(fh)


def load_thermompnn_checkpoint(path, allow_pickle: bool = None):
    """Read a TransferModelPL checkpoint without importing Lightning (train_thermompnn.py:28-40 registers the
    TransferModel as ``self.model``): keeps ONLY the ``model.*`` entries of ``state_dict`` (Lightning-level buffers,
    metric states, optimizer state and ``hyper_parameters`` are dropped) and strips the prefix.

    The file is read with ``torch.load(weights_only=True)`` (tensors and plain containers only). Checkpoints whose
    ``hyper_parameters`` hold arbitrary Python objects (the published ``thermoMPNN_default.pt`` carries an OmegaConf
    config) fail that load; they are then read through a RESTRICTED unpickler that builds tensors and plain containers
    and turns every other pickled global into an inert placeholder — nothing from the file is imported or executed, so
    the reference's standard checkpoint loads out of the box (thermompnn_benchmarking.py:78-84 loads it with Lightning).
    ``allow_pickle=True`` / TMPNN_ALLOW_PICKLE=1 selects the unrestricted pickle loader instead (files you trust)."""
    import os
    import pickle
    if allow_pickle is None:
        allow_pickle = os.environ.get("TMPNN_ALLOW_PICKLE") == "1"
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
    except (pickle.UnpicklingError, RuntimeError) as e:
        if allow_pickle:
            ckpt = torch.load(path, map_location="cpu", weights_only=False)
        else:
            try:
                ckpt = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_safe_pickle_module())
            except Exception as e2:
                raise RuntimeError(f"{path}: not loadable with weights_only=True ({str(e).splitlines()[0][:200]}) nor with the "
                                   f"restricted unpickler ({str(e2).splitlines()[0][:200]}). If you trust this file, pass "
                                   "allow_pickle=True (or set TMPNN_ALLOW_PICKLE=1) to use the pickle loader, which can "
                                   "execute code embedded in the checkpoint.") from e2
    out = OrderedDict()
    if isinstance(ckpt, dict) and "state_dict" in ckpt:
        for k, v in ckpt["state_dict"].items():
            if k.startswith("model.") and isinstance(v, torch.Tensor):
                out[k[len("model."):]] = v
        if not out:
            raise KeyError(f"{path}: 'state_dict' has no 'model.*' entries (not a TransferModelPL checkpoint?)")
    elif isinstance(ckpt, dict):                # a bare TransferModel state dict
        for k, v in ckpt.items():
            if isinstance(v, torch.Tensor):
                out[k[len("model."):] if k.startswith("model.") else k] = v
    else:
        raise KeyError(f"{path}: not a checkpoint dictionary")
    return out


RAW_MAGIC = b"TMPNNRAW"


def export_raw(state_dict, path) -> int:
    """The weight set as a flat file a host WITHOUT Python can read (examples/scan_native.cpp): ``RAW_MAGIC``, int32 tensor
    count, then per tensor — in the library's canonical order, ``tmpnn_tensor_name(i)`` — int64 element count + little-endian
    fp32 data. Accepts a TransferModel state dict (``prot_mpnn.*`` + head: 130 tensors) or a bare ProteinMPNN one (118).
    -> bytes written."""
    import struct
    from . import _lib
    names = _lib.tensor_names()
    sd = dict(state_dict)
    with_head = all(n in sd for n in names[_lib.N_MPNN_TENSORS:])
    n = _lib.N_TENSORS if with_head else _lib.N_MPNN_TENSORS
    lib = _lib.load()
    total = 0
    with open(path, "wb") as fh:
        total += fh.write(RAW_MAGIC + struct.pack("<i", n))
        for i, name in enumerate(names[:n]):
            key = name if i >= _lib.N_MPNN_TENSORS else ("prot_mpnn." + name if ("prot_mpnn." + name) in sd else name)
            if key not in sd:
                raise KeyError(f"state dict lacks {key!r}")
            a = np.ascontiguousarray(sd[key].detach().cpu().numpy().astype("<f4"))
            if a.size != lib.tmpnn_tensor_numel(i):
                raise ValueError(f"{key}: {a.size} elements, the engine expects {lib.tmpnn_tensor_numel(i)}")
            total += fh.write(struct.pack("<q", a.size))
            total += fh.write(a.tobytes())
    return total


def _main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="checkpoint -> flat fp32 weight file for hosts that bind libtmpnn.so without Python")
    ap.add_argument("out")
    ap.add_argument("--model_path", default="", help="a TransferModelPL checkpoint (thermoMPNN_default.pt)")
    ap.add_argument("--vanilla_path", default="", help="the ProteinMPNN checkpoint (v_48_020.pt) the transfer model was trained on; "
                    "needed when the Lightning checkpoint does not carry prot_mpnn.*")
    ap.add_argument("--synthetic_weights", type=int, default=None, help="seed of a synthetic weight set instead of files")
    a = ap.parse_args(argv)
    if a.synthetic_weights is not None:
        sd = synthetic_state_dict(a.synthetic_weights)
    else:
        sd = load_thermompnn_checkpoint(a.model_path)
        if a.vanilla_path and not any(k.startswith("prot_mpnn.") for k in sd):
            for k, v in load_vanilla_checkpoint(a.vanilla_path)[1].items():
                sd["prot_mpnn." + k] = v
    print(f"{a.out}: {export_raw(sd, a.out)} bytes")


if __name__ == "__main__":
    _main()
