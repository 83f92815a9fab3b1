"""`nn.EmbeddingBag(mode="sum")`-shaped modules over the engine -- the operator surface the reference
plugs into.  dlrm_s_pytorch.py::create_emb builds `emb_l = nn.ModuleList([nn.EmbeddingBag(n, m,
mode="sum", sparse=True) ...])` and apply_emb calls `emb_l[k](indices_k, offsets_k)` per table [EXT:
that file is an empty submodule here; call shape from README.md:6,10,14].  The reference's fork reroutes
exactly that call to populate_mram / lookup.  Here:

    emb_l[k] = EmbeddingBag.from_torch(emb_l[k])          # one table, same forward(input, offsets)
    ebc = FusedEmbeddingBags.from_torch(emb_l)            # all tables, ONE launch per apply_emb

Inference only (the engine has no backward), `mode="sum"` only, no per-sample weights / padding_idx /
max_norm -- the reference has none of them either.

Input checking.  The C ABI is as unchecked as the reference (an out-of-range index is a wild read there,
emb_dpu_lookup.c:113); these modules are what user tensors reach first, so by default every forward goes through
emb_lookup_batched_checked: indices / offsets are validated on the GPU first and IndexError is raised, like
nn.EmbeddingBag does, before anything is launched.  That costs one small kernel (7 us) and a wait for its one-word verdict per call (no
allocation, no device-wide synchronize); pass `trusted_inputs=True` (constructor or attribute) for the unchecked
fast path once the producer of the indices is known to be sound, or `deferred_check=True` to keep the check (a bad call's
lookup is still kept from running, on the GPU) without the wait: the IndexError then comes out of a LATER forward, of
`engine.check_report()` or of `engine.close()`, naming the call -- the wait inside the call is what a checked forward costs
on host-bound shapes (26 tables x 2048 bags: 14.4 us checked, 9.7 deferred, 6.6 unchecked; apply_emb with per-table lists: 31.5 / 20.7 /
17.5; 26 x 39 292 bags is GPU-bound: 28.8 either way, 20.9 unchecked -- profiles/r06/checked_calls/).

Checkpoints.  `state_dict()` carries `<prefix>weight` read out of HBM and `load_state_dict` uploads it, so a
DLRM whose `emb_l[k]` were swapped for these modules saves / loads the same keys and shapes as before."""
from __future__ import annotations

import threading

import torch

from . import lib as _l
from .engine import EmbeddingEngine

_engines: dict[int, EmbeddingEngine] = {}
_next_id: dict[int, int] = {}
_lock = threading.Lock()
_MAX_TABLES = 4096


def default_engine(device: int = 0) -> EmbeddingEngine:
    """One shared engine per GPU for the modules below (created on first use)."""
    with _lock:
        e = _engines.get(device)
        if e is None:
            e = _engines[device] = EmbeddingEngine(device=device, max_tables=_MAX_TABLES)
            _next_id[device] = 0
        return e


def _new_table_id(device: int) -> int:
    with _lock:
        t = _next_id[device]
        if t >= _MAX_TABLES:
            raise RuntimeError("default engine is full; pass engine= / table_id= explicitly")
        _next_id[device] = t + 1
        return t


def _bags_from(input, offsets, include_last_offset: bool):
    """torch's calling conventions -> (1-D indices, 1-D bag starts)."""
    if input.dim() == 2:
        if offsets is not None:
            raise ValueError("if input is 2D, then offsets has to be None")     # torch's own message
        B, L = input.shape
        return input.reshape(-1), torch.arange(0, B * L, L, dtype=input.dtype, device=input.device)
    if input.dim() != 1 or offsets is None or offsets.dim() != 1:
        raise ValueError("input has to be 1D with 1D offsets, or 2D without offsets")
    if offsets.dtype != input.dtype:
        offsets = offsets.to(input.dtype)
    if include_last_offset:
        offsets = offsets[:-1]      # the extra entry must be len(input): the last bag runs to the end either way
    return input, offsets


class EmbeddingBag(torch.nn.Module):
    """One table.  forward(input, offsets) -> [B, embedding_dim] fp32, like nn.EmbeddingBag(mode="sum")."""

    def __init__(self, num_embeddings: int, embedding_dim: int, mode: str = "sum", sparse: bool = False,
                 _weight=None, include_last_offset: bool = False, device: int = 0, engine: EmbeddingEngine | None = None,
                 table_id: int | None = None, dtype=torch.float32, trusted_inputs: bool = False, deferred_check: bool = False):
        super().__init__()
        self.trusted_inputs = bool(trusted_inputs)
        self.deferred_check = bool(deferred_check)
        self._table_dtype = dtype
        if mode != "sum":
            raise NotImplementedError("only mode='sum' (the reference pools by summation, emb_dpu_lookup.c:114)")
        self.num_embeddings, self.embedding_dim = int(num_embeddings), int(embedding_dim)
        self.mode, self.sparse, self.include_last_offset = mode, sparse, include_last_offset
        self.engine = engine if engine is not None else default_engine(device)
        self.device_index = self.engine.device
        self.table_id = table_id if table_id is not None else _new_table_id(self.device_index)
        self._id_list = [self.table_id]
        dev = torch.device("cuda", self.device_index)
        if _weight is None:      # nn.EmbeddingBag's default init is N(0, 1)
            _weight = torch.randn((self.num_embeddings, self.embedding_dim), device=dev)
        w = torch.as_tensor(_weight)
        if tuple(w.shape) != (self.num_embeddings, self.embedding_dim):
            raise ValueError(f"weight shape {tuple(w.shape)} != ({self.num_embeddings}, {self.embedding_dim})")
        self.engine.load_table(self.table_id, w.detach().to(dtype).contiguous())

    @classmethod
    def from_pretrained(cls, embeddings, mode: str = "sum", include_last_offset: bool = False, **kw):
        return cls(embeddings.shape[0], embeddings.shape[1], mode=mode, _weight=embeddings,
                   include_last_offset=include_last_offset, **kw)

    @classmethod
    def from_torch(cls, module: torch.nn.EmbeddingBag, **kw):
        if module.padding_idx is not None or module.max_norm is not None:
            raise NotImplementedError("padding_idx / max_norm are not part of the reference path")
        return cls(module.num_embeddings, module.embedding_dim, mode=module.mode, sparse=module.sparse,
                   _weight=module.weight.detach(), include_last_offset=module.include_last_offset, **kw)

    @property
    def weight(self):
        """The table's rows in HBM as a torch tensor (zero-copy view; not a Parameter -- inference only)."""
        return self.engine.table_tensor(self.table_id)

    def forward(self, input, offsets=None, per_sample_weights=None):
        if per_sample_weights is not None:
            raise NotImplementedError("per_sample_weights are not part of the reference path")
        idx, off = _bags_from(input, offsets, self.include_last_offset)
        idx, off = idx.contiguous(), off.contiguous()
        check = False if self.trusted_inputs else ("deferred" if self.deferred_check else True)
        return self.engine.lookup_batched(self._id_list, [idx], [off], check=check)[0]

    # ---- checkpoints: the same key and shape as nn.EmbeddingBag ("<prefix>weight", [num_embeddings, dim]) ----
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        destination[prefix + "weight"] = self.weight.detach().clone()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        key = prefix + "weight"
        if key not in state_dict:
            if strict:
                missing_keys.append(key)
            return
        w = state_dict[key]
        if tuple(w.shape) != (self.num_embeddings, self.embedding_dim):
            error_msgs.append(f"size mismatch for {key}: copying a param with shape {tuple(w.shape)} from checkpoint, "
                              f"the shape in current model is ({self.num_embeddings}, {self.embedding_dim}).")
            return
        self.engine.load_table(self.table_id, w.detach().to(self._table_dtype).contiguous())

    def extra_repr(self) -> str:
        return f"{self.num_embeddings}, {self.embedding_dim}, mode='sum', table_id={self.table_id}, engine=MI355X"


class FusedEmbeddingBags(torch.nn.Module):
    """All tables of a model: forward(lS_o, lS_i) -> list of [B, m] with ONE fused launch (apply_emb)."""

    def __init__(self, bags, trusted_inputs: bool | None = None, deferred_check: bool | None = None):
        super().__init__()
        self.bags = torch.nn.ModuleList(bags)
        # unchecked only if every table says so (or the caller does); the same for the deferred verdict
        self.trusted_inputs = all(b.trusted_inputs for b in bags) if trusted_inputs is None else bool(trusted_inputs)
        self.deferred_check = all(b.deferred_check for b in bags) if deferred_check is None else bool(deferred_check)
        engines = {id(b.engine) for b in self.bags}
        if len(engines) != 1:
            raise ValueError("all tables of a FusedEmbeddingBags must live in one engine")
        self.engine = self.bags[0].engine
        self._ids = [b.table_id for b in self.bags]

    @classmethod
    def from_torch(cls, emb_l, **kw):
        return cls([EmbeddingBag.from_torch(m, **kw) for m in emb_l])

    def forward(self, lS_o, lS_i):
        check = False if self.trusted_inputs else ("deferred" if self.deferred_check else True)
        if hasattr(lS_i, "dim") and lS_i.dim() == 2 and hasattr(lS_o, "dim") and lS_o.dim() == 2 and lS_i.is_cuda:
            return list(self.engine.lookup_stacked(self._ids, lS_i, lS_o, check=check).unbind(0))
        idx, off = [], []
        for b, i, o in zip(self.bags, lS_i, lS_o):
            i1, o1 = _bags_from(i, o, b.include_last_offset)
            idx.append(i1.contiguous())
            off.append(o1.contiguous())
        return self.engine.lookup_batched(self._ids, idx, off, check=check)

    apply_emb = forward


__all__ = ["EmbeddingBag", "FusedEmbeddingBags", "default_engine"]
_ = _l  # (lib is imported for its side effect: torch first, then libpimemb.so)
