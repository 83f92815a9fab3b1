"""On-disk formats either side of the hot path.

The reference's CLI names them (`--processed-data-file=…kaggleAdDisplayChallenge_processed.npz`,
`--load-model=…`, README.md:6,10; upmem/run.sh:117-118; data/model dirs in .gitignore:149,152) but
ships neither files nor readers (they live in the empty `PIM-dlrm-new` submodule).  Formats below
follow upstream facebookresearch/dlrm [EXT, not verifiable from the checkout]:

  * processed Criteo-Kaggle `.npz`: arrays `X_int` [N,13], `X_cat` [N,26] (already re-indexed to
    0..count-1 per feature), `y` [N], `counts` [26] (cardinality per categorical feature);
  * DLRM checkpoint `.pt`: a dict with `state_dict` (or the state dict itself) holding
    `emb_l.<k>.weight` [N_k, D] per table.

Only what the embedding path needs is read: indices/offsets per table, table shapes, weights."""
from __future__ import annotations

import numpy as np


class CriteoKaggleNpz:
    """Categorical side of a processed Criteo-Kaggle file as embedding-lookup batches."""

    def __init__(self, path: str, mmap: bool = True):
        z = np.load(path, mmap_mode="r" if mmap else None)
        if "X_cat" not in z or "counts" not in z:
            raise ValueError(f"{path}: expected arrays X_cat and counts (processed DLRM npz)")
        self.x_cat = z["X_cat"]
        self.counts = [int(c) for c in z["counts"]]
        if self.x_cat.ndim != 2 or self.x_cat.shape[1] != len(self.counts):
            raise ValueError("X_cat must be [N, len(counts)]")
        self.n_samples = int(self.x_cat.shape[0])

    @property
    def table_rows(self) -> list[int]:
        """`--arch-embedding-size` as DLRM derives it from `counts`."""
        return list(self.counts)

    def batch(self, start: int, size: int, index_dtype=np.int64):
        """(lS_o, lS_i) for samples [start, start+size): one index per bag (Criteo is one-hot), so
        lS_o[k] = arange(B) and lS_i[k] = X_cat[start:start+B, k] -- the layout
        `dlrm_s_pytorch.py::apply_emb` consumes [EXT]."""
        stop = min(start + size, self.n_samples)
        x = np.asarray(self.x_cat[start:stop])
        if x.size and (x.min() < 0 or (x.max(axis=0) >= np.asarray(self.counts)).any()):
            raise ValueError("X_cat holds an index outside [0, counts[k])")
        B = stop - start
        off = np.arange(B, dtype=index_dtype)
        return [off] * x.shape[1], [np.ascontiguousarray(x[:, k]).astype(index_dtype) for k in range(x.shape[1])]

    def batches(self, batch_size: int, index_dtype=np.int64):
        for s in range(0, self.n_samples, batch_size):
            yield self.batch(s, batch_size, index_dtype)


def load_dlrm_embedding_weights(path: str):
    """Embedding tables of a DLRM checkpoint: list of float32 [N_k, D] numpy arrays in table order."""
    import torch
    obj = torch.load(path, map_location="cpu", weights_only=True)
    sd = obj.get("state_dict", obj) if isinstance(obj, dict) else obj
    keys = sorted((k for k in sd if k.startswith("emb_l.") and k.endswith(".weight")),
                  key=lambda k: int(k.split(".")[1]))
    if not keys:
        raise ValueError(f"{path}: no emb_l.<k>.weight entries")
    if [int(k.split(".")[1]) for k in keys] != list(range(len(keys))):
        raise ValueError("emb_l indices are not 0..T-1")
    return [sd[k].detach().to(torch.float32).contiguous().numpy() for k in keys]


def save_dlrm_embedding_weights(path: str, tables) -> None:
    """Write tables back in the same layout (tests, round trips)."""
    import torch
    torch.save({"state_dict": {f"emb_l.{k}.weight": torch.as_tensor(np.asarray(t)) for k, t in enumerate(tables)}},
               path)
