"""Synthetic workloads of the shapes BASELINE.json / SURVEY.md section 8(d) name.  Data only:
nothing here computes a lookup."""
from __future__ import annotations

import numpy as np

# Criteo-Kaggle categorical cardinalities (the 26 `--arch-embedding-size`
# values the reference's kaggle runs use, README.md:6,10,14).  Sum = 33 762 577 rows.
KAGGLE_ROWS = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194,
               27, 14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
# BASELINE config C4, "Criteo-Terabyte-shaped 26 tables (largest ~200M rows) dim=128".  The reference
# holds no such list ([EXT]); these are the categorical cardinalities of the
# Criteo Terabyte click logs without the 40M hashing cap, as the public DLRM recipes quote them:
# 882 774 559 rows, 452 GB at dim 128 fp32 -> ~56.5 GB per GPU on 8.
TERABYTE_ROWS = [227605432, 39060, 17295, 7424, 20265, 3, 7122, 1543, 63, 130229467, 3067956, 405282, 10,
                 2209, 11938, 155, 4, 976, 14, 292775614, 40790948, 187188510, 590152, 12973, 108, 36]
TERABYTE_DIM = 128
TERABYTE_BATCH = 16384

KAGGLE_DIM = 16       # --arch-sparse-feature-size=16
KAGGLE_BATCH = 39292  # --mini-batch-size=39292 (README.md:14)


def dlrm_table(rng: np.random.Generator, nr_rows: int, dim: int, dtype=np.float32) -> np.ndarray:
    """DLRM init W ~ U(-sqrt(1/n), sqrt(1/n))."""
    a = np.sqrt(1.0 / nr_rows)
    return rng.uniform(-a, a, size=(nr_rows, dim)).astype(dtype)


def uniform_indices(rng: np.random.Generator, nr_rows: int, n: int, dtype=np.uint32) -> np.ndarray:
    """load_generator.c:90: (uint32)((double)rand()/RAND_MAX*nr_rows), i.e. uniform over rows."""
    return rng.integers(0, nr_rows, size=n, dtype=np.int64).astype(dtype)


def zipf_indices(rng: np.random.Generator, nr_rows: int, n: int, alpha: float = 1.2,
                 dtype=np.uint32, permute: bool = True) -> np.ndarray:
    """Zipf(alpha)-distributed ranks over [0, nr_rows) (continuous bounded-Pareto inverse CDF:
    vectorised, exact enough for a traffic pattern), scattered over the table by a fixed
    multiplicative hash so that hot rows are not neighbours in memory."""
    u = rng.random(n)
    if abs(alpha - 1.0) < 1e-9:
        ranks = np.exp(u * np.log(nr_rows)) - 1.0
    else:
        oma = 1.0 - alpha
        ranks = np.power((nr_rows ** oma - 1.0) * u + 1.0, 1.0 / oma) - 1.0
    ranks = np.minimum(ranks.astype(np.int64), nr_rows - 1)
    if permute:
        ranks = (ranks * 2654435761 + 12345) % nr_rows
    return ranks.astype(dtype)


def fixed_offsets(n_bags: int, pooling: int, dtype=np.uint32) -> np.ndarray:
    """load_generator.c:88: offsets[i] = i * indices_per_batch."""
    return (np.arange(n_bags, dtype=np.int64) * pooling).astype(dtype)


def ragged_offsets(rng: np.random.Generator, n_bags: int, max_len: int, p_empty: float = 0.1,
                   dtype=np.uint32):
    lens = rng.integers(1, max_len + 1, size=n_bags)
    lens[rng.random(n_bags) < p_empty] = 0
    off = np.zeros(n_bags, dtype=np.int64)
    if n_bags > 1:
        off[1:] = np.cumsum(lens)[:-1]
    return off.astype(dtype), int(lens.sum())


def algorithmic_bytes(n_idx: int, n_bags: int, dim: int, elem: int, idx_bytes: int,
                      has_offsets: bool = True) -> int:
    """n_idx*(D*elem + idx) + n_bags*off + n_bags*D*4."""
    return n_idx * (dim * elem + idx_bytes) + (n_bags * idx_bytes if has_offsets else 0) + n_bags * dim * 4


def table_set(name: str, rows_scale: float = 1.0):
    """(rows per table, dim, default bags per table per rank, label) of a named multi-GPU table set:
    "c2" = the 26 Criteo-Kaggle tables (BASELINE configs[1]); "c4" = the Terabyte-shaped set
    (configs[3]).  rows_scale < 1 shrinks every table (at least one row) so that an N-rank layout can
    be rehearsed on fewer GPUs."""
    if name == "c4":
        rows, dim, batch, label = TERABYTE_ROWS, TERABYTE_DIM, TERABYTE_BATCH, "C4: 26 Criteo-Terabyte-shaped tables"
    elif name == "c5":
        # BASELINE configs[4]: 512 tables x 50M rows x dim 64 fp16 = 3.28 TB does not fit 8 x 288 GB; rows scaled to 30M
        # (1.97 TB, 246 GB per GPU at 8), as the single-GPU `--workload c5` share does (64 of these tables)
        rows, dim, batch, label = [30_000_000] * 512, 64, 16384, "C5: 512 tables x 30M rows (50M as written does not fit), fp16"
    else:
        rows, dim, batch, label = KAGGLE_ROWS, KAGGLE_DIM, KAGGLE_BATCH, "C2: 26 Criteo-Kaggle tables"
    if rows_scale != 1.0:
        rows = [max(1, int(n * rows_scale)) for n in rows]
        label += " (rows x %g)" % rows_scale
    return list(rows), dim, batch, label


# per table set: table dtype, default indices per bag, default index distribution ("mixed" = Zipf(1.2) on even tables,
# uniform on odd ones -- configs[4]'s "mixed hot/cold")
TABLE_SET_EXTRAS = {"c2": dict(dtype="f32", pooling=1, dist="uniform"), "c4": dict(dtype="f32", pooling=1, dist="uniform"),
                    "c5": dict(dtype="f16", pooling=32, dist="mixed")}


def top_rows(indices: np.ndarray, k: int, min_share: float = 0.0) -> np.ndarray:
    """The k most frequent row ids of an index sample, most frequent first (for emb_set_hot_rows).
    min_share > 0: return no rows unless those k cover at least that share of the sample -- a table
    with near-uniform accesses gains nothing from an LDS copy of k of its rows."""
    indices = np.asarray(indices)
    ids, counts = np.unique(indices, return_counts=True)
    order = np.argsort(-counts, kind="stable")[:k]
    if min_share > 0.0 and counts[order].sum() < min_share * max(indices.shape[0], 1):
        return np.zeros(0, dtype=np.uint64)
    return ids[order].astype(np.uint64)
