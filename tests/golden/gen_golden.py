#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (run in the BUILD container only).

Expected outputs come from torch CPU `torch.nn.functional.embedding_bag(mode="sum")` -- the routine
the north star names as the parity target (nn.EmbeddingBag sum; the reference reaches it through
`dlrm_s_pytorch.py`, README.md:6,10,14, whose source is an empty submodule in this checkout) -- and,
for the toy case, from the reference's one known-answer vector (upmem/c_test.py:40,55-57).

Nothing here runs on the GPU box: the .npz files are data (inputs + expected outputs) and travel
with the repo; this script is committed so the fixtures can be regenerated and audited.

    python tests/golden/gen_golden.py          # rewrites tests/golden/*.npz
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))

# Criteo-Kaggle categorical cardinalities
KAGGLE_ROWS = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194,
               27, 14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]


def torch_bag_sum(table_f32, indices, offsets):
    w = torch.from_numpy(np.ascontiguousarray(table_f32, dtype=np.float32))
    i = torch.from_numpy(indices.astype(np.int64))
    o = torch.from_numpy(offsets.astype(np.int64))
    with torch.no_grad():
        return F.embedding_bag(i, w, o, mode="sum").numpy().copy()


def ragged_offsets(rng, n_bags, max_len, p_empty=0.15):
    lens = rng.integers(1, max_len + 1, size=n_bags)
    lens[rng.random(n_bags) < p_empty] = 0
    off = np.zeros(n_bags, dtype=np.int64)
    off[1:] = np.cumsum(lens)[:-1]
    return off, int(lens.sum())


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path)} bytes")


def main():
    torch.set_num_threads(1)
    torch.manual_seed(0)

    # ---- 1. reference known-answer vector: upmem/c_test.py:40 (table) and :55-57 (bags) --------
    table = np.array([[(r + 1) * (c + 1) for c in range(8)] for r in range(4)], dtype=np.int32)
    nb = 32
    indices = np.tile(np.array([1, 3, 2, 0], dtype=np.uint32), nb)
    offsets = (np.arange(nb) * 4).astype(np.uint32)
    expect_int = np.tile(np.arange(10, 90, 10, dtype=np.int32), (nb, 1))  # [10..80] per bag
    expect_f32 = torch_bag_sum(table.astype(np.float32), indices, offsets)
    assert (expect_f32 == expect_int.astype(np.float32)).all()
    save("toy_ctest", table_i32=table, indices=indices, offsets=offsets,
         expect_int=expect_int, expect_f32=expect_f32)

    # ---- 2. torch EmbeddingBag(sum) fixtures: D x pooling x index width x init scale ----------
    rng = np.random.default_rng(1234)
    case = 0
    for dim in (16, 64, 128):
        for kind in ("one_hot", "fixed32", "ragged"):
            for scale in ("dlrm", "unit"):
                n_rows = int(rng.integers(50, 400))
                n_bags = int(rng.integers(5, 40))
                if scale == "dlrm":  # DLRM init U(-sqrt(1/n), sqrt(1/n))
                    a = np.sqrt(1.0 / n_rows)
                    tab = rng.uniform(-a, a, size=(n_rows, dim)).astype(np.float32)
                else:
                    tab = rng.standard_normal((n_rows, dim)).astype(np.float32)
                if kind == "one_hot":
                    off = np.arange(n_bags, dtype=np.int64)
                    n_idx = n_bags
                elif kind == "fixed32":
                    off = np.arange(n_bags, dtype=np.int64) * 32
                    n_idx = n_bags * 32
                else:
                    off, n_idx = ragged_offsets(rng, n_bags, 40)
                idx = rng.integers(0, n_rows, size=n_idx)
                exp = torch_bag_sum(tab, idx, off)
                save(f"bag_{case:02d}_d{dim}_{kind}_{scale}", table=tab,
                     indices=idx.astype(np.int64), offsets=off.astype(np.int64), expect=exp)
                case += 1

    # ---- 3. edge cases the domain has: all-empty bags, trailing empty bags, one giant bag,
    #         duplicate indices, single row table ----------------------------------------------
    tab = rng.standard_normal((7, 16)).astype(np.float32)
    off = np.zeros(5, dtype=np.int64)
    idx = np.zeros(0, dtype=np.int64)
    save("edge_all_empty", table=tab, indices=idx, offsets=off, expect=torch_bag_sum(tab, idx, off))
    off = np.array([0, 3, 3, 5, 5, 5], dtype=np.int64)
    idx = np.array([6, 0, 6, 2, 2], dtype=np.int64)
    save("edge_trailing_empty_dups", table=tab, indices=idx, offsets=off,
         expect=torch_bag_sum(tab, idx, off))
    tab1 = rng.standard_normal((1, 64)).astype(np.float32)
    off = np.array([0], dtype=np.int64)
    idx = np.zeros(1000, dtype=np.int64)
    save("edge_one_giant_bag_single_row", table=tab1, indices=idx, offsets=off,
         expect=torch_bag_sum(tab1, idx, off))

    # ---- 4. fp16 storage, fp32 accumulate (BASELINE config C5) --------------------------------
    tab16 = rng.standard_normal((300, 64)).astype(np.float16)
    off, n_idx = ragged_offsets(rng, 33, 20)
    idx = rng.integers(0, 300, size=n_idx)
    save("f16_d64_ragged", table=tab16, indices=idx.astype(np.int64), offsets=off,
         expect=torch_bag_sum(tab16.astype(np.float32), idx, off))

    # ---- 5. C1-shaped plumbing case: 26 Kaggle tables, D=16, B=4, one index per bag; row counts
    #         capped at 512 so the fixture stays small (ragged tiny tables 3,4,10,... kept) -----
    arrays = {}
    B = 4
    for t, n in enumerate(KAGGLE_ROWS):
        n = min(n, 512)
        a = np.sqrt(1.0 / n)
        tab = rng.uniform(-a, a, size=(n, 16)).astype(np.float32)
        idx = rng.integers(0, n, size=B)
        off = np.arange(B, dtype=np.int64)
        arrays[f"table_{t}"] = tab
        arrays[f"indices_{t}"] = idx.astype(np.int64)
        arrays[f"offsets_{t}"] = off
        arrays[f"expect_{t}"] = torch_bag_sum(tab, idx, off)
    save("kaggle26_capped_b4", **arrays)

    # ---- 6. row widths off the 16-byte grid (dlrm_s_pytorch.py's default sparse feature size is 2) and
    #         beyond 1 KiB: the any-dim kernels.  Own generator, so sections 1-5 stay byte-identical. ----
    rng6 = np.random.default_rng(4321)
    for dim, dt in ((2, np.float32), (3, np.float32), (10, np.float32), (30, np.float32), (300, np.float32),
                    (6, np.float16), (5, np.float16)):
        n_rows, n_bags = int(rng6.integers(40, 300)), int(rng6.integers(10, 50))
        tab = rng6.standard_normal((n_rows, dim)).astype(dt)
        off, n_idx = ragged_offsets(rng6, n_bags, 12)
        idx = rng6.integers(0, n_rows, size=n_idx)
        save(f"anydim_d{dim}_{np.dtype(dt).name}_ragged", table=tab, indices=idx.astype(np.int64), offsets=off,
             expect=torch_bag_sum(tab.astype(np.float32), idx, off))


if __name__ == "__main__":
    main()
