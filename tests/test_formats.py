"""CPU: on-disk formats either side of the path (synthetic files written by the test)."""
import numpy as np
import pytest


def _formats():
    from importlib import import_module
    import pim_embedding_lookup_amd  # noqa: F401
    return import_module("pim-embedding-lookup_amd.formats")


def test_criteo_npz_batches(tmp_path):
    fm = _formats()
    rng = np.random.default_rng(0)
    counts = np.array([7, 3, 1000, 12], dtype=np.int32)
    N = 50
    x_cat = np.stack([rng.integers(0, c, size=N) for c in counts], axis=1).astype(np.int32)
    p = tmp_path / "kaggle_processed.npz"
    np.savez(p, X_int=rng.integers(0, 9, size=(N, 13)), X_cat=x_cat, y=rng.integers(0, 2, size=N), counts=counts)
    d = fm.CriteoKaggleNpz(str(p), mmap=False)
    assert d.table_rows == [7, 3, 1000, 12] and d.n_samples == N
    lS_o, lS_i = d.batch(10, 16)
    assert len(lS_o) == len(lS_i) == 4
    for k in range(4):
        assert np.array_equal(lS_o[k], np.arange(16)) and lS_i[k].dtype == np.int64
        assert np.array_equal(lS_i[k], x_cat[10:26, k])
    sizes = [o[0].shape[0] for o, _ in d.batches(16)]
    assert sizes == [16, 16, 16, 2]                      # ragged tail batch
    lS_o32, lS_i32 = d.batch(0, 8, index_dtype=np.uint32)
    assert lS_i32[0].dtype == np.uint32
    bad = x_cat.copy(); bad[3, 1] = 3                     # index == count: out of range
    np.savez(p, X_cat=bad, counts=counts)
    with pytest.raises(ValueError):
        fm.CriteoKaggleNpz(str(p), mmap=False).batch(0, 8)
    np.savez(p, X_int=np.zeros(3))
    with pytest.raises(ValueError):
        fm.CriteoKaggleNpz(str(p), mmap=False)


def test_dlrm_checkpoint_round_trip(tmp_path):
    fm = _formats()
    rng = np.random.default_rng(1)
    tabs = [rng.standard_normal((n, 16)).astype(np.float32) for n in (5, 300, 17)]
    p = tmp_path / "model.pt"
    fm.save_dlrm_embedding_weights(str(p), tabs)
    got = fm.load_dlrm_embedding_weights(str(p))
    assert len(got) == 3 and all(np.array_equal(a, b) for a, b in zip(got, tabs))
    import torch
    torch.save({"bot_l.0.weight": torch.zeros(2, 2)}, p)
    with pytest.raises(ValueError):
        fm.load_dlrm_embedding_weights(str(p))
