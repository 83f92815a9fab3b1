"""CPU: the host-side arithmetic of the counts-first exchange that lives in the C ABI (emb_route_exchange_sizes,
emb_route_serve_descs: no GPU work, no engine) against a numpy restatement of the layout rule in include/pimemb.h."""
import ctypes as C

import numpy as np
import pytest


@pytest.mark.parametrize("N,K,dim,seed", [(1, 1, 4, 0), (2, 5, 16, 1), (8, 8, 128, 2), (5, 64, 64, 3), (200, 2, 8, 4)])
def test_exchange_sizes_and_serve_descriptors(pel, N, K, dim, seed):
    L = pel.lib.load()
    rng = np.random.default_rng(seed)

    def counts():
        c = np.zeros((N, K + 1, 2), dtype=np.uint32)
        c[:, :K, 0] = rng.integers(0, 50, size=(N, K)) * (rng.random((N, K)) > 0.3)       # some empty pieces
        c[:, :K, 1] = c[:, :K, 0] * rng.integers(1, 9, size=(N, K))                        # >= one index per sub-bag
        return c

    pad4 = lambda v: (v + 3) // 4 * 4
    sent, received = counts(), counts()
    words = lambda c: (pad4(c[:, :K, 0].astype(np.int64)) + pad4(c[:, :K, 1].astype(np.int64))).sum(axis=1)
    # the peaks entry: every sender's largest request piece (words) / most partial rows from one peer, the same in every
    # message it sends; what a rank RECEIVES is one such pair per source
    received[:, K, 0] = rng.integers(0, 10_000, size=N)
    received[:, K, 1] = rng.integers(0, 5_000, size=N)
    both = np.ascontiguousarray(np.stack([sent, received]))
    out = (C.c_uint64 * (4 * N))()
    pr, pt = C.c_uint64(), C.c_uint64()
    base = C.addressof(out)
    rc = L.emb_route_exchange_sizes(both.ctypes.data, both.ctypes.data + sent.nbytes, K, N, dim, base, base + 8 * N,
                                    base + 16 * N, base + 24 * N, C.byref(pr), C.byref(pt))
    assert rc == 0
    v = np.array(out[:], dtype=np.int64).reshape(4, N)
    assert np.array_equal(v[0], words(sent)) and np.array_equal(v[1], words(received))
    assert np.array_equal(v[2], sent[:, :K, 0].sum(axis=1)) and np.array_equal(v[3], received[:, :K, 0].sum(axis=1))
    assert pr.value == int(received[:, K, 0].max()) * 4 and pt.value == int(received[:, K, 1].max()) * dim * 4

    ids = (C.c_uint32 * K)(*range(100, 100 + K))
    descs = (pel.lib.EmbLookupDesc * (N * K))()
    n, nbytes = C.c_uint32(), C.c_uint64()
    recv_base, ret_base = 0x10000000, 0x7000000000
    assert L.emb_route_serve_descs(received.ctypes.data, K, N, dim, ids, recv_base, ret_base, descs, C.byref(n),
                                   C.byref(nbytes)) == 0
    start = row0 = want_bytes = j = 0
    for s in range(N):
        for k in range(K):
            ns, ni = int(received[s, k, 0]), int(received[s, k, 1])
            if ns:
                d = descs[j]
                j += 1
                assert (d.table_id, d.fixed_pooling, d.n_bags, d.n_indices) == (100 + k, 0, ns, ni)
                assert d.offsets == recv_base + 4 * start and d.indices == recv_base + 4 * (start + pad4(ns))
                assert d.pooled == ret_base + 4 * dim * row0
                want_bytes += ni * (dim * 4 + 4) + ns * (4 + dim * 4)
            start += pad4(ns) + pad4(ni)
            row0 += ns
    assert n.value == j and nbytes.value == want_bytes
    assert start == int(v[1].sum()) and row0 == int(v[3].sum())


def test_exchange_arithmetic_rejects_bad_arguments(pel):
    L = pel.lib.load()
    z = (C.c_uint64 * 8)()
    assert L.emb_route_exchange_sizes(None, None, 1, 1, 4, z, z, z, z, None, None) == pel.lib.EMB_ERR_INVALID
    c = np.zeros((1, 2, 2), dtype=np.uint32)
    assert L.emb_route_exchange_sizes(c.ctypes.data, c.ctypes.data, 0, 1, 4, C.addressof(z), C.addressof(z), C.addressof(z),
                                      C.addressof(z), None, None) == pel.lib.EMB_ERR_INVALID
    assert L.emb_route_serve_descs(c.ctypes.data, 1, 1, 0, None, None, None, None, None, None) == pel.lib.EMB_ERR_INVALID
