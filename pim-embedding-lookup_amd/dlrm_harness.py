"""`apply_emb`-shaped harness over the engine.

Stands where `dlrm_s_pytorch.py::apply_emb` stands in the reference's stack [EXT: that file is an
empty submodule here; flags from README.md:6,10,14 and upmem/run.sh:72-82,111-121]: a list of
`nn.EmbeddingBag(n_k, m, mode="sum")` called per table with (indices_k, offsets_k), each returning
[B, m].  Here all tables are served by ONE fused HIP launch.  No autograd (inference only, as the
reference's `--inference-only` runs), no MLPs -- the rest of the model is out of scope.

    python -m pim_embedding_lookup_amd.dlrm_harness --arch-sparse-feature-size=16 \
        --arch-embedding-size=1460-583-10131227-… --mini-batch-size=39292 --inference-only
"""
from __future__ import annotations

import argparse
import sys
import time

import numpy as np

from . import workloads
from .engine import EmbeddingEngine


class EmbeddingBagCollection:
    """emb_l of a DLRM: T tables of dim m in HBM; forward(lS_o, lS_i) -> list of [B, m]."""

    def __init__(self, ln_emb, m_spa: int, device: int = 0, weights=None, seed: int = 0, dtype="f32",
                 trusted_inputs: bool = False, deferred_check: bool = False):
        import torch
        self.torch = torch
        self.trusted_inputs = bool(trusted_inputs)     # False: every apply_emb checks its indices first (IndexError)
        self.deferred_check = bool(deferred_check)     # ... without waiting for the verdict (a later call / close() raises it)
        self.device = torch.device("cuda", device)
        self.ln_emb, self.m = [int(n) for n in ln_emb], int(m_spa)
        self.engine = EmbeddingEngine(device=device, max_tables=len(self.ln_emb))
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)
        for k, n in enumerate(self.ln_emb):
            if weights is not None:
                w = torch.as_tensor(np.asarray(weights[k]), dtype=torch.float32)
                if tuple(w.shape) != (n, self.m):
                    raise ValueError(f"table {k}: weight shape {tuple(w.shape)} != ({n}, {self.m})")
            else:  # DLRM init: U(-sqrt(1/n), sqrt(1/n))
                a = float(np.sqrt(1.0 / n))
                w = torch.empty((n, self.m), dtype=torch.float32, device=self.device).uniform_(-a, a, generator=g)
            if dtype == "f16":
                w = w.to(torch.float16)
            self.engine.load_table(k, w)
        self._plans = {}
        self._ids = list(range(len(self.ln_emb)))

    @classmethod
    def from_checkpoint(cls, path: str, device: int = 0):
        from .formats import load_dlrm_embedding_weights
        ws = load_dlrm_embedding_weights(path)
        return cls([w.shape[0] for w in ws], ws[0].shape[1], device=device, weights=ws)

    def apply_emb(self, lS_o, lS_i):
        """lS_o[k], lS_i[k]: offsets / indices of table k (torch CUDA tensors, int64 or int32, or
        numpy arrays -> host path).  Returns ly: list of [B_k, m] fp32, freshly allocated -- one fused
        launch on torch's current stream, no plan and no state kept (every batch brings new tensors)."""
        if len(lS_o) != len(self.ln_emb) or len(lS_i) != len(self.ln_emb):
            raise ValueError("need one (offsets, indices) pair per table")
        # validated and looked up in ONE engine call; nothing runs on bad input
        check = False if self.trusted_inputs else ("deferred" if self.deferred_check else True)
        if hasattr(lS_i, "dim") and hasattr(lS_o, "dim") and lS_i.dim() == 2 and lS_o.dim() == 2 and lS_i.is_cuda:
            # DLRM stacks fixed-size batches into [T, N] / [T, B] tensors: one [T, B, m] result, unbound
            return list(self.engine.lookup_stacked(self._ids, lS_i, lS_o, check=check).unbind(0))
        return self.engine.lookup_batched(self._ids, list(lS_i), list(lS_o), check=check)

    def validate(self, lS_o, lS_i) -> None:
        """emb_validate_inputs over one batch: IndexError on an index >= table rows or broken offsets, as
        nn.EmbeddingBag raises (the lookup kernels themselves are unchecked, like the reference's DPU program)."""
        stacked = hasattr(lS_i, "dim") and lS_i.dim() == 2
        bad = self.engine.validate(self._ids, list(lS_i.unbind(0)) if stacked else list(lS_i),
                                   list(lS_o.unbind(0)) if stacked else list(lS_o))
        if bad:
            raise IndexError(f"{bad} index / offset value(s) out of range (emb_validate_inputs)")

    def prepare(self, lS_o, lS_i):
        """For callers that reuse the SAME device tensors every step (static-shape serving, hipGraph
        capture): a prepared plan over these buffers.  plan.launch() enqueues the lookup (3-4 us of host
        time), plan.outputs are its fixed output tensors.  The caller keeps the inputs alive."""
        plan = self.engine.plan(self._ids, list(lS_i), list(lS_o))
        self._plans[id(plan)] = plan
        return plan

    forward = __call__ = apply_emb

    def close(self):
        for p in self._plans.values():
            p.destroy()
        self._plans.clear()
        self.engine.close()


class ShardedEmbeddingBagCollection:
    """The same `apply_emb(lS_o, lS_i) -> list of [B, m]` contract with the tables SHARDED over the ranks of one node
    (one process per GPU; every rank passes its own bags and gets its own pooled rows): `plan_shards` places each table --
    replicated, whole on an owner, or split by row range -- and every call is ONE library call (`emb_shard_lookup`).  The
    reference's lookup() likewise serves all its devices from one call (emb_host.h:258-270, 297, 312-321).

        comm = sharding.native_comm(engine, rank, world)   (RCCL; or peer=sharding.PeerGroup(...) for the peer-store exchange)
        ebc = ShardedEmbeddingBagCollection(ln_emb, m, rank, world, comm=comm)
        ly = ebc.apply_emb(lS_o, lS_i)                     # COLLECTIVE: every rank, every batch

    weights: per table [n, m] arrays (every rank passes the same; each keeps its shards), or None: W[k][r][c] from a hash
    of (k, r, c) in DLRM's U(-sqrt(1/n), sqrt(1/n)) range, so that ranks agree without exchanging anything."""

    def __init__(self, ln_emb, m_spa: int, rank: int, world: int, comm=None, peer=None, device: int = 0, weights=None, seed: int = 0,
                 replicate_bytes: int = 64 << 20, pooling: float = 1.0, trusted_inputs: bool = False, engine=None):
        import torch
        from . import sharding
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.ln_emb, self.m, self.rank, self.world = [int(n) for n in ln_emb], int(m_spa), int(rank), int(world)
        self.plan = sharding.plan_shards(self.ln_emb, self.m, 4, world, replicate_bytes=replicate_bytes, pooling=pooling)
        self.engine = engine if engine is not None else EmbeddingEngine(device=device, max_tables=len(self.plan.units) + 1)
        self._own_engine = engine is None
        self.sharded = sharding.ShardedEmbeddingBags(self.plan, self.engine, rank, comm, depth=0, check=not trusted_inputs, peer=peer)

        def rows_of(t, lo, hi):
            if weights is not None:
                return torch.as_tensor(np.asarray(weights[t][lo:hi]), dtype=torch.float32)
            a = float(np.float32(2.0 * np.sqrt(1.0 / self.ln_emb[t])))
            out = torch.empty((hi - lo, self.m), dtype=torch.float32, device=self.device)
            for r0 in range(lo, hi, 1 << 22):
                r1 = min(r0 + (1 << 22), hi)
                e = torch.arange(r0 * self.m, r1 * self.m, dtype=torch.int64, device=self.device)
                h = (e * 2654435761 + (t + 1) * 40503 + seed * 7919) % 2147483647
                out[r0 - lo:r1 - lo] = ((h.to(torch.float32) / 2147483647.0 - 0.5) * a).reshape(r1 - r0, self.m)
            return out

        self.sharded.load_tables(rows_of)

    def apply_emb(self, lS_o, lS_i):
        """lS_o[k], lS_i[k]: this rank's offsets / indices of table k (torch CUDA tensors, int64 or int32).  Returns this
        rank's ly: [B, m] fp32 per table.  With a PeerGroup the tensors of tables other ranks hold are copied into the
        group's arena first (peers gather from them in place)."""
        if len(lS_o) != len(self.ln_emb) or len(lS_i) != len(self.ln_emb):
            raise ValueError("need one (offsets, indices) pair per table")
        peer = self.sharded.peer
        if peer is not None:
            return self._apply_emb_peer(peer, list(lS_o), list(lS_i))
        return self.sharded.forward(list(lS_o), list(lS_i))

    def _apply_emb_peer(self, peer, lS_o, lS_i):
        """Peer stores: the owners gather this rank's indices IN PLACE and store pooled rows straight into its buffers, so all
        three must live in the group's ARENA -- a bump allocator that frees nothing before the group closes.  A training /
        serving loop therefore must not carve fresh blocks per call (round 4's harness did: the arena ran dry after a few
        hundred batches): three rotating slots per table, grown geometrically when a batch outgrows them, reused for ever.
        The rows a call returns stay valid for the next two calls."""
        t = self.torch
        T = len(self.ln_emb)
        if not hasattr(self, "_arena"):
            self._arena = {"slot": 0, "idx": [[None] * T for _ in range(3)], "off": [[None] * T for _ in range(3)], "out": [None] * 3}
        a = self._arena
        slot = a["slot"] = (a["slot"] + 1) % 3
        sh = self.sharded
        want = lS_i[0].dtype         # int64 (DLRM's) or int32: staged as they are -- the library takes both in place

        def staged(cache, k, x):
            x = sh._ids(x, want)
            n = int(x.numel())
            buf = cache[slot][k]
            if buf is None or buf.numel() < n or buf.dtype != want:
                buf = cache[slot][k] = peer.empty((max(2 * n, 64),), want)
            y = buf[:n]
            y.copy_(x)
            return y
        idx = [staged(a["idx"], k, lS_i[k]) for k in range(T)]
        off = [staged(a["off"], k, lS_o[k]) for k in range(T)]
        B = int(off[0].numel())
        one = a["out"][slot]
        if one is None or one.shape[1] < B:
            one = a["out"][slot] = peer.empty((T, max(2 * B, 16), self.m), t.float32)
        outs = [one[k][:B] for k in range(T)]
        return self.sharded.forward(off, idx, outs=outs)

    forward = __call__ = apply_emb

    def close(self):
        self.sharded.close()
        if self._own_engine:
            self.engine.close()


def random_batch(rng, ln_emb, batch: int, num_indices_per_lookup: int, fixed: bool, device=None):
    """DLRM's synthetic generator [EXT]: per sample and table, `num_indices_per_lookup` indices
    (fixed) or 1..num_indices_per_lookup (random), uniform over the table."""
    lS_o, lS_i = [], []
    for n in ln_emb:
        if fixed:
            lens = np.full(batch, num_indices_per_lookup, dtype=np.int64)
        else:
            lens = rng.integers(1, num_indices_per_lookup + 1, size=batch)
        off = np.zeros(batch, dtype=np.int64)
        off[1:] = np.cumsum(lens)[:-1]
        idx = rng.integers(0, n, size=int(lens.sum()), dtype=np.int64)
        lS_o.append(off)
        lS_i.append(idx)
    if device is not None:
        import torch
        lS_o = [torch.from_numpy(o).to(device) for o in lS_o]
        lS_i = [torch.from_numpy(i).to(device) for i in lS_i]
    return lS_o, lS_i


def main(argv=None):
    ap = argparse.ArgumentParser(description="embedding path of DLRM on the MI355X engine")
    ap.add_argument("--arch-sparse-feature-size", type=int, default=16)
    ap.add_argument("--arch-embedding-size", type=str, default="-".join(map(str, workloads.KAGGLE_ROWS)))
    ap.add_argument("--mini-batch-size", type=int, default=1)
    ap.add_argument("--num-indices-per-lookup", type=int, default=1)
    ap.add_argument("--num-indices-per-lookup-fixed", action="store_true")
    ap.add_argument("--num-batches", type=int, default=100)
    ap.add_argument("--inference-only", action="store_true")
    ap.add_argument("--data-set", type=str, default="random", choices=["random", "kaggle"])
    ap.add_argument("--data-generation", type=str, default="random", choices=["random", "dataset"])
    ap.add_argument("--processed-data-file", type=str, default="")
    ap.add_argument("--load-model", type=str, default="")
    ap.add_argument("--save-model", type=str, default="")
    ap.add_argument("--numpy-rand-seed", type=int, default=123)
    args, ignored = ap.parse_known_args(argv)
    if ignored:   # MLP / training / logging flags of the reference command lines (README.md:6,10,14)
        print("ignored (outside the embedding path):", " ".join(ignored))
    if not args.inference_only:
        print("note: only the inference embedding path exists here; running it (--inference-only implied)")
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(args.numpy_rand_seed)
    data = None
    if args.data_set == "kaggle" and args.processed_data_file:
        from .formats import CriteoKaggleNpz
        data = CriteoKaggleNpz(args.processed_data_file)
        ln_emb = data.table_rows
    else:
        ln_emb = [int(x) for x in args.arch_embedding_size.split("-")]
    if args.load_model:
        ebc = EmbeddingBagCollection.from_checkpoint(args.load_model)
    else:
        ebc = EmbeddingBagCollection(ln_emb, args.arch_sparse_feature_size)
    B = args.mini_batch_size
    batches = []
    for b in range(min(args.num_batches, 16)):
        if data is not None:
            o, i = data.batch((b * B) % max(data.n_samples - B, 1), B)
            batches.append(([torch.from_numpy(x).to(dev) for x in o], [torch.from_numpy(x).to(dev) for x in i]))
        else:
            batches.append(random_batch(rng, ebc.ln_emb, B, args.num_indices_per_lookup,
                                        args.num_indices_per_lookup_fixed, dev))
    for o, i in batches:          # checked once here (IndexError on a bad index), unchecked in the timed loops
        ebc.apply_emb(o, i)
    ebc.trusted_inputs = True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(args.num_batches):
        o, i = batches[b % len(batches)]
        ly = ebc.apply_emb(o, i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.num_batches
    n_bags = sum(int(x.shape[0]) for x in ly)
    print(f"apply_emb: {len(ebc.ln_emb)} tables, m={ebc.m}, batch {B}: {dt * 1e3:.4f} ms/batch, "
          f"{n_bags / dt:.3e} pooled lookups/s")
    plans = [ebc.prepare(o, i) for o, i in batches]        # same buffers every step: prepared plans
    for p in plans:
        p.launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(args.num_batches):
        plans[b % len(plans)].launch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.num_batches
    print(f"prepared plans (static buffers): {dt * 1e3:.4f} ms/batch, {n_bags / dt:.3e} pooled lookups/s")
    if args.save_model:      # the embedding part of a DLRM checkpoint: emb_l.<k>.weight, straight out of HBM
        from .formats import save_dlrm_embedding_weights
        save_dlrm_embedding_weights(args.save_model, [ebc.engine.table_tensor(k).float().cpu().numpy()
                                                      for k in range(len(ebc.ln_emb))])
        print("saved", args.save_model)
    ebc.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
