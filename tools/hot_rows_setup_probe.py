"""What emb_set_hot_rows costs per call (one table, dim 128, 400 candidate rows): the accepted rows are copied next to their hash by ONE
gather kernel (round 6; a device-to-device copy per row before).  usage: python tools/hot_rows_setup_probe.py"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pim_embedding_lookup_amd as pel
dev = torch.device("cuda", 0)
eng = pel.EmbeddingEngine(device=0, max_tables=8)
eng.load_table(0, torch.rand((1_000_000, 128), device=dev))
ids = np.arange(0, 100_000, 97, dtype=np.uint64)[:400]
for _ in range(3): eng.set_hot_rows(0, ids)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): eng.set_hot_rows(0, ids)
print("emb_set_hot_rows, dim 128, %d candidate rows: %.3f ms per call" % (len(ids), (time.perf_counter() - t0) / 20 * 1e3))
eng.close()
