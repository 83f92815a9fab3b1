"""pim-embedding-lookup_amd -- MI355X-native replacement for the embedding-lookup hot path of
UBC-ECE-Sasha/PIM-Embedding-Lookup (host runtime upmem/include/emb_host.h + DPU kernel
upmem/src/dpu/emb_dpu_lookup.c), behind the C ABI in include/pimemb.h.

The directory name carries a hyphen (it is the reference's name); import it as
`pim_embedding_lookup_amd` (alias module at the repository root)."""
from . import lib, workloads  # noqa: F401
from .build import build, LIB_PATH  # noqa: F401
from .engine import DeviceBuffer, EmbeddingEngine, NativeExchange, Plan, RequestQueue  # noqa: F401
from .lib import (EMB_F16, EMB_F32, EMB_FIXED32, EMB_IDX_I64, EMB_IDX_U32, EMB_MEM_DEVICE,  # noqa: F401
                  EMB_MEM_HOST, PimembError)

__all__ = ["EmbeddingEngine", "DeviceBuffer", "Plan", "NativeExchange", "RequestQueue", "PimembError", "build", "lib", "workloads",
           "EMB_F32", "EMB_F16", "EMB_FIXED32", "EMB_IDX_U32", "EMB_IDX_I64", "EMB_MEM_HOST",
           "EMB_MEM_DEVICE", "LIB_PATH"]
