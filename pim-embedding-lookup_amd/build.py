"""Build recipe for libpimemb.so (hipcc, gfx950 only).  Used by __graft_entry__.build()."""
from __future__ import annotations

import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "lib", "libpimemb.so")
MARSHAL_PATH = os.path.join(PKG_DIR, "lib", "_pimemb_marshal.so")
SOURCES = ["pimemb_kernels.hip", "pimemb_engine.cpp", "pimemb_compat.cpp", "pimemb_comm.cpp", "pimemb_shard.cpp", "pimemb_peer.cpp", "pimemb_peer.h", "pimemb_internal.h", "pimemb_torch_marshal.cpp",
           "pimemb_bag_kernels.h", "pimemb_xcd_map.h", "pimemb_hot_rows.h", "pimemb_hostcopy.h", "Makefile",
           os.path.join("..", "..", "include", "pimemb.h")]


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC_DIR, s)) > built for s in SOURCES if s != "pimemb_torch_marshal.cpp")


def build_marshal(verbose: bool = False) -> str | None:
    """The OPTIONAL CPython helper that unpacks lists of torch tensors (engine.py's fast path): built with the running
    interpreter's headers and the C++ ABI torch itself was built with.  A box without torch headers, or a compiler that
    cannot build it, gets a warning and the Python unpacking path -- libpimemb.so does not depend on it."""
    src = os.path.join(CSRC_DIR, "pimemb_torch_marshal.cpp")
    if os.path.exists(MARSHAL_PATH) and os.path.getmtime(MARSHAL_PATH) >= os.path.getmtime(src):
        return MARSHAL_PATH
    import sys
    import warnings
    try:
        import torch
        abi = int(bool(torch._C._GLIBCXX_USE_CXX11_ABI))
    except Exception as ex:  # noqa: BLE001 -- no torch: nothing to marshal
        warnings.warn(f"_pimemb_marshal.so not built (torch is not importable: {ex})")
        return None
    res = subprocess.run(["make", "-C", CSRC_DIR, "marshal", f"PYTHON={sys.executable}", f"CXX11_ABI={abi}"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose:
        print(res.stdout)
    if res.returncode != 0 or not os.path.exists(MARSHAL_PATH):
        warnings.warn("_pimemb_marshal.so (optional torch tensor-list helper) could not be built; tensor lists are unpacked "
                      "in Python:\n" + res.stdout[-2000:])
        return None
    return MARSHAL_PATH


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP kernels + C-ABI into pim-embedding-lookup_amd/lib/libpimemb.so (in-tree, so
    the built library travels to the GPU box with the repo snapshot)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC_DIR, "clean"], stdout=subprocess.DEVNULL)
    if force or is_stale():
        cmd = ["make", "-C", CSRC_DIR, "-j4"]
        if verbose:
            subprocess.check_call(cmd)
        else:
            res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if res.returncode != 0:
                raise RuntimeError("building libpimemb.so failed:\n" + res.stdout)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libpimemb.so missing after build")
    build_marshal(verbose)          # optional: a failure is a warning, never an error
    return LIB_PATH


def build_clamped(out_dir: str) -> str:
    """Build the input-clamping flavour (-DPIMEMB_CLAMP_INPUTS=1: malformed indices / offsets give garbage
    rows instead of out-of-bounds accesses, ~2.5 % slower on the headline shape) into out_dir."""
    out = os.path.join(out_dir, "libpimemb_clamp.so")
    flags = "-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -I../../include -I. -DPIMEMB_CLAMP_INPUTS=1"
    res = subprocess.run(["make", "-C", CSRC_DIR, "-j8", f"CXXFLAGS={flags}", f"OUT={out}",
                          f"OBJDIR={os.path.join(out_dir, 'obj_clamp')}"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0 or not os.path.exists(out):
        raise RuntimeError("building the clamped flavour failed:\n" + res.stdout)
    return out


if __name__ == "__main__":
    print(build(verbose=True))
