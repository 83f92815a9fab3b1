"""CPU: the C-ABI library loads and exports every symbol include/pimemb.h declares; the ctypes
binding covers them all; struct layouts match the header.  No compute is called (no GPU here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pimemb.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)       # strip comments
    text = re.sub(r"typedef\s+(struct|enum)\s+\w+\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)
    names = re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{]*\)\s*;", text)
    return sorted(set(names))


def test_header_declares_the_reference_signatures():
    text = open(HEADER).read()
    # emb_host.h:136 and :234, token for token
    assert re.search(r"struct dpu_set_t \*populate_mram\(uint32_t table_id, uint64_t nr_rows, uint32_t col,\s*"
                     r"int32_t \*table_data, dpu_runtime_totals \*runtime\);", text)
    assert re.search(r"int32_t \*lookup\(uint32_t \*\*indices, uint32_t \*\*offsets, float \*\*final_results,\s*"
                     r"void \*dpu_set_ptr_untyped, int64_t latency_print\);", text)


def test_library_exports_every_declared_symbol(pel):
    fns = declared_functions()
    assert {"populate_mram", "lookup", "emb_create", "emb_lookup", "emb_lookup_batched",
            "emb_load_table", "emb_get_stats", "emb_destroy", "emb_last_error",
            "emb_configure"} <= set(fns)
    assert os.path.exists(pel.LIB_PATH), "libpimemb.so must be built in-tree (__graft_entry__.build())"
    out = subprocess.check_output(["nm", "-D", "--defined-only", pel.LIB_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [f for f in fns if f not in exported]
    assert not missing, f"declared in pimemb.h but not exported: {missing}"


def test_ctypes_binding_covers_the_header(pel):
    assert sorted(pel.lib.SIGNATURES) == declared_functions()
    L = pel.lib.load()
    for name in pel.lib.SIGNATURES:
        assert getattr(L, name) is not None
    assert L.emb_version().startswith(b"pimemb")


def test_struct_layouts(pel):
    assert C.sizeof(pel.lib.EmbLookupDesc) == 48
    assert C.sizeof(pel.lib.EmbConfig) == 12
    assert C.sizeof(pel.lib.EmbStats) == 5 * 8 + 6 * 8 + 5 * 8
    assert C.sizeof(pel.lib.DpuRuntimeTotals) == 48      # six doubles, emb_host.h:41-48
    assert C.sizeof(pel.lib.EmbRouteTable) == 32 and pel.lib.EmbRouteTable.rows_per_shard.offset == 28
    assert pel.lib.EmbLookupDesc.indices.offset == 8 and pel.lib.EmbLookupDesc.pooled.offset == 40


def test_header_compiles_as_c_and_cxx(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "pimemb.h"\nint main(void){emb_lookup_desc d; (void)d; '
                   'return (sizeof(emb_stats)==128 && sizeof(emb_route_table)==32)?0:1;}\n')
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        exe = tmp_path / ("t_" + cc)
        subprocess.check_call([cc, std, "-Wall", "-Werror", "-x", "c" if cc == "gcc" else "c++",
                               "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
        assert subprocess.call([str(exe)]) == 0


def test_links_one_hip_runtime_by_unversioned_name(pel):
    """DT_NEEDED must be the unversioned libamdhip64.so so a torch process shares ONE HIP runtime
    with the engine (csrc/Makefile note)."""
    out = subprocess.check_output(["objdump", "-p", pel.LIB_PATH], text=True)
    needed = re.findall(r"NEEDED\s+(\S+)", out)
    assert "libamdhip64.so" in needed
    assert not any(n.startswith("libtorch") or n.startswith("libc10") for n in needed)
    maps = {line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line}
    assert len(maps) <= 1, maps


def test_no_gpu_fails_loudly_not_silently(pel):
    """Without a GPU the product must raise; with one (GPU box) creation simply works."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pel.PimembError) as ei:
        pel.EmbeddingEngine()
    assert ei.value.code == pel.lib.EMB_ERR_DEVICE
    assert pel.lib.load().emb_configure(0, 1, 1, 1) == pel.lib.EMB_ERR_INVALID
    assert b"non-zero" in pel.lib.load().emb_last_error()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import, include, link or
    dlopen it (comments may mention it)."""
    pat = re.compile(r"(^\s*(import|from)\s+\S*oracle|#\s*include\s+\S*oracle|libemb_oracle|oracle/|emb_oracle)",
                     re.M)
    pkg = os.path.join(ROOT, "pim-embedding-lookup_amd")
    seen = 0
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                seen += 1
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not pat.search(text), (dirpath, f)
    assert seen >= 8


def test_marshal_helper_is_built_and_refuses_what_it_does_not_handle():
    """_pimemb_marshal (torch tensor lists -> emb_lookup_desc records) is built next to libpimemb.so and loads; CPU
    tensors, mixed dtypes and unknown tables are not its business: None (the Python path reports the error) / KeyError."""
    import ctypes
    import torch
    from importlib import import_module
    eng = import_module("pim-embedding-lookup_amd.engine")
    m = eng._marshal()
    assert m is not None, "pim-embedding-lookup_amd/lib/_pimemb_marshal.so missing: run __graft_entry__.build()"
    buf = ctypes.create_string_buffer(48 * 2)
    tables = {0: (5, 16, 0), 1: (7, 16, 0)}
    i, o = torch.zeros(3, dtype=torch.int64), torch.zeros(1, dtype=torch.int64)
    assert m.pack(ctypes.addressof(buf), [0, 1], [i, i], [o, o], None, tables) is None          # not CUDA tensors
    assert m.pack(ctypes.addressof(buf), [0, 1], [i, 5], [o, o], None, tables) is None          # not a tensor
    assert m.pack(ctypes.addressof(buf), [0], [i, i], [o, o], None, tables) is None             # lengths differ
    key = m.fill_pooled(ctypes.addressof(buf), 2, 4096, 16)
    assert isinstance(key, bytes) and len(key) == 96


def test_round4_struct_layouts_match_the_header(pel, tmp_path):
    """The structs of the sharded call, the transfer list and their stats as a C compiler lays them out (a small program
    over include/pimemb.h prints sizeof / offsetof) against the ctypes mirrors in lib.py -- field for field."""
    L = pel.lib
    specs = {"emb_comm_op": L.EmbCommOp, "emb_shard_table": L.EmbShardTable, "emb_shard_input": L.EmbShardInput,
             "emb_shard_config": L.EmbShardConfig, "emb_shard_stats": L.EmbShardStats, "emb_lookup_desc": L.EmbLookupDesc,
             "emb_route_table": L.EmbRouteTable}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "pimemb.h"', 'int main(void){']
    for cname, ct in specs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in ct._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, ct in specs.items():
        assert int(got[cname]) == C.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(ct, fname).offset, f"{cname}.{fname}"
    # placement / flag constants
    text = open(HEADER).read()
    for name, val in (("EMB_PLACE_REPLICATED", L.EMB_PLACE_REPLICATED), ("EMB_PLACE_WHOLE", L.EMB_PLACE_WHOLE), ("EMB_PLACE_ROWS", L.EMB_PLACE_ROWS),
                      ("EMB_SHARD_SELF_VIA_COMM", L.EMB_SHARD_SELF_VIA_COMM), ("EMB_SHARD_CHECK_SERVED", L.EMB_SHARD_CHECK_SERVED),
                      ("EMB_SHARD_PEER_STORES", L.EMB_SHARD_PEER_STORES), ("EMB_SHARD_NO_DIRECT", L.EMB_SHARD_NO_DIRECT)):
        assert re.search(r"#define %s %du\b" % (name, val), text), name


def test_boundary_documents_do_not_name_a_retired_transport():
    """The ABI header is the boundary document: the RCCL-mode sharded step has ONE transport --
    emb_comm_exchange issued from C -- and bench.py has no torch.distributed data path any more.  The retired wording
    ("--collective torch", "uses torch.distributed unless") must not come back in the header, the RCCL binding or the
    integration notes."""
    import re
    retired = re.compile(r"--collective\s+torch|torch\.distributed\s+unless|keeps\s+torch\.distributed\s+as\s+the\s+default|all_to_all_rounds", re.S)
    for rel in ("include/pimemb.h", "pim-embedding-lookup_amd/csrc/pimemb_comm.cpp", "pim-embedding-lookup_amd/csrc/pimemb_shard.cpp",
                "INTEGRATION.md", "README.md"):
        text = " ".join(open(os.path.join(ROOT, rel)).read().split())
        m = retired.search(text)
        assert m is None, f"{rel}: retired wording {m.group(0)!r}"
    # ... and what bench.py ships agrees: --collective has the one choice
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert 'choices=["native"]' in bench and '"--collective"' in bench


def test_last_words_survive_a_fatal_signal(tmp_path):
    """emb_peer_last_words (what bench.py's peer-store leg leaves behind before it runs code no link has seen): a process that
    abort()s -- the runtime's answer to a GPU memory fault -- or segfaults while a line is registered writes the line to the
    given descriptor and ends with the given status; once the handlers are taken out again a fatal signal is fatal again."""
    import subprocess
    import sys
    import textwrap
    script = textwrap.dedent("""
        import ctypes, os, sys
        sys.path.insert(0, %r)
        import pim_embedding_lookup_amd as pel
        L = pel.lib.load()
        how = sys.argv[1]
        fd = os.dup(1)
        os.dup2(2, 1)                        # (as dist_bench does: descriptor 1 belongs to the libraries)
        assert L.emb_peer_last_words(b'{"metric": "x", "value": 1}', fd, 0) == 0
        assert L.emb_peer_last_words(b'{"metric": "kept", "value": 2}', fd, 5 if how == "status" else 0) == 0      # replaced, not appended
        if how == "cleared":
            assert L.emb_peer_last_words(None, 1, 0) == 0
        if how == "segv":
            ctypes.string_at(8)
        if how == "term":
            import signal
            os.kill(os.getpid(), signal.SIGTERM)
            signal.pause()
        os.abort()
    """ % ROOT)
    for how, rc_ok in (("abort", True), ("segv", True), ("term", True), ("status", True), ("cleared", False)):
        res = subprocess.run([sys.executable, "-c", script, how], capture_output=True, text=True, timeout=120)
        if rc_ok:
            assert res.returncode == (5 if how == "status" else 0) and res.stdout == '{"metric": "kept", "value": 2}\n', (how, res.returncode, res.stdout, res.stderr[-500:])
        else:
            assert res.returncode != 0 and res.stdout == "", (how, res.returncode, res.stdout)
