"""The gfx950 code object inside libpimemb.so, read without any tool: which device code a build ships, kernel by kernel.

bench.py ties a committed counter profile (profiles/traffic.json) to the DEVICE CODE it measured -- the sha256 of the
dominant kernel's machine code and kernel descriptor -- instead of to the bytes of the whole library or of host sources: a
host-only edit of the engine then leaves every single-GPU profile valid, a change of the kernel (or of a header it is
instantiated from) never does.  The same walk gives the register / spill / LDS figures of every kernel (the ISA facts DESIGN.md
quotes), straight from the code object's metadata note.

Layout walked here (all little endian):
  libpimemb.so (ELF64 x86-64)  ->  section .hip_fatbin  ->  clang offload bundle "__CLANG_OFFLOAD_BUNDLE__": u64 n, then n x
  {u64 offset, u64 size, u64 len, triple[len]}  ->  the entry whose triple ends in gfx950 = an ELF64 AMDGPU code object: symbols
  (STT_FUNC `name` = the kernel's code, STT_OBJECT `name.kd` = its 64-byte kernel descriptor), note NT_AMDGPU_METADATA (msgpack).
Reference counterpart: the DPU program is a separate binary the host loads by path (upmem/include/emb_host.h:155-160,
DPU_BINARY); here the device code travels inside the one library."""
from __future__ import annotations

import hashlib
import struct

BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


class _Elf:
    def __init__(self, blob: bytes):
        if blob[:4] != b"\x7fELF" or blob[4] != 2 or blob[5] != 1:
            raise ValueError("not a little-endian ELF64 image")
        self.b = blob
        (self.shoff,) = struct.unpack_from("<Q", blob, 0x28)
        self.shentsize, self.shnum, self.shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
        self.sections = []
        for i in range(self.shnum):
            name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", blob, self.shoff + i * self.shentsize)
            self.sections.append(dict(name_off=name, type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
        strtab = self.sections[self.shstrndx]
        for s in self.sections:
            s["name"] = self._cstr(strtab["off"] + s["name_off"])

    def _cstr(self, off: int) -> str:
        end = self.b.index(b"\0", off)
        return self.b[off:end].decode(errors="replace")

    def section(self, name: str):
        for s in self.sections:
            if s["name"] == name:
                return s
        return None

    def data(self, s) -> bytes:
        return self.b[s["off"]:s["off"] + s["size"]] if s["type"] != 8 else b""      # SHT_NOBITS

    def symbols(self):
        out = []
        for s in self.sections:
            if s["type"] not in (2, 11):          # SHT_SYMTAB, SHT_DYNSYM
                continue
            strs = self.sections[s["link"]]
            for i in range(s["size"] // 24):
                name, info, other, shndx, value, size = struct.unpack_from("<IBBHQQ", self.b, s["off"] + 24 * i)
                out.append(dict(name=self._cstr(strs["off"] + name), type=info & 0xF, shndx=shndx, value=value, size=size))
            if out:
                break                               # (.symtab and .dynsym list the kernels alike; one is enough)
        return out

    def bytes_at(self, shndx: int, value: int, size: int) -> bytes:
        s = self.sections[shndx]
        start = s["off"] + (value - s["addr"])
        return self.b[start:start + size]


def device_code_object(lib_path: str, arch: str = "gfx950") -> bytes:
    """The code object for `arch` out of the library's .hip_fatbin section."""
    with open(lib_path, "rb") as f:
        host = _Elf(f.read())
    fat = host.section(".hip_fatbin")
    if fat is None:
        raise ValueError(f"{lib_path}: no .hip_fatbin section (not a HIP library?)")
    blob = host.data(fat)
    at = blob.find(BUNDLE_MAGIC)
    if at < 0:
        raise ValueError(f"{lib_path}: .hip_fatbin holds no uncompressed clang offload bundle")
    blob = blob[at:]
    (n,) = struct.unpack_from("<Q", blob, len(BUNDLE_MAGIC))
    p = len(BUNDLE_MAGIC) + 8
    for _ in range(n):
        off, size, tlen = struct.unpack_from("<QQQ", blob, p)
        triple = blob[p + 24:p + 24 + tlen].decode(errors="replace")
        p += 24 + tlen
        if triple.startswith("hip") and triple.rstrip("-").endswith(arch) and size:
            return blob[off:off + size]
    raise ValueError(f"{lib_path}: no {arch} code object in the offload bundle")


def device_code_sha256(lib_path: str, arch: str = "gfx950") -> str:
    return hashlib.sha256(device_code_object(lib_path, arch)).hexdigest()


def kernel_hashes(lib_path: str, arch: str = "gfx950") -> dict[str, str]:
    """{mangled kernel name: sha256(machine code || 64-byte kernel descriptor)} for every kernel of the code object.  The
    descriptor's code-entry offset is masked: it moves whenever ANY kernel of the library grows, the kernel's code does not."""
    co = _Elf(device_code_object(lib_path, arch))
    syms = co.symbols()
    funcs = {s["name"]: s for s in syms if s["type"] == 2 and s["size"]}                 # STT_FUNC
    kds = {s["name"][:-3]: s for s in syms if s["type"] == 1 and s["name"].endswith(".kd")}   # STT_OBJECT
    out = {}
    for name, kd in kds.items():
        f = funcs.get(name)
        if f is None:
            continue
        h = hashlib.sha256()
        h.update(co.bytes_at(f["shndx"], f["value"], f["size"]))
        desc = bytearray(co.bytes_at(kd["shndx"], kd["value"], kd["size"] or 64))
        desc[16:24] = bytes(8)       # kernel_code_entry_byte_offset: WHERE the code sits relative to the descriptor -- layout, not code
        h.update(bytes(desc))
        out[name] = h.hexdigest()
    return out


def symbol_fragments(launch: dict) -> list[str]:
    """Fragments of the mangled name of the kernel instantiation one launch of a plan runs (Plan.describe() record): enough to
    pick it out of kernel_hashes().  The template heads are pimemb_kernels.hip's: bag_sum_<family>_kernel<IdxT, DT, LPR, Cfg
    [, RANGED]>, the any-dim kernels <IdxT, DT, CLAMP>."""
    idx = "j" if launch["itype"] == 0 else "l"
    kind, dt, lpr = launch["kind"], launch["dtype"], launch["lanes_per_row"]
    if kind == 3:
        return ["bag_sum_anydim_vec_kernelI%sLi%dE" % (idx, dt)] if launch.get("anydim_vec") else ["bag_sum_anydim_kernelI%sLi%dE" % (idx, dt)]
    if kind == 1:
        return ["bag_sum_group_kernelI%sLi%dELi%dENS_6BagCfgI" % (idx, dt, lpr)]
    if kind == 4:
        return ["bag_sum_hot_kernelI%sLi%dELi%dENS_6BagCfgI" % (idx, dt, lpr)]
    block = 128 if kind == 2 else 64          # the two-batch geometry runs 128-thread workgroups (pimemb_kernels.hip: Wave2Cfg)
    return ["bag_sum_wavebatch_kernelI%sLi%dELi%dENS_6BagCfgILi%dE" % (idx, dt, lpr, block), "EELb%dEEEvPK" % (1 if launch.get("ranged") else 0)]


def kernel_of_launch(lib_path: str, launch: dict, arch: str = "gfx950") -> tuple[str, str]:
    """(mangled symbol, sha256 of its code) of the kernel a plan's launch runs; raises when the library holds no such kernel or
    more than one (the fragments are then not an identification any more: fix symbol_fragments)."""
    hits = kernel_text_sha256(lib_path, symbol_fragments(launch), arch)
    if len(hits) != 1:
        raise LookupError("launch %s matches %d kernels of %s: %s" % (launch, len(hits), lib_path, sorted(hits)[:4]))
    return next(iter(hits.items()))


def kernel_text_sha256(lib_path: str, must_contain: list[str], arch: str = "gfx950") -> dict[str, str]:
    """The kernels whose mangled name contains every fragment of `must_contain` (e.g. ["bag_sum_wavebatch_kernel", "IjLi0ELi4E"])."""
    return {k: v for k, v in kernel_hashes(lib_path, arch).items() if all(m in k for m in must_contain)}


# ---- kernel metadata (NT_AMDGPU_METADATA: a msgpack map) -------------------------------------------------------------------
def _msgpack(b: bytes, p: int = 0):
    t = b[p]
    if t <= 0x7F:
        return t, p + 1
    if 0x80 <= t <= 0x8F:
        return _mp_map(b, p + 1, t & 0xF)
    if 0x90 <= t <= 0x9F:
        return _mp_arr(b, p + 1, t & 0xF)
    if 0xA0 <= t <= 0xBF:
        n = t & 0x1F
        return b[p + 1:p + 1 + n].decode(errors="replace"), p + 1 + n
    if t == 0xC0:
        return None, p + 1
    if t in (0xC2, 0xC3):
        return t == 0xC3, p + 1
    if t in (0xC4, 0xC5, 0xC6):
        w = {0xC4: 1, 0xC5: 2, 0xC6: 4}[t]
        n = int.from_bytes(b[p + 1:p + 1 + w], "big")
        return b[p + 1 + w:p + 1 + w + n], p + 1 + w + n
    if t in (0xCC, 0xCD, 0xCE, 0xCF):
        w = {0xCC: 1, 0xCD: 2, 0xCE: 4, 0xCF: 8}[t]
        return int.from_bytes(b[p + 1:p + 1 + w], "big"), p + 1 + w
    if t in (0xD0, 0xD1, 0xD2, 0xD3):
        w = {0xD0: 1, 0xD1: 2, 0xD2: 4, 0xD3: 8}[t]
        return int.from_bytes(b[p + 1:p + 1 + w], "big", signed=True), p + 1 + w
    if t in (0xD9, 0xDA, 0xDB):
        w = {0xD9: 1, 0xDA: 2, 0xDB: 4}[t]
        n = int.from_bytes(b[p + 1:p + 1 + w], "big")
        return b[p + 1 + w:p + 1 + w + n].decode(errors="replace"), p + 1 + w + n
    if t in (0xDC, 0xDD):
        w = 2 if t == 0xDC else 4
        return _mp_arr(b, p + 1 + w, int.from_bytes(b[p + 1:p + 1 + w], "big"))
    if t in (0xDE, 0xDF):
        w = 2 if t == 0xDE else 4
        return _mp_map(b, p + 1 + w, int.from_bytes(b[p + 1:p + 1 + w], "big"))
    if t == 0xCA:
        return struct.unpack(">f", b[p + 1:p + 5])[0], p + 5
    if t == 0xCB:
        return struct.unpack(">d", b[p + 1:p + 9])[0], p + 9
    if t >= 0xE0:
        return t - 256, p + 1
    raise ValueError(f"msgpack type 0x{t:02x} not handled")


def _mp_map(b, p, n):
    out = {}
    for _ in range(n):
        k, p = _msgpack(b, p)
        v, p = _msgpack(b, p)
        out[k] = v
    return out, p


def _mp_arr(b, p, n):
    out = []
    for _ in range(n):
        v, p = _msgpack(b, p)
        out.append(v)
    return out, p


def kernel_resources(lib_path: str, arch: str = "gfx950") -> dict[str, dict]:
    """{mangled kernel name: {vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, lds, scratch, max_wg}} from the metadata note."""
    co = _Elf(device_code_object(lib_path, arch))
    out = {}
    for s in co.sections:
        if s["type"] != 7:                         # SHT_NOTE
            continue
        d, p = co.data(s), 0
        while p + 12 <= len(d):
            namesz, descsz, typ = struct.unpack_from("<III", d, p)
            p += 12
            name = d[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = d[p:p + descsz]
            p += (descsz + 3) & ~3
            if name == b"AMDGPU" and typ == 32:    # NT_AMDGPU_METADATA
                meta, _ = _msgpack(desc)
                for k in meta.get("amdhsa.kernels", []):
                    out[k[".name"]] = dict(vgpr=k.get(".vgpr_count"), agpr=k.get(".agpr_count"), sgpr=k.get(".sgpr_count"),
                                           vgpr_spill=k.get(".vgpr_spill_count"), sgpr_spill=k.get(".sgpr_spill_count"),
                                           lds=k.get(".group_segment_fixed_size"), scratch=k.get(".private_segment_fixed_size"),
                                           max_wg=k.get(".max_flat_workgroup_size"))
    return out


if __name__ == "__main__":
    import sys
    from .build import LIB_PATH
    path = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else LIB_PATH
    res, hashes = kernel_resources(path), kernel_hashes(path)
    print("gfx950 code object sha256", device_code_sha256(path))
    for name in sorted(res):
        r = res[name]
        print(f"{r['vgpr']:4d} vgpr {r['vgpr_spill']:3d} spill {r['sgpr']:4d} sgpr {r['lds']:6d} lds {r['scratch']:5d} scratch  {hashes.get(name, '?')[:12]}  {name}")
