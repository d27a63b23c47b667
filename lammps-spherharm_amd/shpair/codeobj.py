"""Identity of the gfx950 kernels inside a built libshpair.so, without a GPU and without LLVM tools.

  kernel_hashes(lib) -> {mangled kernel symbol: first 16 hex digits of the SHA-256 of the kernel's machine code}

`profiles/pmc_traffic.json` stores, with every PMC measurement, the hash of the contact kernel it was taken on
(tools/pmc_table.py); bench.py recomputes the hash of the kernel it actually launches and marks the static
`utilisation` / `roofline.traffic` figures `"stale": true` when the two differ.

Layout read here: the shared library's `.hip_fatbin` section is a sequence of clang offload bundles (one per
translation unit; uncompressed: magic, entry count, then {offset, size, triple} per entry); the gfx950 entry of each
bundle is an ELF64 code object whose `.symtab` lists every kernel as an STT_FUNC with its size.
"""
import hashlib
import struct

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _sections(elf):
    """{name: (offset, size, addr)} of an ELF64 little-endian image, and the list of raw section headers."""
    if elf[:4] != b"\x7fELF" or elf[4] != 2 or elf[5] != 1:
        raise ValueError("not an ELF64 little-endian image")
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    hdr = []
    for k in range(shnum):
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + k * shentsize)
        hdr.append(dict(name=name, type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
    stab = hdr[shstrndx]
    names = elf[stab["off"]:stab["off"] + stab["size"]]
    out = {}
    for h in hdr:
        end = names.index(b"\0", h["name"])
        h["sname"] = names[h["name"]:end].decode()
        out[h["sname"]] = h
    return out, hdr


def code_objects(lib_path, target="gfx950"):
    """The device ELF images bundled in `lib_path` for `target` (one per translation unit that has kernels)."""
    blob = open(lib_path, "rb").read()
    secs, _ = _sections(blob)
    if ".hip_fatbin" not in secs:
        raise ValueError(f"{lib_path}: no .hip_fatbin section")
    fb = secs[".hip_fatbin"]
    fat = blob[fb["off"]:fb["off"] + fb["size"]]
    out = []
    pos = fat.find(_MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", fat, pos + len(_MAGIC))
        p = pos + len(_MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", fat, p)
            triple = fat[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if target in triple and size:
                out.append(fat[pos + off:pos + off + size])
        pos = fat.find(_MAGIC, pos + 1)
    return out


def kernel_hashes(lib_path, target="gfx950"):
    """{symbol: sha256[:16] of the function's bytes} for every function symbol of every code object in the library."""
    res = {}
    for co in code_objects(lib_path, target):
        secs, hdr = _sections(co)
        if ".symtab" not in secs:
            continue
        st = secs[".symtab"]
        strtab = hdr[st["link"]]
        strs = co[strtab["off"]:strtab["off"] + strtab["size"]]
        for k in range(st["size"] // 24):
            name, info, other, shndx, value, size = struct.unpack_from("<IBBHQQ", co, st["off"] + 24 * k)
            if (info & 0xF) != 2 or size == 0 or shndx == 0 or shndx >= len(hdr):   # STT_FUNC with a body
                continue
            sec = hdr[shndx]
            a = sec["off"] + (value - sec["addr"])
            sym = strs[name:strs.index(b"\0", name)].decode()
            res[sym] = hashlib.sha256(co[a:a + size]).hexdigest()[:16]
    return res


def contact_kernel_symbol_fragment(order, needv, weighted, family, waves_per_pair, specialised=0):
    """Itanium-mangled template-argument list of shp::pair_contact_kernel<L, NEEDV, WEIGHTED, JPT, WPP, SPEC>."""
    lit = "n{}".format(-int(order)) if int(order) < 0 else str(int(order))   # the run-time-order kernel is instantiated with L = -1
    return "pair_contact_kernelILi{}ELb{}ELb{}ELb{}ELi{}ELb{}EE".format(lit, int(bool(needv)), int(bool(weighted)), int(bool(family)),
                                                                        int(waves_per_pair), int(bool(specialised)))


def contact_kernel_hash(lib_path, order, needv, weighted, family, waves_per_pair, specialised=0):
    """(symbol, hash) of the contact-kernel instance a workload launches, or (None, None) if the library has none."""
    frag = contact_kernel_symbol_fragment(order, needv, weighted, family, waves_per_pair, specialised)
    for sym, h in kernel_hashes(lib_path).items():
        if frag in sym:
            return sym, h
    return None, None
