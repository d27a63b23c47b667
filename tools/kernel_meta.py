"""Code-object metadata of every kernel in libshpair.so (no GPU needed): registers, spills, scratch, LDS.
  python tools/kernel_meta.py [lib.so] [--json]
Reads the AMDGPU metadata note of the embedded gfx950 code object with llvm-readelf."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object_notes(lib):
    """Text of `llvm-readelf --notes` of every gfx950 code object bundled in `lib` (one per translation unit)."""
    out = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        # the fat binaries sit in section .hip_fatbin as clang offload bundles, one after the other
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for k, a in enumerate(starts):
            b = starts[k + 1] if k + 1 < len(starts) else len(blob)
            one, co = os.path.join(td, f"b{k}.bin"), os.path.join(td, f"co{k}.hsaco")
            open(one, "wb").write(blob[a:b])
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={one}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"],
                                  stderr=subprocess.DEVNULL)
            if os.path.getsize(co):
                out.append(subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True))
    return "\n".join(out)


def scratch_instruction_counts(lib):
    """{mangled kernel name: number of scratch_* / private buffer instructions in its disassembly}."""
    counts = {}
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for k, a in enumerate(starts):
            b = starts[k + 1] if k + 1 < len(starts) else len(blob)
            one, co = os.path.join(td, f"b{k}.bin"), os.path.join(td, f"co{k}.hsaco")
            open(one, "wb").write(blob[a:b])
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={one}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"],
                                  stderr=subprocess.DEVNULL)
            if not os.path.getsize(co):
                continue
            dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True)
            cur = None
            for ln in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
                if m:
                    cur = m.group(1)
                    counts.setdefault(cur, 0)
                elif cur and re.search(r"\bscratch_(load|store)|buffer_(load|store)_\w+ .*\boffen\b.*\bs\[0:3\]", ln):
                    counts[cur] += 1
    return counts


def kernels(lib):
    txt = code_object_notes(lib)
    out = []
    for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
        def g(key, cast=int):
            m = re.search(rf"\.{key}:\s*(\S+)", blk)
            return cast(m.group(1)) if m else None
        name = g("name", str)
        try:
            dem = subprocess.check_output([os.path.join(LLVM, "llvm-cxxfilt"), name], text=True).strip()
        except (OSError, subprocess.CalledProcessError):
            dem = name
        out.append(dict(name=dem, symbol=name, vgprs=g("vgpr_count"), sgprs=g("sgpr_count"), vgpr_spills=g("vgpr_spill_count"),
                        sgpr_spills=g("sgpr_spill_count"), scratch_bytes=g("private_segment_fixed_size"),
                        static_lds=g("group_segment_fixed_size")))
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(ROOT, "lammps-spherharm_amd", "shpair", "libshpair.so")
    ks = kernels(lib)
    if "--json" in sys.argv:
        print(json.dumps(ks, indent=1))
    else:
        for k in sorted(ks, key=lambda k: k["name"]):
            short = re.sub(r"shp::|\(shp::PairParams\)|void ", "", k["name"])[:70]
            print(f"{short:70s} vgpr {k['vgprs']:3d} sgpr {k['sgprs']:3d} spill v{k['vgpr_spills']} s{k['sgpr_spills']:3d} "
                  f"scratch {k['scratch_bytes']} lds {k['static_lds']}")
