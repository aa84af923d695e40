#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in libwavtok_hip.so, read from the code objects' metadata notes
(no GPU needed): python3 tools/kernel_resources.py [--spills-only] [lib.so].  tests/test_host_logic.py uses kernel_table()
to keep spilling instantiations out of the default plans."""
import os, re, struct, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
KEYS = ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
        "group_segment_fixed_size", "max_flat_workgroup_size")


def _code_objects(lib, td):
    """The gfx950 ELF images inside the library's .hip_fatbin section (one clang offload bundle per translation unit)."""
    fat = os.path.join(td, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(td, "scratch.so")],
                   check=True)
    d = open(fat, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), d):
        base = m.start()
        n, = struct.unpack_from("<Q", d, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", d, pos)
            triple = d[pos + 24: pos + 24 + tl].decode()
            pos += 24 + tl
            if "gfx950" in triple and size:
                p = os.path.join(td, f"co{len(out)}.elf")
                open(p, "wb").write(d[base + off: base + off + size])
                out.append(p)
    return out


def kernel_table(lib=None):
    """{demangled kernel name: {vgpr_count, agpr_count, sgpr_count, vgpr_spill_count, private_segment_fixed_size, ...}}"""
    lib = lib or os.path.join(ROOT, "wavtokenizer_amd", "libwavtok_hip.so")
    table = {}
    with tempfile.TemporaryDirectory() as td:
        for co in _code_objects(lib, td):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                                   text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s+-?\s*\.(\w+):\s+(\S+)", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2)
                if k in KEYS:
                    cur[k] = int(v)
                elif k == "symbol":              # '<mangled>.kd'
                    cur["mangled"] = v[:-3] if v.endswith(".kd") else v
                    cur["mangled"] = cur["mangled"].strip("'\"")
                elif k == "wavefront_size":      # last key of a kernel's block in the emitted order
                    pass
                if "mangled" in cur and all(x in cur for x in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size")):
                    table[cur["mangled"]] = cur
                    cur = {}
    names = list(table.keys())
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return {d: table[n] for d, n in zip(dem, names)}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    spills_only = "--spills-only" in sys.argv
    tab = kernel_table(args[0] if args else None)
    for name, t in sorted(tab.items()):
        sp, sc = t.get("vgpr_spill_count", 0), t.get("private_segment_fixed_size", 0)
        if spills_only and not sp and not sc:
            continue
        print(f"vgpr {t.get('vgpr_count', 0):3d} (agpr {t.get('agpr_count', 0):3d}) sgpr {t.get('sgpr_count', 0):3d}  spill {sp:3d}  scratch {sc:4d} B  "
              f"lds {t.get('group_segment_fixed_size', 0):6d}  {name[:170]}")
    print(f"{len(tab)} kernels", file=sys.stderr)
