"""Kernel resource table of libieagan_hip.so, read from the code-object metadata of the embedded gfx950 ELFs (no GPU needed):
name, VGPRs, SGPRs, scratch bytes per lane (private segment: spills / stack), LDS bytes.

    python tools/codeobj_notes.py            # kernels that use scratch
    python tools/codeobj_notes.py --all
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(so_path):
    """The gfx950 ELF images inside the .hip_fatbin section (clang offload bundles)."""
    tmp = tempfile.mkdtemp()
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path], check=True, capture_output=True)
    data = open(fat, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        base = m.start()
        n = struct.unpack_from("<Q", data, base + 24)[0]
        pos = base + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, pos)
            triple = data[pos + 24:pos + 24 + tlen].decode()
            pos += 24 + tlen
            if "gfx950" in triple and size > 0:
                out.append(data[base + off:base + off + size])
    return out


def kernels(so_path=None):
    so_path = so_path or os.path.join(ROOT, "iea-gan_amd", "libieagan_hip.so")
    rows = []
    for k, elf in enumerate(code_objects(so_path)):
        tmp = tempfile.NamedTemporaryFile(suffix=".co", delete=False)
        tmp.write(elf)
        tmp.close()
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", tmp.name], capture_output=True, text=True).stdout
        os.unlink(tmp.name)
        for blk in txt.split("- .agpr_count:")[1:]:
            def field(name, cast=int):
                m = re.search(r"\." + name + r":\s+(\S+)", blk)
                return cast(m.group(1)) if m else None
            name = field("name", str)
            rows.append(dict(name=name, vgpr=field("vgpr_count"), sgpr=field("sgpr_count"), scratch=field("private_segment_fixed_size"),
                             lds=field("group_segment_fixed_size"), spill_vgpr=field("vgpr_spill_count")))
    return rows


def demangle(names):
    import shutil
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if tool is None or not names:
        return list(names)
    r = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.strip().split("\n")


if __name__ == "__main__":
    rows = kernels()
    show = rows if "--all" in sys.argv else [r for r in rows if r["scratch"]]
    for r, d in zip(show, demangle([r["name"] for r in show])):
        print(f"{r['vgpr']:4d} vgpr {r['sgpr']:4d} sgpr {r['scratch']:5d} B scratch ({r['spill_vgpr']} spilled) {r['lds']:6d} B lds  {d[:150]}")
    print(f"{len(rows)} kernels, {sum(1 for r in rows if r['scratch'])} with scratch")
