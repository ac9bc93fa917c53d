#!/usr/bin/env python3
"""Register / scratch / LDS use and an instruction census of the gfx950 kernels inside an object file.

    python tools/kernel_resources.py opencorr_amd/lib/icgn2d.o [name-substring ...]

Unbundles the device code object (clang-offload-bundler), reads the kernel descriptors' metadata (llvm-readelf --notes:
.vgpr_count is the ALLOCATED count per lane -- the number rocprofv3's kernel trace prints is this value in units of two
registers on wave64, see tools/rocpd_summary.py), and counts instructions per kernel from the disassembly.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def unbundle(obj, out):
    fat = out + ".fatbin"   # the host object carries the device bundle in its .hip_fatbin section
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + fat, "--output=" + out, "--unbundle"])


def kernels_meta(co):
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    out = []
    cur = {}
    for line in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("name"):
            out.append(cur)
            cur = {}
        if k in ("agpr_count", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "name",
                 "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size"):
            cur[k] = v
        if k == "wavefront_size" and cur.get("name"):
            out.append(cur)
            cur = {}
    if cur.get("name"):
        out.append(cur)
    return out


def census(co):
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True)
    per = {}
    name = None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            name = m.group(1)
            per[name] = collections.Counter()
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s", line)
        if m and name:
            per[name][m.group(1)] += 1
    return per


def demangle(n):
    try:
        return subprocess.check_output(["c++filt", n], text=True).strip()
    except Exception:
        return n


def main():
    obj = sys.argv[1]
    filt = sys.argv[2:]
    with tempfile.TemporaryDirectory() as d:
        co = os.path.join(d, "dev.co")
        unbundle(obj, co)
        meta = kernels_meta(co)
        cen = census(co)
    for m in meta:
        name = m["name"]
        dn = demangle(name)
        if filt and not any(f in dn for f in filt):
            continue
        c = cen.get(name, {})
        total = sum(c.values())
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        fma = sum(v for k, v in c.items() if k.startswith("v_fma") or k.startswith("v_pk_fma") or k.startswith("v_fmac"))
        pk = sum(v for k, v in c.items() if k.startswith("v_pk_"))
        print("%s\n   vgpr %s agpr %s sgpr %s scratch %s B lds(static) %s B | insts %d valu %d (fma-class %d, packed %d) s_barrier %d buffer_load %d ds %d"
              % (dn, m.get("vgpr_count"), m.get("agpr_count"), m.get("sgpr_count"), m.get("private_segment_fixed_size"),
                 m.get("group_segment_fixed_size"), total, valu, fma, pk, c.get("s_barrier", 0),
                 sum(v for k, v in c.items() if k.startswith("buffer_load")), sum(v for k, v in c.items() if k.startswith("ds_"))))


if __name__ == "__main__":
    main()
