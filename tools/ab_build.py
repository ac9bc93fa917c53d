#!/usr/bin/env python
"""Builds compile-time variants of ONE source file of the library HERE (hipcc cross-compiles without a GPU), each linked with the
library's other objects into tools/ab_build/libab_<name>.so (git-ignored, but they travel to the GPU box with the snapshot), so
that the GPU box only has to RUN them (tools/ab_run.sh):

    python tools/ab_build.py icgn3d "base:-DOC_TAPS_PACKED=0" "pk:-DOC_TAPS_PACKED=1 -mllvm -disable-vector-combine"

Round 5: icgn2d.hip / icgn3d.hip exist twice in the library (separately rounded and fused arithmetic: <name>.o and <name>_fma.o,
opencorr_amd/csrc/oc_device.h).  `icgn3d` replaces the first, `icgn3d:fma` the second (the variant is compiled with -DOC_FMA=1 and
runs under oc_hip_set_tuning("arith_fma", 1)).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opencorr_amd import build as b  # noqa: E402

src, _, mode = sys.argv[1].partition(":")
fma = mode == "fma"
out = os.path.join(ROOT, "tools", "ab_build")
os.makedirs(out, exist_ok=True)
b.build(verbose=False)
replaced = os.path.join(b.LIBDIR, src + ("_fma.o" if fma else ".o"))
objs = [o for _, o, _ in b._units(b.SOURCES, b.LIBDIR, []) if o != replaced]
procs = []
for spec in sys.argv[2:]:
    name, flags = spec.split(":", 1)
    o = os.path.join(out, "%s_%s.o" % (src, name))
    cmd = [b.hipcc(), "--offload-arch=" + b.ARCH, "-c", os.path.join(b.CSRC, src + ".hip"), "-o", o] + b.FLAGS + (["-DOC_FMA=1"] if fma else []) + flags.split()
    procs.append((name, o, subprocess.Popen(cmd)))
for name, o, p in procs:
    if p.wait() != 0:
        raise SystemExit("hipcc failed for variant " + name)
    lib = os.path.join(out, "libab_%s_%s.so" % (src, name))
    subprocess.check_call([b.hipcc(), "--offload-arch=" + b.ARCH, "-shared", "-o", lib] + objs + [o, "-L/opt/rocm/lib", "-lrocfft", "-ldl", "-lpthread"])
    print(lib)
