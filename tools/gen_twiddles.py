#!/usr/bin/env python
"""Prints the Twiddle<N> specialisations of opencorr_amd/csrc/fft_device.h (cos / sin of 2*pi*k/N, k < N) for the
window sides the fused FFTCC2D kernel is instantiated for:   python tools/gen_twiddles.py 24 30 40 48"""
import math
import sys

for n in (int(a) for a in sys.argv[1:]):
    c = ", ".join("%.10ef" % math.cos(2 * math.pi * k / n) for k in range(n))
    s = ", ".join("%.10ef" % math.sin(2 * math.pi * k / n) for k in range(n))
    print("template <>\nstruct Twiddle<%d> {\n    static constexpr float c[%d] = {%s};\n    static constexpr float s[%d] = {%s};\n};\n"
          % (n, n, c, n, s))
