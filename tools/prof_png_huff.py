#!/usr/bin/env python3
"""Cycle breakdown of png_huff_kernel, thread 0 of every workgroup (build with UCFP_HIPCC_EXTRA="-DPNG_PROF -DPNG_PROF_HUFF"):
python tools/prof_png_huff.py [n] [level]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
from bench_png import make  # noqa: E402
from ucfp_amd import _lib, image  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pngs, imgs = make(n, level)
lib = _lib.load()
f = lib.ucfp_debug_png_prof
f.argtypes = [C.c_void_p, C.c_int]
image.decode_pngs(pngs[:8], 256, 256, image.PIX_RGB8)
f(None, 1)
fr, st = image.decode_pngs(pngs, 256, 256, image.PIX_RGB8)
acc = (C.c_ulonglong * 16)()
f(acc, 0)
a = np.array(list(acc), dtype=np.float64) / n
names = ["header+tables", "stage", "warm-up", "chain", "scan", "emit", "round end"]
tot = a[:7].sum()
print(f"ok={not st.any()} per image: total {tot/1e6:.2f} Mcycles; rounds {a[9]:.1f} chain iterations/round {a[10]/max(a[9],1):.2f} "
      f"subsequences confirmed/round {a[11]/max(a[9],1):.1f}")
for i, nm in enumerate(names):
    print(f"  {nm:14s} {a[i]/1e3:9.1f} kcycles  {100*a[i]/tot:5.1f}%   per round {a[i]/max(a[9],1):8.0f}")
