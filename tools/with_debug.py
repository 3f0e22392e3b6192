#!/usr/bin/env python
"""python tools/with_debug.py <switch>[,<switch>...] <script.py> [args...]: run a script (bench.py, a tool) with verification builds of the
blend kernels switched on (the switches are exported by libgmesh_hip.so but are not part of the public header; the product never sets them):
  fwd_exact   gm_debug_forward_exact_exponent(1):  forward blend with the per-pixel exponent
  none        nothing (same command line shape for A/B loops)"""
import ctypes
import os
import runpy
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from gaussianmesh_amd import _lib

lib = ctypes.CDLL(_lib.lib()._name)
for sw in sys.argv[1].split(","):
    if sw == "fwd_exact":
        lib.gm_debug_forward_exact_exponent(1)
    elif sw != "none":
        raise SystemExit("unknown switch %r" % sw)
script = sys.argv[2]
sys.argv = sys.argv[2:]
runpy.run_path(script, run_name="__main__")
