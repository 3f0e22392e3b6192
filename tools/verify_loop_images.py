#!/usr/bin/env python
"""The bench's pipelined frame loop at ITS OWN size (1 M Gaussians, 1920x1080, 64 cameras), every frame compared bit for bit with the
synchronous render of its (mesh frame, camera): tests/test_gpu_parity.py::pipelined_deformed_loop with the C3 parameters."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from test_gpu_parity import pipelined_deformed_loop
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pipelined_deformed_loop(1_000_000, 1920, 1080, 64, frames)
print("pipelined loop at C3 size: %d frames, all bit-identical to the synchronous render" % frames)
