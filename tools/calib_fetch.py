#!/usr/bin/env python3
"""Run the FETCH_SIZE calibration kernel (GPU box; wrap in rocprofv3 --pmc FETCH_SIZE)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maple_amd.runtime import Device
dev = Device(np.zeros(100, dtype=np.uint8), [0.25] * 4, arena_bytes=16 << 20, debug=True)
nbytes = 2 << 30
ms = dev.debug_calib_walk(nbytes, 3)
print(f"calib: {nbytes} bytes x 3 launches in {ms:.3f} ms -> {3 * nbytes / ms / 1e6:.1f} GB/s")
for mode, what in ((1, "coalesced 8-byte stream"), (2, "one 8-byte store per 64-byte line")):
    ms = dev.debug_calib_write(nbytes, mode, 1)
    print(f"calib write mode {mode} ({what}): buffer {nbytes} bytes in {ms:.3f} ms")
