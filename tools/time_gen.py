#!/usr/bin/env python3
"""Where the frame generator's time goes: esn_gen_frames at the benchmark size (2048 blocks x 75 frames) with parts knocked
out through the debug knob gen_ko (bit0 no AWGN draw, bit1 no channel MACs, bit2 no IFFT, bit3 no PA; timing only)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from esn_ofdm_mimo_amd import _lib  # noqa: E402
from esn_ofdm_mimo_amd.montecarlo import FrameSource, LinkParams  # noqa: E402

fs = FrameSource(LinkParams(), seed=1)
G, F = 2048, 75
taps = fs.taps(G, 0, 0)
for name, ko in (("full", 0), ("no noise draw", 1), ("no channel", 2), ("no IFFT", 4), ("no PA", 8), ("nothing but bits + stores", 15)):
    _lib.debug_set("gen_ko", ko)
    for _ in range(2):
        fs.frames(taps, F, 12.0, 0, 0, 1)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in ev:
        a.record(); fs.frames(taps, F, 12.0, 0, 0, 1); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)[2]
    print(f"{name:28s} {ms:7.3f} ms per {G * F} frames (incl. the output allocations of the wrapper)")
_lib.debug_set("gen_ko", 0)
