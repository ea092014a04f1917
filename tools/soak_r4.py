#!/usr/bin/env python3
"""Round-4 soak (GPU box): tests/test_gpu_config_fuzz.py's run_case as a driver over N random configuration x dictionary x format x
size combinations -- every threshold_window 3..12 and, in every third case, 8..33 among them, i.e. the radius-templated threshold kernels, the ring kernel in all its splits and the separable path -- plus
tools/fuzz_soak.py's structured-frame seeds.  Prints a progress line every 100 cases.   python tools/soak_r4.py [cases] [seeds]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import test_gpu_config_fuzz as F
import test_gpu_parity as T
from oracle import a3oracle
from aruco3_amd.dictionaries import ARDictionary


class Dicts:
    new_from_named_dict = staticmethod(ARDictionary.new_from_named_dict)


cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 400
a3oracle.build()
t0 = time.time(); found = 0
for i in range(cases):
    found += F.run_case(a3oracle, (500000 if i % 3 else 1500000) + i)   # (every third case: windows 8..16, widths the fused kernel takes)
    if i % 100 == 99:
        print(f"config fuzz: {i + 1} cases, {found} markers decoded, {time.time() - t0:.0f} s", flush=True)
fn = T.test_randomised_structured_frames_full_parity
fn = getattr(fn, "__wrapped__", fn)
for s in range(seeds):
    fn(Dicts, a3oracle, 70000 + s)
    if s % 50 == 49:
        print(f"structured frames: {s + 1} seeds, {time.time() - t0:.0f} s", flush=True)
print(f"SOAK OK: {cases} configuration cases ({found} markers), {seeds} structured seeds, {time.time() - t0:.0f} s")
