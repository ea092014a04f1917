"""The C-ABI library loads without a GPU and exports every symbol include/aruco3_hip.h declares (no compute calls)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared(path=None):
    text = (path or ROOT / "include" / "aruco3_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(a3_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from aruco3_amd import _lib

    L = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_lib.SYMBOLS) == names
    assert L.a3_abi_version() == 5
    # tuning probes and single-stage hooks live in an internal header, out of the binding surface
    internal = _declared(ROOT / "aruco3_amd" / "csrc" / "a3_internal.h")
    assert sorted(_lib.INTERNAL_SYMBOLS) == internal and not set(internal) & set(names)
    for n in internal:
        assert hasattr(L, n), n
    assert L.a3_detection_record_bytes(32, 0) == 8 + 32 * 56 and L.a3_detection_record_bytes(16, 1) == 8 + 16 * (56 + 104)


def test_struct_layouts_match_header():
    from aruco3_amd import _lib

    assert C.sizeof(_lib.MarkerRec) == 56          # u32 u32 u64 u32[8] u8 u8 u16
    assert C.sizeof(_lib.PoseRec) == 52            # 13 floats
    assert C.sizeof(_lib.Config) == 32
    assert C.sizeof(_lib.Intrinsics) == 24
    cfg = _lib.default_config()                    # src/aruco.rs:32-43
    assert (cfg.threshold_window, cfg.contour_simplification_epsilon, cfg.homography_sample_size, cfg.filter_high_bit_errors) == (7, 0.05, 49, 1)
    assert abs(cfg.min_side_length_factor - 0.2) < 1e-7 and abs(cfg.min_corner_separation_factor - 0.1) < 1e-7


def test_no_device_fails_loudly():
    """Without a GPU the product path must raise, never fall back to host arithmetic."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import numpy as np

    from aruco3_amd import _lib
    from aruco3_amd.aruco import Detector

    with pytest.raises(_lib.A3Error) as e:
        Detector().detect(np.zeros((8, 8, 3), np.uint8))
    assert e.value.code == _lib.ERR_NO_DEVICE
    from aruco3_amd.dictionaries import ARDictionary

    with pytest.raises(_lib.A3Error):
        ARDictionary.new_from_named_dict("ARUCO").find_nearest(0x1084210)


def test_product_library_reads_no_environment():
    """Tuning knobs are compile-time (-DA3_TUNING, `make tuning` into build/tuning/): the product library must not import
    getenv, and the only getenv in the sources sits inside the A3_TUNING block of a3_common.h."""
    import subprocess

    from aruco3_amd import _lib

    csrc = ROOT / "aruco3_amd" / "csrc"
    hits = []
    for f in sorted(csrc.glob("*.hip")) + sorted(csrc.glob("*.h")):
        for i, line in enumerate(f.read_text().splitlines(), 1):
            if "getenv" in line:
                hits.append((f.name, i))
    assert [h[0] for h in hits] == ["a3_common.h"], hits
    text = (csrc / "a3_common.h").read_text()
    block = text[text.index("#ifdef A3_TUNING\ninline int tuning_knob"): text.index("#else\nconstexpr int tuning_knob")]
    assert "getenv" in block
    assert _lib.LIB_PATH == ROOT / "aruco3_amd" / "libaruco3_hip.so"        # (A3_HIP_LIB is for the sweep scripts only)
    syms = subprocess.run(["nm", "-D", "--undefined-only", str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms
