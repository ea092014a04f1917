"""A compiled caller of the ABI that is not Python: tests/c_abi/caller.c includes include/aruco3_hip.h under
`gcc -std=c11 -pedantic -Wall -Wextra -Werror`, links libaruco3_hip.so and
  * (CPU) prints sizeof / offsetof of every struct field -> compared with the `#[repr(C)]` declarations of integration/aruco3_hip.rs
    (laid out by the C rules from their Rust field types) and with the ctypes structures of aruco3_amd/_lib.py;
  * (GPU) runs tests/fixtures/inputs/c1_640x480_aruco.raw through a3_create -> a3_detect_batch -> a3_destroy and prints the markers
    -> compared with tests/golden/c1_640x480_aruco.npz.
The reference's interface it stands for: src/lib.rs:6-9, src/aruco.rs:46-52."""
import ctypes as C
import json
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "c_abi" / "caller.c"


@pytest.fixture(scope="module")
def caller(tmp_path_factory):
    exe = tmp_path_factory.mktemp("c_abi") / "caller"
    lib_dir = ROOT / "aruco3_amd"
    assert (lib_dir / "libaruco3_hip.so").exists(), "build the library first (__graft_entry__.build())"
    cmd = ["gcc", "-std=c11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", str(ROOT / "include"), str(SRC), "-o", str(exe),
           "-L", str(lib_dir), "-laruco3_hip", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath-link,/opt/rocm/lib"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return exe


def _layout(caller):
    p = subprocess.run([str(caller), "layout"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
    sizes, fields = {}, {}
    for ln in p.stdout.splitlines():
        name, off, size = ln.split()
        if off == "-":
            sizes[name] = int(size)
        else:
            s, f = name.split(".")
            fields.setdefault(s, []).append((f, int(off), int(size)))
    return sizes, fields


# C layout of a #[repr(C)] struct from its Rust field types
_RUST = {"u8": 1, "u16": 2, "u32": 4, "u64": 8, "i32": 4, "f32": 4, "f64": 8}
_RUST_OF_C = {"a3_config": "A3Config", "a3_marker": "A3Marker", "a3_pose": "A3Pose", "a3_intrinsics": "A3Intrinsics", "a3_stats": "A3Stats",
              "a3_synth_marker": "A3SynthMarker", "a3_synth_frame": "A3SynthFrame"}


def _rust_layout(name):
    text = re.sub(r"//[^\n]*", "", (ROOT / "integration" / "aruco3_hip.rs").read_text())
    body = re.search(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\]]*\)\]\s*)?pub struct %s \{(.*?)\}" % name, text, flags=re.S).group(1)
    out, off, align_max = [], 0, 1
    for fname, ty in re.findall(r"pub (\w+):\s*([^,\n]+),", body):
        ty = ty.strip()
        m = re.fullmatch(r"\[(\w+);\s*(\d+)\]", ty)
        elem, count = (m.group(1), int(m.group(2))) if m else (ty, 1)
        a = _RUST[elem]
        off = (off + a - 1) // a * a
        out.append((fname, off, a * count))
        off += a * count
        align_max = max(align_max, a)
    return out, (off + align_max - 1) // align_max * align_max


def test_header_compiles_as_c11_and_layouts_match_rust_and_ctypes(caller):
    from aruco3_amd import _lib

    sizes, fields = _layout(caller)
    assert sizes.pop("A3_ABI_VERSION") == _lib.load().a3_abi_version()      # the header the program saw and the library it linked
    assert set(sizes) == set(_RUST_OF_C)
    for c_name, rust_name in _RUST_OF_C.items():
        rust_fields, rust_size = _rust_layout(rust_name)
        assert fields[c_name] == rust_fields, (c_name, fields[c_name], rust_fields)
        assert sizes[c_name] == rust_size, c_name
    # the ctypes mirror the GPU tests go through
    for c_name, cls in (("a3_config", _lib.Config), ("a3_marker", _lib.MarkerRec), ("a3_pose", _lib.PoseRec), ("a3_intrinsics", _lib.Intrinsics),
                        ("a3_stats", _lib.Stats)):
        assert C.sizeof(cls) == sizes[c_name], c_name
        got = [(n, getattr(cls, n).offset, getattr(cls, n).size) for n, _ in cls._fields_ if not n.startswith("_")]
        assert [(o, s) for _, o, s in got] == [(o, s) for _, o, s in fields[c_name]], (c_name, got, fields[c_name])
    assert sizes["a3_marker"] == 56 and sizes["a3_pose"] == 52      # SURVEY 8e's record sizes


def test_usage_line_without_arguments(caller):
    p = subprocess.run([str(caller)], capture_output=True, text=True, timeout=60)
    assert p.returncode == 64 and "usage" in p.stderr


@pytest.mark.gpu
def test_c_caller_detects_the_config1_fixture(caller):
    """BASELINE config 1 (one 640x480 frame, 4 ARUCO_DEFAULT markers) from a C program: markers equal the golden vector"""
    index = json.loads((ROOT / "aruco3_amd" / "data" / "dictionaries.json").read_text())["ARUCO_DEFAULT"]
    z = np.load(ROOT / "tests" / "golden" / "c1_640x480_aruco.npz")
    raw = ROOT / "tests" / "fixtures" / "inputs" / "c1_640x480_aruco.raw"
    assert raw.stat().st_size == 640 * 480 * 3
    p = subprocess.run([str(caller), "detect", str(raw), "640", "480", "0", str(ROOT / "aruco3_amd" / "data" / "dictionaries.bin"),
                        str(index["offset"]), str(index["count"]), str(index["num_bits"]), str(index["tau"])], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.stdout, p.stderr)
    lines = p.stdout.splitlines()
    cfg = next(ln for ln in lines if ln.startswith("config ")).split()[1:]
    assert (int(cfg[0]), float(cfg[1]), float(cfg[2]), float(cfg[3]), int(cfg[4]), int(cfg[5])) == (7, 0.05, np.float32(0.2), np.float32(0.1), 49, 1)   # src/aruco.rs:32-43
    head = next(ln for ln in lines if ln.startswith("markers ")).split()
    assert int(head[1]) == int(head[3]) == len(z["marker_id"]) and int(head[5]) == len(z["candidates_pre"]) and int(head[7]) == len(z["candidates"])
    got = [[int(v) for v in ln.split()[1:]] for ln in lines if ln.startswith("marker ")]
    want = [[0, int(z["marker_id"][i]), int(z["marker_code"][i]), int(z["marker_hamming"][i]), int(z["marker_rotation"][i])] for i in range(len(z["marker_id"]))]
    assert [g[:5] for g in got] == want
    assert [g[6:] for g in got] == [[int(v) for v in z["marker_corners"][i].reshape(-1)] for i in range(len(want))]
    assert next(ln for ln in lines if ln.startswith("too_small_rc")).split()[1] == "-3"        # A3_ERR_CAPACITY, never a clip
    assert next(ln for ln in lines if ln.startswith("null_pixels_rc")).split()[1] == "-1"      # A3_ERR_INVALID
    e0, e1 = (float(v) for v in next(ln for ln in lines if ln.startswith("pose_errors")).split()[1:])
    assert 0.0 <= e0 <= e1
