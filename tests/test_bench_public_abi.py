"""The headline of bench.py must be reproducible through include/aruco3_hip.h alone: everything bench.py does before
its A/B block -- set-up, isolated launches, the burst stepping, the in-burst roofline, the timed regions -- calls no a3_debug_* /
a3_selftest_* symbol, except behind the explicit --overlap measurement flag, which is off by default and reported in the JSON line.
CPU-only: the file is read as text."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
BENCH = (ROOT / "bench.py").read_text()
MARK = "# ==== A/B block:"


def _main_body():
    a = BENCH.index("def main():")
    b = BENCH.index("\ndef launch_ranks(")
    return BENCH[a:b]


def test_headline_path_calls_no_internal_symbol():
    body = _main_body()
    assert body.count(MARK) == 1
    head = body[: body.index(MARK)]
    # comments and help strings may name internal symbols; calls may not
    calls = [m.group(0) for m in re.finditer(r"\bL\.(a3_debug_[a-z0-9_]+|a3_selftest_[a-z0-9_]+)\s*\(", head)]
    assert calls == ["L.a3_debug_set_overlap("], calls                    # the one call there is ...
    guarded = re.search(r"if args\.overlap >= 0:\s*\n\s*assert L\.a3_debug_set_overlap\(args\.overlap\) == 0\s*\n\s*internal_switches_used\.append", head)
    assert guarded, "... sits behind --overlap (default -1 = off) and is reported in library.internal_switches_used"
    assert re.search(r'add_argument\("--overlap", type=int, default=-1', BENCH)
    # no other route to the internal header: no debug_* helper of the binding, no ctypes lookup by name
    assert not re.search(r"\.debug_[a-z_]+\(", head)
    assert "getattr(L" not in head and "a3_internal" not in re.sub(r"#[^\n]*", "", re.sub(r'help="[^"]*"', "", head))


def test_public_names_used_by_the_headline_are_in_the_header():
    """every method the headline calls on a context maps onto a function the public header declares"""
    header = (ROOT / "include" / "aruco3_hip.h").read_text()
    for sym in ("a3_order_after", "a3_detect_batch_submit", "a3_detect_batch_collect", "a3_detect_batch_pose_submit", "a3_detect_batch_pose_collect",
                "a3_set_profiling", "a3_get_profile", "a3_get_stream", "a3_get_stats", "a3_pack_detections", "A3_STEP_HELD_RELEASED_BY_LAST",
                "A3_STEP_BURST_LAST"):
        assert sym in header, sym
    lib = (ROOT / "aruco3_amd" / "_lib.py").read_text()
    for meth, sym in (("order_after", "a3_order_after"), ("submit", "a3_detect_batch_submit"), ("collect", "a3_detect_batch_collect"),
                      ("set_profiling", "a3_set_profiling"), ("profile", "a3_get_profile"), ("stats", "a3_get_stats")):
        m = re.search(r"    def %s\(self.*?(?=\n    def |\nclass |\Z)" % meth, lib, flags=re.S)
        assert m and sym in m.group(0), (meth, sym)
