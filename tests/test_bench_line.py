"""The line the driver parses: bench.py's LAST -- and only -- stdout line is compact: the contract's keys, `roofline`,
`roofline_warp`, `cpu_baseline`, `parity_in_run`, at most 4 KB whatever the run measured; everything else goes to
bench_detail.json / stderr.  CPU-only: the line is built from canned dicts (round 5's full 22 KB line among them)."""
import io
import json
import sys
from contextlib import redirect_stderr, redirect_stdout
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402  (imports numpy only; torch and the library are loaded inside main())

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "timed_region_s_total", "stepping", "gates", "detail")
ROOFLINE = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "bytes_per_launch", "avg_launch_ms", "launches_timed")


def _round5_line():
    log = ROOT / "profiles" / "r05_bench_c2.json.log"
    d = json.loads([ln for ln in log.read_text().splitlines() if ln.startswith("{")][-1])
    d["stepping_word"] = "free-running"
    d["roofline"]["traffic_source"] = "profiles/r05_pmc_bench_c2.json (committed PMC pass of this command, not this run)"
    d["roofline_warp"]["traffic_source"] = "profiles/r05_pmc_chain.json (committed PMC pass, not this run)"
    return d


def _check(line, full):
    text = json.dumps(line)
    assert len(text.encode()) <= bench.LINE_BUDGET == 4096
    for k in REQUIRED:
        assert k in line, k
    for k in ROOFLINE:
        assert k in line["roofline"], k
    assert line["value"] == full["value"] and line["roofline"]["frac"] == full["roofline"]["frac"]
    assert all(len(v) <= 160 for v in _strings(line)), "no prose in the line"


def _strings(o):
    if isinstance(o, str):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, list):
        for v in o:
            yield from _strings(v)


def test_round5_line_shrinks_below_the_budget_and_keeps_what_the_driver_reads():
    full = _round5_line()
    assert len(json.dumps(full)) > 20000           # the line the driver could not read
    line = bench.compact_line(full)
    _check(line, full)
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert (line["parity_in_run"]["frames_equal"], line["parity_in_run"]["frames_compared"]) == (1024, 1024)
    assert line["roofline_warp"]["frac_of_request_ceiling"] == full["roofline_warp"]["request_rate"]["frac_of_ceiling_whole_kernel"]
    assert "not this run" in line["roofline"]["traffic_source"]
    assert line["other_workloads"]["C0_reference_bench_noise_1080p"][:2] == [12380.6, 14040.2]
    assert "ms_per_step_all" not in line and line["stepping"] == "free-running"


def test_line_stays_in_budget_when_everything_grows():
    """ten times the workloads, kilobytes of prose, thousands of regions: optional blocks are dropped, the required keys stay"""
    full = _round5_line()
    full["ms_per_step_all"] = [0.633] * 5000
    full["stepping"] = "prose " * 2000
    row = full["other_workloads"]["C0_reference_bench_noise_1080p"]
    for i in range(200):
        full["other_workloads"][f"another_workload_with_a_long_name_{i:03d}"] = row
    full["gathered"] = {"frames": 1, "global_frame_indices_in_order": True, "trace": ["x" * 100] * 100}
    full["dist"] = {"backend": "nccl", "world_size": 8, "launcher": "y" * 500, "hw_queues": list(range(1000))}
    line = bench.compact_line(full)
    _check(line, full)
    assert "cpu_baseline" in line and "parity_in_run" in line and "roofline_warp" in line


def test_minimal_run_without_the_side_measurements():
    """--no-other-workloads --no-cpu-baseline on N ranks: no roofline_warp, no cpu_baseline, a `gathered` block"""
    full = _round5_line()
    for k in ("roofline_warp", "cpu_baseline", "parity_in_run", "other_workloads", "ab_shared_stream", "ab_burst_gates", "ab_r04_library_default"):
        full.pop(k)
    full["n_gpus"] = 8
    full["gathered"] = {"frames": 8192, "global_frame_indices_in_order": True, "all_ranks_ids_correct": 7700, "collectives": 17}
    line = bench.compact_line(full)
    _check(line, full)
    assert line["gathered"]["frames"] == 8192 and line["n_gpus"] == 8


def test_emit_prints_one_stdout_line_and_the_detail_elsewhere(tmp_path, monkeypatch):
    full = _round5_line()
    monkeypatch.setenv("A3_BENCH_DETAIL", str(tmp_path / "bench_detail.json"))
    so, se = io.StringIO(), io.StringIO()
    with redirect_stdout(so), redirect_stderr(se):
        bench.emit(full)
    lines = so.getvalue().splitlines()
    assert len(lines) == 1 and lines[0].startswith("{") and len(lines[0].encode()) <= 4096
    assert json.loads(lines[0])["detail"] == str(tmp_path / "bench_detail.json")
    assert json.loads((tmp_path / "bench_detail.json").read_text()) == full
    det = [ln for ln in se.getvalue().splitlines() if ln.startswith("bench_detail ")]
    assert len(det) == 1 and json.loads(det[0][len("bench_detail "):]) == full
