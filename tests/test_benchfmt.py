"""bench.py's stdout contract: the LAST line is one compact JSON record that survives the driver's 8 000-character tail
(round 4's 26.6 KB line did not: BENCH_r04.parsed = null).  Runs the formatter on round 4's recorded full result."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config")


def _recorded(world):
    out = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    if world > 1:       # what an --gpus 8 run adds: per-rank lists and one device record per rank
        out = copy.deepcopy(out)
        out["n_gpus"] = world
        out["config"]["global_batch"] = 4 * world
        r0 = out["distributed"]["ranks"][0]
        out["distributed"].update(world_size=world, backend="nccl", distinct_devices=world,
                                  ranks=[dict(r0, rank=i, local_rank=i, device_index=i, pci_bus_id=100 + i) for i in range(world)])
        out["per_rank"] = {"ms_per_step": [23.1 + 0.01 * i for i in range(world)],
                           "allreduce_exposed_ms_per_step": [1.9 + 0.01 * i for i in range(world)], "note": "x" * 200}
        out.pop("cpu_baseline", None)
        out.pop("operating_points", None)
    return out


@pytest.mark.parametrize("world", [1, 8])
def test_compact_line_fits_the_driver_tail(world):
    out = _recorded(world)
    line = bench.compact(out)
    assert "\n" not in line and len(line) < 4096 and len(line) <= bench.MAX_LINE
    tail = ("x" * 9000 + "\n" + line + "\n")[-8000:]            # what the driver keeps of stdout
    rec = json.loads(tail.strip().splitlines()[-1])
    for k in CONTRACT:
        assert k in rec, k
    assert rec["value"] == pytest.approx(out["value"], rel=1e-4) and rec["ms_per_step"] == pytest.approx(out["ms_per_step"], rel=1e-4)
    assert rec["n_gpus"] == world and rec["config"]["global_batch"] == 4 * world
    ro = rec["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "launches", "alg_bytes_per_launch",
              "selective_scan_op", "shared_chip"):
        assert k in ro, k
    assert ro["frac"] == pytest.approx(ro["achieved"] / ro["peak"], rel=1e-3)
    assert set(ro["shared_chip"]) >= {"frac", "op_frac"}
    assert set(rec["distributed"]) >= {"world_size", "backend", "rccl_version", "distinct_devices"} and "ranks" not in rec["distributed"]
    if world == 1:
        assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["cores"] == out["cpu_baseline"]["cores"]
        assert set(rec["operating_points"]) == set(out["operating_points"])
        for p in rec["operating_points"].values():
            assert set(p) >= {"value", "ms_per_step", "scan_op_frac"}
    else:
        assert rec["distributed"]["allreduce_exposed_ms_per_step"] == pytest.approx(1.97)
        assert rec["distributed"]["ms_per_step_max"] == pytest.approx(23.17)


def test_oversized_fields_are_dropped_not_emitted():
    out = _recorded(1)
    out["operating_points"] = {f"point_{i}": dict(v) for i in range(40) for v in [next(iter(out["operating_points"].values()))]}
    line = bench.compact(out)
    assert len(line) <= bench.MAX_LINE
    rec = json.loads(line)
    assert rec["operating_points"] == {"see": rec["detail"]} and "roofline" in rec and "value" in rec


def test_emit_prints_the_compact_line_last(tmp_path, capsys):
    out = _recorded(1)
    bench.emit(out, str(tmp_path / "bench_detail.json"))
    cap = capsys.readouterr()
    lines = cap.out.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) <= bench.MAX_LINE
    full = json.load(open(tmp_path / "bench_detail.json"))
    assert "kernels" in full["roofline"] and json.loads(lines[0])["detail"] == "bench_detail.json"


def test_compact_line_with_in_graph_collectives():
    """world > 1 on RCCL: the collectives are branches of the step's graph and cannot be bracketed by events — the per-rank exposed
    times are null and the line says where the collectives run."""
    out = _recorded(8)
    out["per_rank"]["allreduce_exposed_ms_per_step"] = [None] * 8
    out["per_rank"]["collectives"] = "branches of the step's graph (RCCL captured; MPD gradient as bf16)"
    rec = json.loads(bench.compact(out))
    assert rec["distributed"]["allreduce_exposed_ms_per_step"] is None and rec["distributed"]["collectives"].startswith("branches of the step")
    assert len(bench.compact(out)) <= bench.MAX_LINE
