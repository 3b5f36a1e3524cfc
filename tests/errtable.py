"""Achieved-error bookkeeping for the parity tests: every `_close` comparison of the kernel / module / MPD parity
files reports  max|got - want|, max|want|  and the tolerance it was held to; with VMASR_PARITY_TABLE=<path> the session writes one
markdown table (profiles/r06_parity_table.md), so that "1e-4 absolute" and "1e-4 of the tensor's max" are visible per tensor."""
import os

import numpy as np

ROWS = {}


def record(what, got, want, tol_abs):
    """got / want: float64 numpy arrays of one comparison; tol_abs: the largest absolute deviation the assertion allowed at the worst
    element (atol + rtol |want| there).  Kept: the worst comparison of each (test, tensor)."""
    if not os.environ.get("VMASR_PARITY_TABLE"):
        return
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    test = test.replace("tests/", "")
    d = np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64))
    err = float(d.max()) if d.size else 0.0
    scale = float(np.abs(want).max()) if np.size(want) else 0.0
    key = (test, what.split(" ")[-1] if what else "")
    row = dict(err=err, scale=scale, tol=float(tol_abs), n=int(np.size(want)))
    old = ROWS.get(key)
    if old is None or err / max(row["tol"], 1e-300) > old["err"] / max(old["tol"], 1e-300):
        ROWS[key] = row


def write(path):
    if not ROWS:
        return
    by_file = {}
    for (test, what), r in ROWS.items():
        by_file.setdefault(test.split("::")[0], []).append((test.split("::", 1)[-1], what, r))
    with open(path, "w") as f:
        f.write("# Parity table — achieved errors of the GPU parity tests (one row per test x tensor: its worst comparison)\n\n"
                "`abs err` = max|HIP - expected| over the tensor, `max|want|` the expected tensor's largest magnitude, `scaled` = abs err / max(1, max|want|)\n"
                "(the north-star gate is 1e-4 fp32 / 1e-2 bf16), `allowed` the absolute deviation the assertion permitted at its worst element, `used` = abs err / allowed.\n"
                "Expected values: committed goldens generated from the reference (tests/golden/make_golden.py) or the CPU oracle (oracle/) on the same inputs.\n\n")
        for fname in sorted(by_file):
            rows = sorted(by_file[fname], key=lambda t: -(t[2]["err"] / max(t[2]["tol"], 1e-300)))
            f.write(f"## {fname}  ({len(rows)} rows; worst first, top 40 shown)\n\n| test | tensor | abs err | max\\|want\\| | scaled | allowed | used |\n|---|---|---|---|---|---|---|\n")
            for test, what, r in rows[:40]:
                f.write(f"| {test[:90]} | {what[:30]} | {r['err']:.2e} | {r['scale']:.2e} | {r['err'] / max(1.0, r['scale']):.2e} | {r['tol']:.2e} | {r['err'] / max(r['tol'], 1e-300):.2f} |\n")
            worst_scaled = max(t[2]["err"] / max(1.0, t[2]["scale"]) for t in rows)
            worst_abs = max(t[2]["err"] for t in rows)
            f.write(f"\nall {len(rows)} rows: worst abs err {worst_abs:.2e}, worst scaled {worst_scaled:.2e}\n\n")
