"""the reporter's stdout lines equal the reference AsyncReporter's for the same inputs
(tests/golden/golden_data.json "telemetry", captured from the reference's own handlers by make_golden_data.py)."""
import contextlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from make_golden_data import TELEMETRY_CASES                     # noqa: E402
from aozora_sdxl_training_amd import telemetry as T              # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_data.json")))["telemetry"]


def test_lines_equal_reference():
    rep = T.Reporter(total_steps=1000, asynchronous=False)
    got = []
    for case in TELEMETRY_CASES:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            rep._last_line_len = 0
            rep.log_step(case["global_step"], case["timing"], case["diag"])
        got.append(buf.getvalue())
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rep._last_line_len = 17
        rep.log_message("hello")
    got.append(buf.getvalue())
    assert got == GOLD["lines"]
    assert [T.format_time(x) for x in (None, float("inf"), 0, 59.9, 3600, 86399, 360000)] == GOLD["times"]


def test_async_reporter_drains_in_order():
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rep = T.Reporter(total_steps=10)
        for i in range(5):
            rep.log_step(i, dict(loss=float(i), timestep=str(i)))
        rep.log_message("done")
        rep.shutdown()
    out = buf.getvalue()
    pos = [out.index(f" {i + 1}/10[") for i in range(5)]
    assert pos == sorted(pos) and "done" in out and "Waiting for pending tasks..." in out and out.index("done") > pos[-1]
