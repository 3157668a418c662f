"""bench.py's own launcher (`python bench.py --gpus N` started bare): child supervision.  CPU-only: the rank program is a stub."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _stub(tmp_path, body):
    p = tmp_path / "rank_stub.py"
    p.write_text("import os, sys, time\nrank = int(os.environ['RANK'])\n" + body)
    return str(p)


def test_launcher_stops_the_other_ranks_when_one_dies(tmp_path, monkeypatch, capfd):
    """A rank that dies before rendezvous must not leave the launcher (and the surviving ranks) waiting out a collective timeout:
    the first non-zero exit terminates the rest and is the launcher's exit code."""
    import bench
    monkeypatch.setattr(bench, "__file__", _stub(tmp_path, "if rank == 1:\n    sys.exit(3)\ntime.sleep(120)\n"))
    t0 = time.monotonic()
    rc = bench.spawn_ranks(2, [])
    assert rc == 3 or rc == -15, rc          # worst code by magnitude: the dead rank's 3 or the terminated survivor's SIGTERM
    assert time.monotonic() - t0 < 30
    assert "stopping the other ranks" in capfd.readouterr().err


def test_launcher_relays_rank0_json_and_returns_zero(tmp_path, monkeypatch, capfd):
    import bench
    monkeypatch.setattr(bench, "__file__", _stub(tmp_path, "print('chatter')\nif rank == 0:\n    print('{\"ok\": 1}')\n"))
    assert bench.spawn_ranks(2, []) == 0
    out = capfd.readouterr().out
    assert out.strip().splitlines() == ['{"ok": 1}']


def test_launcher_overall_timeout(tmp_path, monkeypatch, capfd):
    import bench
    monkeypatch.setattr(bench, "__file__", _stub(tmp_path, "time.sleep(120)\n"))
    monkeypatch.setenv("BENCH_SPAWN_TIMEOUT", "2")
    t0 = time.monotonic()
    rc = bench.spawn_ranks(2, [])
    assert rc != 0 and time.monotonic() - t0 < 30
    assert "overall timeout" in capfd.readouterr().err
