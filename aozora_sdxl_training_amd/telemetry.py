"""Telemetry contract of the training loop: the stdout lines the reference's AsyncReporter prints
(train.py:381-458) and its GUI parses (gui.py:1853-1885), so that the unmodified GUI can monitor this trainer.

`progress_line` / `optimizer_block` build the strings; `Reporter` is the queue-fed printer with the reference's method
names (log_step / log_message / shutdown).  tests/test_telemetry.py compares the strings with what the reference's own
handlers printed for the same inputs (tests/golden/golden_data.json "telemetry")."""
from __future__ import annotations

import math
import queue
import threading

import torch

BAR_WIDTH = 30


def format_time(seconds) -> str:
    if seconds is None or not math.isfinite(seconds):
        return "N/A"
    s = int(seconds)
    return f"{s // 3600:02}:{(s % 3600) // 60:02}:{s % 60:02}"


def optimizer_block(d) -> str:
    """Per-optimizer-step block (train.py:416-423); `d` has the keys of diag_data_to_log (train.py:2791-2800)."""
    status = "[OK]" if d["update_delta"] > 1e-12 else "[NO UPDATE!]"
    reserved = torch.cuda.memory_reserved() / 1e9 if torch.cuda.is_available() else 0.0
    allocated = torch.cuda.memory_allocated() / 1e9 if torch.cuda.is_available() else 0.0
    return (f"\n--- Optimizer Step: {d['optim_step']:<5} | Loss: {d['avg_loss']:<8.5f} | LR: {d['current_lr']:.2e} ---\n"
            f"  Time: {d['optim_step_time']:.2f}s/step | Avg Speed: {d['avg_optim_step_time']:.2f}s/step\n"
            f"  Grad Norm (Raw/Clipped): {d['raw_grad_norm']:<8.4f} / {d['clipped_grad_norm']:<8.4f}\n"
            f"  VRAM: Training={reserved:.2f}GB | Model={allocated:.2f}GB\n"
            f"  |- Update Magnitude : {d['update_delta']:.4e} {status}\n")


def progress_line(global_step, total_steps, timing) -> str:
    """The carriage-return progress line (train.py:425-441)."""
    frac = (global_step + 1) / total_steps
    filled = int(BAR_WIDTH * frac)
    sigma = timing.get("sigma")
    ticket = timing.get("timestep", "N/A")
    sampling = f"Ticket: {ticket}, Sigma: {float(sigma):.6f}" if sigma is not None else f"Timestep: {ticket}"
    return (f"Training |{'#' * filled}{'-' * (BAR_WIDTH - filled)}| {global_step + 1}/{total_steps}[{frac:.2%}]"
            f"[Loss: {timing.get('loss', 0.0):.4f}, {sampling}]"
            f"[{timing.get('raw_step_time', 0):.2f}s/step, ETA: {format_time(timing.get('eta'))}, Elapsed: {format_time(timing.get('elapsed_time'))}]")


class Reporter:
    """Prints from a worker thread so the training thread never blocks on stdout (train.py:381-458)."""

    def __init__(self, total_steps, test_param_name="conv_in", asynchronous=True):
        self.total_steps, self.test_param_name = total_steps, test_param_name
        self._last_line_len = 0
        self._async = asynchronous
        if asynchronous:
            self._q = queue.Queue()
            self._stop = threading.Event()
            self._t = threading.Thread(target=self._run, daemon=True)
            self._t.start()

    def _clear_line(self):
        if self._last_line_len > 0:
            print("\r" + " " * self._last_line_len + "\r", end="", flush=True)
            self._last_line_len = 0

    def _emit_step(self, global_step, timing_data, diag_data):
        if diag_data:
            self._clear_line()
            print(optimizer_block(diag_data))
        line = progress_line(global_step, self.total_steps, timing_data)
        print("\r" + line, end="", flush=True)
        self._last_line_len = len(line)

    def _emit_message(self, text):
        self._clear_line()
        print(text)

    def _run(self):
        while not self._stop.is_set():
            try:
                kind, payload = self._q.get(timeout=0.05)
            except queue.Empty:
                continue
            (self._emit_step if kind == "step" else self._emit_message)(*payload)
            self._q.task_done()

    def log_step(self, global_step, timing_data, diag_data=None):
        if self._async:
            self._q.put(("step", (global_step, timing_data, diag_data)))
        else:
            self._emit_step(global_step, timing_data, diag_data)

    def log_message(self, text):
        if self._async:
            self._q.put(("msg", (text,)))
        else:
            self._emit_message(text)

    def shutdown(self):
        self._clear_line()
        print("\nShutting down async reporter. Waiting for pending tasks...")
        if self._async:
            self._q.join()
            self._stop.set()
            self._t.join()
