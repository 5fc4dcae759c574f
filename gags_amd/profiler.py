"""Per-stage GPU timing with HIP events on the stream the kernels are launched on
(torch's current stream).  Disabled by default; bench.py turns it on to obtain the average
launch duration of the dominant kernel for the roofline line."""
from collections import defaultdict
from contextlib import contextmanager

import torch

ENABLED = False
_events = defaultdict(list)


def enable(flag=True):
    global ENABLED
    ENABLED = bool(flag)
    _events.clear()


@contextmanager
def stage(name):
    if not ENABLED:
        yield
        return
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    try:
        yield
    finally:
        b.record()
        _events[name].append((a, b))


_notes = {}


def note(key, value):
    """Remember a scalar fact about the last call (e.g. row count of the staged backward)."""
    if ENABLED:
        _notes[key] = value


def notes():
    return dict(_notes)


def summary():
    """{stage: (mean_ms, count)}; synchronizes."""
    torch.cuda.synchronize()
    out = {}
    for k, evs in _events.items():
        ms = [a.elapsed_time(b) for a, b in evs]
        out[k] = (sum(ms) / max(len(ms), 1), len(ms))
    return out


def reset():
    _events.clear()
