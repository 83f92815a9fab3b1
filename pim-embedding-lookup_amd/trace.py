"""Stage-latency export: the reference's two profiling artefacts, fed
from the engine's stage timers instead of UPMEM's --

  * interval CSV with header "DPU, Start, Stop" (upmem/dputypes.py:87-98 `write_results`; plotted as
    a Gantt chart by graph/runtime_intervals/main.py, which reads column 0 as the lane and columns
    1, 2 as start/stop).  Lane = stage id here (0 copy-in, 1 descriptors, 2 launch, 3 copy-out, 4 sync);
  * Chrome trace JSON of begin/end events like upmem/test.json (`chrome://tracing`, Perfetto)."""
from __future__ import annotations

import csv
import ctypes as C
import json

from . import lib as _l

STAGE_NAMES = {0: "copy_in_indices_offsets", 1: "copy_in_descriptors", 2: "launch", 3: "copy_out_results", 4: "sync"}
# names the reference's trace uses for the corresponding SDK calls (upmem/test.json)
REFERENCE_NAMES = {0: "dpu_push_xfer(input_indices,input_offsets)", 1: "dpu_push_xfer(input_lengths)",
                   2: "dpu_launch", 3: "dpu_push_xfer(results)", 4: "dpu_sync"}


def enable(engine, capacity: int = 65536) -> None:
    _l.check(engine._L.emb_trace_enable(engine._h, capacity))


def read(engine, max_events: int = 65536):
    buf = (_l.EmbTraceEvent * max_events)()
    n = C.c_uint32()
    _l.check(engine._L.emb_trace_read(engine._h, buf, max_events, C.byref(n)))
    return [dict(stage=buf[i].stage, call_id=buf[i].call_id, start_us=buf[i].start_us, stop_us=buf[i].stop_us)
            for i in range(n.value)]


def write_interval_csv(events, path: str, headers=("DPU", "Start", "Stop")) -> None:
    """Same shape as upmem/dputypes.py::write_results: one row per interval, times in milliseconds
    with 16 decimals."""
    with open(path, "w", newline="") as f:
        w = csv.writer(f, delimiter=",", quotechar='"', quoting=csv.QUOTE_MINIMAL)
        if headers:
            w.writerow(headers)
        for ev in events:
            w.writerow([ev["stage"], f'{ev["start_us"] / 1e3:.16f}', f'{ev["stop_us"] / 1e3:.16f}'])


def write_chrome_trace(events, path: str, pid: int = 1) -> None:
    out = []
    for ev in events:
        name = STAGE_NAMES.get(ev["stage"], str(ev["stage"]))
        args = {"call_id": ev["call_id"], "reference_call": REFERENCE_NAMES.get(ev["stage"], "")}
        out.append({"name": name, "cat": "pimemb", "pid": str(pid), "tid": str(pid), "ts": ev["start_us"], "ph": "B",
                    "args": args})
        out.append({"name": name, "cat": "pimemb", "pid": str(pid), "tid": str(pid), "ts": ev["stop_us"], "ph": "E",
                    "args": {}})
    with open(path, "w") as f:
        json.dump({"traceEvents": out}, f)
