"""Per-phase clock of one command (tools/cli_timeline.py): with SVX_CLI_TIMELINE=<file> in the environment every
mark(name) notes wall time (time.time(), comparable with the parent's clock), the CPU seconds of the whole process so
far (all threads) and the calling thread; dump() writes them as JSON lines.  Without the variable mark() is one
dictionary look-up.  Needs nothing but the standard library (bin/svim-asm marks before numpy is imported)."""
import os
import threading
import time

_PATH = os.environ.get("SVX_CLI_TIMELINE")
_marks = []
_lock = threading.Lock()


def enabled():
    return _PATH is not None


def mark(name, **extra):
    if _PATH is None:
        return
    rec = {"name": name, "t": time.time(), "cpu": time.process_time(), "thread": threading.current_thread().name}
    rec.update(extra)
    with _lock:
        _marks.append(rec)


def dump():
    if _PATH is None:
        return
    import json
    with _lock:
        rows = list(_marks)
    with open(_PATH, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
