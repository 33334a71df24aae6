"""Per-kernel statistics from a rocprofv3 results database (`rocprofv3 --kernel-trace -d DIR -o NAME` writes NAME_results.db).
usage: python tools/db_stats.py path/to/results.db [rows] [steps]
With `steps` (rollout steps the run made, warm-up included) the last column is the kernel's time per step and the summary line gives
the busy time per step against the span from the first kernel's start to the last one's end."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 30
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
q = "select name, count(*), avg(end - start), min(end - start), max(end - start), sum(end - start) from kernels group by name order by 6 desc"
res = list(db.execute(q))
tot = sum(r[5] for r in res)
for name, calls, avg, mn, mx, s in res[:rows]:
    per = f" per-step={s / steps / 1e3:8.1f}us" if steps else ""
    print(f"{name[:84]:84s} calls={calls:5d} avg={avg / 1e3:8.1f}us min={mn / 1e3:8.1f} max={mx / 1e3:8.1f} {100 * s / tot:5.2f}%{per}")
if steps:
    t0, t1 = db.execute("select min(start), max(end) from kernels").fetchone()
    print(f"busy per step {tot / steps / 1e3:.1f} us; span per step {(t1 - t0) / steps / 1e3:.1f} us (includes set-up before the first step)")
