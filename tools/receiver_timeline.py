"""Runs ON THE GPU BOX under rocprofv3 (see the command below): the receiver pipeline's last repetition as one time axis -- host-to-device
copies, kernels, device-to-host copies -- from the trace database.
  cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/rt -- python tools/bench_receiver.py 8192 1280 128 0.1
  python tools/receiver_timeline.py $(find /tmp/rt -name '*.db' | head -1)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kt = "kernels" if "kernels" in names else None
mt = next((n for n in ("memory_copies", "memory_copy") if n in names), None)
ev = []
if kt:
    for name, s, e in db.execute("select name, start, end from %s" % kt):
        ev.append((s, e, "kernel", name.split("(")[0][:60], 0))
if mt:
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % mt)]
    size_col = "size" if "size" in cols else ("bytes" if "bytes" in cols else None)
    name_col = "name" if "name" in cols else cols[0]
    q = "select %s, start, end%s from %s" % (name_col, (", " + size_col) if size_col else "", mt)
    for r in db.execute(q):
        ev.append((r[1], r[2], "copy", str(r[0])[:40], r[3] if size_col else 0))
ev.sort()
if not ev:
    print("no events; tables:", names)
    sys.exit(0)
# the last repair_all: from the last burst of big host-to-device copies on
t_end = ev[-1][1]
window = [x for x in ev if x[0] >= t_end - 60e6]   # the last 60 ms
t0 = window[0][0]
print("# the last 60 ms of the run (the last add_symbols_async + repair_all of tools/bench_receiver.py); times in ms from the window's start")
print("# kind    start     end      MB   what")
agg = {}
for s, e, kind, what, size in window:
    key = (kind, what)
    a = agg.setdefault(key, [s, e, 0, 0.0])
    a[0] = min(a[0], s); a[1] = max(a[1], e); a[2] += 1; a[3] += size / 1e6
for (kind, what), (s, e, n, mb) in sorted(agg.items(), key=lambda kv: kv[1][0]):
    print("%-7s %7.2f  %7.2f  %7.1f  %s x%d" % (kind, (s - t0) / 1e6, (e - t0) / 1e6, mb, what, n))
