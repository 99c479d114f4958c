"""How much the sub-batch chains overlap, from a rocprofv3 --kernel-trace CSV: for the step kernel's dispatches of the
steady state (the last `n` of them) the average number in flight (sum of durations / covered span), the share of the
span with >= 2 / >= 3 / 4 of them running, the duration of a dispatch and the period of a chain (start to start on
one queue).  usage: trace_overlap.py kernel_trace.csv [kernel_substring] [n]"""
import csv
import sys

import numpy as np

path = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_step_roles"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
rows = [r for r in csv.DictReader(open(path)) if kern in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
st = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
en = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
q = np.array([int(r.get("Queue_Id", 0)) for r in rows])
span = en.max() - st.min()
ev = sorted([(t, 1) for t in st] + [(t, -1) for t in en])
cur, last, hist = 0, ev[0][0], {}
for t, dlt in ev:
    hist[cur] = hist.get(cur, 0) + (t - last)
    cur += dlt
    last = t
print("kernel %s: %d dispatches on %d queues over %.1f us" % (kern, len(rows), len(set(q)), span / 1e3))
print("dispatch duration (us): median %.1f  mean %.1f  p95 %.1f" % (np.median(en - st) / 1e3, (en - st).mean() / 1e3, np.percentile(en - st, 95) / 1e3))
print("average number in flight: %.2f" % ((en - st).sum() / span))
for k in sorted(hist):
    print("  %d in flight: %5.1f %% of the span" % (k, 100.0 * hist[k] / span))
for qq in sorted(set(q)):
    s = np.sort(st[q == qq])
    if len(s) > 2:
        print("  queue %d: %d dispatches, period median %.1f us" % (qq, len(s), np.median(np.diff(s)) / 1e3))
first = st.min()
print("first dispatches (us from the first start):")
for i in range(min(12, len(rows))):
    print("  queue %-3d start %8.1f  end %8.1f  dur %6.1f" % (q[i], (st[i] - first) / 1e3, (en[i] - first) / 1e3, (en[i] - st[i]) / 1e3))
