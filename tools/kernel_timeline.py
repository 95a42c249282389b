import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(list(rows[0].keys()))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sh::", ""), r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows), key=lambda e: e[0])
# last run: from the last rref_inverse_table to the last rref_tmp_to_rows
starts = [i for i, e in enumerate(ev) if e[2].startswith("rref_inverse_table")]
ends = [i for i, e in enumerate(ev) if e[2].startswith("rref_tmp_to_rows")]
a, b = starts[-1], ends[-1]
run = ev[a:b + 1]
t0, t1 = run[0][0], max(e[1] for e in run)
print("run wall %.1f us, kernels %d, sum of durations %.1f us" % ((t1 - t0) / 1e3, len(run), sum(e[1] - e[0] for e in run) / 1e3))
# busy time (union of intervals)
busy = 0; cur_s, cur_e = run[0][0], run[0][1]
for s_, e_, _, _ in run[1:]:
    if s_ > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
busy += cur_e - cur_s
print("union busy %.1f us, idle %.1f us" % (busy / 1e3, (t1 - t0 - busy) / 1e3))
per = defaultdict(lambda: [0, 0.0])
for s_, e_, k, q in run:
    per[(k, q)][0] += 1; per[(k, q)][1] += (e_ - s_) / 1e3
for (k, q), (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-34s q%-4s %5d %9.1f us %7.1f avg" % (k[:34], q, c, t, t / c))
