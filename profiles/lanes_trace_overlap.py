"""How much of bench.py's timed region has two kernels resident?  Reads the kernel trace of the default command
(`rocprofv3 --kernel-trace ... -- python3 bench.py ...`: profiles/run_profiles.sh, stats_bench_lanes2/s_kernel_trace.csv),
finds the timed loop - the longest run of fp32 stem launches that alternate between the two lanes' queues less than 5 ms
apart (the warm-up before it ends on the lane the loop starts with) - and sweeps the launches of both queues from its first
stem to the end of its last forward.
Usage: python profiles/lanes_trace_overlap.py <s_kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
stems = [e for e in ev if "stem_dw_kernel<float>" in e[2]]
runs, cur = [], [stems[0]]
for prev, e in zip(stems, stems[1:]):
    if e[3] != prev[3] and e[0] - prev[0] < 5_000_000:
        cur.append(e)
    else:
        runs.append(cur)
        cur = [e]
runs.append(cur)
run = max(runs, key=len)
queues = {e[3] for e in run}
a = run[0][0]
# the forward that starts at the run's last stem ends with the MLP head's last launch on that queue
tail = [e for e in ev if e[0] >= run[-1][0] and e[3] in queues and "pw_kernel<1, false, 1, false>" in e[2]]
b = max(e[1] for e in tail[:2])
sel = [e for e in ev if e[0] >= a and e[1] <= b and e[3] in queues]
pts = sorted([(s, 1) for s, _, _, _ in sel] + [(e, -1) for _, e, _, _ in sel])
act, last, share = 0, a, [0, 0, 0]
for x, d in pts:
    share[min(act, 2)] += x - last
    last, act = x, act + d
span = b - a
print(f"timed loop: {len(run)} forwards on queues {sorted(queues)}, {len(sel)} launches, {span / 1e6:.2f} ms = "
      f"{span / 1e6 / len(run):.3f} ms per forward (under the profiler)")
print("share of the span with 0 / 1 / 2 kernels resident: " + " / ".join(f"{v / span:.3f}" for v in share))
print(f"sum of kernel durations / span = {sum(e - s for s, e, _, _ in sel) / span:.3f}")
