"""Where a multi-lane decode step's wall time goes, from a rocprofv3 kernel trace (run on the GPU box):
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --no-cpu-baseline --streams 0 --wide-streams 256 --no-lanes \
        --prefill 0 --generate 0 --serve 0 --steps 16 --warmup 4 --fill prefill
    python3 tools/lane_timeline.py DIR > gpurun_out/lane_timeline.txt
Takes a window of four steps of decoder dispatches (k_dec_*), splits them by hardware queue (a lane's launch chain stays on one), and
reports per queue: kernel time, gaps between consecutive kernels (launch boundaries), and across queues: time with 0 / 1 / 2+
kernels in flight.  A profiled run: every launch is slower than in the graph replay, the SHARES are what this is for."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no kernel_trace.csv under " + d
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if "k_dec_" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"].split("(")[0][:48]))
    rows.sort()
    # the last 4 steps' worth: attention launches mark (block, lane); take the window behind the 8 x 44-th attention from the end
    att = [i for i, r in enumerate(rows) if "k_dec_attn_hm" in r[3]]
    # (the bench's per-family timing replays follow the timed steps in the trace: the window is steps 10 .. 13 of the run itself)
    assert len(att) > 44 * 16, "too few steps in the trace"
    win = rows[att[44 * 10]:att[44 * 14]]
    t0, t1 = win[0][0], max(r[1] for r in win)
    print(f"window: {len(win)} decoder dispatches, {(t1 - t0) / 1e3:.1f} us = {(t1 - t0) / 1e3 / 4:.1f} us per step (4 steps, profiled)")
    byq = defaultdict(list)
    for r in win:
        byq[r[2]].append(r)
    for q, rs in sorted(byq.items()):
        busy = sum(r[1] - r[0] for r in rs)
        gaps = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
        pos = [g for g in gaps if g > 0]
        print(f"queue {q}: {len(rs)} dispatches, kernel time {busy / 1e3:.1f} us, gaps {sum(pos) / 1e3:.1f} us over {len(pos)} boundaries "
              f"(median {sorted(pos)[len(pos) // 2] / 1e3 if pos else 0:.2f} us), overlapping-next {sum(1 for g in gaps if g <= 0)}")
        per = defaultdict(lambda: [0, 0])
        for r in rs:
            per[r[3]][0] += r[1] - r[0]; per[r[3]][1] += 1
        for k, (t, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:8]:
            print(f"      {k:50s} {n:5d} x {t / n / 1e3:7.2f} us")
    # concurrency histogram over the window
    ev = []
    for r in win:
        ev.append((r[0], 1)); ev.append((r[1], -1))
    ev.sort()
    depth, last, hist = 0, t0, defaultdict(int)
    for t, dlt in ev:
        hist[min(depth, 3)] += t - last
        last = t
        depth += dlt
    tot = float(t1 - t0)
    print("kernels in flight: " + ", ".join(f"{k}{'+' if k == 3 else ''}: {v / tot:.3f}" for k, v in sorted(hist.items())))


if __name__ == "__main__":
    main()
