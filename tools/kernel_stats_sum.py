"""Sum of the decode kernels' average durations per step from a rocprofv3 --kernel-trace --stats file, beside the step time of the
same run (VERDICT r3 #4d): python3 tools/kernel_stats_sum.py profiles/r04_fused_q4_kernel_stats.csv [bench json of the same run]"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
dec = [r for r in rows if "k_dec_" in r["Name"]]
steps = sum(int(r["Calls"]) for r in dec if "k_dec_argmax" in r["Name"])
total = 0.0
print(f"{steps} decode steps in the trace")
for r in sorted(dec, key=lambda r: -float(r["TotalDurationNs"])):
    per_step = int(r["Calls"]) / steps
    us = float(r["AverageNs"]) / 1e3
    total += per_step * us
    print(f"  {r['Name'].split('(')[0][:60]:60s} {per_step:6.2f} launches per step x {us:7.2f} us = {per_step * us:7.1f} us")
print(f"sum of kernel time per step: {total:.1f} us")
if len(sys.argv) > 2:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"ms_per_step of the same run (graph replay, under the profiler): {d['ms_per_step']} ms = {d['ms_per_step'] * 1e3:.1f} us; "
          f"ratio sum / step = {total / (d['ms_per_step'] * 1e3):.3f}")
