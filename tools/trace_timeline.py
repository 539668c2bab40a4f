"""Per-launch durations and inter-launch gaps of the demod kernel from a rocprofv3 kernel trace.
   python tools/trace_timeline.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import statistics
import sys

fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = [r for r in csv.DictReader(open(fs[0])) if "demod_" in r["Kernel_Name"] and "kernel_t" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
gap = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
print("launches", len(dur))
for lo in range(0, len(dur), max(1, len(dur) // 10)):
    seg = dur[lo: lo + max(1, len(dur) // 10)]
    g = gap[lo: lo + max(1, len(dur) // 10)]
    print(f"  launches {lo:4d}+: dur median {statistics.median(seg) / 1e3:7.2f} us  min {min(seg) / 1e3:7.2f}  "
          f"max {max(seg) / 1e3:7.2f}   gap median {statistics.median(g) / 1e3 if g else 0:6.2f} us")
print("overall: dur median %.2f us, mean %.2f; gap median %.2f us, mean %.2f" % (
    statistics.median(dur) / 1e3, statistics.mean(dur) / 1e3, statistics.median(gap) / 1e3,
    statistics.mean(gap) / 1e3))
