"""Ordered kernel dispatches of the last iterations of a rocprofv3 --kernel-trace run: name, duration, gap to the
previous dispatch.  usage: python tools/dbg/launch_seq.py <dir with *_kernel_trace.csv> [last N dispatches]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 400
fn = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-last:]
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("po::", "")
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%-44s %9.1f us  gap %7.1f us" % (name[:44], (e - s) / 1e3, gap))
    prev_end = e
