"""Print the per-dispatch durations of one transformer layer from the newest rocprofv3 kernel trace under gpurun_out/pm."""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "pm", "*", "*_kernel_trace.csv")),
           key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
seq = [(r["Kernel_Name"].replace("void hg::", "").replace("_ZN2hg", "")[:34], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
i0 = len(seq) * 2 // 3
while "attention" not in seq[i0][0]:
    i0 += 1
print(" | ".join("%s %.0f" % (n[:18], t) for n, t in seq[i0:i0 + 7]))
