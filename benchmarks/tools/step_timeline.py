"""One step's kernel timeline out of a rocprofv3 --kernel-trace CSV: the launches between two consecutive extend
launches, start / end / duration in microseconds relative to the first, with the queue each ran on.
usage: python benchmarks/tools/step_timeline.py <..._kernel_trace.csv> [step index from the end = 3]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ext = [i for i, r in enumerate(rows) if "k_loc_extend" in r["Kernel_Name"]]
a, b = ext[-back - 1], ext[-back]
t0 = int(rows[a]["Start_Timestamp"])
print("start_us end_us dur_us queue kernel")
for r in rows[a:b + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f} q{r.get('Queue_Id', '?')} {r['Kernel_Name'][:72]}")
