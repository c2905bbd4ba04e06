"""Kernel-by-kernel timeline of ONE solve of the bench loop from a rocprofv3 kernel trace (…_kernel_trace.csv): duration of every
launch and the gap to its predecessor.  A solve starts at an lmBeginKernel launch (the restore to the snapshot rides in it).

    rocprofv3 --kernel-trace --output-format csv -d out -o prof -- python scripts/profile_target.py c1|large_loop
    python scripts/one_solve_timeline.py out/.../prof_kernel_trace.csv > profiles/rNN/<w>_one_solve_timeline.csv"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n).replace("dsopp_hip::", "")
    m = re.match(r"(void )?([^\(]+?)(\(|$)", n)
    return (m.group(2) if m else n).strip()


names = [short(r["Kernel_Name"]) for r in rows]
begins = [i for i, n in enumerate(names) if n.startswith("lmBeginKernel")]
a, b = begins[-3], begins[-2]  # a solve in the middle of the last batch
print("kernel,duration_us,gap_before_us")
prev = None
total = 0.0
for i in range(a, b):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    print('"%s",%.2f,%.2f' % (names[i], (e - s) / 1000, 0.0 if prev is None else (s - prev) / 1000))
    total += (e - s) / 1000
    prev = e
print('"(sum of kernel durations; start of this solve to start of the next: %.2f us)",%.2f,' %
      ((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1000, total))
