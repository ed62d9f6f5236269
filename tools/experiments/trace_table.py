"""kernel_trace.csv of rocprofv3 -> per-kernel mean duration and mean gap to the previous kernel's end, over the last `reps` repetitions
of a fixed kernel sequence.   python trace_table.py <kernel_trace.csv> [first_kernel_substring]"""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "colsum_partial"
rows = [r for r in rows if "vsel::" in r["Kernel_Name"]]
short = lambda r: r["Kernel_Name"].replace("void ", "").replace("vsel::", "", 1)   # noqa: E731
idx = [i for i, r in enumerate(rows) if short(r).startswith(first)]
idx = idx[len(idx) // 2:]                      # second half: warm
seq_len = idx[1] - idx[0]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for i0 in idx[:-1]:
    for j in range(seq_len):
        r = rows[i0 + j]
        nm = f'{j:02d} ' + r["Kernel_Name"].split("(")[0].replace("void vsel::", "")[:60]
        dur[nm].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        gap[nm].append(int(r["Start_Timestamp"]) - int(rows[i0 + j - 1]["End_Timestamp"]))
tot = 0
for nm in sorted(dur):
    d_ = sum(dur[nm]) / len(dur[nm]) / 1e3; g_ = sum(gap[nm]) / len(gap[nm]) / 1e3
    tot += d_ + g_
    print(f"{nm:64s} dur {d_:7.2f} us   gap before {g_:6.2f} us")
print(f"sum of (dur + gap) per repetition: {tot:.1f} us over {len(idx) - 1} repetitions")
