"""Per-step kernel time of the occupancy-grid training step from a rocprofv3 --kernel-trace CSV of tools/prof_occupancy.py: the last
STEPS steps only (steady state: sample budget set, no warm-up launches), a step = the launches between two k_march_count_wave.
Library kernels (this repo's) against torch's own; sum of durations per step.  usage: occ_step_summary.py <kernel_trace.csv> [steps]"""
import csv, re, sys
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_march_count" in r["Kernel_Name"]]            # the step's first library kernel
lo, hi = marks[-STEPS - 1], marks[-1]                     # STEPS whole steps (the last, incomplete one is dropped)
acc = {}
for r in rows[lo:hi]:
    name = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc.setdefault(name, [0.0, 0])
    a[0] += d; a[1] += 1
span = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3 / STEPS
lib, other = [], []
for name, (us, calls) in acc.items():
    short = re.sub(r"\(.*", "", name).replace("void ", "")
    is_lib = re.search(r"\bk_[a-z]|_Z\d+k_", name) is not None
    (lib if is_lib else other).append((us / STEPS, calls / STEPS, short[:72]))
print(f"steady state, {STEPS} steps: {span:.1f} us between step starts")
for title, lst in (("library kernels", lib), ("torch / runtime kernels", other)):
    lst.sort(reverse=True)
    print(f"--- {title}: {sum(x[0] for x in lst):.1f} us/step, {sum(x[1] for x in lst):.1f} launches/step")
    for us, calls, n in lst[:24]:
        print(f"  {us:8.1f} us  x{calls:5.2f}  {n}")
