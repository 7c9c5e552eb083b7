"""tools/kernel_avg.py <stats dir> <substring>...: average duration (us) and call count of the kernels whose name contains a substring, from a
rocprofv3 --kernel-trace --stats output directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in sys.argv[2:]):
        print(f'{r["Name"][:64]:64s} calls {int(r["Calls"]):5d} avg_us {float(r["AverageNs"]) / 1e3:8.2f}')
