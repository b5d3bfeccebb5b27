"""Print the top kernels of a rocprofv3 --stats run: python tools/top_kernels.py gpurun_out/prof_x [n]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in list(csv.DictReader(open(f)))[:n]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(7), "%10.1f us avg" % (float(r["AverageNs"]) / 1e3), r["Percentage"].rjust(7), "%")
