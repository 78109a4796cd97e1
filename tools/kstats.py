"""Print the head of a rocprofv3 kernel_stats.csv found under a directory: python tools/kstats.py DIR [N]."""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>7s} avg {float(r["AverageNs"]) / 1e3:9.2f} us  {r["Percentage"]}%')
