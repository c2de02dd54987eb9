#!/usr/bin/env python
"""Print the top rows of a rocprofv3 *kernel_stats.csv (name, calls, average us, share).  usage: kstat_top.py file [n]"""
import csv
import sys


def main(path, n=14):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows[:n]:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"{name[:48]:48s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e3:8.1f} us  "
              f"{100 * float(r['TotalDurationNs']) / tot:5.1f} %")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 14)
