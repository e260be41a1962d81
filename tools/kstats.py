import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time ms %.2f  calls %d" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(r["Name"][:72].ljust(72), r["Calls"].rjust(6), "%8.2f ms %8.1f us" % (float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
