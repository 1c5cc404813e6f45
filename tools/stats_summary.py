import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 22]:
    print("%-82s %7s %9.1f ms %6.2f%% avg %8.1f us" % (r["Name"][:82], r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                                        float(r["Percentage"]), float(r["AverageNs"]) / 1e3))
