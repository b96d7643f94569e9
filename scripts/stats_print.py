import csv, sys
f = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 3), "per step ms", round(tot / steps / 1e6, 3))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = r["Name"].split("(")[0][-64:]
    print("%-66s calls %5s avg_us %9.1f total_ms %8.2f %s%%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
