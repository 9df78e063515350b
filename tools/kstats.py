import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void dlaf_mi355x::", "").replace("(anonymous namespace)::", "")
    print(n[:64].ljust(64), r["Calls"].rjust(5), "%9.2f ms %9.1f us" % (float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3), r["Percentage"])
