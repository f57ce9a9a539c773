import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:int(sys.argv[2]) if len(sys.argv)>2 else 13]:
    print("%-70s calls %5s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
