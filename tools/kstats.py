import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<int(sys.argv[2]) : print(r["Name"][:80], r["Calls"], int(float(r["AverageNs"]))/1e3, r["Percentage"])
