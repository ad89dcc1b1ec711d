import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:3]:
    print("   %-50s avg_us %8.1f"%(r["Name"][:50],float(r["AverageNs"])/1e3))
