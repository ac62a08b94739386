import csv,glob,sys
d=sys.argv[1]; nsteps=int(sys.argv[2]) if len(sys.argv)>2 else 7
f=glob.glob(d+'/*/*_kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 16]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} ms/step={float(r['TotalDurationNs'])/1e6/nsteps:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={float(r['Percentage']):5.1f}")
print("total ms/step", tot/1e6/nsteps)
