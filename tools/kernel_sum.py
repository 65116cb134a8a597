"""Per-kernel digest of a `rocprofv3 --kernel-trace --stats --output-format csv -d <dir>` run: launches, average and total microseconds per
forward.  usage: kernel_sum.py <dir> <number of forwards the run made>"""
import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
n=float(sys.argv[2])
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernels per forward {sum(int(r['Calls']) for r in rows)/n:.1f}, busy us per forward {tot/n/1e3:.0f}")
for r in rows[:22]: print(f"{r['Name'][:100]:100s} {int(r['Calls'])/n:5.1f} {float(r['AverageNs'])/1e3:7.1f} {float(r['TotalDurationNs'])/n/1e3:7.1f}")
