"""Device-busy time (union of kernel intervals over both streams) per alpha-learning step from a rocprofv3 kernel trace of
tests/tools/train_bench.py -- steps are delimited by the one distill_rows_kernel launch each has.
Usage: rocprofv3 --kernel-trace --output-format csv -d /tmp/tb -- python3 tests/tools/train_bench.py; python tools/train_busy.py <kernel_trace.csv>"""
import csv, sys
ev=[]
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
ev.sort()
marks=[s for s,e,n in ev if "distill_rows_kernel" in n]
a,b=marks[-11],marks[-1]
seg=[(s,e) for s,e,n in ev if s>=a and s<b]
busy=0; cur_s,cur_e=seg[0]
for s,e in seg[1:]:
    if s>cur_e: busy+=cur_e-cur_s; cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print(f"steps 10: wall {(b-a)/1e7:.3f} ms/step, device busy (union) {busy/1e7:.3f} ms/step, idle {(b-a-busy)/1e7:.3f} ms/step, launches/step {len(seg)/10:.0f}")
