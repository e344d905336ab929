#!/usr/bin/env python3
"""One steady-state step of `rocprofv3 --kernel-trace -- python3 bench.py ...` as a timeline (start / end / duration in us relative to
the conv launch, queue, kernel): what sits on the critical path between two conv launches.   python tools/step_timeline.py <rocprof output dir>"""
import csv, glob, sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
convs=[i for i,r in enumerate(rows) if 'conv3_wino63' in r['Kernel_Name']]
i0=convs[12]; i1=convs[13]
t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1+1]:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    n=r['Kernel_Name']; n=n[n.find('::')+2:][:40] if '::' in n else n[:40]
    print(f"{s:10.1f} {e:10.1f} {e-s:9.1f}  q{r.get('Queue_Id','?'):>3} {n}")
