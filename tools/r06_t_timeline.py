# round 6: upload cadence, walk and download times of the movi_pml_*_host calls traced by tools/r06_t.sh (reads gpurun_out/r06_t/trace/*.csv)
import csv, glob
O="gpurun_out/r06_t"
ev=[]
for f in glob.glob(O+"/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_",""), ""))
for f in glob.glob(O+"/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"][12:60]))
ev.sort()
big=[e for e in ev if e[2]=="HOST_TO_DEVICE" and e[1]-e[0]>120_000]
# split into calls by gaps > 0.5 ms between consecutive big copies
calls=[[big[0]]]
for a,b in zip(big,big[1:]):
    if b[0]-a[1] > 500_000: calls.append([])
    calls[-1].append(b)
print([len(c) for c in calls])
for c in calls[-5:]:
    t0=c[0][0]
    starts=[(e[0]-t0)/1e6 for e in c]; durs=[(e[1]-e[0])/1e6 for e in c]
    gaps=[(b[0]-a[1])/1e6 for a,b in zip(c,c[1:])]
    ks=[e for e in ev if e[2]=="K" and "flatp" in e[3] and e[0]>=t0 and e[0]<=c[-1][1]+1_000_000]
    lastk=max(k[1] for k in ks) if ks else 0
    lastd=max(e[1] for e in ev if e[2]=="DEVICE_TO_HOST" and e[0]>=t0 and e[0]<=c[-1][1]+1_500_000)
    print("copies %d: mean dur %.3f, mean gap %.3f, uploads end %.3f, last walk ends %.3f, last D2H ends %.3f; kernel mean %.3f" % (len(c), sum(durs)/len(durs), sum(gaps)/max(1,len(gaps)), (c[-1][1]-t0)/1e6, (lastk-t0)/1e6, (lastd-t0)/1e6, sum((k[1]-k[0]) for k in ks)/1e6/max(1,len(ks))))
    print("  gaps:", " ".join("%.3f"%g for g in gaps))
