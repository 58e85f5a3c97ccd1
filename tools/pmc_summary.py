import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float); cnt = collections.defaultdict(int); seen=set()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-60:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    d = r["Dispatch_Id"]
    if d not in seen:
        seen.add(d); dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k]+=1
for k in sorted(dur, key=lambda k:-dur[k])[:8]:
    print(f"{k}: n={cnt[k]} total_ms={dur[k]/1e6:.3f} vgpr={[r['VGPR_Count']+'/'+r['Accum_VGPR_Count'] for r in rows if r['Kernel_Name'].split('(')[0][-60:]==k][0]}")
    for c,v in sorted(agg[k].items()):
        print(f"    {c:32s} {v:.4g}  per_ns={v/dur[k]:.4g}")
