import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    k=(r["Kernel_Name"][:48], r.get("Grid_Size_X",""), r.get("Grid_Size_Y",""))
    agg[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:40]:
    print(k, len(v), round(sum(v)/len(v)/1000,2))
