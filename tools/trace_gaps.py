import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if "k_normal_fused" in r["Kernel_Name"] or "k_demod" in r["Kernel_Name"] or "k_tsc" in r["Kernel_Name"]]
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    prev=None
    out=[]
    for r in rows[-40:]:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        out.append((r["Kernel_Name"][:40],(e-s)/1e3,(s-prev)/1e3 if prev else 0))
        prev=e
    for o in out[-14:]: print("%-42s dur %8.1f us  gap-before %8.1f us"%o)
