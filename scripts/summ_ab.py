import csv,sys,glob
for d in sys.argv[1:]:
    f=glob.glob(d+"/**/*kernel_stats.csv", recursive=True)
    if not f: continue
    out=[]
    for row in csv.DictReader(open(f[0])):
        n=row["Name"]
        for t in ("k_cs_march","k_forces_table","k_forces_parity","k_forces_gather","k_obst_paint"):
            if t in n: out.append("%s %.1f" % (t[2:], float(row["AverageNs"])/1e3))
    print(d.split("/")[-1], " | ".join(out))
