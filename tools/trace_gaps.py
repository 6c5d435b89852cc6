import csv,glob,statistics as st,sys
d0=sys.argv[1]
f=glob.glob(d0+"/**/*kernel_trace.csv",recursive=True)[0]
allrows=list(csv.DictReader(open(f)))
for name in ("k_oj_round","k_pcx_step","k_pcb_block"):
    rows=[r for r in allrows if name in r["Kernel_Name"]]
    if not rows: continue
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows]
    g=[int(rows[i+1]["Start_Timestamp"])-int(rows[i]["End_Timestamp"]) for i in range(len(rows)-1)]
    sd=sorted(d)
    print(name,"n",len(d),"dur median",st.median(d),"p10",sd[len(d)//10],"p90",sd[9*len(d)//10],"gap median",st.median(g),"gap mean",sum(g)/len(g), "sum dur ms", sum(d)/1e6, "span ms", (int(rows[-1]["End_Timestamp"])-int(rows[0]["Start_Timestamp"]))/1e6)
