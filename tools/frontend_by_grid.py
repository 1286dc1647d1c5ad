import csv, glob, sys, collections, re
d=sys.argv[1]
agg=collections.defaultdict(list)
for f in glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if not any(k in n for k in ("gemm256","gemm_kernel","encoder_attention","layernorm","mel_","convert")): continue
        wg=int(r["Grid_Size_X"])//max(1,int(r["Workgroup_Size_X"]))
        agg[(re.sub(r"\(.*","",n)[:60],wg)].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    print(f"{k[0]:62s} wgs {k[1]:7d} calls {len(v):5d} mean {sum(v)/len(v)/1e3:10.1f} us total {sum(v)/1e6:9.1f} ms")
