import csv,glob,sys
f=glob.glob(f"gpurun_out/prof_{sys.argv[1]}/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
names=sys.argv[2].split(",")
sel=[r for r in rows if any(n in r["Kernel_Name"] for n in names)]
for r in sel[-int(sys.argv[3]):]:
    print(r["Kernel_Name"][:44], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
