import sys; sys.path.insert(0,'/root/repo')
import ntedit_amd
p=ntedit_amd.Polisher(0)
for lg in (20,21,22,23,24,25,26,27,28,29,30,32):
    pps,ms=p.gather_bench(1<<lg, 8_000_000_000)
    print("buf 2^%d B = %8.1f MB : %.1f G probes/s  (%.2f ms)"%(lg,(1<<lg)/1e6,pps/1e9,ms), flush=True)
