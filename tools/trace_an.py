import sys, collections
rows=int(sys.argv[2]); 
d=[tuple(map(int,l.split())) for l in open(sys.argv[1])]
t0=min(x[1] for x in d if x[1]); tend=max(x[2] for x in d)
print("span us", (tend-t0)/100.0)
lv=collections.defaultdict(list)
for u,a,b,hw in d: lv[u//rows].append((a-t0,b-t0,hw))
for w in sorted(lv):
    v=lv[w]; dur=[(b-a)/100.0 for a,b,_ in v]
    print("level",w,"n",len(v),"start[min,max]us",min(a for a,_,_ in v)/100.0,max(a for a,_,_ in v)/100.0,"end max",max(b for _,b,_ in v)/100.0,"dur mean %.1f min %.1f max %.1f"%(sum(dur)/len(dur),min(dur),max(dur)))
