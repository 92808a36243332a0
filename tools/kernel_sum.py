import sqlite3, sys, collections, re
c=sqlite3.connect(sys.argv[1])
rows=c.execute("select name,start,end from kernels order by start").fetchall()
tot=collections.defaultdict(lambda:[0,0])
for n,s,e in rows:
    k=re.sub(r"\(anonymous namespace\)::","",n); k=re.sub(r"^void ","",k).split("(")[0].split("<")[0][-44:]
    tot[k][0]+=e-s; tot[k][1]+=1
span=rows[-1][2]-rows[0][1]
print("span ms",span/1e6,"busy",sum(v[0] for v in tot.values())/span)
for k,v in sorted(tot.items(),key=lambda kv:-kv[1][0])[:16]:
    print(f"{k:44s} {v[0]/1e6:9.3f} ms {v[1]:6d} launches avg {v[0]/v[1]/1e3:9.1f} us")
