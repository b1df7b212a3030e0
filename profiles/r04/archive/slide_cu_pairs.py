import numpy as np, csv, sys, collections
f=open(sys.argv[1]); hd=f.readline(); r=list(csv.reader(f)); d=np.array([[int(v) for v in x] for x in r[1:]],dtype=np.uint64)
w0=d[d[:,1]==0]
t=w0[:,5:].astype(float); used=t>0
t0=t[used].min()
first=np.array([t[i][used[i]].min() for i in range(len(t))]); last=np.array([t[i][used[i]].max() for i in range(len(t))])
life=(last-first)*0.01; start=(first-t0)*0.01
hw=w0[:,3]; xcc=w0[:,2]&0xf
key=((xcc*8+((hw>>13)&7))*2+((hw>>12)&1))*16+((hw>>8)&0xf)
simd=(hw>>4)&3; wid=hw&0xf
groups=collections.defaultdict(list)
for i,k in enumerate(key.tolist()): groups[k].append(i)
both_slow=0; one_slow=0; none=0; rows=[]
for k,idx in groups.items():
    if len(idx)!=2: continue
    a,b=idx
    s=(life[a]>190)+(life[b]>190)
    if s==2: both_slow+=1
    elif s==1: one_slow+=1
    else: none+=1
    rows.append((round(float(start[a]),1),round(float(start[b]),1),round(float(life[a]),0),round(float(life[b]),0),int(w0[a,0]),int(w0[b,0])))
print('pairs: both slow',both_slow,' one slow',one_slow,' none',none)
print('sample pairs (startA,startB,lifeA,lifeB,idA,idB):')
for r_ in rows[:24]: print('  ',r_)
