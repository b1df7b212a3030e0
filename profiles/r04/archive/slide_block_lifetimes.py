import numpy as np, csv, sys
f=open(sys.argv[1]); hd=f.readline(); r=list(csv.reader(f)); d=np.array([[int(v) for v in x] for x in r[1:]],dtype=np.uint64)
w0=d[d[:,1]==0]
t=w0[:,5:].astype(float); used=t>0
t0=t[used].min()
first=np.array([t[i][used[i]].min() for i in range(len(t))]); last=np.array([t[i][used[i]].max() for i in range(len(t))])
life=(last-first)*0.01; nm=used.sum(axis=1)
print(hd.strip(), 'blocks',len(w0))
print('marks per block: ', np.unique(nm, return_counts=True))
print('start spread us: max',((first-t0)*0.01).max().round(1), ' end: min/med/max', ((last-t0)*0.01).min().round(1), np.median((last-t0)*0.01).round(1), ((last-t0)*0.01).max().round(1))
o=np.argsort(life)
print('life percentiles 0/10/50/90/100:', [round(float(np.percentile(life,p)),1) for p in (0,10,50,90,100)])
print('slowest blocks: id, life, marks:', [(int(w0[i,0]), round(float(life[i]),1), int(nm[i])) for i in o[-12:]])
# typical lean block: per-tile mark deltas
i=o[len(o)//2]
m=(t[i][used[i]]-first[i])*0.01
print('median block',int(w0[i,0]),'marks:', m.round(1))
D=6
per=D+1
dm=np.diff(m)
print('per-tile phase durations (level1, step1..5):')
for k in range(0,len(dm)-per+1,per): print('   ', dm[k:k+per].round(1), ' tile total', dm[k:k+per].sum().round(1))
# residency: which CU (xcc, se, sh, cu) each block ran on; lifetimes by number of blocks on the CU
hw=w0[:,3]; xcc=w0[:,2]&0xf
cuid=(hw>>8)&0xf; sh=(hw>>12)&1; se=(hw>>13)&0x7
key=((xcc*8+se)*2+sh)*16+cuid
import collections
cnt=collections.Counter(key.tolist())
print('CUs used',len(cnt),'blocks/CU histogram',collections.Counter(cnt.values()))
alone=np.array([cnt[k]==1 for k in key.tolist()])
print('life alone: n',alone.sum(),' median',np.median(life[alone]).round(1),' paired: median',np.median(life[~alone]).round(1),' p90',np.percentile(life[~alone],90).round(1))
gen=nm<49
print('general blocks life:', np.sort(life[gen]).round(0))
# paired with general?
kg=set(key[gen].tolist())
pg=np.array([(k in kg) for k in key.tolist()]) & ~gen
print('lean blocks sharing a CU with a general block: n',pg.sum(),'median life',np.median(life[pg]).round(1))
slow=life>200
print('slow lean blocks: n',slow.sum(),' of which alone',(slow&alone).sum(),' xcc hist',collections.Counter(xcc[slow].tolist()))
print('all blocks xcc hist',collections.Counter(xcc.tolist()))
for i in o[-3:]:
    m=(t[i][used[i]]-first[i])*0.01; dm=np.diff(m)
    print('slow block',int(w0[i,0]),'xcc',int(xcc[i]),'tile totals',[round(float(dm[k:k+7].sum()),1) for k in range(0,len(dm)-6,7)], 'step5s',[round(float(dm[k+5]),1) for k in range(0,len(dm)-6,7)])
