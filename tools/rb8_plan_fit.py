#!/usr/bin/env python3
"""Fit of rb8_plan's cost model (ao_amd/csrc/rb8_kernels.hip) to the (slab rows, tile columns, K parts) grid measured by tools/midm_sweep.py:

    python tools/rb8_plan_fit.py profiles/rb8_grid_r06_fp8.jsonl profiles/rb8_grid_r06_int8.jsonl

Least squares on log(model / measured) over every single-round form of every cell the weight-streaming kernel takes, then a hill climb within
+- 30 % of that fit for the least mean regret of the model's pick (measured time of the pick / the cell's best form).  Prints the constants
(per slab height: fixed, step by 32 / 64 / 128 columns, meeting <= 4 parts, meeting > 4 parts, gather; then the stream rate in TB/s) and the
cells where the pick is furthest from the best form.  Host-only: reads the committed measurements.

The grid itself (on the box; 36 forms = slab rows {128, 64} x tile columns {32, 64, 128} x K parts {1, 2, 3, 4, 6, 8}, plus the default):

    FORMS=default; for bm in 128 64; do for bn in 32 64 128; do for s in 1 2 3 4 6 8; do FORMS=$FORMS,bm$bm+bn$bn+s$s; done; done; done
    python tools/midm_sweep.py --ms 80,96,128,160,192,256,320,384,512 --kinds fp8 --no-core --forms $FORMS > profiles/rb8_grid_r06_fp8.jsonl
    python tools/midm_sweep.py --ms 128,256,512 --kinds int8 --no-core --forms $FORMS > profiles/rb8_grid_r06_int8.jsonl
    (one 64-row slab, M = 40 / 48 / 64, the bm64 forms only: profiles/rb8_grid_r06_small.jsonl)
and the check of the refitted plan next to 128-row slabs forced, 64-row slabs forced and hipBLASLt (profiles/midm_rb8_plan_r06.jsonl):
    python tools/midm_sweep.py --ms 80,96,128,160,192,256,384,512,768,1024 --kinds fp8,int8 --check --forms default,bm128,bm64
"""
import json,collections,math,re,sys
import numpy as np
from scipy.optimize import least_squares
rows=[]
for f in sys.argv[1:]:
    rows+= [json.loads(l) for l in open(f)]
pts=[]; cells=collections.defaultdict(dict)
for r in rows:
    if 'us' not in r: continue
    key=(r['kind'],r['shape'],r['M'],r['N'],r['K'])
    if r['form']=='default':
        cells[key]['default']=(r['us'],r.get('kernel')); continue
    m=re.fullmatch(r'bm(\d+)\+bn(\d+)\+s(\d)',r['form'])
    if not m: continue
    bm,bn,S=map(int,m.groups())
    ks=r['K']>>7
    Seff=max(1,min(S,16,ks))
    cells[key][(bm,bn,S)]=r['us']
# drop cells where the rb kernel is not what runs (all forms equal): detect by default's kernel name
def feats(M,N,K,bm,bn,S):
    slabs=(M+bm-1)//bm; ks=K>>7
    tiles=((N+bn-1)//bn)*slabs
    wgs=tiles*S; rounds=(wgs+255)//256; steps=(ks+S-1)//S
    return slabs,ks,tiles,wgs,rounds,steps
BN={32:0,64:1,128:2}
def model(P,M,N,K,bm,bn,S):
    o=0 if bm==128 else 8
    F,c32,c64,c128,m1,m2,g,_=P[o:o+8]
    BW=P[16]
    slabs,ks,tiles,wgs,rounds,steps=feats(M,N,K,bm,bn,S)
    c=(c32,c64,c128)[BN[bn]]
    loop=max(steps*c*rounds, N*K/(BW*1e6))
    meet=0.0 if S==1 else ((m1 if S<=4 else m2)+g*(S-1)*bn/128*bm/128)
    return F*rounds+loop+meet
data=[]
for key,v in cells.items():
    kind,shape,M,N,K=key
    if 'default' not in v or v['default'][1]!='rb8_kernel': continue
    for f,us in v.items():
        if f=='default': continue
        bm,bn,S=f
        slabs,ks,tiles,wgs,rounds,steps=feats(M,N,K,bm,bn,S)
        if S>ks//4 and S>1: continue
        if S>1 and wgs>256: continue
        data.append((M,N,K,bm,bn,S,us))
print(len(data),'points', len(cells),'cells')
P0=np.array([5.5,.333,.35,.52,3.17,4.53,.4,0, 5.5,.28,.29,.43,1.6,2.3,.4,0, 5.85])
def resid(P): return [math.log(model(P,*d[:6])/d[6]) for d in data]
res=least_squares(resid,P0,bounds=(np.array([2,.1,.1,.1,0,0,0,-1]*2+[3.0]),np.array([12,1,1,1,10,10,3,1]*2+[8.0])))
P=res.x
print('fit',np.round(P,3),'rms log err',np.sqrt(np.mean(np.square(res.fun))))
def pick(P,M,N,K,bms=(128,64)):
    best=None
    for bm in bms:
        if bm==128 and M<=64: continue
        for bn in (128,64,32):
            for S in (1,2,3,4,6,8):
                slabs,ks,tiles,wgs,rounds,steps=feats(M,N,K,bm,bn,S)
                if S>1 and (S>max(1,ks//4) or wgs>256): continue
                t=model(P,M,N,K,bm,bn,S)
                if best is None or t<best[0]: best=(t,bm,bn,S)
    return best
tot=collections.Counter()
for key,v in sorted(cells.items(),key=lambda kv:(kv[0][0],kv[0][1],kv[0][2])):
    kind,shape,M,N,K=key
    if 'default' not in v or v['default'][1]!='rb8_kernel': continue
    meas={f:us for f,us in v.items() if f!='default'}
    bestf=min(meas,key=meas.get)
    p=pick(P,M,N,K); p128=pick(P,M,N,K,(128,))
    tp=meas.get(p[1:]); t128=meas.get(p128[1:]) if p128 else None
    print(kind,shape,M,'default',v['default'][0],'best',bestf,meas[bestf],'pick',p[1:],tp,'model',round(p[0],1),'regret',round(tp/meas[bestf],3) if tp else None,'| bm128 pick',p128[1:] if p128 else None,t128)
    if tp: tot['pick']+=tp; tot['best']+=meas[bestf]; tot['default']+=v['default'][0]; tot['n']+=1
    if t128: tot['p128']+=t128
print(dict(tot))

# ---- regret-minimising refinement over all cells (fp8 + int8) ----
import random
random.seed(1)
allcells=[]
for key,v in cells.items():
    kind,shape,M,N,K=key
    if 'default' not in v or v['default'][1]!='rb8_kernel': continue
    meas={f:us for f,us in v.items() if f!='default'}
    allcells.append((M,N,K,meas,min(meas.values()),v['default'][0]))
def total(P):
    t=0
    for M,N,K,meas,b,d in allcells:
        p=pick(P,M,N,K)
        t+=meas.get(p[1:], 2*b)/b
    return t/len(allcells)
cur=P.copy(); curv=total(cur); print('start regret',curv)
for it in range(1500):
    cand=cur.copy()
    for j in random.sample(range(17),3):
        if j in (7,15): continue
        cand[j]*=math.exp(random.gauss(0,0.08))
    v=total(cand)
    if v<=curv: cur,curv=cand,v
print('refined regret',curv,np.round(cur,3))
dsum=sum(d/b for M,N,K,meas,b,d in allcells)/len(allcells); print('default regret',dsum)
worst=sorted(((meas.get(pick(cur,M,N,K)[1:],9e9)/b,M,N,K,pick(cur,M,N,K)[1:]) for M,N,K,meas,b,d in allcells),reverse=True)[:8]
print(worst)

print('---- bounded refinement (+-30% of the LS fit) ----')
random.seed(2)
cur=P.copy(); curv=total(cur)
for it in range(1500):
    cand=cur.copy()
    for j in random.sample(range(17),3):
        if j in (7,15): continue
        cand[j]=min(max(cand[j]*math.exp(random.gauss(0,0.05)),0.7*P[j]),1.3*P[j])
    v=total(cand)
    if v<=curv: cur,curv=cand,v
print('bounded refined regret',curv,repr(np.round(cur,3)))
worst=sorted(((meas.get(pick(cur,M,N,K)[1:],9e9)/b,M,N,K,pick(cur,M,N,K)[1:]) for M,N,K,meas,b,d in allcells),reverse=True)[:8]
print(worst)
# how often bm64 / by M
cnt=collections.Counter((M,pick(cur,M,N,K)[1]) for M,N,K,meas,b,d in allcells); print(sorted(cnt.items()))
