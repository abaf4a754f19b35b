"""Design study for the map's bulk kNN launch (round 4, the review's item 1): what would LANE-HOMOGENEOUS waves buy?

CPU only (numpy); no product code.  For a contiguous stretch of the cell-sorted c-main map every query's cost drivers are replayed the
way knn_point_sp walks them -- quads of candidates scanned (rows nearest first, a row out of reach skipped), keys that enter the chain
after the 24-candidate sorting-network fill -- and waves of 64 queries are formed (a) as the kernel forms them, 64 consecutive queries in
cell order, and (b) after sorting the queries of a POOL (256 = a workgroup, 512, 1024, 4096 = one XCD run, all) by their quad count or
by their appended keys.  A wave pays for its slowest lane: trips = max quads, insert rounds ~ max appended keys; cost = 70 x (trips - 6)
+ 27 x rounds VALU instructions (profiles/r03_knn_isa_mix.json), of ~5500 per wave.

    python scripts/sim_wave_grouping.py [first query] [queries]      (needs /tmp/map1m.npy: the c-main map, scripts/sim_candidates.py)

Result (profiles/r04_knn_grouping_sim.json): consecutive waves 32.3 trips / 27.9 rounds; a workgroup-sized pool sorted by quads 29.0 /
27.6 (-240 instructions per wave, 4 %) -- and the re-ordering itself needs every query's row ranges BEFORE the waves are formed (18
start[] loads + the sums: ~130 instructions per wave), so the net is ~2 %; 1024-query pools 27.4 / 25.1 (-415, 7.5 %; one workgroup per
CU, a wave per SIMD less); the ideal -- every lane alone -- is -12 %.  The insert rounds do not follow the block's population: sorting by
one driver un-sorts the other.  Not built: the review's 'third of the kernel' is a tenth, and half of that is spent forming the waves.
"""
import sys, numpy as np
sys.path.insert(0,'/root/repo/scripts')
from sim_candidates import build, gaps, ring_order
K=20; L=22
P=np.load('/tmp/map1m.npy'); g=build(P,1.0)
Ps,start,dim=g["P"],g["start"],g["dim"]
rows=ring_order(1)
def query(i):
    q=Ps[i].astype(np.float64); c=g["c"][i]; gap=gaps(g,q,c)
    ds=[];bounds=[]
    tau=np.inf; chain=None
    seen=0; quads=0; app=0; first=[]
    filled=False
    for (dy,dz) in rows:
        y,z=c[1]+dy,c[2]+dz
        if not(0<=y<dim[1] and 0<=z<dim[2]): continue
        base=(z*dim[1]+y)*dim[0]
        a,b=start[base+c[0]-1],start[base+c[0]+2]
        if b<=a: continue
        bound=gap(1,dy)**2+gap(2,dz)**2
        if filled and bound>=tau: continue
        d=((Ps[a:b].astype(np.float64)-q)**2).sum(1)
        nq=(len(d)+3)//4
        for k in range(nq):
            dd=d[4*k:4*k+4]
            quads+=1
            if not filled:
                first.extend(dd.tolist())
                if quads==6:
                    chain=np.sort(np.array(first+[np.inf]*L))[:L]; tau=chain[-1]; filled=True
            else:
                m=dd<tau
                if m.any():
                    app+=int(m.sum())
                    chain=np.sort(np.concatenate([chain,dd[m]]))[:L]; tau=chain[-1]
    if not filled:
        pass
    return quads,app
lo=int(sys.argv[1]) if len(sys.argv)>1 else 300000
n=int(sys.argv[2]) if len(sys.argv)>2 else 8192
res=np.array([query(i) for i in range(lo,lo+n)])
np.save('/tmp/sim/qa_%d_%d.npy'%(lo,n),res)
t,a=res[:,0],res[:,1]
print("mean quads %.1f appended %.1f"%(t.mean(),a.mean()))
def cost(order):
    tt=t[order].reshape(-1,64); aa=a[order].reshape(-1,64)
    trips=tt.max(1); rounds=aa.max(1)
    return trips.mean(), rounds.mean(), (70*np.maximum(trips-6,0)+27*rounds).mean()
base=np.arange(n)
print("consecutive      trips %.1f rounds %.1f cost %.0f"%cost(base))
for pool in (256,512,1024,4096,n):
    o=base.reshape(-1,pool)
    key=(t*64+np.minimum(a,63)).reshape(-1,pool)
    srt=np.argsort(key,axis=1,kind='stable')
    order=(o[np.arange(o.shape[0])[:,None],srt]).reshape(-1)
    print("pool %5d sorted trips %.1f rounds %.1f cost %.0f"%((pool,)+cost(order)))
    key2=(a*64+np.minimum(t,63)).reshape(-1,pool)
    srt=np.argsort(key2,axis=1,kind='stable')
    order=(o[np.arange(o.shape[0])[:,None],srt]).reshape(-1)
    print("pool %5d by app  trips %.1f rounds %.1f cost %.0f"%((pool,)+cost(order)))
print("ideal (mean lane) cost %.0f"%((70*np.maximum(t-6,0)+27*a).mean()))
