"""tools/dispatch_rounds.py (CPU): how many ROUNDS of workgroups a level-0 launch of k_spconv_w is, per unit shape -- the bench
pair's stride-1 map as built and occupancy-sorted (oracle/imf_cpu_twins), units dispatched greedily onto (CUs x resident
workgroups) slots per XCD chunk, in launch order and heaviest-first.  Shows that part of what a unit shape gains or loses is
quantisation: 2 156 48-row units on 768 slots are 2.8 rounds (efficiency 0.93), 1 617 whole tiles on 768 are 2.1 -> 3 rounds
(0.70), 3 234 half tiles on 768 / 1 024 are 4.2 / 3.2 -> 5 / 4 rounds (0.84 / 0.79).  LAB_NOTES 4g-8."""
import sys, numpy as np, heapq
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import bench, imf_cpu_twins as T
pts2, imgs2 = bench.load_pair(1.7)
lvs=[]
for p in pts2:
    lv,err=T.voxelize(np.asarray(p,dtype=np.float64),0.025); lvs.append(lv)
print([l.n for l in lvs])
# batch: concatenate maps per fragment (tiles don't mix fragments? they do in the batch (one coordinate set with batch index)); approximate: per fragment separately, concatenated
def masks_of(lv, sort):
    rows,nbr,mask=T.rulebook_conv(lv,lv,1,3)
    if sort:
        rows,nbr,mask=T.rulebook_sort_by_occupancy(nbr,lv.n)
    return mask[:,0].astype(np.uint32), len(rows)
def popc(x): return np.array([bin(int(v)).count('1') for v in x])
def sim(work, slots, order):
    # greedy in-order dispatch of work items onto `slots` identical servers; returns makespan
    h=[0.0]*slots; heapq.heapify(h)
    for w in work[order]:
        t=heapq.heappop(h); heapq.heappush(h,t+w)
    return max(h)
for sort in (False, True):
    m=np.concatenate([masks_of(l,sort)[0] for l in lvs]); nt=len(m)
    for name,UR,slots_per_cu in (("half",32,3),("half x4",32,4),("u48",48,3),("whole x2",64,2),("whole x3",64,3)):
        n_units=(nt*64+UR-1)//UR
        u=np.arange(n_units); t0=(u*UR)//64; t1=np.minimum((u*UR+UR-1)//64, nt-1)
        mm=m[t0]|m[t1]
        nk=popc(mm)
        work=(4.0+2*nk)*(UR/64.0)**0.7      # sub-stage count x cost per sub-stage (sub-linear in rows) + skeleton
        work=4.0*(UR/64)**0.3+2*nk*(UR/64.0)**0.7
        # per XCD contiguous chunks
        res={}
        for mode in ("inorder","lpt"):
            mk=0
            chunk=(n_units+7)//8
            for x in range(8):
                idx=np.arange(x*chunk, min((x+1)*chunk,n_units))
                if mode=="lpt": idx=idx[np.argsort(-work[idx],kind='stable')]
                mk=max(mk, sim(work, 32*slots_per_cu, idx))
            res[mode]=mk
        ideal=work.sum()/(256*slots_per_cu)
        print(f"sorted={sort} {name:9s} units {n_units:5d} mean nk {nk.mean():.1f} ideal {ideal:.1f} inorder {res['inorder']:.1f} ({res['inorder']/ideal:.3f}) lpt {res['lpt']:.1f} ({res['lpt']/ideal:.3f})")
