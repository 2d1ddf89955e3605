import numpy as np, sys
np.set_printoptions(linewidth=220, precision=5)
def images(d):
    r=d[8192:8192+16384].view(np.uint32).reshape(512,64)
    s=d[8192+16384:8192+16384+160*32].view(np.uint32).reshape(160,64)
    vec={}
    def add(name, lo, hi):
        v=(lo.astype(np.uint64)|(hi.astype(np.uint64)<<np.uint64(32)))
        vec[name]=v
    for i in range(0,512,2): add(('v%d'%i) if i<256 else ('a%d'%(i-256)), r[i], r[i+1])
    for k in range(0,159): add('s%d'%(4*k), s[k], s[k+1])
    return vec
def interesting(v):
    f=v.view(np.float64)
    if not np.all(np.isfinite(f)): return False
    if len(np.unique(v))<8: return False
    a=np.abs(f[f!=0])
    return a.size>0 and a.max()<1e12 and a.min()>1e-30
def key(v): return v.tobytes()
B=np.load(sys.argv[1]); D=np.load(sys.argv[2])
gB=images(B['a_dump']); bB=images(B['b_dump']); gD=images(D['a_dump']); bD=images(D['b_dump'])
sets={n:{key(v):nm for nm,v in im.items() if interesting(v)} for n,im in (('gB',gB),('bB',bB),('gD',gD),('bD',bD))}
persistent=[k for k in sets['gB'] if k in sets['gD'] and k in sets['bB']]
print("interesting vectors: goodB %d badB %d goodD %d badD %d ; persistent (goodB & goodD & badB): %d" % (len(sets['gB']),len(sets['bB']),len(sets['gD']),len(sets['bD']),len(persistent)))
missing=[k for k in persistent if k not in sets['bD']]
print("persistent vectors missing from bad build at the later point:", len(missing))
for k in missing:
    v=np.frombuffer(k,dtype=np.uint64)
    print(" goodB %s goodD %s badB %s" % (sets['gB'][k], sets['gD'][k], sets['bB'][k]))
    # closest in bD
    best=None
    for nm,w in bD.items():
        m=int((w==v).sum())
        if best is None or m>best[1]: best=(nm,m,w)
    print("   closest at bad-later: %s with %d lanes equal; differing lanes: %s" % (best[0], best[1], np.where(best[2]!=v)[0]))
    print("   expected:", v.view(np.float64).reshape(4,16)[:, :6], "\n   found:", best[2].view(np.float64).reshape(4,16)[:, :6])
