import numpy as np, re, sys
np.set_printoptions(linewidth=200, precision=6)
Z=np.load('gpurun_out/fd6/dump_M.npz')
def state(d):
    r=d[8192:8192+16384].view(np.uint32).reshape(512,64).copy()
    return r[:256].copy(), r[256:].copy(), d[:5120].copy().view(np.uint8).copy()
V,A,LDS=state(Z['a_dump']); V1,A1,LDS1=state(Z['b_dump'])
lines=open('build/frag/st/k_g.s').read().split('\n')
start=32814-1   # first v_mfma (1-based line 32814)
endl=33267-1    # 61st v_mfma
def reg(tok):
    tok=tok.strip()
    m=re.match(r'^([av])\[(\d+):(\d+)\]$',tok)
    if m: return m.group(1),int(m.group(2)),int(m.group(3))-int(m.group(2))+1
    m=re.match(r'^([av])(\d+)$',tok)
    if m: return m.group(1),int(m.group(2)),1
    return None
def bank(f): return V if f=='v' else A
def rd64(f,i): b=bank(f); return (b[i].astype(np.uint64)|(b[i+1].astype(np.uint64)<<np.uint64(32))).view(np.float64)
def wr64(f,i,x): b=bank(f); u=x.view(np.uint64); b[i]=(u&np.uint64(0xffffffff)).astype(np.uint32); b[i+1]=(u>>np.uint64(32)).astype(np.uint32)
lane=np.arange(64)
trace=[]
nm=0
for ln in range(start,endl):
    s=lines[ln].split(';')[0].strip()
    if not s or s.startswith('.'): continue
    op,_,rest=s.partition(' ')
    ops=[o.strip() for o in rest.split(',')] if rest else []
    if op=='v_mfma_f64_16x16x4_f64':
        nm+=1
        d=reg(ops[0]); a=reg(ops[1]); b=reg(ops[2]); c=reg(ops[3])
        Am=np.zeros((16,4)); Bm=np.zeros((4,16))
        av=rd64(a[0],a[1]); bv=rd64(b[0],b[1])
        Am[lane%16, lane//16]=av; Bm[lane//16, lane%16]=bv
        C=np.zeros((16,16))
        if c:
            for p in range(4):
                cv=rd64(c[0],c[1]+2*p); C[(lane>>4)+4*p, lane&15]=cv
        with np.errstate(all='ignore'): D=Am@Bm+C
        for p in range(4): wr64(d[0],d[1]+2*p, D[(lane>>4)+4*p, lane&15].copy())
        trace.append((nm,ln+1,s,np.abs(D).max()))
    elif op=='v_accvgpr_read_b32':
        d=reg(ops[0]); a=reg(ops[1]); V[d[1]]=A[a[1]]
    elif op=='v_accvgpr_write_b32':
        d=reg(ops[0]); a=reg(ops[1]); A[d[1]]=V[a[1]]
    elif op=='v_xor_b32_e32':
        d=reg(ops[0]); assert ops[1]=='0x80000000'; a=reg(ops[2]); V[d[1]]=V[a[1]]^np.uint32(0x80000000)
    elif op=='v_mov_b32_e32':
        d=reg(ops[0]); a=reg(ops[1]); assert a and a[0]=='v', s; V[d[1]]=V[a[1]]
    elif op=='v_add_u32_e32':
        d=reg(ops[0]); a0=ops[1]; b=reg(ops[2])
        x=V[reg(a0)[1]] if reg(a0) else np.uint32(int(a0,0))
        V[d[1]]=(x+V[b[1]]).astype(np.uint32); print("note:",s)
    elif op=='ds_read2_b64':
        d=reg(ops[0]); rest2=ops[1].split(); ad=reg(rest2[0]); o0=o1=0
        for t in rest2[1:]:
            k,v=t.split(':'); 
            if k=='offset0': o0=int(v)
            if k=='offset1': o1=int(v)
        addr=V[ad[1]].astype(np.int64)
        if addr.max()+8*max(o0,o1)+8>len(LDS): print("OOB", s, addr[:8], "line", ln+1); addr=np.minimum(addr, len(LDS)-8*max(o0,o1)-8)
        for half,o in ((0,o0),(1,o1)):
            vals=np.array([np.frombuffer(LDS[a_+8*o:a_+8*o+8].tobytes(),dtype=np.float64)[0] for a_ in addr])
            wr64('v',d[1]+2*half,vals)
    elif op in ('s_nop','s_waitcnt'): pass
    else: raise SystemExit("unhandled: "+s)
print("emulated", nm, "mfma")
# compare emulated final state with the hardware state at the later dump
def diffregs(E,H,name):
    bad=[i for i in range(256) if not np.array_equal(E[i],H[i])]
    return bad
bv=diffregs(V,V1,'v'); ba=diffregs(A,A1,'a')
print("arch VGPRs differing emu vs hw:", bv); print("acc VGPRs differing emu vs hw:", ba)
for i in range(0,64,2):
    e=rd64('a',i); h=(A1[i].astype(np.uint64)|(A1[i+1].astype(np.uint64)<<np.uint64(32))).view(np.float64)
    col0=[0,16,32,48]
    print("a[%d:%d] emu col0 %s | hw col0 %s | max|emu-hw| %.3e" % (i,i+1,e[col0],h[col0],np.nanmax(np.abs(e-h))))
import pickle; pickle.dump(trace,open('/tmp/trace.pkl','wb'))
