import numpy as np, sys, math
FIRST=8
def load(path, mrd, N=4096):
    a=np.fromfile(path,dtype=np.int32).reshape(N,N,3)
    cnt=a[...,0]; mu=a[...,1]; p=a[...,2]
    K=(mrd-1-FIRST)//8+1
    never=cnt==0
    has=never&(p>0)
    Lc=np.zeros(cnt.shape,np.int32)
    pp=p[has].astype(np.int64)
    Lc[has]=(pp//np.gcd(pp,8)).astype(np.int32)   # lcm(8,p)/8
    kmu=np.zeros(cnt.shape,np.int32)
    kmu[has]=np.maximum(0,(mu[has]-FIRST+7)//8)
    Lc=np.minimum(Lc,K+1)
    return dict(cnt=cnt,never=never,has=has,Lc=Lc,kmu=kmu,K=K,mrd=mrd,N=N)
def schedule(nextwin, win0, K):
    U=[0]; W=[]; win=win0; u=0
    while u<=K+1:
        W.append(win); u+=win; U.append(u); win=nextwin(win)
    W.append(win)
    return np.array(U),np.array(W)
def evaluate(d, U, W):
    K=d['K']; mrd=d['mrd']
    INF=10**6
    tab=np.full((K+2,K+2),INF,np.int64)   # [Lc, kmu]
    for L in range(1,K+2):
        ok=W>=L
        cand=np.where(ok,U+L,INF)
        # for each kmu: min over i with U[i]>=kmu of cand[i]; U increasing -> suffix min
        suf=np.minimum.accumulate(cand[::-1])[::-1]
        idx=np.searchsorted(U,np.arange(K+2),side='left')
        idx=np.minimum(idx,len(U)-1)
        tab[L,:]=np.where(U[idx]>=np.arange(K+2),suf[idx],INF)
    ex=np.where(d['never'],mrd-1,d['cnt']).astype(np.int64)
    has=d['has']
    k=tab[d['Lc'][has],np.minimum(d['kmu'][has],K+1)]
    ex[has]=np.minimum(FIRST+8*k,mrd-1)
    px=ex.sum()
    blk=ex.reshape(d["N"]//8,8,d["N"]//8,8).max(axis=(1,3))
    return px/1e6, blk.sum()/1e6, (ex[d['never']]<mrd-1).mean()
if __name__=="__main__":
    views=[("cfg2","/tmp/cw/cfg2.bin",1000),("chunk(1,0,0)","/tmp/cw/c100.bin",1024),("chunk(4,1,1)","/tmp/cw/c411.bin",1024)]
    data=[(n,load(p,m)) for n,p,m in views]
    pols={"doubling (now)":(lambda w:2*w,1),
          "x1.5":(lambda w:w+max(1,w>>1),1),
          "x1.25":(lambda w:w+max(1,w>>2),1),
          "x1.25 from 4":(lambda w:w+max(1,w>>2),4),
          "x1.25 from 8":(lambda w:w+max(1,w>>2),8),
          "x1.125 from 8":(lambda w:w+max(1,w>>3),8),
          "fixed 8":(lambda w:w,8),"fixed 12":(lambda w:w,12),"fixed 16":(lambda w:w,16),"fixed 24":(lambda w:w,24),
          "+1":(lambda w:w+1,1), "+2 from 4":(lambda w:w+2,4), "+1 from 8":(lambda w:w+1,8),"+2 from 8":(lambda w:w+2,8),"+4 from 8":(lambda w:w+4,8),
          }
    base={}
    for pn,(f,w0) in pols.items():
        out=[]
        for n,d in data:
            U,W=schedule(f,w0,d['K'])
            px,ws,early=evaluate(d,U,W)
            if pn.startswith("doubling"): base[n]=ws
            out.append(f"{n}: px {px:7.1f} M ws {ws:7.3f} M ({100*(ws/base[n]-1):+5.1f} %) early {100*early:4.1f}%")
        print(f"{pn:16s} "+" | ".join(out))
