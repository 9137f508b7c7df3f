import itertools
def plan(LOGN):
    LT=LOGN-4; T=1<<LT; RL=1 if LOGN&1 else 2; REM=LOGN-RL; R0=4 if REM%4==0 else REM%4
    NMID=(REM-R0)//4; NG=NMID+2; NL=min(LT,6)
    R=lambda g: R0 if g==0 else (RL if g==NG-1 else 4)
    S=lambda g: sum(R(i) for i in range(g))
    LO=lambda g: LOGN-S(g)-R(g)
    def EB(g,b):
        if g==0: return LOGN-4+b
        if g==NG-1: return b if b<RL else RL+NL+(b-RL)
        return LO(g)+b
    def TB(g,t):
        if g==0: return t
        if g==NG-1: return RL+t if t<NL else t+4
        return t if t<LO(g) else t+4
    return dict(LOGN=LOGN,LT=LT,T=T,NG=NG,EB=EB,TB=TB)
def ibase(P,g,t): return sum(((t>>b)&1)<<P['TB'](g,b) for b in range(P['LT']))
def thread_of(P,g,i): return sum(((i>>P['TB'](g,b))&1)<<b for b in range(P['LT']))
def slot_of(P,g,i): return sum(((i>>P['EB'](g,b))&1)<<b for b in range(4))
def wconf(P,gw,gr,addrfn):
    """max conflict degree of ds_write_b64 (16-lane groups, 16 8B-columns)"""
    worst=1
    T=P['T']
    for t0 in range(0,T,16):
        cnt={}
        for l in range(min(16,T)):
            i=ibase(P,gw,t0+l); a=addrfn(slot_of(P,gr,i),thread_of(P,gr,i))
            cnt[a%16]=cnt.get(a%16,0)+1
        worst=max(worst,max(cnt.values()))
    return worst
def rconf(P,gr,addrfn):
    """ds_read_b64: 32-lane halves, 32 8B-columns; reader reads slot e, threads consecutive"""
    worst=1; T=P['T']
    for e in range(16):
      for t0 in range(0,T,32):
        cnt={}
        for l in range(min(32,T)):
            a=addrfn(e,t0+l); cnt[a%32]=cnt.get(a%32,0)+1
        worst=max(worst,max(cnt.values()))
    return worst
for LOGN in range(8,15):
    P=plan(LOGN); T=P['T']
    print("LOGN",LOGN,"NG",P['NG'])
    for g in range(P['NG']-1):
        for (gw,gr) in ((g,g+1),(g+1,g)):
            res=[]
            for pad in range(0,20):
                f=lambda s,t,pad=pad:(s*(T+pad)+t)
                res.append(wconf(P,gw,gr,f))
            print("  x %d->%d pad conflicts:"%(gw,gr),res)
