import itertools, sys
def plan(LOGN):
    LT=LOGN-4; RL=1 if LOGN&1 else 2; REM=LOGN-RL; R0=4 if REM%4==0 else REM%4
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
    return LT,NG,NL,EB,TB
def solve(LOGN):
    LT,NG,NL,EB,TB=plan(LOGN)
    lanes=[[TB(g,t) for t in range(NL)] for g in range(NG)]   # lane index bits per group (default order)
    slots=[[EB(g,b) for b in range(4)] for g in range(NG)]
    nw=min(4,NL)
    for pad in [1,2,3,4,5,6,7,8,0,9,10,11,12]:
        # candidates: ordered choice of first nw lane bits for each group
        cands=[list(itertools.permutations(lanes[g],nw)) for g in range(NG)]
        def contrib(gr,perm_r,bit):
            # reader gr with lane ordering perm_r (tuple of first nw bits; remaining lane bits >= position nw)
            if bit in slots[gr]: return (pad<<slots[gr].index(bit))%16 if True else 0
            if bit in perm_r: return (1<<perm_r.index(bit))%16
            if bit in lanes[gr]: return 0 if nw>=4 else None  # position >= nw : 2^t with t>=4 -> 0 mod 16 (only if nw==4)
            return 0  # wave bit of reader: >= 64
        def ok(gw,pw,gr,pr):
            cs=[]
            for bit in pw:
                c=contrib(gr,pr,bit)
                if c is None:
                    # lanes<4 bits case: remaining lanes positions
                    rest=[b for b in lanes[gr] if b not in pr]
                    c=(1<<(nw+rest.index(bit)))%16
                cs.append(c)
            sums=set()
            for m in range(1<<len(cs)):
                s=sum(c for k,c in enumerate(cs) if (m>>k)&1)%16
                if s in sums: return False
                sums.add(s)
            return True
        # chain search
        def rec(g,chosen):
            if g==NG: return chosen
            for p in cands[g]:
                if g>0 and not (ok(g-1,chosen[-1],g,p) and ok(g,p,g-1,chosen[-1])): continue
                r=rec(g+1,chosen+[p])
                if r: return r
            return None
        r=rec(0,[])
        if r:
            full=[]
            for g in range(NG):
                rest=[b for b in lanes[g] if b not in r[g]]
                full.append(list(r[g])+rest)
            return pad,full
    return None
for LOGN in range(6,15):
    res=solve(LOGN)
    print(LOGN,res)
