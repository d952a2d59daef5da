import itertools
G128 = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128 = G128 + [[l+32 for l in g] for g in G128]
def cycles(addrs_by_lane, groups, width_dw):
    tot=0
    for g in groups:
        banks={}
        for l in g:
            a=addrs_by_lane[l]
            for d in range(width_dw):
                b=((a//4)+d)%64
                banks.setdefault(b,set()).add((a//4)+d)
        tot+=max(len(v) for v in banks.values())
    return tot
def frag_rows_addr(RS, rbase, ks, lane, swz=None):
    row=rbase+(lane&15); col=(32*ks+8*(lane>>4))*2
    if swz: col=swz(row,col)
    return row*RS+col
def frag_tr_addr(RS, dt, c, lane, half, swz=None):
    i=lane&15; g=lane>>4
    row=32*c+4*g+(i>>2)+16*half; col=(16*dt+4*(i&3))*2
    if swz: col=swz(row,col)
    return row*RS+col
def evaluate(HDP, RS, swz=None):
    KS=HDP//32; DT=HDP//16
    t1=0;n1=0
    for j in range(4):
        for ks in range(KS):
            a=[frag_rows_addr(RS,16*j,ks,l,swz) for l in range(64)]
            t1+=cycles(a,G128,4); n1+=1
    t2=0;n2=0
    G64=[list(range(32)),list(range(32,64))]
    for c in range(2):
        for dt in range(DT):
            for half in range(2):
                a=[frag_tr_addr(RS,dt,c,l,half,swz) for l in range(64)]
                t2+=cycles(a,G64,2); n2+=1
    return t1/n1, t2/n2   # ideal 4 and 2
for HDP in (96,64,128):
    print("HDP",HDP)
    for pad in (0,16,32,48,64,80,96,112):
        RS=HDP*2+pad
        print("  pad",pad,"RS",RS,evaluate(HDP,RS))
    # xor swizzle on 16B chunks: col ^= ((row&7)<<4) with RS=HDP*2 (only valid if row chunks power of 2) - test for 128
    for m in (7,15):
        sw=lambda row,col,m=m:(col ^ ((row&m)<<4))
        if HDP in (64,128):
            print("  xor",m,evaluate(HDP,HDP*2,sw))
