import numpy as np, sys
def count(k, R, C, WR, WC, rows_pad=None):
    n = k + 7                      # bordered rows
    BR, BC = R*WR, C*WC
    nr = -(-n // R) * R            # rows padded to wave-row granularity? (a sub-block is issued only if live)
    total = 0
    npan = -(-k // 4)
    # enumerate sub-blocks
    subs = []
    for r0 in range(0, n, R):
        for c0 in range(0, n, C):
            if c0 > r0 + R - 1: continue
            subs.append((r0, c0))
    subs = np.array(subs)
    r1 = np.minimum(subs[:,0] + R - 1, n - 1)      # last row of the sub-block
    c0 = subs[:,1]
    for p in range(npan):
        lo = 4*p + 4
        # live iff exists i in [max(r0,lo), r1], j in [max(c0,lo), c1] with j <= i  <=> max(c0,lo) <= r1 and r1 >= lo
        live = (np.maximum(c0, lo) <= r1) & (r1 >= lo) & (c0 + C - 1 >= lo)
        total += 4 * live.sum()
    useful = 0
    for p in range(npan):
        lo = 4*p+4
        m = n - lo
        useful += 4 * m*(m+1)/2
    return total, useful/64.0
for k in (40, 60, 76, 96, 104, 120, 136, 147):
    out = []
    for name,(R,C,WR,WC) in (("cur16x4",(16,4,1,1)),("T4x16",(4,16,1,1)),("8x8",(8,8,1,1)),("2x32",(2,32,1,1)),("1x64",(1,64,1,1))):
        t,u = count(k,R,C,WR,WC)
        out.append("%s %5d (%.2f)" % (name, t, t/u))
    print(k, "ideal %.0f |" % u, " | ".join(out))
print("---- Z variant (RHS as k x 8 column block) for the 16x4 layout, and k-averaged comparison")
def count_n(n, k, R, C):
    subs=[(r0,c0) for r0 in range(0,n,R) for c0 in range(0,n,C) if c0 <= r0+R-1]
    subs=np.array(subs); r1=np.minimum(subs[:,0]+R-1,n-1); c0=subs[:,1]
    tot=0
    for p in range(-(-k//4)):
        lo=4*p+4
        live=(np.maximum(c0,lo)<=r1)&(r1>=lo)&(c0+C-1>=lo)
        tot+=4*live.sum()
    return tot
def zvar(k):
    c = count_n(k, k, 16, 4)
    z = 0
    for p in range(-(-k//4)):
        lo = 4*p+4
        nbr = sum(1 for r0 in range(0,k,16) if min(r0+15,k-1) >= lo)
        z += nbr * 2 * 4
    return c + z
for k in (40, 52, 60, 68, 76, 84, 96, 104, 112, 120, 136, 147):
    n=k+7
    useful=sum(4*(n-4*p-4)*(n-4*p-3)/2 for p in range(-(-k//4)))/64
    cur=count_n(n,k,16,4); zv=zvar(k); t=count_n(n,k,4,16)
    print(k, "ideal %.0f cur %d (%.2f)  Z %d (%.2f)  T %d (%.2f)  best/cur %.2f" % (useful, cur, cur/useful, zv, zv/useful, t, t/useful, min(cur,zv,t)/cur))
print("---- round 5: multi-wave kernels, bordered (k_uk<NB, NW>) against border-as-columns with REPLICATED border registers (k_ukz<NB, NW>)")
def uk(NB, NW, k, z=False):
    """fmac wave-instructions per SYSTEM of the panel updates, as the kernels unroll them (twx_uk.h): 4 per live block and
    panel; z: + 8 per live block row and panel and wave for the border columns, + 8 per panel for the 7x7 corner."""
    CB = 4 * NW
    NBC = 16 * NB // CB
    a_tot = z_tot = 0
    for bc in range(NBC):
        ncb = k - CB * bc
        if ncb <= 0:
            continue
        a0 = CB * bc // 16
        for s in range(min(NW, (ncb + 3) // 4)):
            for wvp in range(NW):
                for a in range(a0, NB):
                    a_tot += 4 * max(0, 16 * (a + 1) // CB - (bc + 1)) + (4 if wvp > s else 0)
                    z_tot += 8 if z else 0
                z_tot += 8 if (z and wvp == 0) else 0
    return a_tot, z_tot
for name, NB, NW, k, z in (("k_uk<8,4>  k=110", 8, 4, 110, False), ("k_ukz<7,2> k=110", 7, 2, 110, True), ("k_uk<9,2>  k=121", 9, 2, 121, False),
                           ("k_ukz<8,2> k=121", 8, 2, 121, True), ("k_uk<7,2>  k=100", 7, 2, 100, False)):
    a, zt = uk(NB, NW, k, z)
    print("%-18s matrix %5d + border %5d = %5d fmac instructions per system" % (name, a, zt, a + zt))
