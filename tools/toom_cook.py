"""Exact Toom-Cook matrices (A^T, G, B^T) of F(m, r) for a list of interpolation points (+ infinity), in Fractions; check() verifies the
convolution identity on random integers.  Used by tools/wino_accuracy.py."""
import numpy as np, itertools
from fractions import Fraction as Fr
def poly_mul(a,b):
    r=[Fr(0)]*(len(a)+len(b)-1)
    for i,x in enumerate(a):
        for j,y in enumerate(b): r[i+j]+=x*y
    return r
def toom(m,r,pts):
    n=m+r-1; assert len(pts)==n-1
    pts=[Fr(p) for p in pts]
    AT=[[ (pts[j]**i if not (pts[j]==0 and i==0) else Fr(1)) for j in range(n-1)]+[Fr(1) if i==m-1 else Fr(0)] for i in range(m)]
    G=[]
    for j in range(n-1):
        N=Fr(1)
        for l in range(n-1):
            if l!=j: N*= (pts[j]-pts[l])
        G.append([ (pts[j]**k if not (pts[j]==0 and k==0) else Fr(1))/N for k in range(r)])
    G.append([Fr(0)]*(r-1)+[Fr(1)])
    BT=[]
    for j in range(n-1):
        p=[Fr(1)]
        for l in range(n-1):
            if l!=j: p=poly_mul(p,[-pts[l],Fr(1)])
        # multiply by ... degree n-2 -> pad
        BT.append(p+[Fr(0)]*(n-len(p)))
    p=[Fr(1)]
    for l in range(n-1): p=poly_mul(p,[-pts[l],Fr(1)])
    BT.append(p)
    return np.array(AT,dtype=object),np.array(G,dtype=object),np.array(BT,dtype=object)
def check(m,r,pts):
    AT,G,BT=toom(m,r,pts)
    n=m+r-1
    import random
    d=[Fr(random.randint(-9,9)) for _ in range(n)]; g=[Fr(random.randint(-9,9)) for _ in range(r)]
    U=[sum(G[i][k]*g[k] for k in range(r)) for i in range(n)]
    T=[sum(BT[i][k]*d[k] for k in range(n)) for i in range(n)]
    y=[sum(AT[i][j]*U[j]*T[j] for j in range(n)) for i in range(m)]
    ref=[sum(d[i+k]*g[k] for k in range(r)) for i in range(m)]
    return y==ref
if __name__=="__main__":
    print(check(2,3,[0,1,-1]), check(4,3,[0,1,-1,2,-2]), check(4,3,[0,1,-1,Fr(1,2),-2]))
    AT,G,BT=toom(4,3,[0,1,-1,2,-2])
    print(AT); print(G); print(BT)
