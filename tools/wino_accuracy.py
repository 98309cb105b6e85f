"""CPU emulation of the fp32 pipeline of the row-transform conv kernels (csrc/conv_wino.hip, conv_wino4.hip): transforms evaluated term
by term in fp32, products accumulated per 16-channel group with SIX roundings (one per bf16 term of the bf16x3 product), output transform
in fp32 -- max error against fp64 relative to the output maximum, for F(2,3) and F(4,3) with several interpolation-point sets.
Usage: python tools/wino_accuracy.py   (profiles/r05/wino_accuracy.txt)"""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from toom_cook import toom
from fractions import Fraction as Fr
rng=np.random.default_rng(1)
f32=lambda x: x.astype(np.float32)
def matf(M): return np.array([[float(v) for v in row] for row in M],np.float64)
def seq_apply(M, X):
    """fp32 sequential evaluation of rows of M applied to X (first axis): out[i] = sum_k M[i,k]*X[k], fma-style (one rounding per term)"""
    out=[]
    for i in range(M.shape[0]):
        acc=None
        for k in range(M.shape[1]):
            if M[i,k]==0: continue
            term=M[i,k]*X[k].astype(np.float64)
            acc = f32(term) if acc is None else f32(acc.astype(np.float64)+term)
        out.append(acc if acc is not None else np.zeros_like(X[0],dtype=np.float32))
    return np.stack(out,0)
def acc_mfma(a,b):
    M,K=a.shape; out=np.zeros((M,b.shape[1]),np.float32)
    A=2.0**-8; Bq=2.0**-16
    for k0 in range(0,K,16):
        p=a[:,k0:k0+16].astype(np.float64)@b[k0:k0+16].astype(np.float64)
        for part in (p*Bq/3,p*Bq/3,p*Bq/3,p*A/2,p*A/2,p*(1-A-Bq)):
            out=f32(out.astype(np.float64)+part)
    return out
def run(C,pts,m=4,Wd=48,relu=True,nout=16,scale=None):
    AT,G,BT=[matf(x) for x in toom(m,3,pts)]
    n=m+2
    if scale is not None:   # diagonal rescale: T_c *= s_c, U_c /= s_c
        s=np.array(scale,np.float64); BT=BT*s[:,None]; G=G/s[:,None]
    x=rng.standard_normal((3,Wd+2,C)); 
    if relu: x=np.maximum(x,0)
    x=f32(x); w=f32(rng.standard_normal((3,3,C,nout))/np.sqrt(9*C))
    ref=np.zeros((Wd,nout))
    for kx in range(3): ref+=np.einsum('kxc,kcn->xn',x[:,kx:kx+Wd,:].astype(np.float64),w[:,kx].astype(np.float64))
    nq=Wd//m
    mm=[np.zeros((nq,nout),np.float32) for _ in range(n)]
    for ky in range(3):
        dd=np.stack([x[ky,i:i+m*nq:m] for i in range(n)],0)
        T=seq_apply(BT,dd)
        U=seq_apply(G,w[ky])
        for c in range(n): mm[c]=f32(mm[c].astype(np.float64)+acc_mfma(T[c],U[c]).astype(np.float64)) if False else acc_add(mm[c],T[c],U[c])
    Y=seq_apply(AT,np.stack(mm,0))
    o=Y.transpose(1,0,2).reshape(Wd,nout)
    return np.abs(o-ref).max()/np.abs(ref).max()
def acc_add(acc,a,b):
    M,K=a.shape; out=acc
    A=2.0**-8; Bq=2.0**-16
    for k0 in range(0,K,16):
        p=a[:,k0:k0+16].astype(np.float64)@b[k0:k0+16].astype(np.float64)
        for part in (p*Bq/3,p*Bq/3,p*Bq/3,p*A/2,p*A/2,p*(1-A-Bq)):
            out=f32(out.astype(np.float64)+part)
    return out
h=Fr(1,2)
cands={'F23 {0,1,-1}':(2,[0,1,-1]),
 'F43 {0,1,-1,2,-2}':(4,[0,1,-1,2,-2]),
 'F43 {0,1,-1,1/2,-1/2}':(4,[0,1,-1,h,-h]),
 'F43 {0,1,-1,1/2,-2}':(4,[0,1,-1,h,-2]),
 'F43 {0,1,-1,2,-1/2}':(4,[0,1,-1,2,-h]),
 'F43 {0,1/2,-1/2,3/2,-3/2}':(4,[0,h,-h,Fr(3,2),-Fr(3,2)]),
 'F43 {0,1,-1,3/2,-3/2}':(4,[0,1,-1,Fr(3,2),-Fr(3,2)]),
 'F43 {0,2/3,-2/3,4/3,-4/3}':(4,[0,Fr(2,3),-Fr(2,3),Fr(4,3),-Fr(4,3)]),
 'F43 {0,1/2,-1/2,1,-1} ':(4,[0,h,-h,1,-1]),
 'F43 {0,3/4,-3/4,3/2,-3/2}':(4,[0,Fr(3,4),-Fr(3,4),Fr(3,2),-Fr(3,2)]),
}
for name,(m,pts) in cands.items():
    res=[]
    for C,relu in ((64,True),(64,False),(256,True),(256,False)):
        e=max(run(C,pts,m=m,relu=relu) for _ in range(3))
        res.append('%.2e'%e)
    print(name.ljust(30),' C64 relu/N01: %s %s   C256: %s %s'%tuple(res))

# Which rounding dominates: the accumulator roundings (six per 16-channel group, one per bf16 term; one if only the leading term meets the
# big accumulator, as conv_wino4.hip does; none = exact accumulation) or the transforms
print()
for mode in ('six', 'one', 'exact'):
    def acc_add(acc, a, b, mode=mode):
        M, K = a.shape; out = acc
        A = 2.0 ** -8; Bq = 2.0 ** -16
        for k0 in range(0, K, 16):
            p = a[:, k0:k0 + 16].astype(np.float64) @ b[k0:k0 + 16].astype(np.float64)
            if mode == 'one':
                out = f32(out.astype(np.float64) + p)
            elif mode == 'six':
                for part in (p * Bq / 3, p * Bq / 3, p * Bq / 3, p * A / 2, p * A / 2, p * (1 - A - Bq)):
                    out = f32(out.astype(np.float64) + part)
            else:
                out = out.astype(np.float64) + p
        return out
    for name, (m, pts) in {'F23': (2, [0, 1, -1]), 'F43 {0,1,-1,2,-2}': (4, [0, 1, -1, 2, -2])}.items():
        res = []
        for C, relu in ((64, True), (64, False), (256, True), (256, False)):
            res.append('%.2e' % max(run(C, pts, m=m, relu=relu) for _ in range(3)))
        print(('accumulator roundings per group: ' + mode).ljust(40), name.ljust(20), ' C64 relu/N01: %s %s   C256: %s %s' % tuple(res))
