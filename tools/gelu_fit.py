# tools/gelu_fit.py -- minimax (LP) fit of the transcendental-free GELU in csrc/common.h: x*clamp(0.5 + xc P(xc^2), 0, 1).
# MARGIN keeps the fp32 Horner value at xc = R safely >= 1 so the clamp makes both tails exact (x, -0).
import numpy as np
MARGIN=4e-6
from scipy.special import erf
from scipy.optimize import linprog
Phi=lambda x:0.5*(1+erf(x/np.sqrt(2)))
def gelu(x): return x*Phi(x)
def solve(n,R):
    x=np.linspace(0,R,3001)
    # vars: c0..c{n-1}, t ; err = x*(0.5 + x*sum c_k x^{2k}) - gelu = sum c_k x^{2k+2} + 0.5x - gelu
    A=np.stack([x**(2*k+2) for k in range(n)],1); b=gelu(x)-0.5*x
    Aub=np.block([[A,-np.ones((len(x),1))],[-A,-np.ones((len(x),1))]]); bub=np.concatenate([b,-b])
    # phi(R)>=1:  -R*sum c_k R^{2k} <= -0.5
    row=np.concatenate([-R*np.array([R**(2*k) for k in range(n)]),[0]])
    Aub=np.vstack([Aub,row]); bub=np.append(bub,-0.5-MARGIN)
    c=np.zeros(n+1); c[-1]=1
    r=linprog(c,A_ub=Aub,b_ub=bub,bounds=[(None,None)]*n+[(0,None)],method='highs')
    return r.x[:n],r.x[-1]
def approx32(c,x,R):
    x=x.astype(np.float32); xc=np.clip(x,-np.float32(R),np.float32(R)); x2=xc*xc
    p=np.full_like(x,np.float32(c[-1]))
    for k in range(len(c)-2,-1,-1): p=p*x2+np.float32(c[k])
    return x*np.clip(np.float32(0.5)+xc*p,0,1)
xs=np.linspace(-100,100,1000001)
for n in (8,9,10):
    best=None
    for R in np.arange(3.0,6.01,0.125):
        try: c,t=solve(n,R)
        except Exception as e: continue
        e=np.max(np.abs(approx32(c,xs,R).astype(np.float64)-gelu(xs)))
        if best is None or e<best[0]: best=(e,R,c,t)
    print(n,'err32 %.3g'%best[0],'R',best[1],'lp t %.3g'%best[3]); print('   ',[float('%.9g'%v) for v in best[2]])
