import numpy as np
# float64 model of the IF AGC recurrence g <- g*(1 + r(1 - g^2 e)), one Newton multiple-shooting round from a flat guess
r=1e-4; C=256
rng=np.random.default_rng(1)
def run(sigma, amp, g0, N=2_000_000):
    x=amp*np.exp(1j*rng.uniform(0,2*np.pi,N)) + sigma*(rng.standard_normal(N)+1j*rng.standard_normal(N))
    e=(x.real**2+x.imag**2)
    # serial truth
    g=g0; truth=np.empty(N//C+1); truth[0]=g
    for c in range(N//C):
        for i in range(c*C,(c+1)*C):
            g=g*(1+r*(1-g*g*e[i]))
        truth[c+1]=g
    nc=N//C
    nodes=np.full(nc+1,g0)
    res=[]
    for rnd in range(3):
        G=np.empty(nc); M=np.empty(nc)
        # vectorised over chunks
        g=nodes[:nc].copy(); dg=np.ones(nc)
        E=e[:nc*C].reshape(nc,C)
        for i in range(C):
            nrm=g*g*E[:,i]; z=1+r*(1-nrm); dg*= (z-2*r*nrm); g=g*z
        G=g; M=dg
        new=np.empty(nc+1); new[0]=nodes[0]
        v=nodes[0]
        for c in range(nc):
            v=G[c]+M[c]*(v-nodes[c]); new[c+1]=v
        maxrel=np.max(np.abs(new[1:]-nodes[1:])/np.abs(new[1:]))
        err=np.max(np.abs(new-truth)/truth); err_end=abs(new[-1]-truth[-1])/truth[-1]
        res.append((maxrel,err,err_end))
        nodes=new
    return res
for sigma,amp,g0 in [(1e-3,0.5,2.0),(1e-2,0.5,2.0),(3e-2,0.5,2.0),(1e-1,0.5,2.0),(1e-2,0.5,2.02),(3e-2,0.5,1.9)]:
    print(sigma,amp,g0,[tuple(float(f"{v:.3g}") for v in t) for t in run(sigma,amp,g0,N=400_000)])
