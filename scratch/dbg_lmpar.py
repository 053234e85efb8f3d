import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import nonlin_amd._lib as L
L.LIB_PATH = os.path.join(os.getcwd(), "scratch", "libnonlin_hip_dbg.so")
from nonlin_amd.device import DeviceSolver
from oracle import pyoracle as O
m,n,dsc = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
ds = DeviceSolver(0)
A,b,xt,x0 = O.dq_generate(11,m,n,gamma=2.0,sigma=0.1,spread=2.0,square_shift=(m==n))
f0 = O.dq_residual(A,b,2.0,x0)
J = O.dq_fd_jacobian(A,b,2.0,x0,fv=f0)
a, ip, rd, acn = O.lmfactor(J)
w=f0.copy()
for j in range(n):
    if a[j,j]!=0:
        t=-np.dot(a[j:,j],w[j:])/a[j,j]; w[j:]+=a[j:,j]*t
    a[j,j]=rd[j]
qtf=w[:n].copy(); diag=np.where(acn==0,1.0,acn)
delta=dsc*np.linalg.norm(diag*x0)
os.environ["NLO_DEBUG_LMPAR"]="1"
par_o,x_o,sd_o,_=O.lmpar(a,ip,diag,qtf,delta,0.0,w)
dev="cuda"
R = torch.tensor(np.ascontiguousarray(a[:n,:n].T), device=dev).unsqueeze(0)
par,x,sd = ds.lmpar(R, torch.tensor(ip,dtype=torch.int32,device=dev).unsqueeze(0), torch.tensor(diag,device=dev).unsqueeze(0),
   torch.tensor(qtf,device=dev).unsqueeze(0), torch.tensor([delta],device=dev), torch.tensor([float(np.sum(w[n:]**2))],device=dev), torch.tensor([0.0],device=dev))
torch.cuda.synchronize()
print("oracle par", par_o, "gpu par", float(par[0]))
print("x diff", np.abs(x[0].cpu().numpy()-x_o).max(), "sdiag diff", np.abs(sd[0].cpu().numpy()-sd_o).max())
