import sys, numpy as np, torch
sys.path.insert(0, '.')
import levelsetfortran_amd as L
from levelsetfortran_amd import fields
N=int(sys.argv[1]); order=sys.argv[2]; K=int(sys.argv[3])
phi0, dx = fields.two_sphere_phi0((N,N,N)); h=fields.reinit_step(dx)
phi=torch.from_numpy(phi0.reshape(-1,order='F')).cuda(); phiS=phi.clone()
L.reinit(phi,None,None,N-1,N-1,N-1,K-1,dx,h,tol=0.0,order=order,arith='fast',phiS=phiS)
torch.cuda.synchronize()
