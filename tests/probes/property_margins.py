import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/enlsip.jl_amd/python')
import torch; torch.zeros(1,device='cuda:0')
from oracle import synth
import importlib.util
spec=importlib.util.spec_from_file_location('tp','/root/repo/tests/test_gpu_defining_properties.py'); tp=importlib.util.module_from_spec(spec); spec.loader.exec_module(tp)
from enlsip_gn import GNSolver
s=GNSolver(device=0)
def margins(p,J,r,A,c):
    pd,Z=tp.direction_by_null_space(J,r,A,c)
    e=np.linalg.norm(p-pd)/max(np.linalg.norm(pd),1e-300)
    f=np.linalg.norm(A@p+c)/(np.linalg.norm(A,2)*np.linalg.norm(p)+np.linalg.norm(c)) if A.shape[0] else 0
    g=np.linalg.norm(Z.T@(J.T@(J@p+r)))/(np.linalg.norm(J,2)*(np.linalg.norm(J,2)*np.linalg.norm(p)+np.linalg.norm(r))) if Z.shape[1] else 0
    return e,f,g
worst=[0,0,0]
for (m,n,t) in tp.SHAPES+[(4096,512,64)]:
    J,rx,A,cx=synth.make_problem(9100+m+3*n+7*t,m,n,t)
    o=s.solve(J,rx,A,cx); mm=margins(o.p,J,rx,A,cx); worst=[max(a,b) for a,b in zip(worst,mm)]
    print(m,n,t,['%.1e'%x for x in mm])
for (m,n,t) in [(64,16,4),(300,40,6),(4096,96,32),(1024,320,100),(4096,512,64)]:
    J,rx,A,cx=synth.make_rank_deficient_A(4400+m+n,m,n,t)
    o=s.solve(J,rx,A,cx); mm=margins(o.p,J,rx,A[:-1],cx[:-1]); worst=[max(a,b) for a,b in zip(worst,mm)]
    print('dupA',m,n,t,['%.1e'%x for x in mm])
print('worst (tolerances 1e-10, 1e-11, 1e-10):',['%.1e'%x for x in worst])
