import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import orc
from voldor_amd import synth
sc = synth.make_scene(w=320, h=240, n_flows=5, fx=160, fy=160, cx=160, cy=120, seed=233)
fx, fy, cx, cy = sc["K"]; K = np.array([fx,0,cx,0,fy,cy,0,0,1],np.float32)
fl=sc["flows"]; N,h,w,_=fl.shape
ok,Ro,to=orc.estimate_pose_epipolar(fl[0],K)
do=orc.estimate_depth_closed_form(fl[0],K,Ro,to)
Rs=np.stack([Ro]+[np.eye(3,dtype=np.float32)]*(N-1)); ts=np.stack([to]+[np.zeros(3,np.float32)]*(N-1))
rig=np.ones((N,h,w),np.float32)
p2,p3=orc.compact_p3p(*orc.collect_p3p(fl,rig,do,K,Rs,ts,0))
rv,tv=orc.solve_batch_p3p(p3,p2,K,8192)
f=np.isfinite(rv.sum(1)+tv.sum(1)); pool=np.concatenate([rv[f]*25,tv[f]],1).astype(np.float32)
init=np.concatenate([orc.rotmat_to_angle_axis(Ro)*25,to]).astype(np.float32)
m,c,i=orc.meanshift(pool,0.2,init,False)
cov0=(np.eye(6)*0.2*1e4).astype(np.float32)
rng=np.random.default_rng(0)
base=orc.fit_robust_gaussian(pool*100,m*100,cov0)
print("base",base[0],base[1]/100,base[3],base[4])
for k in range(6):
    pp=(pool*(1+rng.normal(0,1e-6,pool.shape))).astype(np.float32)
    r=orc.fit_robust_gaussian(pp*100,m*100,cov0)
    print("pert",r[0],np.abs(r[1]-base[1]).max()/100,r[3],r[4])
from voldor_amd import kernels
g=kernels.fit_robust_gaussian(pool*100,m*100,cov0)
print("gpu ",g[0],g[1]/100,g[3],g[4], "diff", np.abs(g[1]-base[1]).max()/100)
print("cov diag oracle", np.diag(base[2]), "gpu", np.diag(g[2]))
