import sys; sys.path.insert(0,'.')
import numpy as np
from voldor_amd import kernels, synth
def K9(fx,fy,cx,cy): return np.array([fx,0,cx,0,fy,cy,0,0,1],np.float32)
for (w,h,N,ndp) in ((640,360,10,1),(640,360,10,0),(640,360,6,1),(320,240,10,0)):
    sc=synth.make_scene(w=w,h=h,n_flows=N,fx=w/2,fy=w/2,cx=w/2,cy=h/2,seed=7,basefocal=w/4)
    rng=np.random.default_rng(3)
    flows=sc['flows']; gt=sc['poses_gt']
    Rs=np.stack([synth.rodrigues(gt[i,:3]) for i in range(N)]).astype(np.float32); ts=gt[:,3:].astype(np.float32)
    depth=(sc['depth_gt']*(1+rng.normal(0,0.3,(h,w)))).astype(np.float32); rig=rng.uniform(0.2,1,(N,h,w)).astype(np.float32)
    pri=pc=cf=dR=dt=None
    if ndp:
        pri=((w/4)/sc['disparity'])[None].astype(np.float32); pc=np.ones_like(pri); cf=np.ones_like(pri); dR=np.eye(3,dtype=np.float32)[None]; dt=np.zeros((1,3),np.float32)
    out=[]
    for thr,order in ((1<<62,1<<62),(0,1<<62),(0,0)):
        kernels.set_frame_major_threshold(thr,order); kernels.set_rand_epoch(5)
        out.append(kernels.optimize_depth_gpu(flows,rig,pri,pc,cf,depth,K9(w/2,w/2,w/2,h/2),Rs,ts,dR,dt,1.0,N,ndp,w,h,w/4 if ndp else 0.0,10,0,0,0.15,0.15,1.0 if ndp else -1.0,0.2,0,0.5,0.9,1.0,0))
    d0=out[0][0]
    for o in out[1:]:
      d1=o[0]; bad=np.argwhere(d0!=d1)
      print((w,h,N,ndp),'depth mismatches',len(bad),'of',w*h, 'rig mismatches',int(np.sum(out[0][1]!=o[1])))
    for y,x in bad[:5]: print('   ',x,y,d0[y,x],d1[y,x])
