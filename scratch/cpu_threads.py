import sys, time, os, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'oracle'); sys.path.insert(0,'tests')
import _pkg; pkg=_pkg.load()
import iblnerf_oracle as O
from conftest import load_lut_rgb
from threadpoolctl import threadpool_limits, threadpool_info
ck=pkg.checkpoint
sdc,sdf=ck.synthetic_state_dict(0),ck.synthetic_state_dict(1)
lut=load_lut_rgb()
K=np.array([[692.82,0,400],[0,692.82,400],[0,0,1]],dtype=np.float32); c2w=np.concatenate([np.eye(3),np.zeros((3,1))],1).astype(np.float32)
ro,rd=O.get_rays(800,800,K,c2w); ro,rd=ro.reshape(-1,3),rd.reshape(-1,3)
print(os.cpu_count(), [ (p['internal_api'],p['num_threads']) for p in threadpool_info()])
for nt in (8,16,32,64,128):
    with threadpool_limits(limits=nt):
        O.render_rays(sdc,sdf,ro[:64],rd[:64],0.5,8.0,lut)
        t=time.perf_counter(); O.render_rays(sdc,sdf,ro[1000:1256],rd[1000:1256],0.5,8.0,lut); dt=time.perf_counter()-t
    print(nt, 'threads: %.1f rays/s'%(256/dt))
