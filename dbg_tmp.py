import os, sys, ctypes as C
os.environ["MDH_HOUGH_DEBUG"]="1"
sys.path.insert(0,'.')
import numpy as np
from metdetpy_amd import M3Detector, _lib
from metdetpy_amd.synth import SkyStream
from bench import make_cfg
H,W,n,B=1080,1920,8,128
s=SkyStream(H,W,20220413); frames=s.take(B)
det=M3Detector(1,n,np.ones((H,W),np.uint8),10,make_cfg(),None,max_batch=B)
ctx=det._ctx
pool=ctx.device_alloc(frames.nbytes); ctx.device_upload(pool,frames)
for it in range(3):
    info,raw,nms,prob=ctx.process_batch(pool,K=B,on_device=True)
out=np.zeros((B,8),np.int64)
L=_lib.load(); L.mdh_debug_hough.argtypes=[C.c_void_p,C.c_void_p,C.c_int]
L.mdh_debug_hough(ctx.handle,out.ctypes.data,B)
order=np.argsort(-out[:,2])
print("N build ppht | shuffle mask vote key line")
for i in order[:6]: print(i, out[i].tolist())
