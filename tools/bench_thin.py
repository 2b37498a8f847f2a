import torch, torch.nn.functional as F, sys, os
sys.path.insert(0, os.getcwd())
import deepinpainting_amd
from deepinpainting_amd import ops
deepinpainting_amd.use_shipped_miopen_db()
B=8
cb=torch.ops.aten.convolution_backward
cases=[("vgg1_1 conv 3->64 k3 @256",False,3,64,256,3,1,1),("netG conv 6->64 k3 @256",False,6,64,256,3,1,1),("netG convT 128->3 k3 @256",True,128,3,256,3,1,1),
       ("netP/D conv 3->64 k4s2 @256",False,3,64,256,4,2,1),("netP convT 128->3 k4s2 @128",True,128,3,128,4,2,1)]
for name,tr,ci,co,H,k,s,p in cases:
    x=torch.randn(B,ci,H,H,device="cuda"); w=torch.randn((ci,co,k,k) if tr else (co,ci,k,k),device="cuda")*0.05
    with torch.no_grad():
        y=F.conv_transpose2d(x,w,None,s,p) if tr else F.conv2d(x,w,None,s,p)
        dy=torch.randn_like(y)
        args=(dy,x,w,None,[s,s],[p,p],[1,1],tr,[0,0],1)
        for _ in range(5):
            (F.conv_transpose2d(x,w,None,s,p) if tr else F.conv2d(x,w,None,s,p))
            cb(*args,[True,False,False]); cb(*args,[False,True,False])
        torch.cuda.synchronize()
        def t20(fn):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1)/20
        if k==3:
            fop,bop=(ops.CONVT_FWD,ops.CONVT_BWD_DATA) if tr else (ops.CONV_FWD,ops.CONV_BWD_DATA)
            yo=torch.empty_like(y); dxo=torch.empty_like(x); dwo=torch.empty_like(w)
            print("%-32s thin: fwd %.4f  bwdD %.4f  wrw %.4f ms"%(name,t20(lambda:ops.conv3x3_thin(fop,x,w,(B,ci,H,H),co,out=yo)),t20(lambda:ops.conv3x3_thin(bop,dy,w,(B,ci,H,H),co,out=dxo)),t20(lambda:ops.conv3x3_thin_wrw(tr,x,dy,out=dwo))),flush=True)
        for nm,fn in (("fwd",lambda:(F.conv_transpose2d(x,w,None,s,p) if tr else F.conv2d(x,w,None,s,p))),("bwdD",lambda:cb(*args,[True,False,False])),("wrw",lambda:cb(*args,[False,True,False]))):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            print("%-32s %-4s %.4f ms (20 back-to-back calls: device-bound)"%(name,nm,e0.elapsed_time(e1)/20),flush=True)
