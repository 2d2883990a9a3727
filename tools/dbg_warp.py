import numpy as np, torch, sys
sys.path.insert(0,'.')
import imagestitch_amd as I
from imagestitch_amd import synth
from oracle import capi as O
rng=np.random.default_rng(3)
for (w,h) in [(433,269),(432,269),(64,40)]:
    f=300.0
    K,Rs=synth.camera_pair(w,h,f,yaw=0.3)
    img=rng.integers(0,256,(h,w,3),dtype=np.uint8)
    wp=I.CylindricalWarper().create(f)
    c,wi,wm=wp.warp_with_mask(torch.from_numpy(img).cuda(),K,Rs[1])
    oc,oi,_=O.warp_u8(O.CYL,f,K,Rs[1],img,1,2)
    g=wi.cpu().numpy()
    d=(g!=oi)
    print((w,h), g.shape, 'mismatch per channel', d[...,0].sum(), d[...,1].sum(), d[...,2].sum(), 'of', d[...,0].size)
    ys,xs=np.nonzero(d[...,0])
    print('  x%4 hist', np.bincount(xs%4,minlength=4), 'rows min/max', ys.min() if len(ys) else None, ys.max() if len(ys) else None, 'cols', xs.min() if len(xs) else None, xs.max() if len(xs) else None)
    if len(xs):
        for y,x in list(zip(ys,xs))[:6]:
            print('   ',y,x,g[y,x],oi[y,x], 'neighbors exp', oi[y,max(x-1,0)], oi[y,min(x+1,g.shape[1]-1)])
