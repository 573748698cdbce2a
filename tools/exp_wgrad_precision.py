import sys, torch, numpy as np
sys.path.insert(0,'.')
from musicfpaugment_amd import ops_train as T
g=torch.Generator().manual_seed(0)
B,H,W,Ci,Co=4,64,62,256,256
x=torch.randn(B,H,W,Ci,generator=g).relu().cuda(); dz=(torch.randn(B,H,W,Co,generator=g)*0.01).cuda()
ref=torch.zeros(9,Co,Ci,device='cuda'); T.wgrad_mfma(dz,x,ref,Co,precision=0)
for prec in (1, 2):
    got=torch.zeros(9,Co,Ci,device='cuda'); T.wgrad_mfma(dz,x,got,Co,precision=prec)
    rel=((got-ref).abs().sum()/ref.abs().sum()).item()
    print(f"precision {prec}: relative L1 of the weight gradient vs fp32 MFMA: {rel:.3e}")
