import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from blurred_gan_amd import ops
B,H,W,C=64,256,256,3
x=torch.rand(B,H,W,C,device="cuda"); y=torch.empty_like(x)
ks,se,nt=ops.blur_policy(5.0,H,W); taps=torch.tensor(ops.gauss_kernel_1d(se,ks),device="cuda")
dbg=torch.zeros(1<<20,dtype=torch.float32,device="cuda")
for _ in range(30): ops.blur_nhwc(x,y,taps,nt,dbg)
torch.cuda.synchronize(); dbg.zero_(); ops.blur_nhwc(x,y,taps,nt,dbg); torch.cuda.synchronize()
d=dbg.view(torch.int64)[:2048*4].view(-1,4).cpu().double()
d=d[d[:,0]>0]
cyc=(d[:,2]-d[:,0]); ref=(d[:,3]-d[:,1])
print("waves",len(d),"mean wave life: %.0f shader ticks, %.0f ref ticks (100 MHz) = %.2f us -> clock %.3f GHz"%(cyc.mean(),ref.mean(),ref.mean()/100, (cyc/ref).mean()*0.1))
print("span all waves: %.2f us"%((d[:,3].max()-d[:,1].min())/100))
