import sys; sys.path.insert(0,'/root/repo')
import torch
from gpar_amd import hip
dev=torch.device('cuda:0')
for n,bad in [(10000,7777),(16384,100),(16384,16000),(3000,2999),(700,0)]:
    g=torch.Generator().manual_seed(1)
    X=torch.rand(n,3,generator=g,dtype=torch.float64).to(dev)
    A=hip.alloc_matrix(n,n,dev); A.copy_(torch.exp(-0.5*torch.cdist(X,X)**2/0.25)); A.diagonal().add_(0.1)
    A[bad,bad] = -5.0
    logdet,info=hip.potrf_(A)
    torch.cuda.synchronize()
    print(n,bad,int(info.item()), "OK" if int(info.item())==bad+1 else "MISMATCH")
