import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpar_amd import hip
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = torch.Generator(device="cpu"); g.manual_seed(n)
X = torch.rand(n, 4, generator=g, dtype=torch.float64).to(dev)
K = hip.alloc_matrix(n, n, dev); K.copy_(torch.exp(-0.5 * torch.cdist(X, X) ** 2 / 0.25)); K.diagonal().add_(0.1)
hip.potrf_(K); torch.cuda.synchronize()
hip.chol_inverse(K); torch.cuda.synchronize()
K.add_(0.0)  # marker (elementwise kernel) before the traced inverse
hip.chol_inverse(K); torch.cuda.synchronize()
