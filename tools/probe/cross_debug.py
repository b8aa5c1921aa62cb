import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from recman_amd import ops
torch.manual_seed(0)
for (B, FD, Dn, L) in [(5, 8, 0, 1), (5, 40, 3, 3), (5, 416, 13, 6)]:
    d = FD + Dn
    xe = torch.randn(B, FD).cuda(); xd = torch.randn(B, Dn).cuda() if Dn else None
    w = (torch.randn(L, d) * 0.1).cuda(); b = (torch.randn(L, d) * 0.1).cuda(); wo = (torch.randn(d) * 0.1).cuda()
    logit = torch.empty(B).cuda(); p = torch.zeros(B, ops.cross_p_ld(L)).cuda()
    ops.cross_fwd(xe, xd, w, b, wo, logit, p)
    x0 = torch.cat([xe] + ([xd] if Dn else []), 1).double()
    want_p = torch.cat([x0 @ w.double().t(), (x0 @ wo.double()).view(-1, 1)], 1)
    x = x0
    for l in range(L):
        x = x0 * (x * w[l].double()).sum(1, keepdim=True) + b[l].double() + x
    print(B, FD, Dn, L)
    print(" p got ", p[:2, :L + 1].cpu().numpy())
    print(" p want", want_p[:2].cpu().numpy())
    print(" logit got ", logit.cpu().numpy()[:5])
    print(" logit want", (x @ wo.double()).cpu().numpy()[:5])
