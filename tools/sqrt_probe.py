"""Developer probe: is torch's exp(lv) ** 0.5 on the device the IEEE square root (what vbq_prep_planes_f32 takes)?"""
import numpy as np
import torch
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vbq_amd import ops

rng = np.random.default_rng(0)
lv = rng.normal(-4, 3, (4096, 256)).astype(np.float32)
lv.reshape(-1)[:6] = [-200.0, 88.0, -87.5, 0.0, -103.0, -95.0]
var = torch.exp(torch.from_numpy(lv).cuda())
mine = ops.prep_planes(var, var, spread="variance")[1].t().contiguous()
p = var ** 0.5
s = torch.sqrt(var)
ref = torch.sqrt(var.double()).float()
for name, t in (("pow0.5", p), ("torch.sqrt", s), ("prep_planes", mine)):
    bad = (t != ref)
    print(name, "differs from f64-rounded sqrt in", int(bad.sum()), "of", t.numel())
    if bad.any():
        i = bad.nonzero()[:5]
        for r, c in i.tolist():
            print("   var", float(var[r, c]), var[r, c].view(torch.int32).item(), "got", float(t[r, c]), "ref", float(ref[r, c]),
                  "denormal" if float(var[r, c]) < 1.1754944e-38 else "")
print("pow vs sqrt differ:", int((p != s).sum()))
dn = var < 1.1754944e-38
print("denormal variances:", int(dn.sum()), "pow==mine on normals:", bool(torch.equal(p[~dn], mine[~dn])))
