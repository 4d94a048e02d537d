import os, sys, subprocess
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from wav2lip_amd import _lib, bf16
from wav2lip_amd._lib import ACT_NONE, ConvGeom
def run(tr):
    torch.manual_seed(0)
    dev = torch.device("cuda")
    w = torch.randn((64, 64, 3, 3)) / 24
    x = torch.randn(1, 64, 16, 16)
    g = ConvGeom(int(tr), 64, 64, 3, 3, 1, 1, 1, 1, 0, 0, ACT_NONE)
    layer = bf16.ConvB(g, w.to(dev))
    xb = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)
    yb = torch.zeros(1, 16, 16, 64, dtype=torch.bfloat16, device=dev)
    layer.run(bf16.ActB(xb, 0, 64), bf16.ActB(yb, 0, 64), None, None, None, 0)
    torch.cuda.synchronize()
    return yb.float().cpu().numpy()
tr = int(sys.argv[1])
y = run(tr)
np.save("/tmp/y_%s_%d.npy" % (os.environ.get("W2L_CONVB_BOX", "1"), tr), y)
if os.environ.get("W2L_CONVB_BOX", "1") == "1":
    env = dict(os.environ, W2L_CONVB_BOX="0")
    subprocess.run([sys.executable, __file__, str(tr)], env=env, check=True)
    ref = np.load("/tmp/y_0_%d.npy" % tr)
    d = np.abs(y - ref)
    print("tr", tr, "max err", d.max(), "bad frac", (d > 0.05).mean())
    bad = d > 0.05
    print("bad per channel:", bad.reshape(-1, 64).sum(0).tolist())
    print("bad per row:", bad.sum(axis=(0, 2, 3)).tolist())
    print("bad per col:", bad.sum(axis=(0, 1, 3)).tolist())
    print("y[0,5,5,:16]", y[0, 5, 5, :16].round(2).tolist()); print("r[0,5,5,:16]", ref[0, 5, 5, :16].round(2).tolist())
