"""CPU experiment behind conv_wino4.hip: Winograd F(4x4,3x3) in fp32 (points 0, +-1, +-2, inf; weights transformed in fp64 and
rounded once) on every eligible 3x3 layer of the generator vs the direct fp32 / fp64 evaluation.  Result on the seeded weights:
per-layer relative error 2.6e-6 (direct fp32: 4.2e-7); pixel L-inf of the whole generator 7.3e-7 against fp64 (direct fp32:
3.1e-7); 1 of 110 592 uint8 values differs.  The north-star tolerance is 1e-3.

    python tools/wino_f4x4_accuracy.py
"""
import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0,'/root/repo')
from oracle import models_ref, datagen_ref
from wav2lip_amd import synthetic as synth, models
torch.set_num_threads(8)
# F(4x4,3x3) matrices (Lavin & Gray), points 0, +-1, +-2, inf
AT = torch.tensor([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]], dtype=torch.float64)
G = torch.tensor([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]], dtype=torch.float64)
BT = torch.tensor([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]], dtype=torch.float64)
def wino43(x, w, dt=torch.float32):
    """3x3 s1 p1 conv via F(4x4,3x3), transforms and products in dtype dt (weights transformed in fp64, rounded once)"""
    N,C,H,W = x.shape; K = w.shape[0]
    TH, TW = (H+3)//4, (W+3)//4
    xp = F.pad(x, (1, 4*TW - W + 1, 1, 4*TH - H + 1))
    # tiles 6x6 stride 4
    t = xp.unfold(2,6,4).unfold(3,6,4)            # N,C,TH,TW,6,6
    Bt = BT.to(dt); At = AT.to(dt)
    V = torch.einsum('ij,nctujk,lk->nctuil', Bt, t.to(dt), Bt)      # N,C,TH,TW,6,6
    U = torch.einsum('ij,kcjl,ml->kcim', G, w.double(), G).to(dt)   # K,C,6,6
    M = torch.einsum('nctuil,kcil->nktuil', V, U)
    Y = torch.einsum('ij,nktujl,ml->nktuim', At, M, At)             # N,K,TH,TW,4,4
    Y = Y.permute(0,1,2,4,3,5).reshape(N,K,4*TH,4*TW)[:,:,:H,:W]
    return Y
# quick self-check
x = torch.randn(2,16,13,10); w = torch.randn(8,16,3,3)
ref = F.conv2d(x.double(), w.double(), padding=1)
print("F(4,3) fp64 err", (wino43(x.double(), w, torch.float64)-ref).abs().max().item())
print("F(4,3) fp32 rel err", ((wino43(x, w)-ref).abs().max()/ref.abs().max()).item(), " direct fp32 rel err", ((F.conv2d(x,w,padding=1)-ref).abs().max()/ref.abs().max()).item())
# whole generator with F(4,3) on every 3x3 s1 p1 layer with cin%8==0 (and spatial >= 12)
G_ = models.Wav2Lip()
sd = synth.synthetic_state_dict({k: tuple(v.shape) for k, v in G_.state_dict().items()}, seed=0)
faces = synth.face_crops_u8(4, seed=1); mels = synth.mel_windows(4, seed=1)
img, mel = datagen_ref.to_model_inputs(*datagen_ref.datagen_batch(faces, mels))
img, mel = torch.from_numpy(img), torch.from_numpy(mel)
ref32 = models_ref.wav2lip_forward(sd, mel, img)
sd64 = {k:(v.double() if v.is_floating_point() else v) for k,v in sd.items()}
ref64 = models_ref.wav2lip_forward(sd64, mel.double(), img.double())
orig = F.conv2d
used=[0]
def conv_sub(x, w, b=None, stride=1, padding=0, **kw):
    s = stride if isinstance(stride,int) else stride[0]; s2 = stride if isinstance(stride,int) else stride[1]
    p = padding if isinstance(padding,int) else padding[0]
    if w.shape[2]==3 and w.shape[3]==3 and s==1 and s2==1 and p==1 and w.shape[1]%8==0 and x.shape[2]>=12 and x.dtype==torch.float32:
        used[0]+=1
        y = wino43(x, w)
        return y + b.view(1,-1,1,1) if b is not None else y
    return orig(x, w, b, stride=stride, padding=padding, **kw)
F.conv2d = conv_sub
out43 = models_ref.wav2lip_forward(sd, mel, img)
F.conv2d = orig
print("layers on F(4,3):", used[0])
print("pixel Linf: direct fp32 vs fp64 %.3e | F(4,3) fp32 vs fp64 %.3e | F(4,3) vs direct fp32 %.3e" % ((ref32.double()-ref64).abs().max().item(), (out43.double()-ref64).abs().max().item(), (out43-ref32).abs().max().item()))
u8a = datagen_ref.frames_to_u8(ref32.numpy()); u8b = datagen_ref.frames_to_u8(out43.numpy())
print("uint8 mismatches:", int((u8a!=u8b).sum()), "of", u8a.size)
