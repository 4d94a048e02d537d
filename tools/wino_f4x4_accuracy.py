"""CPU experiment behind conv_wino4.hip: Winograd F(4x4,3x3) in fp32 (points 0, +-1, +-2, inf; weights transformed in fp64 and
rounded once) on every eligible 3x3 layer of the generator vs the direct fp32 / fp64 evaluation.  Result on the seeded weights:
per-layer relative error 2.6e-6 (direct fp32: 4.2e-7); pixel L-inf of the whole generator 7.3e-7 against fp64 (direct fp32:
3.1e-7); 1 of 110 592 uint8 values differs.  The north-star tolerance is 1e-3.

    python tools/wino_f4x4_accuracy.py [--split]

--split (DESIGN 8.0, next round's kernel): the 36 position GEMMs of F(4x4) with both operands - transformed input, transformed
weights - as three bf16 pieces and the six piece products with i + j <= 2 accumulated in ONE fp32 accumulator per 16-channel chunk,
smallest first (what v_mfma_f32_32x32x16_bf16 would do).  Result on the seeded weights: per-layer relative error 1.8e-6 (fp32
products 4.3e-6), pixel L-inf of the whole generator 5.7e-7 against fp64 (fp32 products 7.3e-7), the same single uint8 value
differs: the split costs no accuracy on top of Winograd's.
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
def split3(t):
    """fp32 tensor -> three bf16-valued fp32 tensors that sum to it exactly (RNE each, of what the pieces before left)"""
    h = t.bfloat16().float(); r = t - h
    m = r.bfloat16().float()
    return h, m, (r - m).bfloat16().float()
PAIRS = [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]      # smallest piece products first
def split_products(V, U, chunk=16):
    """M[n,k,t,u,i,l] = sum_c V[n,c,t,u,i,l] U[k,c,i,l] from three-piece operands: per 16-channel chunk six exact piece products
    (fp64 here, exact in the matrix core's wide inner sum) added one after the other into an fp32 accumulator"""
    pv, pu = split3(V), split3(U)
    N, C = V.shape[:2]; K = U.shape[0]
    acc = torch.zeros((N, K) + tuple(V.shape[2:]), dtype=torch.float32)
    for c in range(0, C, chunk):
        for a, b in PAIRS:
            acc = acc + torch.einsum('nctuil,kcil->nktuil', pv[a][:, c:c + chunk].double(), pu[b][:, c:c + chunk].double()).float()
    return acc
def wino43(x, w, dt=torch.float32, split=False):
    """3x3 s1 p1 conv via F(4x4,3x3), transforms and products in dtype dt (weights transformed in fp64, rounded once);
    split: the position products from three-piece bf16 operands (dt must be fp32)"""
    N,C,H,W = x.shape; K = w.shape[0]
    TH, TW = (H+3)//4, (W+3)//4
    xp = F.pad(x, (1, 4*TW - W + 1, 1, 4*TH - H + 1))
    # tiles 6x6 stride 4
    t = xp.unfold(2,6,4).unfold(3,6,4)            # N,C,TH,TW,6,6
    Bt = BT.to(dt); At = AT.to(dt)
    V = torch.einsum('ij,nctujk,lk->nctuil', Bt, t.to(dt), Bt)      # N,C,TH,TW,6,6
    U = torch.einsum('ij,kcjl,ml->kcim', G, w.double(), G).to(dt)   # K,C,6,6
    M = split_products(V, U) if split else torch.einsum('nctuil,kcil->nktuil', V, U)
    Y = torch.einsum('ij,nktujl,ml->nktuim', At, M, At)             # N,K,TH,TW,4,4
    Y = Y.permute(0,1,2,4,3,5).reshape(N,K,4*TH,4*TW)[:,:,:H,:W]
    return Y
# quick self-check
x = torch.randn(2,16,13,10); w = torch.randn(8,16,3,3)
ref = F.conv2d(x.double(), w.double(), padding=1)
print("F(4,3) fp64 err", (wino43(x.double(), w, torch.float64)-ref).abs().max().item())
print("F(4,3) fp32 rel err", ((wino43(x, w)-ref).abs().max()/ref.abs().max()).item(), " direct fp32 rel err", ((F.conv2d(x,w,padding=1)-ref).abs().max()/ref.abs().max()).item())
SPLIT = "--split" in sys.argv
if SPLIT:
    print("F(4,3) split-bf16 products rel err", ((wino43(x, w, split=True)-ref).abs().max()/ref.abs().max()).item())
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
        y = wino43(x, w, split=SPLIT_NOW[0])
        return y + b.view(1,-1,1,1) if b is not None else y
    return orig(x, w, b, stride=stride, padding=padding, **kw)
SPLIT_NOW = [False]
F.conv2d = conv_sub
out43 = models_ref.wav2lip_forward(sd, mel, img)
if SPLIT:
    SPLIT_NOW[0] = True
    n43 = used[0]
    out43s = models_ref.wav2lip_forward(sd, mel, img)
    used[0] = n43
F.conv2d = orig
print("layers on F(4,3):", used[0])
print("pixel Linf: direct fp32 vs fp64 %.3e | F(4,3) fp32 vs fp64 %.3e | F(4,3) vs direct fp32 %.3e" % ((ref32.double()-ref64).abs().max().item(), (out43.double()-ref64).abs().max().item(), (out43-ref32).abs().max().item()))
u8a = datagen_ref.frames_to_u8(ref32.numpy()); u8b = datagen_ref.frames_to_u8(out43.numpy())
print("uint8 mismatches:", int((u8a!=u8b).sum()), "of", u8a.size)
if SPLIT:
    print("pixel Linf with split-bf16 position products: vs fp64 %.3e | vs F(4,3) fp32 products %.3e" %
          ((out43s.double()-ref64).abs().max().item(), (out43s-out43).abs().max().item()))
    u8c = datagen_ref.frames_to_u8(out43s.numpy())
    print("uint8 mismatches against direct fp32:", int((u8a!=u8c).sum()), "of", u8a.size)
