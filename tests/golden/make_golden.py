"""Generates the committed golden fixtures from the REAL reference.  Runs only in the build container
(needs /root/reference); the fixtures travel, the reference does not.

    python tests/golden/make_golden.py

For each network: load the seeded synthetic state_dict (wav2lip_amd/synthetic.py: pure data generation) into the reference's own nn.Module
(imported from /root/reference/models), run it on seeded inputs in eval mode on CPU fp32, store the outputs.
Also checks that oracle/models_ref.py reproduces the reference bit-for-bit, and stores the audio/datagen
known-answer vectors (those are oracle outputs: the librosa boundary is unpinned, see oracle/audio_ref.py).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import audio_ref, datagen_ref, models_ref  # noqa: E402
from wav2lip_amd import synthetic as synth  # noqa: E402


def ref_models():
    sys.path.insert(0, REF)
    import importlib
    m = importlib.import_module("models")
    assert m.__file__.startswith(REF), m.__file__
    sys.path.remove(REF)
    return m


def main():
    torch.set_num_threads(8)
    rm = ref_models()
    out = {}

    # ---- generator, 4-D (B=2) and 5-D (B=1, T=2)
    G = rm.Wav2Lip().eval()
    shapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    sd = synth.synthetic_state_dict(shapes, seed=0)
    G.load_state_dict(sd)
    faces = synth.face_crops_u8(2, seed=1)
    mels = synth.mel_windows(2, seed=1)
    img, mel = datagen_ref.to_model_inputs(*datagen_ref.datagen_batch(faces, mels))
    with torch.no_grad():
        y = G(torch.from_numpy(mel), torch.from_numpy(img))
    yo = models_ref.wav2lip_forward(sd, torch.from_numpy(mel), torch.from_numpy(img))
    print("generator: ref vs oracle max|d| =", (y - yo).abs().max().item(),
          " out range", y.min().item(), y.max().item(), " std", y.std().item())
    assert torch.equal(y, yo), "oracle/models_ref.py does not reproduce the reference generator bit-for-bit"
    out["gen_out_b2"] = y.numpy()
    img5 = torch.from_numpy(img).unsqueeze(0).permute(0, 2, 1, 3, 4).contiguous()   # (1, 6, 2, 96, 96)
    mel5 = torch.from_numpy(mel).unsqueeze(0)                                        # (1, 2, 1, 80, 16)
    with torch.no_grad():
        y5 = G(mel5, img5)
    assert torch.equal(y5, models_ref.wav2lip_forward(sd, mel5, img5))
    out["gen_out_5d"] = y5.numpy()

    # ---- SyncNet (B=2)
    S = rm.SyncNet_color().eval()
    sds = synth.synthetic_state_dict({k: tuple(v.shape) for k, v in S.state_dict().items()}, seed=2)
    S.load_state_dict(sds)
    sf = torch.from_numpy(synth.sync_faces(2, seed=3))
    sm = torch.from_numpy(synth.mel_windows(2, seed=3)).unsqueeze(1)
    with torch.no_grad():
        a, v = S(sm, sf)
    ao, vo = models_ref.syncnet_forward(sds, sm, sf)
    print("syncnet: ref vs oracle", (a - ao).abs().max().item(), (v - vo).abs().max().item(),
          " cos", torch.nn.functional.cosine_similarity(a, v).tolist())
    assert torch.equal(a, ao) and torch.equal(v, vo)
    out["sync_audio_emb"] = a.numpy()
    out["sync_face_emb"] = v.numpy()

    # ---- quality discriminator (B=1, T=2)
    D = rm.Wav2Lip_disc_qual().eval()
    sdd = synth.synthetic_state_dict({k: tuple(v.shape) for k, v in D.state_dict().items()}, seed=4)
    D.load_state_dict(sdd)
    df = torch.from_numpy(synth.disc_frames(1, 2, seed=5))
    with torch.no_grad():
        p = D(df)
    po = models_ref.disc_forward(sdd, df)
    print("disc: ref vs oracle", (p - po).abs().max().item(), p.view(-1).tolist())
    assert torch.equal(p, po)
    out["disc_pred"] = p.numpy()

    # ---- audio known answers (oracle outputs; librosa boundary unpinned)
    out["mel_sine3s"] = audio_ref.melspectrogram(synth.sine_wav())
    out["mel_noise1s"] = audio_ref.melspectrogram(synth.noise_wav(16000, seed=7))
    out["mel_basis_rowsum"] = audio_ref.mel_basis().sum(axis=1)
    out["chunk_starts_T241_fps25"] = np.asarray(datagen_ref.mel_chunk_starts(241, 25.0), dtype=np.int32)
    out["chunk_starts_T1000_fps30"] = np.asarray(datagen_ref.mel_chunk_starts(1000, 30.0), dtype=np.int32)

    path = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
