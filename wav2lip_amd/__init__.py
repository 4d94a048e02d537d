"""wav2lip_amd — MI355X (gfx950) engine for the Wav2Lip hot path behind the reference's Python API:
`models.Wav2Lip / SyncNet_color / Wav2Lip_disc_qual`, `audio.load_wav / melspectrogram`, `inference.datagen`
and the per-batch loop, `hparams`.  Compute lives in csrc/ (hand-written HIP, C ABI in include/w2l_hip.h)."""
__version__ = "0.1.0"
