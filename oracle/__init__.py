"""CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package, and only as the
checker / reported baseline.  The product (wav2lip_amd/) never imports it and has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * models_ref  — PINNED: bit-for-bit checked against the reference's own models/ (imported from /root/reference
                  in the build container) by tests/golden/make_golden.py; its outputs are committed in tests/golden/.
  * datagen_ref — restates pure-numpy arithmetic of inference.py; no reference test exists; checked by properties.
  * audio_ref   — PARITY UNPINNED at the librosa boundary: librosa 0.7.0 is a third-party dependency absent from
                  /root/reference (requirements.txt:1) and from this image; the restatement follows its published
                  algorithm and is cross-checked against torch.stft and known-answer properties only.
"""
