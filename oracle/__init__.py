"""CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package, and only as the
checker / reported baseline.  The product (wav2lip_amd/) never imports it and has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * models_ref  — PINNED: bit-for-bit checked against the reference's own models/ (imported from /root/reference
                  in the build container) by tests/golden/make_golden.py; its outputs are committed in tests/golden/.
  * datagen_ref — PINNED: tests/golden/make_golden_datapath.py executes the reference's own inference.py (datagen, the
                  chunking, main()) with stub cv2 / librosa and freezes what it produced; test_golden_datapath.py compares.
  * audio_ref   — PINNED to the reference's audio.py (same script: audio.melspectrogram with a stub librosa == this oracle,
                  bit for bit); PARITY UNPINNED INSIDE librosa: librosa 0.7.0's stft / filters.mel / load are a third-party
                  dependency absent from /root/reference (requirements.txt:1) and from this image; restated from the
                  published algorithm, cross-checked against torch.stft and known-answer properties only.
  * resample_ref, resize_ref — PARITY UNPINNED (resampy / OpenCV absent): restatements of the published algorithms.
  * s3fd_ref, lse_ref — PINNED to the reference's modules / expressions (tests/golden/make_golden_s3fd.py).
"""
