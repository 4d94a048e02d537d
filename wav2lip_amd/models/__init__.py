"""Drop-in mirrors of the reference's `models` package (models/__init__.py:1-2): same class names, constructor
signatures, module trees and state-dict keys; `forward` runs on the HIP engine (libw2l_hip.so)."""
from .syncnet import SyncNet_color
from .wav2lip import Wav2Lip, Wav2Lip_disc_qual

__all__ = ["Wav2Lip", "Wav2Lip_disc_qual", "SyncNet_color"]
