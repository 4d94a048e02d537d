"""`face_detection` with the reference's surface (face_detection/__init__.py, api.py): `FaceAlignment`, `LandmarksType`,
`NetworkSize`; the S3FD detector behind it runs on the HIP path (wav2lip_amd/face_detection/s3fd.py)."""
from .api import FaceAlignment, LandmarksType, NetworkSize  # noqa: F401
from .s3fd import s3fd  # noqa: F401
