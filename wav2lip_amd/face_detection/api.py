"""`FaceAlignment` (face_detection/api.py:41-77) over the HIP S3FD detector: `get_detections_for_batch(images)` takes the
reference's numpy uint8 BGR batch [B,H,W,3] and returns one (x1, y1, x2, y2) int tuple or None per image."""
import os
from enum import Enum

import numpy as np
import torch

from .s3fd import nms, nms_batch, s3fd


class LandmarksType(Enum):
    _2D = 1
    _2halfD = 2
    _3D = 3


class NetworkSize(Enum):
    LARGE = 4

    def __int__(self):
        return self.value


DEFAULT_WEIGHTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "s3fd.pth")   # detection/sfd/sfd_detector.py:17


class FaceAlignment:
    def __init__(self, landmarks_type=LandmarksType._2D, network_size=NetworkSize.LARGE, device="cuda", flip_input=False,
                 face_detector="sfd", verbose=False, path_to_detector=None, state_dict=None):
        if face_detector != "sfd":
            raise NotImplementedError("only the 'sfd' detector of the reference is mirrored")
        if not torch.cuda.is_available() or "cuda" not in str(device):
            raise RuntimeError("wav2lip_amd.face_detection: needs a HIP device (no CPU path)")
        self.device, self.flip_input, self.landmarks_type, self.verbose = device, flip_input, landmarks_type, verbose
        net = s3fd()
        if state_dict is None:
            path = path_to_detector or DEFAULT_WEIGHTS
            if not os.path.isfile(path):
                raise FileNotFoundError("S3FD weights %s not found (the reference downloads s3fd-619a316812.pth, "
                                        "sfd_detector.py:10-24; there is no network here): pass path_to_detector or "
                                        "state_dict" % path)
            state_dict = torch.load(path, map_location="cpu")
        net.load_state_dict(state_dict)
        self.face_detector = net.to(device).eval()

    def detect_from_batch(self, images_bgr):
        """sfd_detector.py:39-45 semantics per image: candidates (score > 0.05), NMS 0.3, keep score > 0.5.
        images: numpy uint8 [B,H,W,3] BGR (or a torch uint8 tensor already on the device)"""
        if isinstance(images_bgr, np.ndarray):
            images_bgr = torch.from_numpy(np.ascontiguousarray(images_bgr)).to(self.device)
        with torch.no_grad():
            levels = self.face_detector.dense_boxes(images_bgr)
            table = torch.cat(levels, dim=1).contiguous()       # [B, sum FH*FW, 5]
            keep, counts = nms_batch(table, 0.05, 0.3)          # gate + NMS on the device: only the survivors cross PCIe
            counts_h = counts.cpu().tolist()
            out = []
            for b, n in enumerate(counts_h):
                d = table[b].index_select(0, keep[b, :n].long()).cpu().numpy()
                out.append([x for x in d if x[-1] > 0.5])
        return out

    def get_detections_for_batch(self, images):
        """api.py:61-77 (the BGR->RGB flip of :62 happens inside the device pack kernel)"""
        def first_rect(dets):
            # the best-scoring detection, clipped at the image origin and truncated to ints; None when nothing survived NMS
            if len(dets) == 0:
                return None
            box = np.maximum(np.asarray(dets[0][:4]), 0)
            return tuple(int(v) for v in box)

        return [first_rect(dets) for dets in self.detect_from_batch(images)]
