#!/usr/bin/env python
"""S3FD face-detection throughput on one MI355X: batches of uint8 frames resident in HBM ->
dense decoded boxes on device -> candidate gate + greedy NMS on device (w2l_s3fd_nms), survivors to the host.
    python tools/s3fd_bench.py [--batch 16] [--height 480] [--width 640] [--steps 5]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    from wav2lip_amd.synthetic import s3fd_state_dict as seeded_state_dict
    from wav2lip_amd import face_detection as fd
    fa = fd.FaceAlignment(fd.LandmarksType._2D, device="cuda", state_dict=seeded_state_dict())
    net = fa.face_detector
    B, H, W = args.batch, args.height, args.width
    img = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
    net.dense_boxes(img)                     # builds the graph, autotunes every conv launch
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        net.dense_boxes(img)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    g = net._graph(B, H, W, img.device)
    macs = sum(op[1].macs() for op in g.ops if op[0] == "convs")
    t0 = time.perf_counter()
    fa.get_detections_for_batch(img)
    host_ms = (time.perf_counter() - t0) * 1e3 - ms
    print(json.dumps({"what": "S3FD detector, fp32, network + decode on device", "batch": B, "frame": [H, W],
                      "ms_per_batch": round(ms, 3), "frames_per_s": round(B / ms * 1e3, 1),
                      "gflop_per_frame": round(2 * macs / B / 1e9, 1), "tflops": round(2 * macs / ms / 1e9, 1),
                      "gate_nms_ms_per_batch": round(host_ms, 1), "gate_nms": "device (w2l_s3fd_nms) + the copy of the survivors; round 4: host numpy, 3 570 ms"}))


if __name__ == "__main__":
    main()
