#!/usr/bin/env python
"""Which torch operators run inside one training step, and what they launch (GPU box): torch.profiler over one cfg4 step of
tools/train_bench.py's setup - the tiny launches (copies, fills, elementwise) that do not come from libw2l_hip.so.
    python tools/train_step_ops.py [--precision bf16] [--cfg 4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--cfg", type=int, default=4)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--stacks", type=int, default=0, help="group the operator table by this many Python frames (call sites)")
    args = ap.parse_args()
    from wav2lip_amd import engine, models, optim, train
    engine.set_train_precision(args.precision)
    dev = torch.device("cuda")
    r = np.random.default_rng(0)

    def rand(shape, lo=0., hi=1.):
        return torch.from_numpy(r.uniform(lo, hi, shape).astype(np.float32)).to(dev)
    S = models.SyncNet_color().to(dev)
    for p in S.parameters():
        p.requires_grad = False
    G = models.Wav2Lip().to(dev)
    optG = optim.Adam([p for p in G.parameters() if p.requires_grad], lr=1e-4, betas=(0.5, 0.999))
    B, T = args.batch, 5
    gt = rand((B, 3, T, 96, 96))
    xin = torch.cat([gt.clone(), rand((B, 3, T, 96, 96))], dim=1)
    xin[:, :3, :, 48:] = 0.
    indiv, melw = rand((B, T, 1, 80, 16), -4, 4), rand((B, 1, 80, 16), -4, 4)
    if args.cfg == 5:
        D = models.Wav2Lip_disc_qual().to(dev)
        optD = optim.Adam([p for p in D.parameters() if p.requires_grad], lr=1e-4, betas=(0.5, 0.999))
        step = lambda: train.hq_train_step(G, D, S, optG, optD, xin, indiv, melw, gt, syncnet_wt=0.03, disc_wt=0.07)   # noqa: E731
    else:
        step = lambda: train.wav2lip_train_step(G, S, optG, xin, indiv, melw, gt, syncnet_wt=0.03)   # noqa: E731
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=args.stacks > 0) as prof:
        step()
        torch.cuda.synchronize()
    if args.stacks:
        rows = [e for e in prof.key_averages(group_by_stack_n=args.stacks) if e.key.startswith("aten::") or "Memcpy" in e.key
                or "Memset" in e.key]
        rows.sort(key=lambda e: -e.count)
        for e in rows[:60]:
            site = " <- ".join(f.split("/")[-1] for f in e.stack if ".py" in f and "torch/" not in f)[:200]
            print("%5d  %-34s dev %8.1f us  %s" % (e.count, e.key[:34], e.device_time_total, site))
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=70))
    print(prof.key_averages().table(sort_by="count", row_limit=40, max_name_column_width=70))


if __name__ == "__main__":
    main()
