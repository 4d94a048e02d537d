#!/usr/bin/env python
"""Generator inference at small / ragged batch sizes on one stream: ms per step, where each launch's configuration comes from
(tune table or heuristic), and - with --add-table - the same sweep again after MORE table entries were loaded on top of the
committed ones, with the L-inf distance between the two sets of output frames (a table entry only changes a layer's launch
shape and summation order, so the distance must sit at fp32 rounding level).

    python tools/batch_sweep.py --batches 1,2,3,4,5,6,7,8 [--add-table gpurun_out/x/tune_table_add.json]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,2,3,4,5,6,7,8,16,37")
    ap.add_argument("--add-table", default="", help="tune-table JSON loaded on top of the committed one for a second sweep")
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--dump-configs", default="", help="write {batch: [[launch name, config id, split-K], ...]} as the launches "
                    "resolve at the END of the run (after --add-table) - the per-plan lists of wav2lip_amd/plan_configs.json (engine.apply_plan_configs)")
    ap.add_argument("--table-batches", default="1,8,16,32,64,128,256", help="batch sizes whose launches the committed tune table "
                    "holds (tools/make_tune_table.py): listed in the dump, never applied")
    args = ap.parse_args()
    if args.dump_configs:
        os.environ["W2L_PLAN_CONFIGS"] = "0"     # dump what the TABLE resolves, not what an older dump says
    from wav2lip_amd import _lib, models
    from wav2lip_amd import synthetic as synth
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    G = models.Wav2Lip()
    G.load_state_dict(synth.synthetic_state_dict({k: tuple(v.shape) for k, v in G.state_dict().items()}, seed=0))
    G = G.to(dev).eval()
    batches = [int(b) for b in args.batches.split(",") if b]
    r = np.random.default_rng(1)
    inputs = {B: (torch.from_numpy(r.uniform(-4, 4, (B, 1, 80, 16)).astype(np.float32)).to(dev),
                  torch.from_numpy(r.uniform(0, 1, (B, 6, 96, 96)).astype(np.float32)).to(dev)) for B in batches}

    def sweep(tag):
        outs, line = {}, []
        G._graphs.clear()
        for B in batches:
            g = G.graph(B, 96, 96, dev)
            g.load_nchw(*inputs[B])
            for _ in range(5):
                g.run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                g.run()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.reps * 1e3
            outs[B] = g.output_nchw().clone()
            split = sum(1 for _, _, _, (_, k) in g.plan.resolved() if k > 1)
            line.append("B=%d %.3f (%d split-K)" % (B, ms, split))
        print("%s, %d table entries, ms per step: %s" % (tag, lib.w2l_tune_count(), "  ".join(line)), flush=True)
        return outs

    def dump(path):
        import json
        table = [int(b) for b in args.table_batches.split(",") if b]
        plans = ",\n".join('  "%d": %s' % (B, json.dumps([[n, c, k] for n, _, _, (c, k) in G.graph(B, 96, 96, dev).plan.resolved()],
                                                          separators=(",", ":"))) for B in batches)
        note = "tools/batch_sweep.py --dump-configs on %s; table batches resolve through tune_table.json" % torch.cuda.get_device_name(0)
        with open(path, "w") as fh:
            fh.write('{"generator_96": {"note": %s, "table": %s, "plans": {\n%s\n}}}\n' % (json.dumps(note), json.dumps(table), plans))
        print("wrote the resolved configurations of %d plans to %s" % (len(batches), path), flush=True)

    base = sweep("committed table")
    if args.add_table:
        n = _lib.load_tune_table(lib, args.add_table)
        print("loaded %d entries from %s" % (n, args.add_table), flush=True)
        new = sweep("with added entries")
        print("L-inf between the two sweeps' frames: " +
              "  ".join("B=%d %.2e" % (B, float((base[B] - new[B]).abs().max())) for B in batches), flush=True)
    if args.dump_configs:
        dump(args.dump_configs)


if __name__ == "__main__":
    main()
