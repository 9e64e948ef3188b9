#!/usr/bin/env python3
"""Throughput of the guided (external-evaluator) MCTS path on one MI355X: 11x11 Copenhagen, S simulations per root.

  engine-only   nnet = a constant float32 prior tensor + zero values already resident in HBM: times k_gmcts_step /
                k_gmcts_leaves alone (what the library adds per network call)
  torch net     a small random-init conv policy/value network evaluated with PyTorch-ROCm on the same stream of leaves;
                inputs and outputs stay in HBM (device pointers)
Prints one JSON line per mode."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=65536)
    ap.add_argument("--sims", type=int, default=64)
    ap.add_argument("--edges-per-node", type=int, default=192)
    ap.add_argument("--channels", type=int, default=32)
    args = ap.parse_args()
    import torch
    from alphazeroforhnefatafl_amd import BatchedGameLogic, GuidedMCTS, MCTSArgs, boards, rules
    dev = torch.device("cuda:0")
    n, side = args.games, 11
    lg = BatchedGameLogic(rules.COPENHAGEN, side)
    A = lg.action_size
    bt = torch.empty((n, side, side), dtype=torch.uint8, device=dev)
    st = torch.empty(n, dtype=torch.uint8, device=dev)
    wt = torch.empty(n, dtype=torch.uint8, device=dev)
    bufs = (bt.data_ptr(), st.data_ptr(), wt.data_ptr())

    class Const:
        def __init__(self):
            self.p = torch.rand((n, A), dtype=torch.float32, device=dev)
            self.v = torch.zeros(n, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()

        def predict_batch(self, *_):
            return self.p.data_ptr(), self.v.data_ptr()

    class Conv:
        def __init__(self):
            torch.manual_seed(0)
            c = args.channels
            self.body = torch.nn.Sequential(torch.nn.Conv2d(2, c, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(c, c, 3, padding=1), torch.nn.ReLU(),
                                            torch.nn.Conv2d(c, 20, 1)).to(dev).eval().half()
            self.vhead = torch.nn.Linear(20 * side * side, 1).to(dev).eval().half()
            self.keep = None

        def predict_batch(self, *_):
            with torch.no_grad():
                x = torch.stack([bt.half() / 35.0, (st.half() / 8.0)[:, None, None].expand(-1, side, side)], 1)
                y = self.body(x)                                   # [n, 20, 11, 11]: one logit per (tile, slot) = the action layout
                logits = y.permute(0, 2, 3, 1).reshape(n, A).float()
                p = torch.softmax(logits, 1).contiguous()
                v = torch.tanh(self.vhead(y.reshape(n, -1))).float().reshape(n).contiguous()
            torch.cuda.synchronize()
            self.keep = (p, v)
            return p.data_ptr(), v.data_ptr()

    for name, net in (("engine_only_constant_priors", Const()), ("torch_conv_fp16", Conv())):
        b = lg.new_batch(n, boards.COPENHAGEN)
        m = GuidedMCTS(b, net, MCTSArgs(numMCTSSims=args.sims, cpuct=1.0), edges_per_node=args.edges_per_node, device=True, buffers=bufs)
        bt.zero_(); st.zero_(); wt.zero_()
        for _ in range(3):
            net.predict_batch()                                  # MIOpen / hipBLASLt kernel selection happens on the first calls
        m.search_all()                                           # first search: allocates the arena (26 GB at the default size)
        b.reset_fen(boards.COPENHAGEN, rules.COPENHAGEN.starting_side)
        m.rounds = 0
        lg.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.search_all()
        lg.sync(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        s = b.gmcts_stats()
        print(json.dumps({"mode": name, "games": n, "sims_per_root": args.sims, "seconds": round(dt, 3), "sims_per_sec": s.sims / dt, "rounds": m.rounds,
                          "ms_per_round": 1e3 * dt / max(1, m.rounds), "predicts": s.predicts, "terminal_hits": s.terminal_hits, "faults": s.faults,
                          "mean_select_depth": s.select_depth_sum / max(1, s.sims)}))
        b.close()


if __name__ == "__main__":
    main()
