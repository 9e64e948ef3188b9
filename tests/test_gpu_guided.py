"""Guided MCTS through the C-ABI on a real MI355X: k_gmcts_step / k_gmcts_leaves against the vectors of the reference's
mcts.py (stub network), against the literal oracle on a mixed batch, and host-pointer vs device-pointer (torch) plumbing."""
import ctypes as C
import json
import os

import pytest

from alphazeroforhnefatafl_amd import abi
from oracle import oracle as orc
from tests import guided_util as gu
from tests import parity_util as pu
from tests.stub_net import matrix_bytes_of, stub_predict

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "mcts_golden.json")) as f:
    GOLD = json.load(f)


def run_device_guided(batch, n, side_len, n_sims, c_puct, salts, edges_per_node=256, max_children=600):
    A = abi.action_size(side_len)
    batch.gmcts_begin(n_sims, edges_per_node)
    w = batch.gmcts_step(None, None, c_puct, n_sims)
    rounds = 0
    while w:
        boards, sides, waiting = batch.gmcts_leaves()
        assert sum(waiting) == w
        pri, val = gu.stub_batch(boards, sides, waiting, n, side_len, A, salts)
        w = batch.gmcts_step(pri, val, c_puct, n_sims)
        rounds += 1
    kids, cnt = batch.gmcts_root_children(max_children)
    out = [[(kids[g * max_children + i].action, kids[g * max_children + i].visits, float(kids[g * max_children + i].q).hex()) for i in range(cnt[g])]
           for g in range(n)]
    return out, batch.gmcts_stats(), rounds


@pytest.mark.parametrize("case", GOLD["guided_cases"], ids=[c["name"] for c in GOLD["guided_cases"]])
def test_kernels_match_reference_mcts_py(case):
    from alphazeroforhnefatafl_amd import BatchedGameLogic
    n, wb = case["side_len"], case["word_bits"]
    lg = BatchedGameLogic(abi.rules.BY_NAME[case["rules"]], n, wb)
    b = lg.new_batch(3)
    st = abi.TaflState.from_buffer_copy(bytes.fromhex(case["state_hex"]))
    b.upload((abi.TaflState * 3)(st, st, st))
    kids, stats, rounds = run_device_guided(b, 3, n, case["n_sims"], case["cpuct"], [case["salt"]] * 3)
    for g in range(3):
        assert [[a, v, q] for a, v, q in kids[g]] == case["root_children"]
    assert stats.sims == 3 * case["n_sims"] and stats.predicts == 3 * case["predict_calls"] and stats.faults == 0
    # dense getters: visit counts and the probs of mcts.py:48-53
    A = abi.action_size(n)
    visits, probs = b.gmcts_root_visits(), b.gmcts_policy(1.0)
    want = {a: v for a, v, _ in case["root_children"]}
    assert {a: visits[a] for a in range(A) if visits[a]} == want
    assert [(a, float(probs[a]).hex()) for a in range(A) if probs[a] != 0] == [tuple(x) for x in case["probs_temp1_nonzero"]]
    hot = b.gmcts_policy(0.0)
    best = max(want.values())
    assert [a for a in range(A) if hot[a] != 0] == [min(a for a, v in want.items() if v == best)]


@pytest.mark.parametrize("cfg", ["brandubh7", "copenhagen11", "copenhagen13", "tablut9"])
def test_kernels_match_oracle_on_a_batch(cfg):
    from alphazeroforhnefatafl_amd import BatchedGameLogic
    rules, fen, wb = pu.CONFIGS[cfg]
    n = abi.fen_side_len(fen)
    G, S = 96, 48
    olg = orc.GameLogic(rules, n)
    base = orc.GameState(fen, rules.starting_side, wb)
    states = (abi.TaflState * G)(*[olg.random_advance(base, 33, g, (g * 5) % 70).to_abi() for g in range(G)])
    salts = [(7 * g + 3) % 256 for g in range(G)]
    lg = BatchedGameLogic(rules, n, wb)
    b = lg.new_batch(G)
    b.upload(states)
    kids, stats, _ = run_device_guided(b, G, n, S, 1.5, salts)
    A = abi.action_size(n)
    for g in range(0, G, 5):
        gs = orc.GameState.from_abi(states[g], wb)
        ok, _, _, _ = olg.gmcts(gs, S, 1.5, lambda s, g=g: stub_predict(matrix_bytes_of(s.board_to_matrix()), int(s.side_to_play), A, salts[g]), wb)
        assert [(a, v, float(q).hex()) for (_p, a, v, q) in ok] == kids[g], f"game {g}"
    assert stats.sims == G * S and stats.faults == 0


def test_device_pointer_plumbing_with_a_torch_network():
    """GuidedMCTS with a small random-init torch policy/value net, inputs and outputs staying on the GPU; the same network
    evaluated through host buffers must give bit-identical search results."""
    import torch
    from alphazeroforhnefatafl_amd import BatchedGameLogic, GuidedMCTS, MCTSArgs, boards, rules
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    n, side = 64, 11
    lg = BatchedGameLogic(rules.COPENHAGEN, side)
    A = lg.action_size
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Flatten(), torch.nn.Linear(8 * side * side, A + 1)).to(dev).eval()

    class Net:
        def forward(self, boards_t, sides_t):
            with torch.no_grad():
                x = torch.stack([boards_t.float() / 35.0, (sides_t.float() / 8.0)[:, None, None].expand(-1, side, side)], 1)
                y = net(x)
                return torch.softmax(y[:, :A], 1).contiguous(), torch.tanh(y[:, A]).contiguous()

    class DeviceNet(Net):
        def __init__(self):
            self.boards = torch.empty((n, side, side), dtype=torch.uint8, device=dev)
            self.sides = torch.empty(n, dtype=torch.uint8, device=dev)
            self.waiting = torch.empty(n, dtype=torch.uint8, device=dev)
            self.keep = None

        def predict_batch(self, *_ptrs):
            p, v = self.forward(self.boards, self.sides)
            torch.cuda.synchronize()
            self.keep = (p, v)
            return p.data_ptr(), v.data_ptr()

    class HostNet(Net):
        def predict_batch(self, boards, sides, waiting):
            bt = torch.frombuffer(bytearray(bytes(boards)), dtype=torch.uint8).reshape(n, side, side).to(dev)
            st = torch.frombuffer(bytearray(bytes(sides)), dtype=torch.uint8).to(dev)
            p, v = self.forward(bt, st)
            p, v = p.cpu().numpy(), v.cpu().numpy()
            self.keep = (p, v)
            return p.ctypes.data_as(C.POINTER(C.c_float)), v.ctypes.data_as(C.POINTER(C.c_float))

    args = MCTSArgs(numMCTSSims=24, cpuct=1.0)
    b1 = lg.new_batch(n, boards.COPENHAGEN)
    b1.random_advance(5, (C.c_uint32 * n)(*[g % 30 for g in range(n)]))
    states = b1.download()
    dn = DeviceNet()
    m1 = GuidedMCTS(b1, dn, args, device=True, buffers=(dn.boards.data_ptr(), dn.sides.data_ptr(), dn.waiting.data_ptr()))
    k1, c1 = m1.root_children()
    b2 = lg.new_batch(n)
    b2.upload(states)
    m2 = GuidedMCTS(b2, HostNet(), args)
    k2, c2 = m2.root_children()
    assert list(c1) == list(c2) and m1.rounds == m2.rounds == 24
    for g in range(n):
        for i in range(c1[g]):
            a, b_ = k1[g * 512 + i], k2[g * 512 + i]
            assert (a.action, a.visits, a.q) == (b_.action, b_.visits, b_.q)
    st = b1.gmcts_stats()
    assert st.sims == n * 24 and st.faults == 0
    # the policy tensor can be written straight into a torch tensor
    pol = torch.zeros((n, A), dtype=torch.float64, device=dev)
    b1.gmcts_policy(1.0, out_device_ptr=pol.data_ptr())
    assert torch.allclose(pol.sum(1), torch.ones(n, dtype=torch.float64, device=dev))
