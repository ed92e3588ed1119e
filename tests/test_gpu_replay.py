"""GPU parity: replay ring write / index draw / gather vs the oracle and the reference golden (G5)."""
import os

import numpy as np
import pytest
import torch

from avddpg_amd import vec
from oracle import philox as ophilox
from oracle import replay as oreplay
from tests.gpu_util import need_gpu, t

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def test_ring_and_gather_match_reference_golden():
    """Reference rows (f64) -> device ring (f32): ring contents after wrap, counter, and the batches
    gathered at the reference's own sampled indices are the f32 roundings of the reference's."""
    need_gpu()
    g = np.load(os.path.join(G, "g5_replay.npz"))
    for key in g["keys"]:
        cap = int(key.split("_")[0][3:])
        B = int(key.split("_")[1][1:])
        seed = int(key.split("seed")[1])
        rb = vec.VecReplay(1, cap, B, 4, 1)
        rows = g[key + "__rows"]
        for k in range(len(rows)):
            rb.add(t(rows[k:k + 1, 0:4]), t(rows[k:k + 1, 4:5]), t(rows[k:k + 1, 5]), t(rows[k:k + 1, 6:10]), 4)
        assert rb.buffer_counter == int(g[key + "__counter"])
        if cap == 8:
            ring = rb.ring.cpu().numpy()[0]
            assert np.array_equal(ring[:, 0:4], g[key + "__ring_s"].astype(np.float32))
            assert np.array_equal(ring[:, 4:5], g[key + "__ring_a"].astype(np.float32))
            assert np.array_equal(ring[:, 5:6], g[key + "__ring_r"].astype(np.float32))
            assert np.array_equal(ring[:, 6:10], g[key + "__ring_s2"].astype(np.float32))
        np.random.seed(seed)
        s, a, r, s2 = rb.sample()  # host mode: np.random.choice(range, B), the reference's own draw
        assert np.array_equal(rb.idx.cpu().numpy()[0].astype(np.int64), g[key + "__idx"])  # bit-exact integers
        assert np.array_equal(s.cpu().numpy()[0], g[key + "__s"].astype(np.float32))
        assert np.array_equal(a.cpu().numpy()[0], g[key + "__a"].astype(np.float32))
        assert np.array_equal(r.cpu().numpy()[0], g[key + "__r"].astype(np.float32).ravel())
        assert np.array_equal(s2.cpu().numpy()[0], g[key + "__s2"].astype(np.float32))


@pytest.mark.parametrize("S,x_stride", [(4, 4), (3, 4)])
def test_many_agents_ring_wrap_and_gather_vs_oracle(S, x_stride):
    """4096x5 agents (BASELINE config #2 shape), small capacity so the ring wraps; Model A (S=3) reads
    only the first 3 columns of the 4-wide state rows."""
    need_gpu()
    n, cap, B, A = 4096 * 5, 7, 64, 1
    rb = vec.VecReplay(n, cap, B, S, A, rng="device", seed=11)
    refs = [oreplay.RefReplayBuffer(cap, B, S, A, dtype=np.float32) for _ in range(3)]
    pick = [0, 777, n - 1]
    rs = np.random.RandomState(0)
    for k in range(17):
        sp = rs.normal(size=(n, x_stride)).astype(np.float32)
        sn = rs.normal(size=(n, x_stride)).astype(np.float32)
        ac = rs.normal(size=(n, A)).astype(np.float32)
        rw = rs.normal(size=n).astype(np.float32)
        assert oreplay.ring_index(rb.buffer_counter, cap) == k % cap
        rb.add(t(sp), t(ac), t(rw), t(sn), x_stride)
        for ref, ag in zip(refs, pick):
            ref.add((sp[ag, :S], ac[ag], rw[ag], sn[ag, :S]))
    assert rb.sample_range() == oreplay.sample_range(17, cap) == cap
    ring = rb.ring.cpu().numpy()
    for ref, ag in zip(refs, pick):
        assert np.array_equal(ring[ag, :, :S], ref.state_buffer)
        assert np.array_equal(ring[ag, :, S:S + A], ref.action_buffer)
        assert np.array_equal(ring[ag, :, S + A], ref.reward_buffer[:, 0])
        assert np.array_equal(ring[ag, :, S + A + 1:], ref.next_state_buffer)
    s, a, r, s2 = rb.sample()
    idx = rb.idx.cpu().numpy()
    # device indices are bit-exact with the oracle's restatement of the Philox draw
    assert np.array_equal(idx, ophilox.replay_indices(n, B, cap, seed=11, counter=0))
    assert idx.min() >= 0 and idx.max() < cap
    ar = np.arange(n)[:, None]
    assert np.array_equal(s.cpu().numpy(), ring[ar, idx][:, :, :S])
    assert np.array_equal(a.cpu().numpy(), ring[ar, idx][:, :, S:S + A])
    assert np.array_equal(r.cpu().numpy(), ring[ar, idx][:, :, S + A])
    assert np.array_equal(s2.cpu().numpy(), ring[ar, idx][:, :, S + A + 1:])
    # uniformity of the device draw over a larger range
    rb2 = vec.VecReplay(64, 100000, 64, 4, 1, rng="device", seed=5)
    rb2.buffer_counter = 100000
    counts = np.zeros(10)
    for _ in range(50):
        counts += np.histogram(rb2.draw_indices().cpu().numpy(), bins=10, range=(0, 100000))[0]
    assert np.all(np.abs(counts / counts.sum() - 0.1) < 0.005)


def test_normal_kernel_matches_oracle_philox():
    need_gpu()
    from avddpg_amd._hip import call, ptr, stream_handle
    out = torch.empty(10000, device="cuda")
    call("avd_normal_f32", 10000, ptr(out), 0.1, 42, 7, stream_handle())
    ref = ophilox.normals(10000, seed=42, counter=7, stream=ophilox.STREAM_NORMAL) * np.float32(0.1)
    assert np.allclose(out.cpu().numpy(), ref, rtol=0, atol=2e-6)  # libm vs device log/cos: few ulp of 0.4
    assert abs(out.std().item() / 0.1 - 1) < 0.03


def test_full_capacity_ring_of_20480_agents_wraps_at_100000_rows():
    """BASELINE configs[1] at the reference's buffer_size (src/config.py:107): 20480 rings x 100000 rows x 10 floats = 82 GB
    in HBM. The write index wraps at capacity (replaybuffer.py:40), the sample range saturates at it (:52), indices stay in
    range and the gather returns the rows it indexed."""
    need_gpu()
    n, cap, B, S, A = 4096 * 5, 100000, 64, 4, 1
    if torch.cuda.mem_get_info()[0] < 100 * 2**30:
        pytest.skip("needs ~85 GB of free HBM")
    rb = vec.VecReplay(n, cap, B, S, A, rng="device", seed=3)
    assert rb.ring.numel() * 4 == n * cap * (2 * S + A + 1) * 4 == 81_920_000_000
    rb.buffer_counter = cap - 2  # as after cap - 2 adds
    rows = []
    for k in range(5):  # slots cap-2, cap-1, 0, 1, 2
        sp = torch.full((n, S), float(10 + k), device="cuda") + torch.arange(n, device="cuda").view(n, 1) * 1e-3
        ac = torch.full((n, A), float(-k), device="cuda")
        rw = torch.full((n,), 0.5 * k, device="cuda")
        sn = sp + 0.25
        assert oreplay.ring_index(rb.buffer_counter, cap) == (cap - 2 + k) % cap
        rb.add(sp, ac, rw, sn, S)
        rows.append((sp, ac, rw, sn))
    assert rb.buffer_counter == cap + 3 and rb.sample_range() == oreplay.sample_range(cap + 3, cap) == cap
    for k, slot in enumerate((cap - 2, cap - 1, 0, 1, 2)):
        sp, ac, rw, sn = rows[k]
        for ag in (0, 12345, n - 1):
            row = rb.ring[ag, slot]
            assert torch.equal(row[:S], sp[ag]) and torch.equal(row[S:S + A], ac[ag]) and row[S + A] == rw[ag]
            assert torch.equal(row[S + A + 1:], sn[ag])
    assert float(rb.ring[77, 3].abs().sum()) == 0.0  # untouched slots stay as allocated
    s, a, r, s2 = rb.sample()
    idx = rb.idx
    assert int(idx.min()) >= 0 and int(idx.max()) < cap
    assert np.array_equal(idx[:64].cpu().numpy(), ophilox.replay_indices(n, B, cap, seed=3, counter=0)[:64])
    for ag in (0, 9999, n - 1):
        got = rb.ring[ag][idx[ag].long()]
        assert torch.equal(s[ag], got[:, :S]) and torch.equal(r[ag], got[:, S + A]) and torch.equal(s2[ag], got[:, S + A + 1:])
