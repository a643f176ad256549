"""Long replays of the captured step: hundreds of frames with CHANGING point counts through the graphs, every result
compared with the un-captured step on the same frame.  The round-4 memory fault (a graph memset node that corrupted K1's tables,
profiles/r05_k1_memset_node_dump.txt) needed only one replay; these runs are the guard against the next bug of that kind."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _frames(n_agents, n_frames, small):
    from heal_amd import synth
    from heal_amd.pipeline import Scene
    frames = []
    for i in range(n_frames):
        sc = Scene(n_agents, seed=70 + i, device="cuda:0", modalities=["m1"] * n_agents)
        lim = 28.0 if small else 1e9
        keep = 6000 - 450 * (i % 7) if small else None            # point counts move up AND down between consecutive frames
        sc.points = {k: p[(p[:, 0].abs() < lim) & (p[:, 1].abs() < lim)][:keep].contiguous() for k, p in sc.points.items()}
        if small:
            sc.pairwise = synth.pairwise_t_matrix(synth.agent_poses(70 + i, n_agents, r_min=3.0, r_max=10.0), 5)[None]
        frames.append(sc)
    return frames


@pytest.mark.parametrize("small", [True, False])
def test_frames_in_flight_200_frames_equal_eager(small):
    """pipeline.FramesInFlight, LiDAR PyramidFusion, 3 agents: 210 submissions over 7 distinct frames (small: +-25.6 m, where agents x cells
    == table capacity -- the layout of the r4 fault; full: BASELINE's +-102.4 m); every result bit-equal to the eager step's."""
    from heal_amd import configs
    from heal_amd.pipeline import FramesInFlight, ScenePipeline
    rng = [-25.6, -25.6, -3, 25.6, 25.6, 1] if small else None
    hypes = configs.lidar_pyramid(rng) if small else configs.lidar_pyramid(max_cav=5)
    frames = _frames(3, 7, small)
    side = torch.cuda.Stream()
    with torch.no_grad(), torch.cuda.stream(side):
        pipe = ScenePipeline(hypes, "cuda:0", seed=3)
        pipe.calibrate_cls_bias(frames[0], target_candidates=200 if small else 600)
        want = []
        for f in frames:
            b, sc = pipe.step(f)                                 # eager, un-padded clouds
            want.append((None, None) if b is None else (b.clone(), sc.clone()))
        assert sum(b is not None for b, _ in want) >= 5
        ring = FramesInFlight(pipe, frames[0], depth=2, warmup=1)
        got = []
        for k in range(210):
            r = ring.step(frames[k % 7])
            if r is not None:
                got.append(r)
        got += ring.drain()
    torch.cuda.synchronize()
    assert len(got) == 210
    for k, (b, sc) in enumerate(got):
        wb, ws = want[k % 7]
        assert (b is None) == (wb is None), k
        if b is not None:
            assert torch.equal(b, wb) and torch.equal(sc, ws), k


def test_fill_bytes_inside_a_busy_graph():
    """heal_fill_bytes captured between other nodes (torch kernels, zero fills, a second captured graph) writes exactly its pattern: ragged
    heads / tails, 0xFF and 0x00, 50 replays."""
    from heal_amd import ops
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        big = torch.zeros(197376 + 64, dtype=torch.uint8, device="cuda:0")
        x = torch.zeros(4096, device="cuda:0")
        small = torch.ones(16, dtype=torch.int32, device="cuda:0")
        views = [big[4:4 + 197376], big[16:16 + 1024], big[12:12 + 20], big[8:8 + 4]]
        for v in views:                                          # eager
            big.fill_(0x5A)
            ops.fill_bytes(v, 0xFF)
            st.synchronize()
            ref = torch.full_like(big, 0x5A)
            ref[v.data_ptr() - big.data_ptr():v.data_ptr() - big.data_ptr() + v.numel()] = 0xFF
            assert torch.equal(big, ref)
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=st):
            x.mul_(1.0)
            ops.fill_bytes(small[:1], 0)
        with torch.cuda.graph(g1, stream=st):
            z = torch.zeros(3, dtype=torch.int32, device="cuda:0")
            x.add_(1.0)
            ops.fill_bytes(small[1:2], 0)
            ops.fill_bytes(views[0], 0xFF)
            ops.fill_bytes(small[2:3], 0)
            x.add_(1.0)
        for rep in range(50):
            big.zero_()
            small.fill_(7)
            g1.replay()
            g2.replay()
            st.synchronize()
            assert int((views[0] != 0xFF).sum()) == 0, rep
            assert int(big[:4].sum()) == 0 and int(big[4 + 197376:].sum()) == 0, rep
            assert small[:4].tolist() == [0, 0, 0, 7], rep
        del z
    with pytest.raises(Exception):
        ops.fill_bytes(big[1:6], 0)                              # not 4-byte aligned / sized: refused, not rounded


def test_graph_guard_names_a_freed_address():
    """HEAL_GRAPH_GUARD (heal_amd/_capi.py): addresses handed to kernels during a capture are logged; the check passes
    while their tensors live -- those of the ordinary pool AND the ones allocated and released inside the capture (graph-private pool) --
    and names the entry point once one of them has been freed."""
    from heal_amd import _capi, ops
    st = torch.cuda.Stream()
    old = _capi._GUARD
    _capi._GUARD = True
    try:
        with torch.cuda.stream(st):
            _capi.guard_take()
            keep = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda:0")
            victim = torch.zeros(3 << 20, dtype=torch.uint8, device="cuda:0")
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                ops.fill_bytes(keep, 1)
                ops.fill_bytes(victim, 2)
                tmp = torch.empty(1 << 16, dtype=torch.uint8, device="cuda:0")     # lives in the graph's private pool only
                ops.fill_bytes(tmp, 3)
                del tmp
            log = _capi.guard_take()
            assert {a for _, a in log} >= {keep.data_ptr(), victim.data_ptr()} and len(log) == 3
            _capi.guard_check(log)                       # everything alive (the private-pool block is free but still the graph's)
            g.replay()
            st.synchronize()
            addr = victim.data_ptr()
            del victim                                   # block returns to the allocator: 'inactive'
            with pytest.raises(_capi.HealAmdError, match="heal_fill_bytes.*%x" % addr):
                _capi.guard_check(log)
            torch.cuda.empty_cache()                     # ... and now back to the driver: unmapped
            with pytest.raises(_capi.HealAmdError, match="heal_fill_bytes"):
                _capi.guard_check(log)
    finally:
        _capi._GUARD = old
        _capi.guard_take()


def test_graph_guard_passes_on_the_pipeline(monkeypatch):
    """The whole captured step under the guard: ~100 logged addresses (inputs, scratch per stream, weight layouts, outputs), all live
    across 10 replays on changing frames, two slots in flight."""
    from heal_amd import _capi, configs
    from heal_amd.pipeline import FramesInFlight, ScenePipeline
    monkeypatch.setattr(_capi, "_GUARD", True)
    frames = _frames(2, 3, True)
    side = torch.cuda.Stream()
    with torch.no_grad(), torch.cuda.stream(side):
        pipe = ScenePipeline(configs.lidar_pyramid([-25.6, -25.6, -3, 25.6, 25.6, 1]), "cuda:0", seed=3)
        pipe.calibrate_cls_bias(frames[0], target_candidates=200)
        pipe.capture(frames[0], warmup=1)
        assert len(pipe._guard) > 50
        for k in range(4):
            pipe.replay(frames[k % 3])
        ring = FramesInFlight(pipe, frames[0], depth=2, warmup=1)
        assert all(len(s.guard) > 50 for s in ring.slots)
        for k in range(6):
            ring.step(frames[k % 3])
        ring.drain()
    torch.cuda.synchronize()
