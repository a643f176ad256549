"""The N>1 path's exchange step on CPU: world_size-2 gloo processes run the same ownership /
pack / all-gather / unpack code the GPU ranks run (heal_amd/dist.py); the result must equal the
single-process stacking in scene agent order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from heal_amd import dist as hd

SHAPES = [(4, 8, 8), (6, 4, 4), (8, 2, 2)]


def _agent_maps(a):
    g = torch.Generator().manual_seed(100 + a)
    feats = [torch.randn((c, h, w), generator=g) for c, h, w in SHAPES]
    scores = [torch.rand((1, h, w), generator=g) for c, h, w in SHAPES]
    return feats, scores


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, n_agents, port, tmp, wire=None, collective="all_gather"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = hd.owned_agents(n_agents, rank, world)
    lf = [torch.stack([_agent_maps(a)[0][l] for a in mine]) if mine else torch.zeros((0,) + SHAPES[l])
          for l in range(len(SHAPES))]
    ls = [torch.stack([_agent_maps(a)[1][l] for a in mine]) if mine else torch.zeros((0, 1) + SHAPES[l][1:])
          for l in range(len(SHAPES))]
    buf = hd.pack_levels(lf, ls, hd.slots_per_rank(n_agents, world))
    if wire is not None:  # the optional half-size wire format of ShardedCollab (cast, gather, cast back)
        buf = buf.to(getattr(torch, wire))
    if collective == "gather":   # round 3 default: only rank 0 (the fusion tail) receives the shards
        gathered = hd.gather_packed(buf, world, rank)
        assert (gathered is None) == (rank != 0)
    else:
        gathered = hd.all_gather_packed(buf, world)
    if rank == 0:
        levels = hd.unpack_levels(gathered.float(), SHAPES, n_agents, world)
        torch.save([(f.clone(), s.clone()) for f, s in levels], tmp)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("collective", ["all_gather", "gather"])
@pytest.mark.parametrize("n_agents", [5, 2, 1])
def test_all_gather_of_packed_maps_world2(tmp_path, n_agents, collective):
    world = 2
    out = str(tmp_path / "levels.pt")
    mp.spawn(_worker, args=(world, n_agents, _free_port(), out, None, collective), nprocs=world, join=True)
    levels = torch.load(out)
    for l, (f, s) in enumerate(levels):
        want_f = torch.stack([_agent_maps(a)[0][l] for a in range(n_agents)])
        want_s = torch.stack([_agent_maps(a)[1][l] for a in range(n_agents)])
        assert torch.equal(f, want_f) and torch.equal(s, want_s)


def test_ownership_and_padding_rules():
    # round-robin starting at rank 1: rank 0 also runs the fusion tail, so it gets the smallest share ...
    assert hd.owned_agents(5, 0, 2) == [1, 3] and hd.owned_agents(5, 1, 2) == [0, 2, 4]
    # ... and no agent at all when there are more ranks than agents (its tail then overlaps the others' next local stage)
    assert hd.owned_agents(5, 7, 8) == [] and hd.owned_agents(5, 0, 8) == [] and hd.owned_agents(5, 1, 8) == [0]
    assert hd.owned_agents(5, 5, 8) == [4] and hd.owned_agents(8, 0, 8) == [7]
    for world in (1, 2, 3, 4, 5, 8):   # a partition, and agent a sits in slot a // world of its owner
        owned = [hd.owned_agents(5, r, world) for r in range(world)]
        assert sorted(a for o in owned for a in o) == list(range(5))
        assert all(o.index(a) == a // world for o in owned for a in o)
    assert hd.slots_per_rank(5, 2) == 3 and hd.slots_per_rank(5, 8) == 1 and hd.slots_per_rank(5, 4) == 2
    # padding slots are all-zero: score 0 -> masked out by the fusion kernel
    lf = [torch.ones((1,) + s) for s in SHAPES]
    ls = [torch.ones((1, 1) + s[1:]) for s in SHAPES]
    buf = hd.pack_levels(lf, ls, 3)
    assert buf.shape[0] == 3 and float(buf[1:].abs().sum()) == 0.0 and float(buf[0].min()) == 1.0
    # single-process gather is the identity
    g = hd.all_gather_packed(buf, 1)
    lv = hd.unpack_levels(g, SHAPES, 1, 1)
    assert all(torch.equal(f, torch.ones((1,) + s)) for (f, _), s in zip(lv, SHAPES))


def test_all_gather_fp16_wire_format_world2(tmp_path):
    """the exchange buffer may travel as fp16 (opt-in): same ownership / order, values within fp16
    rounding (2^-11 relative), zero padding slots stay exactly zero (they must still be masked by the fusion kernel)."""
    world, n_agents = 2, 3
    out = str(tmp_path / "levels16.pt")
    mp.spawn(_worker, args=(world, n_agents, _free_port(), out, "float16"), nprocs=world, join=True)
    levels = torch.load(out)
    for l, (f, s) in enumerate(levels):
        want_f = torch.stack([_agent_maps(a)[0][l] for a in range(n_agents)])
        want_s = torch.stack([_agent_maps(a)[1][l] for a in range(n_agents)])
        assert float((f - want_f).abs().max()) <= 2.0 ** -11 * float(want_f.abs().max()) * 1.01
        assert float((s - want_s).abs().max()) <= 2.0 ** -11 * 1.01
        assert not torch.equal(f, want_f)  # the cast really happened


def _map_worker(rank, world, n_agents, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = hd.owned_agents(n_agents, rank, world)
    x = torch.stack([_agent_maps(a)[0][0] for a in mine]) if mine else torch.zeros((0,) + SHAPES[0])
    gathered = hd.all_gather_packed(hd.pack_maps(x, hd.slots_per_rank(n_agents, world)), world)
    if rank == 0:
        torch.save(hd.unpack_maps(gathered, SHAPES[0], n_agents, world).clone(), tmp)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_agents", [8, 3, 1])
def test_all_gather_of_single_scale_maps_world2(tmp_path, n_agents):
    """The exchange of ShardedBaseline (HeterModelBaseline: one ego-frame map per agent, no scores)."""
    out = str(tmp_path / "maps.pt")
    mp.spawn(_map_worker, args=(2, n_agents, _free_port(), out), nprocs=2, join=True)
    want = torch.stack([_agent_maps(a)[0][0] for a in range(n_agents)])
    assert torch.equal(torch.load(out), want)


def _stripe_args():
    return {"num_blocks": 1, "depth": 2, "use_roi_mask": True, "use_RTE": False, "RTE_ratio": 0,
            "cav_att_config": {"dim": 64, "use_hetero": True, "use_RTE": False, "RTE_ratio": 0, "heads": 4, "dim_head": 16, "dropout": 0.0},
            "pwindow_att_config": {"dim": 64, "heads": [4, 2, 1], "dim_head": [16, 32, 64], "dropout": 0.0, "window_size": [2, 4, 8],
                                   "relative_pos_embedding": True, "fusion_method": "split_attn64"},
            "feed_forward": {"mlp_dim": 64, "dropout": 0.0},
            "sttf": {"voxel_size": [0.4, 0.4, 4], "downsample_rate": 4}}


def _stripe_worker(rank, world, port, tmp):
    """The exchanges of dist.ShardedBaselineStriped on CPU: all-to-all of row stripes, the V2X-ViT encoder on a stripe with split
    attention's average pool all-gathered (v2xvit_basic.SplitAttn._stripe), gather of the ego stripe."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from heal_amd.opencood.models.sub_modules.v2xvit_basic import PyramidWindowAttention, SplitAttn, V2XTEncoder
    torch.manual_seed(3)
    enc = V2XTEncoder(_stripe_args())      # training mode with zero dropout: the torch path of every operator (CPU)
    L, H, W, C = 3, 16, 8, 64
    Hs = H // world
    x = torch.randn(L, H, W, C, generator=torch.Generator().manual_seed(4))
    n_slots = hd.slots_per_rank(L, world)
    comm = hd._StripeComm(rank, world)
    mine = hd.owned_agents(L, rank, world)
    send = torch.zeros((world, n_slots, Hs, W, C))
    send[:, :len(mine)] = x[mine].reshape(len(mine), world, Hs, W, C).transpose(0, 1)
    recv = comm.all_to_all(send)
    xs = torch.stack([recv[hd.agent_owner(a, world), a // world] for a in range(L)])
    assert torch.equal(xs, x[:, rank * Hs:(rank + 1) * Hs])
    mods = [m for m in enc.modules() if isinstance(m, (PyramidWindowAttention, SplitAttn))]
    for m in mods:
        m._stripe = comm
    y = enc(xs)[0].detach().contiguous()
    for m in mods:
        m._stripe = None
    g = comm.gather0(y)
    assert (g is None) == (rank != 0)
    if rank == 0:
        torch.save((g.reshape(H, W, C), enc(x)[0].detach(), enc(xs)[0].detach()), tmp)
    dist.barrier()
    dist.destroy_process_group()


def test_v2xvit_encoder_on_row_stripes_world2(tmp_path):
    out = str(tmp_path / "stripes.pt")
    mp.spawn(_stripe_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got, want, local_only = torch.load(out)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) < 1e-5 * scale
    # without the exchange a stripe pools only its own rows: the split-attention weights differ -> the test would see it
    assert float((local_only - want[:8]).abs().max()) > 1e-4 * scale


def _window_worker(rank, world, port, tmp):
    """dist.PeerWindow on host memory: rank 0's buffer mapped into both ranks, rows written in place, `free` / `ready` fences."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_slots, per_slot = 2, 1000
    w = hd.PeerWindow(rank, world, n_slots, per_slot, torch.float32, "cpu")
    assert tuple(w.full.shape) == (world, n_slots, per_slot) and tuple(w.mine.shape) == (n_slots, per_slot)
    seen = []
    for frame in range(3):
        w.fence()                                     # free: rank 0 has read the previous frame
        w.mine.copy_(torch.full((n_slots, per_slot), float(10 * frame + rank + 1)))
        w.fence()                                     # ready
        if rank == 0:
            seen.append(w.full.clone())
    if rank == 0:
        torch.save(seen, tmp)
    dist.barrier()
    del w
    dist.destroy_process_group()


def test_peer_window_rows_written_in_place_world2(tmp_path):
    """HEAL_COLLECTIVE=p2p (dist.PeerWindow, SURVEY 8e "prefer direct P2P"): what every rank writes into ITS rows is what rank 0 reads after
    the `ready` fence, frame after frame -- the N > 1 exchange without a data collective, on gloo and host shared memory (the GPU form,
    hipIpcMemHandle, is tests/test_gpu_dist.py::test_sharded_p2p_window_equals_gather_and_single_process)."""
    out = str(tmp_path / "win.pt")
    mp.spawn(_window_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    seen = torch.load(out)
    assert len(seen) == 3
    for frame, full in enumerate(seen):
        for r in range(2):
            assert torch.equal(full[r], torch.full((2, 1000), float(10 * frame + r + 1))), (frame, r)
