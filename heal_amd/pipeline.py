"""One-scene perception pipeline on one GPU: device point clouds -> voxelise (K1) -> model forward
(K2, dense BEV convs, K5) -> decode + rotated NMS (K8).  This is what `tools/inference.py`'s loop
body does per frame (to_device, model(batch['ego']), dataset.post_process), minus disk
I/O: inputs are already resident in HBM.
"""
import numpy as np
import torch

from heal_amd import _capi, synth
from heal_amd.opencood.data_utils.post_processor.voxel_postprocessor import VoxelPostprocessor
from heal_amd.opencood.tools.train_utils import create_model


def fill_deterministic(module, seed=0):
    """Random-init weights of the named architecture (there are no checkpoints on the box); BN
    statistics are kept non-trivial so that folding is exercised."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if not t.dtype.is_floating_point:
                continue
            leaf = name.rsplit(".", 1)[-1]
            if leaf == "running_var":
                t.copy_(0.7 + 0.6 * torch.rand(t.shape, generator=g))
            elif leaf == "running_mean":
                t.copy_(0.1 * torch.randn(t.shape, generator=g))
            elif t.dim() == 1 and leaf == "weight":
                t.copy_(1.0 + 0.1 * torch.randn(t.shape, generator=g))
            elif leaf in ("bias", "gamma"):
                t.copy_(0.05 * torch.randn(t.shape, generator=g))
            elif t.dim() >= 2:
                fan_in = int(np.prod(t.shape[1:]))
                t.copy_(torch.randn(t.shape, generator=g) * (1.6 / fan_in) ** 0.5)
    return module


RIG_KEYS = ("rots", "trans", "intrins", "post_rots", "post_trans")


def pack_rig(rig, device):
    """The five small rig tensors of a camera agent as VIEWS of one flat device tensor (`_rig`), so that loading a frame into
    the static buffers of a captured graph is one copy for the rig instead of five."""
    parts = [rig[k].to(torch.float32).reshape(-1) for k in RIG_KEYS]
    flat = torch.cat(parts).to(device)
    out, off = {"_rig": flat}, 0
    for k, p in zip(RIG_KEYS, parts):
        out[k] = flat[off:off + p.numel()].view(tuple(rig[k].shape))
        off += p.numel()
    return out


class Scene:
    """Synthetic OPV2V-shaped scene resident on one device.  `modalities[k]` names agent k's sensor
    suite: LiDAR agents ('m1', 'm3') get a 64-line sweep, camera agents ('m2', 'm4') get four
    normalised images and a camera rig."""

    CAMERA_DIMS = {"m2": (384, 512), "m4": (336, 448)}

    def __init__(self, n_agents, seed, device, max_cav=None, modalities=None):
        self.n_agents = n_agents
        self.device = torch.device(device)
        self.modalities = list(modalities) if modalities else ["m1"] * n_agents
        assert len(self.modalities) == n_agents
        self.points, self.cameras = {}, {}
        for k, mod in enumerate(self.modalities):
            if mod in self.CAMERA_DIMS:
                H, W = self.CAMERA_DIMS[mod]
                g = torch.Generator().manual_seed(seed * 1000 + k)
                rig = synth.camera_rig(seed * 1000 + k, 4, H, W)
                cam = pack_rig({name: torch.from_numpy(v) for name, v in rig.items()}, self.device)
                cam["imgs"] = torch.randn((4, 3, H, W), generator=g).to(self.device)
                self.cameras[k] = cam
            else:
                self.points[k] = torch.from_numpy(synth.lidar_frame(seed * 1000 + k)).to(self.device)
        self.poses = synth.agent_poses(seed, n_agents)
        L = max(max_cav or 5, n_agents)
        self.pairwise = synth.pairwise_t_matrix(self.poses, L)[None]  # [1,L,L,4,4] float64 (host metadata)
        self.record_len = [n_agents]

    def inputs_for(self, agents):
        """`inputs_mX` dictionaries (the reference's collated layout) for a subset of agents, in order."""
        out = {}
        for mod in sorted(set(self.modalities[a] for a in agents)):
            mine = [a for a in agents if self.modalities[a] == mod]
            if mod in self.CAMERA_DIMS:
                keys = ("imgs", "rots", "trans", "intrins", "post_rots", "post_trans")
                # one agent of the modality: a view (no copy of the images); several: the reference's collated stack
                out[f"inputs_{mod}"] = {k: (self.cameras[mine[0]][k].unsqueeze(0) if len(mine) == 1 else
                                            torch.stack([self.cameras[a][k] for a in mine])) for k in keys}
            else:
                out[f"inputs_{mod}"] = {"points": [self.points[a] for a in mine]}
        return out

    def model_input(self):
        d = self.inputs_for(list(range(self.n_agents)))
        d.update({"agent_modality_list": list(self.modalities), "record_len": self.record_len,
                  "pairwise_t_matrix": self.pairwise})
        return d


class StaticInputs:
    """Fixed-address device buffers holding one scene's inputs, so that a HIP graph captured around them can be replayed
    on OTHER frames (opencood/tools/inference.py:131-165 sees a new batch every iteration): `load(scene)` copies a scene
    in, `model_input()` / `inputs_for()` hand out views with the same layout `Scene` produces.

    * point clouds: one buffer per LiDAR agent with `slack` x the first scene's point count as capacity; the unused tail
      holds NaN points, which the voxeliser drops (same rows, same order as the un-padded cloud);
    * camera agents: images and rig tensors, fixed shapes;
    * `pairwise_t_matrix`: float64 [1,L,L,4,4] ON THE DEVICE -- the warp kernels read the poses at run time
      (heal_warp_fuse / heal_warp_agent `affine_dev`), so a replay uses the poses of the frame that was loaded."""

    CAMERA_DIMS = Scene.CAMERA_DIMS

    def __init__(self, scene, slack=1.25, agents=None):
        """agents: restrict the sensor buffers to these agent ids (a rank of the agent-sharded job only feeds its own
        agents); the pose matrices are always held."""
        self.device = scene.device
        self.modalities = list(scene.modalities)
        self.n_agents = scene.n_agents
        self.record_len = list(scene.record_len)
        self.points, self.cameras = {}, {}
        keep = set(range(scene.n_agents)) if agents is None else set(agents)
        # the clouds are slices of ONE buffer, in agent order: the collated voxeliser (ops.voxelize_collated) then reads the agents
        # of a modality in place instead of concatenating them first (a 32-us copy kernel per frame)
        caps = {k: max(1024, (int(int(p.shape[0]) * slack) + 1023) // 1024 * 1024) for k, p in scene.points.items() if k in keep}
        flat = torch.full((max(1, sum(caps.values())), 4), float("nan"), dtype=torch.float32, device=self.device)
        off = 0
        for k in sorted(caps):
            self.points[k] = flat[off:off + caps[k]]
            off += caps[k]
        self._n_points = {}
        for k, cam in scene.cameras.items():
            if k in keep:
                if "_rig" in cam:   # rig tensors as views of one flat buffer: one copy per frame
                    self.cameras[k] = pack_rig({name: torch.zeros_like(cam[name], device="cpu") for name in RIG_KEYS}, self.device)
                    self.cameras[k]["imgs"] = torch.empty_like(cam["imgs"])
                else:
                    self.cameras[k] = {name: torch.empty_like(t) for name, t in cam.items()}
        self.pairwise = torch.empty(tuple(scene.pairwise.shape), dtype=torch.float64, device=self.device)
        self._pairwise_pinned = torch.empty(tuple(scene.pairwise.shape), dtype=torch.float64).pin_memory()
        self.load(scene)

    def load(self, scene):
        """Copy `scene` into the static buffers (asynchronous on the current stream apart from the pinned staging of the
        4x4 pose matrices).  The scene must have the captured layout: same modalities, clouds within capacity."""
        if list(scene.modalities) != self.modalities:
            raise ValueError(f"scene layout {scene.modalities} differs from the captured layout {self.modalities}")
        for k, buf in self.points.items():
            p = scene.points[k]
            n = int(p.shape[0])
            if n > buf.shape[0]:
                raise ValueError(f"agent {k}: {n} points exceed the static capacity {buf.shape[0]} of the captured graph "
                                 "(capture with a larger `slack`)")
            buf[:n].copy_(p, non_blocking=True)
            prev = self._n_points.get(k, int(buf.shape[0]))   # rows >= prev already hold NaN points
            if n < prev:
                buf[n:prev].fill_(float("nan"))
            self._n_points[k] = n
        for k, cam in self.cameras.items():
            src = scene.cameras[k]
            if "_rig" in cam and "_rig" in src:
                cam["_rig"].copy_(src["_rig"], non_blocking=True)
                cam["imgs"].copy_(src["imgs"], non_blocking=True)
            else:
                for name, t in cam.items():
                    if name != "_rig":
                        t.copy_(src[name], non_blocking=True)
        pw = scene.pairwise
        if isinstance(pw, torch.Tensor) and pw.is_cuda:
            self.pairwise.copy_(pw.to(torch.float64), non_blocking=True)
        else:
            # the pinned staging buffer is reused: make sure the previous frame's H2D copy has left it (an event, not a stream
            # synchronisation: the stream may already hold work that waits for an earlier frame, FramesInFlight)
            ev = getattr(self, "_pin_ev", None)
            if ev is not None:
                ev.synchronize()
            self._pairwise_pinned.copy_(torch.as_tensor(np.asarray(pw), dtype=torch.float64))
            self.pairwise.copy_(self._pairwise_pinned, non_blocking=True)
            self._pin_ev = torch.cuda.Event()
            self._pin_ev.record(torch.cuda.current_stream(self.device))

    def scene_meta(self):
        """What the agent-sharded runner reads besides a rank's own sensor inputs: layout and (device) poses."""
        return {"agent_modality_list": list(self.modalities), "record_len": self.record_len,
                "pairwise_t_matrix": self.pairwise}

    # the two accessors below mirror Scene's
    def inputs_for(self, agents):
        return Scene.inputs_for(self, agents)

    def model_input(self):
        return Scene.model_input(self)


@torch.no_grad()
def calibrate_heads(model, model_input, score_threshold, target_candidates=600):
    """Random-init heads put an arbitrary fraction of the 131 072 anchors above the score threshold.  Shift the
    classification bias (a parameter like any other) so that about `target_candidates` anchors pass -- a busy but
    realistic frame for decode + NMS -- and keep regressed boxes near their anchors (std 0.15), otherwise exp(delta)
    sizes fail the reference's size / z filters and nothing reaches the NMS.  Returns the bias shift."""
    out = model(model_input)
    std = float(out["reg_preds"].std())
    if std > 0:
        for name, p in model.named_parameters():
            if name.startswith("reg_head"):
                p.mul_(0.15 / std)
    logits = out["cls_preds"].flatten()
    k = min(max(int(target_candidates), 1), logits.numel() - 1)
    kth = torch.topk(logits, k).values[-1]
    shift = float(np.log(score_threshold / (1.0 - score_threshold))) - float(kth)
    for name, p in model.named_parameters():
        if name.startswith("cls_head") and name.endswith("bias"):
            p.add_(shift)
    return shift


class ScenePipeline:
    def __init__(self, hypes, device, seed=0):
        self.hypes = hypes
        self.device = torch.device(device)
        self.model = fill_deterministic(create_model(hypes), seed).to(self.device).eval()
        self.post = VoxelPostprocessor(hypes["postprocess"], train=False)
        self.anchor_box = torch.from_numpy(self.post.generate_anchor_box()).to(self.device)
        self.tfm = torch.eye(4)

    @torch.no_grad()
    def calibrate_cls_bias(self, scene, target_candidates=600):
        return calibrate_heads(self.model, scene.model_input(),
                               self.hypes["postprocess"]["target_args"]["score_threshold"], target_candidates)

    @torch.no_grad()
    def forward(self, scene):
        return self.model(scene.model_input())

    @torch.no_grad()
    def step(self, scene):
        """One scene end to end; returns (pred_box3d [K,8,3] | None, scores | None)."""
        out = self.model(scene.model_input())
        batch = {"ego": {"transformation_matrix": self.tfm, "anchor_box": self.anchor_box}}
        return self.post.post_process(batch, {"ego": out})

    # ---- hipGraph replay ---------------------------------------------------------------------------
    # A scene is ~400 launches, most of them a few microseconds long: launch-bound on the host.  With
    # static shapes (inputs resident in fixed buffers; variable-size clouds can be padded with NaN
    # points, which the voxeliser drops) the whole step -- voxelise, forward, decode + NMS -- is
    # captured once into a HIP graph and replayed; the only host interaction left per scene is reading
    # the box count.
    @torch.no_grad()
    def capture_slot(self, scene, warmup=3, slack=1.25):
        """Capture the whole step around its OWN static input buffers (initialised from `scene`) on the current, non-default
        stream -> a `_Slot` (buffers, graph, output buffers, stream).  Several slots of one pipeline share the model's
        parameters and nothing else: scratch buffers are per stream (ops._workspace) and so are the modality side streams."""
        from heal_amd import ops
        dir_args = self.post.params.get("dir_args", {"dir_offset": 0.7853, "num_bins": 2})
        anchors = self.post._anchors_f32(self.anchor_box, self.device)
        static = StaticInputs(scene, slack)

        def body():
            out = self.model(static.model_input())
            return ops.decode_nms(out["cls_preds"], out["reg_preds"], out.get("dir_preds"), anchors,
                                  self.post.params["target_args"]["score_threshold"], dir_args["dir_offset"],
                                  dir_args["num_bins"], self.post.params["nms_thresh"],
                                  np.eye(4, dtype=np.float32), self.post.params["gt_range"], sync=False)

        # Capture on the stream the caller already runs on (it must be a non-default stream): MIOpen keeps
        # its solver choices per stream, and the first convolutions on a fresh stream trigger a search that
        # takes minutes for the ~100 convolution shapes of the heterogeneous model.
        cur = torch.cuda.current_stream(self.device)
        if cur == torch.cuda.default_stream(self.device):
            raise RuntimeError("capture() must be called under a non-default stream (torch.cuda.stream(s))")
        for _ in range(warmup):
            body()
        cur.synchronize()
        ops.verify_sparse_capacity()
        graph = torch.cuda.CUDAGraph()
        _capi.guard_begin()
        try:
            with torch.cuda.graph(graph, stream=cur):
                static_out = body()
        except BaseException:
            _capi.guard_begin()               # a failed capture owns nothing
            raise
        # the capacity counters written inside the graph live in its private pool: keep them to re-check after every replay
        slot = _Slot(static, graph, static_out, ops.take_sparse_checks(), cur)
        slot.guard = _capi.guard_take()       # HEAL_GRAPH_GUARD=1: every device address the capture handed to a kernel
        return slot

    @torch.no_grad()
    def capture(self, scene, warmup=3, slack=1.25):
        """Capture the whole step around STATIC input buffers initialised from `scene` (StaticInputs); `replay()` re-runs
        it on whatever the buffers hold, `replay(other_scene)` loads another frame of the same layout first."""
        slot = self.capture_slot(scene, warmup, slack)
        self._static_in, self._graph, self._static_out, self._graph_checks = slot.static_in, slot.graph, slot.out, slot.checks
        self._guard = slot.guard
        return slot.graph

    def check_sparse_capacity(self):
        """SECOND encoders fed with device point clouds size their strided layers by capacity (no host round trip);
        this host-side check (it synchronises) raises if any layer found more active sites than its capacity."""
        from heal_amd import ops
        ops.verify_sparse_capacity(getattr(self, "_graph_checks", None))

    def replay(self, scene=None):
        """Replay the captured step -- on `scene` (loaded into the static input buffers first) or on whatever the buffers
        hold; returns (pred_box3d | None, scores | None) like step()."""
        if scene is not None:
            self._static_in.load(scene)
        if getattr(self, "_guard", None):
            _capi.guard_check(self._guard, "ScenePipeline.replay")
        self._graph.replay()
        corners, scores, count = self._static_out
        k = int(count.item())
        if self._graph_checks:   # after the sync above: a denser frame than the capacity policy allows must not pass silently
            self.check_sparse_capacity()
        if k == 0:
            return None, None
        return corners[:k], scores[:k]


class _Slot:
    """One captured copy of the step: static input buffers, HIP graph, static output buffers, the stream it replays on."""

    def __init__(self, static_in, graph, out, checks, stream):
        self.static_in, self.graph, self.out, self.checks, self.stream = static_in, graph, out, checks, stream
        self.done = torch.cuda.Event()
        self.guard = []


class FramesInFlight:
    """Throughput mode of the replayed step: `depth` captured copies of the step, each on its own stream with its own static
    input / output buffers; frame k is loaded into slot k % depth and replayed there while frame k - 1 is still running on the
    other slot, and its boxes are read one `step()` later.  What overlaps is the latency-bound part of one frame (voxeliser,
    image trunks at 1/16 resolution, decode + NMS: small launches that leave most CUs idle) with the matrix-bound part of its
    neighbour -- the same effect the concurrent modality stems have inside one frame.  Every frame still runs the whole step;
    per-frame latency is that of the plain replay (or a little more), the rate is what changes.

        ring = FramesInFlight(pipe, scene, depth=2)
        for frame in frames:  res = ring.step(frame)      # result of the frame submitted len(ring.slots) - 1 steps earlier (None at first)
        rest = ring.drain()                               # results still in flight, oldest first
    """

    def __init__(self, pipe, scene, depth=2, warmup=3, slack=1.25, queue_ahead=0):
        """depth: frames that RUN concurrently.  queue_ahead: extra captured copies whose frame is loaded and launched by the host
        while `depth` frames are still running, but whose stream waits (on the device) for the frame `depth` places ahead of it to
        finish -- the next frame starts the moment a running one ends instead of after the host's read-back + load + launch.
        Measured with queue_ahead = 1: SLOWER (scene5 7.50 vs 7.10 ms per frame, config 5 24.1 vs 23.7): the frames then run in
        lock-step phase; the host's gap happens to stagger them so that one frame's small launches meet the other's big ones.
        Hence the default 0."""
        from collections import deque
        self.pipe = pipe
        self.depth = depth
        self.slots = []
        for _ in range(depth + max(0, int(queue_ahead))):
            stream = torch.cuda.Stream(device=pipe.device)
            with torch.cuda.stream(stream):
                self.slots.append(pipe.capture_slot(scene, warmup, slack))
            stream.synchronize()
        self._next = 0
        self._inflight = deque()
        self._done = deque(maxlen=depth)     # completion events of the last `depth` submitted frames

    def _collect(self):
        from heal_amd import ops
        slot = self._inflight.popleft()
        with torch.cuda.stream(slot.stream):
            corners, scores, count = slot.out
            k = int(count.item())          # waits for this slot's stream only
            # the slot's output buffers are overwritten by its next frame: copy on the SLOT's stream (ordered before that
            # replay), then let the caller's stream wait for the copy
            res = (None, None) if k == 0 else (corners[:k].clone(), scores[:k].clone())
        torch.cuda.current_stream(self.pipe.device).wait_stream(slot.stream)
        if slot.checks:
            ops.verify_sparse_capacity(slot.checks)
        return res

    def step(self, scene):
        """Submit `scene`; returns the (boxes, scores) of the oldest frame in flight once every slot is taken, else None."""
        slot = self.slots[self._next % len(self.slots)]      # free: its previous frame was collected
        self._next += 1
        with torch.cuda.stream(slot.stream):
            if len(self._done) == self.depth:
                slot.stream.wait_event(self._done[0])        # at most `depth` frames run at a time
            slot.static_in.load(scene)
            if slot.guard:
                _capi.guard_check(slot.guard, "FramesInFlight.step")
            slot.graph.replay()
            ev = torch.cuda.Event()
            ev.record(slot.stream)
        self._done.append(ev)
        self._inflight.append(slot)
        if len(self._inflight) == len(self.slots):
            return self._collect()
        return None

    def drain(self):
        out = []
        while self._inflight:
            out.append(self._collect())
        return out

