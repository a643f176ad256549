"""The local stage of a rank that owns ONE camera agent of scene5 (what bounds the 8-GPU period in scripts/scaling_model.py: 3.07 ms for
the m2 agent, 1.81 ms for m4): captured as `_Sharded.capture` does, replayed alone.  Prints the replay period; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel list (sum of kernel time vs the period = idle between the launches).

    python scripts/cam_stage_probe.py [--agent 3] [--iters 20]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agent", type=int, default=3)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--depth", type=int, default=1, help="copies of the stage in flight (own stream, own static inputs and graph each)")
    a = ap.parse_args()
    from bench import WORKLOADS
    from heal_amd import configs
    from heal_amd.dist import agent_owner, make_sharded, owned_agents
    from heal_amd.pipeline import Scene, ScenePipeline, StaticInputs
    from scripts.scaling_model import timed_graph
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    mods, _ = WORKLOADS["scene5"]
    n_agents, N = len(mods), 8
    pipe = ScenePipeline(configs.heal_heter(tuple(sorted(set(mods))), max_cav=5), dev, seed=0)
    scene = Scene(n_agents, seed=4, device=dev, modalities=mods)
    r = agent_owner(a.agent, N)
    mine = owned_agents(n_agents, r, N)
    assert mine == [a.agent], mine
    with torch.no_grad():
        slots = []
        for d in range(a.depth):
            st = stream if d == 0 else torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                runner = make_sharded(pipe.model, r, N, collective="gather")
                static = StaticInputs(scene, agents=mine)
                static.load(scene)
                li, inp = static.inputs_for(mine), static.scene_meta()
                buf, ms, g = timed_graph(lambda: runner.local(inp, n_agents, li), st, iters=a.iters)
            slots.append((g, st, buf, static, runner))
        print(f"agent {a.agent} ({mods[a.agent]}): local stage {ms:.3f} ms per replay (alone)", flush=True)
        if a.depth > 1:
            import time
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            total = a.iters * a.depth
            for k in range(total):
                g, st = slots[k % a.depth][:2]
                with torch.cuda.stream(st):
                    g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3 / total
            print(f"agent {a.agent} ({mods[a.agent]}): {a.depth} frames in flight: {dt:.3f} ms per frame", flush=True)


if __name__ == "__main__":
    main()
