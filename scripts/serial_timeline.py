"""Where one frame at a time spends its wall time: parse a `rocprofv3 --kernel-trace` CSV of `bench.py --frames-in-flight 1`, take the LAST
replay of the step graph (from the first K1 kernel to the last NMS kernel) and report the frame's span, the time with no kernel running, the
time with exactly one kernel running, and the longest single-kernel stretches (name, duration).
CAVEAT (measured, round 6): under `--kernel-trace` the dispatches of all streams are SERIALISED on one queue with ~20 us between them -- a
7.2 ms frame takes 13.4 ms and never shows two kernels at once -- so this gives the frame's kernel-time composition when every kernel runs
alone (7.84 ms for scene5: pointwise 2.96 ms in 112 launches, Winograd 2.35 ms), not the concurrency of the un-profiled run.
    python scripts/serial_timeline.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict


def main(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "k_voxb_insert" in r[2]]
    ends = [i for i, r in enumerate(rows) if "k_nms_reduce" in r[2]]
    # the last WHOLE frame: a K1 launch followed by an NMS launch with no other K1 launch and >= 250 kernels in between (bench.py also
    # replays the K1 / K8 chains alone for their graph period: those are not frames)
    s_i = e_i = None
    for c in reversed(starts):
        later = [i for i in ends if i > c]
        nxt = [i for i in starts if i > c]
        if later and (not nxt or later[0] < nxt[0]) and later[0] - c >= 250:
            s_i, e_i = c, later[0]
            break
    assert s_i is not None, "no complete frame in the trace"
    frame = [r for r in rows if r[0] >= rows[s_i][0] and r[1] <= rows[e_i][1]]
    t0, t1 = rows[s_i][0], rows[e_i][1]
    ev = []
    for a, b, n, q in frame:
        ev.append((a, 1, n)); ev.append((b, -1, n))
    ev.sort()
    depth, last, hist = 0, t0, defaultdict(int)
    single = []          # stretches with exactly one kernel running
    cur = set()
    for t, d, n in ev:
        hist[min(depth, 4)] += t - last
        if depth == 1 and t > last:
            single.append((t - last, next(iter(cur)) if cur else "?"))
        last = t
        if d == 1:
            cur.add(n)
        else:
            cur.discard(n)
        depth += d
    span = t1 - t0
    print(f"frame span {span / 1e3:.1f} us, {len(frame)} kernels, queues {len(set(q for *_, q in frame))}")
    for k in sorted(hist):
        print(f"  {k}{'+' if k == 4 else ''} kernels running: {hist[k] / 1e3:8.1f} us ({100.0 * hist[k] / span:4.1f} %)")
    agg = defaultdict(lambda: [0, 0])
    for d, n in single:
        agg[n[:70]][0] += d; agg[n[:70]][1] += 1
    print("alone on the chip (top 25 by total):")
    for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"  {d / 1e3:8.1f} us in {c:3d} stretches  {n}")


if __name__ == "__main__":
    main(sys.argv[1])
