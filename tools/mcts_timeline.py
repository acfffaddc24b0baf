"""Summarise a DTK_TRACE_MCTS file (infer/batching.py): when the trees were ready, how long rollouts / rewards took, how many
trees were decoding over time.   DTK_TRACE_MCTS=/tmp/t.json python bench.py ... ; python tools/mcts_timeline.py /tmp/t.json"""
import json, sys
ev = json.load(open(sys.argv[1]))
t0 = ev[0][0]
ev = [(t - t0, tree, name) for t, tree, name in ev]
end = max(t for t, _, _ in ev)
ready = sorted(t for t, _, n in ev if n == "tree_ready")
print(f"total {end:.2f} s; trees ready between {ready[0]:.3f} and {ready[-1]:.3f} s")
for name in ("rollout", "reward"):
    starts = {}
    spans = []
    for t, tree, n in sorted(ev):
        if n == name + "_start":
            starts[tree] = t
        elif n == name + "_end" and tree in starts:
            spans.append((starts.pop(tree), t))
    if spans:
        d = sorted(b - a for a, b in spans)
        print(f"{name}: {len(spans)} spans, median {d[len(d) // 2]:.3f} s, min {d[0]:.3f}, max {d[-1]:.3f}; first start {min(a for a, _ in spans):.3f}, last end {max(b for _, b in spans):.3f}")
# trees inside a rollout over time, 100 ms buckets
edges = sorted([(t, +1) for t, _, n in ev if n == "rollout_start"] + [(t, -1) for t, _, n in ev if n == "rollout_end"])
cur, i, line = 0, 0, []
for b in range(int(end * 10) + 1):
    while i < len(edges) and edges[i][0] <= b / 10:
        cur += edges[i][1]; i += 1
    line.append(cur)
print("trees in a rollout, every 100 ms:", " ".join(map(str, line)))

# per tree: what happens between the end of a rollout and the start of the next one
by_tree = {}
for t, tree, n in sorted(ev):
    if tree >= 0:
        by_tree.setdefault(tree, []).append((t, n))
gaps = {"rollout_end -> reward_start": [], "reward (ViT + image work)": [], "reward_end -> next rollout_start": [], "rollout_end -> next rollout_start": []}
for tree, es in by_tree.items():
    last_end = last_rw_end = None
    for t, n in es:
        if n == "rollout_end":
            last_end, last_rw_end = t, None
        elif n == "reward_start" and last_end is not None:
            gaps["rollout_end -> reward_start"].append(t - last_end); rs = t
        elif n == "reward_end" and last_end is not None:
            gaps["reward (ViT + image work)"].append(t - rs); last_rw_end = t
        elif n == "rollout_start" and last_end is not None:
            if last_rw_end is not None:
                gaps["reward_end -> next rollout_start"].append(t - last_rw_end)
            gaps["rollout_end -> next rollout_start"].append(t - last_end)
            last_end = None
for k, v in gaps.items():
    if v:
        v.sort()
        print(f"{k:36s}: n {len(v):3d}  median {1e3 * v[len(v) // 2]:7.1f} ms  p90 {1e3 * v[int(len(v) * .9)]:7.1f} ms  max {1e3 * v[-1]:7.1f} ms")
