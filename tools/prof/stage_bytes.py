"""Per-stage algorithmic bytes (SURVEY 8d definitions) of cfg3 frames, combined with the per-kernel times of the
committed rocprofv3 summary -> profiles/r01_stage_bytes_cfg3.txt"""
import sys, os, csv, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STATS = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r02_bench_cfg3_kernel_stats.csv")
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "stage_bytes_cfg3.txt")
sys.path.insert(0, ROOT)
import svoslam_pkg
pkg = svoslam_pkg.load()
import importlib
synth = importlib.import_module("octree_slam_amd.synth")
pl = importlib.import_module("octree_slam_amd.pipeline")
W, H, D, edge, center = 640, 480, 12, 4.096, (0.0, 1.5, 0.0)
K = 105
depth, rgb = synth.render_stream(K, W, H, device="cuda")
P = pl.SlamPipeline(W, H, D, center, edge, count_steps=True, pool_capacity_nodes=1 << 28)
acc = {"V": 0, "UD": 0, "K": 0, "U": np.zeros(D), "S": 0, "L": 0}
frames = 0
lo = torch.tensor(center, device="cuda") - edge
for k in range(K):
    P.track(depth[k], rgb[k], k); P.backproject(depth[k])
    pts = P.points.view(-1, 3)
    st = P.fuse(rgb[k], blocking=True)
    P.counters.zero_()
    P.render(pl.ground_truth_view(k, synth))
    if k >= 5:
        ok = torch.isfinite(pts[:, 0]) & torch.isfinite(pts[:, 2])
        q = pts[ok]
        acc["V"] += int(ok.sum())
        acc["K"] += int(st.num_split)
        for l in range(1, D + 1):
            cell = torch.clamp(((q - lo) / (2 * edge / (1 << l))).floor().long(), 0, (1 << l) - 1)
            key = (cell[:, 0] << 40) | (cell[:, 1] << 20) | cell[:, 2]
            u = int(torch.unique(key).numel())
            if l == D: acc["UD"] += u
            else: acc["U"][l] += u
        c = P.counters.cpu().tolist()
        acc["S"] += c[0]; acc["L"] += c[1]
        frames += 1
f = float(frames)
V, UD, Kn, S, L = acc["V"] / f, acc["UD"] / f, acc["K"] / f, acc["S"] / f, acc["L"] / f
U = acc["U"] / f
N = [W * H, W * H // 4, W * H // 16]
its = [10, 5, 4]   # level 0, 1, 2 (rgbd_camera.cpp:19)
stages = {
    "bilateral": 4.0 * N[0],
    "maps (vertex + normal, 3 levels)": sum(n * (2 + 12 + 12 + 12) for n in N),
    "pyramid (subsample)": sum(N[l] * 2.5 for l in range(2)),
    "ICP (19 iterations)": sum(its[l] * N[l] * 48 for l in range(3)),
    "fuse": V * 15 + V * 4 * D + UD * 8 + Kn * 72 + U[1:].sum() * 36,
    "raycast": 4.0 * (L + S) + 4.0 * W * H,
}
# kernel time per frame from the committed rocprof summary
t = {}
for r in csv.reader(l for l in open(STATS) if not l.startswith("#")):
    if r[0] == "Name": continue
    t[r[0].replace("svoslam::", "").replace("void ", "")] = float(r[2]) / 105.0 / 1e3   # us per frame
def tsum(*names): return sum(v for k, v in t.items() if any(k.startswith(n) for n in names))
times = {
    "bilateral": tsum("bilateral_kernel"),
    "maps (vertex + normal, 3 levels)": tsum("vertex_normal_kernel"),
    "pyramid (subsample)": tsum("subsample_depth_kernel"),
    "ICP (19 iterations)": tsum("icp_accumulate_kernel", "cam_reduce_solve_kernel", "track_persistent_kernel"),
    "fuse": tsum("compute_keys", "keys_packed", "packed_", "radix_", "row_scan", "plan_", "split_all", "fill_mip_local", "mip_straddle", "vertex_map_kernel", "transform_kernel", "bbox_"),
    "raycast": tsum("cone_trace_kernel", "build_accel_kernel"),
}
out = ["# cfg3 (640x480, depth 12), mean over frames 5..104 of the bench stream; algorithmic bytes per SURVEY.md 8d;",
       "# kernel time per frame = sum of the per-kernel totals of %s / 105 frames" % os.path.basename(STATS),
       "# (kernels of different stages overlap in the four-stream pipeline, so the times do not add up to the frame period)",
       "# V = %.0f valid points, U_D = %.0f unique leaves, K = %.0f splits, S = %.3g march steps, mean levels per step %.2f" % (V, UD, Kn, S, L / S),
       "stage,alg_MB_per_frame,kernel_us_per_frame,GB_per_s,percent_of_8TBps"]
for k in stages:
    gbs = stages[k] / (times[k] * 1e-6) / 1e9
    out.append("%s,%.2f,%.1f,%.0f,%.2f" % (k, stages[k] / 1e6, times[k], gbs, gbs / 80.0))
open(OUT, "w").write("\n".join(out) + "\n")
print("\n".join(out))
