"""The DEFERRED commit (svoslam_svo_fuse_commit_deferred + svoslam_svo_fuse_apply) alone on the GPU, next to the direct commit, on the old
cfg3 map: per-kernel times from the stage brackets and (under rocprofv3 --kernel-trace --stats) from the kernel statistics.
    python tools/prof/deferred_commit_alone.py [frames_in_map] [timed_frames]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import svoslam_pkg
pkg = svoslam_pkg.load()
synth = importlib.import_module("octree_slam_amd.synth")
pl = importlib.import_module("octree_slam_amd.pipeline")
hist = int(sys.argv[1]) if len(sys.argv) > 1 else 290
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w, h, depth, center, edge = 640, 480, 12, (0.0, 1.5, 0.0), 4.096
total = hist + 2 * nt
d, c = synth.render_stream(total, w, h, device="cuda")
views = [pl.ground_truth_view(k, synth) for k in range(total)]
P = pl.SlamPipeline(w, h, depth, center, edge, count_steps=True, pool_capacity_nodes=(1 << 30) - 8)
P.run_stream(d[:hist], c[:hist], list(range(hist)), views[:hist])
torch.cuda.synchronize()
n = w * h
for mode in ("deferred", "direct"):
    tot = {"plan": 0.0, "commit": 0.0}
    t_apply = 0.0
    for k in range(hist if mode == "deferred" else hist + nt, (hist if mode == "deferred" else hist + nt) + nt):
        P.track(d[k], c[k], k)
        pkg.svo_fuse_sort_frame(P.ws, d[k], P.cam.fusion_transform_ptr(), P.focal, P.focal, depth, center, edge, P.bbox)
        torch.cuda.synchronize()
        pkg.stage_timing([pkg.STAGE_FUSE_PLAN, pkg.STAGE_FUSE_COMMIT])
        pkg.svo_fuse_plan(P.ws, n, depth, P.pool)
        if mode == "deferred":
            pkg.svo_fuse_commit_deferred(P.ws, c[k].view(-1, 3), depth, P.pool)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); pkg.svo_fuse_apply(P.ws, P.pool); e1.record(); torch.cuda.synchronize()
            t_apply += e0.elapsed_time(e1)
        else:
            pkg.svo_fuse_split_early(P.ws, n, depth, P.pool)
            pkg.svo_fuse_commit(P.ws, c[k].view(-1, 3), depth, P.pool)
        torch.cuda.synchronize()
        tot["plan"] += pkg.stage_timing_read(pkg.STAGE_FUSE_PLAN)[0]
        tot["commit"] += pkg.stage_timing_read(pkg.STAGE_FUSE_COMMIT)[0]
        pkg.stage_timing([])
        P.render(views[k])
    print("%-8s plan %.1f us  commit %.1f us  apply %.1f us   (mean of %d frames, alone on the GPU, map of %d frames)"
          % (mode, tot["plan"] / nt * 1e3, tot["commit"] / nt * 1e3, t_apply / nt * 1e3, nt, hist))
