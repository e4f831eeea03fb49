#!/usr/bin/env python
"""bench.py -- SLAM frames/s (bilateral + ICP + fuse + raycast) on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step is one SLAM frame of the synthetic RGB-D stream of BASELINE.md config 3
(640x480, depth-12 SVO, root centre (0,1.5,0), half-edge 4.096 m): the body of
mainLoop() (src/main.cpp:31-84) with the tracker enabled -- bilateral + pyramids
+ 19 ICP iterations + back-projection + fusion + one cone-traced frame.  All
frames are resident in HBM before the timed region.  `--workload cfg4` runs the
1920x1080 / depth-14 stream of config 4.

With N > 1 the total work is fixed: scaling = "strong".  --exchange deltas (default
for N > 1): frames are tracked in parallel -- rank k % N runs the
19 ICP iterations of frame k (a function of depth images k-1 and k only) and ray-marches
it; the 80-byte update_trans records are all-gathered over RCCL, every rank composes the
same poses and applies every fusion to its own replica of the map.  --exchange none:
every rank tracks and fuses whole frames, only the raycast is cut into N row bands, no
collective in the frame loop.  --exchange allreduce: SURVEY 8e (ICP accumulation and
back-projection per band; ICP sums all-reduced, point bands all-gathered, fusion applied
to every replica).  --emulate-rank R/N: rank R of an N-rank "deltas" session on ONE GPU,
the other ranks' records precomputed outside the timed region (what one rank of N costs).

cfg3 is BASELINE config 3: a stream of 300 frames.  With K + W < 300 the first 300 - K - W frames are fused UNTIMED
through the same four-stream path, so that the W warm-up and K timed frames are the LAST frames of the 300-frame
config (a young map has short rays and flatters the number: VERDICT r02); `config.frames_in_map_at_end` = 300.
--map-frames overrides (0 = K + W, the young map).

The default single-GPU cfg3 line also carries `other_configs` (round 5): cfg4 through a child run of this script (`--workload cfg4 --lean`),
cfg2 and the cfg5 stand-in through tools/mesh_bench.py (stage times from HIP events inside the C calls, algorithmic bytes, fractions, renders),
and `wall` (seconds in this process's GPU legs, in the children, in the CPU oracle -- which runs LAST).

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel = whichever of the march kernel and the tracker
kernel EXECUTES longer (each alone on the GPU in the sequential pass after the timed region: the order of the rocprofv3
kernel statistics; an event interval on a busy device also contains the launch's wait for its turn, which is most of
the difference between the tracker's 0.25 ms and the 0.31-0.35 ms between its events).  Its numbers are the LIVE ones:
both kernels are timed with HIP events on their launch streams inside the timed region (svoslam_stage_timing),
algorithmic bytes per launch (march 4*(levels+steps) + 4*W*H, ICP 48 B per pixel per iteration) over the
mean interval.  `roofline_stages` carries the same for march, tracker and
fusion; the fusion's 18 launches are timed in a short SEQUENTIAL pass over three more frames after the timed region
(event pairs on the map stream would cost the timed region ~1.5 %).  `traffic` = HBM bytes per frame from the committed
rocprofv3 PMC passes (profiles/pmc_traffic.json, written by tools/prof/profile_round.sh; counters cannot be sampled
from inside the process) -- missing entries for the chosen workload are an ERROR, not a null.
`cpu_baseline` times the single-thread CPU oracle (kind "port") on the TIMED frames of the same stream, continued from the
GPU's map and pose before them.  `value` is the median of --repeats windows (each from an empty map); `runs` lists them all,
`pipeline_fill` states the share of the cold five-stream pipeline's fill inside the window.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

# The frame pipeline uses four HIP streams next to the default one; the HIP runtime multiplexes streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams sharing a queue run in order.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (width, height, max_depth, centre, half_edge)
    "cfg3": (640, 480, 12, (0.0, 1.5, 0.0), 4.096),
    "cfg4": (1920, 1080, 14, (0.0, 1.5, 0.0), 8.192),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_MEASURED_COPY_GBS = 6290.0  # the same guide's measured copy ceiling; quoted beside the spec peak


LINE_LIMIT = 6000   # bytes: the driver keeps ~8 KB of stdout; round 5's 20.9 KB line was cut and its record lost


def _sig(x, nd=4):
    """floats to `nd` significant digits (the full-precision numbers are in the details file)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float("%.*g" % (nd, x))
    if isinstance(x, (list, tuple)):
        return [_sig(v, nd) for v in x]
    if isinstance(x, dict):
        return {k: _sig(v, nd) for k, v in x.items()}
    return x


def _cut(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3] + "..."


def compact_line(full, details_path=None, limit=LINE_LIMIT):
    """The ONE stdout line of the contract, built from the full record: every contract key (metric, value, unit, n_gpus, steps, warmup,
    ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload, roofline, cpu_baseline) plus a few numbers per stage
    and per other configuration; everything else lives in `details_path`.  Optional parts are dropped, last first, until the line is
    <= `limit` bytes -- the contract keys never are.  Pure function of `full` (tests/test_bench_line.py runs it on canned records)."""
    cfgf = full.get("config") or {}
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    mem = cfgf.get("device_memory_GiB") or {}
    out["config"] = {k: cfgf[k] for k in ("workload", "parallelism", "frames_in_map_at_end", "frames_input", "mrays_per_s", "pool_nodes_end",
                                          "image_coloured_pixels_end", "saturated_nodes_end") if k in cfgf}
    out["config"]["workload"] = _cut(out["config"].get("workload"), 200)
    out["config"]["parallelism"] = _cut(out["config"].get("parallelism"), 160)
    if mem:
        out["config"]["device_GiB_in_use"] = mem.get("device_in_use_all_processes")
    rf = full.get("roofline")
    if rf:
        out["roofline"] = {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "alg_bytes_per_launch")}
        out["roofline"]["limiter"] = _cut(rf.get("limiter"), 140)
    else:
        out["roofline"] = None
    cb = full.get("cpu_baseline")
    out["cpu_baseline"] = ({k: (_cut(cb.get(k), 260) if k == "sample" else cb.get(k)) for k in ("value", "unit", "cores", "kind", "sample")}
                           if cb else None)
    if "runs" in full:
        out["runs"] = full["runs"]

    def stage_row(r):
        row = {"stage": r.get("stage"), "kernel": _cut(r.get("kernel"), 48), "ms": r.get("kernel_ms"), "alg_MB": (r.get("alg_bytes_per_launch") or 0) / 1e6,
               "frac": r.get("frac")}
        if r.get("traffic") is not None:
            row["traffic_MB"] = r["traffic"] / 1e6
        if r.get("floor_bytes") is not None:
            row["floor_MB"] = r["floor_bytes"] / 1e6
            if r.get("traffic_over_floor"):
                row["traffic_over_floor"] = r["traffic_over_floor"]
        if r.get("launches_per_frame") is not None:
            row["launches"] = r["launches_per_frame"]
        return row

    optional = []   # (key, value): appended in this order, dropped from the END when the line is too long
    if full.get("roofline_stages"):
        optional.append(("roofline_stages", [stage_row(r) for r in full["roofline_stages"]]))
    oc = full.get("other_configs")
    if oc:
        c = {}
        for name, d in oc.items():
            if not isinstance(d, dict):
                continue
            if "error" in d:
                c[name] = {"error": _cut(d["error"], 120)}
                continue
            e = {}
            if "value" in d:       # a stream configuration (cfg4) through this script
                e.update({"value": d["value"], "unit": d.get("unit"), "ms_per_step": d.get("ms_per_step")})
                r4 = d.get("roofline") or {}
                e.update({"kernel": _cut(r4.get("kernel"), 40), "kernel_ms": r4.get("kernel_ms"), "frac": r4.get("frac")})
                if d.get("cpu_baseline"):
                    e["cpu_value"] = d["cpu_baseline"].get("value")
                if d.get("roofline_stages"):
                    e["stages"] = [{"stage": r.get("stage"), "ms": r.get("kernel_ms"), "frac": r.get("frac")} for r in d["roofline_stages"]]
            else:                  # a mesh configuration (cfg2 / cfg5) through tools/mesh_bench.py
                st = d.get("stages") or {}
                for k in ("voxelize", "svo_from_voxel_grid"):
                    if k in st:
                        e[k + "_ms"] = st[k].get("ms")
                        e[k + "_frac"] = st[k].get("frac")
                rs = d.get("renders") or []
                e["renders"] = [{"view": r.get("view"), "mode": (r.get("mode") or "")[:3], "ms": r.get("trace_kernel_ms"), "Mrays_s": r.get("Mrays_per_s"),
                                 "frac": r.get("frac"), "lit": r.get("lit_pixels")} for r in rs]
                e["kernel"] = _cut(d.get("render_kernel"), 60)
                e["voxels"] = d.get("voxels")
            c[name] = e
        optional.append(("other_configs", c))
    for k in ("stages_sequential", "corrected_tracker", "latency", "pipeline_fill", "other_partition", "wall"):
        v = full.get(k)
        if isinstance(v, dict):
            optional.append((k, {a: b for a, b in v.items() if not isinstance(b, (str, dict, list)) or a in ("exchange", "error")}))
    if details_path:
        out["details"] = details_path
    out = _sig(out, 5)
    optional = [(k, _sig(v, 4)) for k, v in optional]
    while True:
        line = json.dumps({**out, **dict(optional)}, separators=(",", ":"))
        if len(line.encode()) <= limit or not optional:
            break
        optional.pop()
    if len(line.encode()) > limit:     # (cannot happen with the cuts above; never print a line the driver would truncate)
        for k in ("runs",):
            out.pop(k, None)
        out["cpu_baseline"] = out["cpu_baseline"] and {**out["cpu_baseline"], "sample": _cut(out["cpu_baseline"].get("sample"), 80)}
        line = json.dumps(out, separators=(",", ":"))
    assert len(line.encode()) <= limit, len(line)
    return line


def write_details(full, path):
    """the full record (every number of every leg) beside the compact line; a failure to write is reported, never fatal"""
    try:
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f)
            f.write("\n")
        return path
    except OSError as e:
        print("bench.py: could not write %s: %r" % (path, e), file=sys.stderr)
        return None


def cpu_baseline(depth_frames, rgb_frames, views, first, width, height, max_depth, center, edge, seed_words, seed_pose, budget_s=20.0, min_frames=1):
    """Single-thread CPU oracle on the TIMED frames of the same stream (reported baseline only): its pool starts from the
    words of the GPU's map after frame first - 1 (bit-identical to the oracle's own, tests/test_gpu_fullsize.py), its tracker
    from the GPU's pose and the maps of frame first - 1 (built untimed), then frames first, first + 1, ... until the budget."""
    import numpy as np
    from oracle import oracle as ora
    try:
        L = ora.lib(native=True)   # rebuilt on this box with -O3 -march=native (results stay bit-identical)
        flags = "-O3 -march=native"
    except (OSError, RuntimeError, subprocess.CalledProcessError) as e:   # no compiler on the box: say so IN the record
        print("bench.py: native oracle build failed (%r): cpu_baseline uses the prebuilt -O2 library" % (e,), file=sys.stderr)
        L = ora.lib()
        flags = "-O2 (prebuilt; the -O3 -march=native rebuild FAILED on this box: %s)" % type(e).__name__
    import ctypes as C
    focal = 570.3 * width / 640.0
    cam = ora.Camera(width, height, focal, focal, L=L)
    pool = ora.Pool(L=L)
    if first > 0:
        pool.load_words(seed_words)
        k0 = first - 1
        cam.update(depth_frames[k0].cpu().numpy().view(np.uint16), rgb_frames[k0].cpu().numpy(), k0)   # the "last frame" maps
        cam.set_pose(*seed_pose)
    n_done, t_total = 0, 0.0
    for k in range(first, len(depth_frames)):
        d = depth_frames[k].cpu().numpy().view(np.uint16)
        c = rgb_frames[k].cpu().numpy()
        t0 = time.perf_counter()
        cam.update(d, c, k)
        v = np.empty((height, width, 3), np.float32)
        L.ora_generate_vertex_map(d.ctypes.data_as(C.POINTER(C.c_uint16)), v.ctypes.data_as(C.POINTER(C.c_float)), width, height,
                                  focal, focal, width, height)
        m = cam.fusion_transform()
        L.ora_transform_vertex_map(v.ctypes.data_as(C.POINTER(C.c_float)), m.ctypes.data_as(C.POINTER(C.c_float)), width * height)
        b0 = np.zeros(3, np.float32); b1 = np.zeros(3, np.float32)
        L.ora_point_cloud_bbox(v.ctypes.data_as(C.POINTER(C.c_float)), width * height, b0.ctypes.data_as(C.POINTER(C.c_float)),
                               b1.ctypes.data_as(C.POINTER(C.c_float)))
        pool.insert_cloud(v.reshape(-1, 3), c.reshape(-1, 3), max_depth, center, edge)
        ora.cone_trace(pool, width, height, 45.0, views[k], center, edge, ora.RENDER_REFERENCE, L=L)
        t_total += time.perf_counter() - t0
        n_done += 1
        if t_total > budget_s and n_done >= min_frames:
            break
    return {"value": n_done / t_total, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "frames %d..%d of the same %dx%d depth-%d stream = the first %d of the GPU's timed frames, from the map and pose the GPU "
                      "had before them (%d nodes); CPU oracle (gcc %s, single thread), %.1f s"
                      % (first + 1, first + n_done, width, height, max_depth, n_done, pool.size, flags, t_total)}


def device_memory(P, torch):
    """what the session holds on the device: the pool's reservation, the deferred commits' shadow array (8 B per
    node of capacity, when deferred commits are on), the occupancy bricks' dense field (2 x 2048^3 bytes, taken only when three times
    its size is free), the level grid (256^3 x 8 B), and what the process has allocated in total"""
    GiB = float(1 << 30)
    free_b, total_b = torch.cuda.mem_get_info()
    cap = int(P.pool.capacity)
    return {"pool_reserved": cap * 8 / GiB, "pool_used": int(P.pool.size) * 8 / GiB, "deferred_shadow_if_on": cap * 8 / GiB,
            "brick_field_if_taken": 16.0, "level_grid": (256 ** 3) * 8 / GiB, "march_accel": P.pool.march_accel(),
            "device_in_use_all_processes": (total_b - free_b) / GiB, "device_total": total_b / GiB}


def other_configs(args):
    """the other BASELINE configurations, each in a child process on the same GPU; a failure is IN the record"""
    out = {}

    def child(cmd, timeout):
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                return {"error": "exit %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:])}, time.perf_counter() - t0
            return json.loads(lines[-1]), time.perf_counter() - t0
        except (subprocess.TimeoutExpired, OSError, ValueError) as e:
            return {"error": repr(e)}, time.perf_counter() - t0

    me = os.path.abspath(__file__)
    d, w = child([sys.executable, me, "--workload", "cfg4", "--steps", "40", "--warmup", "5", "--repeats", "3", "--lean", "--full-line", "--cpu-budget", "7", "--cpu-min-frames", "5"]
                 + (["--no-cpu-baseline"] if args.no_cpu_baseline else []), 300)
    if "error" in d:
        out["cfg4"] = d
    else:
        keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "runs", "value_min", "value_max", "dtype", "roofline", "roofline_stages",
                "stages_sequential", "cpu_baseline")
        out["cfg4"] = {k: d[k] for k in keep if k in d}
        out["cfg4"]["workload"] = d["config"]["workload"]
        out["cfg4"]["overlap"] = d["config"]["overlap"]
        out["cfg4"]["pool_nodes_end"] = d["config"].get("pool_nodes_end")
        out["cfg4"]["mrays_per_s"] = d["config"].get("mrays_per_s")
        out["cfg4"]["device_memory_GiB"] = d["config"].get("device_memory_GiB")   # incl. march_accel: a degraded child run is visible
        out["cfg4"]["what"] = ("BASELINE config 4 on ONE GPU (its 8-GPU tiling is bench.py --gpus 8 --workload cfg4): `python bench.py --workload cfg4 --steps 40 "
                               "--warmup 5 --repeats 3 --lean` in a child process; median of 3 windows of 40 frames from an empty map")
    out["cfg4"]["wall_s"] = w
    mb = os.path.join(ROOT, "tools", "mesh_bench.py")
    for cfg in ("cfg2", "cfg5"):
        d, w = child([sys.executable, mb, "--config", cfg, "--reps", "2" if cfg == "cfg5" else "3"], 300)
        d["wall_s"] = w
        out[cfg] = d
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--render-mode", default="reference", choices=["reference", "carry"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tracker", default="reference", choices=["reference", "corrected"],
                    help="reference (the headline): RGBDCamera::update as the reference has it, Q14's Jacobian included -- its pose "
                         "drifts by tens of degrees over the stream; corrected: this build's own specification "
                         "(svoslam_camera_set_strict_reference(cam, 0)): a SECOND, labelled line, never the headline")
    ap.add_argument("--no-corrected-line", action="store_true",
                    help="single GPU, reference tracker: do not add the short `corrected_tracker` measurement to the line")
    ap.add_argument("--no-overlap", action="store_true", help="one stream, stages strictly in sequence")
    ap.add_argument("--include-h2d", action="store_true",
                    help="frames start in pinned HOST memory and are uploaded inside the timed region (the reference's frame "
                         "includes the 1.5 MB cudaMemcpy of openni_device.cpp:122,144); default: frames resident in HBM")
    ap.add_argument("--exchange", default=None, choices=["deltas", "none", "allreduce", "keyrange"],
                    help="N > 1: 'deltas' (default) = frame-parallel tracking, all-gather of update_trans records, every rank fuses "
                         "every frame and ray-marches its own; 'none' = every rank tracks and fuses whole frames, only the raycast "
                         "is split into row bands; 'allreduce' = SURVEY 8e row bands with 19 ICP all-reduces + one point all-gather "
                         "per frame; 'keyrange' = 'deltas' with the fusion cut by key range: every rank plans + commits its slice of every "
                         "frame's sorted keys, one all-gather of the ranks' deltas per frame (svoslam_svo_fuse_keyrange_*)")
    ap.add_argument("--emulate-rank", default=None, metavar="R/N",
                    help="one GPU: run rank R of an N-rank 'deltas' session, the other ranks' records precomputed (untimed)")
    ap.add_argument("--map-frames", type=int, default=None,
                    help="frames in the map when the timed region ENDS (default: 300 for cfg3 = BASELINE config 3, K + W for "
                         "cfg4); the frames before the warm-up are fused untimed; 0 = K + W")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-frame window is measured this many times, each from an empty map (reset + history + warm-up + K timed "
                         "frames); `value` is the median run, `runs` lists them all")
    ap.add_argument("--no-other-partition", action="store_true",
                    help="N > 1: do not also measure the other partition (by default the line carries both north_star's row-band "
                         "'allreduce' scheme and the frame-sharded 'deltas' scheme: `value` is --exchange's, `other_partition` the other's)")
    ap.add_argument("--no-stage-pass", action="store_true", help="skip the sequential per-stage pass after the timed region")
    ap.add_argument("--allow-missing-traffic", action="store_true",
                    help="profiles/pmc_traffic.json incomplete for this workload: report traffic null instead of failing "
                         "(used by the profiling script that GENERATES that file)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default single-GPU cfg3 run: do not add `other_configs` (cfg4 through this script in a child process, the mesh "
                         "configurations cfg2 / cfg5 through tools/mesh_bench.py)")
    ap.add_argument("--lean", action="store_true",
                    help="the line without its side measurements (latency window, pipeline fill, corrected-tracker line): what the "
                         "`other_configs.cfg4` child process runs")
    ap.add_argument("--full-line", action="store_true",
                    help="print the FULL record as the stdout line (tens of KB) instead of the compact one: what the child runs and "
                         "tools/prof/*.sh ask for (also SVOSLAM_BENCH_FULL_LINE=1); the driver's run never does")
    ap.add_argument("--details", default=os.path.join(ROOT, "bench_details.json"),
                    help="where the full record goes beside the compact line ('' = nowhere)")
    ap.add_argument("--cpu-min-frames", type=int, default=1, help="`cpu_baseline` runs at least this many frames, whatever --cpu-budget says")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of single-thread CPU oracle work for `cpu_baseline`")
    ap.add_argument("--stages", action="store_true",
                    help="add `stages` (per-stage durations from HIP-event marks at the stage boundaries, svoslam_runner_timeline); "
                         "the ~10 extra event records per frame cost ~6 %% of the frame rate, so they are off for the headline line")
    args = ap.parse_args()

    t_start = time.perf_counter()
    wall = {}
    import numpy as np
    import torch
    import svoslam_pkg
    pkg = svoslam_pkg.load()
    synth = importlib.import_module("octree_slam_amd.synth")
    pl = importlib.import_module("octree_slam_amd.pipeline")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # testing aid for one-GPU boxes: SVOSLAM_BENCH_ONE_DEVICE=1 puts every rank on device 0 and uses the gloo backend
    # (RCCL refuses two ranks on one device) -- the multi-rank code path of this file end to end, not a measurement
    one_device = os.environ.get("SVOSLAM_BENCH_ONE_DEVICE") == "1"
    one_device_chain = False
    if one_device:
        local_rank = 0
        # several PROCESSES on one device: the one-launch tracker's workgroups wait for each other and assume an idle
        # device (csrc/track_persistent.hip); the launch chain has no such assumption
        one_device_chain = True
        # likewise the peer-to-peer mailbox: its collect kernel polls until the peers have posted, and processes that share ONE
        # device are time-sliced by the hardware scheduler (milliseconds per exchange instead of microseconds)
        os.environ.setdefault("SVOSLAM_MAILBOX", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libsvoslam_hip has no CPU fallback")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks HERE (one process per GPU under torch.distributed.run)
        # -- never measure one GPU and print it under --gpus N (VERDICT r03 missing 3)
        if args.emulate_rank:
            raise SystemExit("--emulate-rank runs ONE rank on one GPU: use --gpus 1")
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and not one_device:
            raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s); refusing to run fewer ranks than asked for" % (args.gpus, ndev))
        import socket
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print("bench.py: --gpus %d without WORLD_SIZE: launching %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1 and not one_device and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but %d visible GPU(s)" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = os.environ.get("SVOSLAM_FORCE_DIST") == "1"   # exercise the row-band/RCCL code path on one GPU
    if args.exchange is None:
        args.exchange = "deltas" if world > 1 else "none"
    if args.no_overlap and (args.exchange in ("deltas", "keyrange") or args.emulate_rank):
        raise SystemExit("--no-overlap has no frame-sharded form: use --exchange none")
    emu = None
    if args.emulate_rank:
        er, en = (int(x) for x in args.emulate_rank.split("/"))
        assert world == 1 and 0 <= er < en
        args.exchange = "keyrange" if args.exchange == "keyrange" else "deltas"
        emu = pl.EmulatedRank(er, en, exchange=args.exchange)
        dist = emu
    elif world > 1 or force_dist:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if one_device:
            tdist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            tdist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        if tdist.get_world_size() != world:
            raise SystemExit("bench.py: the process group has %d ranks, --gpus says %d" % (tdist.get_world_size(), world))
        dist = pl.DistContext(rank, world, force=force_dist, exchange=args.exchange)
    arch = pkg.device_arch()
    assert arch and arch.startswith("gfx950"), arch
    # library settings (include/svoslam.h svoslam_config; SVOSLAM_CONFIG presets them for A/B runs)
    if args.stages:
        pkg.configure(runner_timeline=1)
    if one_device_chain and "track_mode" not in os.environ.get("SVOSLAM_CONFIG", ""):
        pkg.configure(track_mode=1)
    cfg = pkg.get_config()

    width, height, max_depth, center, edge = WORKLOADS[args.workload]
    K, Wm = args.steps, args.warmup
    map_frames = args.map_frames if args.map_frames is not None else (300 if args.workload == "cfg3" else 0)
    pre = max(0, map_frames - (K + Wm))     # fused untimed before the warm-up
    total = pre + Wm + K
    single = world == 1 and emu is None and not force_dist
    stage_pass = single and not args.no_overlap and not args.no_stage_pass
    extra = 3 if stage_pass else 0          # frames of the sequential per-stage pass after the timed region
    # every rank generates the same stream (same seeds) directly in HBM
    depth, rgb = synth.render_stream(total + extra, width, height, device="cuda")
    views = [pl.ground_truth_view(k, synth) for k in range(total + extra)]
    mode = pkg.RENDER_REFERENCE if args.render_mode == "reference" else pkg.RENDER_CARRY
    strict = args.tracker == "reference"
    # pool reservation: room for the map at the end of the stream plus the worst-case reservation of the frames in flight (sum_d min(8^d, n)
    # splits per frame), so that no fusion waits for a size readback: 2^29 nodes (4 GiB; + 4 GiB of shadow words) for cfg3's 279 M-node
    # map, the whole 30-bit index range of the node format (8 GiB + 8) for 1080p frames
    pool_cap = (1 << 29) if width * height <= 400000 else (1 << 30) - 8
    P = pl.SlamPipeline(width, height, max_depth, center, edge, render_mode=mode, dist=dist, count_steps=True, strict_reference=strict,
                        pool_capacity_nodes=pool_cap)

    per_rank = None
    if emu is not None:   # the records the other ranks would deliver, for the whole stream
        per_rank = max(1, 16 // emu.world)
        dcam = pkg.Camera(width, height, P.focal, P.focal)
        if not strict:
            dcam.set_strict_reference(False)
        table = torch.zeros((total, pkg.DELTA_FLOATS), dtype=torch.float32, device="cuda")   # (no stage pass with an emulated rank)
        for k in range(1, total):
            dcam.pair_delta(depth[k - 1], rgb[k - 1], depth[k], rgb[k], table[k])
        torch.cuda.synchronize()
        tab_k = tab_i = None
        if P.shard_sort:   # what the owners of the other frames would all-gather after sorting them
            tab_k = torch.empty((total, width * height), dtype=torch.int64, device="cuda")
            tab_i = torch.empty((total, width * height), dtype=torch.int32, device="cuda")
            scam, sws = pkg.Camera(width, height, P.focal, P.focal), pkg.Workspace()
            if not strict:
                scam.set_strict_reference(False)
            for k in range(total):
                scam.apply_delta(table[k], k)
                pkg.svo_fuse_sort_frame(sws, depth[k], scam.fusion_transform_ptr(), P.focal, P.focal, max_depth, center, edge)
                pkg.svo_fuse_export_sorted(sws, width * height, tab_k[k], tab_i[k])
            torch.cuda.synchronize()
        kr_bytes = None
        if args.exchange == "keyrange":   # the deltas the other ranks would all-gather, frame by frame, from a pool fused in one piece
            assert tab_k is not None
            kr_deltas, kr_shared, kr_bytes = pl.keyrange_delta_table(tab_k, tab_i, rgb, pre, emu.rank, emu.world, max_depth, pool_cap)
            emu.expect_keyrange(kr_deltas)    # (None for the untimed history before frame `pre`: committed in one piece)

    def expect(lo, hi):
        if emu is not None:
            emu.expect(table[lo:hi], lo, per_rank, tab_k[lo:hi] if tab_k is not None else None, tab_i[lo:hi] if tab_i is not None else None)

    def barrier():
        if (world > 1 or force_dist) and emu is None:
            import torch.distributed as tdist
            tdist.barrier()
        torch.cuda.synchronize()

    # Initialisation of the runtime, independent of --warmup: the library records its launch sequences as HIP graphs
    # the first time it sees a set of buffers (3 map sets x 2 workspaces: six frames cover all of them), the streams
    # and the second workspace are created.  The map and the tracker are then reset (allocations and graphs stay), so
    # the W warm-up and K timed frames below start from an empty map exactly as they would without this pass.
    if not args.no_overlap:
        ninit = min(6, total)
        expect(0, ninit)
        P.run_stream(depth[:ninit], rgb[:ninit], list(range(ninit)), views[:ninit])
        barrier()
        P.reset()

    cur = {"P": P}

    def history(upto):
        """the map's history (untimed) and the warm-up through the same streamed path as the timed region: frames [0, upto)
        from an EMPTY map and a fresh tracker (allocations, streams and recorded launch graphs are kept)"""
        P = cur["P"]
        P.reset()
        if args.no_overlap:
            for k in range(upto):
                P.frame(depth[k], rgb[k], k, views[k])
        else:
            a = min(pre, upto)
            if a:
                expect(0, a)
                P.run_stream(depth[:a], rgb[:a], list(range(a)), views[:a])
                barrier()
            if upto > a:
                expect(a, upto)
                P.run_stream(depth[a:upto], rgb[a:upto], list(range(a, upto)), views[a:upto])
        barrier()

    def timed_window(first, include_h2d=False):
        """frames [first, total) timed between barrier + synchronize on both sides; the state before it is history(first)"""
        P = cur["P"]
        history(first)
        P.counters.zero_()
        barrier()
        # HIP events around each trace kernel and each tracker launch (group), recorded by the library on the launch streams
        pkg.stage_timing([pkg.STAGE_MARCH, pkg.STAGE_TRACKER])
        h_depth = h_rgb = None
        if include_h2d:     # the timed frames leave the device; what stays are pinned host copies
            h_depth, h_rgb = depth[first:total].cpu().pin_memory(), rgb[first:total].cpu().pin_memory()
            depth[first:total].zero_(); rgb[first:total].zero_()
            barrier()
        t0 = time.perf_counter()
        if include_h2d:     # enqueued ahead of the frame loop on the same stream (not overlapped: an upper bound of its cost)
            depth[first:total].copy_(h_depth, non_blocking=True)
            rgb[first:total].copy_(h_rgb, non_blocking=True)
        if args.no_overlap:
            for k in range(first, total):
                P.track(depth[k], rgb[k], k)
                if dist is None:
                    P.fuse_frame(depth[k], rgb[k])     # the kernels of the frame loop (fused front end), one after the other
                else:
                    P.backproject(depth[k])
                    P.fuse(rgb[k])
                P.render(views[k])
        else:
            # the HIP streams of the frame loop (pipeline.run_stream); every frame still goes through
            # track -> back-project -> fuse -> render with the same results
            expect(first, total)
            P.run_stream(depth[first:total], rgb[first:total], list(range(first, total)), views[first:total])
        barrier()
        el = time.perf_counter() - t0
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        if world > 1:
            import torch.distributed as tdist
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        st, lv = (int(x) for x in P.counters.cpu().tolist())
        runner_ = getattr(P, "_runner", None)
        rec = {"timeline": runner_.timeline() if (args.stages and runner_ is not None and not args.no_overlap) else None,
               "elapsed": float(t.item()), "steps": st, "levels": lv, "march": pkg.stage_timing_read(pkg.STAGE_MARCH),
               "tracker": pkg.stage_timing_read(pkg.STAGE_TRACKER),
               "marches": P.marched_last_call if getattr(P, "frame_sharded", False) else total - first}
        pkg.stage_timing([])
        return rec

    # The K-frame window is measured R times, each time from an EMPTY map: reset, the 300 - K - W frames of history, the W
    # warm-up frames, barrier, K timed frames (the whole stream is ~0.15 s of GPU time).  `value` is the MEDIAN run (VERDICT
    # r03: the schedule has two steady states and one window is a coin flip); every run is listed in `runs`.
    t0w = pre + Wm                                # first timed frame
    R = max(1, args.repeats)
    runs = [timed_window(t0w, args.include_h2d) for _ in range(R)]
    order = sorted(range(R), key=lambda i: runs[i]["elapsed"])
    med = runs[order[(R - 1) // 2]]               # (an actual run: its kernel timings go with it)
    elapsed = med["elapsed"]
    # the cold pipeline's fill: the window starts on an idle device (the contract's barrier), so its first frames pay the
    # latency of the five-stream schedule.  T(K) = fill + K x period, measured with a second window over the last K / 2
    # frames of the same stream (same map at the end): period = (T(K) - T(K/2)) / (K - K/2), fill = T(K) - K x period.
    fill = None
    if R > 1 and K >= 8 and not args.no_overlap and not args.include_h2d and not args.lean:
        half = sorted(timed_window(t0w + K // 2)["elapsed"] for _ in range(3))[1]
        period = (elapsed - half) / (K // 2)
        if period > 0:
            fill = {"pipeline_fill_ms": (elapsed - K * period) * 1e3, "steady_period_ms": period * 1e3,
                    "steady_frames_per_s": 1.0 / period,
                    "how": "T(K) = fill + K x period from the median window of K = %d frames and the median of three windows over its last "
                           "%d frames; the contract's barrier before the timed frames empties the pipeline, so the fill is INSIDE `value`" % (K, K - K // 2)}

    # ---- what the map and the tracker look like when the timed region ends (say what the tracker does)
    def pool_i32(Pp=None):
        """the pool's node words as a device tensor (int32 view of the uint32 words; no copy)"""
        Pp = Pp or P
        class _Words:
            __cuda_array_interface__ = {"shape": (2 * Pp.pool.size,), "typestr": "<i4", "data": (Pp.pool.data_ptr, False), "version": 2}
        return torch.as_tensor(_Words(), device="cuda")

    def pose_error(frame, Pp=None):
        """estimated sensor pose of `frame` against the generator's ground truth, both in the map frame (= camera frame of
        frame 0): main.cpp:40 maps a camera point x to orientation * (x + position)"""
        p, o = (Pp or P).cam.pose()
        M = o.reshape(3, 3).T.astype(np.float64)                       # column-major mat3 -> M[row][col]
        (p0, yaw0), (pk, yawk) = synth.camera_pose(0), synth.camera_pose(frame)
        th = yawk - yaw0
        Rgt = np.array([[np.cos(th), 0.0, np.sin(th)], [0.0, 1.0, 0.0], [-np.sin(th), 0.0, np.cos(th)]])
        c0, s0 = np.cos(yaw0), np.sin(yaw0)
        d = np.array(pk) - np.array(p0)
        eye = np.array([c0 * d[0] - s0 * d[2], d[1], s0 * d[0] + c0 * d[2]])
        E = M @ Rgt.T
        ang = float(np.degrees(np.arccos(np.clip((np.trace(E) - 1.0) / 2.0, -1.0, 1.0))))
        return ang, float(np.linalg.norm(M @ p.astype(np.float64) - eye))

    def end_facts(Pp):
        words = pool_i32(Pp)
        alpha = (words[1::2] >> 24) & 0xFF
        err_deg, err_m = pose_error(total - 1, Pp)
        return {"pool_nodes_end": Pp.pool.size, "saturated_nodes_end": int((alpha >= 254).sum().item()),
                "image_coloured_pixels_end": int((Pp.image[..., :3].amax(-1) > 0).sum().item()),
                "pose_error_deg_end": err_deg, "pose_error_m_end": err_m, "tracking_lost_levels": Pp.cam.tracking_lost_count()}

    end_state = None
    if rank == 0:
        end_state = end_facts(P)
        end_state["tracker"] = ("reference: RGBDCamera::update with every quirk; its rotational Jacobian (Q14, localization_kernels.cu:207-213: not "
                                "v x n) drifts by degrees per frame on clean data -- pose_error_*_end is against the generator's ground truth after "
                                "%d frames; tracking_lost_levels counts abandoned ICP levels and is NOT a health indicator" % total) if strict else \
                               "CORRECTED (own specification, svoslam_camera_set_strict_reference(cam, 0)): NOT the reference's tracker, not the headline"

    # ---- live kernel timings of the timed region: march and tracker (HIP events on their launch streams)
    steps, levels = med["steps"], med["levels"]
    march_total_ms, march_launches = med["march"]
    trk_total_ms, trk_launches = med["tracker"]
    marches = med["marches"]                              # frame-sharded: this rank ray-marches its own frames only
    assert march_launches == marches and marches > 0, (march_launches, marches)
    kern_ms = march_total_ms / marches
    rows = P.rows
    n0 = width * height
    march_alg = (4.0 * (levels + steps) + 4.0 * width * rows * marches) / marches     # per launch (this rank's band / frames)
    # ICP reads 48 B per pixel per iteration, 10 / 5 / 4 iterations on the three pyramid levels = 552 N0
    icp_alg = 48.0 * (10 * n0 + 5 * (n0 // 4) + 4 * (n0 // 16))
    chain = cfg["track_mode"] == 1 or (n0 > 640 * 480 and cfg["track_stream"] == 0)
    one_launch = not chain
    streaming = one_launch and n0 > 640 * 480
    trk_kernel = ("track_persistent_kernel (streaming form)" if streaming else "track_persistent_kernel") if one_launch else \
        "icp_accumulate_work_kernel + cam_reduce_solve_kernel (38 launches)"
    trk_ms = trk_total_ms / trk_launches if trk_launches else None

    # ---- HBM traffic per frame from the committed rocprofv3 PMC passes (cannot be sampled from inside the process)
    traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    want_traffic = world == 1 and emu is None and not force_dist

    def stage_traffic(stage):
        """(bytes per frame, source) of one stage = sum over its kernels of calls_per_frame * (2 FETCH + WRITE); raises
        when the committed file does not cover the workload (a silent null hides a broken measurement)"""
        if not want_traffic:
            return None, "not applicable: the PMC passes are single-GPU runs"
        try:
            tj = json.load(open(traffic_file))
            ent = tj[args.workload]["stages"][stage]
            tot = 0.0
            for kname, kv in ent["kernels"].items():
                tot += kv["calls_per_frame"] * (2.0 * kv["fetch_kb"] + kv["write_kb"]) * 1024.0
            if not ent["kernels"]:
                raise KeyError("no kernels")
            return tot, "not sampled in this run: calls per frame x (2 x FETCH_SIZE + WRITE_SIZE) of %s" % tj[args.workload]["source"]
        except (OSError, KeyError, ValueError, TypeError) as e:
            if args.allow_missing_traffic:
                return None, "MISSING (%s: %r)" % (os.path.relpath(traffic_file, ROOT), e)
            raise SystemExit("bench.py: %s has no complete '%s' / stage '%s' entry (%r): run tools/prof/profile_round.sh, or pass "
                             "--allow-missing-traffic" % (traffic_file, args.workload, stage, e))

    def roof(stage, kernel, alg_bytes, ms, launches_per_frame, note):
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        tr, src = stage_traffic(stage)
        return {"stage": stage, "bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "peak_measured": HBM_MEASURED_COPY_GBS, "frac_of_measured": achieved / HBM_MEASURED_COPY_GBS, "traffic": tr, "traffic_source": src, "alg_bytes_per_launch": alg_bytes,
                "kernel_ms": ms, "launches_per_frame": launches_per_frame, "limiter": note}

    # which march runs: pools fused to depth <= 14 are marched over occupancy bricks in reference mode (csrc/pool_grid.hpp)
    bricks = max_depth <= 14 and args.render_mode == "reference" and cfg["march_bricks"] != 0
    march_kernel = "cone_trace_brick_kernel" if bricks else "cone_trace_kernel"   # (its template parameter B = 3 when svoslam_config.march_ahead >= 0: bursts)
    roofs = [roof("march", march_kernel, march_alg, kern_ms, 1,
                  ("instruction issue: a step is ONE memory round trip (brick entry + level-grid entry requested together, mostly L1 / L2 "
                   "hits: counter traffic is a few percent of the algorithmic bytes) and ~160 instructions; a lone wavefront of the tail "
                   "issues them in ~0.5 us, the full chip is VALU-bound (profiles/r03_brick_march_anatomy.txt); not HBM bandwidth") if bricks else
                  ("dependent chain of the longest rays (~1.0 us per march step on the critical path; the walk is L2-resident: "
                   "counter traffic is a few percent of the algorithmic bytes), not HBM bandwidth"))]
    roofs[0].update({"steps_per_launch": steps / marches, "levels_per_launch": levels / marches, "timed": "live, timed region"})
    if trk_ms:
        roofs.append(roof("tracker", trk_kernel, icp_alg, trk_ms, 1 if one_launch else 38,
                          ("one launch for the 19 iterations, the two finer levels streamed through work maps (72 B per pixel and iteration) on "
                           "176 workgroups: HBM bandwidth of the ten 1080p iterations + 19 hand-offs") if streaming else
                          ("one launch for the 19 iterations: pixels live in registers (48 B per pixel read once per LEVEL, not per "
                           "iteration); bound by the 19 cross-workgroup fan-in / solve / broadcast hand-offs (~11 us each)") if one_launch else
                          "launch chain with work maps: 38 dependent launches beside the march"))
        roofs[-1]["timed"] = "live, timed region (%d tracked frames)" % trk_launches

    # ---- sequential per-stage pass over three MORE frames (after the timed region): the fusion's launches, the maps
    stage_seq = None
    if stage_pass:
        fus = {"sort": [0.0, 0], "plan": [0.0, 0], "commit": [0.0, 0], "maps": [0.0, 0], "tracker": [0.0, 0]}
        alg_f, marches_seq, floors_f = [], [], []
        lo = torch.tensor(center, device="cuda", dtype=torch.float32) - edge
        for k in range(total, total + extra):
            torch.cuda.synchronize()
            pkg.stage_timing([pkg.STAGE_FUSE_SORT, pkg.STAGE_FUSE_PLAN, pkg.STAGE_FUSE_COMMIT, pkg.STAGE_MAPS, pkg.STAGE_MARCH, pkg.STAGE_TRACKER])
            P.track(depth[k], rgb[k], k)
            size0 = P.pool.size
            P.fuse_frame(depth[k], rgb[k])
            P.render(views[k])
            torch.cuda.synchronize()
            for nm, st in (("sort", pkg.STAGE_FUSE_SORT), ("plan", pkg.STAGE_FUSE_PLAN), ("commit", pkg.STAGE_FUSE_COMMIT), ("maps", pkg.STAGE_MAPS),
                           ("tracker", pkg.STAGE_TRACKER)):
                ms, n = pkg.stage_timing_read(st)
                fus[nm][0] += ms; fus[nm][1] += 1
            marches_seq.append(pkg.stage_timing_read(pkg.STAGE_MARCH)[0])
            pkg.stage_timing([])
            # algorithmic bytes of this fusion: V 15 + V 4 D + U_D 8 + sum_l K_l 72 + sum_{l<D} U_l 36
            splits = (P.pool.size - size0) // 8
            P.backproject(depth[k])                       # (statistics only: the frame loop keeps no point cloud)
            q = P.points.view(-1, 3)
            q = q[torch.isfinite(q[:, 0]) & torch.isfinite(q[:, 2])]
            V = int(q.shape[0])
            U = []
            for l in range(1, max_depth + 1):
                cell = torch.clamp(((q - lo) / (2 * edge / (1 << l))).floor().long(), 0, (1 << l) - 1)
                U.append(int(torch.unique((cell[:, 0] << 40) | (cell[:, 1] << 20) | cell[:, 2]).numel()))
            alg_f.append(V * 15.0 + V * 4.0 * max_depth + U[-1] * 8.0 + splits * 72.0 + 36.0 * sum(U[:-1]))
            # ---- sector floor of THIS build's fusion: the distinct 64-byte sectors its kernels must move, from the
            # frame's own counts.  A tile of eight sibling nodes IS one 64-byte sector (node index = 8 x tile), T_l = U_(l-1) distinct
            # tiles hold the touched nodes of level l (U_0 = 1).  Streams are dense (n = width x height lanes of 8-byte packed keys).
            n_px = width * height
            tiles = [1] + U[:-1]                      # T_1 .. T_D
            key_bits = 3 * max_depth + 1
            digit = 11 if n_px <= (4 << 20) else 9   # radix_packed_digit_bits_for
            passes = -(-key_bits // digit)
            floors_f.append({
                "front": 2.0 * n_px + 8.0 * n_px,                               # depth in, packed keys out
                "sort": passes * 24.0 * n_px,                                    # per pass: keys read by the count, read + written by the scatter
                "plan": 16.0 * n_px + 64.0 * sum(tiles) + 12.0 * splits,        # keys read by count and emit, one walk's tiles, the split records
                "commit": 128.0 * splits + 11.0 * n_px + 128.0 * sum(tiles)})   # new tile + parent's sector per split; keys + colours in; every touched tile read + written once (leaf blend, mip)
        fuse_ms = sum(fus[nm][0] for nm in ("sort", "plan", "commit")) / extra
        fuse_alg = sum(alg_f) / len(alg_f)
        roofs.append(roof("fusion", "keys_packed + packed sort (12 launches) + plan (3) + split_all + fill_mip_local + mip_straddle",
                          fuse_alg, fuse_ms, 18, "18 short dependent launches of gathers into a multi-GB pool: dependent-load latency and launch "
                          "floors, not bandwidth"))
        roofs[-1]["timed"] = "sequential pass over %d frames after the timed region (sum of the event-bracketed launch groups)" % extra
        roofs[-1]["parts_ms"] = {nm: fus[nm][0] / extra for nm in ("sort", "plan", "commit")}
        fl = {k: sum(f[k] for f in floors_f) / len(floors_f) for k in floors_f[0]}
        roofs[-1]["floor_bytes"] = sum(fl.values())
        roofs[-1]["floor_parts"] = fl
        roofs[-1]["floor_is"] = ("distinct 64-byte sectors this build's fusion kernels must move, from the frame's own counts: front 10 n; sort passes x 24 n; "
                                 "plan 16 n + 64 sum_l T_l + 12 K; commit 128 K + 11 n + 128 sum_l T_l (n pixels, T_l = U_(l-1) touched tiles of level l, "
                                 "K splits); traffic_over_floor = counter traffic of the group's kernels over it")
        try:   # counter traffic per kernel group (profiles/pmc_traffic.json), against its floor
            with open(traffic_file) as f:
                kern = json.load(f)[args.workload]["stages"]["fusion"]["kernels"]
            grp = {"front": ("keys_packed",), "sort": ("packed_",), "plan": ("plan_",), "commit": ("split_", "fill_", "mip_")}
            tr_g = {g: sum(v["calls_per_frame"] * (2.0 * v["fetch_kb"] + v["write_kb"]) * 1024.0 for k, v in kern.items() if k.startswith(pre))
                    for g, pre in grp.items()}
            roofs[-1]["traffic_parts"] = tr_g
            roofs[-1]["traffic_over_floor"] = {g: tr_g[g] / fl[g] for g in fl}
            roofs[-1]["traffic_over_floor"]["all"] = sum(tr_g.values()) / sum(fl.values())
        except (OSError, KeyError, ValueError, TypeError, ZeroDivisionError):
            pass
        stage_seq = {"maps_ms": fus["maps"][0] / extra, "tracker_ms": fus["tracker"][0] / extra, "fuse_sort_ms": fus["sort"][0] / extra, "fuse_plan_ms": fus["plan"][0] / extra,
                     "fuse_commit_ms": fus["commit"][0] / extra, "march_ms": sum(marches_seq) / extra,
                     "note": "stages one after the other on an otherwise idle GPU, frames %d..%d" % (total, total + extra - 1)}
    # the dominant kernel.  Both are timed live with HIP events on their launch streams, and an event interval contains what the
    # launch WAITS for its turn on the device (the one-launch tracker's 151 workgroups queue behind the march's wavefronts: 0.31-0.35
    # ms between its events for a kernel that executes 0.25 ms, as rocprofv3 --kernel-trace shows: profiles/r03_bench_cfg3_kernel_
    # trace_timed_frames.txt).  The choice therefore follows the kernels' EXECUTION times -- the sequential pass below, each kernel
    # alone on the GPU, which ranks them as the rocprofv3 kernel statistics do -- and falls back to the live intervals when that
    # pass did not run; every number of the object stays the live one.
    dom = max(roofs[:2], key=lambda r: r["kernel_ms"])
    tie_note = "ALWAYS"
    roofline = None
    def make_roofline(dom, how):
        r = {k: dom[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "peak_measured", "frac_of_measured", "traffic", "traffic_source", "limiter",
                                 "alg_bytes_per_launch", "kernel_ms")}
        r["chosen_by"] = how + ": " + ", ".join("%s %.3f ms" % (q["kernel"].split(" ")[0], q["kernel_ms"]) for q in roofs[:2])
        return r
    roofline = make_roofline(dom, "largest mean launch duration among the kernels timed live")
    if dom["stage"] == "march":
        roofline.update({"steps_per_launch": steps / marches, "levels_per_launch": levels / marches})

    if tie_note and stage_seq and stage_seq.get("tracker_ms") and stage_seq.get("march_ms"):
        alone = {"march": stage_seq["march_ms"], "tracker": stage_seq["tracker_ms"]}
        dom = max(roofs[:2], key=lambda r: alone[r["stage"]])
        roofline = make_roofline(dom, "largest execution time alone on the GPU (march %.3f ms, tracker %.3f ms: stages_sequential; the "
                                      "order of the rocprofv3 kernel statistics); live event intervals, which include the wait for the device"
                                      % (alone["march"], alone["tracker"]))
    if dom["stage"] == "march":
        roofline.update({"steps_per_launch": steps / marches, "levels_per_launch": levels / marches})
    # stage durations from the scheduler's HIP-event marks (mean over the timed frames; stages overlap across streams)
    stages = None
    if med.get("timeline") is not None:
        tl = med["timeline"]            # (of the median window)
        if len(tl) == K and K > 8:
            a = tl[4:-2]
            stages = {"maps_ms": float((a[:, 1] - a[:, 0]).mean()), "tracker_ms": float((a[:, 3] - a[:, 2]).mean()),
                      "backproject_sort_ms_incl_waits": float((a[:, 5] - a[:, 4]).mean()), "plan_ms": float((a[:, 6] - a[:, 5]).mean()),
                      "commit_ms": float((a[:, 8] - a[:, 7]).mean()), "accel_build_plus_march_ms": float((a[:, 9] - a[:, 8]).mean()),
                      "frame_period_ms": float(np.diff(a[:, 9]).mean()),
                      "frame_latency_ms_maps_begin_to_march_end": float((a[:, 9] - a[:, 0]).mean())}
    # ---- frame latency (a SLAM consumer cares): the multi-stream schedule trades latency for rate.  One more
    # window on a pipeline whose runner records HIP-event marks at the stage boundaries (the marks cost ~6 % of the rate, which
    # is why the timed windows above run without them); mean over its frames of first map kernel -> end of the frame's march.
    latency = None
    if single and not args.no_overlap and not args.stages and not args.include_h2d and K > 8 and not args.lean:
        try:
            pkg.configure(runner_timeline=1)     # (read when the pipeline's runner is created: at its first stream call, inside the window)
            cur["P"] = pl.SlamPipeline(width, height, max_depth, center, edge, render_mode=mode, count_steps=True, strict_reference=strict,
                                       pool_capacity_nodes=pool_cap)
            timed_window(t0w)
            tl = cur["P"]._runner.timeline()
            if len(tl) == K:
                a = tl[4:-2]
                latency = {"frame_latency_ms": float((a[:, 9] - a[:, 0]).mean()), "frame_period_ms_with_marks": float(np.diff(a[:, 9]).mean()),
                           "what": "first map kernel of a frame to the end of its march, mean over the timed frames of one extra window with "
                                   "stage marks on (svoslam_config.runner_timeline)"}
        except Exception as e:
            latency = {"frame_latency_ms": None, "error": repr(e)}
        finally:
            pkg.configure(runner_timeline=0)
        cur["P"] = P
    # ---- single GPU, reference tracker: the same windows once more with the CORRECTED tracker, as a labelled second measurement
    # (with it the stream exercises alpha saturation, ray retirement through the bricks' A >= 254 bits,
    # fusion dominated by read-modify-writes of existing leaves).  Never `value`.
    corrected_line = None
    if single and strict and not args.no_overlap and not args.no_corrected_line and not args.include_h2d and not args.lean:
        try:
            cur["P"] = pl.SlamPipeline(width, height, max_depth, center, edge, render_mode=mode, count_steps=True, strict_reference=False,
                                       pool_capacity_nodes=pool_cap)
            r2 = [timed_window(t0w) for _ in range(min(R, 3))]
            e2 = sorted(r["elapsed"] for r in r2)[(len(r2) - 1) // 2]
            m2 = [r for r in r2 if r["elapsed"] == e2][0]
            corrected_line = {"what": "the same %d timed frames with svoslam_camera_set_strict_reference(cam, 0): this build's corrected tracker "
                                      "(own specification; include/svoslam.h)" % K,
                              "value": K / e2, "unit": "frames/s", "ms_per_step": e2 / K * 1e3, "runs": [K / r["elapsed"] for r in r2],
                              "march_kernel_ms": m2["march"][0] / max(1, m2["march"][1]), "march_steps_per_launch": m2["steps"] / max(1, m2["marches"]),
                              **end_facts(cur["P"])}
        except Exception as e:   # (in the record, not swallowed)
            corrected_line = {"value": None, "error": repr(e)}
        cur["P"] = P

    # ---- N > 1: the OTHER partition as well, same windows (a SCALE run then measures both what north_star
    # describes -- row bands, ICP all-reduce, all-gather of sorted band key lists -- and the frame-sharded scheme recommended here)
    other = None
    if world > 1 and emu is None and not args.no_other_partition and args.exchange in ("deltas", "allreduce") and not args.no_overlap:
        oex = "allreduce" if args.exchange == "deltas" else "deltas"
        try:
            d2 = pl.DistContext(rank, world, force=force_dist, exchange=oex)
            cur["P"] = pl.SlamPipeline(width, height, max_depth, center, edge, render_mode=mode, dist=d2, count_steps=True,
                                       pool_capacity_nodes=pool_cap)
            r2 = [timed_window(t0w) for _ in range(min(R, 3))]
            e2 = sorted(r["elapsed"] for r in r2)[(len(r2) - 1) // 2]
            other = {"exchange": oex, "value": K / e2, "unit": "frames/s", "ms_per_step": e2 / K * 1e3, "runs": [K / r["elapsed"] for r in r2],
                     "what": ("SURVEY 8e / north_star: image rows in %d bands, ICP sums all-reduced (19 per frame), sorted band key lists all-gathered "
                              "and merged, replicated pool" % world) if oex == "allreduce" else
                             "frames tracked + ray-marched by rank k %% %d, update_trans records all-gathered, every rank applies every fusion" % world}
        except Exception as e:   # the line still carries the primary scheme; the failure is IN the record, not swallowed
            other = {"exchange": oex, "value": None, "error": repr(e)}
        cur["P"] = P
    if rank == 0:
        out = {
            "metric": "SLAM frames/sec (fuse+ICP+raycast)", "value": K / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": elapsed / K * 1e3,
            "value_is": "median of %d windows of K = %d timed frames, each from an empty map (reset, %d frames of history, %d of warm-up)" % (R, K, pre, Wm),
            "runs": [K / r["elapsed"] for r in runs], "value_min": K / max(r["elapsed"] for r in runs),
            "value_max": K / min(r["elapsed"] for r in runs),
            # the schedule's two steady states: a run more than 7 % under the best one of this process fell into the slow one
            "runs_steady_state": ["fast" if min(q["elapsed"] for q in runs) / r["elapsed"] >= 0.93 else "slow" for r in runs],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32/f32 (ICP sums exact fixed-point in f64)", "data": "synthetic",
            "config": {"workload": "%s: synthetic %dx%d RGB-D stream, depth-%d SVO, half-edge %.3f m, bilateral+ICP(19 it)+fuse+raycast(%s mode)%s"
                                   % (args.workload, width, height, max_depth, edge, args.render_mode,
                                      "" if strict else " -- CORRECTED TRACKER (own specification, not the reference's): a labelled second line, not the headline"),
                       "parallelism": ("EMULATED rank %d of %d (one GPU; the other ranks' update_trans records, sorted arrays and key-range DELTAS precomputed): "
                                       "frames tracked, sorted and ray-marched by rank k %% N; every frame's plan + commit cut by key range, this rank's slice "
                                       "computed here, all ranks' deltas applied here" % (emu.rank, emu.world) if emu is not None and args.exchange == "keyrange" else
                                       "EMULATED rank %d of %d (one GPU; the other ranks' update_trans records precomputed): frames tracked "
                                       "and ray-marched by rank k %% N, every fusion applied here" % (emu.rank, emu.world) if emu is not None else
                                       "single GPU" if world == 1 and not force_dist else
                                       "frames tracked + ray-marched by rank k %% %d, all-gather of 80-byte update_trans records "
                                       "(one per chunk of frames), every rank applies every fusion to its replica" % world
                                       if args.exchange == "deltas" else
                                       "raycast in %d row bands; tracker + fusion on every rank, no data-path collective" % world
                                       if args.exchange == "none" else
                                       "%d row bands: ICP all-reduce (19 per frame) + point all-gather, replicated pool" % world),
                       "multi_gpu_scheme": (None if world == 1 and emu is None and not force_dist else
                                            "%s (bench.py's default for N > 1 is 'deltas': on ONE GPU, emulating one rank of N on the 300-frame map, "
                                            "it gives 1.2x / 1.7x / 2.1x of the single-GPU rate for N = 2 / 4 / 8, against < 1x for the row-band scheme "
                                            "'allreduce' = SURVEY 8e with sorted-key all-gather + merge (profiles/r03_bench_cfg3_emulated_*.json, "
                                            "r03_bench_cfg3_forced_dist_*.json); UNMEASURED on multi-GPU hardware)" % args.exchange),
                       "overlap": "none" if args.no_overlap else
                                  ("5 HIP streams: maps(k+2) | ICP(k+1) | back-project+sort+plan(k+1) | deferred commit(k+1) beside | apply+bricks+raycast(k)"
                                   if (width * height <= 400000 if cfg["runner_deferred"] < 0 else cfg["runner_deferred"] == 1) and world == 1 and emu is None and not force_dist
                                   else "4 HIP streams: maps(k+2) | ICP(k+1) | back-project+sort+plan(k+1) | commit+raycast(k)"),
                       "frames_in_map_at_end": total, "frames_fused_untimed_before_warmup": pre, "frames_input": "pinned host memory, uploaded inside the timed region" if args.include_h2d else "resident in HBM",
                       "raycast_views": "ground-truth sensor poses (the reference renders from a free GLFW camera)",
                       "mrays_per_s": width * rows / (kern_ms * 1e-3) / 1e6,
                       "device_memory_GiB": device_memory(P, torch), **(end_state or {})},
            "roofline": roofline,
            "roofline_stages": roofs,
        }
        if stage_seq:
            out["stages_sequential"] = stage_seq
        if stages:
            out["stages"] = stages
        if fill:
            out["pipeline_fill"] = fill
        if other:
            out["other_partition"] = other
        if args.exchange == "keyrange" and hasattr(P, "_kr"):
            P.keyrange_check()          # (raises if an apply was refused that the schedule did not expect)
            kb = kr_bytes[t0w:total] if emu is not None and kr_bytes is not None else None
            out["keyrange"] = {
                "splitter_level": 3, "ranks": emu.world if emu is not None else world,
                "frames_timed": K, "frames_timed_with_records_above_the_splitter_level": (sum(1 for k in range(t0w, total) if kr_shared[k]) if emu is not None else None),
                "delta_bytes_all_ranks_per_frame_mean": (sum(sum(b) for b in kb) / len(kb)) if kb else None,
                "delta_bytes_all_ranks_per_frame_max": max(sum(b) for b in kb) if kb else None,
                "delta_bytes_this_rank_per_frame_mean": (sum(b[emu.rank] for b in kb) / len(kb)) if kb else None,
                "all_gather": "ONE per frame, between svoslam_svo_fuse_keyrange_commit and _apply; an emulated rank takes the other ranks' deltas from a "
                              "table produced before the timed region (pipeline.keyrange_delta_table): the all-gather itself is NOT in this number",
                "hardware": "UNMEASURED on multi-GPU hardware: one rank of N on one GPU"}
        if corrected_line:
            out["corrected_tracker"] = corrected_line
        if latency:
            out["latency"] = latency
        # the CPU leg's starting point, read while the pipeline still exists: the map and the pose the timed frames started from
        seed_words = seed_pose = None
        if not args.no_cpu_baseline:
            history(t0w)
            seed_words = pool_i32().cpu().numpy().view(np.uint32).copy() if t0w > 0 else None
            seed_pose = P.cam.pose()
        wall["gpu_legs_s"] = time.perf_counter() - t_start
        # ---- the other BASELINE configurations on the same line: cfg4 = this script in a child process
        # (3 windows of 40 frames, per-stage rooflines, a short CPU leg), cfg2 / cfg5 = tools/mesh_bench.py (voxelize, SVO build,
        # renders in both modes).  GPU legs, ahead of this process's CPU oracle -- and AFTER this process has given its pool
        # reservation, shadow array and brick field back (the children measured beside 40 GiB held by an idle parent).
        if single and args.workload == "cfg3" and not args.no_other_configs and not args.lean and not args.no_overlap and not args.include_h2d:
            t_o0 = time.perf_counter()
            cur["P"] = None
            P.close()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            out["other_configs"] = other_configs(args)
            wall["other_configs_s"] = time.perf_counter() - t_o0
        if not args.no_cpu_baseline:
            t_cpu0 = time.perf_counter()
            out["cpu_baseline"] = cpu_baseline(depth[:total], rgb[:total], views, t0w, width, height, max_depth, center, edge, seed_words, seed_pose,
                                               budget_s=args.cpu_budget, min_frames=args.cpu_min_frames)
            wall["cpu_baseline_s"] = time.perf_counter() - t_cpu0
        wall["total_s"] = time.perf_counter() - t_start
        wall["note"] = ("wall clock of this process: gpu_legs_s = import + stream generation + every GPU window of this workload; other_configs_s = "
                        "the child processes (GPU); cpu_baseline_s = the single-thread CPU oracle (the GPU idles: a 5-second SMI sample taken "
                        "there reads 0 % busy)")
        out["wall"] = wall
        if args.full_line or os.environ.get("SVOSLAM_BENCH_FULL_LINE") == "1":
            print(json.dumps(out), flush=True)
        else:
            # the contract's ONE line, <= LINE_LIMIT bytes; the full record goes beside it (and to stderr-free disk only)
            dp = write_details(out, args.details) if args.details else None
            print(compact_line(out, os.path.relpath(dp, ROOT) if dp else None), flush=True)
    if world > 1 or force_dist:
        import torch.distributed as tdist
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
