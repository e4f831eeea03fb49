"""rocprofv3 --pmc counter_collection CSVs -> profiles/pmc_traffic.json (what bench.py's `traffic` fields read).

  python tools/prof/pmc_to_json.py OUT.json TAG  WORKLOAD:FRAMES:TIMED:FETCH.csv:WRITE.csv[:TRACKER_FETCH.csv:TRACKER_WRITE.csv:TRACKED] ...

FRAMES = frames the profiled bench processed (map history + warm-up + timed), TIMED = its timed frames.  Per kernel:
calls per frame = dispatches / FRAMES, and the counter mean is taken over the dispatches of the TIMED frames only (the
last calls_per_frame x TIMED rows in dispatch order) -- the traffic of the march, the planner and the leaf kernel grows
with the age of the map, and the bench times the last frames of the 300-frame config.  Optional second pair: a pass
over tools/prof/track_only.py for the one-launch tracker (the bench pass runs the launch chain: counter collection
serialises dispatches, which the sequential form of the bench tolerates)."""
import collections
import csv
import json
import sys

STAGES = {
    "march": ("cone_trace_brick_ahead_kernel", "cone_trace_brick_kernel", "cone_trace_kernel"),
    "march_accel": ("pool_refresh_kernel", "pool_grid_update_kernel", "pool_grid_build_kernel", "build_tables_kernel", "build_accel_kernel",
                    "brick_rebuild_kernel", "brick_clear_kernel"),
    "tracker": ("track_persistent_kernel", "icp_accumulate_work_kernel", "icp_accumulate_kernel", "cam_reduce_solve_kernel", "cam_frame_end_kernel"),
    "fusion": ("keys_packed_kernel", "packed_upsweep_kernel", "packed_column_scan_kernel", "packed_downsweep_kernel", "plan_count_kernel",
               "plan_scan_finish_kernel", "plan_emit_kernel", "split_all_kernel", "fill_mip_local_kernel", "mip_straddle_kernel",
               "mip_straddle2_kernel", "commit_apply_kernel"),
    "maps": ("bilateral_kernel", "vertex_normal_kernel", "subsample_depth_kernel"),
}


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("svoslam::", "")
    return n.split("<")[0]


def read(path, counter):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            rows[short(r["Kernel_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(vs)] for k, vs in rows.items()}


def main():
    out_path, tag = sys.argv[1], sys.argv[2]
    out = {"_comment": "HBM traffic per kernel from rocprofv3 PMC passes (one counter per pass, --kernel-include-regex svoslam, sequential form of "
                       "bench.py); KB per dispatch = mean over the dispatches of the TIMED frames. bench.py reports, per stage, traffic = sum over "
                       "its kernels of calls_per_frame x (2 x fetch_kb + write_kb) x 1024 bytes: FETCH_SIZE is doubled as MI355X_MICROARCH.md "
                       "prescribes for gfx950 (calibrated there on wide coalesced reads; 8-byte gathers make the factor an upper bound)."}
    for spec in sys.argv[3:]:
        f = spec.split(":")
        wl, frames, timed, fetch_csv, write_csv = f[0], int(f[1]), int(f[2]), f[3], f[4]
        fe, wr = read(fetch_csv, "FETCH_SIZE"), read(write_csv, "WRITE_SIZE")
        trk = None
        if len(f) >= 8:
            trk = (read(f[5], "FETCH_SIZE"), read(f[6], "WRITE_SIZE"), int(f[7]))
        entry = {"source": "profiles/%s_pmc_fetch_write_per_kernel.txt" % tag, "frames_profiled": frames, "timed_frames": timed, "stages": {}}
        for stage, names in STAGES.items():
            ks = {}
            for n in names:
                if n == "track_persistent_kernel":
                    if trk and n in trk[0] and n in trk[1]:
                        a, b = trk[0][n], trk[1][n]
                        ks[n] = {"calls_per_frame": 1.0, "fetch_kb": sum(a) / len(a), "write_kb": sum(b) / len(b), "dispatches": len(a),
                                 "from": "tools/prof/track_only.py pass (one launch per tracked frame)"}
                    continue
                if n not in fe or n not in wr:
                    continue
                a, b = fe[n], wr[n]
                cpf = len(a) / float(frames)
                take = max(1, int(round(cpf * timed)))
                a2, b2 = a[-take:], b[-take:]
                ks[n] = {"calls_per_frame": cpf, "fetch_kb": sum(a2) / len(a2), "write_kb": sum(b2) / len(b2), "dispatches": len(a)}
            if stage == "tracker" and "track_persistent_kernel" in ks:   # the one-launch tracker is what the frame loop runs at this size
                ks = {"track_persistent_kernel": ks["track_persistent_kernel"]}
            entry["stages"][stage] = {"kernels": ks,
                                      "bytes_per_frame": sum(v["calls_per_frame"] * (2 * v["fetch_kb"] + v["write_kb"]) * 1024.0 for v in ks.values())}
        out[wl] = entry
    json.dump(out, open(out_path, "w"), indent=1)
    for wl in out:
        if wl != "_comment":
            print(wl, {s: round(v["bytes_per_frame"] / 1e6, 2) for s, v in out[wl]["stages"].items()}, "MB per frame")


if __name__ == "__main__":
    main()
