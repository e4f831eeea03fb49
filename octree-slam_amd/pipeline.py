"""One SLAM frame = the body of mainLoop() (src/main.cpp:31-84) with the tracker
call of main.cpp:35 enabled:

    camera_estimation_->update(rawFrame)                      RGBDCamera::update      rgbd_camera.cpp:53
    generateVertexMap(raw depth -> points_)                   main.cpp:39
    transformVertexMap(points_, mat4(orientation)*T(position)) main.cpp:40
    computePointCloudBoundingBox(points_)                     main.cpp:43
    scene_->addPointCloudToOctree(...) -> svoFromPointCloud   main.cpp:44 / octree.cpp:290
    cuda_renderer_->coneTraceSVO(svo, camera)                 main.cpp:56-58

Single GPU: frame() enqueues everything on the current stream with no host round trip
(asynchronous fusion; the pool size stays on the device); run_stream() processes a
sequence of frames on four HIP streams (map generation, ICP, fusion preparation, commit +
raycast) with the same results.

Several GPUs (one process per GPU, torch.distributed over RCCL): the image is cut
into row bands and each rank ray-marches its band of the output image against its
own full replica of the node pool.  Two ways to keep the replicas identical
(DistContext.exchange):
  "none"       every rank tracks and fuses the whole frame itself -- no collective in the
               frame loop.  The tracker and the fusion are chains of short dependent launches
               whose duration hardly depends on the pixel count, so a band saves almost nothing
               while 19 all-reduces per frame cost ~0.5 ms: this is the faster choice at
               640x480 and 1080p and the default.
  "allreduce"  SURVEY 8e: each rank accumulates the ICP normal equations of its band and the
               27 exact fixed-point sums are all-reduced (float64 sum: integer-valued, so every
               rank gets the same bits in any reduction order); each rank back-projects and
               transforms its band of points and the bands are all-gathered, after which every
               rank applies the same fusion.
  "deltas"     frame-parallel tracking (the default of bench.py for N > 1).  The ICP of frame k
               starts from the identity and reads the maps of frames k-1 and k only (rgbd_camera.cpp:100-168), so rank
               k % N tracks frame k (svoslam_camera_pair_delta) and ray-marches it; the 80-byte update_trans records
               are all-gathered (one small collective per chunk of frames), every rank composes the same pose chain
               (svoslam_camera_apply_delta) and applies every fusion to its own replica.  No stage runs on a band.
Either way the replicas stay byte-identical to the 1-GPU pool.
"""
import os

import numpy as np
import torch

import octree_slam_amd as pkg

FOV = 45.0  # glfw_camera_controller.h:13 default


def frame_shards(n, first_index, world, per_rank):
    """Frame-sharded schedule of n frames whose global indices start at first_index: frame i belongs to rank
    (first_index + i) % world, which tracks it against frame i-1 and ray-marches it.  The frames are cut into chunks of
    world * per_rank; returns [(begin, end, slots)] with slots[i - begin] = (owner rank, row of that rank's delta block)."""
    chunks, size = [], world * per_rank
    for a in range(0, n, size):
        b = min(n, a + size)
        rows = [0] * world
        slots = []
        for i in range(a, b):
            r = (first_index + i) % world
            slots.append((r, rows[r]))
            rows[r] += 1
        chunks.append((a, b, slots))
    return chunks


def band_rows(height, rank, world):
    """Contiguous row band [first, first+rows) of rank `rank`."""
    base, rem = divmod(height, world)
    first = rank * base + min(rank, rem)
    rows = base + (1 if rank < rem else 0)
    return first, rows


class DistContext:
    """Thin torch.distributed wrapper (backend nccl == RCCL on ROCm; gloo for the CPU tests)."""

    def __init__(self, rank=0, world=1, group=None, force=False, exchange="none"):
        """exchange = "none": every rank tracks and fuses the whole frame itself (no collective in the frame loop; only
        the raycast is split into row bands).  exchange = "allreduce": SURVEY 8e -- ICP accumulation and back-projection
        per row band, 19 all-reduces of the 27 normal-equation sums and one all-gather of the point bands per frame."""
        assert exchange in ("none", "allreduce", "deltas", "keyrange")
        self.rank, self.world, self.group, self.force, self.exchange = rank, world, group, force, exchange
        if self.enabled:
            # create the RCCL communicator and its streams NOW (first use is lazy and was observed to
            # mis-order against kernels queued around it), then drain the device once
            import torch.distributed as dist
            dev = "cuda" if torch.cuda.is_available() else "cpu"   # cpu: the gloo tests
            t = torch.zeros(27, dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            g = torch.empty(self.world * 4, dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(g, torch.zeros(4, dtype=torch.float32, device=dev), group=self.group)
            if dev == "cuda":
                torch.cuda.synchronize()
        # Small records (the 27 ICP sums, the 80-byte pose records) go through the peer-to-peer mailbox instead of a collective
        # (csrc/mailbox.hip): a rank stores its record straight into every peer's inbox and polls its own -- no host-issued
        # collective, no ~25 us of RCCL latency per 216 bytes.  OPT-IN (SVOSLAM_MAILBOX=1) since round 4: it has only ever run
        # between processes on one device, so the default is torch.distributed for everything (the required baseline:
        # ncclAllReduce / ncclAllGather) until it has been validated across real devices.  A wait that gives up
        # poisons its output and sets a sticky flag that check_mailbox() turns into an exception at the end of every stream
        # call.  The inbox handles travel once, through the process group.
        # The mailbox has only ever run between processes on ONE device (no multi-GPU node was available to this build): its
        # set-up is therefore guarded -- every rank reports whether it could map its peers' inboxes, a first all-gather of
        # rank-stamped records is checked against the collective's, and unless EVERY rank succeeded the session falls back
        # to torch.distributed for these records too (a note on stderr says so).
        self.mailbox = None
        if self.enabled and torch.cuda.is_available() and os.environ.get("SVOSLAM_MAILBOX", "0") == "1":
            import sys
            import torch.distributed as dist
            backend = dist.get_backend(self.group)
            dev = "cuda" if backend == "nccl" else "cpu"

            def all_agree(ok):
                f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(f, op=dist.ReduceOp.MIN, group=self.group)
                return int(f.item()) == 1

            mb, why = None, ""
            try:
                mb = pkg.Mailbox(self.rank, self.world)
                mine = torch.from_numpy(mb.handle()).to(dev)
                have = True
            except Exception as e:                      # (e.g. hipIpcGetMemHandle refused)
                mine, have, why = torch.zeros(64, dtype=torch.uint8, device=dev), False, repr(e)
            allh = torch.empty((self.world, 64), dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(allh.view(-1), mine, group=self.group)
            if all_agree(have):
                try:
                    mb.connect(allh.cpu().numpy())
                    connected = True
                except Exception as e:                  # (e.g. no peer access between the two devices)
                    connected, why = False, repr(e)
                if all_agree(connected):
                    dist.barrier(group=self.group)   # every inbox is mapped before anyone posts
                    try:                              # first use, checked against the collective
                        rec = torch.full((1, pkg.DELTA_FLOATS), float(self.rank + 1), dtype=torch.float32, device="cuda")
                        got = torch.zeros((self.world, 1, pkg.DELTA_FLOATS), dtype=torch.float32, device="cuda")
                        mb.all_gather(rec, got)
                        torch.cuda.synchronize()
                        want = torch.arange(1, self.world + 1, dtype=torch.float32, device="cuda").view(-1, 1, 1).expand_as(got)
                        good = bool(torch.equal(got, want)) and not mb.failed()
                    except Exception as e:
                        good, why = False, repr(e)
                    if all_agree(good):
                        self.mailbox = mb
                    else:
                        why = why or "first all-gather through the mailbox differs from the expected records"
            if self.mailbox is None:
                if self.rank == 0:
                    print("svoslam: peer-to-peer mailbox unavailable (%s): small records go through torch.distributed" % (why or "a peer failed"),
                          file=sys.stderr, flush=True)
                if mb is not None:
                    try:
                        mb.close()
                    except Exception:
                        pass

    @property
    def enabled(self):
        return self.world > 1 or self.force

    def check_mailbox(self):
        """raises when a wait of the peer-to-peer mailbox has given up (a peer that posted late or never): the records built
        from it are poisoned, the session must not go on silently (blocks until the device is idle)"""
        if self.mailbox is not None and self.mailbox.failed():
            raise pkg.SvoslamError("peer-to-peer mailbox: a wait for a peer's record timed out (rank %d of %d); the session's "
                                   "ICP sums / pose records are invalid from that exchange on" % (self.rank, self.world))

    def all_reduce_sum(self, t):
        if self.mailbox is not None and t.is_cuda and t.dtype == torch.float64 and t.numel() <= 256:
            self.mailbox.all_reduce_f64(t)       # added in rank order from the gathered records: the same bits on every rank
        elif self.enabled:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_gather_deltas(self, out, mine):
        """out[world, m, DELTA_FLOATS] <- every rank's mine[m, DELTA_FLOATS] (enqueued on the current stream)"""
        if self.mailbox is not None and mine.is_cuda and mine.numel() * mine.element_size() <= 2048:
            self.mailbox.all_gather(mine, out)
        elif self.enabled:
            import torch.distributed as dist
            dist.all_gather_into_tensor(out.view(-1), mine.view(-1), group=self.group)
        else:
            out.view(-1).copy_(mine.view(-1))
        return out

    def all_gather_sorted(self, out, mine):
        """out[world, m, n] <- every rank's mine[m, n] (sorted keys or point indices of the frames it owns)"""
        if self.enabled:
            import torch.distributed as dist
            dist.all_gather_into_tensor(out.view(-1), mine.view(-1), group=self.group)
        else:
            out.view(-1).copy_(mine.view(-1))
        return out

    def all_gather_keyrange(self, frame, mine):
        """key-range sharded fusion: every rank's delta buffer of global frame `frame` (rank order; `mine` at this rank's place) -- ONE
        all-gather of the buffers' capacity (a production exchange would ship the used words, word 10 of a delta: see DESIGN.md 7)"""
        if not self.enabled:
            return [mine]
        import torch.distributed as dist
        if getattr(self, "_kr_recv", None) is None or self._kr_recv[0].shape[1] != mine.numel():
            self._kr_recv = [torch.empty((self.world, mine.numel()), dtype=mine.dtype, device=mine.device) for _ in range(2)]
        out = self._kr_recv[frame & 1]
        dist.all_gather_into_tensor(out.view(-1), mine.view(-1), group=self.group)
        return [out[s] for s in range(self.world)]

    def all_gather_rows(self, full, height):
        """`full` is [height, ...]; each rank has filled its own band; returns with all bands filled.
        Bands may differ by one row, so gather into per-rank views of padded size."""
        if not self.enabled:
            return full
        import torch.distributed as dist
        base, rem = divmod(height, self.world)
        maxrows = base + (1 if rem else 0)
        first, rows = band_rows(height, self.rank, self.world)
        row_shape = tuple(full.shape[1:])
        send = torch.zeros((maxrows,) + row_shape, dtype=full.dtype, device=full.device)
        send[:rows] = full[first:first + rows]
        recv = torch.empty((self.world * maxrows,) + row_shape, dtype=full.dtype, device=full.device)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        for r in range(self.world):
            f, n = band_rows(height, r, self.world)
            full[f:f + n] = recv[r * maxrows:r * maxrows + n]
        return full


class EmulatedRank:
    """Rank `rank` of a `world`-rank frame-sharded session WITHOUT a process group (tests; bench.py --emulate-rank on a
    one-GPU box): the all-gather delivers this rank's own records and takes the other ranks' from `table` -- one record
    per frame of the next run_stream call, produced beforehand by svoslam_camera_pair_delta (what RCCL would deliver)."""
    enabled, exchange, force, group = False, "deltas", False, None

    def __init__(self, rank, world, exchange="deltas"):
        self.rank, self.world, self.exchange = rank, world, exchange
        self.kr_deltas = None
        self.table, self.first, self.per_rank, self.calls = None, 0, 1, 0
        self.sorted, self.sorted_calls = (None, None), [0, 0]

    @property
    def has_sorted(self):
        """the emulation was given the other ranks' sorted arrays (else the pipeline sorts every frame itself)"""
        return self.sorted[0] is not None

    def expect(self, table, first_index, per_rank, sorted_keys=None, sorted_idx=None):
        """records [n, DELTA_FLOATS] of the frames of the next call, whose first frame has global index first_index;
        sorted_keys / sorted_idx [n, pixels] (optional): what the owners of those frames would all-gather after sorting them"""
        self.table, self.first, self.per_rank, self.calls = table, first_index, per_rank, 0
        self.sorted = (sorted_keys, sorted_idx)
        self.sorted_calls = [0, 0]

    def expect_keyrange(self, deltas):
        """deltas[g][s]: rank s's delta of global frame g (None at this rank's place); deltas[g] = None: no table for frame g -- it is
        committed in one piece (the untimed history of a bench run).  pipeline.keyrange_delta_table; set once for the whole stream"""
        self.kr_deltas = deltas

    def all_gather_keyrange(self, frame, mine):
        row = self.kr_deltas[frame]
        return [mine if s == self.rank else row[s] for s in range(self.world)]

    def all_gather_sorted(self, out, mine):
        which = 0 if out.dtype == torch.int64 else 1
        tab = self.sorted[which]
        size = self.world * self.per_rank
        a = self.sorted_calls[which] * size
        rows = [0] * self.world
        for i in range(a, min(tab.shape[0], a + size)):
            r = (self.first + i) % self.world
            if r != self.rank:
                out[r, rows[r]] = tab[i]
            rows[r] += 1
        out[self.rank] = mine
        self.sorted_calls[which] += 1
        return out

    def all_gather_deltas(self, out, mine):
        size = self.world * self.per_rank
        a = self.calls * size
        rows = [0] * self.world
        out.zero_()
        for i in range(a, min(self.table.shape[0], a + size)):
            r = (self.first + i) % self.world
            if r != self.rank:
                out[r, rows[r]] = self.table[i]
            rows[r] += 1
        out[self.rank] = mine          # this rank's own records: produced here, on the delta stream
        self.calls += 1
        return out


class SlamPipeline:
    def __init__(self, width, height, max_depth, center, half_edge, render_mode=pkg.RENDER_REFERENCE, dist=None,
                 pool_capacity_nodes=1 << 20, count_steps=False, frame_to_model=False, model_min_coverage=0.5, strict_reference=True):
        self.w, self.h, self.depth = width, height, max_depth
        self.center, self.edge = tuple(float(c) for c in center), float(half_edge)
        self.mode = render_mode
        self.dist = dist or DistContext()
        self.focal = 570.3 * width / 640.0
        self.cam = pkg.Camera(width, height, self.focal, self.focal)
        # strict_reference=False: the corrected tracker (include/svoslam.h svoslam_camera_set_strict_reference) on every camera of
        # the session -- own specification, never the headline
        self.strict_reference = bool(strict_reference)
        if not self.strict_reference:
            self.cam.set_strict_reference(False)
        self.ws = pkg.Workspace()
        self.pool = pkg.Pool(pool_capacity_nodes)
        dev = "cuda"
        self.points = torch.empty((height, width, 3), dtype=torch.float32, device=dev)
        self.bbox = torch.zeros(7, dtype=torch.float32, device=dev)
        self.image = torch.zeros((height, width, 4), dtype=torch.uint8, device=dev)
        self.counters = torch.zeros(2, dtype=torch.int64, device=dev) if count_steps else None
        self.keyrange = self.dist.exchange == "keyrange"      # frame-sharded tracking / marches + key-range sharded fusion
        self.frame_sharded = self.dist.exchange in ("deltas", "keyrange")   # (also without a process group: world 1, or an emulated rank)
        self.shard_sort = False
        if self.frame_sharded:
            self.first, self.rows = 0, height                    # whole images of this rank's frames
            self.delta_cam = pkg.Camera(width, height, self.focal, self.focal)   # scratch camera of pair_delta
            if not self.strict_reference:
                self.delta_cam.set_strict_reference(False)
            self.frames_seen, self._prev = 0, None
            # the SORT can be sharded too (round 3): the owner of a frame sorts it -- with the pose a second chain of apply_delta
            # gives -- and the sorted keys / point indices (12 bytes per pixel) are all-gathered per chunk; needs the packed key
            # to fit 64 bits.  Measured with emulated ranks on the 300-frame map (profiles/r03_*emulated*): at 640x480 a rank
            # of 8 is bound by commit + its marches, not by the sort (4550 frames/s either way), and ranks of 2 / 4 LOSE 14-17 %
            # to the extra stream; at 1920x1080, where the sort is 0.25 ms, a rank of 8 gains 21 % (1556 -> 1882).  Default: on
            # for images of a megapixel and more; SVOSLAM_SHARD_SORT=0 / 1 overrides.
            idx_bits = max(1, (width * height - 1).bit_length())
            want = os.environ.get("SVOSLAM_SHARD_SORT")
            want = (width * height >= (1 << 20)) if want is None else want != "0"
            want = want or self.keyrange            # (the key-range commit takes the frame's sorted arrays: the owners sort)
            self.shard_sort = want and 3 * max_depth + 1 + idx_bits <= 64
            assert self.shard_sort or not self.keyrange, "key-range fusion needs the packed sort (3 x depth + 1 + index bits <= 64)"
            if self.shard_sort:
                self.sort_cam = pkg.Camera(width, height, self.focal, self.focal)   # composes the poses the owner sorts with
                if not self.strict_reference:
                    self.sort_cam.set_strict_reference(False)
                self.ws_sort = pkg.Workspace()
        else:
            self.first, self.rows = band_rows(height, self.dist.rank, self.dist.world)
        self.band_exchange = self.dist.enabled and self.dist.exchange == "allreduce"
        if self.band_exchange:
            self.acc = torch.zeros(27, dtype=torch.float64, device=dev)
            self.cam.set_acc(self.acc)
            self.cam.set_band(self.first, self.rows)
            # SURVEY 8e's sharded fusion (round 3): each rank computes and SORTS the keys of its band, the sorted (key, pixel)
            # lists are all-gathered and merged (identical global list on every rank -> identical node numbering), instead
            # of all-gathering the band's points and sorting the whole frame everywhere.  SVOSLAM_BAND_FUSION=points: round 2.
            idx_bits = max(1, (width * height - 1).bit_length())
            # (the merge kernel takes at most 16 lists, svoslam_svo_fuse_merge_sorted; a forced pair sort has no packed word to export)
            self.band_keys = (os.environ.get("SVOSLAM_BAND_FUSION", "keys") != "points" and 3 * max_depth + 1 + idx_bits <= 64
                              and self.dist.world <= 16 and pkg.get_config()["sort_pairs"] == 0)
            if self.band_keys:
                self.ws_band = pkg.Workspace()
                base, rem = divmod(height, self.dist.world)
                self._band_pad = (base + (1 if rem else 0)) * width
        self.last_stats = None
        # frame-to-model tracking (own specification: include/svoslam.h): after a frame has been fused, the map is
        # ray-cast into a depth image from that frame's pose and the next frame is tracked against it instead of against
        # the previous frame's maps.  frame() only (one GPU, stage by stage); the native frame loop does not carry it.
        # A node answers a model ray only once it has been observed ~64 times (A >= 254, what retires a ray in coneTrace): a
        # young map has no model.  The model is used for the next frame when at least model_min_coverage of its pixels are
        # valid, otherwise that frame is tracked against the previous frame (one host read of a pixel count per frame).
        self.frame_to_model = bool(frame_to_model)
        self.model_min_coverage = float(model_min_coverage)
        self.model_used = 0        # frames whose model was accepted
        if self.frame_to_model:
            assert not self.dist.enabled and not self.frame_sharded
            self.model_depth = torch.zeros((height, width), dtype=torch.int16, device=dev)   # (uint16 bit pattern, as the sensor frames)
            self.model_steps = torch.zeros(1, dtype=torch.int64, device=dev) if count_steps else None
            self.cam.set_frame_to_model(True)

    def close(self):
        """give the device back: runner (streams, events) first, then the pool's reservation (+ shadow array, brick field), workspaces
        and cameras; the object is unusable afterwards.  bench.py calls it before its child processes measure the other configurations."""
        kr = self.__dict__.pop("_kr", None)
        if kr is not None:
            for w in kr["ws"]:
                w.close()
        for name in ("_runner", "pool", "ws", "ws_sort", "ws_band", "cam", "delta_cam", "sort_cam"):
            o = self.__dict__.pop(name, None)
            if o is not None:
                o.close()
        for name in ("points", "image", "counters", "model_depth", "acc"):
            self.__dict__.pop(name, None)

    def refresh_model(self):
        """the map as a depth image from the pose of the frame just tracked -> the maps the next frame is tracked against"""
        pkg.raycast_model_depth(self.model_depth, self.focal, self.focal, self.pool.data_ptr, self.center, self.edge,
                                cam_to_world_ptr=self.cam.fusion_transform_ptr(), counters=self.model_steps)
        covered = int((self.model_depth != 0).sum().item())
        if covered >= self.model_min_coverage * self.w * self.h:
            self.cam.set_model_depth(self.model_depth)
            self.model_used += 1
        else:
            self.cam.set_model_depth(None)
        return self.model_depth

    def reset(self):
        """empty map and a fresh tracker; allocations, streams and the launch graphs recorded so far are kept"""
        self.pool.reset()
        self.cam.reset()
        if self.frame_sharded:
            self.delta_cam.reset()
            if self.shard_sort:
                self.sort_cam.reset()
            self.frames_seen, self._prev = 0, None
        if self.counters is not None:
            self.counters.zero_()

    # -- stages (each enqueues on the current stream) -------------------------------------
    def track(self, depth, rgb, timestamp):
        if not self.band_exchange:
            return self.cam.update(depth, rgb, timestamp)
        used = self.cam.begin(depth, rgb, timestamp)
        if used:
            for level in (2, 1, 0):
                for it in range(pkg.PYRAMID_ITERS[level]):
                    self.cam.icp_accumulate(level, it)      # this rank's row band
                    self.dist.all_reduce_sum(self.acc)      # 27 x float64 over xGMI, exact
                    self.cam.icp_solve(level, it)           # every rank: same 6x6 solve, same bits
            self.cam.end()
        return used

    def backproject(self, depth):
        # (multi-GPU: this rank's row band only, absolute pixel coordinates, then all-gather the bands)
        self._backproject_with(depth, self.cam.fusion_transform_ptr())

    def fuse(self, rgb, blocking=False):
        """svoFromPointCloud.  Default: the asynchronous entry point (no host round trip; pool.size is
        fetched from the device when it is read).  blocking=True uses the reference-shaped call and
        returns its statistics."""
        if blocking:
            self.last_stats = pkg.svo_from_point_cloud(self.ws, self.points.view(-1, 3), rgb.view(-1, 3), self.depth,
                                                       self.pool, self.center, self.edge)
            return self.last_stats
        pkg.svo_from_point_cloud_async(self.ws, self.points.view(-1, 3), rgb.view(-1, 3), self.depth, self.pool,
                                       self.center, self.edge)
        return None

    def sort_bands(self, depth, ws, pose_ptr=None):
        """keys + sort of THIS rank's row band, all-gather of the bands' sorted (key, pixel) lists, k-way merge;
        `ws` adopts the merged list as the outcome of its sort phase (plan / commit follow).  Same list as one sort of the
        whole frame, on every rank."""
        w, world = self.w, self.dist.world
        nb = self.rows * w
        pkg.svo_fuse_sort_frame_band(self.ws_band, depth, pose_ptr if pose_ptr is not None else self.cam.fusion_transform_ptr(), self.focal,
                                     self.focal, self.depth, self.center, self.edge, self.first, self.rows)
        pad = self._band_pad
        mine_k = torch.zeros(pad, dtype=torch.int64, device="cuda")
        mine_i = torch.zeros(pad, dtype=torch.int32, device="cuda")
        pkg.svo_fuse_export_sorted(self.ws_band, nb, mine_k, mine_i)
        all_k = torch.empty((world, pad), dtype=torch.int64, device="cuda")
        all_i = torch.empty((world, pad), dtype=torch.int32, device="cuda")
        self.dist.all_gather_sorted(all_k, mine_k)
        self.dist.all_gather_sorted(all_i, mine_i)
        counts = [band_rows(self.h, r, world)[1] * w for r in range(world)]
        merged_k = torch.empty(w * self.h, dtype=torch.int64, device="cuda")
        merged_i = torch.empty(w * self.h, dtype=torch.int32, device="cuda")
        pkg.svo_fuse_merge_sorted([all_k[r, :counts[r]] for r in range(world)], [all_i[r, :counts[r]] for r in range(world)], merged_k, merged_i)
        pkg.svo_fuse_adopt_sorted(ws, merged_k, merged_i, self.depth)
        self._merged = getattr(self, "_merged", [])[-3:] + [(merged_k, merged_i, all_k, all_i)]   # alive until the commits have run
        return merged_k, merged_i

    def fuse_bands(self, depth, rgb):
        """one frame's fusion in band mode: sort_bands + plan + commit (the pool of the one-GPU loop, byte for byte)"""
        n = self.w * self.h
        self.sort_bands(depth, self.ws)
        pkg.svo_fuse_plan(self.ws, n, self.depth, self.pool)
        pkg.svo_fuse_commit(self.ws, rgb.view(-1, 3), self.depth, self.pool)

    def fuse_frame(self, depth, rgb):
        """backproject() + fuse() as the native frame loop runs them (csrc/runner.hip): back-projection, pose transform,
        bounding box and keys in ONE launch straight from the depth image (no point cloud in memory), sort, plan, the
        splits' tiles ahead of the commit, commit.  Same pool and bounding box as backproject() + fuse()."""
        assert not self.band_exchange
        n = self.w * self.h
        idx_bits = max(1, (n - 1).bit_length())
        if 3 * self.depth + 1 + idx_bits > 64:      # the packed key does not fit one word (e.g. depth 16 at 640x480): the stand-alone kernels
            self.backproject(depth)
            self.fuse(rgb)
            return
        pkg.svo_fuse_sort_frame(self.ws, depth, self.cam.fusion_transform_ptr(), self.focal, self.focal, self.depth, self.center,
                                self.edge, self.bbox)
        pkg.svo_fuse_plan(self.ws, n, self.depth, self.pool)
        pkg.svo_fuse_split_early(self.ws, n, self.depth, self.pool)
        pkg.svo_fuse_commit(self.ws, rgb.view(-1, 3), self.depth, self.pool)

    def render(self, view):
        if not self.dist.enabled:
            pkg.cone_trace_svo(self.image, FOV, view, self.pool.data_ptr, self.center, self.edge, self.mode, self.counters)
        else:
            pkg.cone_trace_svo_band(self.image, self.first, self.rows, FOV, view, self.pool.data_ptr, self.center, self.edge,
                                    self.mode, self.counters)
        return self.image

    def frame(self, depth, rgb, timestamp, view):
        self.track(depth, rgb, timestamp)
        if self.band_exchange and self.band_keys:
            self.fuse_bands(depth, rgb)
        else:
            self.backproject(depth)
            self.fuse(rgb)
        if self.frame_to_model:
            self.refresh_model()
        return self.render(view)

    # -- software-pipelined stream of frames ------------------------------------------------
    def run_stream(self, depths, rgbs, timestamps, views, on_render=None):
        """Processes the frames in order on four HIP streams, every frame still going through
        track -> back-project -> fuse -> render with the results of frame():

          P  bilateral filter + vertex/normal pyramids of frame k+2 (three rotating map sets)
          T  19 ICP iterations of frame k+1; touches only the camera state
          S  back-projection, keys + sort and split planning of frame k+1 (reads the pool's tree)
          M  commit of frame k (splits, leaf blend, mip levels -- the only writer of the pool), raycast of frame k

        While frame k is ray-marched on M, S prepares frame k+1 and T tracks it; M then only has to commit.
        Cross-stream order (events): S waits for the pose of its frame (from T) and, before planning, for
        the commit of the previous frame (from M); M waits for the plan of its frame; T waits, before it
        overwrites a slot of the 4-deep pose ring, for the back-projection that reads that slot.
        on_render(i, None) / on_render(i, image) are called around each raycast (stream-ordered on M; the
        image is valid until the next frame)."""
        n = len(timestamps)
        if n == 0:
            return
        assert not self.frame_to_model, "frame-to-model tracking runs through frame()"
        if self.frame_sharded:
            assert on_render is None
            return self.run_stream_sharded(depths, rgbs, timestamps, views)
        if (not self.band_exchange and on_render is None and not os.environ.get("SVOSLAM_TIMELINE")
                and os.environ.get("SVOSLAM_PY_SCHEDULER") != "1"):
            # the same schedule inside the library (csrc/runner.hip): one call, ~0.1 ms of host time per frame
            # instead of the 0.45 ms of the loop below
            if not hasattr(self, "_runner"):
                self._runner = pkg.Runner(self.cam, self.pool, self.w, self.h, self.depth, self.center, self.edge, self.focal,
                                          self.focal, self.mode)
            self._runner.run(depths, rgbs, timestamps, views, self.image, self.first, self.rows, self.counters)
            return
        if not hasattr(self, "_s_track"):
            self._s_maps, self._s_track = torch.cuda.Stream(), torch.cuda.Stream()
            self._s_prep, self._s_map = torch.cuda.Stream(), torch.cuda.Stream()
            self._ws2 = [self.ws, pkg.Workspace()]
            self._points2 = [self.points, torch.empty_like(self.points)]
            # fixed input addresses per stream: the library replays its launch sequences as HIP graphs keyed
            # on the pointers it is given (csrc/graph_cache.hpp), so each frame is copied into a staging buffer
            self._in_track = torch.empty_like(depths[0])
            self._in_prep = torch.empty_like(depths[0])
            self._in_rgb = torch.empty_like(rgbs[0])
        cur = torch.cuda.current_stream()
        for st in (self._s_maps, self._s_track, self._s_prep, self._s_map):
            st.wait_stream(cur)
        ev_pose = [torch.cuda.Event() for _ in range(n)]
        ev_bp = [torch.cuda.Event() for _ in range(n)]
        ev_plan = [torch.cuda.Event() for _ in range(n)]
        ev_commit = [torch.cuda.Event() for _ in range(n)]
        fusion_ptr = [0] * n
        npts = self.w * self.h
        tl = {} if os.environ.get("SVOSLAM_TIMELINE") else None   # diagnostic: timing events at stage boundaries

        def mark(name, i):
            if tl is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                tl[(name, i)] = e

        four = not self.band_exchange   # whole-frame tracker: the maps of a frame are built on their own stream, one frame ahead
        ev_maps = [torch.cuda.Event() for _ in range(n)]

        def enqueue_maps(i):
            """bilateral filter + pyramids of frame i (no dependence on earlier poses)"""
            with torch.cuda.stream(self._s_maps):
                if i >= 2:
                    self._s_maps.wait_event(ev_pose[i - 2])      # its map set was the "last" set of frame i-2
                mark("maps0", i)
                self._in_track.copy_(depths[i])
                if not self.cam.prepare(self._in_track, rgbs[i], timestamps[i]):
                    raise ValueError("run_stream needs strictly increasing timestamps")
                ev_maps[i].record()
                mark("maps1", i)

        def enqueue_track(i):
            with torch.cuda.stream(self._s_track):
                if i >= 4:
                    self._s_track.wait_event(ev_bp[i - 4])       # ring slot i % 4 has been consumed
                if four:
                    self._s_track.wait_event(ev_maps[i])
                    mark("track0", i)
                    self.cam.track_prepared()
                    mark("track1", i)
                else:
                    self._in_track.copy_(depths[i])
                    self.track(self._in_track, rgbs[i], timestamps[i])
                fusion_ptr[i] = self.cam.fusion_transform_ptr()   # ring slot of frame i
                ev_pose[i].record()

        def enqueue_prepare(i):
            ws, pts = self._ws2[i & 1], self._points2[i & 1]
            with torch.cuda.stream(self._s_prep):
                self._s_prep.wait_event(ev_pose[i])
                mark("prep0", i)
                self.points = pts
                self._in_prep.copy_(depths[i])
                if self.band_exchange and self.band_keys:
                    self.sort_bands(self._in_prep, ws, fusion_ptr[i])
                    ev_bp[i].record()
                else:
                    self._backproject_with(self._in_prep, fusion_ptr[i])
                    ev_bp[i].record()
                    pkg.svo_fuse_sort(ws, pts.view(-1, 3), self.depth, self.center, self.edge)
                if i > 0:
                    self._s_prep.wait_event(ev_commit[i - 1])    # the tree the plan reads
                mark("plan0", i)
                pkg.svo_fuse_plan(ws, npts, self.depth, self.pool)
                ev_plan[i].record()
                mark("plan1", i)

        if four:
            enqueue_maps(0)
        enqueue_track(0)
        if four and n > 1:
            enqueue_maps(1)
        enqueue_prepare(0)
        for i in range(n):
            if i + 1 < n:
                enqueue_track(i + 1)
            if four and i + 2 < n:
                enqueue_maps(i + 2)
            with torch.cuda.stream(self._s_map):
                self._s_map.wait_event(ev_plan[i])
                mark("commit0", i)
                self._in_rgb.copy_(rgbs[i])
                pkg.svo_fuse_commit(self._ws2[i & 1], self._in_rgb.view(-1, 3), self.depth, self.pool)
                ev_commit[i].record()
                mark("commit1", i)
            if i + 1 < n:
                enqueue_prepare(i + 1)      # host order: after ev_commit[i] has been recorded
            with torch.cuda.stream(self._s_map):
                if on_render is not None:
                    on_render(i, None)        # "before render" hook (event timing)
                self.render(views[i])
                mark("render1", i)
                if on_render is not None:
                    on_render(i, self.image)
        for st in (self._s_maps, self._s_track, self._s_prep, self._s_map):
            cur.wait_stream(st)
        if self.band_exchange:
            self.dist.check_mailbox()
        if tl is not None:
            torch.cuda.synchronize()
            base = tl[("commit0", 0)]
            self.timeline = {k: base.elapsed_time(e) for k, e in tl.items()}

    def run_stream_sharded(self, depths, rgbs, timestamps, views, images=None, per_rank=None):
        """One rank of a frame-sharded session (exchange "deltas").  Frame g (counted from the camera's first frame)
        belongs to rank g % world:
          D  that rank tracks it against frame g-1 -- maps of both + 19 ICP iterations on the scratch camera
             (svoslam_camera_pair_delta), a function of the two depth images only -- in chunks of world * per_rank frames,
             each followed by ONE all-gather of the ranks' 80-byte update_trans records;
          then the native runner (svoslam_runner_run_sharded) on its usual streams: T composes the poses of ALL frames from
          the gathered records (waiting for a chunk's all-gather), S back-projects / sorts / plans and M commits ALL frames
          (every rank keeps a byte-identical replica of the map), M ray-marches this rank's frames only.
        Everything is enqueued up front: D runs ahead of the fusion as far as the records go.  images: per-frame output
        tensors for the frames of this rank (default: self.image for each).  Poses, replicas and images are those of the
        one-GPU loop."""
        n = len(timestamps)
        world, rank = self.dist.world, self.dist.rank
        if per_rank is None:
            per_rank = max(1, 16 // world)
        if not hasattr(self, "_s_delta"):
            if not self.keyrange:
                self._runner = pkg.Runner(self.cam, self.pool, self.w, self.h, self.depth, self.center, self.edge, self.focal,
                                          self.focal, self.mode)
            self._s_delta = torch.cuda.Stream()
            self._s_sort = torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        self._s_delta.wait_stream(cur)
        self._s_sort.wait_stream(cur)
        npix = self.w * self.h
        skeys, sidx, sevents = [None] * n, [None] * n, [None] * n
        use_sort = self.shard_sort and getattr(self.dist, "has_sorted", True)   # (one choice per session: the sort camera sees every frame or none)
        g0 = self.frames_seen
        deltas, events, march, outs = [None] * n, [None] * n, [False] * n, [None] * n
        keep = []
        for (a, b, slots) in frame_shards(n, g0, world, per_rank):
            with torch.cuda.stream(self._s_delta):
                # (allocated and zeroed ON the delta stream: a fill enqueued on the caller's stream could land after the records)
                mine = torch.zeros((per_rank, pkg.DELTA_FLOATS), dtype=torch.float32, device="cuda")
                allr = torch.empty((world, per_rank, pkg.DELTA_FLOATS), dtype=torch.float32, device="cuda")
                for i in range(a, b):
                    r, row = slots[i - a]
                    if r == rank and g0 + i > 0:   # (a camera's first frame has no ICP)
                        pd, pr = (depths[i - 1], rgbs[i - 1]) if i > 0 else self._prev
                        self.delta_cam.pair_delta(pd, pr, depths[i], rgbs[i], mine[row])
                self.dist.all_gather_deltas(allr, mine)
                ev = torch.cuda.Event()
                ev.record()
            keep.append((mine, allr))
            for i in range(a, b):
                r, row = slots[i - a]
                deltas[i] = allr[r, row]
                march[i] = r == rank
                if march[i]:
                    outs[i] = images[i] if images is not None else self.image
            events[a] = ev      # the pose stream is in order: the chunk's first frame waits for the gather
            if use_sort:
                with torch.cuda.stream(self._s_sort):
                    self._s_sort.wait_event(ev)
                    mine_k = torch.empty((per_rank, npix), dtype=torch.int64, device="cuda")
                    mine_i = torch.empty((per_rank, npix), dtype=torch.int32, device="cuda")
                    all_k = torch.empty((world, per_rank, npix), dtype=torch.int64, device="cuda")
                    all_i = torch.empty((world, per_rank, npix), dtype=torch.int32, device="cuda")
                    for i in range(a, b):
                        r, row = slots[i - a]
                        self.sort_cam.apply_delta(deltas[i], timestamps[i])     # main.cpp:40's pose of frame i, on this stream
                        if r == rank:
                            pkg.svo_fuse_sort_frame(self.ws_sort, depths[i], self.sort_cam.fusion_transform_ptr(), self.focal, self.focal,
                                                    self.depth, self.center, self.edge)
                            pkg.svo_fuse_export_sorted(self.ws_sort, npix, mine_k[row], mine_i[row])
                    self.dist.all_gather_sorted(all_k, mine_k)
                    self.dist.all_gather_sorted(all_i, mine_i)
                    evs = torch.cuda.Event()
                    evs.record()
                keep.append((mine_k, mine_i, all_k, all_i))
                for i in range(a, b):
                    r, row = slots[i - a]
                    skeys[i], sidx[i] = all_k[r, row], all_i[r, row]
                sevents[a] = evs
        if self.keyrange:
            self._run_keyrange(n, g0, rgbs, timestamps, views, deltas, events, march, outs, skeys, sidx, sevents)
        elif use_sort:
            self._runner.run_sharded_presorted(depths, rgbs, timestamps, views, deltas, events, march, outs, skeys, sidx, sevents, 0, self.h,
                                               self.counters)
        else:
            self._runner.run_sharded(depths, rgbs, timestamps, views, deltas, events, march, outs, 0, self.h, self.counters)
        self._keep_sharded = keep
        self._prev = (depths[n - 1], rgbs[n - 1])
        self.frames_seen = g0 + n
        self.marched_last_call = sum(march)
        if getattr(self.dist, "mailbox", None) is not None:
            self.dist.check_mailbox()

    def _run_keyrange(self, n, g0, rgbs, timestamps, views, deltas, events, march, outs, skeys, sidx, sevents):
        """The fusion of a frame-sharded session cut by KEY RANGE (csrc/svo_build.hip "key-range sharded commit"; include/svoslam.h
        svoslam_svo_fuse_keyrange_*): per frame, on two streams,
          C  keyrange_commit -- this rank's slice of the frame's sorted keys planned and committed where no replica sees it, its
             delta packed -- then the all-gather of the ranks' deltas; frame k+1's may run beside the march of frame k;
          M  pose of the frame (the 80-byte record), keyrange_apply -- every rank's delta into this replica --, and the ray march of
             the frames this rank owns.
        Driven from Python (five calls per frame): the collective sits between the two halves of every frame.  An emulated rank commits
        the frames its table does not cover in one piece."""
        world, rank = self.dist.world, self.dist.rank
        npix = self.w * self.h
        if not hasattr(self, "_kr"):
            cap_words = pkg.KEYRANGE_FIXED_WORDS + 16 * npix          # 64 bytes per pixel of a whole frame: ample for any slice
            self._kr = {"sc": torch.cuda.Stream(), "sm": torch.cuda.Stream(), "ws": [pkg.Workspace(), pkg.Workspace()],
                        "buf": [torch.zeros(cap_words, dtype=torch.int32, device="cuda") for _ in range(2)], "whole": 0, "frames": 0,
                        "used_bytes": []}
        kr = self._kr
        sc, sm = kr["sc"], kr["sm"]
        cur = torch.cuda.current_stream()
        sc.wait_stream(cur); sm.wait_stream(cur)
        table = getattr(self.dist, "kr_deltas", None)
        ev_apply = None
        for i in range(n):
            k = i & 1
            ws, buf = kr["ws"][k], kr["buf"][k]
            colors = rgbs[i].view(-1, 3)
            whole = table is not None and table[g0 + i] is None       # (an emulated rank without the other ranks' deltas for this frame)
            with torch.cuda.stream(sc):
                if sevents[i] is not None:
                    sc.wait_event(sevents[i])          # the chunk's sorted arrays (in-order stream: once per chunk)
                if ev_apply is not None:
                    sc.wait_event(ev_apply)            # the plan reads the structure apply k-1 left
                if not whole:
                    pkg.svo_fuse_keyrange_commit(ws, skeys[i], sidx[i], colors, self.depth, self.pool, rank, world, buf)
                    gathered = self.dist.all_gather_keyrange(g0 + i, buf)
                evc = torch.cuda.Event(); evc.record()
            with torch.cuda.stream(sm):
                sm.wait_event(evc)
                if events[i] is not None:
                    sm.wait_event(events[i])           # the chunk's pose records
                self.cam.apply_delta(deltas[i], timestamps[i])
                if not whole:
                    pkg.svo_fuse_keyrange_apply(ws, skeys[i], self.depth, self.pool, gathered)
                else:
                    kr["whole"] += 1
                    pkg.svo_fuse_adopt_sorted(ws, skeys[i], sidx[i], self.depth)
                    pkg.svo_fuse_plan(ws, npix, self.depth, self.pool)
                    pkg.svo_fuse_commit(ws, colors, self.depth, self.pool)
                ev_apply = torch.cuda.Event(); ev_apply.record()
                if march[i]:
                    pkg.cone_trace_svo(outs[i], FOV, views[i], self.pool.data_ptr, self.center, self.edge, self.mode, counters=self.counters)
            kr["frames"] += 1
        cur.wait_stream(sc); cur.wait_stream(sm)

    def keyrange_check(self):
        """after a key-range run (blocking): raises if an apply was refused (a delta overflowed its buffer, or deltas of different frames
        met: the replica is then behind the others)"""
        if hasattr(self, "_kr"):
            for w in self._kr["ws"]:
                f = pkg.svo_fuse_keyrange_status(w) if w is not None else 0
                if f:
                    raise RuntimeError("key-range apply refused a frame: flags %d (2 overflow, 4 mismatch)" % f)

    def _backproject_with(self, depth, fusion_ptr):
        if not self.band_exchange:
            pkg.generate_vertex_map(depth, self.points, self.focal, self.focal, self.w, self.h)
            pkg.transform_vertex_map_dmat(self.points, fusion_ptr)
        else:
            pkg.generate_vertex_map_rows(depth, self.points, self.first, self.rows, self.focal, self.focal, self.w, self.h)
            pkg.transform_vertex_map_dmat(self.points[self.first:self.first + self.rows], fusion_ptr)
            self.dist.all_gather_rows(self.points, self.h)
        pkg.point_cloud_bbox_device(self.ws, self.points, self.bbox)


def keyrange_delta_table(keys_tab, idx_tab, rgb, first, rank, world, max_depth, pool_capacity_nodes):
    """What the OTHER ranks of a key-range session deliver, for an emulated rank (bench.py --exchange keyrange --emulate-rank R/N): a truth
    pool fuses the stream frame by frame in one piece (keys_tab[k] / idx_tab[k]: the frame's sorted keys and point indices, rgb[k] its
    colours); before frame k >= first is fused, every rank s != rank plans and commits ITS slice on that pool and its delta is kept
    (svoslam_svo_fuse_keyrange_commit + _discard: the pool is not touched).  Returns (deltas, shared, bytes): deltas[k][s] (used words only;
    deltas[k] = None for k < first, None at s == rank), shared[k] = records above the splitter level in frame k (planned by several ranks,
    ranked in their union: the first frames of a map), bytes[k] = [used bytes of every rank's delta] -- what one all-gather of frame k
    moves (the emulated rank's own share measured with the same call)."""
    total, npix = keys_tab.shape[0], keys_tab.shape[1]
    truth = pkg.Pool(pool_capacity_nodes)
    ws_t, ws_s = pkg.Workspace(), pkg.Workspace()
    buf = torch.zeros(pkg.KEYRANGE_FIXED_WORDS + 16 * npix, dtype=torch.int32, device="cuda")
    deltas, shared, nbytes = [None] * total, [0] * total, [None] * total
    for k in range(total):
        colors = rgb[k].view(-1, 3)
        if k >= first:
            row, used, top = [None] * world, [0] * world, 0
            for s_ in range(world):
                pkg.svo_fuse_keyrange_commit(ws_s, keys_tab[k], idx_tab[k], colors, max_depth, truth, s_, world, buf)
                pkg.svo_fuse_keyrange_discard(ws_s, truth)
                head = buf[:512].cpu().numpy().view(np.uint32)
                used[s_] = int(head[pkg.KEYRANGE_USED_WORD]) * 4
                assert head[7] == 0, "a delta overflowed its buffer"
                top = max(top, int(head[13]))            # records above the splitter level in this rank's slice
                if s_ != rank:
                    row[s_] = buf[: used[s_] // 4].clone()
            deltas[k], shared[k], nbytes[k] = row, top, used
        pkg.svo_fuse_adopt_sorted(ws_t, keys_tab[k], idx_tab[k], max_depth)
        pkg.svo_fuse_plan(ws_t, npix, max_depth, truth)
        pkg.svo_fuse_commit(ws_t, colors, max_depth, truth)
    torch.cuda.synchronize()
    ws_t.close(); ws_s.close(); truth.close()
    return deltas, shared, nbytes


def ground_truth_view(frame, synth):
    """View matrix (glm lookAt convention) of the synthetic sensor pose of `frame`, expressed in the map
    frame = camera frame of frame 0 (the tracker starts at identity)."""
    (p0, yaw0), (pk, yawk) = synth.camera_pose(0), synth.camera_pose(frame)
    c0, s0 = np.cos(yaw0), np.sin(yaw0)

    def to_map(v):  # R0^T v with R0 = yaw about +y, camera forward +z
        x, y, z = v
        return np.array([c0 * x - s0 * z, y, s0 * x + c0 * z])

    eye = to_map(np.array(pk) - np.array(p0))
    fwd = to_map(np.array([np.sin(yawk), 0.0, np.cos(yawk)]))
    return look_at(eye, eye + fwd, (0.0, 1.0, 0.0))


def look_at(eye, center, up):
    """glm::lookAt (gtc/matrix_transform.inl:416-441), float32, column-major flat[16]."""
    f32 = np.float32
    eye, center, up = (np.asarray(a, f32) for a in (eye, center, up))

    def norm(v):
        return (v * (f32(1.0) / np.sqrt(f32((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])))).astype(f32)

    def cross(x, y):
        return np.array([x[1] * y[2] - y[1] * x[2], x[2] * y[0] - y[2] * x[0], x[0] * y[1] - y[0] * x[1]], f32)

    def dot(a, b):
        return f32((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2])

    f = norm(center - eye)
    s = norm(cross(f, up))
    u = cross(s, f)
    m = np.zeros((4, 4), f32)  # m[col][row]
    m[0] = [s[0], u[0], -f[0], 0]
    m[1] = [s[1], u[1], -f[1], 0]
    m[2] = [s[2], u[2], -f[2], 0]
    m[3] = [-dot(s, eye), -dot(u, eye), dot(f, eye), 1]
    return m.reshape(16)
