"""The peer-to-peer mailbox (csrc/mailbox.hip): all-gather and rank-ordered all-reduce of small records without a
collective library.  Here several mailboxes of ONE process on one device stand in for the ranks, driven from one stream:
every rank posts, then every rank collects (a collect kernel occupies its stream until its peers have posted, so the
combined calls need a stream -- in practice a process and a device -- per rank); the two-process arrangement over hipIpc handles runs in
tests/test_gpu_sharded.py::test_two_process_sharded_session_on_one_gpu (its pose records travel through the mailbox)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import svoslam_pkg
    return svoslam_pkg.load(), torch


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_mailbox_all_gather_and_rank_ordered_reduce(env, world):
    pkg, torch = env
    boxes = [pkg.Mailbox(r, world) for r in range(world)]
    for b in boxes:
        b.connect_local(boxes)
    rng = np.random.default_rng(world)
    for epoch in range(11):                                   # more epochs than ring slots: the inbox wraps
        recs = [rng.integers(-2 ** 31, 2 ** 31 - 1, 20, dtype=np.int64).astype(np.int32) for _ in range(world)]     # 80-byte records
        # integer-valued doubles up to 2^52 (the ICP sums are such): exact, so any order gives the same sum -- and
        # non-integers, where only the RANK ORDER of the additions makes every rank agree
        sums = [np.concatenate([rng.integers(-2 ** 50, 2 ** 50, 20).astype(np.float64), rng.normal(size=7) * 1e6]) for _ in range(world)]
        srcs = [torch.from_numpy(recs[r]).cuda() for r in range(world)]
        outs = [torch.zeros((world, 20), dtype=torch.int32, device="cuda") for r in range(world)]
        accs = [torch.from_numpy(sums[r]).cuda() for r in range(world)]
        for r in range(world):
            boxes[r].post(srcs[r])
        for r in range(world):
            boxes[r].collect(outs[r], 80)
        for r in range(world):
            boxes[r].post(accs[r])
        for r in range(world):
            boxes[r].collect(accs[r], 27 * 8, reduce_f64=True)
        torch.cuda.synchronize()
        want = np.stack(recs)
        total = np.zeros(27)
        for r in range(world):
            total = total + sums[r]                           # rank order
        for r in range(world):
            assert np.array_equal(outs[r].cpu().numpy(), want), (epoch, r)
            assert np.array_equal(accs[r].cpu().numpy().view(np.uint64), total.view(np.uint64)), (epoch, r)
    assert not any(b.failed() for b in boxes)


def test_mailbox_rejects_bad_sizes(env):
    pkg, torch = env
    b = pkg.Mailbox(0, 1)
    b.connect_local([b])
    with pytest.raises(pkg.SvoslamError):
        b.all_gather(torch.zeros(3, dtype=torch.uint8, device="cuda"), torch.zeros(3, dtype=torch.uint8, device="cuda"))     # not a multiple of 4
    with pytest.raises(pkg.SvoslamError):
        b.all_reduce_f64(torch.zeros(300, dtype=torch.float64, device="cuda"))                                                # > 256 doubles
    fresh = pkg.Mailbox(0, 2)
    with pytest.raises(pkg.SvoslamError):
        fresh.all_reduce_f64(torch.zeros(4, dtype=torch.float64, device="cuda"))                                              # not connected


def test_mailbox_timeout_is_loud(env):
    """a peer that never posts: the wait gives up, the missing record arrives as all-ones granules (NaN), the sum is NaN, the
    sticky flag is set and DistContext.check_mailbox() raises (a silent stale granule let the maps diverge)"""
    pkg, torch = env
    import importlib
    pl = importlib.import_module("octree_slam_amd.pipeline")
    boxes = [pkg.Mailbox(r, 2) for r in range(2)]
    for b in boxes:
        b.connect_local(boxes)
        b.set_wait_limit(2000)
    rec = torch.arange(20, dtype=torch.int32, device="cuda")
    out = torch.zeros((2, 20), dtype=torch.int32, device="cuda")
    boxes[0].post(rec)                 # rank 1 never posts
    boxes[0].collect(out, 80)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert np.array_equal(o[0], np.arange(20)) and np.all(o[1].view(np.uint32) == 0xFFFFFFFF)
    assert boxes[0].failed() and not boxes[1].failed()
    acc = torch.ones(27, dtype=torch.float64, device="cuda")
    boxes[0].post(acc)
    boxes[0].collect(acc, 27 * 8, reduce_f64=True)
    torch.cuda.synchronize()
    assert bool(torch.isnan(acc).all())
    ctx = pl.DistContext(0, 1)
    ctx.mailbox = boxes[0]
    with pytest.raises(pkg.SvoslamError):
        ctx.check_mailbox()
