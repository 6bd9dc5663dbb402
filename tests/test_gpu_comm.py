"""The C ABI's data-parallel exchange entries (include/coati_hip.h: coati_comm_*, coati_allgather_rows, coati_reducescatter_rows,
coati_allreduce_bucket) on RCCL, world size 1 on the one GPU of the box: the calls a torch-free host makes around
coati_engine_forward(train | 2) / coati_engine_infonce / coati_engine_backward.  At world size 1 every collective is the identity
(the average too), which is exactly what is asserted -- what the test proves is that the library finds RCCL, builds a communicator on
the current device and that the three calls run stream-ordered on a caller's stream.  Two ranks: test_two_gpu_* below (skips on a
one-GPU box)."""
import ctypes
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def test_comm_world_size_one_collectives_are_identities():
    from coati_amd import _lib
    l = _lib.lib()
    torch.cuda.set_device(0)
    uid = ctypes.create_string_buffer(128)
    _lib.check(l.coati_comm_unique_id(uid, 128), "unique_id")
    assert any(b != 0 for b in uid.raw)
    h = ctypes.c_void_p()
    _lib.check(l.coati_comm_init(uid, 0, 1, ctypes.byref(h)), "comm_init")
    assert l.coati_comm_rank(h) == 0 and l.coati_comm_world(h) == 1
    side = torch.cuda.Stream()
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(96, 256, device="cuda", generator=g)
    xb = x.bfloat16()
    with torch.cuda.stream(side):
        st = ctypes.c_void_p(side.cuda_stream)
        out = torch.empty_like(x)
        _lib.check(l.coati_allgather_rows(h, _ptr(x), _ptr(out), 96, 256, 0, st), "allgather")
        outb = torch.empty_like(xb)
        _lib.check(l.coati_allgather_rows(h, _ptr(xb), _ptr(outb), 96, 256, 1, st), "allgather bf16")
        rs = torch.empty_like(x)
        _lib.check(l.coati_reducescatter_rows(h, _ptr(x), _ptr(rs), 96, 256, 0, st), "reducescatter")
        bucket = x.clone().view(-1)
        _lib.check(l.coati_allreduce_bucket(h, _ptr(bucket), bucket.numel(), 0, 1, st), "allreduce avg")
        bucket2 = x.clone().view(-1)
        _lib.check(l.coati_allreduce_bucket(h, _ptr(bucket2), bucket2.numel(), 0, 0, st), "allreduce sum")
    side.synchronize()
    assert torch.equal(out, x) and torch.equal(outb, xb) and torch.equal(rs, x)
    assert torch.equal(bucket.view_as(x), x) and torch.equal(bucket2.view_as(x), x)
    # error codes, not exceptions
    assert l.coati_allgather_rows(h, _ptr(x), _ptr(out), 96, 256, 7, None) == -1 and b"dtype" in l.coati_last_error()
    assert l.coati_allreduce_bucket(h, None, 4, 0, 1, None) == -1
    assert l.coati_comm_init(uid, 3, 2, ctypes.byref(ctypes.c_void_p())) == -1 and b"rank" in l.coati_last_error()
    _lib.check(l.coati_comm_destroy(h), "comm_destroy")
    assert l.coati_comm_destroy(None) == 0


def _rank_main(rank, world, idfile, q):
    sys.path.insert(0, ROOT)
    import time
    from coati_amd import _lib
    l = _lib.lib()
    torch.cuda.set_device(rank)
    if rank == 0:
        uid = ctypes.create_string_buffer(128)
        _lib.check(l.coati_comm_unique_id(uid, 128), "unique_id")
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid.raw)
        os.replace(idfile + ".tmp", idfile)
    else:
        for _ in range(600):
            if os.path.exists(idfile):
                break
            time.sleep(0.1)
        uid = ctypes.create_string_buffer(open(idfile, "rb").read(), 128)
    h = ctypes.c_void_p()
    _lib.check(l.coati_comm_init(uid, rank, world, ctypes.byref(h)), "comm_init")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows, cols = 8, 64
    x = torch.full((rows, cols), float(rank + 1), device="cuda")
    allx = torch.empty(world * rows, cols, device="cuda")
    _lib.check(l.coati_allgather_rows(h, _ptr(x), _ptr(allx), rows, cols, 0, st), "allgather")
    gsum = torch.empty(rows, cols, device="cuda")
    contrib = torch.arange(world * rows, device="cuda", dtype=torch.float32)[:, None].expand(world * rows, cols).contiguous() * (rank + 1)
    _lib.check(l.coati_reducescatter_rows(h, _ptr(contrib), _ptr(gsum), rows, cols, 0, st), "reducescatter")
    b = torch.full((1000,), float(rank), device="cuda")
    _lib.check(l.coati_allreduce_bucket(h, _ptr(b), 1000, 0, 1, st), "allreduce")
    torch.cuda.synchronize()
    tot = sum(r + 1 for r in range(world))
    ok = all(bool((allx[r * rows:(r + 1) * rows] == r + 1).all()) for r in range(world))
    ok = ok and torch.equal(gsum[:, 0], torch.arange(rank * rows, (rank + 1) * rows, device="cuda", dtype=torch.float32) * tot)
    ok = ok and bool(torch.allclose(b, torch.full_like(b, sum(range(world)) / world)))
    _lib.check(l.coati_comm_destroy(h), "destroy")
    q.put((rank, bool(ok)))


def test_two_gpu_comm_entries(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    idfile = str(tmp_path / "rccl_id")
    procs = [ctx.Process(target=_rank_main, args=(r, 2, idfile, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == {0: True, 1: True}
