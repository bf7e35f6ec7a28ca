"""world_size-2 `gloo` test of the N > 1 path on CPU: the single all-gather of the per-image
payload reproduces the whole scene on every rank, and the shards partition the images."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from limap_amd import dist as ltdist, synthetic as syn
        # ragged segment counts per image exercise the padding of the gathered buffer
        sc = syn.make_scene(n_views=7, n_segs=30, n_neighbors=3, seed=2)
        keep = [slice(sc.seg_off[i], sc.seg_off[i] + 30 - 3 * i) for i in range(7)]
        segs = np.concatenate([sc.segs[s] for s in keep], 0)
        seg_off = np.zeros(8, np.int64)
        seg_off[1:] = np.cumsum([30 - 3 * i for i in range(7)])
        g = ltdist.SceneGather(sc.img_ids, seg_off, rank, world, torch.device("cpu"))
        # poison everything that is not this rank's so that only the collective can fill it in
        a, b = g.bounds[rank], g.bounds[rank + 1]
        kv, qv, tv, sg = sc.kvec.copy(), sc.qvec.copy(), sc.tvec.copy(), segs.copy()
        mask = np.ones(7, bool); mask[a:b] = False
        kv[mask] = np.nan; qv[mask] = np.nan; tv[mask] = np.nan
        smask = np.ones(len(sg), bool); smask[seg_off[a]:seg_off[b]] = False
        sg[smask] = np.nan
        g.load_local(kv, qv, tv, sg)
        k, q_, t, s = g.all_gather()
        ok = (np.array_equal(k.numpy(), sc.kvec) and np.array_equal(q_.numpy(), sc.qvec)
              and np.array_equal(t.numpy(), sc.tvec) and np.array_equal(s.numpy()[:len(segs)], segs))
        # the asynchronous form used by bench.py (collective of the next step overlapped with the kernels):
        # wipe the receive buffer, launch, wait, and read the chunks in place
        g.recv.fill_(float("nan"))
        h = g.gather_async()
        h.wait()
        ib, pk, pq, pt, ps = g.chunk_pointers()
        ok = ok and ib == [int(x) for x in g.bounds[:-1]] and not bool(torch.isnan(g.recv[:g.sizes[0]]).any())
        for r in range(world):
            a_, b_ = g.bounds[r], g.bounds[r + 1]
            o = r * g.max_size
            ok = ok and np.array_equal(g.recv[o:o + 4 * (b_ - a_)].numpy().reshape(-1, 4), sc.kvec[a_:b_])
        mine = ltdist.shard_images(sc.img_ids, rank, world).tolist()
        # the tail's gather: packed per-image results travel to rank 0 only, as two tensors
        rng = np.random.default_rng(10 + rank)
        fake = []
        for i in mine:
            m = int(seg_off[sc.img_ids.tolist().index(i) + 1] - seg_off[sc.img_ids.tolist().index(i)])
            cnt = rng.integers(0, 4, m)
            eoff = np.zeros(m + 1, np.int64); eoff[1:] = np.cumsum(cnt)
            fake.append(dict(img_id=int(i), nb_ids=rng.integers(0, 7, 3).astype(np.int32), line=rng.normal(size=(m, 10)),
                             score=rng.random(m), src=rng.integers(0, 30, (m, 2)).astype(np.int32),
                             n_tris=rng.integers(0, 9, m).astype(np.int32), edge_off=eoff,
                             edges=rng.integers(0, 30, (int(eoff[-1]), 2)).astype(np.int32)))
        ints, flts = ltdist.pack_image_results(fake)
        parts = ltdist.gather_packed_to_rank0(ints, flts, rank, world, torch.device("cpu"))
        if rank == 0:
            back = ltdist.unpack_image_results(*parts[0])
            ok = ok and len(back) == len(fake) and all(
                a["img_id"] == b["img_id"] and all(np.array_equal(a[k], b[k]) for k in
                                                   ("nb_ids", "line", "score", "src", "n_tris", "edge_off", "edges"))
                for a, b in zip(back, fake))
            other = ltdist.unpack_image_results(*parts[1])
            allimgs = sorted([r["img_id"] for r in back] + [r["img_id"] for r in other])
            ok = ok and allimgs == sc.img_ids.tolist()
        else:
            ok = ok and parts is None
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_all_gather_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:  # a port the OS says is free (not derived from the pid: parallel CI runs)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def _worker8(rank, world, port, q):
    """Config-3-shaped sharding (BASELINE configs[2]: 1000 views x 1000 segs over 8 ranks) without the segments' bulk:
    1000 images with ragged segment counts and connection weights, the weighted shard bounds, the all-gather with
    padded per-rank payloads, and the packed gather of per-image results to rank 0 with eight senders."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from limap_amd import dist as ltdist
        rng = np.random.default_rng(123)  # the same scene on every rank
        n_img = 1000
        img_ids = np.sort(rng.choice(5000, n_img, replace=False)).astype(np.int32)
        n_seg = rng.integers(3, 40, n_img)  # ragged (a real scene has ~1000 per image; the layout logic is the same)
        seg_off = np.zeros(n_img + 1, np.int64); seg_off[1:] = np.cumsum(n_seg)
        kvec = rng.uniform(300, 900, (n_img, 4)); qvec = rng.normal(size=(n_img, 4)); tvec = rng.normal(size=(n_img, 3))
        segs = rng.uniform(0, 800, (int(seg_off[-1]), 4))
        weights = n_seg.astype(float) * rng.integers(5, 21, n_img) * 10  # connections an image brings (matched, topk 10)
        g = ltdist.SceneGather(img_ids, seg_off, rank, world, torch.device("cpu"), weights=weights)
        b = g.bounds
        ok = b[0] == 0 and b[-1] == n_img and all(b[r] < b[r + 1] for r in range(world))
        # connection-weighted bounds: no shard more than 1.35x the mean load
        loads = np.array([weights[b[r]:b[r + 1]].sum() for r in range(world)])
        ok = ok and loads.max() <= 1.35 * loads.mean()
        kv, qv, tv, sg = kvec.copy(), qvec.copy(), tvec.copy(), segs.copy()
        mask = np.ones(n_img, bool); mask[b[rank]:b[rank + 1]] = False
        kv[mask] = np.nan; qv[mask] = np.nan; tv[mask] = np.nan
        smask = np.ones(len(sg), bool); smask[seg_off[b[rank]]:seg_off[b[rank + 1]]] = False
        sg[smask] = np.nan
        g.load_local(kv, qv, tv, sg)
        k, q_, t, s = g.all_gather()
        ok = ok and np.array_equal(k.numpy(), kvec) and np.array_equal(q_.numpy(), qvec) and np.array_equal(t.numpy(), tvec)
        ok = ok and np.array_equal(s.numpy()[:len(segs)], segs)
        mine = ltdist.shard_images(img_ids, rank, world, weights)
        ok = ok and np.array_equal(mine, img_ids[b[rank]:b[rank + 1]])
        # packed per-image results of this shard -> rank 0 (sizes differ per rank: padding + size exchange)
        r2 = np.random.default_rng(1000 + rank)
        fake = []
        for i in mine:
            m = int(n_seg[int(np.searchsorted(img_ids, i))])
            cnt = r2.integers(0, 3, m)
            eoff = np.zeros(m + 1, np.int64); eoff[1:] = np.cumsum(cnt)
            fake.append(dict(img_id=int(i), nb_ids=r2.integers(0, 5000, 4).astype(np.int32), line=r2.normal(size=(m, 10)),
                             score=r2.random(m), src=r2.integers(0, 40, (m, 2)).astype(np.int32),
                             n_tris=r2.integers(0, 9, m).astype(np.int32), edge_off=eoff,
                             edges=r2.integers(0, 40, (int(eoff[-1]), 2)).astype(np.int32)))
        ints, flts = ltdist.pack_image_results(fake)
        parts = ltdist.gather_packed_to_rank0(ints, flts, rank, world, torch.device("cpu"))
        if rank == 0:
            seen = []
            for r in range(world):
                got = ltdist.unpack_image_results(*parts[r])
                seen += [x["img_id"] for x in got]
                if r == 0:
                    ok = ok and all(np.array_equal(a_["line"], b_["line"]) and np.array_equal(a_["edges"], b_["edges"])
                                    for a_, b_ in zip(got, fake))
            ok = ok and sorted(seen) == img_ids.tolist()
        else:
            ok = ok and parts is None
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_config3_shaped_sharding_world8_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(8)]


def _stream_worker(rank, world, port, q):
    """The streamed job's control flow on CPU (limap_amd/stream.py; the device part is tests/test_gpu_stream.py): every
    rank plans the same chunks, takes chunk k iff k % world == rank, builds the closure sub-scene, and its per-image results
    reach rank 0 through the ONE gather -- every image exactly once, from the rank its chunk belongs to."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from limap_amd import stream as ltstream, synthetic as syn
        sc = syn.make_scene(n_views=23, n_segs=20, n_neighbors=4, seed=4)
        plan = ltstream.plan_chunks(sc.img_ids, sc.neighbors, 5, world)
        ok = [c.rank for c in plan] == [k % world for k in range(len(plan))] and len(plan) == 5
        ok = ok and sorted(int(i) for c in plan for i in c.images) == sc.img_ids.tolist()
        mine = [c for c in plan if c.rank == rank]
        results = []
        for c in mine:
            need = set(int(i) for i in c.images) | {int(n) for i in c.images for n in sc.neighbors[int(i)]}
            ok = ok and set(c.closure.tolist()) == need and bool(np.all(np.diff(c.closure) > 0))
            ids, k, qv, t, off, sg = ltstream.closure_arrays(c, sc.img_ids, sc.kvec, sc.qvec, sc.tvec, sc.seg_off, sc.segs)
            for j, i in enumerate(ids):  # the sub-scene's rows are the model's rows of the same image
                n = sc.img_ids.tolist().index(int(i))
                ok = ok and np.array_equal(sg[off[j]:off[j + 1]], sc.segs[sc.seg_off[n]:sc.seg_off[n + 1]])
                ok = ok and np.array_equal(k[j], sc.kvec[n]) and np.array_equal(qv[j], sc.qvec[n]) and np.array_equal(t[j], sc.tvec[n])
            for i in c.images:  # stand-in for lt_export_image_results: tagged with the chunk and the rank
                n = sc.img_ids.tolist().index(int(i))
                m = int(sc.seg_off[n + 1] - sc.seg_off[n])
                eoff = np.arange(m + 1, dtype=np.int64)
                results.append(dict(img_id=int(i), nb_ids=np.asarray(sc.neighbors[int(i)], np.int32),
                                    line=np.full((m, 10), float(c.index)), score=np.full(m, float(rank)),
                                    src=np.zeros((m, 2), np.int32), n_tris=np.full(m, c.index, np.int32), edge_off=eoff,
                                    edges=np.zeros((m, 2), np.int32)))
        from limap_amd import dist as ltdist
        # one blob per chunk, as StreamedTriangulation keeps them
        blobs, k0 = [], 0
        for c in mine:
            blobs.append(ltdist.pack_image_results(results[k0:k0 + len(c.images)]))
            k0 += len(c.images)
        others = ltstream.gather_results(blobs, rank, world, torch.device("cpu"))
        if rank == 0:
            got_other = [r for part in others for r in ltdist.unpack_image_results(*part)]
            got = sorted([r["img_id"] for r in results] + [r["img_id"] for r in got_other])
            ok = ok and got == sc.img_ids.tolist() and len(others) == world - 1
            by_img = {int(i): c for c in plan for i in c.images}
            for r in got_other:
                c = by_img[r["img_id"]]
                ok = ok and c.rank != 0 and float(r["score"][0]) == float(c.rank) and int(r["n_tris"][0]) == c.index
        else:
            ok = ok and others is None
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_streamed_chunks_round_robin_and_one_gather(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + world + (os.getpid() % 500)
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
    assert res == [(r, True) for r in range(world)], res
