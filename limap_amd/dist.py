"""Multi-GPU sharding of the triangulation path: one process per GPU, images sharded by rank,
ONE all-gather (RCCL over xGMI; gloo on CPU for tests) of the per-image payload before scoring.

`TriangulateImage(img)` reads only replicated data (all 2D segments, all poses, the neighbours and
matches of `img`) and writes only the results of `img`'s own nodes
(global_line_triangulator.cc:138-151 of the reference), so after the exchange every rank runs
generation + scoring for its shard with no further communication; the serial tail
(`ComputeLineTracks`) runs once on rank 0 over the gathered per-node results.

What travels: kvec[4] | qvec[4] | tvec[3] | segs[M,4] (FP64) of each rank's own images.  What is
replicated on the host beforehand: the image ids and the per-image segment counts (the layout).
"""
import numpy as np


def shard_bounds(n_items, world, weights=None):
    """Contiguous blocks [b[r], b[r+1]) of the id-ordered image list, balanced by `weights`
    (e.g. connections per image) -- contiguity keeps the gathered arrays a plain concatenation."""
    if weights is None:
        weights = np.ones(n_items)
    w = np.asarray(weights, float)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        b = int(np.searchsorted(cum, target, side="left"))
        b = min(max(b, bounds[-1]), n_items)
        bounds.append(b)
    bounds.append(n_items)
    return bounds


def shard_images(img_ids, rank, world, weights=None):
    ids = np.sort(np.asarray(img_ids))
    b = shard_bounds(len(ids), world, weights)
    return ids[b[rank]:b[rank + 1]]


class SceneGather:
    """Packs this rank's images into one buffer, all-gathers once, unpacks into the global
    kvec / qvec / tvec / segs arrays (ascending image id) on the device."""

    def __init__(self, img_ids, seg_off, rank, world, device, weights=None, force_collective=False):
        import torch
        self.torch = torch
        self.rank, self.world, self.device = rank, world, device
        # force_collective: run the all-gather even in a one-rank job (smoke test of the RCCL path on one GPU)
        self.collective = world > 1 or force_collective
        ids = np.asarray(img_ids)
        assert np.all(np.diff(ids) > 0), "image ids must be ascending"
        self.n_img = len(ids)
        self.seg_off = np.asarray(seg_off, np.int64)
        self.bounds = shard_bounds(self.n_img, world, weights)
        # per-rank payload sizes in doubles
        self.sizes = []
        for r in range(world):
            a, b = self.bounds[r], self.bounds[r + 1]
            self.sizes.append(11 * (b - a) + 4 * int(self.seg_off[b] - self.seg_off[a]))
        self.max_size = max(self.sizes) if self.sizes else 0
        G = int(self.seg_off[-1])
        self.local = torch.zeros(max(self.max_size, 1), dtype=torch.float64, device=device)
        self.recv = torch.zeros(max(self.max_size, 1) * world, dtype=torch.float64, device=device)
        self.kvec = torch.zeros((self.n_img, 4), dtype=torch.float64, device=device)
        self.qvec = torch.zeros((self.n_img, 4), dtype=torch.float64, device=device)
        self.tvec = torch.zeros((self.n_img, 3), dtype=torch.float64, device=device)
        self.segs = torch.zeros((max(G, 1), 4), dtype=torch.float64, device=device)

    def load_local(self, kvec, qvec, tvec, segs):
        """Host arrays of the WHOLE scene are accepted for convenience; only this rank's slice is
        copied to the device."""
        torch = self.torch
        a, b = self.bounds[self.rank], self.bounds[self.rank + 1]
        s0, s1 = int(self.seg_off[a]), int(self.seg_off[b])
        buf = np.concatenate([np.asarray(kvec, np.float64)[a:b].reshape(-1), np.asarray(qvec, np.float64)[a:b].reshape(-1),
                              np.asarray(tvec, np.float64)[a:b].reshape(-1), np.asarray(segs, np.float64)[s0:s1].reshape(-1)])
        assert len(buf) == self.sizes[self.rank]
        self.local[:len(buf)].copy_(torch.from_numpy(buf))

    def gather_only(self):
        """The collective alone: afterwards `chunk_pointers()` describe the scene in place."""
        if self.collective:
            import torch.distributed as dist
            dist.all_gather_into_tensor(self.recv, self.local)

    def gather_async(self):
        """Launch the collective without waiting (returns the c10d work handle, or None for a one-rank job
        without a forced collective).  `handle.wait()` makes the CURRENT stream wait for the gathered data;
        the collective itself is ordered after everything already enqueued on the current stream, so it may
        be launched as soon as the previous contents of the receive buffer have been consumed."""
        if not self.collective:
            return None
        import torch.distributed as dist
        return dist.all_gather_into_tensor(self.recv, self.local, async_op=True)

    def chunk_pointers(self):
        """(img_begin, k_ptrs, q_ptrs, t_ptrs, s_ptrs): per-rank views into the receive buffer
        (the local buffer for world == 1), for lt_set_scene_chunks.  Empty shards are skipped."""
        buf = self.recv if self.collective else self.local
        base = buf.data_ptr()
        ib, pk, pq, pt, ps = [], [], [], [], []
        for r in range(self.world):
            a, b = self.bounds[r], self.bounds[r + 1]
            n = b - a
            if n == 0 and r > 0:
                continue
            o = base + 8 * r * max(self.max_size, 1)
            ib.append(a); pk.append(o); pq.append(o + 8 * 4 * n); pt.append(o + 8 * 8 * n); ps.append(o + 8 * 11 * n)
        return ib, pk, pq, pt, ps

    def all_gather(self):
        """One collective; returns (kvec, qvec, tvec, segs) device tensors of the whole scene."""
        torch = self.torch
        if self.collective:
            import torch.distributed as dist
            dist.all_gather_into_tensor(self.recv, self.local)
            recv = self.recv
        else:
            recv = self.local
        for r in range(self.world):
            a, b = self.bounds[r], self.bounds[r + 1]
            n = b - a
            s0, s1 = int(self.seg_off[a]), int(self.seg_off[b])
            base = r * max(self.max_size, 1)
            if n == 0:
                continue
            self.kvec[a:b].copy_(recv[base:base + 4 * n].view(n, 4))
            self.qvec[a:b].copy_(recv[base + 4 * n:base + 8 * n].view(n, 4))
            self.tvec[a:b].copy_(recv[base + 8 * n:base + 11 * n].view(n, 3))
            if s1 > s0:
                self.segs[s0:s1].copy_(recv[base + 11 * n:base + 11 * n + 4 * (s1 - s0)].view(s1 - s0, 4))
        return self.kvec, self.qvec, self.tvec, self.segs


def pack_image_results(results):
    """Per-image result dicts (`Context.export_image_results`) -> (int32 blob, float64 blob).
    int32: n_images, then per image  img_id, n_nb, m, ne, nb_ids[n_nb], src[m,2], n_tris[m], edge_cnt[m], edges[ne,2];
    float64: per image  line[m,10], score[m]  (104 B best candidate + score per node, 4 B per valid edge ...)."""
    ints, flts = [np.array([len(results)], np.int32)], []
    for r in results:
        nb = np.asarray(r["nb_ids"], np.int32).reshape(-1)
        m = len(r["score"])
        eoff = np.asarray(r["edge_off"], np.int64).reshape(-1)
        edges = np.asarray(r["edges"], np.int32).reshape(-1, 2)
        ints += [np.array([r["img_id"], len(nb), m, len(edges)], np.int32), nb,
                 np.asarray(r["src"], np.int32).reshape(-1), np.asarray(r["n_tris"], np.int32).reshape(-1),
                 np.diff(eoff).astype(np.int32), edges.reshape(-1)]
        flts += [np.asarray(r["line"], np.float64).reshape(-1), np.asarray(r["score"], np.float64).reshape(-1)]
    return np.concatenate(ints), (np.concatenate(flts) if flts else np.zeros(0))


def unpack_image_results(ints, flts):
    """Inverse of pack_image_results."""
    ints, flts = np.asarray(ints, np.int32), np.asarray(flts, np.float64)
    n, ip, fp, out = int(ints[0]), 1, 0, []
    for _ in range(n):
        img_id, n_nb, m, ne = (int(x) for x in ints[ip:ip + 4]); ip += 4
        nb = ints[ip:ip + n_nb].copy(); ip += n_nb
        src = ints[ip:ip + 2 * m].reshape(m, 2).copy(); ip += 2 * m
        nt = ints[ip:ip + m].copy(); ip += m
        eoff = np.zeros(m + 1, np.int64); eoff[1:] = np.cumsum(ints[ip:ip + m]); ip += m
        edges = ints[ip:ip + 2 * ne].reshape(ne, 2).copy(); ip += 2 * ne
        line = flts[fp:fp + 10 * m].reshape(m, 10).copy(); fp += 10 * m
        score = flts[fp:fp + m].copy(); fp += m
        out.append(dict(img_id=img_id, nb_ids=nb, line=line, score=score, src=src, n_tris=nt, edge_off=eoff, edges=edges))
    return out


def gather_packed_to_rank0(ints, flts, rank, world, device):
    """The second, small collective of SURVEY 8(e): every rank's packed per-node results to rank 0 ONLY
    (`gather`, padded to the largest shard; the sizes travel first in one tiny all-gather).  Returns the list
    of (ints, flts) per rank on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    sizes = torch.tensor([len(ints), len(flts)], dtype=torch.int64, device=device)
    all_sizes = torch.zeros((world, 2), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(all_sizes.view(-1), sizes)
    all_sizes = all_sizes.cpu().numpy()
    mi, mf = int(all_sizes[:, 0].max()), max(int(all_sizes[:, 1].max()), 1)
    ti = torch.zeros(mi, dtype=torch.int32, device=device)
    ti[:len(ints)].copy_(torch.from_numpy(np.ascontiguousarray(ints, np.int32)))
    tf = torch.zeros(mf, dtype=torch.float64, device=device)
    if len(flts):
        tf[:len(flts)].copy_(torch.from_numpy(np.ascontiguousarray(flts, np.float64)))
    gi = [torch.zeros_like(ti) for _ in range(world)] if rank == 0 else None
    gf = [torch.zeros_like(tf) for _ in range(world)] if rank == 0 else None
    dist.gather(ti, gi, dst=0)
    dist.gather(tf, gf, dst=0)
    if rank != 0:
        return None
    return [(gi[r][:int(all_sizes[r, 0])].cpu().numpy(), gf[r][:int(all_sizes[r, 1])].cpu().numpy()) for r in range(world)]


def merge_shards_device(ctx, node_range, rank, world, device=None, all_ranges=None, key_cap=None):
    """The shards' way to rank 0 WITHOUT the host -- every rank builds the undirected valid-edge keys of its own nodes on
    its device, the per-node results are slices [g_lo, g_hi) of the arrays the device tail reads (images are sharded in
    id order); rank 0 copies what arrives into place on its device (`lt_shard_import`) and its `compute_tracks()` then
    runs the device form of the tail over the whole scene.  No per-image export, no numpy packing, no host tail.

    Since round 5 everything a rank sends is ONE blob -- a 64-byte header (key count, node range, truncation flag), the
    node slices, the keys -- through ONE tensor `gather`:
      * default: a small all-gather of (key count, node range) first, so that the blobs are padded to the largest shard
        exactly: 2 collectives per merge (round 4: 3, the node slices and the keys as separate gathers);
      * `all_ranges` (every rank's (g_lo, g_hi): the sharding is deterministic, every rank can compute it) together with
        `key_cap` (keys per rank the blob has room for, the same number on every rank): no size exchange, the header
        carries what rank 0 needs -- 1 collective.  A rank with more keys than `key_cap` cannot send them: it and rank 0
        both raise after the gather (nobody hangs, the merge fails loudly); choose the capacity from a bound the job
        knows (valid edges per node x nodes of the largest shard).
    With the node filter on (`min_num_outer_edges > 0`) the keys travel DIRECTED (source, target) and rank 0 runs
    `filterNodeByNumOuterEdges` over the merged list on its device (round 6; refused before).
    node_range = (g_lo, g_hi): this rank's node range.  Under a gloo group the buffers are host tensors (the copies in
    `lt_shard_export` / `_import` take either).  Returns the number of keys merged on rank 0 (0 elsewhere)."""
    if world == 1:
        return 0
    import torch
    import torch.distributed as dist
    if device is None:
        backend = dist.get_backend() if dist.is_initialized() else "gloo"
        device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    g_lo, g_hi = int(node_range[0]), int(node_range[1])
    n_keys = ctx.shard_count()
    one_collective = all_ranges is not None and key_cap is not None
    if one_collective:
        ranges = np.asarray(all_ranges, np.int64).reshape(world, 2)
        max_keys = max(int(key_cap), 1)
        counts = None
    else:
        mine = torch.tensor([n_keys, g_lo, g_hi], dtype=torch.int64, device=device)
        allv = torch.zeros(3 * world, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(allv, mine)
        allv = allv.cpu().numpy().reshape(world, 3)
        ranges, counts = allv[:, 1:3], allv[:, 0]
        max_keys = max(int(counts.max()), 1)
    nb = ctx.shard_node_bytes()
    max_nodes = max(int((ranges[:, 1] - ranges[:, 0]).max()), 1)
    truncated = rank != 0 and n_keys > max_keys  # (rank 0 sends no keys: its own are already in place)
    o_nodes, o_keys = 64, 64 + ((max_nodes * nb + 63) // 64) * 64
    blob = torch.empty(o_keys + max_keys * 8, dtype=torch.uint8, device=device)
    hdr = torch.tensor([n_keys, g_lo, g_hi, int(truncated), 0, 0, 0, 0], dtype=torch.int64)
    blob[:64].copy_(hdr.view(torch.uint8))
    if counts is not None:
        ctx.shard_build(int(counts.sum()) if rank == 0 else n_keys)
    elif rank != 0:
        ctx.shard_build(n_keys)
    if rank != 0 and not truncated:  # rank 0 sends nothing but its header: its own slices are already in place
        ctx.shard_export(g_lo, g_hi, blob.data_ptr() + o_nodes, blob.data_ptr() + o_keys)
    got = [torch.empty_like(blob) for _ in range(world)] if rank == 0 else None
    dist.gather(blob, got, dst=0)
    if truncated:
        raise RuntimeError(f"merge_shards_device: rank {rank} has {n_keys} valid-edge keys, the blob has room for {max_keys} "
                           "(key_cap): raise the capacity or let the merge exchange the sizes (key_cap=None)")
    if rank != 0:
        return 0
    if device.type == "cuda":
        torch.cuda.synchronize(device)  # the gathered blobs are read on the context's own stream
    if counts is None:  # one collective: the headers carry the counts
        heads = torch.stack([g[:64] for g in got]).cpu().numpy().view(np.int64).reshape(world, 8)
        if heads[:, 3].any():
            bad = [int(r) for r in np.nonzero(heads[:, 3])[0]]
            raise RuntimeError(f"merge_shards_device: ranks {bad} could not send their keys (key_cap = {max_keys} too small)")
        counts = heads[:, 0].copy()
        counts[0] = n_keys  # rank 0's own keys are already in place
        # the headers are the peers' word: a count beyond the blob's key room would make lt_shard_import read past it,
        # another node range would put the slices in the wrong place (a peer that sharded with different weights)
        for r in range(1, world):
            if not (0 <= counts[r] <= max_keys) or heads[r, 1] != ranges[r, 0] or heads[r, 2] != ranges[r, 1]:
                raise RuntimeError(f"merge_shards_device: rank {r}'s header says {int(counts[r])} keys for nodes "
                                   f"[{int(heads[r, 1])}, {int(heads[r, 2])}); expected at most {max_keys} keys for "
                                   f"[{int(ranges[r, 0])}, {int(ranges[r, 1])})")
        ctx.shard_build(int(counts.sum()))
    total = int(counts.sum())
    for r in range(1, world):
        ctx.shard_import(int(ranges[r, 0]), int(ranges[r, 1]), got[r].data_ptr() + o_nodes, int(counts[r]),
                         got[r].data_ptr() + o_keys)
    return total


def merge_shards_on_rank0(ctx, my_imgs, rank, world, device=None):
    """After every rank has triangulated its shard: ship the packed per-image results (best candidate + score
    per node, valid edges) to rank 0 with one tensor `gather` -- nothing is sent to the other ranks -- and
    import them into rank 0's context, which can then run `compute_tracks()` for the whole scene.
    Returns the number of images imported (0 on the other ranks)."""
    if world == 1:
        return 0
    import torch
    import torch.distributed as dist
    if device is None:
        # the collective's tensors live where the process group's backend can move them: gloo gathers host tensors,
        # nccl (= RCCL) device tensors
        backend = dist.get_backend() if dist.is_initialized() else "gloo"
        device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = [ctx.export_image_results(int(i)) for i in my_imgs] if rank != 0 else []
    ints, flts = pack_image_results(mine)
    parts = gather_packed_to_rank0(ints, flts, rank, world, device)
    n = 0
    if rank == 0:
        for r in range(1, world):
            for res in unpack_image_results(*parts[r]):
                ctx.import_image_results(res)
                n += 1
    return n
