"""Streamed triangulation of a LARGE model: the library path behind BASELINE.json configs[4] ("Rome16K / large COLMAP
model (>= 5k images) streamed triangulation, 8 GPUs"; reference caller: runners/rome16k/triangulation.py:15-45 ->
limap.runners.line_triangulation, whose per-image loop -- src/limap/runners/line_triangulation.py:158-168 -- reads one
image's matches from disk, triangulates it and forgets the matches again).

`TriangulateImage(img)` reads the 2D segments and poses of `img` and of its neighbours only, and writes only `img`'s own
per-node results (global_line_triangulator.cc:138-151).  So a large model need not be resident as a whole:

  * the id-ordered image list is cut into CHUNKS of `chunk_images` consecutive images;
  * a chunk's CLOSURE = its images + every neighbour of one of them; a worker context is initialised with the closure
    ONLY (`Init` on the sub-scene: segments, poses and derived tables of ~chunk + halo images in HBM, not of the model),
    takes the chunk's match rows, runs generation + scoring, and hands back the chunk's per-image results in terms of
    IMAGE IDS and line ids (`lt_export_image_results`: best candidate per line, candidate counts, valid edges);
  * chunks are dealt ROUND-ROBIN to the ranks of the job (chunk k -> rank k % world): no data-path collective while
    the chunks run; at the end every rank's packed results go to rank 0 through ONE `gather`
    (`dist.gather_packed_to_rank0`) and rank 0 -- whose ACCUMULATOR context holds the whole model's segments, as the
    reference's triangulator does -- imports them (`lt_import_image_results`) and runs `ComputeLineTracks` once.

Results do not depend on the chunking: the closure holds everything `TriangulateImage` reads, image and neighbour
ids ascend in the sub-scene as in the model, valid edges name (neighbour slot, line) against the image's own neighbour
list.  tests/test_gpu_stream.py holds a 1000 x 600 scene in 4+ chunks to committed whole-scene digests of the CPU checker.
"""
import time

import numpy as np


class Chunk:
    """One unit of a streamed run: `images` (ascending ids) are triangulated, `closure` (ascending ids, a superset) is what
    the worker context holds; `rank` = who runs it."""
    __slots__ = ("index", "rank", "images", "closure")

    def __init__(self, index, rank, images, closure):
        self.index, self.rank, self.images, self.closure = index, rank, images, closure

    def __repr__(self):
        return f"Chunk({self.index}, rank={self.rank}, images={len(self.images)}, closure={len(self.closure)})"


def plan_chunks(img_ids, neighbors, chunk_images, world=1):
    """Cuts the ascending image list into chunks of `chunk_images` consecutive images, chunk k to rank k % world, each with
    its neighbour closure.  `neighbors`: {img_id: iterable of neighbour ids}.  Deterministic: every rank computes the same
    plan, nothing is exchanged."""
    ids = np.sort(np.asarray(img_ids, np.int64))
    if chunk_images <= 0:
        raise ValueError("chunk_images must be positive")
    known = set(int(i) for i in ids)
    chunks = []
    for k, a in enumerate(range(0, len(ids), int(chunk_images))):
        images = ids[a:a + int(chunk_images)]
        clo = set(int(i) for i in images)
        for i in images:
            for n in neighbors[int(i)]:
                n = int(n)
                if n not in known:
                    raise ValueError(f"image {int(i)} names neighbour {n}, which is not an image of the model")
                clo.add(n)
        chunks.append(Chunk(k, k % max(int(world), 1), images.astype(np.int32), np.array(sorted(clo), np.int32)))
    return chunks


def closure_arrays(chunk, img_ids, kvec, qvec, tvec, seg_off, segs):
    """The sub-scene of a chunk's closure as `Init` arrays: (ids, kvec, qvec, tvec, seg_off, segs) -- gathered from the
    model's arrays (ascending `img_ids`, CSR `seg_off` / `segs`)."""
    ids = np.asarray(img_ids)
    idx = np.searchsorted(ids, chunk.closure)
    assert np.array_equal(ids[idx], chunk.closure), "closure names an image the model does not have"
    seg_off = np.asarray(seg_off, np.int64)
    cnt = seg_off[idx + 1] - seg_off[idx]
    off = np.zeros(len(idx) + 1, np.int64)
    off[1:] = np.cumsum(cnt)
    # one gather of the segment rows: row r of closure image j is model row seg_off[idx[j]] + r
    rows = np.repeat(seg_off[idx] - off[:-1], cnt) + np.arange(int(off[-1]), dtype=np.int64)
    return (chunk.closure, np.ascontiguousarray(np.asarray(kvec)[idx]), np.ascontiguousarray(np.asarray(qvec)[idx]),
            np.ascontiguousarray(np.asarray(tvec)[idx]), off, np.ascontiguousarray(np.asarray(segs)[rows]))


def merge_blobs(blobs):
    """Several packed result blobs (int32, float64) -> one: the counts add up, the bodies follow each other."""
    if not blobs:
        return np.zeros(1, np.int32), np.zeros(0, np.float64)
    n = sum(int(b[0][0]) for b in blobs)
    return (np.concatenate([np.array([n], np.int32)] + [np.asarray(b[0], np.int32)[1:] for b in blobs]),
            np.concatenate([np.asarray(b[1], np.float64) for b in blobs]))


def gather_results(blobs, rank, world, device=None):
    """The streamed job's ONE collective: every rank's packed per-image results (a list of blobs, one per chunk) to rank 0
    (`gather` of two tensors, dist.gather_packed_to_rank0).  Returns the other ranks' blobs -- one (ints, dbls) pair per
    rank, ready for `import_images_packed` -- on rank 0, None elsewhere."""
    from . import dist as ltdist
    import torch
    import torch.distributed as dist
    if device is None:
        backend = dist.get_backend()
        device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    ints, flts = merge_blobs(blobs if rank != 0 else [])
    parts = ltdist.gather_packed_to_rank0(ints, flts, rank, world, device)
    if rank != 0:
        return None
    return [parts[r] for r in range(1, world)]


class StreamedTriangulation:
    """Runs this rank's chunks of a streamed job and, on rank 0, the tail over the whole model.

        st = StreamedTriangulation(cfg, img_ids, kvec, qvec, tvec, seg_off, segs, neighbors, ranges,
                                   chunk_images=250, rank=rank, world=world, device=local_rank)
        for ch in st.my_chunks():
            st.run_chunk(ch, matches_of)       # matches_of(img_id) -> {neighbour id: int32 (K, 2)}
        tracks_ctx = st.finish()               # rank 0: the accumulator context after ComputeLineTracks; else None

    `per_chunk` collects, per chunk: images, closure size, connections, candidates, host ms of Init / buffering / upload /
    download+export, device ms of the run and -- with `fine_timers` -- of stage A, stage B and the scoring stage.
    """

    def __init__(self, cfg, img_ids, kvec, qvec, tvec, seg_off, segs, neighbors, ranges=None, chunk_images=250, rank=0,
                 world=1, device=0, comm_device=None, accumulate=True):
        from . import _capi
        self._capi = _capi
        self.cfg = cfg
        self.img_ids = np.asarray(img_ids, np.int32)
        assert np.all(np.diff(self.img_ids) > 0), "image ids must be ascending"
        self.kvec, self.qvec, self.tvec = np.asarray(kvec, np.float64), np.asarray(qvec, np.float64), np.asarray(tvec, np.float64)
        self.seg_off, self.segs = np.asarray(seg_off, np.int64), np.asarray(segs, np.float64)
        self.neighbors = neighbors
        self.ranges = ranges
        self.rank, self.world, self.device, self.comm_device = int(rank), int(world), device, comm_device
        self.chunks = plan_chunks(self.img_ids, neighbors, chunk_images, world)
        self.worker = _capi.Context(cfg_dict=cfg, device=device)
        if ranges is not None:
            self.worker.set_ranges(*ranges)
        self.acc = None
        self.accumulate = accumulate
        self.results = []      # this rank's packed results, one blob per chunk (rank 0 imports its own as it goes)
        self.per_chunk = []
        self.n_imported = 0

    def my_chunks(self):
        return [c for c in self.chunks if c.rank == self.rank]

    def _accumulator(self):
        """rank 0's whole-model context (created on first use: the other ranks never hold the model on their device)"""
        if self.acc is None:
            self.acc = self._capi.Context(cfg_dict=self.cfg, device=self.device)
            if self.ranges is not None:
                self.acc.set_ranges(*self.ranges)
            self.acc.init(self.img_ids, self.kvec, self.qvec, self.tvec, self.seg_off, self.segs)
        return self.acc

    def run_chunk(self, chunk, matches_of, fine_timers=False):
        """Init on the chunk's closure, TriangulateImage for its images, run, export.  Returns the chunk's record."""
        import os
        W = self.worker
        rec = {"chunk": chunk.index, "images": int(len(chunk.images)), "closure_images": int(len(chunk.closure))}
        t = time.perf_counter()
        ids, k, q, tv, off, sg = closure_arrays(chunk, self.img_ids, self.kvec, self.qvec, self.tvec, self.seg_off, self.segs)
        W.init(ids, k, q, tv, off, sg)
        rec["closure_segments"] = int(off[-1])
        rec["init_ms"] = 1e3 * (time.perf_counter() - t)
        # the chunk's match rows in ONE call (lt_triangulate_all_rows: one pass of the host team over all blocks; the per-image
        # form cost 66 us of Python and ctypes per image, 25x the chunk's device time)
        t = time.perf_counter()
        ms = [matches_of(int(i)) for i in chunk.images]  # stands in for reading matches_{id}.npy (line_triangulation.py:160-165)
        rec["matches_ms"] = 1e3 * (time.perf_counter() - t)
        t = time.perf_counter()
        nbs, arrs = [], []
        for m in ms:
            nbs.append([int(x) for x in m.keys()])
            arrs.append([a if (a.dtype == np.int32 and a.flags.c_contiguous and a.ndim == 2) else
                         np.ascontiguousarray(np.asarray(a).reshape(-1, 2), np.int32) for a in m.values()])
        W.triangulate_all_rows(chunk.images, nbs, arrs)
        rec["buffer_ms"] = 1e3 * (time.perf_counter() - t)
        t = time.perf_counter()
        W.upload()
        rec["upload_ms"] = 1e3 * (time.perf_counter() - t)
        prev = os.environ.get("LT_FINE_TIMERS")
        if fine_timers:
            os.environ["LT_FINE_TIMERS"] = "2"  # read per run: stage A / stage B carry their own events too
        try:
            t = time.perf_counter()
            W.run_device()
            rec["run_wall_ms"] = 1e3 * (time.perf_counter() - t)
        finally:
            if fine_timers:
                if prev is None:
                    os.environ.pop("LT_FINE_TIMERS", None)
                else:
                    os.environ["LT_FINE_TIMERS"] = prev
        tm = W.timers()
        rec["device_ms"] = float(tm["run"])
        for key in ("gen", "place", "score", "select", "k_gates", "k_tri_rows", "k_score3", "survivors", "line_slots"):
            if key in tm:
                rec[key] = float(tm[key])
        st = W.stats()
        rec["connections"], rec["candidates"] = int(st["connections"]), int(st["candidates"])
        rec["valid_edges"] = int(st["valid_edges"])
        t = time.perf_counter()
        W.download()
        blob = W.export_images_packed(chunk.images)  # one call, two flat arrays (lt_export_images_packed)
        if self.rank == 0 and self.accumulate:
            self._accumulator().import_images_packed(*blob)
            self.n_imported += len(chunk.images)
        else:
            self.results.append(blob)
        rec["export_ms"] = 1e3 * (time.perf_counter() - t)
        self.per_chunk.append(rec)
        return rec

    def finish(self):
        """Every rank's results to rank 0 (one `gather` when world > 1), then ComputeLineTracks there.  Returns the
        accumulator context on rank 0, None elsewhere."""
        if self.world > 1:
            others = gather_results(self.results, self.rank, self.world, self.comm_device)
            if self.rank == 0:
                A = self._accumulator()
                for ints, dbls in others:
                    A.import_images_packed(ints, dbls)
                    self.n_imported += int(ints[0])
        if self.rank != 0:
            return None
        A = self._accumulator()
        if self.n_imported != len(self.img_ids):
            raise RuntimeError(f"streamed run: {self.n_imported} of {len(self.img_ids)} images arrived on rank 0")
        A.compute_tracks()
        return A
