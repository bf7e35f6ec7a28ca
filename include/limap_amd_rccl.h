/* limap_amd_rccl.h -- the multi-GPU exchange of the triangulation path for a C / C++ host (north_star:
 * "shard by image, a single RCCL all-gather of the per-image payload").  Companion of limap_amd.h in its own shared
 * library, liblimap_amd_rccl.so, which links librccl: a host that shards with torch.distributed (limap_amd/dist.py does
 * exactly what is declared here, through c10d) never loads it, and a process never holds two RCCL copies because of us.
 *
 * No reference counterpart: cvg/limap is one process (runners/line_triangulation.py:160-167 loops over the images).
 * TriangulateImage(img) reads only replicated data -- all poses and 2D segments, img's neighbours and matches -- and
 * writes only the results of img's own nodes (triangulation/global_line_triangulator.cc:138-151), so the images are
 * sharded in id order over the ranks; what has to be exchanged is
 *   (1) before the run: kvec[4] | qvec[4] | tvec[3] | segs[M, 4] of every rank's own images -- ONE ncclAllGather;
 *   (2) after the run: the shards' per-node results and valid-edge keys to the rank that runs ComputeLineTracks
 *       (global_line_triangulator.cc:234-351) -- one blob per rank, ONE grouped ncclSend / ncclRecv.
 * The communicator is the caller's (ncclCommInitRank ... in rccl.h); this library creates none.
 * Every function returns LT_OK or an lt_status code; lt_dist_last_error() holds the text (RCCL / HIP errors included).
 */
#ifndef LIMAP_AMD_RCCL_H
#define LIMAP_AMD_RCCL_H

#include "limap_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lt_dist lt_dist;

/* Contiguous blocks [bounds[r], bounds[r+1]) of the id-ordered image list, balanced by `weights` (e.g. connections per
 * image; NULL = equal): the rule of limap_amd.dist.shard_bounds, so that a C host and a Python host shard alike. */
int lt_dist_shard_bounds(int n_img, int world, const double *weights, int64_t *bounds /* world + 1 */);

/* ctx: this rank's context (not yet initialised with a scene).  rccl_comm: ncclComm_t of `world` ranks, this one `rank`.
 * hip_stream: the stream collectives AND kernels run on (it becomes the context's stream, lt_set_stream).
 * img_ids ascending, seg_off[n_img + 1] = first node of every image: the replicated layout of the scene. */
lt_dist *lt_dist_create(lt_ctx *ctx, void *rccl_comm, void *hip_stream, int rank, int world, int n_img,
                        const int32_t *img_ids, const int64_t *seg_off, const double *weights);
void lt_dist_destroy(lt_dist *d);
const char *lt_dist_last_error(const lt_dist *d);
/* this rank's images [first, last) as indices into img_ids */
int lt_dist_my_images(const lt_dist *d, int *first, int *last);

/* Host arrays of the WHOLE scene are accepted (kvec[n_img][4], qvec[n_img][4], tvec[n_img][3], segs[G][4], FP64); only
 * this rank's slice is copied to the device -- the rest arrives through the collective. */
int lt_dist_load_local(lt_dist *d, const double *kvec, const double *qvec, const double *tvec, const double *segs);

/* (1) ONE ncclAllGather of the packed slices.  First call: the gathered scene initialises the context (lt_init_device)
 * and its chunks are registered (lt_set_scene_chunks); later calls (a new batch of poses / segments loaded with
 * lt_dist_load_local): lt_refresh_scene_chunks rebuilds the invariants straight from the receive buffer.  Everything is
 * enqueued on the stream; nothing waits. */
int lt_dist_all_gather_scene(lt_dist *d);

/* (2) After lt_run_device on every rank, nothing read back: rank 0 receives one blob [64-byte header | node slices | keys]
 * from every other rank (grouped ncclSend / ncclRecv) and imports them (lt_shard_import); its lt_compute_tracks then covers
 * the whole scene with the device form of the tail.  key_cap > 0: room for that many valid-edge keys per rank (the sizes
 * ride in the header: one collective; a rank with more keys fails, and so does rank 0); key_cap <= 0: the key counts are
 * exchanged first (one more small ncclAllGather).  *n_keys_merged: keys on rank 0 afterwards (0 elsewhere). */
int lt_dist_merge_shards(lt_dist *d, int64_t key_cap, int64_t *n_keys_merged);

#ifdef __cplusplus
}
#endif
#endif
