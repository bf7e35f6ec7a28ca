/* rccl_host.c -- the multi-GPU triangulation path driven from PLAIN C through the two C ABIs alone
 * (include/limap_amd.h, include/limap_amd_rccl.h): one process per GPU, images sharded in id order, ONE ncclAllGather of
 * the per-image payload, generation + scoring per rank with no further collective, ONE grouped send / receive of the shards
 * to rank 0, ComputeLineTracks there (the reference is one process: runners/line_triangulation.py:158-168).
 *
 *   rccl_host <rank> <world> <scene.bin> <id_file> [out.txt]
 *
 * scene.bin (written by tools/write_scene_bin.py; little endian): int64 n_img, G, n_blocks, n_rows | int32 img_ids[n_img] |
 * int64 seg_off[n_img + 1] | f64 kvec[n_img][4] qvec[n_img][4] tvec[n_img][3] segs[G][4] | f64 lo[3] hi[3] |
 * int64 nb_off[n_img + 1] | int32 nb_ids[n_blocks] | int64 m_off[n_blocks + 1] | int32 rows[n_rows][2].
 * id_file: rank 0 writes the ncclUniqueId there (then renames it into place), the others wait for it.
 * Rank 0 prints / writes: "tracks <T> members <M> fnv <hash of the member arrays>" -- tests/test_rccl_abi.py compares it
 * with the one-process result.  Build: see limap_amd/csrc/Makefile (target rccl_host); no Python, no torch, no C++. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "limap_amd.h"
#include "limap_amd_rccl.h"

#define DIE(...) do { fprintf(stderr, "rccl_host[%d]: ", g_rank); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(2); } while (0)
static int g_rank = -1;

static void *read_exact(FILE *f, size_t bytes) {
  void *p = malloc(bytes ? bytes : 1);
  if (!p || (bytes && fread(p, 1, bytes, f) != bytes)) DIE("short read of the scene file");
  return p;
}
static uint64_t fnv(uint64_t h, const void *data, size_t n) {
  const unsigned char *p = (const unsigned char *)data;
  for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
  return h;
}

int main(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "usage: rccl_host <rank> <world> <scene.bin> <id_file> [out.txt]\n"); return 2; }
  const int rank = atoi(argv[1]), world = atoi(argv[2]);
  g_rank = rank;
  FILE *f = fopen(argv[3], "rb");
  if (!f) DIE("cannot open %s", argv[3]);
  int64_t hdr[4];
  if (fread(hdr, 8, 4, f) != 4) DIE("short header");
  const int64_t n_img = hdr[0], G = hdr[1], n_blocks = hdr[2], n_rows = hdr[3];
  int32_t *img_ids = (int32_t *)read_exact(f, 4 * (size_t)n_img);
  int64_t *seg_off = (int64_t *)read_exact(f, 8 * (size_t)(n_img + 1));
  double *kvec = (double *)read_exact(f, 32 * (size_t)n_img), *qvec = (double *)read_exact(f, 32 * (size_t)n_img);
  double *tvec = (double *)read_exact(f, 24 * (size_t)n_img), *segs = (double *)read_exact(f, 32 * (size_t)G);
  double *ranges = (double *)read_exact(f, 48);
  int64_t *nb_off = (int64_t *)read_exact(f, 8 * (size_t)(n_img + 1));
  int32_t *nb_ids = (int32_t *)read_exact(f, 4 * (size_t)n_blocks);
  int64_t *m_off = (int64_t *)read_exact(f, 8 * (size_t)(n_blocks + 1));
  int32_t *rows = (int32_t *)read_exact(f, 8 * (size_t)n_rows);
  fclose(f);

  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) DIE("no HIP device");
  const int dev = rank % n_dev;
  if (hipSetDevice(dev) != hipSuccess) DIE("hipSetDevice(%d)", dev);
  hipStream_t st;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) DIE("hipStreamCreate");

  /* the communicator is the caller's: unique id through a file */
  ncclUniqueId id;
  if (rank == 0) {
    char tmp[4096];
    if (ncclGetUniqueId(&id) != ncclSuccess) DIE("ncclGetUniqueId");
    snprintf(tmp, sizeof(tmp), "%s.tmp", argv[4]);
    FILE *o = fopen(tmp, "wb");
    if (!o || fwrite(&id, sizeof(id), 1, o) != 1) DIE("cannot write %s", tmp);
    fclose(o);
    if (rename(tmp, argv[4]) != 0) DIE("rename");
  } else {
    int tries = 0;
    FILE *i = NULL;
    while (!(i = fopen(argv[4], "rb")) && tries++ < 6000) usleep(10000);
    if (!i || fread(&id, sizeof(id), 1, i) != 1) DIE("no unique id in %s", argv[4]);
    fclose(i);
  }
  ncclComm_t comm;
  if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess) DIE("ncclCommInitRank");

  lt_config cfg;
  lt_config_default(&cfg);
  lt_ctx *ctx = lt_create(&cfg, dev);
  if (!ctx) DIE("lt_create failed (no CPU fallback)");
  if (lt_set_ranges(ctx, ranges, ranges + 3)) DIE("lt_set_ranges: %s", lt_last_error(ctx));

  /* shard by image, balanced by the match rows every image brings */
  double *weights = (double *)malloc(8 * (size_t)n_img);
  for (int64_t i = 0; i < n_img; ++i) weights[i] = (double)(m_off[nb_off[i + 1]] - m_off[nb_off[i]]);
  lt_dist *d = lt_dist_create(ctx, comm, st, rank, world, (int)n_img, img_ids, seg_off, weights);
  if (!d) DIE("lt_dist_create");
#define CHK(call) do { if ((call) != LT_OK) DIE("%s: %s / %s", #call, lt_dist_last_error(d), lt_last_error(ctx)); } while (0)
  CHK(lt_dist_load_local(d, kvec, qvec, tvec, segs));
  CHK(lt_dist_all_gather_scene(d)); /* (1) the ONE collective before the run */
  int first = 0, last = 0;
  CHK(lt_dist_my_images(d, &first, &last));
  for (int i = first; i < last; ++i) {
    const int64_t b0 = nb_off[i], b1 = nb_off[i + 1];
    /* offsets relative to the image's first row */
    int64_t *off = (int64_t *)malloc(8 * (size_t)(b1 - b0 + 1));
    for (int64_t b = b0; b <= b1; ++b) off[b - b0] = m_off[b] - m_off[b0];
    CHK(lt_triangulate_image(ctx, img_ids[i], (int)(b1 - b0), nb_ids + b0, off, rows + 2 * m_off[b0]));
    free(off);
  }
  CHK(lt_upload(ctx));
  CHK(lt_run_device(ctx));
  int64_t n_keys = 0;
  CHK(lt_dist_merge_shards(d, 0, &n_keys)); /* (2) every shard to rank 0 */
  if (rank == 0) {
    CHK(lt_compute_tracks(ctx));
    const int64_t T = lt_num_tracks(ctx), M = lt_num_track_members(ctx);
    double *line7 = (double *)malloc(56 * (size_t)(T ? T : 1));
    int64_t *toff = (int64_t *)malloc(8 * (size_t)(T + 1));
    int32_t *ti = (int32_t *)malloc(4 * (size_t)(M ? M : 1)), *tl = (int32_t *)malloc(4 * (size_t)(M ? M : 1));
    int32_t *tn = (int32_t *)malloc(4 * (size_t)(M ? M : 1));
    double *ts = (double *)malloc(8 * (size_t)(M ? M : 1)), *t3 = (double *)malloc(80 * (size_t)(M ? M : 1));
    CHK(lt_get_tracks(ctx, line7, toff, ti, tl, tn, ts, t3));
    uint64_t h = 1469598103934665603ull;
    h = fnv(h, toff, 8 * (size_t)(T + 1));
    h = fnv(h, ti, 4 * (size_t)M);
    h = fnv(h, tl, 4 * (size_t)M);
    char line[256];
    snprintf(line, sizeof(line), "tracks %lld members %lld keys %lld fnv %016llx", (long long)T, (long long)M,
             (long long)n_keys, (unsigned long long)h);
    puts(line);
    if (argc > 5) {
      FILE *o = fopen(argv[5], "w");
      if (o) { fprintf(o, "%s\n", line); fclose(o); }
    }
  }
  lt_dist_destroy(d);
  lt_destroy(ctx);
  ncclCommDestroy(comm);
  return 0;
}
