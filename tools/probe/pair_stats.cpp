// tools/probe/pair_stats.cpp -- developer probe (CPU): where do the pairs of scoreOneNode end?  For every node of a scene
// (candidates from the oracle) it walks the ordered pairs (i, j) of the node the way k_score3 does -- conservative
// sweep guards, then the dense pair_score -- and counts at which gate of pair_score a dense pair dies.  Decides whether a
// staged dense evaluation (compaction between the gates) can pay.  Built by tools/probe/pair_stats.py.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../limap_amd/csrc/lt_geom.h"
using namespace lt;

extern "C" void build_cams(int n, const double *k4, const double *q4, const double *t3, Cam *out) {
  for (int i = 0; i < n; ++i) cam_build(k4 + 4 * i, q4 + 4 * i, t3 + 3 * i, out + i);
}
extern "C" int cam_bytes() { return (int)sizeof(Cam); }
extern "C" int score_cfg_bytes() { return (int)sizeof(ScoreCfg); }

// second entry: with the neighbour segments (seg4 per candidate) the 2D part gate by gate
extern "C" int pair_stats2(long long G, const long long *off, const double *line10, const double *seg4, const int *src_img,
                           const int *slot, const Cam *cams, const ScoreCfg *cfgp, double guard2, long long *out, int *dense_of /*[C] dense pairs with this i, or null*/) {
  const ScoreCfg &cfg = *cfgp;
  const LinkCfg2 &c = cfg.l2;
  memset(out, 0, 32 * 8);
  for (long long g = 0; g < G; ++g) {
    const long long o = off[g];
    const int n = (int)(off[g + 1] - o);
    for (int a = 0; a < n; ++a) {
      const double *li = line10 + 10 * (o + a);
      const d3 si = mk3(li[0], li[1], li[2]), ei = mk3(li[3], li[4], li[5]);
      const d3 di = unit(sub(ei, si));
      const double zs = li[6] + kEps, ze = li[7] + kEps;
      const double gs2 = zs > 0 ? guard2 * zs * zs : 1e300, ge2 = ze > 0 ? guard2 * ze * ze : 1e300;
      for (int b = 0; b < n; ++b) {
        if (b == a || slot[o + b] == slot[o + a]) continue;
        const double *lj = line10 + 10 * (o + b);
        const d3 sj = mk3(lj[0], lj[1], lj[2]), ej = mk3(lj[3], lj[4], lj[5]);
        const d3 dj = unit(sub(ej, sj));
        const double cc = fabs(dot(di, dj));
        if (cc < cfg.cos_guard || sqn(sub(si, sj)) > gs2 || sqn(sub(ei, ej)) > ge2) continue;
        out[0]++;  // dense
        if (dense_of) dense_of[o + a]++;
        const LinkCfg3 &c3 = cfg.l3;
        double s3 = dmin(1.0, gate(expscore(angle_deg_from_cos(cc), c3.th_angle * c3.mult), c3.score_th));
        if (s3 < c3.score_th) { out[1]++; continue; }
        const double ds = sqrt(sqn(sub(si, sj))), de = sqrt(sqn(sub(ei, ej)));
        s3 = dmin(s3, gate(expscore(dmax(ds / zs, de / ze), c3.th_scaleinv * c3.mult), c3.score_th));
        if (s3 == 0) { out[2]++; continue; }
        const Cam &cj = cams[src_img[o + b]];
        const L2 l1{cam_project(cj, si), cam_project(cj, ei)};
        const double *sg = seg4 + 4 * (o + b);
        const L2 l2{mk2(sg[0], sg[1]), mk2(sg[2], sg[3])};
        double score = 1.0, ang = angle_between(l1, l2);
        score = dmin(score, gate(expscore(ang, c.th_angle * c.mult), c.score_th));
        if (score < c.score_th) { out[3]++; continue; }  // 2D angle
        const double ov = bioverlap(l1, l2);
        score = dmin(score, ov > c.th_overlap ? 1.0 : 0.0);
        if (score < c.score_th) { out[4]++; continue; }  // overlap
        double th = c.th_angle;
        if (ov < c.th_smartoverlap) {
          double ratio = dmin((c.th_smartoverlap - ov) / (c.th_smartoverlap - c.th_overlap), 1.0);
          th = c.th_angle - ratio * (c.th_angle - c.th_smartangle);
        }
        score = dmin(score, gate(expscore(ang, th * c.mult), c.score_th));
        if (score < c.score_th) { out[5]++; continue; }  // smart angle
        score = dmin(score, gate(expscore(perp_dist(l1, l2), c.th_perp * c.mult), c.score_th));
        if (score < c.score_th) { out[6]++; continue; }  // perpendicular distance
        out[7]++;  // positive score
        // cross-check with the shared function
        if (score2d(c, l1, l2) != score) out[31]++;
      }
    }
  }
  return 0;
}
