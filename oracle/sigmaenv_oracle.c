/*
 * sigmaenv_oracle.c -- CPU restatement of SigmaRL's environment-step hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP product
 * (sigmarl_amd/csrc).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product never links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function below
 * against golden vectors produced by running the reference itself in the build
 * container (tests/golden/ npz files, generator tests/golden/gen/gen_golden.py).
 * Third-party pieces the reference pulls from packages absent from
 * /root/reference are restated from their published behaviour and are "unpinned":
 *   torchdiffeq==0.2.5 odeint(method="euler")  -> one explicit Euler step
 *   vmas==1.4.3 call order                     -> world.step; reward(a) for all a; observation(a) for all a; done()
 *
 * Arithmetic contract (shared with the HIP kernels, stated in DESIGN.md):
 *   - fp32 everywhere, one IEEE operation per reference torch op, no FMA contraction
 *     (-ffp-contract=off) EXCEPT torch.norm over a length-2 dim, which PyTorch-CPU
 *     evaluates as sqrt(fma(y, y, x*x)) [measured in the build container];
 *   - sin/cos/tan/atan are the correctly rounded fp32 value, (float) of a shared fp64 evaluation
 *     (include/sigma_trig_f32.h; the reference's vector math is within 1 ulp of it); atan2 likewise, through libm;
 *   - argmin/top-k ties resolve to the lowest index (torch.min semantics).
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference/sigmarl).
 */
#include "../include/sigmaenv.h"
#include "../include/sigma_trig_f32.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NS SIGMAENV_N_SHORT_TERM
#define PI32 3.14159274101257324f  /* float32(math.pi)   */
#define TWO_PI32 6.28318548202514648f /* float32(2*math.pi) */

typedef struct sigmaenv_oracle {
  sigmaenv_config_t cfg;
  int B, N, K, D, P, n_paths, yaw_stride;
  uint32_t observe_calls, obs_salt;  /* stand-alone observe calls so far; the salt of the observation being assembled (0 inside steps and resets) */
  float *center, *left, *right; /* [n_paths][P][2] padded as world_state_rt.py:313-420 */
  float* yaw;                   /* [n_paths][yaw_stride] */
  int32_t *n_center, *n_left, *n_right;
  uint8_t* is_loop;
  float *state, *prev_pos, *vertices, *short_term, *dist_ref, *dist_left, *dist_right, *dist_bound, *dist_agents;
  float *reward, *reward_info, *obs, *action, *cbf_nominal;
  int32_t *path, *closest, *nearing, *timer;
  uint8_t *col_agents, *col_flags, *done;
  sigmaenv_cbf_config_t cbf;    /* sigmaenv_oracle_cbf_attach */
  uint8_t* fresh;               /* [B,N] 1: the agent was (re)placed and has not been stepped since (the boundary points of the observation
                                 * are taken with another index shift then, world_state_rt.py:531-576 vs :686-724) */
  int32_t* cbf_groups;          /* [B,N] group index of every vehicle (grouped CBF-QPs), formed at the first sigmaenv_oracle_cbf_qp call */
  int cbf_groups_valid;
  float *seg_left, *seg_right;  /* [n_paths][seg_stride][5] */
  int seg_stride;
  float* lanelet_centers;       /* [n_lanelets][lanelet_pts][2] zero-padded centre lines of parser.lanelets_all (sigmaenv_oracle_set_lanelets) */
  uint64_t* lanelet_neigh;      /* [n_lanelets] bit j: lanelet j is in parser.neighboring_lanelets_idx[i] */
  int n_lanelets, lanelet_pts;
  const float* cbf_centers_inject;   /* test hook (sigmaenv_oracle_cbf_inject_centers): [B,N,C,2] circle centres that replace the computed ones, or NULL */
  int n_lists, list_first[4], list_count[4];   /* cpm_mixed sub-scenario path lists (sigmaenv_oracle_set_scenario_lists) */
  float list_cdf[4];
  char err[256];
} oracle_t;

/* ---- scalar helpers ------------------------------------------------------------------------------------------- */
/* sin / cos / tan / atan: correctly rounded fp32 through the shared fp64 algorithm (include/sigma_trig_f32.h states what was measured
 * about the reference's own trig); atan2 (observation only; the HIP path has none): correctly rounded through libm */
static inline float cr_sin(float x) { return sigma_sinf(x); }
static inline float cr_cos(float x) { return sigma_cosf(x); }
static inline float cr_tan(float x) { return sigma_tanf(x); }
static inline float cr_atan(float x) { return sigma_atanf(x); }
static inline float cr_atan2(float y, float x) { return (float)atan2((double)y, (double)x); }
/* torch.norm(..., dim=<len-2 dim>) on PyTorch-CPU == sqrt(fma(y,y,x*x)) */
static inline uint32_t rng_u32(uint64_t seed, uint64_t counter, uint32_t env, uint32_t agent, uint32_t draw);  /* below (device-side resets) */
static inline float norm2(float x, float y) { return sqrtf(fmaf(y, y, x * x)); }
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
/* torch.remainder(a, b), b > 0 (fmod, then shift negatives): dynamics.py:158, helper_scenario.py:1286-1289 */
static inline float remainder_pos(float a, float b) {
  float m = fmodf(a, b);
  if (m != 0.0f && m < 0.0f) m += b;
  return m;
}
static inline float angle_eliminate_two_pi(float a) { /* helper_scenario.py:1276-1289 */
  float r = remainder_pos(a, TWO_PI32);
  if (r > PI32) r -= TWO_PI32;
  return r;
}
/* decreasing_fcn(type="linear"), helper_scenario.py:960-996 */
static inline float decreasing_lin(float x, float x0, float x1) {
  x = clampf(x, x0, x1);
  float denom = x1 - x0;
  return 1.0f - (x - x0) / denom;
}

/* ---- K1: WorldCustom.step + KinematicBicycleModel, helper_training.py:797-861, dynamics.py:62-192 ----------------- */
static void bicycle_step(const sigmaenv_config_t* c, float* s /*8*/, const float* u_in, float* u_clamped) {
  float a0 = clampf(u_in[0], -c->max_speed, c->max_speed);        /* helper_training.py:807-812 */
  float a1 = clampf(u_in[1], -c->max_steering, c->max_steering);  /* :813-818 */
  u_clamped[0] = a0;
  u_clamped[1] = a1;
  float x = s[0], y = s[1], psi = s[2], v = s[3], delta = s[4];
  float u_acc = (a0 - v) / c->dt;                                  /* :821 */
  float u_sr = (a1 - delta) / c->dt;                               /* :822-824 */
  u_acc = clampf(u_acc, c->min_acc, c->max_acc);                   /* :829-831 */
  u_sr = clampf(u_sr, c->min_steering_rate, c->max_steering_rate); /* :832-836 */
  float l_wb = (float)((double)c->l_f + (double)c->l_r);
  float k_beta = (float)((double)c->l_r / ((double)c->l_f + (double)c->l_r));
  float beta = cr_atan(k_beta * cr_tan(delta));                    /* dynamics.py:103 */
  float dx0 = v * cr_cos(psi + beta);                              /* :107 */
  float dx1 = v * cr_sin(psi + beta);                              /* :108 */
  float dx2 = (v / l_wb) * cr_tan(delta) * cr_cos(beta);           /* :109-111 */
  float dt = c->dt;                                                /* t = linspace(0, dt, 2); one Euler step :149-156 */
  x = x + dt * dx0;
  y = y + dt * dx1;
  psi = psi + dt * dx2;
  v = v + dt * u_acc;
  delta = delta + dt * u_sr;
  delta = remainder_pos(delta + PI32, TWO_PI32) - PI32;            /* :158 */
  float beta1 = cr_atan(k_beta * cr_tan(delta));                   /* :161-163 */
  float course = psi + beta1;                                      /* :166 */
  s[0] = x; s[1] = y; s[2] = psi; s[3] = v; s[4] = delta;
  s[5] = v * cr_cos(course);                                       /* :167 */
  s[6] = v * cr_sin(course);                                       /* :168 */
  s[7] = beta1;
}

/* ---- K2: get_rectangle_vertices, helper_scenario.py:695-826 (is_close_shape=True) ------------------------------- */
static void rect_vertices(const sigmaenv_config_t* c, float px, float py, float psi, float* v /*5x2*/) {
  float lh = (float)((double)c->length / 2.0), wh = (float)((double)c->width / 2.0);
  const float bx[5] = {lh, lh, -lh, -lh, lh};
  const float by[5] = {wh, -wh, -wh, wh, wh};
  float cs = cr_cos(psi), sn = cr_sin(psi);
  float nsn = -sn;
  for (int k = 0; k < 5; ++k) { /* bmm [[c,-s],[s,c]] x [bx;by], plain mul/add (measured) then + centre :819-824 */
    v[2 * k] = (cs * bx[k] + nsn * by[k]) + px;
    v[2 * k + 1] = (sn * bx[k] + cs * by[k]) + py;
  }
}

/* ---- K3: get_perpendicular_distances, helper_scenario.py:829-889 ------------------------------------------------- */
/* Entries >= n-1 are overwritten with entry n-2 (:874-879), so (min, first argmin) runs over the n-1 real segments. */
static void point_polyline(float px, float py, const float* poly, int n, float* dist, int32_t* idx_plus1) {
  float best = INFINITY;
  int bi = 0;
  for (int k = 0; k + 1 < n; ++k) {
    float sx = poly[2 * k], sy = poly[2 * k + 1];
    float lx = poly[2 * k + 2] - sx, ly = poly[2 * k + 3] - sy; /* line_vecs :858 */
    float vx = px - sx, vy = py - sy;                           /* point_vecs :859 */
    float len2 = lx * lx + ly * ly;                             /* :862 */
    float proj = (vx * lx + vy * ly) / len2;                    /* :863 */
    float t = clampf(proj, 0.0f, 1.0f);                         /* :866 */
    float cx = sx + lx * t, cy = sy + ly * t;                   /* :869 */
    float d = norm2(cx - px, cy - py);                          /* :872 */
    if (d < best) { best = d; bi = k; }                         /* torch.min: first minimal index :883 */
  }
  *dist = best;
  *idx_plus1 = bi + 1; /* :885-887 */
}

/* ---- K6: get_short_term_reference_path, helper_scenario.py:892-957 (sample_interval=2, n_points_shift=1) ---------- */
static void short_term_path(const float* center, int n, int is_loop, int32_t cp, float* out /*NSx2*/) {
  for (int k = 0; k < NS; ++k) {
    int id = k * 2 + cp + 1;                                    /* :930-934 */
    if (is_loop && id >= n - 1) id = (id + 1) % n;              /* :941-947 */
    out[2 * k] = center[2 * id];
    out[2 * k + 1] = center[2 * id + 1];
  }
}

/* ---- K5: interX, helper_scenario.py:1148-1229 (is_return_points=False) -------------------------------------------- */
static int interx(const float* L1, int n1, const float* L2, int n2) {
  int hit = 0;
  for (int i = 0; i + 1 < n1; ++i) {
    float x1a = L1[2 * i], y1a = L1[2 * i + 1], x1b = L1[2 * i + 2], y1b = L1[2 * i + 3];
    float dx1 = x1b - x1a, dy1 = y1b - y1a;                     /* :1170 */
    float S1 = dx1 * y1a - dy1 * x1a;                           /* :1174 */
    for (int j = 0; j + 1 < n2; ++j) {
      float x2a = L2[2 * j], y2a = L2[2 * j + 1], x2b = L2[2 * j + 2], y2b = L2[2 * j + 3];
      float dx2 = x2b - x2a, dy2 = y2b - y2a;                   /* :1171 */
      float S2 = dx2 * y2a - dy2 * x2a;                         /* :1175 */
      float ma = dx1 * y2a - dy1 * x2a, mb = dx1 * y2b - dy1 * x2b;   /* :1183 */
      int C1 = ((ma - S1) * (mb - S1)) < 0.0f;                  /* D(), :1178-1187 */
      float wa = y1a * dx2 - x1a * dy2, wb = y1b * dx2 - x1b * dy2;   /* :1191 */
      int C2 = ((wa - S2) * (wb - S2)) < 0.0f;                  /* :1188-1196 */
      hit |= (C1 & C2);
    }
  }
  return hit;
}

/* ---- K4: get_distances_between_agents "mtv", helper_scenario.py:1030-1138 ----------------------------------------- */
static void rect_axes(const float* v, float ax[2][2]) { /* :1042-1045 */
  for (int k = 0; k < 2; ++k) {
    float ex = v[2 * (k + 1)] - v[2 * k], ey = v[2 * (k + 1) + 1] - v[2 * k + 1];
    float nrm = norm2(ex, ey);
    ax[k][0] = ex / nrm;
    ax[k][1] = ey / nrm;
  }
}
/* one direction: vertices of rectangle `a` against rectangle `b` on b's axes (:1059-1083) */
static void mtv_half(const float* va, const float* vb, float axb[2][2], float pos_out[4], float* omin_out, int* any_inside) {
  float maxbb[2], minbb[2], maxab[2], minab[2], pab[4][2];
  for (int k = 0; k < 2; ++k) {
    maxbb[k] = -INFINITY; minbb[k] = INFINITY; maxab[k] = -INFINITY; minab[k] = INFINITY;
    for (int v = 0; v < 4; ++v) {
      float pb = vb[2 * v] * axb[k][0] + vb[2 * v + 1] * axb[k][1];
      float pa = va[2 * v] * axb[k][0] + va[2 * v + 1] * axb[k][1];
      pab[v][k] = pa;
      maxbb[k] = fmaxf(maxbb[k], pb); minbb[k] = fminf(minbb[k], pb);
      maxab[k] = fmaxf(maxab[k], pa); minab[k] = fminf(minab[k], pa);
    }
  }
  float ov0 = fminf(maxbb[0], maxab[0]) - fmaxf(minbb[0], minab[0]); /* :1079 */
  float ov1 = fminf(maxbb[1], maxab[1]) - fmaxf(minbb[1], minab[1]);
  float omin = fminf(ov0, ov1);
  *omin_out = omin;
  int inside_any = 0;
  for (int v = 0; v < 4; ++v) {
    float g[2];
    int inside = 1;
    for (int k = 0; k < 2; ++k) {
      float p = pab[v][k];
      g[k] = (p - minbb[k]) * (p <= minbb[k] ? 1.0f : 0.0f) + (maxbb[k] - p) * (p >= maxbb[k] ? 1.0f : 0.0f); /* :1072-1076 */
      inside &= (p > minbb[k]) && (p < maxbb[k]);              /* :1081-1083 */
    }
    pos_out[v] = norm2(g[0], g[1]);                             /* :1077 */
    float neg = -omin * (inside ? 1.0f : 0.0f);                 /* :1080 */
    if (fabsf(neg) > 0.0f) inside_any = 1;                      /* :1118 */
  }
  *any_inside = inside_any;
}
static float mtv_pair(const float* vi, const float* vj) {
  float axi[2][2], axj[2][2], pij[4], pji[4], omin_j, omin_i;
  int neg_ij, neg_ji;
  rect_axes(vi, axi);
  rect_axes(vj, axj);
  mtv_half(vi, vj, axj, pij, &omin_j, &neg_ij);
  mtv_half(vj, vi, axi, pji, &omin_i, &neg_ji);
  float d = INFINITY;                                            /* :1112-1117 */
  for (int v = 0; v < 4; ++v) d = fminf(d, pij[v]);
  for (int v = 0; v < 4; ++v) d = fminf(d, pji[v]);
  if (neg_ij | neg_ji) d = -fminf(omin_j, omin_i);              /* :1118-1123 */
  return d;
}

/* update_mutual_distances, world_state_rt_sim.py:360-373 (diagonal := sqrt(x_semidim^2+y_semidim^2), helper_scenario.py:1140-1143) */
static void mutual_distances(oracle_t* o, int b) {
  int N = o->N;
  float diag = sqrtf(o->cfg.world_x_dim * o->cfg.world_x_dim + o->cfg.world_y_dim * o->cfg.world_y_dim);
  float* D = o->dist_agents + (size_t)b * N * N;
  for (int i = 0; i < N; ++i) {
    for (int j = 0; j < N; ++j) {
      float d;
      if (i == j) d = diag;
      else if (o->cfg.distance_type == SIGMAENV_DIST_C2C) {     /* helper_scenario.py:1012-1029 */
        const float* si = o->state + ((size_t)b * N + i) * 8;
        const float* sj = o->state + ((size_t)b * N + j) * 8;
        float dx = si[0] - sj[0], dy = si[1] - sj[1];
        d = sqrtf(dx * dx + dy * dy);
      } else {
        int a = i < j ? i : j, c = i < j ? j : i;               /* computed once for i<j, mirrored :1135-1138 */
        d = mtv_pair(o->vertices + ((size_t)b * N + a) * 10, o->vertices + ((size_t)b * N + c) * 10);
      }
      D[i * N + j] = d;
    }
  }
}

/* update_distances for one agent, world_state_rt.py:582-656 (corner queries use whatever is in o->vertices) */
static void agent_distances(oracle_t* o, int b, int i) {
  int N = o->N, P = o->P;
  size_t bi = (size_t)b * N + i;
  int path = o->path[bi * 4];
  const float* s = o->state + bi * 8;
  const float* ctr = o->center + (size_t)path * P * 2;
  const float* lb = o->left + (size_t)path * P * 2;
  const float* rb = o->right + (size_t)path * P * 2;
  int nl = o->n_left[path], nr = o->n_right[path];
  float wh = (float)((double)o->cfg.width / 2.0);
  float d;
  int32_t id;
  point_polyline(s[0], s[1], ctr, o->n_center[path], &o->dist_ref[bi], &o->closest[bi * 3 + 0]); /* :587-596 */
  point_polyline(s[0], s[1], lb, nl, &d, &o->closest[bi * 3 + 1]);                                 /* :598-610 */
  o->dist_left[bi * 5] = d - wh;
  point_polyline(s[0], s[1], rb, nr, &d, &o->closest[bi * 3 + 2]);                                 /* :612-624 */
  o->dist_right[bi * 5] = d - wh;
  const float* v = o->vertices + bi * 10;
  for (int c = 0; c < 4; ++c) {                                                                    /* :626-646 */
    point_polyline(v[2 * c], v[2 * c + 1], lb, nl, &o->dist_left[bi * 5 + c + 1], &id);
    point_polyline(v[2 * c], v[2 * c + 1], rb, nr, &o->dist_right[bi * 5 + c + 1], &id);
  }
  float m = INFINITY;                                                                              /* :648-656 */
  for (int c = 0; c < 5; ++c) m = fminf(m, o->dist_left[bi * 5 + c]);
  for (int c = 0; c < 5; ++c) m = fminf(m, o->dist_right[bi * 5 + c]);
  o->dist_bound[bi] = m;
}

static void agent_short_term(oracle_t* o, int b, int i) { /* update_ref_paths_agent_related, world_state_rt.py:668-684 */
  size_t bi = (size_t)b * o->N + i;
  int path = o->path[bi * 4];
  short_term_path(o->center + (size_t)path * o->P * 2, o->n_center[path], o->is_loop[path], o->closest[bi * 3], o->short_term + bi * NS * 2);
}

/* update_collisions, world_state_rt_sim.py:379-424 */
static void update_collisions(oracle_t* o, int b) {
  int N = o->N, P = o->P;
  uint8_t* CA = o->col_agents + (size_t)b * N * N;
  for (int i = 0; i < N; ++i) {
    size_t bi = (size_t)b * N + i;
    const float* vi = o->vertices + bi * 10;
    if (o->cfg.distance_type == SIGMAENV_DIST_C2C) {            /* :382-393 */
      for (int j = i + 1; j < N; ++j) {
        if (interx(vi, 5, o->vertices + ((size_t)b * N + j) * 10, 5)) { CA[i * N + j] = 1; CA[j * N + i] = 1; }
      }
    } else {                                                     /* :394-396 two agents collide iff their mtv distance == 0 */
      for (int j = 0; j < N; ++j) CA[i * N + j] = (o->dist_agents[(size_t)b * N * N + i * N + j] == 0.0f);
    }
    int path = o->path[bi * 4];
    const float* lb = o->left + (size_t)path * P * 2;
    const float* rb = o->right + (size_t)path * P * 2;
    int nl = o->n_left[path], nr = o->n_right[path];
    /* padded tail segments are degenerate (dx2=dy2=S2=0 -> C2 false), so only the real points matter :399-411 */
    if (interx(vi, 5, lb, nl) | interx(vi, 5, rb, nr)) o->col_flags[bi * 4 + 0] = 1;
    if (!o->is_loop[path]) {                                     /* :413-424 */
      float entry[4] = {lb[0], lb[1], rb[0], rb[1]};             /* world_state_rt.py:394-399 */
      float exit_[4] = {lb[2 * (nl - 1)], lb[2 * (nl - 1) + 1], rb[2 * (nr - 1)], rb[2 * (nr - 1) + 1]}; /* :401-406 */
      o->col_flags[bi * 4 + 1] = (uint8_t)interx(vi, 5, entry, 2);
      o->col_flags[bi * 4 + 2] = (uint8_t)interx(vi, 5, exit_, 2);
    }
  }
}

/* _apply_ttc_near_agent_penalty, road_traffic.py:1255-1332 */
static float ttc_penalty(const oracle_t* o, int b, int i) {
  int N = o->N;
  const sigmaenv_config_t* c = &o->cfg;
  const float eps = 1e-6f;
  double d_safe = (double)c->threshold_near_other_agents_low;   /* :1281 */
  float d_safe_sq = (float)(d_safe * d_safe);
  float d_safe32 = c->threshold_near_other_agents_low, d_gate = c->threshold_near_other_agents_high;
  const float* si = o->state + ((size_t)b * N + i) * 8;
  float risk_sum = 0.0f;
  for (int j = 0; j < N; ++j) {
    const float* sj = o->state + ((size_t)b * N + j) * 8;
    float px = sj[0] - si[0], py = sj[1] - si[1];               /* :1275 */
    float vx = sj[5] - si[5], vy = sj[6] - si[6];               /* :1276 */
    float a = vx * vx + vy * vy;                                 /* :1288 */
    float bq = 2.0f * (px * vx + py * vy);                       /* :1289 */
    float pp = px * px + py * py;
    float cq = pp - d_safe_sq;                                   /* :1290 */
    float disc = bq * bq - 4.0f * a * cq;                        /* :1292 */
    float sq = sqrtf(fmaxf(disc, 0.0f));                         /* :1293-1294 */
    float dist = sqrtf(fmaxf(pp, 0.0f));                         /* :1297 */
    int valid = (a > eps) && (disc > 0.0f) && (bq < 0.0f);       /* :1303 */
    float cand = (-bq - sq) / (2.0f * a + eps);                  /* :1307 */
    float ttc = INFINITY;
    if (valid && cand > 0.0f) ttc = cand;                        /* :1309 */
    if (dist <= d_safe32) ttc = 0.0f;                            /* :1312 */
    if (j == i) ttc = INFINITY;                                  /* :1315 */
    if (!(dist <= d_gate)) ttc = INFINITY;                       /* :1318 */
    float x = fminf(ttc, c->ttc_high);                           /* :1322 */
    risk_sum += decreasing_lin(x, c->ttc_low, c->ttc_high);      /* :1323-1328 */
  }
  float risk = risk_sum / (float)(N - 1 > 1 ? N - 1 : 1);
  return risk * c->penalty_near_other_agents;                    /* :1330 */
}

/* weighting_ref_directions = linspace(1, 0.2, n_points_short_term) / sum, road_traffic.py:536-543 (bit patterns read from the reference) */
#include "../include/sigmaenv_ref_weights.h"
static const uint32_t W_REF_BITS[NS] = SIGMAENV_W_REF_BITS;

/* ScenarioRoadTraffic.reward for one agent, road_traffic.py:925-1253 (state updates excluded) */
static float agent_reward(oracle_t* o, int b, int i, float* near_other_out, int* has_near, float* goal_out, float* pca_out, float* pcl_out) {
  int N = o->N;
  const sigmaenv_config_t* c = &o->cfg;
  size_t bi = (size_t)b * N + i;
  const float* s = o->state + bi * 8;
  const float* pp = o->prev_pos + bi * 2;
  const float* st = o->short_term + bi * NS * 2;
  float w[NS];
  memcpy(w, W_REF_BITS, sizeof(w));
  float mvx = s[0] - pp[0], mvy = s[1] - pp[1];                 /* :972-974 */
  float acc = 0.0f;
  for (int k = 0; k < NS; ++k) {
    float rx = st[2 * k] - pp[0], ry = st[2 * k + 1] - pp[1];   /* :976-980 */
    float mp = mvx * rx + mvy * ry;                              /* :981 */
    acc = acc + mp * w[k];                                       /* :982-984 (gemv; summation order not pinned) */
  }
  float denom = (float)((double)c->max_speed * (double)c->dt);
  float rew = 0.0f;
  rew += acc / denom * c->reward_progress;                       /* :986-991 */
  int goal = o->col_flags[bi * 4 + 2];                           /* :996 */
  float reward_goal = (float)goal * c->reward_reach_goal;        /* :997 */
  int col_a = 0;
  for (int j = 0; j < N; ++j) col_a |= o->col_agents[(size_t)b * N * N + i * N + j]; /* :1008-1013 */
  float pca = (float)col_a * c->penalty_collide_with_agents;
  int col_l = o->col_flags[bi * 4 + 0];
  float pcl = (float)col_l * c->penalty_collide_with_boundaries; /* :1021-1026 */
  o->timer[b * 4 + 2] += goal;                                   /* task_success_times :998-1002 */
  if (col_a | col_l | goal) o->timer[b * 4 + 1] += 1;            /* num_task_tries :1030-1035 */
  float pen_lane = decreasing_lin(o->dist_bound[bi], c->threshold_near_boundary_low, c->threshold_near_boundary_high) * c->penalty_near_boundary; /* :1040-1048 */
  *has_near = 0;
  if (c->is_testing_mode) {                                      /* :1050-1055 */
    rew += reward_goal; rew += pca; rew += pcl;
  } else {
    if (c->rew_flags & SIGMAENV_REW_EXACT_SPARSE) { rew += pca; rew += pcl; }     /* :1058-1062 */
    if (c->rew_flags & SIGMAENV_REW_TTC) {                                           /* :1064-1084 */
      float p = ttc_penalty(o, b, i);
      *near_other_out = p; *has_near = 1;
      rew += p; rew += pen_lane; rew += pca; rew += pcl;
      if (c->rew_flags & SIGMAENV_REW_HAS_SPARSE) { rew += pca; rew += pcl; }
    }
    if (c->rew_flags & SIGMAENV_REW_DISTANCE) {                                      /* :1086-1110 */
      float ssum = 0.0f;
      for (int j = 0; j < N; ++j)
        ssum += decreasing_lin(o->dist_agents[(size_t)b * N * N + i * N + j], c->threshold_near_other_agents_low, c->threshold_near_other_agents_high);
      float p = ssum * c->penalty_near_other_agents;
      *near_other_out = p; *has_near = 1;
      rew += p; rew += pen_lane;
      if (c->rew_flags & SIGMAENV_REW_HAS_SPARSE) { rew += pca; rew += pcl; }
    }
    if (c->rew_flags & SIGMAENV_REW_CBF_QP) {                                        /* :1112-1139, is_solve_qp == True */
      const float* nomv = o->cbf_nominal + bi * 2;                   /* world_state.nominal_action_{vel,steer}, left by the CBF-QP */
      const float* cur = o->action + bi * 2;                         /* agent.action.u after WorldCustom.step's clamp */
      float pv = c->penalty_deviate_from_cbf_vel * (fabsf(cur[0] - nomv[0]) / c->max_speed);        /* :1127-1129 */
      float ps = c->penalty_deviate_from_cbf_steer * (fabsf(cur[1] - nomv[1]) / c->max_steering);   /* :1130-1133 */
      rew += pv + ps;
      if (c->rew_flags & SIGMAENV_REW_HAS_SPARSE) { rew += pca; rew += pcl; }
    }
    if (c->rew_flags & SIGMAENV_REW_CBF) {                                           /* :1112-1151, is_solve_qp == False */
      size_t BNn = (size_t)o->B * N;
      const float* RI = o->reward_info;                              /* written by CBFQP.update_qp before the step */
      float cbf_rew = ((RI[5 * BNn + bi] + RI[6 * BNn + bi]) + RI[4 * BNn + bi]) / 3.0f;   /* :1141-1146 */
      rew += cbf_rew;
      if (c->rew_flags & SIGMAENV_REW_HAS_SPARSE) { rew += pca; rew += pcl; }
    }
  }
  *goal_out = reward_goal; *pca_out = pca; *pcl_out = pcl;
  return clampf(rew, -1.0f, 1.0f);                               /* :1249 */
}

/* ego-view transform, helper_scenario.py:1241-1273 */
static inline void ego_transform(float pix, float piy, float rot_i, float pjx, float pjy, float* ox, float* oy) {
  float dx = pjx - pix, dy = pjy - piy;
  float ab = norm2(dx, dy);
  float rr = cr_atan2(dy, dx) - rot_i;
  *ox = cr_cos(rr) * ab;
  *oy = cr_sin(rr) * ab;
}

/* observation of one agent (ego view, partial observation): observation_provider_rt.py:345-588 (latest slot only), :594-925.  obs_flags = 0 is
 * the default layout; the SIGMAENV_OBS_* switches add / drop / replace columns exactly where _observe_self / _observe_other_agents do. */
/* MapManager.determine_current_lanelet (map_manager.py:41-92): squared distances of the position to every centre-line point of every lanelet (the
 * stacked tensor is zero-padded, the padding counts), minimum per lanelet, argmin over the lanelets (first index on ties) */
static int current_lanelet(const oracle_t* o, float px, float py) {
  int best = 0;
  float bd = INFINITY;
  for (int l = 0; l < o->n_lanelets; ++l) {
    float md = INFINITY;
    const float* c = o->lanelet_centers + (size_t)l * o->lanelet_pts * 2;
    for (int q = 0; q < o->lanelet_pts; ++q) {
      float dx = px - c[2 * q], dy = py - c[2 * q + 1];
      float d = dx * dx + dy * dy;                               /* torch.sum((a - c) ** 2, dim=4) */
      if (d < md) md = d;
    }
    if (md < bd) { bd = md; best = l; }
  }
  return best;
}

int sigmaenv_oracle_set_lanelets(oracle_t* o, int32_t n_lanelets, int32_t max_points, const float* centers, const uint64_t* neighbors) {
  if (!o || n_lanelets < 1 || n_lanelets > 64 || max_points < 1 || !centers || !neighbors) return SIGMAENV_EINVAL;
  free(o->lanelet_centers); free(o->lanelet_neigh);
  o->lanelet_centers = (float*)malloc((size_t)n_lanelets * max_points * 2 * sizeof(float));
  o->lanelet_neigh = (uint64_t*)malloc((size_t)n_lanelets * sizeof(uint64_t));
  if (!o->lanelet_centers || !o->lanelet_neigh) return SIGMAENV_ENOMEM;
  memcpy(o->lanelet_centers, centers, (size_t)n_lanelets * max_points * 2 * sizeof(float));
  memcpy(o->lanelet_neigh, neighbors, (size_t)n_lanelets * sizeof(uint64_t));
  o->n_lanelets = n_lanelets;
  o->lanelet_pts = max_points;
  return SIGMAENV_OK;
}

/* opponent_modeling (helper_training.py:1117-1137): the tentative action of the k-th observed neighbour into the k-th placeholder pair at the row's end */
int sigmaenv_oracle_opponent_fill(oracle_t* o, const float* actions) {
  if (!o || !actions || !(o->cfg.obs_flags & SIGMAENV_OBS_OPPONENT_PAD)) return SIGMAENV_EINVAL;
  const int N = o->N, K = o->K, D = o->D;
  for (size_t b = 0; b < (size_t)o->B; ++b)
    for (int ego = 0; ego < N; ++ego)
      for (int j = 0; j < K; ++j) {
        const int sur = o->nearing[(b * N + ego) * K + j];
        float* dst = o->obs + (b * N + ego) * D + (D - (K - j) * 2);
        dst[0] = actions[(b * N + sur) * 2];
        dst[1] = actions[(b * N + sur) * 2 + 1];
      }
  return SIGMAENV_OK;
}

static void agent_observation(oracle_t* o, int b, int i) {
  int N = o->N, K = o->K, D = o->D;
  const sigmaenv_config_t* c = &o->cfg;
  const int F = c->obs_flags;
  size_t bi = (size_t)b * N + i;
  const float* si = o->state + bi * 8;
  float* ob = o->obs + bi * D;
  float n_pos = (float)((double)c->length * 10.0);               /* normalizers.pos, road_traffic.py:588-592 */
  float n_v = c->max_speed;                                      /* :596 */
  float n_dl = (float)((double)c->lane_width * 3.0);             /* :599-601 */
  float n_rot = (float)(2.0 * 3.141592653589793);                /* normalizers.rot = 2 * torch.pi, :597 */
  float n_da = (float)((double)c->length * 10.0);                /* normalizers.distance_agent, :605-607 */
  const float* Drow = o->dist_agents + (size_t)b * N * N + (size_t)i * N;
  /* top-k smallest, ascending, lowest index on ties: observation_provider_rt.py:629-636 */
  int32_t* near = o->nearing + bi * K;
  uint64_t taken = 0;
  const int full = (F & SIGMAENV_OBS_FULL) != 0;                 /* is_partial_observation == False: no top-k, nearing_agents_indices keeps its zeros (:627-636) */
  for (int k = 0; full && k < K; ++k) near[k] = 0;
  for (int k = 0; !full && k < K; ++k) {
    int bj = -1;
    float bd = INFINITY;
    for (int j = 0; j < N; ++j) {
      if ((taken >> j) & 1) continue;
      if (bj < 0 || Drow[j] < bd) { bd = Drow[j]; bj = j; }
    }
    taken |= 1ull << bj;
    near[k] = bj;
  }
  int p = 0;
  const int bird = (F & SIGMAENV_OBS_BIRD_VIEW) != 0;            /* is_ego_view == False: world frame, normalizers.pos_world (:537-575) */
  /* mask by lanelet relation (:638-665, map_manager.py:94-118): the agents' lanelets exist in bird view only (:577-588), the neighbour table on
   * OSM maps only; an ego lanelet beyond the table (interchange_3 lists 20 of its 22 lanelets) raises IndexError in the reference, masks here */
  const int lane_mask = bird && c->is_apply_mask && o->n_lanelets > 0;
  const uint64_t ego_neigh = lane_mask ? o->lanelet_neigh[current_lanelet(o, si[0], si[1])] : ~0ull;
  const float nwx = c->world_x_dim, nwy = c->world_y_dim;
#define OBS_POINT(tx, ty, ox, oy) do { if (bird) { ox = (tx) / nwx; oy = (ty) / nwy; } else { float ex_, ey_; ego_transform(si[0], si[1], si[2], (tx), (ty), &ex_, &ey_); ox = ex_ / n_pos; oy = ey_ / n_pos; } } while (0)
  if (bird) {                                                    /* [own] position and rotation (:862-877) */
    ob[p++] = si[0] / nwx; ob[p++] = si[1] / nwy;
    ob[p++] = angle_eliminate_two_pi(si[2]) / n_rot;
    ob[p++] = si[5] / n_v; ob[p++] = si[6] / n_v;                /* [own] velocity, both components (:547-549, :878-880) */
  } else {
    /* [own] longitudinal speed: past_vel[b,i,i,0] = ||v_i|| * cos(wrap(psi_i - psi_i)) / v  (:441-449, :885-887) */
    float rr = angle_eliminate_two_pi(si[2] - si[2]);
    ob[p++] = (norm2(si[5], si[6]) * cr_cos(rr)) / n_v;
  }
  if (F & SIGMAENV_OBS_STEERING) ob[p++] = angle_eliminate_two_pi(si[4]) / n_rot;   /* [own] steering, :356-360, :392, :888-892 */
  /* [own] short-term reference path in the ego frame (:451-460, :893-897) */
  for (int k = 0; k < NS; ++k) {
    float ox, oy;
    OBS_POINT(o->short_term[bi * NS * 2 + 2 * k], o->short_term[bi * NS * 2 + 2 * k + 1], ox, oy);
    ob[p++] = ox;
    ob[p++] = oy;
  }
  /* [own] distances, all normalised by distance_lanelet (:373-389, :898-922) */
  if (!(F & SIGMAENV_OBS_NO_DIST_CENTER)) ob[p++] = o->dist_ref[bi] / n_dl;
  if (F & SIGMAENV_OBS_BOUNDARY_POINTS) {
    /* [own] the 5 points of each boundary around its closest point instead of the distances (:905-922; world_state_rt.py:686-724:
     * get_short_term_reference_path with sample_interval 1, n_points_shift -2 on the PADDED boundary polyline, the loop rule with the
     * centre line's point count; a negative index counts from the end of the padded tensor) */
    int path = o->path[bi * 4];
    int n = o->n_center[path], loop = o->is_loop[path] != 0;
    for (int side = 0; side < 2; ++side) {
      const float* poly = (side ? o->right : o->left) + (size_t)path * o->P * 2;
      int cp = o->closest[bi * 3 + 1 + side];
      const int shift = o->fresh[bi] ? 1 : -2;                   /* n_points_shift: 1 at a reset (:531-576), -2 in update_distances (:686-724) */
      for (int k = 0; k < 5; ++k) {
        int id = k + cp + shift;
        if (loop && id >= n - 1) id = (id + 1) % n;
        if (id < 0) id += o->P;
        float ox, oy;
        OBS_POINT(poly[2 * id], poly[2 * id + 1], ox, oy);
        ob[p++] = ox;
        ob[p++] = oy;
      }
    }
  } else {
    float ml = INFINITY, mr = INFINITY;
    for (int q = 0; q < 5; ++q) { ml = fminf(ml, o->dist_left[bi * 5 + q]); mr = fminf(mr, o->dist_right[bi * 5 + q]); }
    ob[p++] = ml / n_dl;
    ob[p++] = mr / n_dl;
  }
  /* [others] per observed neighbour (:803-853): vertices (8) -- or position (2), relative rotation, length, width --, velocity (2), steering,
   * distance, its short-term reference path; masked by distance (:638-749): positions / vertices / reference path / distance := 1,
   * rotation / steering / velocity := 0 (lengths and widths are not masked) */
  if (full) {
    /* full observation (bird view; the ego view raises in the reference: :756-800 with indexing_tuple_2 = (env_idx,)): every feature tensor holds ALL N
     * agents in index order -- vertices [B,N,4,2], velocities [B,N,2], ..., and the mutual distances the whole [B,N,N] matrix, which
     * `obs_distance_other_agents[indexing_tuple_2] = 0` (:776-778) zeroes entirely in this view --, is reshaped to [B, n_nearing_agents, -1] (:790-816),
     * i.e. cut into K = n_nearing equal chunks of its flat per-env array, and the chunks of the features are concatenated chunk by chunk (:819-851).
     * No mask applies (:638-749 is the partial branch).  sigmaenv_obs_dim_full refuses the shapes torch.reshape refuses (N * width not divisible by K). */
    enum { F_VERT, F_POS, F_ROT, F_LEN, F_WID, F_VEL, F_STEER, F_DIST, F_REF };
    int kind[9], wid[9], nf = 0;
    if (!(F & SIGMAENV_OBS_NO_VERTICES)) { kind[nf] = F_VERT; wid[nf++] = 8; }
    else { kind[nf] = F_POS; wid[nf++] = 2; kind[nf] = F_ROT; wid[nf++] = 1; kind[nf] = F_LEN; wid[nf++] = 1; kind[nf] = F_WID; wid[nf++] = 1; }
    kind[nf] = F_VEL; wid[nf++] = 2;
    if (F & SIGMAENV_OBS_STEERING) { kind[nf] = F_STEER; wid[nf++] = 1; }
    if (!(F & SIGMAENV_OBS_NO_DIST_AGENTS)) { kind[nf] = F_DIST; wid[nf++] = N; }
    if (F & SIGMAENV_OBS_REF_OTHERS) { kind[nf] = F_REF; wid[nf++] = 2 * NS; }
    for (int ck = 0; ck < K; ++ck) {
      for (int f = 0; f < nf; ++f) {
        const int w = wid[f], chunk = N * w / K;
        for (int e = 0; e < chunk; ++e) {
          const int flat = ck * chunk + e, j = flat / w, q = flat - j * w;
          const size_t bj = (size_t)b * N + j;
          const float* sj = o->state + bj * 8;
          float v = 0.0f;
          switch (kind[f]) {
            case F_VERT: v = o->vertices[bj * 10 + q] / ((q & 1) ? nwy : nwx); break;              /* :555-563 */
            case F_POS: v = sj[q] / (q ? nwy : nwx); break;                                          /* :539-546 */
            case F_ROT: v = angle_eliminate_two_pi(sj[2]) / n_rot; break;                            /* :550-553 */
            case F_LEN: v = c->length / n_da; break;                                                 /* :387-389 */
            case F_WID: v = c->width / n_da; break;                                                  /* :390-391 */
            case F_VEL: v = sj[5 + q] / n_v; break;                                                  /* :547-549 */
            case F_STEER: v = angle_eliminate_two_pi(sj[4]) / n_rot; break;                          /* :356-360, :392 */
            case F_DIST: v = 0.0f; break;                                                            /* :776-778 */
            case F_REF: v = o->short_term[bj * NS * 2 + q] / ((q & 1) ? nwy : nwx); break;          /* :564-571 */
          }
          ob[p++] = v;
        }
      }
    }
  }
  for (int k = 0; !full && k < K; ++k) {
    int j = near[k];
    size_t bj = (size_t)b * N + j;
    const float* sj = o->state + bj * 8;
    const float* vj = o->vertices + bj * 10;
    int masked = c->is_apply_mask && Drow[j] >= c->distance_mask_agents;
    if (lane_mask) masked = masked || !((ego_neigh >> current_lanelet(o, sj[0], sj[1])) & 1ull);
    if (!(F & SIGMAENV_OBS_NO_VERTICES)) {
      for (int q = 0; q < 4; ++q) {
        float ox, oy;
        OBS_POINT(vj[2 * q], vj[2 * q + 1], ox, oy);
        ob[p++] = masked ? 1.0f : ox;                            /* :734-737 */
        ob[p++] = masked ? 1.0f : oy;
      }
    } else {
      float ox, oy;
      OBS_POINT(sj[0], sj[1], ox, oy);                           /* :429-434 */
      ob[p++] = masked ? 1.0f : ox;                              /* :672-680 */
      ob[p++] = masked ? 1.0f : oy;
      ob[p++] = masked ? 0.0f : (bird ? angle_eliminate_two_pi(sj[2]) : angle_eliminate_two_pi(sj[2] - si[2])) / n_rot;   /* :437, :551-553, :683-688 */
      ob[p++] = c->length / n_da;                                /* :387-389, :691-694 */
      ob[p++] = c->width / n_da;                                 /* :390-391, :695-698 */
    }
    if (bird) {
      ob[p++] = masked ? 0.0f : sj[5] / n_v;                     /* :547-549 */
      ob[p++] = masked ? 0.0f : sj[6] / n_v;
    } else {
      float rr = angle_eliminate_two_pi(sj[2] - si[2]);         /* :439 */
      float va = norm2(sj[5], sj[6]);                            /* :444 */
      ob[p++] = masked ? 0.0f : (va * cr_cos(rr)) / n_v;         /* :717-719 */
      ob[p++] = masked ? 0.0f : (va * cr_sin(rr)) / n_v;
    }
    if (F & SIGMAENV_OBS_STEERING) ob[p++] = masked ? 0.0f : angle_eliminate_two_pi(sj[4]) / n_rot;   /* :699-706 */
    if (!(F & SIGMAENV_OBS_NO_DIST_AGENTS)) ob[p++] = masked ? 1.0f : Drow[j] / n_dl;   /* :373-375, :747-749 */
    if (F & SIGMAENV_OBS_REF_OTHERS) {                           /* :451-460, :721-729 */
      for (int q = 0; q < NS; ++q) {
        float ox, oy;
        OBS_POINT(o->short_term[bj * NS * 2 + 2 * q], o->short_term[bj * NS * 2 + 2 * q + 1], ox, oy);
        ob[p++] = masked ? 1.0f : ox;
        ob[p++] = masked ? 1.0f : oy;
      }
    }
  }
#undef OBS_POINT
  if (F & SIGMAENV_OBS_OPPONENT_PAD) {                             /* F.pad(obs, (0, n_nearing * n_actions)), :606-611: before the noise */
    for (int k = 0; k < 2 * K; ++k) ob[p++] = 0.0f;
  }
  /* sensor noise, observation_provider_rt.py:613-618: obs + obs_noise_level * rand_like(obs), uniform in [0, level).  The draw is the shared
   * specification of sigmaenv_config_t.obs_noise_level: the counter-based generator keyed on the env's own counters (episodes_reset, timer.step) */
  if (c->obs_noise_level > 0.0f) {
    const uint64_t seed = ((uint64_t)c->obs_noise_seed_hi << 32) | c->obs_noise_seed_lo;
    /* + the salt of an observation that is TAKEN AGAIN at the same counters (sigmaenv_oracle_observe: its n-th call on the handle; 0 inside steps and resets):
     * the reference draws rand_like on every observation() call */
    const uint64_t counter = (uint64_t)(uint32_t)o->timer[b * 4 + 3] * 65537ull + (uint64_t)(uint32_t)o->timer[b * 4] + (uint64_t)(uint32_t)(o->obs_salt * 0x632BE5ABu);
    for (int k = 0; k < p; ++k) {
      const float u = (float)(rng_u32(seed, counter, (uint32_t)(c->env_index_base + b), (uint32_t)i, 9000u + (uint32_t)k) >> 8) * (1.0f / 16777216.0f);
      ob[k] = ob[k] + c->obs_noise_level * u;
    }
  }
}

/* done(), road_traffic.py:1368-1487 (flags only; the resets it triggers are requests to the host) */
static void env_done(oracle_t* o, int b) {
  int N = o->N;
  const sigmaenv_config_t* c = &o->cfg;
  int col_a = 0, col_l = 0;
  for (int k = 0; k < N * N; ++k) col_a |= o->col_agents[(size_t)b * N * N + k];
  for (int i = 0; i < N; ++i) col_l |= o->col_flags[((size_t)b * N + i) * 4];
  int max_reached = o->timer[b * 4] == (c->max_steps - 1);      /* :1413 */
  int fixed = 0;                                                 /* :1388-1397: t = timer.step * dt is an fp32 tensor */
  if (c->reset_agent_fixed_duration > 0.0f) {
    float tt = (float)o->timer[b * 4] * c->dt;
    fixed = (remainder_pos(tt, c->reset_agent_fixed_duration) == 0.0f) && (tt != 0.0f);
  }
  int done;
  if (c->is_testing_mode) done = max_reached | fixed;            /* :1429-1433 */
  else done = max_reached | col_a | col_l | fixed;               /* :1450-1455 */
  o->done[b] = (uint8_t)done;
  for (int i = 0; i < N; ++i) {
    size_t bi = (size_t)b * N + i;
    int rq = 0;
    if (c->is_testing_mode) {                                    /* :1436-1447 */
      int ca = 0;
      for (int j = 0; j < N; ++j) ca |= o->col_agents[(size_t)b * N * N + i * N + j];
      rq = ca | o->col_flags[bi * 4] | o->col_flags[bi * 4 + 1] | o->col_flags[bi * 4 + 2];
    } else if (c->has_entry_exit) {                              /* :1456-1473 */
      rq = o->col_flags[bi * 4 + 1] | o->col_flags[bi * 4 + 2];
    }
    o->col_flags[bi * 4 + 3] = (uint8_t)(rq && !done);
  }
}

/* one env, the VMAS >= 1.4 call order: world.step; reward(a) for all a; observation(a) for all a; done() */
static void step_env(oracle_t* o, int b, const float* actions) {
  int N = o->N;
  for (int i = 0; i < N; ++i) {
    size_t bi = (size_t)b * N + i;
    o->fresh[bi] = 0;
    bicycle_step(&o->cfg, o->state + bi * 8, actions + bi * 2, o->action + bi * 2);
  }
  float* RI = o->reward_info;
  size_t BN = (size_t)o->B * N;
  for (int i = 0; i < N; ++i) {
    size_t bi = (size_t)b * N + i;
    if (i == 0) {
      o->timer[b * 4] += 1;                                      /* road_traffic.py:954-962 */
      mutual_distances(o, b);                                    /* world_state_rt.py:583-584: BEFORE update_vertices -> mtv sees last step's vertices */
    }
    agent_distances(o, b, i);                                    /* agent 0 still sees last step's vertices for its corners */
    if (i == 0) {                                                /* world_state_rt_sim.py:442-448 */
      memset(o->col_agents + (size_t)b * N * N, 0, (size_t)N * N);
      for (int a = 0; a < N; ++a) { uint8_t* f = o->col_flags + ((size_t)b * N + a) * 4; f[0] = f[1] = f[2] = 0; }
      for (int a = 0; a < N; ++a) {
        const float* s = o->state + ((size_t)b * N + a) * 8;
        rect_vertices(&o->cfg, s[0], s[1], s[2], o->vertices + ((size_t)b * N + a) * 10);
      }
      update_collisions(o, b);
    }
    float near_other = 0.0f, goal, pca, pcl;
    int has_near;
    float r = agent_reward(o, b, i, &near_other, &has_near, &goal, &pca, &pcl);
    o->reward[bi] = r;
    /* RewardInfo.reset() at the top of every reward() zeroes all fields of ALL agents except the three it skips
     * (helper_scenario.py:128-138), so after the loop only the last agent's entries of the other fields survive. */
    int last = (i == N - 1);
    RI[1 * BN + bi] = last ? goal : 0.0f;
    RI[8 * BN + bi] = last ? pcl : 0.0f;
    RI[7 * BN + bi] = last ? pca : 0.0f;
    RI[11 * BN + bi] = last ? r : 0.0f;
    if (has_near) RI[4 * BN + bi] = near_other;
    agent_short_term(o, b, i);                                   /* update_state_after_rewarding, world_state_rt_sim.py:450-454 */
  }
  for (int i = 0; i < N; ++i) {                                  /* state_buffer.add at the last agent, road_traffic.py:1226-1240 */
    size_t bi = (size_t)b * N + i;
    o->prev_pos[bi * 2] = o->state[bi * 8];
    o->prev_pos[bi * 2 + 1] = o->state[bi * 8 + 1];
  }
  for (int i = 0; i < N; ++i) agent_observation(o, b, i);
  env_done(o, b);
}

/* ---- reset ------------------------------------------------------------------------------------------------------ */
/* reset_init_distances_and_short_term_ref_path for one agent, world_state_rt.py:422-529: vertices first, then corners */
static void reset_agent_derived(oracle_t* o, int b, int i) {
  size_t bi = (size_t)b * o->N + i;
  o->fresh[bi] = 1;
  const float* s = o->state + bi * 8;
  rect_vertices(&o->cfg, s[0], s[1], s[2], o->vertices + bi * 10);
  agent_distances(o, b, i);
  agent_short_term(o, b, i);
}
static void reset_env_tail(oracle_t* o, int b, int full_env) { /* road_traffic.py:902-923 */
  int N = o->N;
  mutual_distances(o, b);
  memset(o->col_agents + (size_t)b * N * N, 0, (size_t)N * N);
  for (int a = 0; a < N; ++a) {
    size_t ba = (size_t)b * N + a;
    uint8_t* f = o->col_flags + ba * 4;
    f[0] = f[1] = f[2] = f[3] = 0;
    o->prev_pos[ba * 2] = o->state[ba * 8];
    o->prev_pos[ba * 2 + 1] = o->state[ba * 8 + 1];
    if (full_env) { o->action[ba * 2] = 0.0f; o->action[ba * 2 + 1] = 0.0f; }
  }
  if (full_env) { o->timer[b * 4] = 0; o->timer[b * 4 + 3] += 1; o->done[b] = 0; }
}

/* TEST REPLAYS ONLY.  road_traffic.py:889-907: `env_j = slice(None); if env_index: env_j = env_index` -- for env 0 the condition is false, so a
 * reset of env 0 (whole env or single agent) also recomputes, for EVERY env, the derived state of the reset agent(s) with the reset-time rules
 * (corner queries on the current vertices, boundary points with the reset shift), the mutual distances, and clears every env's collision flags.
 * Nothing the learner sees depends on it (the next step recomputes all of it from the states, and observations of untouched envs are not taken
 * again), so the product does not reproduce it; the replay of reference trajectories calls this after an event in env 0 so that the snapshots the
 * golden generator takes right after the resets compare exactly.  agent < 0: all agents. */
int sigmaenv_oracle_env0_reset_side_effect(oracle_t* o, int32_t agent) {
  if (!o || agent >= o->N) return SIGMAENV_EINVAL;
  for (int b = 1; b < o->B; ++b) {
    for (int i = 0; i < o->N; ++i) if (agent < 0 || i == agent) reset_agent_derived(o, b, i);
    mutual_distances(o, b);
    memset(o->col_agents + (size_t)b * o->N * o->N, 0, (size_t)o->N * o->N);
    for (int a = 0; a < o->N; ++a) { uint8_t* f = o->col_flags + ((size_t)b * o->N + a) * 4; f[0] = f[1] = f[2] = f[3] = 0; }
  }
  return SIGMAENV_OK;
}

/* counter-based RNG shared (as a specification) with the HIP kernel: 32-bit multiplicative mix + murmur3 finalisers over (seed, counter, env, agent, draw) */
static inline uint32_t rng_u32(uint64_t seed, uint64_t counter, uint32_t env, uint32_t agent, uint32_t draw) {
  uint32_t h = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B9u);
  h ^= ((uint32_t)counter + 0x7F4A7C15u) * 0x85EBCA6Bu;
  h ^= (env + 0x165667B1u) * 0xC2B2AE35u;
  h ^= (agent + 0x27D4EB2Fu) * 0x9E3779B1u;
  h ^= (draw + 0x61C88647u) * 0x85EBCA77u;
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;   /* murmur3 fmix32, twice */
  h += 0x9E3779B9u;
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
#define AUTO_RESET_MAX_TRIES 64
/* Exclusive upper end of the centre-line points try `t` (0-based) may draw from (world_state_rt_sim.py:253-263): the first half of the
 * path in training; in testing mode the range starts at the path's beginning and grows with the tries -- end_point_idx starts at 3 and
 * gains random_count (= t + 1) per try, i.e. 3 + (t + 1)(t + 2) / 2 -- capped by the half.  At least one point (3) is always allowed. */
static inline int reset_end_point(int testing, int t, int n) {
  int half = n / 2;
  int end = half;
  if (testing) {
    long grow = 3 + (long)(t + 1) * (t + 2) / 2;
    end = grow < half ? (int)grow : half;
  }
  return end < 4 ? 4 : end;
}

/* Device-style auto reset of one env: rejection sampling of world_state_rt_sim.py:215-311 (non-testing mode: point in
 * [3, n/2), min centre distance 1.5*sqrt(l^2+w^2)), bounded to AUTO_RESET_MAX_TRIES per agent, then :143-213 and the tail. */
static void auto_reset_env(oracle_t* o, int b, uint64_t seed, uint64_t counter, int path_first, int path_count) {
  /* cpm_mixed (world_state_rt_sim.py:313-358): the env draws its sub-scenario (torch.multinomial(cpm_scenario_probabilities) there; draw 5000 of
   * agent 0 against the cumulative distribution here) and takes that sub-scenario's path list */
  int scenario_id = 0;
  if (path_count == SIGMAENV_SCENARIO_LISTS) {
    const float us = (float)(rng_u32(seed, counter, (uint32_t)(o->cfg.env_index_base + b), 0u, 5000u) >> 8) * (1.0f / 16777216.0f);
    scenario_id = o->n_lists;
    for (int k = o->n_lists - 2; k >= 0; --k) if (us < o->list_cdf[k]) scenario_id = k + 1;
    path_first = o->list_first[scenario_id - 1]; path_count = o->list_count[scenario_id - 1];
  }
  int N = o->N;
  const sigmaenv_config_t* c = &o->cfg;
  float min_d = sqrtf((float)((double)c->length * (double)c->length + (double)c->width * (double)c->width)) * 1.5f; /* road_traffic.py:679-684 */
  float min_d_sq = min_d * min_d;
  for (int i = 0; i < N; ++i) {
    size_t bi = (size_t)b * N + i;
    float* s = o->state + bi * 8;
    int path = path_first, pt = 3;
    for (int t = 0; t < AUTO_RESET_MAX_TRIES; ++t) {
      path = path_first + (int)(((uint64_t)rng_u32(seed, counter, (uint32_t)(c->env_index_base + b), (uint32_t)i, 2u * t) * (uint64_t)(uint32_t)path_count) >> 32);
      int n = o->n_center[path];
      int end = reset_end_point(c->is_testing_mode, t, n);
      pt = 3 + (int)(((uint64_t)rng_u32(seed, counter, (uint32_t)(c->env_index_base + b), (uint32_t)i, 2u * t + 1u) * (uint64_t)(uint32_t)(end - 3)) >> 32);
      float px = o->center[((size_t)path * o->P + pt) * 2], py = o->center[((size_t)path * o->P + pt) * 2 + 1];
      s[0] = px; s[1] = py;
      int ok = 1;
      for (int j = 0; j < i; ++j) {
        const float* sj = o->state + ((size_t)b * N + j) * 8;
        float dx = px - sj[0], dy = py - sj[1];
        float d2 = dx * dx + dy * dy;
        if (!(d2 >= min_d_sq)) ok = 0;
      }
      if (ok) break;
    }
    float u = (float)(rng_u32(seed, counter, (uint32_t)(c->env_index_base + b), (uint32_t)i, 1000u) >> 8) * (1.0f / 16777216.0f);
    int ny = o->yaw_stride;
    int yi = pt < ny ? pt : ny - 1;
    float rot = o->yaw[(size_t)path * o->yaw_stride + yi];
    float speed = u * c->max_speed;                              /* world_state_rt_sim.py:195-198 */
    s[2] = rot; s[3] = speed; s[4] = 0.0f; s[7] = 0.0f;
    s[5] = speed * cr_cos(0.0f + rot);                           /* :199-204 */
    s[6] = speed * cr_sin(0.0f + rot);
    o->path[bi * 4 + 0] = path; o->path[bi * 4 + 1] = scenario_id; o->path[bi * 4 + 2] = path - path_first; o->path[bi * 4 + 3] = pt;
  }
  for (int i = 0; i < N; ++i) reset_agent_derived(o, b, i);
  reset_env_tail(o, b, 1);
  for (int i = 0; i < N; ++i) agent_observation(o, b, i);
}

/* ---- C-ABI twin --------------------------------------------------------------------------------------------------- */
int sigmaenv_oracle_n_short_term(void) { return NS; }
int sigmaenv_oracle_obs_dim(int32_t n_nearing) { return 1 + 2 * NS + 3 + n_nearing * 11; }
int sigmaenv_oracle_obs_dim_ex(int32_t n_nearing, int32_t f) {   /* observation_provider_rt.py:803-925 */
  int s = (f & SIGMAENV_OBS_STEERING) ? 1 : 0, r = (f & SIGMAENV_OBS_REF_OTHERS) ? 1 : 0;
  int own = 1 + s + 2 * NS + ((f & SIGMAENV_OBS_NO_DIST_CENTER) ? 0 : 1) + ((f & SIGMAENV_OBS_BOUNDARY_POINTS) ? 20 : 2) + ((f & SIGMAENV_OBS_BIRD_VIEW) ? 4 : 0);
  int other = ((f & SIGMAENV_OBS_NO_VERTICES) ? 5 : 8) + 2 + s + ((f & SIGMAENV_OBS_NO_DIST_AGENTS) ? 0 : 1) + r * 2 * NS;
  return own + n_nearing * other + ((f & SIGMAENV_OBS_OPPONENT_PAD) ? 2 * n_nearing : 0);
}

int sigmaenv_oracle_obs_dim_full(int32_t n_agents, int32_t n_nearing, int32_t f) {   /* :622-800, see agent_observation */
  if (!(f & SIGMAENV_OBS_FULL)) return sigmaenv_oracle_obs_dim_ex(n_nearing, f);
  if (!(f & SIGMAENV_OBS_BIRD_VIEW) || n_nearing < 1 || n_agents < 1) return SIGMAENV_EINVAL;
  const int N = n_agents, K = n_nearing;
  int wid[9], nf = 0;
  if (!(f & SIGMAENV_OBS_NO_VERTICES)) wid[nf++] = 8; else { wid[nf++] = 2; wid[nf++] = 1; wid[nf++] = 1; wid[nf++] = 1; }
  wid[nf++] = 2;
  if (f & SIGMAENV_OBS_STEERING) wid[nf++] = 1;
  if (!(f & SIGMAENV_OBS_NO_DIST_AGENTS)) wid[nf++] = N;
  if (f & SIGMAENV_OBS_REF_OTHERS) wid[nf++] = 2 * NS;
  if (N % K != 0) return SIGMAENV_EINVAL;  /* the reference reshapes all nine feature tensors to [B, n_nearing, -1], the width-1 ones (rotation, length, width, steering) included, whether
                                             * or not they end up in the row (observation_provider_rt.py:790-816) */
  int others = 0;
  for (int q = 0; q < nf; ++q) {
    if ((N * wid[q]) % K != 0) return SIGMAENV_EINVAL;          /* torch.reshape(B, n_nearing_agents, -1) raises */
    others += N * wid[q];
  }
  const int own = sigmaenv_oracle_obs_dim_ex(0, f & ~SIGMAENV_OBS_OPPONENT_PAD);
  return own + others + ((f & SIGMAENV_OBS_OPPONENT_PAD) ? 2 * K : 0);
}

static void* xcalloc(size_t n, size_t sz) { return calloc(n ? n : 1, sz); }

int sigmaenv_oracle_create(const sigmaenv_config_t* cfg, const sigmaenv_map_t* map, int device_id, void* stream, oracle_t** out) {
  (void)device_id; (void)stream;
  if (!cfg || !map || !out) return SIGMAENV_EINVAL;
  if (cfg->abi_version != SIGMAENV_ABI_VERSION) return SIGMAENV_EINVAL;
  if (cfg->n_points_short_term != 0 && cfg->n_points_short_term != NS) return SIGMAENV_EINVAL;
  if (cfg->n_envs < 1 || cfg->n_agents < 1 || cfg->n_agents > SIGMAENV_MAX_AGENTS) return SIGMAENV_EINVAL;
  if (cfg->n_nearing < 0 || cfg->n_nearing > SIGMAENV_MAX_NEARING || cfg->n_nearing > cfg->n_agents - 1) return SIGMAENV_EINVAL;
  if (cfg->distance_type != SIGMAENV_DIST_C2C && cfg->distance_type != SIGMAENV_DIST_MTV) return SIGMAENV_EINVAL;
  oracle_t* o = (oracle_t*)calloc(1, sizeof(oracle_t));
  if (!o) return SIGMAENV_ENOMEM;
  o->cfg = *cfg;
  int B = o->B = cfg->n_envs, N = o->N = cfg->n_agents, K = o->K = cfg->n_nearing;
  o->D = sigmaenv_oracle_obs_dim_full(N, K, cfg->obs_flags);
  if (o->D < 0) { free(o); return SIGMAENV_EINVAL; }
  int np = o->n_paths = map->n_paths, S = map->stride_points;
  int maxc = 0;
  for (int p = 0; p < np; ++p) {
    if (map->n_center[p] > maxc) maxc = map->n_center[p];
  }
  /* max_ref_path_points = max centre-line points + n_points_short_term*sample_interval + 2, road_traffic.py:520-530 */
  int P = maxc + NS * 2 + 2;
  for (int p = 0; p < np; ++p) {
    if (map->n_left[p] > P) P = map->n_left[p];
    if (map->n_right[p] > P) P = map->n_right[p];
  }
  o->P = P;
  o->yaw_stride = S;
  o->center = xcalloc((size_t)np * P * 2, 4); o->left = xcalloc((size_t)np * P * 2, 4); o->right = xcalloc((size_t)np * P * 2, 4);
  o->yaw = xcalloc((size_t)np * S, 4);
  o->n_center = xcalloc(np, 4); o->n_left = xcalloc(np, 4); o->n_right = xcalloc(np, 4); o->is_loop = xcalloc(np, 1);
  memcpy(o->yaw, map->yaw, (size_t)np * S * 4);
  for (int p = 0; p < np; ++p) {
    int n = map->n_center[p], nl = map->n_left[p], nr = map->n_right[p];
    o->n_center[p] = n; o->n_left[p] = nl; o->n_right[p] = nr; o->is_loop[p] = map->is_loop[p];
    const float* c = map->center + (size_t)p * S * 2;
    float* dc = o->center + (size_t)p * P * 2;
    memcpy(dc, c, (size_t)n * 8);
    /* _extend_map_related_ref_path, world_state_rt.py:279-293: centre[-1] + k * (centre[-1]-centre[-2]), k = 1..6 */
    float dirx = c[2 * (n - 1)] - c[2 * (n - 2)], diry = c[2 * (n - 1) + 1] - c[2 * (n - 2) + 1];
    int ne = NS * 2;
    for (int k = 1; k <= ne; ++k) {
      dc[2 * (n + k - 1)] = c[2 * (n - 1)] + (float)k * dirx;
      dc[2 * (n + k - 1) + 1] = c[2 * (n - 1) + 1] + (float)k * diry;
    }
    for (int k = n + ne; k < P; ++k) { dc[2 * k] = dc[2 * (n + ne - 1)]; dc[2 * k + 1] = dc[2 * (n + ne - 1) + 1]; } /* :337-345 */
    const float* l = map->left + (size_t)p * S * 2;
    float* dl = o->left + (size_t)p * P * 2;
    memcpy(dl, l, (size_t)nl * 8);
    for (int k = nl; k < P; ++k) { dl[2 * k] = l[2 * (nl - 1)]; dl[2 * k + 1] = l[2 * (nl - 1) + 1]; }              /* :372-374 */
    const float* r = map->right + (size_t)p * S * 2;
    float* dr = o->right + (size_t)p * P * 2;
    memcpy(dr, r, (size_t)nr * 8);
    for (int k = nr; k < P; ++k) { dr[2 * k] = r[2 * (nr - 1)]; dr[2 * k + 1] = r[2 * (nr - 1) + 1]; }              /* :386-388 */
  }
  size_t BN = (size_t)B * N;
  o->state = xcalloc(BN * 8, 4); o->prev_pos = xcalloc(BN * 2, 4); o->vertices = xcalloc(BN * 10, 4);
  o->short_term = xcalloc(BN * NS * 2, 4); o->dist_ref = xcalloc(BN, 4); o->dist_left = xcalloc(BN * 5, 4);
  o->dist_right = xcalloc(BN * 5, 4); o->dist_bound = xcalloc(BN, 4); o->dist_agents = xcalloc(BN * N, 4);
  o->reward = xcalloc(BN, 4); o->reward_info = xcalloc(BN * SIGMAENV_N_REWARD_INFO, 4); o->obs = xcalloc(BN * o->D, 4);
  o->action = xcalloc(BN * 2, 4); o->fresh = xcalloc(BN, 1); o->cbf_nominal = xcalloc(BN * 2, 4); o->path = xcalloc(BN * 4, 4); o->closest = xcalloc(BN * 3, 4);
  o->nearing = xcalloc(BN * (K ? K : 1), 4); o->timer = xcalloc((size_t)B * 4, 4);
  o->col_agents = xcalloc(BN * N, 1); o->col_flags = xcalloc(BN * 4, 1); o->done = xcalloc(B, 1);
  *out = o;
  return SIGMAENV_OK;
}

void sigmaenv_oracle_destroy(oracle_t* o) {
  if (!o) return;
  void* ptrs[] = {o->center, o->left, o->right, o->yaw, o->n_center, o->n_left, o->n_right, o->is_loop, o->state, o->prev_pos,
                  o->vertices, o->short_term, o->dist_ref, o->dist_left, o->dist_right, o->dist_bound, o->dist_agents, o->reward,
                  o->reward_info, o->obs, o->action, o->path, o->closest, o->nearing, o->timer, o->col_agents, o->col_flags, o->done,
                  o->seg_left, o->seg_right, o->cbf_nominal, o->cbf_groups, o->fresh, o->lanelet_centers, o->lanelet_neigh};
  for (size_t k = 0; k < sizeof(ptrs) / sizeof(ptrs[0]); ++k) free(ptrs[k]);
  free(o);
}

const char* sigmaenv_oracle_last_error(const oracle_t* o) { return o ? o->err : "null handle"; }

int sigmaenv_oracle_reset(oracle_t* o, int32_t n, const int32_t* env_idx, const int32_t* agent_idx, const int32_t* path_ids,
                          const float* state8, int32_t full_env) {
  if (!o || n < 0) return SIGMAENV_EINVAL;
  for (int k = 0; k < n; ++k) {
    int b = env_idx[k], i = agent_idx[k];
    if (b < 0 || b >= o->B || i < 0 || i >= o->N || path_ids[4 * k] < 0 || path_ids[4 * k] >= o->n_paths) {
      snprintf(o->err, sizeof(o->err), "reset entry %d out of range", k);
      return SIGMAENV_EINVAL;
    }
  }
  for (int k = 0; k < n; ++k) {
    size_t bi = (size_t)env_idx[k] * o->N + agent_idx[k];
    memcpy(o->state + bi * 8, state8 + 8 * (size_t)k, 32);
    memcpy(o->path + bi * 4, path_ids + 4 * (size_t)k, 16);
  }
  for (int k = 0; k < n; ++k) reset_agent_derived(o, env_idx[k], agent_idx[k]);
  for (int k = 0; k < n; ++k) {
    int seen = 0;
    for (int q = 0; q < k; ++q) seen |= (env_idx[q] == env_idx[k]);
    if (!seen) reset_env_tail(o, env_idx[k], full_env);
  }
  return SIGMAENV_OK;
}

int sigmaenv_oracle_step(oracle_t* o, const float* actions) {
  if (!o || !actions) return SIGMAENV_EINVAL;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < o->B; ++b) step_env(o, b, actions);
  return SIGMAENV_OK;
}

int sigmaenv_oracle_observe(oracle_t* o) {
  if (!o) return SIGMAENV_EINVAL;
  o->obs_salt = ++o->observe_calls;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < o->B; ++b)
    for (int i = 0; i < o->N; ++i) agent_observation(o, b, i);
  o->obs_salt = 0u;
  return SIGMAENV_OK;
}

/* per-agent resets of an unfinished env (road_traffic.py:1435-1447, :1456-1473): agents with a reset request in index order, each
 * sampled against ALL other agents' current positions (is_reset_single_agent, world_state_rt_sim.py:287-309); tries 0..63 use the
 * draws 2000 + 2t / 2001 + 2t, the speed draw 3000.  Then the single-agent reset of road_traffic.py:888-923 (derived state of the
 * agent, env-wide mutual distances, all collision flags of the env cleared, prev_pos := pos) and a fresh observation. */
static void auto_reset_agents(oracle_t* o, int b, uint64_t seed, uint64_t counter, int path_first, int path_count) {
  const int mixed = path_count == SIGMAENV_SCENARIO_LISTS;
  int N = o->N;
  const sigmaenv_config_t* c = &o->cfg;
  float min_d = sqrtf((float)((double)c->length * (double)c->length + (double)c->width * (double)c->width)) * 1.5f;
  float min_d_sq = min_d * min_d;
  int any = 0;
  uint64_t req = 0;
  for (int i = 0; i < N; ++i) if (o->col_flags[((size_t)b * N + i) * 4 + 3]) { req |= 1ull << i; any = 1; }
  if (!any) return;
  for (int i = 0; i < N; ++i) {
    if (!((req >> i) & 1)) continue;
    size_t bi = (size_t)b * N + i;
    float* s = o->state + bi * 8;
    if (mixed) {                                                 /* the agent keeps its env's sub-scenario (:325-328) */
      const int sid = o->path[bi * 4 + 1];
      const int k = (sid >= 1 && sid <= o->n_lists) ? sid - 1 : (o->n_lists > 0 ? o->n_lists - 1 : 0);   /* else branch of world_state_rt_sim.py:345-356: the last list */
      path_first = o->list_first[k]; path_count = o->list_count[k];
    }
    int path = path_first, pt = 3;
    float px = 0.f, py = 0.f;
    for (int t = 0; t < AUTO_RESET_MAX_TRIES; ++t) {
      path = path_first + (int)(((uint64_t)rng_u32(seed, counter, (uint32_t)(c->env_index_base + b), (uint32_t)i, 2000u + 2u * t) * (uint64_t)(uint32_t)path_count) >> 32);
      int n = o->n_center[path];
      int end = reset_end_point(c->is_testing_mode, t, n);
      pt = 3 + (int)(((uint64_t)rng_u32(seed, counter, (uint32_t)(c->env_index_base + b), (uint32_t)i, 2001u + 2u * t) * (uint64_t)(uint32_t)(end - 3)) >> 32);
      px = o->center[((size_t)path * o->P + pt) * 2]; py = o->center[((size_t)path * o->P + pt) * 2 + 1];
      int ok = 1;
      for (int j = 0; j < N; ++j) {
        if (j == i) continue;
        const float* sj = o->state + ((size_t)b * N + j) * 8;
        float dx = px - sj[0], dy = py - sj[1];
        float d2 = dx * dx + dy * dy;
        if (!(d2 >= min_d_sq)) ok = 0;
      }
      if (ok) break;
    }
    float u = (float)(rng_u32(seed, counter, (uint32_t)(c->env_index_base + b), (uint32_t)i, 3000u) >> 8) * (1.0f / 16777216.0f);
    int yi = pt < o->yaw_stride ? pt : o->yaw_stride - 1;
    float rot = o->yaw[(size_t)path * o->yaw_stride + yi];
    float speed = u * c->max_speed;
    s[0] = px; s[1] = py; s[2] = rot; s[3] = speed; s[4] = 0.0f; s[7] = 0.0f;
    s[5] = speed * cr_cos(0.0f + rot);
    s[6] = speed * cr_sin(0.0f + rot);
    o->path[bi * 4 + 0] = path; o->path[bi * 4 + 2] = path - path_first; o->path[bi * 4 + 3] = pt;
  }
  for (int i = 0; i < N; ++i) if ((req >> i) & 1) reset_agent_derived(o, b, i);
  reset_env_tail(o, b, 0);
  for (int i = 0; i < N; ++i) agent_observation(o, b, i);
}

int sigmaenv_oracle_set_scenario_lists(oracle_t* o, int32_t n_lists, const int32_t* first, const int32_t* count, const float* probabilities) {
  if (!o || n_lists < 1 || n_lists > 4 || !first || !count || !probabilities) return SIGMAENV_EINVAL;
  double tot = 0.0, acc = 0.0;
  for (int k = 0; k < n_lists; ++k) {
    if (first[k] < 0 || count[k] < 1 || first[k] + count[k] > o->n_paths || !(probabilities[k] >= 0.0f)) return SIGMAENV_EINVAL;
    tot += (double)probabilities[k];
  }
  if (!(tot > 0.0)) return SIGMAENV_EINVAL;
  for (int k = 0; k < 4; ++k) {
    o->list_first[k] = k < n_lists ? first[k] : 0;
    o->list_count[k] = k < n_lists ? count[k] : 1;
    if (k < n_lists) acc += (double)probabilities[k] / tot;
    o->list_cdf[k] = k + 1 >= n_lists ? 1.0f : (float)acc;
  }
  o->n_lists = n_lists;
  return SIGMAENV_OK;
}

int sigmaenv_oracle_auto_reset(oracle_t* o, uint64_t seed, uint64_t counter, int32_t path_first, int32_t path_count) {
  if (!o) return SIGMAENV_EINVAL;
  if (path_count == SIGMAENV_SCENARIO_LISTS ? o->n_lists < 1 : (path_first < 0 || path_count < 1 || path_first + path_count > o->n_paths)) return SIGMAENV_EINVAL;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < o->B; ++b) {
    if (o->done[b]) auto_reset_env(o, b, seed, counter, path_first, path_count);
    else auto_reset_agents(o, b, seed, counter, path_first, path_count);
  }
  return SIGMAENV_OK;
}

int sigmaenv_oracle_get(oracle_t* o, sigmaenv_buf_t which, void** ptr, size_t* bytes) {
  if (!o || !ptr || !bytes) return SIGMAENV_EINVAL;
  size_t BN = (size_t)o->B * o->N;
  switch (which) {
    case SIGMAENV_BUF_STATE: *ptr = o->state; *bytes = BN * 32; break;
    case SIGMAENV_BUF_PREV_POS: *ptr = o->prev_pos; *bytes = BN * 8; break;
    case SIGMAENV_BUF_VERTICES: *ptr = o->vertices; *bytes = BN * 40; break;
    case SIGMAENV_BUF_PATH: *ptr = o->path; *bytes = BN * 16; break;
    case SIGMAENV_BUF_SHORT_TERM: *ptr = o->short_term; *bytes = BN * NS * 8; break;
    case SIGMAENV_BUF_DIST_REF: *ptr = o->dist_ref; *bytes = BN * 4; break;
    case SIGMAENV_BUF_DIST_LEFT: *ptr = o->dist_left; *bytes = BN * 20; break;
    case SIGMAENV_BUF_DIST_RIGHT: *ptr = o->dist_right; *bytes = BN * 20; break;
    case SIGMAENV_BUF_DIST_BOUND: *ptr = o->dist_bound; *bytes = BN * 4; break;
    case SIGMAENV_BUF_CLOSEST: *ptr = o->closest; *bytes = BN * 12; break;
    case SIGMAENV_BUF_DIST_AGENTS: *ptr = o->dist_agents; *bytes = BN * o->N * 4; break;
    case SIGMAENV_BUF_COL_AGENTS: *ptr = o->col_agents; *bytes = BN * o->N; break;
    case SIGMAENV_BUF_COL_FLAGS: *ptr = o->col_flags; *bytes = BN * 4; break;
    case SIGMAENV_BUF_REWARD: *ptr = o->reward; *bytes = BN * 4; break;
    case SIGMAENV_BUF_REWARD_INFO: *ptr = o->reward_info; *bytes = BN * SIGMAENV_N_REWARD_INFO * 4; break;
    case SIGMAENV_BUF_OBS: *ptr = o->obs; *bytes = BN * o->D * 4; break;
    case SIGMAENV_BUF_NEARING: *ptr = o->nearing; *bytes = BN * o->K * 4; break;
    case SIGMAENV_BUF_DONE: *ptr = o->done; *bytes = (size_t)o->B; break;
    case SIGMAENV_BUF_TIMER: *ptr = o->timer; *bytes = (size_t)o->B * 16; break;
    case SIGMAENV_BUF_ACTION: *ptr = o->action; *bytes = BN * 8; break;
    case SIGMAENV_BUF_CBF_NOMINAL: *ptr = o->cbf_nominal; *bytes = BN * 8; break;
    default: return SIGMAENV_EINVAL;
  }
  return SIGMAENV_OK;
}

int sigmaenv_oracle_sync(oracle_t* o) { (void)o; return SIGMAENV_OK; }

/* padded path table access for tests (checks the padding against the reference's per-agent copies) */
int sigmaenv_oracle_path_table(oracle_t* o, int32_t* P, const float** center, const float** left, const float** right) {
  if (!o) return SIGMAENV_EINVAL;
  *P = o->P; *center = o->center; *left = o->left; *right = o->right;
  return SIGMAENV_OK;
}

/* ---- standalone entry points for the function-level goldens (tests only) ---------------------------------------- */
void sigmaenv_oracle_fn_bicycle(const sigmaenv_config_t* cfg, int n, const float* in7 /*x,y,psi,v,delta,u0,u1*/, float* out10) {
  for (int k = 0; k < n; ++k) {
    float s[8] = {in7[7 * k], in7[7 * k + 1], in7[7 * k + 2], in7[7 * k + 3], in7[7 * k + 4], 0, 0, 0};
    float uc[2];
    bicycle_step(cfg, s, in7 + 7 * k + 5, uc);
    memcpy(out10 + 10 * k, s, 32);
    out10[10 * k + 8] = uc[0]; out10[10 * k + 9] = uc[1];
  }
}
void sigmaenv_oracle_fn_vertices(const sigmaenv_config_t* cfg, int n, const float* in3, float* out10) {
  for (int k = 0; k < n; ++k) rect_vertices(cfg, in3[3 * k], in3[3 * k + 1], in3[3 * k + 2], out10 + 10 * k);
}
void sigmaenv_oracle_fn_point_polyline(int n, const float* pts, const float* poly, int n_points, float* dist, int32_t* idx) {
  for (int k = 0; k < n; ++k) point_polyline(pts[2 * k], pts[2 * k + 1], poly, n_points, dist + k, idx + k);
}
void sigmaenv_oracle_fn_short_term(int n, const float* poly, int n_points, int is_loop, const int32_t* cp, float* out6) {
  for (int k = 0; k < n; ++k) short_term_path(poly, n_points, is_loop, cp[k], out6 + 6 * k);
}
void sigmaenv_oracle_fn_interx(int n, const float* L1, int n1, int stride1, const float* L2, int n2, int stride2, uint8_t* hit) {
  for (int k = 0; k < n; ++k) hit[k] = (uint8_t)interx(L1 + (size_t)k * stride1, n1, L2 + (size_t)k * stride2, n2);
}
void sigmaenv_oracle_fn_mtv(int n, const float* verts /*[n,2,5,2]*/, float* d) {
  for (int k = 0; k < n; ++k) d[k] = mtv_pair(verts + 20 * (size_t)k, verts + 20 * (size_t)k + 10);
}
void sigmaenv_oracle_fn_ego(int n, const float* pi, const float* roti, int m, const float* pj, float* out) {
  for (int k = 0; k < n; ++k)
    for (int q = 0; q < m; ++q)
      ego_transform(pi[2 * k], pi[2 * k + 1], roti[k], pj[((size_t)k * m + q) * 2], pj[((size_t)k * m + q) * 2 + 1],
                    out + ((size_t)k * m + q) * 2, out + ((size_t)k * m + q) * 2 + 1);
}
void sigmaenv_oracle_fn_wrap(int n, const float* a, float* out) {
  for (int k = 0; k < n; ++k) out[k] = angle_eliminate_two_pi(a[k]);
}

#include "sigmaenv_cbf_oracle.inc"

/* the contract's trig on arrays (tests/test_trig.py): kind 0 sin, 1 cos, 2 tan, 3 atan (b unused), 4 atan2(a, b) */
void sigmaenv_oracle_fn_trig(int kind, int n, const float* a, const float* b, float* out) {
  for (int k = 0; k < n; ++k) {
    switch (kind) {
      case 0: out[k] = cr_sin(a[k]); break;
      case 1: out[k] = cr_cos(a[k]); break;
      case 2: out[k] = cr_tan(a[k]); break;
      case 3: out[k] = cr_atan(a[k]); break;
      default: out[k] = cr_atan2(a[k], b[k]); break;
    }
  }
}
