// sigmaenv.hip -- fused vectorized multi-agent CAV environment step for MI355X (gfx950) + its C-ABI (include/sigmaenv.h).
//
// One workgroup = one environment.  Phases of the fused step kernel (one launch per step, all agents x envs):
//   A  per agent      : action clamp + kinematic bicycle Euler step, new rectangle vertices            (K1, K2)
//   B1 per agent pair : mutual distances (c2c / mtv) and rectangle-rectangle collision masks          (K4, K5)
//   B2 wave per agent : 11 point->polyline queries and 2 rectangle->boundary collision scans,
//                       64 lanes across polyline segments, wavefront shuffles for (min, first argmin)   (K3, K5)
//   C  per agent      : reward terms, short-term reference path, counters                               (K7, K6)
//   D  per obs item   : top-k nearest agents, ego-view transforms, observation vector in LDS            (K8)
//   E  per env        : done flag and per-agent reset requests, coalesced write-out                     (K9)
// Per-env agent poses / vertices / distance rows live in LDS between the phases; HBM sees each state word once in, once out.
// MFMA is unused on purpose: there is no dense contraction in this path.
//
// Reference citations are relative to /root/reference/sigmarl; see sigmaenv_device.h for the arithmetic contract.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sigmaenv_device.h"

using namespace sigmadev;

// weighting_ref_directions = linspace(1, 0.2, 3) / sum (road_traffic.py:536-543); bit patterns of the reference's tensor
__device__ __constant__ uint32_t W_REF_BITS[3] = {0x3F0E38E3u, 0x3EAAAAABu, 0x3DE38E39u};

#define AUTO_RESET_MAX_TRIES 64

// ---------------------------------------------------------------------------------------------------------------------
// LDS carve-up for one environment
// ---------------------------------------------------------------------------------------------------------------------
struct Smem {
  float *st, *vold, *vnew, *shrt, *dref, *dleft, *dright, *dbound, *dist, *obs;
  int *path, *cp, *near, *flags;
  uint8_t* col;
  __device__ Smem(char* base, int N, int K, int D) {
    float* f = reinterpret_cast<float*>(base);
    st = f; f += N * 8;
    vold = f; f += N * 10;
    vnew = f; f += N * 10;
    shrt = f; f += N * NS * 2;
    dref = f; f += N;
    dleft = f; f += N * 5;
    dright = f; f += N * 5;
    dbound = f; f += N;
    dist = f; f += N * N;
    obs = f; f += N * D;
    int* i = reinterpret_cast<int*>(f);
    path = i; i += N;
    cp = i; i += N * 3;
    near = i; i += N * (K > 0 ? K : 1);
    flags = i; i += N * 4;
    col = reinterpret_cast<uint8_t*>(i);
  }
  static size_t bytes(int N, int K, int D) {
    size_t f = (size_t)N * 8 + N * 10 * 2 + N * NS * 2 + N + N * 5 * 2 + N + (size_t)N * N + (size_t)N * D;
    size_t i = (size_t)N + N * 3 + N * (K > 0 ? K : 1) + N * 4;
    return (f + i) * 4 + (size_t)N * N + 16;
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// B2: all distance queries + boundary collisions of one agent, cooperatively by one wavefront.
// update_distances (world_state_rt.py:582-656) + the boundary part of update_collisions (world_state_rt_sim.py:398-411).
// qv: the 4 corner QUERY points (agent 0 uses last step's vertices, see step kernel), ev: the 5 closed-rectangle vertices
// used for the collision scan.  Results are valid in every lane.
// ---------------------------------------------------------------------------------------------------------------------
struct AgentScan {
  float d_ref, dl[5], dr[5];
  int cp_ref, cp_l, cp_r;
  bool hit;
};

template <bool COLLIDE>
__device__ __forceinline__ void boundary_scan(const float* __restrict__ poly, int n, int lane, float cgx, float cgy, const float* qv,
                                              const Edge* e, float d_out[5], int& cp_out, bool& hit) {
  float bd[5];
  int bk = 0;
#pragma unroll
  for (int q = 0; q < 5; ++q) bd[q] = INFINITY;
  bool h = false;
  const float2* p2 = reinterpret_cast<const float2*>(poly);
  for (int k = lane; k + 1 < n; k += 64) {
    float2 a = p2[k], b = p2[k + 1];
    float lx = b.x - a.x, ly = b.y - a.y;
    float len2 = lx * lx + ly * ly;
    float d0 = point_segment(cgx, cgy, a.x, a.y, lx, ly, len2);
    if (d0 < bd[0]) { bd[0] = d0; bk = k; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float d = point_segment(qv[2 * q], qv[2 * q + 1], a.x, a.y, lx, ly, len2);
      bd[q + 1] = fminf(bd[q + 1], d);
    }
    if (COLLIDE) {
      float S2 = lx * a.y - ly * a.x;
#pragma unroll
      for (int i = 0; i < 4; ++i) h |= edge_hits_segment(e[i], a.x, a.y, b.x, b.y, lx, ly, S2);
    }
  }
  wave_argmin(bd[0], bk);
#pragma unroll
  for (int q = 1; q < 5; ++q) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) bd[q] = fminf(bd[q], __shfl_xor(bd[q], off, 64));
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) d_out[q] = bd[q];
  cp_out = bk + 1;
  if (COLLIDE) hit = hit || (__ballot(h) != 0ull);
}

template <bool COLLIDE>
__device__ inline void agent_scan(const DevMap& m, const sigmaenv_config_t& c, int path, int lane, float cgx, float cgy, const float* qv,
                                  const float* ev, AgentScan& r) {
  const float* ctr = m.center + (size_t)path * m.P * 2;
  const float* lb = m.left + (size_t)path * m.P * 2;
  const float* rb = m.right + (size_t)path * m.P * 2;
  int n = m.n_center[path], nl = m.n_left[path], nr = m.n_right[path];
  // centre line: CG only
  {
    float bd = INFINITY;
    int bk = 0;
    const float2* p2 = reinterpret_cast<const float2*>(ctr);
    for (int k = lane; k + 1 < n; k += 64) {
      float2 a = p2[k], b = p2[k + 1];
      float lx = b.x - a.x, ly = b.y - a.y;
      float d = point_segment(cgx, cgy, a.x, a.y, lx, ly, lx * lx + ly * ly);
      if (d < bd) { bd = d; bk = k; }
    }
    wave_argmin(bd, bk);
    r.d_ref = bd;
    r.cp_ref = bk + 1;
  }
  Edge e[4];
  if (COLLIDE) {
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = make_edge(ev[2 * i], ev[2 * i + 1], ev[2 * i + 2], ev[2 * i + 3]);
  }
  r.hit = false;
  boundary_scan<COLLIDE>(lb, nl, lane, cgx, cgy, qv, e, r.dl, r.cp_l, r.hit);
  boundary_scan<COLLIDE>(rb, nr, lane, cgx, cgy, qv, e, r.dr, r.cp_r, r.hit);
  float wh = (float)((double)c.width / 2.0);
  r.dl[0] = r.dl[0] - wh;  // world_state_rt.py:608-610
  r.dr[0] = r.dr[0] - wh;
}

// ---------------------------------------------------------------------------------------------------------------------
// B2, pruned: the same (min, first argmin) / collision results as the full scan, but only the polyline chunks that can matter
// are evaluated.  Exactness argument (DESIGN.md "Pruned scan"): a run of 8 consecutive segments is skipped only if the
// distance from the agent's centre to the run's bounding box exceeds T, where T bounds (with a 1e-4 m margin, >> fp32 error)
//   * the distance of every query point to the segment that was closest last step (an upper bound of its minimum), and
//   * the rectangle's circumradius (a segment farther than that cannot intersect an edge).
// So every segment whose computed distance can equal or beat the running minimum, and every segment that can hit the
// rectangle, is still evaluated with the very same arithmetic; ties still resolve to the lowest index.
// Lanes 0-31 scan the left boundary, lanes 32-63 the right one; the centre line uses all 64 lanes.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t chunk_mask(const float4* __restrict__ box, int nch, int lane, float px, float py, float T) {
  bool cand = false;
  if (lane < nch) {
    float4 b = box[lane];
    float dx = fmaxf(fmaxf(b.x - px, px - b.z), 0.0f);
    float dy = fmaxf(fmaxf(b.y - py, py - b.w), 0.0f);
    float lb = sqrtf(dx * dx + dy * dy);
    cand = !(lb > T);  // a NaN threshold keeps every chunk
  }
  return (uint32_t)__ballot(cand);
}
__device__ __forceinline__ int nth_set_bit(uint32_t m, int c) {
  for (int t = 0; t < c; ++t) m &= (m - 1u);
  return m ? (__ffs((int)m) - 1) : -1;
}
__device__ __forceinline__ float guess_distance(const float* __restrict__ poly, int n, int cp_guess, float px, float py) {
  int k = cp_guess - 1;
  k = k < 0 ? 0 : (k > n - 2 ? n - 2 : k);
  const float2* p2 = reinterpret_cast<const float2*>(poly);
  float2 a = p2[k], b = p2[k + 1];
  float lx = b.x - a.x, ly = b.y - a.y;
  return point_segment(px, py, a.x, a.y, lx, ly, lx * lx + ly * ly);
}

template <bool COLLIDE>
__device__ inline void agent_scan_pruned(const DevMap& m, const sigmaenv_config_t& c, int path, int lane, float cgx, float cgy, const float* qv,
                                         const float* ev, const int* cp_guess, AgentScan& r) {
  const float* ctr = m.center + (size_t)path * m.P * 2;
  const float* lb = m.left + (size_t)path * m.P * 2;
  const float* rb = m.right + (size_t)path * m.P * 2;
  const int n = m.n_center[path], nl = m.n_left[path], nr = m.n_right[path];
  const float4* box = m.chunk_box + (size_t)path * 3 * m.nch;
  const float MARGIN = 1e-4f;
  // radii of the query points / rectangle vertices around the centre of gravity
  float Rq = 0.0f, Rv = 0.0f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float ax = qv[2 * q] - cgx, ay = qv[2 * q + 1] - cgy;
    Rq = fmaxf(Rq, sqrtf(ax * ax + ay * ay));
    float bx = ev[2 * q] - cgx, by = ev[2 * q + 1] - cgy;
    Rv = fmaxf(Rv, sqrtf(bx * bx + by * by));
  }
  // ---- centre line (CG only), 64 lanes over the candidate segments
  {
    float T = guess_distance(ctr, n, cp_guess[0], cgx, cgy) + MARGIN;
    int nch = (n - 1 + SIGMAENV_CHUNK - 1) / SIGMAENV_CHUNK;
    uint32_t mc = chunk_mask(box, nch, lane, cgx, cgy, T);
    int cnt = __popc(mc) * SIGMAENV_CHUNK;
    float bd = INFINITY;
    int bk = 0;
    const float2* p2 = reinterpret_cast<const float2*>(ctr);
    for (int base = 0; base < cnt; base += 64) {
      int j = base + lane;
      int ch = nth_set_bit(mc, j >> 3);
      int k = ch * SIGMAENV_CHUNK + (j & 7);
      if (ch >= 0 && k + 1 < n) {
        float2 a = p2[k], b = p2[k + 1];
        float lx = b.x - a.x, ly = b.y - a.y;
        float d = point_segment(cgx, cgy, a.x, a.y, lx, ly, lx * lx + ly * ly);
        if (d < bd || (d == bd && k < bk)) { bd = d; bk = k; }
      }
    }
    wave_argmin(bd, bk);
    r.d_ref = bd;
    r.cp_ref = bk + 1;
  }
  // ---- boundaries: half-wave per side
  const int half = lane >> 5, hl = lane & 31;
  const float* poly = half ? rb : lb;
  const int np = half ? nr : nl;
  float Tl = fmaxf(guess_distance(lb, nl, cp_guess[1], cgx, cgy) + 2.0f * Rq, COLLIDE ? Rv : 0.0f) + MARGIN;
  float Tr = fmaxf(guess_distance(rb, nr, cp_guess[2], cgx, cgy) + 2.0f * Rq, COLLIDE ? Rv : 0.0f) + MARGIN;
  uint32_t ml = chunk_mask(box + m.nch, (nl - 1 + SIGMAENV_CHUNK - 1) / SIGMAENV_CHUNK, lane, cgx, cgy, Tl);
  uint32_t mr = chunk_mask(box + 2 * m.nch, (nr - 1 + SIGMAENV_CHUNK - 1) / SIGMAENV_CHUNK, lane, cgx, cgy, Tr);
  const uint32_t mm = half ? mr : ml;
  const int cnt_max = max(__popc(ml), __popc(mr)) * SIGMAENV_CHUNK;
  Edge e[4];
  if (COLLIDE) {
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = make_edge(ev[2 * i], ev[2 * i + 1], ev[2 * i + 2], ev[2 * i + 3]);
  }
  float bd[5];
  int bk = 0;
#pragma unroll
  for (int q = 0; q < 5; ++q) bd[q] = INFINITY;
  bool h = false;
  const float2* p2 = reinterpret_cast<const float2*>(poly);
  for (int base = 0; base < cnt_max; base += 32) {
    int j = base + hl;
    int ch = nth_set_bit(mm, j >> 3);
    int k = ch * SIGMAENV_CHUNK + (j & 7);
    if (ch >= 0 && k + 1 < np) {
      float2 a = p2[k], b = p2[k + 1];
      float lx = b.x - a.x, ly = b.y - a.y;
      float len2 = lx * lx + ly * ly;
      float d0 = point_segment(cgx, cgy, a.x, a.y, lx, ly, len2);
      if (d0 < bd[0] || (d0 == bd[0] && k < bk)) { bd[0] = d0; bk = k; }
#pragma unroll
      for (int q = 0; q < 4; ++q) bd[q + 1] = fminf(bd[q + 1], point_segment(qv[2 * q], qv[2 * q + 1], a.x, a.y, lx, ly, len2));
      if (COLLIDE) {
        float S2 = lx * a.y - ly * a.x;
#pragma unroll
        for (int i = 0; i < 4; ++i) h |= edge_hits_segment(e[i], a.x, a.y, b.x, b.y, lx, ly, S2);
      }
    }
  }
  // reductions inside each 32-lane half (xor offsets < 32 never cross the halves)
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) {
    float od = __shfl_xor(bd[0], off, 64);
    int ok = __shfl_xor(bk, off, 64);
    if (od < bd[0] || (od == bd[0] && ok < bk)) { bd[0] = od; bk = ok; }
#pragma unroll
    for (int q = 1; q < 5; ++q) bd[q] = fminf(bd[q], __shfl_xor(bd[q], off, 64));
  }
  float wh = (float)((double)c.width / 2.0);
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    r.dl[q] = __shfl(bd[q], 0, 64);
    r.dr[q] = __shfl(bd[q], 32, 64);
  }
  r.cp_l = __shfl(bk, 0, 64) + 1;
  r.cp_r = __shfl(bk, 32, 64) + 1;
  r.dl[0] = r.dl[0] - wh;
  r.dr[0] = r.dr[0] - wh;
  r.hit = COLLIDE ? (__ballot(h) != 0ull) : false;
}

template <bool COLLIDE>
__device__ __forceinline__ void agent_scan_any(const DevMap& m, const sigmaenv_config_t& c, int path, int lane, float cgx, float cgy,
                                               const float* qv, const float* ev, const int* cp_guess, AgentScan& r) {
  if (m.nch > 0) agent_scan_pruned<COLLIDE>(m, c, path, lane, cgx, cgy, qv, ev, cp_guess, r);
  else agent_scan<COLLIDE>(m, c, path, lane, cgx, cgy, qv, ev, r);
}

// mutual distance of pair (i, j), i != j  (update_mutual_distances, world_state_rt_sim.py:360-373)
__device__ __forceinline__ float pair_distance(const sigmaenv_config_t& c, const float* st, const float* verts, int i, int j) {
  if (c.distance_type == SIGMAENV_DIST_C2C) {
    float dx = st[i * 8] - st[j * 8], dy = st[i * 8 + 1] - st[j * 8 + 1];
    return sqrtf(dx * dx + dy * dy);
  }
  int a = i < j ? i : j, b = i < j ? j : i;
  return mtv_pair(verts + a * 10, verts + b * 10);
}

// _apply_ttc_near_agent_penalty, road_traffic.py:1255-1332
__device__ inline float ttc_penalty(const sigmaenv_config_t& c, const float* st, int N, int i) {
  const float eps = 1e-6f;
  double d_safe = (double)c.threshold_near_other_agents_low;
  float d_safe_sq = (float)(d_safe * d_safe);
  float d_safe32 = c.threshold_near_other_agents_low, d_gate = c.threshold_near_other_agents_high;
  const float* si = st + i * 8;
  float risk_sum = 0.0f;
  for (int j = 0; j < N; ++j) {
    const float* sj = st + j * 8;
    float px = sj[0] - si[0], py = sj[1] - si[1];
    float vx = sj[5] - si[5], vy = sj[6] - si[6];
    float a = vx * vx + vy * vy;
    float bq = 2.0f * (px * vx + py * vy);
    float pp = px * px + py * py;
    float cq = pp - d_safe_sq;
    float disc = bq * bq - 4.0f * a * cq;
    float sq = sqrtf(fmaxf(disc, 0.0f));
    float dist = sqrtf(fmaxf(pp, 0.0f));
    bool valid = (a > eps) && (disc > 0.0f) && (bq < 0.0f);
    float cand = (-bq - sq) / (2.0f * a + eps);
    float ttc = INFINITY;
    if (valid && cand > 0.0f) ttc = cand;
    if (dist <= d_safe32) ttc = 0.0f;
    if (j == i) ttc = INFINITY;
    if (!(dist <= d_gate)) ttc = INFINITY;
    float x = fminf(ttc, c.ttc_high);
    risk_sum += decreasing_lin(x, c.ttc_low, c.ttc_high);
  }
  float risk = risk_sum / (float)(N - 1 > 1 ? N - 1 : 1);
  return risk * c.penalty_near_other_agents;
}

// top-k nearest agents of agent i (observation_provider_rt.py:629-636): ascending, lowest index on ties
__device__ inline void topk_nearest(const float* Drow, int N, int K, int* out) {
  unsigned long long taken = 0ull;
  for (int k = 0; k < K; ++k) {
    int bj = -1;
    float bd = INFINITY;
    for (int j = 0; j < N; ++j) {
      if ((taken >> j) & 1ull) continue;
      if (bj < 0 || Drow[j] < bd) { bd = Drow[j]; bj = j; }
    }
    taken |= 1ull << bj;
    out[k] = bj;
  }
}

// one observation work item (observation_provider_rt.py:345-588 latest slot, :594-925 default flags); writes into s.obs
__device__ inline void obs_item(const sigmaenv_config_t& c, const Smem& s, int N, int K, int D, int i, int q) {
  const float* si = s.st + i * 8;
  float* ob = s.obs + i * D;
  float n_pos = (float)((double)c.length * 10.0);
  float n_v = c.max_speed;
  float n_dl = (float)((double)c.lane_width * 3.0);
  if (q == 0) {
    float rr = angle_eliminate_two_pi(si[2] - si[2]);
    ob[0] = (norm2(si[5], si[6]) * cr_cos(rr)) / n_v;
    ob[1 + 2 * NS] = s.dref[i] / n_dl;
    float ml = INFINITY, mr = INFINITY;
#pragma unroll
    for (int t = 0; t < 5; ++t) { ml = fminf(ml, s.dleft[i * 5 + t]); mr = fminf(mr, s.dright[i * 5 + t]); }
    ob[2 + 2 * NS] = ml / n_dl;
    ob[3 + 2 * NS] = mr / n_dl;
  } else if (q <= NS) {
    int k = q - 1;
    float ox, oy;
    ego_transform(si[0], si[1], si[2], s.shrt[i * NS * 2 + 2 * k], s.shrt[i * NS * 2 + 2 * k + 1], ox, oy);
    ob[1 + 2 * k] = ox / n_pos;
    ob[2 + 2 * k] = oy / n_pos;
  } else if (q < 1 + NS + 4 * K) {
    int w = q - (1 + NS);
    int k = w >> 2, v = w & 3;
    int j = s.near[i * K + k];
    float ox, oy;
    ego_transform(si[0], si[1], si[2], s.vnew[j * 10 + 2 * v], s.vnew[j * 10 + 2 * v + 1], ox, oy);
    int base = 4 + 2 * NS + 11 * k;
    ob[base + 2 * v] = ox / n_pos;
    ob[base + 2 * v + 1] = oy / n_pos;
  } else {
    int k = q - (1 + NS + 4 * K);
    int j = s.near[i * K + k];
    const float* sj = s.st + j * 8;
    float rr = angle_eliminate_two_pi(sj[2] - si[2]);
    float va = norm2(sj[5], sj[6]);
    int base = 4 + 2 * NS + 11 * k;
    ob[base + 8] = (va * cr_cos(rr)) / n_v;
    ob[base + 9] = (va * cr_sin(rr)) / n_v;
    ob[base + 10] = s.dist[i * N + j] / n_dl;
  }
}

// D + write-out of the observation of one env: top-k, items, coalesced store.  All threads of the block participate.
__device__ inline void observe_env(const sigmaenv_config_t& c, const Smem& s, const DevBufs& g, int b, int N, int K, int D) {
  for (int i = threadIdx.x; i < N; i += blockDim.x) topk_nearest(s.dist + i * N, N, K, s.near + i * K);
  __syncthreads();
  const int items = 1 + NS + 5 * K;
  for (int w = threadIdx.x; w < N * items; w += blockDim.x) obs_item(c, s, N, K, D, w / items, w % items);
  __syncthreads();
  for (int k = threadIdx.x; k < N * D; k += blockDim.x) g.obs[(size_t)b * N * D + k] = s.obs[k];
  for (int k = threadIdx.x; k < N * K; k += blockDim.x) g.nearing[(size_t)b * N * K + k] = s.near[k];
}

// ---------------------------------------------------------------------------------------------------------------------
// the fused step kernel: grid = n_envs, block = 64 * WAVES
// VMAS >= 1.4 call order restated per env: world.step(); reward(a) for all a; observation(a) for all a; done()
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) sigmaenv_step_kernel(sigmaenv_config_t c, DevMap m, DevBufs g, const float* __restrict__ actions) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int N = c.n_agents, K = c.n_nearing;
  const int D = 4 + 2 * NS + 11 * K;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
  Smem s(smem_raw, N, K, D);
  const size_t bN = (size_t)b * N;

  // ---- A: dynamics + vertices (one thread per agent) -------------------------------------------------------------
  for (int i = tid; i < N; i += blockDim.x) {
    const size_t bi = bN + i;
    float st[8], uc[2];
    const float4* gs = reinterpret_cast<const float4*>(g.state + bi * 8);
    float4 s0 = gs[0], s1 = gs[1];
    st[0] = s0.x; st[1] = s0.y; st[2] = s0.z; st[3] = s0.w; st[4] = s1.x; st[5] = s1.y; st[6] = s1.z; st[7] = s1.w;
    float2 u = reinterpret_cast<const float2*>(actions)[bi];
    bicycle_step(c, st, u.x, u.y, uc);
#pragma unroll
    for (int k = 0; k < 8; ++k) s.st[i * 8 + k] = st[k];
    reinterpret_cast<float2*>(g.action)[bi] = make_float2(uc[0], uc[1]);
    float4* go = reinterpret_cast<float4*>(g.state + bi * 8);
    go[0] = make_float4(st[0], st[1], st[2], st[3]);
    go[1] = make_float4(st[4], st[5], st[6], st[7]);
    // last step's vertices: agent 0's corner queries and the whole mtv matrix still see them, because update_distances
    // runs before update_vertices (world_state_rt_sim.py:439-448)
#pragma unroll
    for (int k = 0; k < 10; ++k) s.vold[i * 10 + k] = g.vertices[bi * 10 + k];
    float v[10];
    rect_vertices(c, st[0], st[1], st[2], v);
#pragma unroll
    for (int k = 0; k < 10; ++k) { s.vnew[i * 10 + k] = v[k]; g.vertices[bi * 10 + k] = v[k]; }
    s.path[i] = g.path[bi * 4];
    // last step's closest-point indices: the pruned scan derives its distance upper bounds from them
    s.cp[i * 3 + 0] = g.closest[bi * 3 + 0]; s.cp[i * 3 + 1] = g.closest[bi * 3 + 1]; s.cp[i * 3 + 2] = g.closest[bi * 3 + 2];
  }
  __syncthreads();

  // ---- B1: mutual distances + agent-agent collisions (one thread per ordered pair) -------------------------------
  const float diag = sqrtf(c.world_x_dim * c.world_x_dim + c.world_y_dim * c.world_y_dim);  // helper_scenario.py:1140-1143
  for (int p = tid; p < N * N; p += blockDim.x) {
    int i = p / N, j = p - i * N;
    float d = (i == j) ? diag : pair_distance(c, s.st, s.vold, i, j);
    s.dist[p] = d;
    g.dist_agents[bN * N + p] = d;
    uint8_t col = 0;
    if (c.distance_type == SIGMAENV_DIST_C2C) {
      if (i != j) col = interx_rect_rect(s.vnew + (i < j ? i : j) * 10, s.vnew + (i < j ? j : i) * 10) ? 1 : 0;  // world_state_rt_sim.py:382-393
    } else {
      col = (d == 0.0f) ? 1 : 0;  // :394-396
    }
    s.col[p] = col;
    g.col_agents[bN * N + p] = col;
  }

  // ---- B2: distance queries + boundary collisions (one wavefront per agent) --------------------------------------
  for (int i = wave; i < N; i += n_waves) {
    const float* si = s.st + i * 8;
    const float* qv = (i == 0) ? (s.vold) : (s.vnew + i * 10);  // agent 0 queries its corners at last step's vertices
    AgentScan r;
    agent_scan_any<true>(m, c, s.path[i], lane, si[0], si[1], qv, s.vnew + i * 10, s.cp + i * 3, r);
    int path = s.path[i];
    bool entry = false, exit_ = false;
    if (!m.is_loop[path]) {  // world_state_rt_sim.py:413-424
      const float* lb = m.left + (size_t)path * m.P * 2;
      const float* rb = m.right + (size_t)path * m.P * 2;
      int nl = m.n_left[path], nr = m.n_right[path];
      entry = interx_rect_seg(s.vnew + i * 10, lb[0], lb[1], rb[0], rb[1]);
      exit_ = interx_rect_seg(s.vnew + i * 10, lb[2 * (nl - 1)], lb[2 * (nl - 1) + 1], rb[2 * (nr - 1)], rb[2 * (nr - 1) + 1]);
    }
    if (lane == 0) {
      s.dref[i] = r.d_ref;
      float mb = INFINITY;
#pragma unroll
      for (int q = 0; q < 5; ++q) { s.dleft[i * 5 + q] = r.dl[q]; s.dright[i * 5 + q] = r.dr[q]; }
#pragma unroll
      for (int q = 0; q < 5; ++q) mb = fminf(mb, r.dl[q]);
#pragma unroll
      for (int q = 0; q < 5; ++q) mb = fminf(mb, r.dr[q]);
      s.dbound[i] = mb;
      s.cp[i * 3 + 0] = r.cp_ref; s.cp[i * 3 + 1] = r.cp_l; s.cp[i * 3 + 2] = r.cp_r;
      s.flags[i * 4 + 0] = r.hit ? 1 : 0; s.flags[i * 4 + 1] = entry ? 1 : 0; s.flags[i * 4 + 2] = exit_ ? 1 : 0;
    }
  }
  __syncthreads();

  // ---- C: reward, short-term path, bookkeeping (one thread per agent) ---------------------------------------------
  const size_t BN = (size_t)c.n_envs * N;
  for (int i = tid; i < N; i += blockDim.x) {
    const size_t bi = bN + i;
    const float* si = s.st + i * 8;
    float2 pp = reinterpret_cast<const float2*>(g.prev_pos)[bi];
    float w[3];
    w[0] = __uint_as_float(W_REF_BITS[0]); w[1] = __uint_as_float(W_REF_BITS[1]); w[2] = __uint_as_float(W_REF_BITS[2]);
    float mvx = si[0] - pp.x, mvy = si[1] - pp.y;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < NS; ++k) {  // the short-term path of the PREVIOUS step is still in HBM (road_traffic.py:976-984)
      float rx = g.short_term[bi * NS * 2 + 2 * k] - pp.x, ry = g.short_term[bi * NS * 2 + 2 * k + 1] - pp.y;
      float mp = mvx * rx + mvy * ry;
      acc = acc + mp * w[k];
    }
    float denom = (float)((double)c.max_speed * (double)c.dt);
    float rew = 0.0f;
    rew += acc / denom * c.reward_progress;
    int goal = s.flags[i * 4 + 2];
    float reward_goal = (float)goal * c.reward_reach_goal;
    int col_a = 0;
    for (int j = 0; j < N; ++j) col_a |= s.col[i * N + j];
    s.flags[i * 4 + 3] = col_a;
    float pca = (float)col_a * c.penalty_collide_with_agents;
    int col_l = s.flags[i * 4 + 0];
    float pcl = (float)col_l * c.penalty_collide_with_boundaries;
    float pen_lane = decreasing_lin(s.dbound[i], c.threshold_near_boundary_low, c.threshold_near_boundary_high) * c.penalty_near_boundary;
    bool has_near = false;
    float near_other = 0.0f;
    if (c.is_testing_mode) {
      rew += reward_goal; rew += pca; rew += pcl;
    } else {
      if (c.rew_flags & SIGMAENV_REW_EXACT_SPARSE) { rew += pca; rew += pcl; }
      if (c.rew_flags & SIGMAENV_REW_TTC) {
        float p = ttc_penalty(c, s.st, N, i);
        near_other = p; has_near = true;
        rew += p; rew += pen_lane; rew += pca; rew += pcl;
        if (c.rew_flags & SIGMAENV_REW_HAS_SPARSE) { rew += pca; rew += pcl; }
      }
      if (c.rew_flags & SIGMAENV_REW_DISTANCE) {
        float ssum = 0.0f;
        for (int j = 0; j < N; ++j) ssum += decreasing_lin(s.dist[i * N + j], c.threshold_near_other_agents_low, c.threshold_near_other_agents_high);
        float p = ssum * c.penalty_near_other_agents;
        near_other = p; has_near = true;
        rew += p; rew += pen_lane;
        if (c.rew_flags & SIGMAENV_REW_HAS_SPARSE) { rew += pca; rew += pcl; }
      }
    }
    float r = clampf(rew, -1.0f, 1.0f);
    g.reward[bi] = r;
    // RewardInfo.reset() at the top of every reward() zeroes all agents' entries except three fields
    // (helper_scenario.py:128-138): only the last agent's values of the other fields survive the loop.
    bool last = (i == N - 1);
    g.reward_info[1 * BN + bi] = last ? reward_goal : 0.0f;
    g.reward_info[7 * BN + bi] = last ? pca : 0.0f;
    g.reward_info[8 * BN + bi] = last ? pcl : 0.0f;
    g.reward_info[11 * BN + bi] = last ? r : 0.0f;
    if (has_near) g.reward_info[4 * BN + bi] = near_other;
    // update_state_after_rewarding: new short-term path (world_state_rt_sim.py:450-454)
    int path = s.path[i];
    float sp[NS * 2];
    short_term_path(m.center + (size_t)path * m.P * 2, m.n_center[path], m.is_loop[path] != 0, s.cp[i * 3], sp);
#pragma unroll
    for (int k = 0; k < NS * 2; ++k) { s.shrt[i * NS * 2 + k] = sp[k]; g.short_term[bi * NS * 2 + k] = sp[k]; }
    reinterpret_cast<float2*>(g.prev_pos)[bi] = make_float2(si[0], si[1]);  // state_buffer.add, road_traffic.py:1226-1240
    // distances / indices / flags of this agent
    g.dist_ref[bi] = s.dref[i];
#pragma unroll
    for (int q = 0; q < 5; ++q) { g.dist_left[bi * 5 + q] = s.dleft[i * 5 + q]; g.dist_right[bi * 5 + q] = s.dright[i * 5 + q]; }
    g.dist_bound[bi] = s.dbound[i];
    g.closest[bi * 3 + 0] = s.cp[i * 3 + 0]; g.closest[bi * 3 + 1] = s.cp[i * 3 + 1]; g.closest[bi * 3 + 2] = s.cp[i * 3 + 2];
  }
  __syncthreads();

  // ---- E: timer, counters, done(), reset requests (road_traffic.py:954-962,998-1002,1030-1035,1368-1487) --------
  if (tid == 0) {
    int step = g.timer[b * 4] + 1;
    int tries = 0, succ = 0, col_a = 0, col_l = 0;
    for (int i = 0; i < N; ++i) {
      int ca = s.flags[i * 4 + 3], cl = s.flags[i * 4 + 0], goal = s.flags[i * 4 + 2];
      succ += goal;
      tries += (ca | cl | goal) ? 1 : 0;
      col_a |= ca; col_l |= cl;
    }
    g.timer[b * 4] = step;
    g.timer[b * 4 + 1] += tries;
    g.timer[b * 4 + 2] += succ;
    int max_reached = (step == c.max_steps - 1);
    int done = c.is_testing_mode ? max_reached : (max_reached | col_a | col_l);
    g.done[b] = (uint8_t)done;
    for (int i = 0; i < N; ++i) {
      int rq = 0;
      if (c.is_testing_mode) rq = s.flags[i * 4 + 3] | s.flags[i * 4 + 0] | s.flags[i * 4 + 1] | s.flags[i * 4 + 2];
      else if (c.has_entry_exit) rq = s.flags[i * 4 + 1] | s.flags[i * 4 + 2];
      uchar4 f;
      f.x = (uint8_t)s.flags[i * 4 + 0]; f.y = (uint8_t)s.flags[i * 4 + 1]; f.z = (uint8_t)s.flags[i * 4 + 2];
      f.w = (uint8_t)((rq && !done) ? 1 : 0);
      reinterpret_cast<uchar4*>(g.col_flags)[bN + i] = f;
    }
  }

  // ---- D: observations --------------------------------------------------------------------------------------------
  observe_env(c, s, g, b, N, K, D);
}

// ---------------------------------------------------------------------------------------------------------------------
// observation only (observation() called again after resets): grid = n_envs
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline void load_env_for_observation(const Smem& s, const DevBufs& g, int b, int N) {
  const size_t bN = (size_t)b * N;
  for (int k = threadIdx.x; k < N * 8; k += blockDim.x) s.st[k] = g.state[bN * 8 + k];
  for (int k = threadIdx.x; k < N * 10; k += blockDim.x) s.vnew[k] = g.vertices[bN * 10 + k];
  for (int k = threadIdx.x; k < N * NS * 2; k += blockDim.x) s.shrt[k] = g.short_term[bN * NS * 2 + k];
  for (int k = threadIdx.x; k < N; k += blockDim.x) s.dref[k] = g.dist_ref[bN + k];
  for (int k = threadIdx.x; k < N * 5; k += blockDim.x) { s.dleft[k] = g.dist_left[bN * 5 + k]; s.dright[k] = g.dist_right[bN * 5 + k]; }
  for (int k = threadIdx.x; k < N * N; k += blockDim.x) s.dist[k] = g.dist_agents[bN * N + k];
  __syncthreads();
}

__global__ void __launch_bounds__(1024) sigmaenv_observe_kernel(sigmaenv_config_t c, DevBufs g) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int N = c.n_agents, K = c.n_nearing, D = 4 + 2 * NS + 11 * K;
  Smem s(smem_raw, N, K, D);
  load_env_for_observation(s, g, blockIdx.x, N);
  observe_env(c, s, g, blockIdx.x, N, K, D);
}

// ---------------------------------------------------------------------------------------------------------------------
// reset: (1) scatter host-chosen entries, (2) optional device-side sampling, (3) rebuild derived state of marked envs
// ---------------------------------------------------------------------------------------------------------------------
__global__ void sigmaenv_reset_scatter_kernel(DevBufs g, int N, int n, const int32_t* env_idx, const int32_t* agent_idx, const int32_t* path_ids,
                                              const float* state8, int full_env) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int b = env_idx[k], i = agent_idx[k];
  size_t bi = (size_t)b * N + i;
  for (int q = 0; q < 8; ++q) g.state[bi * 8 + q] = state8[(size_t)k * 8 + q];
  for (int q = 0; q < 4; ++q) g.path[bi * 4 + q] = path_ids[(size_t)k * 4 + q];
  atomicOr(&g.reset_mask[b], 1ull << i);
  if (full_env) g.reset_full[b] = 1;
}

// derived state of every marked agent (reset_init_distances_and_short_term_ref_path, world_state_rt.py:422-529) and the
// per-env tail (road_traffic.py:902-923); with_obs: also a fresh observation of the env.  Expects s.st / s.path / s.vnew of
// the env in LDS.  All threads of the block participate.
__device__ inline void reset_derive_body(const sigmaenv_config_t& c, const DevMap& m, const DevBufs& g, const Smem& s, int b, unsigned long long mask,
                                         int full, int with_obs) {
  const int N = c.n_agents, K = c.n_nearing, D = 4 + 2 * NS + 11 * K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
  const size_t bN = (size_t)b * N;
  for (int i = tid; i < N; i += blockDim.x) {
    if ((mask >> i) & 1ull) {
      float v[10];
      rect_vertices(c, s.st[i * 8], s.st[i * 8 + 1], s.st[i * 8 + 2], v);
#pragma unroll
      for (int k = 0; k < 10; ++k) { s.vnew[i * 10 + k] = v[k]; g.vertices[(bN + i) * 10 + k] = v[k]; }
    }
  }
  __syncthreads();
  for (int i = wave; i < N; i += n_waves) {
    if (!((mask >> i) & 1ull)) continue;
    AgentScan r;
    agent_scan_any<false>(m, c, s.path[i], lane, s.st[i * 8], s.st[i * 8 + 1], s.vnew + i * 10, s.vnew + i * 10, s.cp + i * 3, r);
    if (lane == 0) {
      const size_t bi = bN + i;
      float mb = INFINITY;
      g.dist_ref[bi] = r.d_ref;
#pragma unroll
      for (int q = 0; q < 5; ++q) { g.dist_left[bi * 5 + q] = r.dl[q]; g.dist_right[bi * 5 + q] = r.dr[q]; }
#pragma unroll
      for (int q = 0; q < 5; ++q) mb = fminf(mb, r.dl[q]);
#pragma unroll
      for (int q = 0; q < 5; ++q) mb = fminf(mb, r.dr[q]);
      g.dist_bound[bi] = mb;
      g.closest[bi * 3 + 0] = r.cp_ref; g.closest[bi * 3 + 1] = r.cp_l; g.closest[bi * 3 + 2] = r.cp_r;
      int path = s.path[i];
      float sp[NS * 2];
      short_term_path(m.center + (size_t)path * m.P * 2, m.n_center[path], m.is_loop[path] != 0, r.cp_ref, sp);
#pragma unroll
      for (int k = 0; k < NS * 2; ++k) g.short_term[bi * NS * 2 + k] = sp[k];
    }
  }
  // tail: mutual distances, collisions cleared, prev_pos := pos, timer
  const float diag = sqrtf(c.world_x_dim * c.world_x_dim + c.world_y_dim * c.world_y_dim);
  for (int p = tid; p < N * N; p += blockDim.x) {
    int i = p / N, j = p - i * N;
    float d = (i == j) ? diag : pair_distance(c, s.st, s.vnew, i, j);
    g.dist_agents[bN * N + p] = d;
    g.col_agents[bN * N + p] = 0;
  }
  for (int i = tid; i < N; i += blockDim.x) {
    reinterpret_cast<uchar4*>(g.col_flags)[bN + i] = make_uchar4(0, 0, 0, 0);
    reinterpret_cast<float2*>(g.prev_pos)[bN + i] = make_float2(s.st[i * 8], s.st[i * 8 + 1]);
    if (full) reinterpret_cast<float2*>(g.action)[bN + i] = make_float2(0.f, 0.f);
  }
  if (tid == 0) {
    if (full) { g.timer[b * 4] = 0; g.timer[b * 4 + 3] += 1; g.done[b] = 0; }
    g.reset_mask[b] = 0ull;
    g.reset_full[b] = 0;
  }
  if (with_obs) {
    __threadfence_block();
    __syncthreads();
    load_env_for_observation(s, g, b, N);
    observe_env(c, s, g, b, N, K, D);
  }
}

// host-driven resets: grid = n_envs, only blocks whose env has marked agents do work
__global__ void __launch_bounds__(1024) sigmaenv_reset_derive_kernel(sigmaenv_config_t c, DevMap m, DevBufs g, int with_obs) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int b = blockIdx.x;
  const unsigned long long mask = g.reset_mask[b];
  if (mask == 0ull) return;  // uniform per block
  const int N = c.n_agents, K = c.n_nearing, D = 4 + 2 * NS + 11 * K;
  const int full = g.reset_full[b];
  Smem s(smem_raw, N, K, D);
  const size_t bN = (size_t)b * N;
  for (int k = threadIdx.x; k < N * 8; k += blockDim.x) s.st[k] = g.state[bN * 8 + k];
  for (int k = threadIdx.x; k < N * 10; k += blockDim.x) s.vnew[k] = g.vertices[bN * 10 + k];
  for (int k = threadIdx.x; k < N; k += blockDim.x) {
    s.path[k] = g.path[(bN + k) * 4];
    int pt = g.path[(bN + k) * 4 + 3];  // the agent was placed at (or near) this centre-line point: guess for the pruned scan
    s.cp[k * 3 + 0] = pt; s.cp[k * 3 + 1] = pt; s.cp[k * 3 + 2] = pt;
  }
  __syncthreads();
  reset_derive_body(c, m, g, s, b, mask, full, with_obs);
}

// Device-side reset of finished envs (grid = n_envs, blocks of unfinished envs exit at once).  Wavefront 0 runs the rejection
// sampler of world_state_rt_sim.py:215-311 (non-testing mode) from a counter-based RNG: the 64 lanes evaluate tries 0..63 of one
// agent at once and the FIRST feasible try wins, which is exactly the sequential loop's result for the same draws (bounded to
// 64 tries; the reference loops without bound).  Then the deterministic reset as in sigmaenv_reset(full_env=1).
__global__ void __launch_bounds__(1024) sigmaenv_auto_reset_kernel(sigmaenv_config_t c, DevMap m, DevBufs g, uint64_t seed, uint64_t counter,
                                                                   int path_first, int path_count) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int b = blockIdx.x;
  if (!g.done[b]) return;  // uniform per block
  const int N = c.n_agents, K = c.n_nearing, D = 4 + 2 * NS + 11 * K;
  Smem s(smem_raw, N, K, D);
  const size_t bN = (size_t)b * N;
  const int tid = threadIdx.x;
  if (tid < 64) {
    const int t = tid;  // try index of this lane
    const float min_d = sqrtf((float)((double)c.length * (double)c.length + (double)c.width * (double)c.width)) * 1.5f;  // road_traffic.py:679-684
    const float min_d_sq = min_d * min_d;
    for (int i = 0; i < N; ++i) {
      int path = path_first + (int)(rng_u32(seed, counter, (uint32_t)b, (uint32_t)i, 2u * t) % (uint32_t)path_count);
      int n = m.n_center[path];
      int end = n / 2;
      if (end < 4) end = 4;
      int pt = 3 + (int)(rng_u32(seed, counter, (uint32_t)b, (uint32_t)i, 2u * t + 1u) % (uint32_t)(end - 3));
      float px = m.center[((size_t)path * m.P + pt) * 2];
      float py = m.center[((size_t)path * m.P + pt) * 2 + 1];
      bool ok = true;
      for (int j = 0; j < i; ++j) {  // accepted agents 0..i-1 are in LDS (written by the winning lane below)
        float dx = px - s.st[j * 8], dy = py - s.st[j * 8 + 1];
        float d2 = dx * dx + dy * dy;
        if (!(d2 >= min_d_sq)) ok = false;
      }
      unsigned long long feas = __ballot(ok);
      int win = feas ? (__ffsll((long long)feas) - 1) : (AUTO_RESET_MAX_TRIES - 1);
      if (t == win) {
        float u = (float)(rng_u32(seed, counter, (uint32_t)b, (uint32_t)i, 1000u) >> 8) * (1.0f / 16777216.0f);
        int yi = pt < m.yaw_stride ? pt : m.yaw_stride - 1;
        float rot = m.yaw[(size_t)path * m.yaw_stride + yi];
        float speed = u * c.max_speed;
        float st[8] = {px, py, rot, speed, 0.0f, speed * cr_cos(0.0f + rot), speed * cr_sin(0.0f + rot), 0.0f};
#pragma unroll
        for (int k = 0; k < 8; ++k) { s.st[i * 8 + k] = st[k]; g.state[(bN + i) * 8 + k] = st[k]; }
        s.path[i] = path;
        s.cp[i * 3 + 0] = pt; s.cp[i * 3 + 1] = pt; s.cp[i * 3 + 2] = pt;
        g.path[(bN + i) * 4 + 0] = path; g.path[(bN + i) * 4 + 1] = 0; g.path[(bN + i) * 4 + 2] = path - path_first; g.path[(bN + i) * 4 + 3] = pt;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  __threadfence_block();
  __syncthreads();
  const unsigned long long mask = (N >= 64) ? ~0ull : ((1ull << N) - 1ull);
  reset_derive_body(c, m, g, s, b, mask, 1, 1);
}

// =====================================================================================================================
// host side: the C-ABI
// =====================================================================================================================
struct sigmaenv {
  sigmaenv_config_t cfg;
  int device = 0;
  hipStream_t stream = nullptr;
  int B = 0, N = 0, K = 0, D = 0, P = 0, n_paths = 0;
  DevMap map{};
  DevBufs buf{};
  std::vector<void*> allocs;
  size_t smem_bytes = 0;
  int block = 256;
  void* bufs[SIGMAENV_BUF_COUNT] = {nullptr};
  size_t buf_bytes[SIGMAENV_BUF_COUNT] = {0};
  // reset staging
  int32_t *d_env_idx = nullptr, *d_agent_idx = nullptr, *d_path_ids = nullptr;
  float* d_state8 = nullptr;
  size_t staging_cap = 0;
  // step timing
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  std::vector<int> ev_used;
  bool timing = false;
  std::string err;
};

#define HIPCHK(h, call)                                                                              \
  do {                                                                                               \
    hipError_t e_ = (call);                                                                          \
    if (e_ != hipSuccess) {                                                                          \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                  \
      return SIGMAENV_EHIP;                                                                          \
    }                                                                                                \
  } while (0)

static int dev_alloc(sigmaenv* h, void** p, size_t bytes) {
  if (bytes == 0) bytes = 16;
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) { h->err = std::string("hipMalloc: ") + hipGetErrorString(e); return SIGMAENV_ENOMEM; }
  h->allocs.push_back(*p);
  e = hipMemsetAsync(*p, 0, bytes, h->stream);
  if (e != hipSuccess) { h->err = std::string("hipMemsetAsync: ") + hipGetErrorString(e); return SIGMAENV_EHIP; }
  return SIGMAENV_OK;
}

extern "C" int sigmaenv_obs_dim(int32_t n_nearing) { return 1 + 2 * NS + 3 + n_nearing * 11; }

extern "C" const char* sigmaenv_last_error(const sigmaenv_t* h) { return h ? h->err.c_str() : "null handle"; }

extern "C" void sigmaenv_destroy(sigmaenv_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  for (void* p : h->allocs) (void)hipFree(p);
  for (auto& ev : h->ev_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  delete h;
}

static int pick_block(int N, int B, int n_cu) {
  // one wavefront per agent where that still leaves enough workgroups to fill the chip, else fewer waves looping over agents
  int waves = N;
  if (waves > 16) waves = 16;
  while (waves > 4 && (long long)B * waves > (long long)n_cu * 32 * 4) waves >>= 1;
  if (waves < 1) waves = 1;
  // at least N threads are not required (loops stride by blockDim), but phase B1 likes >= 64
  return waves * 64;
}

extern "C" int sigmaenv_create(const sigmaenv_config_t* cfg, const sigmaenv_map_t* map, int device_id, void* hip_stream, sigmaenv_t** out) {
  if (!cfg || !map || !out) return SIGMAENV_EINVAL;
  *out = nullptr;
  if (cfg->abi_version != SIGMAENV_ABI_VERSION) return SIGMAENV_EINVAL;
  if (cfg->n_envs < 1 || cfg->n_agents < 1 || cfg->n_agents > SIGMAENV_MAX_AGENTS) return SIGMAENV_EINVAL;
  if (cfg->n_nearing < 0 || cfg->n_nearing > SIGMAENV_MAX_NEARING || cfg->n_nearing > cfg->n_agents - 1) return SIGMAENV_EINVAL;
  if (cfg->distance_type != SIGMAENV_DIST_C2C && cfg->distance_type != SIGMAENV_DIST_MTV) return SIGMAENV_EINVAL;
  if (map->n_paths < 1 || map->stride_points < 2) return SIGMAENV_EINVAL;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1 || device_id < 0 || device_id >= n_dev) return SIGMAENV_ENODEV;
  if (hipSetDevice(device_id) != hipSuccess) return SIGMAENV_ENODEV;
  sigmaenv* h = new sigmaenv();
  h->cfg = *cfg;
  h->device = device_id;
  h->stream = reinterpret_cast<hipStream_t>(hip_stream);
  const int B = h->B = cfg->n_envs, N = h->N = cfg->n_agents, K = h->K = cfg->n_nearing;
  h->D = sigmaenv_obs_dim(K);
  const int np = h->n_paths = map->n_paths, S = map->stride_points;
  for (int p = 0; p < np; ++p) {
    if (map->n_center[p] < 2 || map->n_left[p] < 2 || map->n_right[p] < 2 || map->n_center[p] > S || map->n_left[p] > S || map->n_right[p] > S) {
      delete h;
      return SIGMAENV_EINVAL;
    }
  }
  // padded path table, exactly as the reference pads its per-(env, agent) copies (world_state_rt.py:279-420)
  int maxc = 0;
  for (int p = 0; p < np; ++p) maxc = map->n_center[p] > maxc ? map->n_center[p] : maxc;
  int P = maxc + NS * 2 + 2;  // max_ref_path_points, road_traffic.py:520-530
  for (int p = 0; p < np; ++p) {
    if (map->n_left[p] > P) P = map->n_left[p];
    if (map->n_right[p] > P) P = map->n_right[p];
  }
  h->P = P;
  std::vector<float> hc((size_t)np * P * 2), hl((size_t)np * P * 2), hr((size_t)np * P * 2);
  for (int p = 0; p < np; ++p) {
    const int n = map->n_center[p], nl = map->n_left[p], nr = map->n_right[p];
    const float* c = map->center + (size_t)p * S * 2;
    float* dc = hc.data() + (size_t)p * P * 2;
    memcpy(dc, c, (size_t)n * 8);
    const float dirx = c[2 * (n - 1)] - c[2 * (n - 2)], diry = c[2 * (n - 1) + 1] - c[2 * (n - 2) + 1];
    const int ne = NS * 2;
    for (int k = 1; k <= ne; ++k) {  // _extend_map_related_ref_path :279-293
      volatile float kx = (float)k * dirx, ky = (float)k * diry;  // product rounded to fp32 before the add (no contraction)
      dc[2 * (n + k - 1)] = c[2 * (n - 1)] + kx;
      dc[2 * (n + k - 1) + 1] = c[2 * (n - 1) + 1] + ky;
    }
    for (int k = n + ne; k < P; ++k) { dc[2 * k] = dc[2 * (n + ne - 1)]; dc[2 * k + 1] = dc[2 * (n + ne - 1) + 1]; }
    const float* l = map->left + (size_t)p * S * 2;
    float* dl = hl.data() + (size_t)p * P * 2;
    memcpy(dl, l, (size_t)nl * 8);
    for (int k = nl; k < P; ++k) { dl[2 * k] = l[2 * (nl - 1)]; dl[2 * k + 1] = l[2 * (nl - 1) + 1]; }
    const float* r = map->right + (size_t)p * S * 2;
    float* dr = hr.data() + (size_t)p * P * 2;
    memcpy(dr, r, (size_t)nr * 8);
    for (int k = nr; k < P; ++k) { dr[2 * k] = r[2 * (nr - 1)]; dr[2 * k + 1] = r[2 * (nr - 1) + 1]; }
  }
  int rc;
#define ALLOC(ptr, bytes)                                         \
  if ((rc = dev_alloc(h, (void**)&(ptr), (bytes))) != SIGMAENV_OK) { \
    std::string e = h->err;                                       \
    sigmaenv_destroy(h);                                          \
    fprintf(stderr, "sigmaenv_create: %s\n", e.c_str());        \
    return rc;                                                    \
  }
  float *d_c, *d_l, *d_r, *d_y;
  int32_t *d_nc, *d_nl, *d_nr;
  uint8_t* d_loop;
  ALLOC(d_c, hc.size() * 4); ALLOC(d_l, hl.size() * 4); ALLOC(d_r, hr.size() * 4); ALLOC(d_y, (size_t)np * S * 4);
  ALLOC(d_nc, (size_t)np * 4); ALLOC(d_nl, (size_t)np * 4); ALLOC(d_nr, (size_t)np * 4); ALLOC(d_loop, (size_t)np);
#define H2D(dst, src, bytes)                                                                     \
  if (hipMemcpyAsync((dst), (src), (bytes), hipMemcpyHostToDevice, h->stream) != hipSuccess) {   \
    sigmaenv_destroy(h);                                                                         \
    return SIGMAENV_EHIP;                                                                        \
  }
  H2D(d_c, hc.data(), hc.size() * 4); H2D(d_l, hl.data(), hl.size() * 4); H2D(d_r, hr.data(), hr.size() * 4);
  H2D(d_y, map->yaw, (size_t)np * S * 4);
  H2D(d_nc, map->n_center, (size_t)np * 4); H2D(d_nl, map->n_left, (size_t)np * 4); H2D(d_nr, map->n_right, (size_t)np * 4);
  H2D(d_loop, map->is_loop, (size_t)np);
  // pruning table: bounding boxes of runs of SIGMAENV_CHUNK consecutive real segments (points 8c .. min(8c+8, n-1))
  int nch = (P - 1 + SIGMAENV_CHUNK - 1) / SIGMAENV_CHUNK;
  bool prune = nch <= 32;  // the candidate masks are 32-bit; longer polylines fall back to the full scan
  if (const char* e = getenv("SIGMAENV_PRUNE")) prune = prune && atoi(e) != 0;
  float4* d_box = nullptr;
  std::vector<float4> hb;  // must outlive the asynchronous upload below
  if (prune) {
    hb.assign((size_t)np * 3 * nch, make_float4(1e30f, 1e30f, -1e30f, -1e30f));
    for (int p = 0; p < np; ++p) {
      const float* polys[3] = {hc.data() + (size_t)p * P * 2, hl.data() + (size_t)p * P * 2, hr.data() + (size_t)p * P * 2};
      const int cnt[3] = {map->n_center[p], map->n_left[p], map->n_right[p]};
      for (int q = 0; q < 3; ++q) {
        for (int k = 0; k + 1 < cnt[q]; ++k) {
          float4& bx = hb[((size_t)p * 3 + q) * nch + k / SIGMAENV_CHUNK];
          for (int e2 = 0; e2 < 2; ++e2) {
            float x = polys[q][2 * (k + e2)], y = polys[q][2 * (k + e2) + 1];
            bx.x = x < bx.x ? x : bx.x; bx.y = y < bx.y ? y : bx.y; bx.z = x > bx.z ? x : bx.z; bx.w = y > bx.w ? y : bx.w;
          }
        }
      }
    }
    ALLOC(d_box, hb.size() * sizeof(float4));
    H2D(d_box, hb.data(), hb.size() * sizeof(float4));
  }
  h->map = DevMap{d_c, d_l, d_r, d_y, d_nc, d_nl, d_nr, d_loop, P, np, S, d_box, prune ? nch : 0};
  const size_t BN = (size_t)B * N;
  DevBufs& g = h->buf;
  struct Spec { int id; void** p; size_t bytes; };
  Spec specs[] = {
      {SIGMAENV_BUF_STATE, (void**)&g.state, BN * 32}, {SIGMAENV_BUF_PREV_POS, (void**)&g.prev_pos, BN * 8},
      {SIGMAENV_BUF_VERTICES, (void**)&g.vertices, BN * 40}, {SIGMAENV_BUF_PATH, (void**)&g.path, BN * 16},
      {SIGMAENV_BUF_SHORT_TERM, (void**)&g.short_term, BN * NS * 8}, {SIGMAENV_BUF_DIST_REF, (void**)&g.dist_ref, BN * 4},
      {SIGMAENV_BUF_DIST_LEFT, (void**)&g.dist_left, BN * 20}, {SIGMAENV_BUF_DIST_RIGHT, (void**)&g.dist_right, BN * 20},
      {SIGMAENV_BUF_DIST_BOUND, (void**)&g.dist_bound, BN * 4}, {SIGMAENV_BUF_CLOSEST, (void**)&g.closest, BN * 12},
      {SIGMAENV_BUF_DIST_AGENTS, (void**)&g.dist_agents, BN * N * 4}, {SIGMAENV_BUF_COL_AGENTS, (void**)&g.col_agents, BN * N},
      {SIGMAENV_BUF_COL_FLAGS, (void**)&g.col_flags, BN * 4}, {SIGMAENV_BUF_REWARD, (void**)&g.reward, BN * 4},
      {SIGMAENV_BUF_REWARD_INFO, (void**)&g.reward_info, BN * SIGMAENV_N_REWARD_INFO * 4}, {SIGMAENV_BUF_OBS, (void**)&g.obs, BN * h->D * 4},
      {SIGMAENV_BUF_NEARING, (void**)&g.nearing, BN * K * 4}, {SIGMAENV_BUF_DONE, (void**)&g.done, (size_t)B},
      {SIGMAENV_BUF_TIMER, (void**)&g.timer, (size_t)B * 16}, {SIGMAENV_BUF_ACTION, (void**)&g.action, BN * 8},
  };
  for (auto& sp : specs) {
    ALLOC(*sp.p, sp.bytes);
    h->bufs[sp.id] = *sp.p;
    h->buf_bytes[sp.id] = sp.bytes;
  }
  ALLOC(g.reset_mask, (size_t)B * 8);
  ALLOC(g.reset_full, (size_t)B);
#undef ALLOC
#undef H2D
  hipDeviceProp_t prop;
  int n_cu = 256;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) n_cu = prop.multiProcessorCount;
  h->block = pick_block(N, B, n_cu);
  if (const char* e = getenv("SIGMAENV_BLOCK")) {
    int v = atoi(e);
    if (v >= 64 && v <= 1024 && v % 64 == 0) h->block = v;
  }
  h->smem_bytes = Smem::bytes(N, K, h->D);
  if (h->smem_bytes > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem_bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_observe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem_bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_reset_derive_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem_bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_auto_reset_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem_bytes);
  }
  if (hipStreamSynchronize(h->stream) != hipSuccess) { sigmaenv_destroy(h); return SIGMAENV_EHIP; }
  *out = h;
  return SIGMAENV_OK;
}

static int launch_derive(sigmaenv* h, int with_obs) {
  hipLaunchKernelGGL(sigmaenv_reset_derive_kernel, dim3(h->B), dim3(h->block), h->smem_bytes, h->stream, h->cfg, h->map, h->buf, with_obs);
  HIPCHK(h, hipGetLastError());
  return SIGMAENV_OK;
}

extern "C" int sigmaenv_reset(sigmaenv_t* h, int32_t n, const int32_t* env_idx, const int32_t* agent_idx, const int32_t* path_ids,
                              const float* state8, int32_t full_env) {
  if (!h || n < 0) return SIGMAENV_EINVAL;
  if (n == 0) return SIGMAENV_OK;
  if (!env_idx || !agent_idx || !path_ids || !state8) return SIGMAENV_EINVAL;
  for (int k = 0; k < n; ++k) {
    if (env_idx[k] < 0 || env_idx[k] >= h->B || agent_idx[k] < 0 || agent_idx[k] >= h->N || path_ids[4 * (size_t)k] < 0 ||
        path_ids[4 * (size_t)k] >= h->n_paths) {
      h->err = "reset entry " + std::to_string(k) + " out of range";
      return SIGMAENV_EINVAL;
    }
  }
  HIPCHK(h, hipSetDevice(h->device));
  if ((size_t)n > h->staging_cap) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    size_t cap = (size_t)n * 2;
    void* p;
    int rc;
    if ((rc = dev_alloc(h, &p, cap * 4)) != 0) return rc;
    h->d_env_idx = (int32_t*)p;
    if ((rc = dev_alloc(h, &p, cap * 4)) != 0) return rc;
    h->d_agent_idx = (int32_t*)p;
    if ((rc = dev_alloc(h, &p, cap * 16)) != 0) return rc;
    h->d_path_ids = (int32_t*)p;
    if ((rc = dev_alloc(h, &p, cap * 32)) != 0) return rc;
    h->d_state8 = (float*)p;
    h->staging_cap = cap;
  }
  HIPCHK(h, hipMemcpyAsync(h->d_env_idx, env_idx, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_agent_idx, agent_idx, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_path_ids, path_ids, (size_t)n * 16, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_state8, state8, (size_t)n * 32, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(sigmaenv_reset_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->buf, h->N, n, h->d_env_idx, h->d_agent_idx,
                     h->d_path_ids, h->d_state8, (int)full_env);
  HIPCHK(h, hipGetLastError());
  int rc = launch_derive(h, 0);
  if (rc) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));  // host inputs may be reused by the caller once we return
  return SIGMAENV_OK;
}

extern "C" int sigmaenv_step(sigmaenv_t* h, const float* actions) {
  if (!h || !actions) return SIGMAENV_EINVAL;
  int slot = -1;
  if (h->timing) {
    slot = (int)h->ev_used.size();
    if (slot >= (int)h->ev_pool.size()) {
      hipEvent_t a, b;
      HIPCHK(h, hipEventCreate(&a));
      HIPCHK(h, hipEventCreate(&b));
      h->ev_pool.emplace_back(a, b);
    }
    h->ev_used.push_back(slot);
    HIPCHK(h, hipEventRecord(h->ev_pool[slot].first, h->stream));
  }
  hipLaunchKernelGGL(sigmaenv_step_kernel, dim3(h->B), dim3(h->block), h->smem_bytes, h->stream, h->cfg, h->map, h->buf, actions);
  HIPCHK(h, hipGetLastError());
  if (slot >= 0) HIPCHK(h, hipEventRecord(h->ev_pool[slot].second, h->stream));
  return SIGMAENV_OK;
}

extern "C" int sigmaenv_observe(sigmaenv_t* h) {
  if (!h) return SIGMAENV_EINVAL;
  hipLaunchKernelGGL(sigmaenv_observe_kernel, dim3(h->B), dim3(h->block), h->smem_bytes, h->stream, h->cfg, h->buf);
  HIPCHK(h, hipGetLastError());
  return SIGMAENV_OK;
}

extern "C" int sigmaenv_auto_reset(sigmaenv_t* h, uint64_t seed, uint64_t counter, int32_t path_first, int32_t path_count) {
  if (!h || path_first < 0 || path_count < 1 || path_first + path_count > h->n_paths) return SIGMAENV_EINVAL;
  hipLaunchKernelGGL(sigmaenv_auto_reset_kernel, dim3(h->B), dim3(h->block), h->smem_bytes, h->stream, h->cfg, h->map, h->buf, seed, counter,
                     (int)path_first, (int)path_count);
  HIPCHK(h, hipGetLastError());
  return SIGMAENV_OK;
}

extern "C" int sigmaenv_get(sigmaenv_t* h, sigmaenv_buf_t which, void** dev_ptr, size_t* bytes) {
  if (!h || !dev_ptr || !bytes || (int)which < 0 || (int)which >= SIGMAENV_BUF_COUNT) return SIGMAENV_EINVAL;
  *dev_ptr = h->bufs[which];
  *bytes = h->buf_bytes[which];
  return SIGMAENV_OK;
}

extern "C" int sigmaenv_sync(sigmaenv_t* h) {
  if (!h) return SIGMAENV_EINVAL;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return SIGMAENV_OK;
}

// Enables HIP-event bracketing of every subsequent step launch (first call) and reports/clears the collected launches.
extern "C" int sigmaenv_step_time_ms(sigmaenv_t* h, double* avg_ms, int32_t* n_launches) {
  if (!h || !avg_ms || !n_launches) return SIGMAENV_EINVAL;
  *avg_ms = 0.0;
  *n_launches = 0;
  if (!h->timing) { h->timing = true; return SIGMAENV_OK; }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  double total = 0.0;
  for (int slot : h->ev_used) {
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev_pool[slot].first, h->ev_pool[slot].second));
    total += ms;
  }
  *n_launches = (int32_t)h->ev_used.size();
  if (*n_launches) *avg_ms = total / *n_launches;
  h->ev_used.clear();
  return SIGMAENV_OK;
}
