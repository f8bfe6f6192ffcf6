// sigmaenv.hip -- fused vectorized multi-agent CAV environment step for MI355X (gfx950) + its C-ABI (include/sigmaenv.h).
//
// The step kernel (sigmaenv_step_wave.inc) gives every WAVEFRONT its own tile of G envs x N agents (G * N <= 64 "slots", one env per
// wavefront at 16 agents) and never synchronises wavefronts with each other; one launch per step:
//   A  two lanes per agent : action clamp + kinematic bicycle Euler step, new rectangle vertices                  (K1, K2)
//   S  lane per work item  : the 11 point->polyline queries per agent as a balanced list of (agent, polyline, chunk of 4 segments)
//                            items -- only chunks that survive bounding-box pruning -- merged by LDS atomics        (K3)
//   E  lane per (agent, near segment, edge) : rectangle->boundary collision tests                                 (K5)
//   B1 lane per pair       : mutual distances (c2c / mtv) and rectangle-rectangle collision masks                  (K4, K5)
//   C  lane per agent      : reward terms, short-term reference path; per-env done / counters by wavefront ballots (K7, K6, K9)
//   D  lane per item       : top-k nearest agents, ego-view transforms, observation rows staged in LDS, rollout record row (K8)
//   R  the tile's wavefront: device-side reset of finished envs / re-placement of agents with a pending request
// Per-env agent poses / vertices / distance rows live in LDS between the phases; HBM sees each state word once in, once out.
// This file holds the shared device code (full-scan fallback, candidate masks, observation, resets), the workgroup kernels of the
// stand-alone entry points (reset, observe, auto-reset, start table) and the C-ABI.
// MFMA is unused on purpose in the step: there is no dense contraction in this path.
//
// Reference citations are relative to /root/reference/sigmarl; see sigmaenv_device.h for the arithmetic contract.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "sigmaenv_device.h"

using namespace sigmadev;

// Phase stamps and ablation switches exist in the profile build only (make prof -> libsigmaenv_prof.so, used by tools/): the product
// library carries no debug branch, no stamp and reads no debug environment variable.
#ifdef SIGMAENV_PROFILE
#define PROF_TS2(g, tid, k) do { if ((g).dbg_ts2 && (tid) == 0) (g).dbg_ts2[(size_t)blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PROF_TS2(g, tid, k) do { } while (0)
#endif

// weighting_ref_directions = linspace(1, 0.2, n_points_short_term) / sum (road_traffic.py:536-543); bit patterns of the reference's tensor (torch CPU)
#include "../../include/sigmaenv_ref_weights.h"
static_assert(NS >= 1 && NS <= SIGMAENV_MAX_SHORT_TERM, "SIGMAENV_N_SHORT_TERM out of range");
__device__ __constant__ uint32_t W_REF_BITS[NS] = SIGMAENV_W_REF_BITS;

#define AUTO_RESET_MAX_TRIES 64

// ---------------------------------------------------------------------------------------------------------------------
// LDS carve-up for one workgroup: S = G*N agent slots (slot = env_local * N + agent)
// ---------------------------------------------------------------------------------------------------------------------
#define DIST_STRIDE(N) ((N) + 1)   /* odd float stride: lane-per-row column walks are bank-conflict free */
#define COL_STRIDE(N) ((N) + 4)    /* byte stride whose word stride ((N+4)/4) is odd for N = 16 */
#define ESCR_BYTES (8 * 4 * 4 * 16) /* up to 8 wavefronts x 4 lane groups x 4 segments x float4 */
#define CAND_LIST 16               /* candidate chunks listed per (agent, polyline); longer masks fall back to bit counting */
#define ITEM_CAP(S) ((S) * 24 > 64 ? (S) * 24 : 64)  /* candidate chunks of a tile listed per round of the balanced scan; never below 64: ONE task can have up to 64 candidate
                                                        chunks (the masks are 64-bit) and must fit the list alone, else a round could list nothing and the scan would spin */
#define NEAR_CAP 8                 /* boundary segments within the circumradius listed per (agent, side); more are tested in place */
struct Smem {
  float *st, *vold, *vnew, *shrt, *dref, *dleft, *dright, *dbound, *dist, *obs, *thr, *cs, *rew;
  float* carry;  // [S][3] tan(steering), cos / sin(yaw + sideslip) of the slot's CURRENT state: what the next bicycle step of the same launch starts from (phase A)
  int *path, *cp, *near, *flags, *npts;
  unsigned long long* cmask;  // candidate-chunk masks of the centre / left / right scan, [S][3]            (block layout only)
  uint8_t* cand;              // the first CAND_LIST set bits of every mask as a list of chunk indices, [S][3][CAND_LIST]   (block layout only)
  uint16_t* pidx;             // (i, j), i < j, of the u-th unordered agent pair of an env: i | j << 8, [N (N - 1) / 2]
  float4* escr;               // B2 staging of the segments close enough to hit the rectangle, [MAX_WAVES][4 lane groups][4]  (block layout only)
  uint8_t* col;
  // wave layout (step kernel): the balanced scan's work list and per-task accumulators
  uint16_t* items;            // (task (pl-major) << 6 | chunk) of every candidate chunk of the tile, [ITEM_CAP(S)]
  unsigned long long* acc64;  // per task: (bits of the minimal centre-point distance) << 32 | its lowest segment index, [S * 3]
  uint32_t* acc32;            // per boundary task: bits of the four minimal SQUARED corner distances, [S * 2][4]
  int* nearn;                 // per boundary task: number of segments within the circumradius, [S * 2]
  uint8_t* nearl;             // indices of the boundary segments within the circumradius, [S * 2][NEAR_CAP]
  uint8_t* fresh;             // [S] 1 = the agent was (re)placed and has not been stepped since (boundary points of the observation: another index shift,
                              // world_state_rt.py:531-576 vs :686-724); the tile's copy of DevBufs::fresh, kept across the steps of one launch
  unsigned long long* key64;  // [S] scratch of the observation's lanelet search (min over (squared distance bits << 32 | lanelet))
  // LEAN = the wave-per-tile layout of the step kernel: no escr / cmask / cand (its scan keeps them in registers), near-segment lists instead
  // floats of the staging region: the observation rows -- and, in the LEAN layout, the scan's work areas IN THE SAME BYTES: the work list and the per-task
  // accumulators live from phase S to phase E of a step, the observation rows (and the reward phase's use of the region as scratch) from phase C to the end of the
  // step, and the next step's scan initialises its areas again.  2 KB per 16-agent tile: the difference between 16 and 20 resident tiles per CU, i.e. room for the
  // wider rows of the non-default observation switches without dropping below the 16 tiles per CU (one resident round at 4096 envs) the default row runs at.
  __host__ __device__ static __forceinline__ size_t scan_scratch_ints(int S) {
    return (size_t)S * 3 * 2 + (size_t)S * 2 * 4 + (size_t)S * 2 + (size_t)(ITEM_CAP(S) + 1) / 2 + (size_t)(S * 2 * NEAR_CAP + 3) / 4;
  }
  __host__ __device__ static __forceinline__ size_t stage_floats(int S, int D, bool lean) {
    const size_t o = ((size_t)S * D + 3) & ~(size_t)3, q = (scan_scratch_ints(S) + 3) & ~(size_t)3;
    return (lean && q > o) ? q : o;
  }
  __device__ __forceinline__ Smem(char* base, int S, int N, int K, int D, bool lean = false) {
    escr = reinterpret_cast<float4*>(base);  // first: the dynamic LDS base is 16-byte aligned
    float* f = reinterpret_cast<float*>(base + (lean ? 0 : ESCR_BYTES));
    obs = f;  // 16-byte aligned (vector copy to HBM), and so is everything up to vold
    {
      int* q = reinterpret_cast<int*>(f);
      acc64 = reinterpret_cast<unsigned long long*>(q); q += S * 3 * 2;
      acc32 = reinterpret_cast<uint32_t*>(q); q += S * 2 * 4;
      nearn = q; q += S * 2;
      items = reinterpret_cast<uint16_t*>(q); q += (ITEM_CAP(S) + 1) / 2;
      nearl = reinterpret_cast<uint8_t*>(q);  // (only meaningful in the LEAN layout)
    }
    f += stage_floats(S, D, lean);
    st = f; f += S * 8;
    vold = f; f += S * 10;
    vnew = f; f += S * 10;
    shrt = f; f += S * NS * 2;
    dref = f; f += S;
    dleft = f; f += S * 5;
    dright = f; f += S * 5;
    dbound = f; f += S;
    dist = f; f += S * DIST_STRIDE(N);
    thr = f; f += S * 3;   // step kernel: last step's position (x, y) of the slot, loaded early for the reward phase
    cs = f; f += S * 2;    // cos / sin of the yaw (shared by the vertices and the ego-view transforms)
    rew = f; f += S * 2;   // reward per slot, then done flag per env (rollout slab record)
    carry = f; f += S * 3;
    int* i = reinterpret_cast<int*>(f);
    path = i; i += S;
    cp = i; i += S * 3;
    near = i; i += S * (K > 0 ? K : 1);
    flags = i; i += S * 4;
    npts = i; i += S * 3;  // point counts of the agent's centre line / left / right boundary
    i += ((reinterpret_cast<uintptr_t>(i) >> 2) & 1);  // 8-byte alignment of the 64-bit words that follow (the base is 16-byte aligned)
    cmask = reinterpret_cast<unsigned long long*>(i);
    if (!lean) i += S * 3 * 2;
    key64 = lean ? reinterpret_cast<unsigned long long*>(i) : cmask;  // LEAN: its own S words (the observation phase cannot borrow the scan's: they share its region)
    if (lean) i += S * 2;
    cand = reinterpret_cast<uint8_t*>(i);
    if (!lean) i += S * 3 * (CAND_LIST / 4);
    pidx = reinterpret_cast<uint16_t*>(i); i += (N * (N - 1) / 2 + 1) / 2;
    col = reinterpret_cast<uint8_t*>(i);
    fresh = col + (size_t)S * COL_STRIDE(N);
  }
  __host__ __device__ static __forceinline__ size_t bytes(int S, int N, int K, int D, bool lean = false) {
    size_t f = stage_floats(S, D, lean) + (size_t)S * 8 + S * 10 * 2 + S * NS * 2 + S + S * 5 * 2 + S + (size_t)S * DIST_STRIDE(N) + S * 3 + S * 2 + S * 2 + S * 3;
    size_t i = (size_t)S + S * 3 + S * (K > 0 ? K : 1) + S * 4 + S * 3 + 1;  // (+ 1: the alignment word)
    i += lean ? (size_t)S * 2 : ((size_t)S * 3 * 2 + (size_t)S * 3 * (CAND_LIST / 4));
    return (lean ? 0 : ESCR_BYTES) + (f + i + (size_t)(N * (N - 1) / 2 + 1) / 2) * 4 + (size_t)S * COL_STRIDE(N) + (((size_t)S + 3) & ~(size_t)3) + 16;
  }
};

// The thread group that owns a tile: a whole workgroup (reset / observe / start-table kernels) or ONE wavefront (the step kernel, whose
// wavefronts never synchronise with each other).  Within a wavefront LDS operations execute in program order; wave_sync only stops the
// compiler from moving LDS accesses across the point where other lanes' data is consumed.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <bool WAVE>
struct Grp {
  // (wavefront groups: the lane index passes through an empty asm, so that inside the step kernel's step loop nothing derived from it is a loop
  // invariant the compiler would keep in registers across the whole loop body -- see sigmaenv_step_wave_kernel)
  __device__ static __forceinline__ int tid() {
    if (!WAVE) return (int)threadIdx.x;
    int x = (int)(threadIdx.x & 63);
    asm volatile("" : "+v"(x));
    __builtin_assume(x >= 0 && x < 64);
    return x;
  }
  __device__ static __forceinline__ int size() { return WAVE ? 64 : (int)blockDim.x; }
  __device__ static __forceinline__ void sync() { if (WAVE) wave_sync(); else __syncthreads(); }
};

// Row layout of the observation for obs_flags != 0 (_observe_self / _observe_other_agents, observation_provider_rt.py:803-925; include/sigmaenv.h):
//   [own]     bird: position 2, rotation 1, velocity 2 | ego: speed 1;  steering?;  short-term path 2 NS;  distance to the centre line?;  boundary distances 2 | points 20
//   [other k] vertices 8 | position 2, rotation, length, width;  velocity 2;  steering?;  distance?;  its short-term path 2 NS?
//   placeholders 2 K?
// SIGMAENV_OBS_FULL (is_partial_observation == False, :756-851; bird view): [own] as above, then the features of ALL N agents, every feature's flat per-env array cut
// into K chunks and the chunks interleaved (see observe_tile).  Plain integers: the same on the host (sizes) and in the kernels.
struct ObsLayout {
  int F, bird, full, p_steer, p_short, p_dcen, p_bnd, n_bnd_pts, own_w, oth_w, q_vel, q_steer, q_dist, q_ref, n_oth_pts, T1, pad, D, DL;
  int W_oth;  // FULL: total width of the [others] part per env (N x the sum of the feature widths)
  enum { F_VERT, F_POS, F_ROT, F_LEN, F_WID, F_VEL, F_STEER, F_DIST, F_REF };
  // FULL: the idx-th feature of the [others] part in row order (no arrays: the struct lives in registers)
  __host__ __device__ __forceinline__ bool feature(int idx, int N, int& kind, int& wid) const {
    int n = 0;
#define SIGMA_FEAT(cond, k_, w_) if (cond) { if (idx == n) { kind = (k_); wid = (w_); return true; } ++n; }
    const bool nv = (F & SIGMAENV_OBS_NO_VERTICES) != 0;
    SIGMA_FEAT(!nv, F_VERT, 8)
    SIGMA_FEAT(nv, F_POS, 2) SIGMA_FEAT(nv, F_ROT, 1) SIGMA_FEAT(nv, F_LEN, 1) SIGMA_FEAT(nv, F_WID, 1)
    SIGMA_FEAT(true, F_VEL, 2)
    SIGMA_FEAT((F & SIGMAENV_OBS_STEERING) != 0, F_STEER, 1)
    SIGMA_FEAT(!(F & SIGMAENV_OBS_NO_DIST_AGENTS), F_DIST, N)
    SIGMA_FEAT((F & SIGMAENV_OBS_REF_OTHERS) != 0, F_REF, 2 * NS)
#undef SIGMA_FEAT
    return false;
  }
  __host__ __device__ __forceinline__ ObsLayout(int F_, int N, int K) {
    F = F_;
    bird = (F & SIGMAENV_OBS_BIRD_VIEW) != 0;
    full = (F & SIGMAENV_OBS_FULL) != 0;
    int p = bird ? 5 : 1;
    p_steer = (F & SIGMAENV_OBS_STEERING) ? p++ : -1;
    p_short = p; p += 2 * NS;
    p_dcen = (F & SIGMAENV_OBS_NO_DIST_CENTER) ? -1 : p++;
    p_bnd = p; n_bnd_pts = (F & SIGMAENV_OBS_BOUNDARY_POINTS) ? 10 : 0; p += n_bnd_pts ? 20 : 2;
    own_w = p;
    int q = (F & SIGMAENV_OBS_NO_VERTICES) ? 5 : 8;
    q_vel = q; q += 2;
    q_steer = (F & SIGMAENV_OBS_STEERING) ? q++ : -1;
    q_dist = (F & SIGMAENV_OBS_NO_DIST_AGENTS) ? -1 : q++;
    q_ref = (F & SIGMAENV_OBS_REF_OTHERS) ? q : -1;
    if (q_ref >= 0) q += 2 * NS;
    oth_w = q;
    n_oth_pts = ((F & SIGMAENV_OBS_NO_VERTICES) ? 1 : 4) + ((F & SIGMAENV_OBS_REF_OTHERS) ? NS : 0);
    pad = (F & SIGMAENV_OBS_OPPONENT_PAD) ? 2 * K : 0;
    W_oth = 0;
    if (full) {
      const int wsum = ((F & SIGMAENV_OBS_NO_VERTICES) ? 5 : 8) + 2 + ((F & SIGMAENV_OBS_STEERING) ? 1 : 0) + ((F & SIGMAENV_OBS_NO_DIST_AGENTS) ? 0 : N) +
                       ((F & SIGMAENV_OBS_REF_OTHERS) ? 2 * NS : 0);
      W_oth = N * wsum;
      T1 = NS + n_bnd_pts;                    // only the [own] points are transformed per agent
      D = own_w + W_oth + pad;
      DL = own_w + (W_oth + N - 1) / N;       // LDS staging: the [own] blocks per agent + ONE copy of the [others] part per env
    } else {
      T1 = NS + n_bnd_pts + K * n_oth_pts;
      D = own_w + K * oth_w + pad;
      DL = D;
    }
  }
};

// what a workgroup (or, in the step kernel, a wavefront) covers
struct Tile {
  int env0, nenv, slots, N, K, D, DL;  // D: width of the observation row; DL: floats per agent slot of its LDS staging area (= D unless SIGMAENV_OBS_FULL)
  size_t a0;  // global agent index of slot 0 (= env0 * N)
  __device__ __forceinline__ Tile(const sigmaenv_config_t& c, int G, int tile_index = -1) {
    N = c.n_agents; K = c.n_nearing; D = 4 + 2 * NS + 11 * K; DL = D;
    if (c.obs_flags != 0) { const ObsLayout L(c.obs_flags, N, K); D = L.D; DL = L.DL; }
    env0 = (tile_index < 0 ? (int)blockIdx.x : tile_index) * G;
    nenv = min(G, c.n_envs - env0);
    slots = nenv * N;
    a0 = (size_t)env0 * N;
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

// full scan of one boundary (fallback when the polyline is too long for the 32-bit candidate masks, and A/B baseline)
template <bool COLLIDE>
__device__ __forceinline__ void boundary_scan(const float* __restrict__ poly, int n, int lane, float cgx, float cgy, const float* qv,
                                              const Edge* e, float d_out[5], int& cp_out, bool& hit) {
  float bd0 = INFINITY, bs[4];
  int bk = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) bs[q] = INFINITY;
  bool h = false;
  const float2* p2 = reinterpret_cast<const float2*>(poly);
  for (int k = lane; k + 1 < n; k += 64) {
    float2 a = p2[k], b = p2[k + 1];
    float lx = b.x - a.x, ly = b.y - a.y;
    float len2 = lx * lx + ly * ly;
    float d0 = point_segment(cgx, cgy, a.x, a.y, lx, ly, len2);
    if (d0 < bd0) { bd0 = d0; bk = k; }
#pragma unroll
    for (int q = 0; q < 4; ++q) bs[q] = fminf(bs[q], point_segment_sq(qv[2 * q], qv[2 * q + 1], a.x, a.y, lx, ly, len2));
    if (COLLIDE) {
      float S2 = lx * a.y - ly * a.x;
#pragma unroll
      for (int i = 0; i < 4; ++i) h |= edge_hits_segment(e[i], a.x, a.y, b.x, b.y, lx, ly, S2);
    }
  }
  wave_argmin(bd0, bk);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) bs[q] = fminf(bs[q], __shfl_xor(bs[q], off, 64));
  }
  d_out[0] = bd0;
#pragma unroll
  for (int q = 0; q < 4; ++q) d_out[q + 1] = sqrtf(bs[q]);
  cp_out = bk + 1;
  if (COLLIDE) hit = hit || (__ballot(h) != 0ull);
}

template <bool COLLIDE>
__device__ inline void agent_scan_full(const DevMap& m, const sigmaenv_config_t& c, int path, int lane, float cgx, float cgy, const float* qv,
                                       const float* ev, AgentScan& r) {
  const float* ctr = m.center + (size_t)path * m.P * 2;
  const float* lb = m.left + (size_t)path * m.P * 2;
  const float* rb = m.right + (size_t)path * m.P * 2;
  int n = m.n_center[path], nl = m.n_left[path], nr = m.n_right[path];
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
// are evaluated.  Exactness argument (DESIGN.md "Pruned scan"): a run of SIGMAENV_CHUNK consecutive segments is skipped only
// if the distance from the agent's centre to the run's bounding box exceeds T, where T bounds (with a 1e-4 m margin, >> fp32
// error)
//   * the distance of every query point to the segment that was closest last step (an upper bound of its minimum), and
//   * the rectangle's circumradius (a segment farther than that cannot intersect an edge).
// So every segment whose computed distance can equal or beat the running minimum, and every segment that can hit the
// rectangle, is still evaluated with the very same arithmetic; ties still resolve to the lowest index.
// ---------------------------------------------------------------------------------------------------------------------
// exact radius of the corner-query points around the centre (used for agent 0, whose query points are last step's vertices)
__device__ __forceinline__ float query_radius(const float* qv, float cgx, float cgy) {
  float R = 0.0f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float ax = qv[2 * q] - cgx, ay = qv[2 * q + 1] - cgy;
    R = fmaxf(R, sqrtf(ax * ax + ay * ay));
  }
  return R * 1.000001f + 1e-6f;
}

// ---------------------------------------------------------------------------------------------------------------------
// B2, pruned, TWO agents per wavefront (the production path).  Lane groups of 16: [A-left | A-right | B-left | B-right] for the
// boundary pass, half-waves [A | B] for the centre-line pass.  Chunks are runs of SIGMAENV_CHUNK consecutive segments; with the
// CPM map ~3 chunks (12-16 segments) survive per boundary, so one 16-lane pass usually covers a boundary.
// Results go straight to LDS (dref, dleft, dright, cp, flags[0]).
// `stale_first`: in the step the first agent of every env queries its corners at last step's vertices (see step kernel).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int nth_set_bit64(unsigned long long m, int c) {
  for (int t = 0; t < c; ++t) m &= (m - 1ull);
  return m ? (__ffsll((long long)m) - 1) : -1;
}
// candidate-chunk masks: bit c of the (agent, polyline) mask is set iff the bounding box of chunk c is within the scan threshold
// of the agent's centre.  Two levels so that the far part of the polyline costs one box per 8 chunks (32 segments): first the <= 8
// group boxes, then the 8 chunk boxes of every group that is within the threshold (usually one or two).  One lane per
// (agent, polyline); the loads of a level are independent and in flight together.  The task also derives the threshold itself.
__device__ __forceinline__ bool box_within(const float4 b, float px, float py, float T2) {
  float dx = fmaxf(fmaxf(b.x - px, px - b.z), 0.0f);
  float dy = fmaxf(fmaxf(b.y - py, py - b.w), 0.0f);
  return !((dx * dx + dy * dy) > T2);  // a NaN threshold keeps every box
}
// The task is written in stages so that the step kernel can issue stage 1 early: it only needs the path row and last step's closest
// index (known before the dynamics), stage 2 needs the new position, stage 3 tests the boxes.
struct MaskTask {
  int sl, pl, path, k, npt, nch;
  float ax, ay, bx, by;          // the segment that was closest last step
  unsigned long long nm_tight, nm_near, nm_far;
  int lvl;                       // neighbour level the threshold selects: 0 tight, 1 near, 2 far, 3 none (two-level search over all boxes)
  int own;                       // chunk of the segment that was closest last step
  const float4* box;
  float px, py, T2;
  unsigned long long rest;  // neighbour-mask bits to test
  bool fast;
};
__device__ __forceinline__ void mask_stage1(const DevMap& m, MaskTask& mt, int task, int path, int cp) {
  mt.sl = task / 3; mt.pl = task - mt.sl * 3; mt.path = path;
  const int pl = mt.pl;
  const float* poly = m.center + (size_t)pl * m.poly_stride + (size_t)path * m.P * 2;
  mt.box = m.chunk_box + ((size_t)path * 3 + pl) * m.nch;
  const ulonglong4* neigh = m.chunk_neigh + ((size_t)path * 3 + pl) * m.nch;
  // One round trip: the point count, and -- speculatively, the index is almost always in range -- the segment that was closest
  // last step together with the neighbour masks of its chunk.
  int k = cp - 1;
  k = k < 0 ? 0 : (k > m.P - 2 ? m.P - 2 : k);
  mt.k = k;
  mt.npt = m.n_center[pl * m.n_paths + path];
  const Seg4 sg = load_segment(reinterpret_cast<const float2*>(poly), k);
  mt.ax = sg.ax; mt.ay = sg.ay; mt.bx = sg.bx; mt.by = sg.by;
  const ulonglong4 nm = neigh[k / SIGMAENV_CHUNK];
  mt.nm_tight = nm.x; mt.nm_near = nm.y; mt.nm_far = nm.z;
  mt.own = k / SIGMAENV_CHUNK;
}
template <bool COLLIDE>
__device__ __forceinline__ void mask_stage2(const DevMap& m, const Smem& s, MaskTask& mt, bool stale_first, int N) {
  const int sl = mt.sl, pl = mt.pl, npt = mt.npt;
  if (mt.k > npt - 2) {  // rare (a caller-provided start index beyond this polyline)
    const float* poly = m.center + (size_t)pl * m.poly_stride + (size_t)mt.path * m.P * 2;
    const int k = npt - 2 < 0 ? 0 : npt - 2;
    const Seg4 sg = load_segment(reinterpret_cast<const float2*>(poly), k);
    mt.ax = sg.ax; mt.ay = sg.ay; mt.bx = sg.bx; mt.by = sg.by;
    const ulonglong4 nm = (m.chunk_neigh + ((size_t)mt.path * 3 + pl) * m.nch)[k / SIGMAENV_CHUNK];
    mt.nm_tight = nm.x; mt.nm_near = nm.y; mt.nm_far = nm.z;
    mt.own = k / SIGMAENV_CHUNK;
  }
  const float px = s.st[sl * 8], py = s.st[sl * 8 + 1];
  mt.px = px; mt.py = py;
  // pruning threshold T (see the exactness argument above): distance to last step's closest segment, plus (boundaries) twice the
  // radius of the query points around the centre -- exact for the agent whose corners are stale -- and at least the circumradius
  // when the rectangle is also tested for collision
  const float MARGIN = 1e-4f;
  const float glx = mt.bx - mt.ax, gly = mt.by - mt.ay;
  // dg only has to be an UPPER bound of the distance to that segment: the distance to ANY point of the segment is one, so the projection parameter may come
  // from the approximate reciprocal and the root from the 1-ulp instruction (their rounding is far inside MARGIN) -- no IEEE division / square-root sequences
  float dg;
  {
    const float vx = px - mt.ax, vy = py - mt.ay;
    const float tq = clampf((vx * glx + vy * gly) * __builtin_amdgcn_rcpf(glx * glx + gly * gly), 0.0f, 1.0f);  // (a zero-length segment: NaN -> 0)
    const float ex = (mt.ax + glx * tq) - px, ey = (mt.ay + gly * tq) - py;
    dg = __builtin_amdgcn_sqrtf(fmaf(ey, ey, ex * ex)) * 1.000001f + 1e-6f;
  }
  float T = dg;
  if (pl != 0) {
    const bool stale = stale_first && (sl % N == 0);  // (only evaluated for the step kernel: N is a kernel-uniform divisor there)
    // the agent whose corner queries are last step's vertices: they lie within the circumradius of last step's position (s.thr, staged by phase A), i.e. within
    // circumradius + |move| of the new centre -- a bound from one root instead of the exact maximum over the four corners (four roots)
    float Rq = m.rect_radius;
    if (stale) {
      const float mx = px - s.thr[sl * 3], my = py - s.thr[sl * 3 + 1];
      Rq = (m.rect_radius + __builtin_amdgcn_sqrtf(fmaf(my, my, mx * mx))) * 1.000001f + 1e-6f;
    }
    T = fmaxf(T + 2.0f * Rq, COLLIDE ? m.rect_radius : 0.0f);
  }
  T += MARGIN;
  mt.T2 = T * T;
  mt.nch = (npt - 1 + SIGMAENV_CHUNK - 1) / SIGMAENV_CHUNK;
  // Every chunk within T of the agent is within T + dg of the chunk of that segment (the segment lies inside its chunk's box and is
  // dg away), i.e. in the precomputed neighbour mask of that radius (the wider one mostly serves the agent whose query points are
  // stale): only those few boxes are tested, eight loads in flight at a time.
  mt.fast = T + dg <= m.neigh_radius_far;
  mt.lvl = mt.fast ? ((T + dg <= m.neigh_radius_tight) ? 0 : ((T + dg <= m.neigh_radius) ? 1 : 2)) : 3;
  mt.rest = mt.fast ? (mt.lvl == 0 ? mt.nm_tight : (mt.lvl == 1 ? mt.nm_near : mt.nm_far)) : 0ull;
}
__device__ __forceinline__ void mask_stage3(const DevMap& m, const Smem& s, MaskTask& mt, int task) {
  const float px = mt.px, py = mt.py, T2 = mt.T2;
  const int nch = mt.nch;
  unsigned long long mk = 0ull;
  unsigned long long rest = mt.rest;
  if (mt.fast && mt.own < nch) { mk |= 1ull << mt.own; rest &= ~(1ull << mt.own); }  // (see scan_tile_balanced)
  while (rest) {  // one pass unless the neighbourhood has more than eight chunks (dense map regions with the wide radius)
    int idx[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { idx[q] = rest ? (__ffsll((long long)rest) - 1) : -1; rest &= rest - 1ull; }
    float4 b[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) b[q] = mt.box[idx[q] < 0 ? 0 : idx[q]];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (idx[q] >= 0 && idx[q] < nch && box_within(b[q], px, py, T2)) mk |= 1ull << idx[q];
    }
  }
  if (!mt.fast) {
    // far from the own path (or a NaN state): two-level search over all boxes -- the <= 8 group boxes, then the 8 chunk boxes of
    // every group within the threshold
    const float4* gbox = m.group_box + ((size_t)mt.path * 3 + mt.pl) * 8;
    unsigned gm = 0u;
#pragma unroll
    for (int gidx = 0; gidx < 8; ++gidx) {
      if (gidx * 8 < nch && box_within(gbox[gidx], px, py, T2)) gm |= 1u << gidx;
    }
    while (gm) {
      const int gidx = __ffs((int)gm) - 1;
      gm &= gm - 1u;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int cidx = gidx * 8 + q;
        if (cidx < nch && box_within(mt.box[cidx], px, py, T2)) mk |= 1ull << cidx;
      }
    }
  }
  s.npts[task] = mt.npt;
  s.cmask[task] = mk;
  uint8_t* cl = s.cand + task * CAND_LIST;
  for (int j = 0; mk && j < CAND_LIST; ++j) {
    cl[j] = (uint8_t)(__ffsll((long long)mk) - 1);
    mk &= mk - 1ull;
  }
}
template <bool COLLIDE>
__device__ inline void scan_mask_task(const DevMap& m, const Smem& s, int task, bool stale_first, int N) {
  // task = (agent slot, polyline): 0 centre line, 1 left boundary, 2 right boundary
  MaskTask mt;
  mask_stage1(m, mt, task, s.path[task / 3], s.cp[task]);
  mask_stage2<COLLIDE>(m, s, mt, stale_first, N);
  mask_stage3(m, s, mt, task);
}

template <bool COLLIDE, bool FASTDIV>
__device__ inline void pair_scan(const DevMap& m, const sigmaenv_config_t& c, const Smem& s, int slotA, int valid_mask, int lane, bool stale_first, int N,
                                 int rA = -1) {
  // valid_mask: bit 0 = scan agent slotA, bit 1 = scan agent slotA + 1 (an unselected half mirrors the selected one, results dropped)
  const int grp = lane >> 4, ag = grp >> 1, side = grp & 1, gl = lane & 15, hl = lane & 31;
  const bool valid = (valid_mask >> ag) & 1;
  const int sl = slotA + (valid ? ag : (ag ^ 1));
  const int path = s.path[sl];
  const float cgx = s.st[sl * 8], cgy = s.st[sl * 8 + 1];
  if (rA < 0) rA = slotA % N;  // agent index of slotA (wavefront-uniform; the step kernel tracks it without a division)
  const int rs = rA + (sl - slotA);
  const bool stale = stale_first && (rs == 0 || rs == N);
  const float* qv = stale ? (s.vold + sl * 10) : (s.vnew + sl * 10);
  const float* ev = s.vnew + sl * 10;
  const float2* ctr2 = reinterpret_cast<const float2*>(m.center + (size_t)path * m.P * 2);
  const float2* pol2 = reinterpret_cast<const float2*>(m.left + (size_t)side * m.poly_stride + (size_t)path * m.P * 2);
  const int n = s.npts[sl * 3], np = s.npts[sl * 3 + 1 + side];
  // ---- candidate chunk masks (scan_mask_task, one lane per (agent, polyline), computed before the scan)
  const unsigned long long mc = s.cmask[sl * 3], mb = s.cmask[sl * 3 + 1 + side];
  const uint8_t* clc = s.cand + (sl * 3) * CAND_LIST;
  const uint8_t* clb = s.cand + (sl * 3 + 1 + side) * CAND_LIST;
  float4* escr = s.escr + ((threadIdx.x >> 6) * 4 + grp) * 4;
  const float near_thr = m.rect_radius + 1e-4f;
  const float q0x = qv[0], q0y = qv[1], q1x = qv[2], q1y = qv[3], q2x = qv[4], q2y = qv[5], q3x = qv[6], q3y = qv[7];
  const int cnt_c = __popcll(mc) * SIGMAENV_CHUNK, cnt_b = __popcll(mb) * SIGMAENV_CHUNK;
  float cd = INFINITY, bd0 = INFINITY, bs0 = INFINITY, bs1 = INFINITY, bs2 = INFINITY, bs3 = INFINITY;
  int ck = 0, bk = 0;
  bool h = false;
  // One pass covers 32 centre-line and 16 boundary segments per agent (side): both polylines in one pass, so that their segment
  // loads are in flight together.  Nearly every scan is ONE pass; the first pass therefore assigns its results instead of merging them
  // into the running minima (FIRST), the rare further passes merge.
  auto pass = [&](int it, auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const int jc = it * 32 + hl, jb = it * 16 + gl;
    const int qc = jc / SIGMAENV_CHUNK, qb = jb / SIGMAENV_CHUNK;
    const int chc = (jc < cnt_c) ? (qc < CAND_LIST ? (int)clc[qc] : nth_set_bit64(mc, qc)) : -1;
    const int chb = (jb < cnt_b) ? (qb < CAND_LIST ? (int)clb[qb] : nth_set_bit64(mb, qb)) : -1;
    const int kc = chc * SIGMAENV_CHUNK + (jc % SIGMAENV_CHUNK), kb = chb * SIGMAENV_CHUNK + (jb % SIGMAENV_CHUNK);
    const bool do_c = chc >= 0 && kc + 1 < n, do_b = chb >= 0 && kb + 1 < np;
    float2 ca = make_float2(0.f, 0.f), cb = ca, ba = ca, bb2 = ca;
    if (do_c) { const Seg4 sg = load_segment(ctr2, kc); ca = make_float2(sg.ax, sg.ay); cb = make_float2(sg.bx, sg.by); }
    if (do_b) { const Seg4 sg = load_segment(pol2, kb); ba = make_float2(sg.ax, sg.ay); bb2 = make_float2(sg.bx, sg.by); }
    if (do_c) {
      float lx = cb.x - ca.x, ly = cb.y - ca.y;
      float len2 = lx * lx + ly * ly;
      float d = point_segment_t<FASTDIV>(cgx, cgy, ca.x, ca.y, lx, ly, len2, FASTDIV ? shared_rcp(len2) : 0.0f);
      if (FIRST || d < cd || (d == cd && kc < ck)) { cd = d; ck = kc; }
    }
    float d0 = INFINITY;
    if (do_b) {
      float lx = bb2.x - ba.x, ly = bb2.y - ba.y;
      float len2 = lx * lx + ly * ly;
      const float rcp = FASTDIV ? shared_rcp(len2) : 0.0f;
      d0 = point_segment_t<FASTDIV>(cgx, cgy, ba.x, ba.y, lx, ly, len2, rcp);
      if (FIRST || d0 < bd0 || (d0 == bd0 && kb < bk)) { bd0 = d0; bk = kb; }
      const float v0 = point_segment_sq_t<FASTDIV>(q0x, q0y, ba.x, ba.y, lx, ly, len2, rcp);
      const float v1 = point_segment_sq_t<FASTDIV>(q1x, q1y, ba.x, ba.y, lx, ly, len2, rcp);
      const float v2 = point_segment_sq_t<FASTDIV>(q2x, q2y, ba.x, ba.y, lx, ly, len2, rcp);
      const float v3 = point_segment_sq_t<FASTDIV>(q3x, q3y, ba.x, ba.y, lx, ly, len2, rcp);
      if (FIRST) { bs0 = v0; bs1 = v1; bs2 = v2; bs3 = v3; }
      else { bs0 = fminf(bs0, v0); bs1 = fminf(bs1, v1); bs2 = fminf(bs2, v2); bs3 = fminf(bs3, v3); }
    }
    if (COLLIDE) {
      // Only a segment within the rectangle's circumradius of the (new) centre can cross an edge (same argument as the pruning,
      // DESIGN.md).  Those few segments (1-3 per boundary) are staged in LDS and the (segment, edge) tests are spread over the
      // lanes of the group, one test per lane, instead of four edge tests in every lane.
      const bool nearseg = do_b && !(d0 > near_thr);
      const unsigned long long nb = __ballot(nearseg);
      if (nb) {
        const unsigned gm16 = (unsigned)(nb >> (grp * 16)) & 0xFFFFu;
        const int rank = __popc(gm16 & ((1u << gl) - 1u)), cntn = __popc(gm16);
        for (int p0 = 0; __any(p0 < cntn); p0 += 4) {
          if (nearseg && rank >= p0 && rank < p0 + 4) escr[rank - p0] = make_float4(ba.x, ba.y, bb2.x, bb2.y);
          if (p0 + (gl >> 2) < cntn) {
            const float4 sg = escr[gl >> 2];
            const float* pv = ev + 2 * (gl & 3);
            const Edge e = make_edge(pv[0], pv[1], pv[2], pv[3]);
            const float lx2 = sg.z - sg.x, ly2 = sg.w - sg.y;
            const float S2 = lx2 * sg.y - ly2 * sg.x;
            h |= edge_hits_segment(e, sg.x, sg.y, sg.z, sg.w, lx2, ly2, S2);
          }
        }
      }
    }
  };
  pass(0, std::true_type{});
  for (int it = 1; __any(it * 32 < cnt_c || it * 16 < cnt_b); ++it) pass(it, std::false_type{});
  // ---- reductions: DPP inside the rows of 16 lanes; the centre line needs one cross-row step.  The lexicographic
  // (distance, index) minimum is taken as: minimum distance first, then the lowest index among the lanes that hold it.
  {
    const float bd_own = bd0, cd_own = cd;
    row16_min6(bd0, bs0, bs1, bs2, bs3, cd);
    cd = fminf(cd, __shfl_xor(cd, 16, 64));
    int kb_c = (bd_own == bd0) ? bk : 0x7FFFFFFF, kc_c = (cd_own == cd) ? ck : 0x7FFFFFFF;
    row16_min2_i32(kb_c, kc_c);
    bk = kb_c;
    ck = min(kc_c, __shfl_xor(kc_c, 16, 64));
  }
  const unsigned long long hb = COLLIDE ? __ballot(h && valid) : 0ull;
  // the four corner minima are row-uniform after the reduction: lanes 0..3 of the row take one square root each
  const float bsq = gl == 0 ? bs0 : (gl == 1 ? bs1 : (gl == 2 ? bs2 : bs3));
  const float bcorner = sqrtf(bsq);
  if (gl < 4 && valid) ((side ? s.dright : s.dleft) + sl * 5)[1 + gl] = bcorner;
  if (gl == 0 && valid) {
    float wh = (float)((double)c.width / 2.0);
    float* dst = (side ? s.dright : s.dleft) + sl * 5;
    dst[0] = bd0 - wh;  // world_state_rt.py:608-610
    s.cp[sl * 3 + 1 + side] = bk + 1;
    if (side == 0) {
      s.dref[sl] = cd;
      s.cp[sl * 3] = ck + 1;
      if (COLLIDE) s.flags[sl * 4 + 0] = ((hb >> (ag * 32)) & 0xFFFFFFFFull) ? 1 : 0;
    }
  }
}

// mutual distance of slots (si, sj) of one env, si != sj  (update_mutual_distances, world_state_rt_sim.py:360-373)
__device__ __forceinline__ float pair_distance(const sigmaenv_config_t& c, const float* st, const float* verts, int si, int sj) {
  if (c.distance_type == SIGMAENV_DIST_C2C) {
    float dx = st[si * 8] - st[sj * 8], dy = st[si * 8 + 1] - st[sj * 8 + 1];
    return sqrtf(dx * dx + dy * dy);
  }
  int a = si < sj ? si : sj, b = si < sj ? sj : si;
  return mtv_pair(verts + a * 10, verts + b * 10);
}

// _apply_ttc_near_agent_penalty, road_traffic.py:1255-1332; st points at the env's first slot
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

// top-k nearest agents of one agent (observation_provider_rt.py:629-636): ascending, lowest index on ties
__device__ inline void topk_nearest(const float* Drow, int N, int K, int* out) {
  if (K == 2) {
    // one pass keeping the two best; the row is read four entries at a time so that the LDS reads are in flight together.
    // `idx < 0` makes the first entries win unconditionally, as the selection loop below does (inf / NaN rows)
    float b0 = INFINITY, b1 = INFINITY;
    int i0 = -1, i1 = -1;
    for (int j0 = 0; j0 < N; j0 += 4) {
      float d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) d[u] = Drow[min(j0 + u, N - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u;
        if (j < N) {
          if (i0 < 0 || d[u] < b0) { b1 = b0; i1 = i0; b0 = d[u]; i0 = j; }
          else if (i1 < 0 || d[u] < b1) { b1 = d[u]; i1 = j; }
        }
      }
    }
    out[0] = i0; out[1] = i1;
    return;
  }
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

// top-k nearest agents of every (selected) slot of the tile into s.near (observation_provider_rt.py:629-636): ascending distance, lowest index on ties
template <bool WAVE, class RealSlot>
__device__ __forceinline__ void topk_tile(const Smem& s, int N, int K, int n_slots, int TID, int NTHR, RealSlot real_slot) {
  if (K == 2 && n_slots * 4 <= NTHR) {
    // Two nearest of every agent with FOUR lanes per agent (a quad): lane q scans the candidates q, q + 4, q + 8, ... in increasing
    // order, then the quads merge their (distance, index)-sorted pairs through DPP quad permutations.  Same result as the selection
    // loop (ascending distance, lowest index on ties, unconditional first entries), a quarter of its dependent chain.
    const int tid = TID, qd = tid & 3, v = tid >> 2;
    const bool act = v < n_slots;
    const int sl = act ? real_slot(v) : 0;
    const float* row = s.dist + sl * DIST_STRIDE(N);
    float b0 = INFINITY, b1 = INFINITY;
    int i0 = 0x7FFFFFFF, i1 = 0x7FFFFFFF;  // "no entry": loses every index tie; replaced by the first real entry unconditionally
    for (int j = qd; j < N; j += 4) {
      const float d = row[j];
      if (i0 == 0x7FFFFFFF || d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = j; }
      else if (i1 == 0x7FFFFFFF || d < b1) { b1 = d; i1 = j; }
    }
    auto less = [](float da, int ia, float db, int ib) {  // (d, i) sorts before (d', i'); an empty entry never sorts first
      if (ia == 0x7FFFFFFF) return false;
      if (ib == 0x7FFFFFFF) return true;
      return da < db || (!(db < da) && ia < ib);
    };
#define SIGMA_QUAD(x, CTRL) __builtin_amdgcn_update_dpp(0, (x), (CTRL), 0xF, 0xF, true)
#define SIGMA_TOP2_MERGE(CTRL)                                                                      \
  {                                                                                                 \
    const float c0 = __int_as_float(SIGMA_QUAD(__float_as_int(b0), CTRL)), c1 = __int_as_float(SIGMA_QUAD(__float_as_int(b1), CTRL)); \
    const int k0 = SIGMA_QUAD(i0, CTRL), k1 = SIGMA_QUAD(i1, CTRL);                                 \
    const bool mine = less(b0, i0, c0, k0);                                                        \
    const float f0 = mine ? b0 : c0, sa = mine ? b1 : b0, sb = mine ? c0 : c1;                       \
    const int g0 = mine ? i0 : k0, ga = mine ? i1 : i0, gb = mine ? k0 : k1;                        \
    const bool second_a = less(sa, ga, sb, gb);                                                    \
    b0 = f0; i0 = g0; b1 = second_a ? sa : sb; i1 = second_a ? ga : gb;                             \
  }
    SIGMA_TOP2_MERGE(0xB1)  // quad_perm [1,0,3,2]
    SIGMA_TOP2_MERGE(0x4E)  // quad_perm [2,3,0,1]
#undef SIGMA_TOP2_MERGE
#undef SIGMA_QUAD
    if (act && qd == 0) { s.near[sl * K] = i0; s.near[sl * K + 1] = i1; }
  } else {
    for (int v = TID; v < n_slots; v += NTHR) {
      const int sl = real_slot(v);
      topk_nearest(s.dist + sl * DIST_STRIDE(N), N, K, s.near + sl * K);
    }
  }
}

// D: observations of all slots of the tile (observation_provider_rt.py:345-588 latest slot, :594-925 default flags).
// Three branch-free passes so that every lane of a wavefront runs the same code:
//   1. ego-view transforms (own short-term path points, observed neighbours' vertices): atan2 + cos + sin each
//   2. relative velocities (self + observed neighbours): angle wrap + cos + sin each
//   3. normalised distances
// Rows are assembled in LDS and written out coalesced.  All threads of the block participate.
// env_sel / n_sel: restrict the work to these envs of the tile (the reset tail only refreshes the envs it touched); the loops then run
// over "virtual" slots v = (position in env_sel) * N + agent, so that the lanes stay densely used.
template <bool WAVE>
__device__ __forceinline__ void observe_tile_default(const sigmaenv_config_t& c, const Smem& s, const DevBufs& g, const Tile& t,
                                                     const int* env_sel, int n_sel, bool write_global, const int* tim, const CfgDerived* dvp = nullptr) {
  // tim: the timer rows [nenv][4] of the tile's envs (LDS copy of the step kernel's step loop; default: SIGMAENV_BUF_TIMER) -- the observation
  // noise is keyed on an env's (episodes_reset, timer.step)
  const int N = t.N, K = t.K, D = t.D;
  const int TID = Grp<WAVE>::tid(), NTHR = Grp<WAVE>::size();
  const int n_slots = env_sel ? n_sel * N : t.slots;
  auto real_slot = [&](int v) {
    if (!env_sel) return v;
    const int q = fdiv(v, g.mN);
    return env_sel[q] * N + (v - q * N);
  };
#define TSO(k) do { } while (0)
  const float n_pos = (float)((double)c.length * 10.0);   // normalizers.pos, road_traffic.py:588-592
  const float n_v = c.max_speed;                            // :596
  const float n_dl = (float)((double)c.lane_width * 3.0);   // :599-601 (distance_lanelet also normalises the agent distances)
  // The observation rows are held to 1e-5, not to the bit (the ego transform is already the rotation form): the normalisers divide through their reciprocals
  // (<= 1 ulp from the quotient; ~10 instructions less per division, 14 divisions per agent row)
  // (dvp: the step kernel hands in the host-derived reciprocals -- derive_config, the same three IEEE divisions -- instead of dividing per lane and call)
  const float r_pos = dvp ? dvp->r_pos : 1.0f / n_pos, r_v = dvp ? dvp->r_v : 1.0f / n_v, r_dl = dvp ? dvp->r_dl : 1.0f / n_dl;
  topk_tile<WAVE>(s, N, K, n_slots, TID, NTHR, real_slot);
  TSO(0);
  Grp<WAVE>::sync();
  TSO(1);
  // The reference goes through atan2 / cos / sin (helper_scenario.py:1261-1271); the same rotation is applied here with the
  // agent's cos(psi), sin(psi) (already needed for the vertices): rel = R(-psi_i) (p_j - p_i).  Identical up to ~1e-7, well inside
  // the 1e-5 bar; no mask or index depends on it.  The oracle keeps the reference's formulation.
  const int T1 = NS + 4 * K;
  for (int w = TID; w < n_slots * T1; w += NTHR) {
    const int v = fdiv(w, g.mT1), q = w - v * T1;
    const int sl = real_slot(v);
    int ebase = fdiv(sl, g.mN) * N;
    const float* si = s.st + sl * 8;
    float tx, ty;
    int pos;
    bool masked = false;
    if (q < NS) {  // [own] short-term reference path (:451-460, :893-897)
      tx = s.shrt[sl * NS * 2 + 2 * q]; ty = s.shrt[sl * NS * 2 + 2 * q + 1];
      pos = 1 + 2 * q;
    } else {       // [others] vertices of the k-th nearest agent (:485-492, :731-737)
      int k = (q - NS) >> 2, v = (q - NS) & 3;
      int sj = ebase + s.near[sl * K + k];
      tx = s.vnew[sj * 10 + 2 * v]; ty = s.vnew[sj * 10 + 2 * v + 1];
      pos = 4 + 2 * NS + 11 * k + 2 * v;
      // masked by distance (observation_provider_rt.py:638-665, 734-737): the vertices read 1
      masked = c.is_apply_mask && s.dist[sl * DIST_STRIDE(N) + s.near[sl * K + k]] >= c.distance_mask_agents;
    }
    float dx = tx - si[0], dy = ty - si[1];
    float ci = s.cs[sl * 2], sn = s.cs[sl * 2 + 1];
    s.obs[sl * D + pos] = masked ? 1.0f : (dx * ci + dy * sn) * r_pos;
    s.obs[sl * D + pos + 1] = masked ? 1.0f : (dy * ci - dx * sn) * r_pos;
  }
  TSO(2);
  // relative velocities (one lane per (agent, self or observed neighbour)) from the front of the block, the per-agent distances
  // from its back, so that both run at the same time when the block is wide enough
  const int T2 = K + 1;
  const int n2 = n_slots * T2, n3 = n_slots;
  const int span = max(n2 + n3, NTHR);
  for (int w0 = TID; w0 < span; w0 += NTHR) {
    const int w3 = (span - 1) - w0;  // the back of the span carries the third pass
    if (w0 < n2) {
      const int w = w0;
      const int v = fdiv(w, g.mT2), q = w - v * T2;
      const int sl = real_slot(v);
      int ebase = fdiv(sl, g.mN) * N;
      int sj = (q == 0) ? sl : (ebase + s.near[sl * K + (q - 1)]);
      const float* sjp = s.st + sj * 8;
      float va = norm2(sjp[5], sjp[6]);                      // :444
      float ci = s.cs[sl * 2], si_ = s.cs[sl * 2 + 1], cj = s.cs[sj * 2], sj_ = s.cs[sj * 2 + 1];
      float cr = cj * ci + sj_ * si_, sr = sj_ * ci - cj * si_;   // cos / sin of (psi_j - psi_i) (:439, :447-449)
      if (q == 0) {
        s.obs[sl * D] = va * r_v;  // [own] only the longitudinal component is observed; cos(0) = 1 (:864-868, :885-887)
      } else {
        int base = 4 + 2 * NS + 11 * (q - 1);
        const bool masked = c.is_apply_mask && s.dist[sl * DIST_STRIDE(N) + s.near[sl * K + (q - 1)]] >= c.distance_mask_agents;  // :717-719
        s.obs[sl * D + base + 8] = masked ? 0.0f : (va * cr) * r_v;
        s.obs[sl * D + base + 9] = masked ? 0.0f : (va * sr) * r_v;
      }
    } else if (w3 < n3) {
      const int sl = real_slot(w3);
      float ml = INFINITY, mr = INFINITY;
#pragma unroll
      for (int q = 0; q < 5; ++q) { ml = fminf(ml, s.dleft[sl * 5 + q]); mr = fminf(mr, s.dright[sl * 5 + q]); }
      s.obs[sl * D + 1 + 2 * NS] = s.dref[sl] * r_dl;        // :376-378, :898-904
      s.obs[sl * D + 2 + 2 * NS] = ml * r_dl;                // :379-382
      s.obs[sl * D + 3 + 2 * NS] = mr * r_dl;                // :383-386
      for (int k = 0; k < K; ++k) {
        const float dk = s.dist[sl * DIST_STRIDE(N) + s.near[sl * K + k]];
        s.obs[sl * D + 4 + 2 * NS + 11 * k + 10] = (c.is_apply_mask && dk >= c.distance_mask_agents) ? 1.0f : dk * r_dl;  // :373-375, :747-749
      }
    }
  }
  TSO(3);
  Grp<WAVE>::sync();
  TSO(4);
  if (c.obs_noise_level > 0.0f) {  // sensor noise on every element of the rows just assembled (observation_provider_rt.py:613-618)
    if (!tim) tim = g.timer + (size_t)t.env0 * 4;
    for (int w = TID; w < n_slots * D; w += NTHR) {
      const int v = w / D, k = w - v * D;
      const int sl = real_slot(v);
      const int e = fdiv(sl, g.mN), i = sl - e * N;
      s.obs[sl * D + k] = s.obs[sl * D + k] + obs_noise(c, c.env_index_base + t.env0 + e, i, k, tim[e * 4 + 3], tim[e * 4], g.obs_salt);
    }
    Grp<WAVE>::sync();
  }
  if (env_sel) {  // only the rows of the selected envs (they are contiguous per env)
    const int ND = N * D, NK = N * K;
    for (int q = 0; q < n_sel; ++q) {
      const int e = env_sel[q];
      for (int k = TID; k < ND; k += NTHR) g.obs[(t.a0 + e * N) * D + k] = s.obs[e * ND + k];
      for (int k = TID; k < NK; k += NTHR) g.nearing[(t.a0 + e * N) * K + k] = s.near[e * NK + k];
    }
  } else if (write_global) {  // (false: an intermediate step of the in-kernel step loop -- the rows stay in LDS for the record, HBM gets the last step's)
    if ((D & 3) == 0) {  // rows are whole float4s: both the LDS staging area and the tile's slice of g.obs are 16-byte aligned
      const float4* so4 = reinterpret_cast<const float4*>(s.obs);
      float4* go4 = reinterpret_cast<float4*>(g.obs + t.a0 * D);
      for (int k = TID; k < t.slots * D / 4; k += NTHR) go4[k] = so4[k];
    } else {
      for (int k = TID; k < t.slots * D; k += NTHR) g.obs[t.a0 * D + k] = s.obs[k];
    }
    for (int k = TID; k < t.slots * K; k += NTHR) g.nearing[t.a0 * K + k] = s.near[k];
  }
  TSO(5);
#undef TSO
}


// ---- the observation for obs_flags != 0 --------------------------------------------------------------------------------------------------------------
// _observe_self / _observe_other_agents of observation_provider_rt.py:594-925 with the quantities update_state prepares (:345-588), for every combination of
// the SIGMAENV_OBS_* switches (include/sigmaenv.h; layout: ObsLayout), assembled from the tile in LDS exactly like the default row: lane per item, rows staged
// in s.obs, noise, coalesced write-out -- so that every launch that refreshes observations (the fused step incl. its T-step loop and the rollout record, the
// resets, sigmaenv_observe) produces the configured row itself.  Ego-view transforms use the rotation form of the default path.
// Value k of agent slot sl's row in SIGMAENV_OBS_FULL mode, from the staging of observe_tile_variant: the [own] blocks of the `slots` agents, then ONE copy of the
// [others] part per env (it does not depend on the observing agent in the bird view), placeholders zero; plus the sensor noise of element k.
__device__ __forceinline__ float full_obs_value(const sigmaenv_config_t& c, const Smem& s, const Tile& t, const ObsLayout& L, int e, int i, int k, const int* tim, uint32_t salt = 0u) {
  const int sl = e * t.N + i;
  float v = k < L.own_w ? s.obs[sl * L.own_w + k] : (k < L.own_w + L.W_oth ? s.obs[t.slots * L.own_w + e * L.W_oth + (k - L.own_w)] : 0.0f);
  if (c.obs_noise_level > 0.0f) v = v + obs_noise(c, c.env_index_base + t.env0 + e, i, k, tim[e * 4 + 3], tim[e * 4], salt);
  return v;
}

template <bool WAVE>
__device__ __forceinline__ void observe_tile_variant(const sigmaenv_config_t& c, const Smem& s, const DevBufs& g, const Tile& t, const int* env_sel, int n_sel,
                                            bool write_global, const int* tim, const CfgDerived* dvp = nullptr) {
  const int N = t.N, K = t.K, F = c.obs_flags;
  const ObsLayout L(F, N, K);
  const int D = L.full ? L.own_w : L.D;   // row stride of the staging area
  const int Kv = L.full ? 0 : K;          // neighbours assembled per agent (FULL: none, the [others] part is assembled once per env below)
  const int TID = Grp<WAVE>::tid(), NTHR = Grp<WAVE>::size();
  const int n_env = env_sel ? n_sel : t.nenv;
  const int n_slots = n_env * N;
  auto real_slot = [&](int v) {
    if (!env_sel) return v;
    const int q = fdiv(v, g.mN);
    return env_sel[q] * N + (v - q * N);
  };
  if (!tim) tim = g.timer + (size_t)t.env0 * 4;
  const float n_pos = (float)((double)c.length * 10.0);      // normalizers.pos, road_traffic.py:588-592
  const float n_v = c.max_speed;                              // :596
  const float n_dl = (float)((double)c.lane_width * 3.0);    // :599-601
  const float n_rot = (float)(2.0 * 3.141592653589793);      // normalizers.rot, :597
  const float n_da = (float)((double)c.length * 10.0);       // normalizers.distance_agent, :605-607
  const float nwx = c.world_x_dim, nwy = c.world_y_dim;      // normalizers.pos_world (bird view, :537-575)
  // As in the default row, the rows are held to 1e-5, not to the bit: every normaliser divides through its reciprocal (<= 1 ulp from the quotient; ~9 instructions
  // less per division, two divisions per observed point); the step kernel hands in the host-derived reciprocals (dvp, derive_config)
  const float r_pos = dvp ? dvp->r_pos : 1.0f / n_pos, r_v = dvp ? dvp->r_v : 1.0f / n_v, r_dl = dvp ? dvp->r_dl : 1.0f / n_dl;
  const float r_rot = 1.0f / n_rot, r_da = dvp ? dvp->r_pos : 1.0f / n_da;  // (n_da == n_pos; n_rot is a literal: its reciprocal folds)
  const float r_wx = dvp ? dvp->r_wx : 1.0f / nwx, r_wy = dvp ? dvp->r_wy : 1.0f / nwy;
  const bool bird = L.bird != 0;
  // ---- nearest neighbours (:629-636); the full observation has none and leaves nearing_agents_indices at zero
  if (L.full) {
    for (int w = TID; w < n_slots * K; w += NTHR) {
      const int v = w / (K > 0 ? K : 1), k = w - v * K;
      s.near[real_slot(v) * K + k] = 0;
    }
  } else {
    topk_tile<WAVE>(s, N, K, n_slots, TID, NTHR, real_slot);
  }
  // ---- mask by lanelet relation (:638-665, map_manager.py:41-118): bird view only -- the agents' lanelets are computed in that branch of update_state only
  // (:577-588) --, on maps whose parser lists neighbouring lanelets.  determine_current_lanelet: squared distance to every (zero-padded) centre-line point of
  // every lanelet, minimum per lanelet, first lanelet with the smallest minimum == the minimum over (squared-distance bits, lanelet) of all (lanelet, point)
  const bool lane_mask = bird && !L.full && c.is_apply_mask && g.n_lanelets > 0;
  if (lane_mask) {
    for (int v = TID; v < n_slots; v += NTHR) s.key64[real_slot(v)] = ~0ull;
    Grp<WAVE>::sync();
    const int LN = g.n_lanelets, LP = g.lanelet_pts;
    for (int w = TID; w < n_slots * LN; w += NTHR) {
      const int v = w / LN, l = w - v * LN;
      const int sl = real_slot(v);
      const float px = s.st[sl * 8], py = s.st[sl * 8 + 1];
      const float2* c2 = reinterpret_cast<const float2*>(g.lanelet_centers) + (size_t)l * LP;
      float md = INFINITY;
      for (int q = 0; q < LP; ++q) {
        const float2 cc = c2[q];
        const float dx = px - cc.x, dy = py - cc.y;
        md = fminf(md, dx * dx + dy * dy);
      }
      atomicMin(&s.key64[sl], ((unsigned long long)__float_as_uint(md) << 32) | (unsigned)l);
    }
  }
  Grp<WAVE>::sync();
  auto masked = [&](int sl, int ebase, int j) -> bool {  // observed neighbour j (agent index) of slot sl
    if (!c.is_apply_mask) return false;
    bool mk = s.dist[sl * DIST_STRIDE(N) + j] >= c.distance_mask_agents;
    if (lane_mask) mk = mk || !((g.lanelet_neigh[(unsigned)s.key64[sl]] >> (unsigned)s.key64[ebase + j]) & 1ull);
    return mk;
  };
  // ---- points: own short-term path, own boundary points, per neighbour its vertices (or position) and its short-term path
  const int T1 = L.T1;
  for (int w = TID; w < n_slots * T1; w += NTHR) {
    const int v = fdiv(w, g.mVT1), q = w - v * T1;
    const int sl = real_slot(v);
    const int ebase = fdiv(sl, g.mN) * N;
    const float* si = s.st + sl * 8;
    float tx, ty;
    int pos;
    bool mk = false;
    if (q < NS) {  // [own] short-term reference path (:451-460, :564-571, :893-897)
      tx = s.shrt[sl * NS * 2 + 2 * q]; ty = s.shrt[sl * NS * 2 + 2 * q + 1];
      pos = L.p_short + 2 * q;
    } else if (q < NS + L.n_bnd_pts) {
      // [own] the 5 points of each boundary around its closest point instead of the distances (:905-922; world_state_rt.py:686-724: get_short_term_reference_path
      // with sample_interval 1 on the PADDED boundary polyline, the loop rule with the centre line's point count; index shift -2 after a step, +1 for an agent
      // that was (re)placed and not stepped since (:531-576); a negative index counts from the end of the padded tensor)
      const int r = q - NS, side = r >= 5 ? 1 : 0, kk = r - 5 * side;
      const int path = s.path[sl];
      const int n = g.bnd_n_center[path];
      int id = kk + s.cp[sl * 3 + 1 + side] + (s.fresh[sl] ? 1 : -2);
      if (g.bnd_is_loop[path] && id >= n - 1) id = (id + 1) % n;
      if (id < 0) id += g.bnd_P;
      const float2 pt = reinterpret_cast<const float2*>(g.bnd_left + (size_t)side * g.bnd_poly_stride + (size_t)path * g.bnd_P * 2)[id];
      tx = pt.x; ty = pt.y;
      pos = L.p_bnd + 2 * r;
    } else {       // [others] (:803-853), masks (:638-749)
      const int r = q - NS - L.n_bnd_pts;
      const int k = (r >= L.n_oth_pts) + (r >= 2 * L.n_oth_pts) + (r >= 3 * L.n_oth_pts), u = r - k * L.n_oth_pts;  // (K <= SIGMAENV_MAX_NEARING = 4)
      const int j = s.near[sl * K + k], sj = ebase + j, base = L.own_w + k * L.oth_w;
      mk = masked(sl, ebase, j);
      const int nv = (F & SIGMAENV_OBS_NO_VERTICES) ? 1 : 4;
      if (u < nv) {
        if (F & SIGMAENV_OBS_NO_VERTICES) { tx = s.st[sj * 8]; ty = s.st[sj * 8 + 1]; }       // position (:429-434, :672-680)
        else { tx = s.vnew[sj * 10 + 2 * u]; ty = s.vnew[sj * 10 + 2 * u + 1]; }              // vertices (:485-492, :731-737)
        pos = base + 2 * u;
      } else {                                                                                 // its short-term path (:451-460, :721-729)
        tx = s.shrt[sj * NS * 2 + 2 * (u - nv)]; ty = s.shrt[sj * NS * 2 + 2 * (u - nv) + 1];
        pos = base + L.q_ref + 2 * (u - nv);
      }
    }
    float ox, oy;
    if (bird) {
      ox = tx * r_wx; oy = ty * r_wy;
    } else {  // ego view: rel = R(-psi_i) (p - p_i), helper_scenario.py:1241-1273
      const float dx = tx - si[0], dy = ty - si[1];
      const float ci = s.cs[sl * 2], sn = s.cs[sl * 2 + 1];
      ox = (dx * ci + dy * sn) * r_pos;
      oy = (dy * ci - dx * sn) * r_pos;
    }
    s.obs[sl * D + pos] = mk ? 1.0f : ox;
    s.obs[sl * D + pos + 1] = mk ? 1.0f : oy;
  }
  // ---- velocities: own (:547-549, :864-887) and the observed neighbours' (:439-449, :717-719)
  const int T2 = Kv + 1;
  for (int w = TID; w < n_slots * T2; w += NTHR) {
    const int v = Kv ? fdiv(w, g.mT2) : w, q = w - v * T2;
    const int sl = real_slot(v);
    const int ebase = fdiv(sl, g.mN) * N;
    const float* si = s.st + sl * 8;
    if (q == 0) {
      if (bird) { s.obs[sl * D + 3] = si[5] * r_v; s.obs[sl * D + 4] = si[6] * r_v; }
      else s.obs[sl * D] = norm2(si[5], si[6]) * r_v;
    } else {
      const int k = q - 1, j = s.near[sl * K + k], sj = ebase + j, base = L.own_w + k * L.oth_w + L.q_vel;
      const bool mk = masked(sl, ebase, j);
      const float* sjp = s.st + sj * 8;
      if (bird) {
        s.obs[sl * D + base] = mk ? 0.0f : sjp[5] * r_v;
        s.obs[sl * D + base + 1] = mk ? 0.0f : sjp[6] * r_v;
      } else {
        const float va = norm2(sjp[5], sjp[6]);
        const float ci = s.cs[sl * 2], si_ = s.cs[sl * 2 + 1], cj = s.cs[sj * 2], sj_ = s.cs[sj * 2 + 1];
        const float cr = cj * ci + sj_ * si_, sr = sj_ * ci - cj * si_;  // cos / sin of (psi_j - psi_i)
        s.obs[sl * D + base] = mk ? 0.0f : (va * cr) * r_v;
        s.obs[sl * D + base + 1] = mk ? 0.0f : (va * sr) * r_v;
      }
    }
  }
  // ---- scalars, one lane per agent
  for (int v = TID; v < n_slots; v += NTHR) {
    const int sl = real_slot(v);
    const int ebase = fdiv(sl, g.mN) * N;
    const float* si = s.st + sl * 8;
    float* ob = s.obs + sl * D;
    if (bird) {                                                                                   // [own] position, rotation (:862-877)
      ob[0] = si[0] * r_wx; ob[1] = si[1] * r_wy;
      ob[2] = angle_eliminate_two_pi(si[2]) * r_rot;
    }
    if (L.p_steer >= 0) ob[L.p_steer] = angle_eliminate_two_pi(si[4]) * r_rot;                    // :356-360, :392, :888-892
    if (L.p_dcen >= 0) ob[L.p_dcen] = s.dref[sl] * r_dl;                                          // :376-378, :898-904
    if (!L.n_bnd_pts) {
      float ml = INFINITY, mr = INFINITY;
#pragma unroll
      for (int q = 0; q < 5; ++q) { ml = fminf(ml, s.dleft[sl * 5 + q]); mr = fminf(mr, s.dright[sl * 5 + q]); }
      ob[L.p_bnd] = ml * r_dl;                                                                    // :379-386
      ob[L.p_bnd + 1] = mr * r_dl;
    }
    for (int k = 0; k < Kv; ++k) {
      const int j = s.near[sl * K + k], sj = ebase + j, base = L.own_w + k * L.oth_w;
      const bool mk = masked(sl, ebase, j);
      const float* sjp = s.st + sj * 8;
      if (F & SIGMAENV_OBS_NO_VERTICES) {                                                         // rotation, length, width (:437, :551-553, :683-698)
        ob[base + 2] = (mk ? 0.0f : (bird ? angle_eliminate_two_pi(sjp[2]) : angle_eliminate_two_pi(sjp[2] - si[2]))) * r_rot;
        ob[base + 3] = c.length * r_da;
        ob[base + 4] = c.width * r_da;
      }
      if (L.q_steer >= 0) ob[base + L.q_steer] = mk ? 0.0f : angle_eliminate_two_pi(sjp[4]) * r_rot;   // :699-706
      if (L.q_dist >= 0) ob[base + L.q_dist] = mk ? 1.0f : s.dist[sl * DIST_STRIDE(N) + j] * r_dl;      // :373-375, :747-749
    }
    if (!L.full) for (int k = 0; k < L.pad; ++k) ob[L.own_w + K * L.oth_w + k] = 0.0f;             // placeholders, padded BEFORE the noise (:606-611)
  }
  // ---- full observation: the features of ALL agents, once per env (:756-851).  Every feature tensor ([B,N,4,2] vertices, [B,N,2] velocities, ..., and the whole
  // [B,N,N] distance matrix, zeroed entirely by `obs_distance_other_agents[indexing_tuple_2] = 0` in this view, :776-778) is reshaped to [B, n_nearing_agents, -1],
  // i.e. its flat per-env array is cut into K equal chunks, and the row takes chunk 0 of every feature, then chunk 1, ...
  if (L.full) {
    const int Wc = L.W_oth / K;  // elements per row chunk
    float* oth = s.obs + t.slots * L.own_w;
    for (int w = TID; w < n_env * L.W_oth; w += NTHR) {
      const int qe = w / L.W_oth, r = w - qe * L.W_oth;
      const int e = env_sel ? env_sel[qe] : qe;
      const int ck = r / Wc;
      int rem = r - ck * Wc, wf = 1, kind = ObsLayout::F_DIST;
      bool found = false;
#pragma unroll
      for (int f = 0; f < 9; ++f) {
        int fk, fw;
        if (!found && L.feature(f, N, fk, fw)) {
          const int cw = N * fw / K;
          if (rem < cw) { wf = fw; kind = fk; found = true; }
          else rem -= cw;
        }
      }
      const int flat = ck * (N * wf / K) + rem, j = flat / wf, q = flat - j * wf;
      const int sj = e * N + j;
      const float* sjp = s.st + sj * 8;
      float val = 0.0f;                                                                            // F_DIST: zero (:776-778)
      if (kind == ObsLayout::F_VERT) val = s.vnew[sj * 10 + q] / ((q & 1) ? nwy : nwx);           // :555-563
      else if (kind == ObsLayout::F_POS) val = sjp[q] / (q ? nwy : nwx);                          // :539-546
      else if (kind == ObsLayout::F_ROT) val = angle_eliminate_two_pi(sjp[2]) * r_rot;            // :550-553
      else if (kind == ObsLayout::F_LEN) val = c.length * r_da;                                   // :387-389
      else if (kind == ObsLayout::F_WID) val = c.width * r_da;                                    // :390-391
      else if (kind == ObsLayout::F_VEL) val = sjp[5 + q] * r_v;                                  // :547-549
      else if (kind == ObsLayout::F_STEER) val = angle_eliminate_two_pi(sjp[4]) * r_rot;          // :356-360, :392
      else if (kind == ObsLayout::F_REF) val = s.shrt[sj * NS * 2 + q] / ((q & 1) ? nwy : nwx);   // :564-571
      oth[e * L.W_oth + r] = val;
    }
  }
  Grp<WAVE>::sync();
  const int DR = L.D;  // public row width
  if (!L.full && c.obs_noise_level > 0.0f) {  // sensor noise on every element of the rows just assembled (observation_provider_rt.py:613-618)
    for (int w = TID; w < n_slots * DR; w += NTHR) {
      const int v = w / DR, k = w - v * DR;
      const int sl = real_slot(v);
      const int e = fdiv(sl, g.mN), i = sl - e * N;
      s.obs[sl * DR + k] = s.obs[sl * DR + k] + obs_noise(c, c.env_index_base + t.env0 + e, i, k, tim[e * 4 + 3], tim[e * 4], g.obs_salt);
    }
    Grp<WAVE>::sync();
  }
  if (!L.full && !env_sel && write_global && ((N * DR) & 3) == 0) {
    // the whole tile's rows are one contiguous, 16-byte aligned block of whole float4s in the staging area and in SIGMAENV_BUF_OBS
    const float4* so4 = reinterpret_cast<const float4*>(s.obs);
    float4* go4 = reinterpret_cast<float4*>(g.obs + t.a0 * DR);
    for (int k = TID; k < t.slots * DR / 4; k += NTHR) go4[k] = so4[k];
    for (int k = TID; k < t.slots * K; k += NTHR) g.nearing[t.a0 * K + k] = s.near[k];
  } else if (env_sel || write_global) {  // (neither: an intermediate step of the in-kernel step loop -- the rows stay in LDS for the record)
    const int ND = N * DR, NK = N * K;
    for (int q = 0; q < n_env; ++q) {
      const int e = env_sel ? env_sel[q] : q;
      float* go = g.obs + (t.a0 + (size_t)e * N) * DR;
      if (L.full) {
        for (int k = TID; k < ND; k += NTHR) { const int i = k / DR; go[k] = full_obs_value(c, s, t, L, e, i, k - i * DR, tim, g.obs_salt); }
      } else {
        for (int k = TID; k < ND; k += NTHR) go[k] = s.obs[e * ND + k];
      }
      for (int k = TID; k < NK; k += NTHR) g.nearing[(t.a0 + e * N) * K + k] = s.near[e * NK + k];
    }
  }
}

// VARIANTS = false compiles the non-default rows out (the fixed-shape instantiations of the step kernel are only launched with obs_flags == 0)
template <bool WAVE = false, bool VARIANTS = true>
__device__ __forceinline__ void observe_tile(const sigmaenv_config_t& c, const Smem& s, const DevBufs& g, const Tile& t, int ts_base = -1,
                                             const int* env_sel = nullptr, int n_sel = 0, bool write_global = true, const int* tim = nullptr,
                                             const CfgDerived* dvp = nullptr) {
  (void)ts_base;
  if (VARIANTS && c.obs_flags != 0) observe_tile_variant<WAVE>(c, s, g, t, env_sel, n_sel, write_global, tim, dvp);
  else observe_tile_default<WAVE>(c, s, g, t, env_sel, n_sel, write_global, tim, dvp);
}

// ---------------------------------------------------------------------------------------------------------------------
// device-side resets (used by the step kernel's phase R and by sigmaenv_auto_reset)
// VMAS >= 1.4 call order restated per env by the step: world.step(); reward(a) for all a; observation(a) for all a; done()
// ---------------------------------------------------------------------------------------------------------------------
// candidate (path, point) of try `tr` for agent `i` of env `b` -- the draw layout shared with the oracle
struct ResetDraw {
  uint64_t seed, counter;
  int path_first, path_count;
  int testing;  // is_testing_mode: the candidate range starts at the path's beginning and grows with the tries
  int env_base; // sigmaenv_config_t.env_index_base: the generator is keyed on the env's index in the WHOLE batch
};
// exclusive upper end of the centre-line points try `tr` may draw from (world_state_rt_sim.py:253-263; specification shared with the oracle)
__device__ __forceinline__ int reset_end_point(int testing, int tr, int n) {
  const int half = n / 2;
  int end = half;
  if (testing) {
    const long long grow = 3 + (long long)(tr + 1) * (tr + 2) / 2;
    end = grow < half ? (int)grow : half;
  }
  return end < 4 ? 4 : end;
}
__device__ __forceinline__ void reset_candidate(const DevMap& m, const ResetDraw& rd, int b, int i, int tr, int& path, int& pt, float& px, float& py,
                                                uint32_t draw0 = 0u) {
  path = rd.path_first + (int)__umulhi(rng_u32(rd.seed, rd.counter, (uint32_t)(rd.env_base + b), (uint32_t)i, draw0 + 2u * tr), (uint32_t)rd.path_count);
  int n = m.n_center[path];
  const int end = reset_end_point(rd.testing, tr, n);
  pt = 3 + (int)__umulhi(rng_u32(rd.seed, rd.counter, (uint32_t)(rd.env_base + b), (uint32_t)i, draw0 + 2u * tr + 1u), (uint32_t)(end - 3));
  const float2 xy = reinterpret_cast<const float2*>(m.center)[(size_t)path * m.P + pt];
  px = xy.x;
  py = xy.y;
}
// cpm_mixed: the path list of sub-scenario `sid` (1-based; any other id -- 0 after a host-placed start -- takes the LAST list, the reference's
// `if 1 / elif 2 / else` chain, world_state_rt_sim.py:345-356), and the sub-scenario a finished env draws
// (torch.multinomial(cpm_scenario_probabilities), world_state_rt_sim.py:330-343 -- here the counter-based generator's draw 5000 of agent 0 against the
// cumulative distribution; specification shared with the oracle)
__device__ __forceinline__ void scenario_list(const DevMap& m, int sid, int& first, int& count) {
  const int k = (sid >= 1 && sid <= m.n_lists) ? sid - 1 : (m.n_lists > 0 ? m.n_lists - 1 : 0);
  first = (int)((m.list_first16 >> (16 * k)) & 0xFFFFull);
  count = (int)((m.list_count16 >> (16 * k)) & 0xFFFFull);
}
__device__ __forceinline__ int draw_scenario(const DevMap& m, uint64_t seed, uint64_t counter, int env_global) {
  const float u = (float)(rng_u32(seed, counter, (uint32_t)env_global, 0u, 5000u) >> 8) * (1.0f / 16777216.0f);
  int sid = m.n_lists;
  if (m.n_lists > 3 && u < m.cdf2) sid = 3;
  if (m.n_lists > 2 && u < m.cdf1) sid = 2;
  if (m.n_lists > 1 && u < m.cdf0) sid = 1;
  return sid;
}
// the tries a wavefront evaluates up front for a finished env: lane -> (agent, try), 64/N tries per agent
struct ResetPrefetch {
  int path, pt;
  float x, y;
  bool have;  // the first env of this wavefront (e == wave) has its candidates here already
};

template <bool WAVE = false, bool VARIANTS = true>
__device__ __forceinline__ void auto_reset_tile(const sigmaenv_config_t& c, const DevMap& m, const DevBufs& g, const Smem& s, const Tile& t,
                                       const unsigned long long* s_mask, const int* s_full, uint64_t seed, uint64_t counter, int path_first,
                                       int path_count, int obs_mode, const ResetPrefetch& pre, int g_cap = 64, int* lds_tim = nullptr, const CfgDerived* dvp = nullptr);
#define MAX_G 64

// ---------------------------------------------------------------------------------------------------------------------
// observation only (observation() called again after resets)
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline void load_tile_for_observation(const Smem& s, const DevBufs& g, const Tile& t) {
  const int N = t.N;
  for (int k = threadIdx.x; k < t.slots * 8; k += blockDim.x) s.st[k] = g.state[t.a0 * 8 + k];
  for (int k = threadIdx.x; k < t.slots * 10; k += blockDim.x) s.vnew[k] = g.vertices[t.a0 * 10 + k];
  for (int k = threadIdx.x; k < t.slots * NS * 2; k += blockDim.x) s.shrt[k] = g.short_term[t.a0 * NS * 2 + k];
  for (int k = threadIdx.x; k < t.slots; k += blockDim.x) s.dref[k] = g.dist_ref[t.a0 + k];
  for (int k = threadIdx.x; k < t.slots * 5; k += blockDim.x) { s.dleft[k] = g.dist_left[t.a0 * 5 + k]; s.dright[k] = g.dist_right[t.a0 * 5 + k]; }
  for (int k = threadIdx.x; k < t.slots * N; k += blockDim.x) s.dist[(k / N) * DIST_STRIDE(N) + (k % N)] = g.dist_agents[t.a0 * N + k];
  for (int k = threadIdx.x; k < t.slots; k += blockDim.x) {
    float psi = g.state[(t.a0 + k) * 8 + 2];
    s.cs[k * 2] = cr_cos(psi);
    s.cs[k * 2 + 1] = cr_sin(psi);
    // (read by the non-default observation rows only: boundary points around the closest boundary point of the agent's path)
    s.path[k] = g.path[(t.a0 + k) * 4];
    s.cp[k * 3 + 0] = g.closest[(t.a0 + k) * 3 + 0]; s.cp[k * 3 + 1] = g.closest[(t.a0 + k) * 3 + 1]; s.cp[k * 3 + 2] = g.closest[(t.a0 + k) * 3 + 2];
    s.fresh[k] = g.fresh ? g.fresh[t.a0 + k] : 0;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256) sigmaenv_observe_kernel(sigmaenv_config_t c, DevBufs g, int G) {
  sigma_poison_lds();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const Tile t(c, G);
  Smem s(smem_raw, G * t.N, t.N, t.K, t.DL);
  load_tile_for_observation(s, g, t);
  observe_tile(c, s, g, t);
}

// ---------------------------------------------------------------------------------------------------------------------
// reset: (1) scatter host-chosen entries, (2) rebuild the derived state of marked agents / envs
// ---------------------------------------------------------------------------------------------------------------------
__global__ void sigmaenv_reset_scatter_kernel(DevBufs g, int N, int n, const int32_t* env_idx, const int32_t* agent_idx, const int32_t* path_ids,
                                              const float* state8, int full_env) {
  sigma_poison_lds();
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int b = env_idx[k], i = agent_idx[k];
  size_t bi = (size_t)b * N + i;
  for (int q = 0; q < 8; ++q) g.state[bi * 8 + q] = state8[(size_t)k * 8 + q];
  for (int q = 0; q < 4; ++q) g.path[bi * 4 + q] = path_ids[(size_t)k * 4 + q];
  if (g.fresh) g.fresh[bi] = 1;
  atomicOr(&g.reset_mask[b], 1ull << i);
  if (full_env) g.reset_full[b] = 1;
}

// derived state of every marked agent (reset_init_distances_and_short_term_ref_path, world_state_rt.py:422-529) and the
// per-env tail (road_traffic.py:902-923) for the envs of the tile whose bit is set in env_bits; with_obs: also a fresh
// observation of the whole tile.  Expects s.st / s.path / s.vnew / s.cp (scan guesses) of the tile in LDS; agent_mask[e] is the
// per-env agent bit mask, full[e] the full-env flag.  All threads of the block participate.
template <bool WAVE = false, bool VARIANTS = true>
__device__ inline void reset_finish_body(const sigmaenv_config_t& c, const DevBufs& g, const Smem& s, const Tile& t,
                                         const unsigned long long* agent_mask, const int* full, int with_obs, int g_cap = 64, int* lds_tim = nullptr,
                                         const CfgDerived* dvp = nullptr);
__device__ inline void reset_derive_body(const sigmaenv_config_t& c, const DevMap& m, const DevBufs& g, const Smem& s, const Tile& t,
                                         const unsigned long long* agent_mask, const int* full, int with_obs) {
  const int N = t.N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
#define TS2(k) PROF_TS2(g, tid, k)
  for (int sl = tid; sl < t.slots; sl += blockDim.x) {
    int e = sl / N, i = sl - e * N;
    if ((agent_mask[e] >> i) & 1ull) {
      float v[10];
      rect_vertices(c, s.st[sl * 8], s.st[sl * 8 + 1], s.st[sl * 8 + 2], v, &s.cs[sl * 2]);
#pragma unroll
      for (int k = 0; k < 10; ++k) { s.vnew[sl * 10 + k] = v[k]; g.vertices[(t.a0 + sl) * 10 + k] = v[k]; }
    }
  }
  __syncthreads();
  if (m.nch > 0) {
    for (int task = tid; task < t.slots * 3; task += blockDim.x) {
      int sl = task / 3;
      if ((agent_mask[sl / N] >> (sl % N)) & 1ull) scan_mask_task<false>(m, s, task, false, N);
    }
    __syncthreads();
    TS2(3);
    for (int pr = wave; 2 * pr < t.slots; pr += n_waves) {
      int sa = 2 * pr, sb = 2 * pr + 1;
      bool ma = (agent_mask[sa / N] >> (sa % N)) & 1ull;
      bool mb = (sb < t.slots) && ((agent_mask[sb / N] >> (sb % N)) & 1ull);
      if (!(ma || mb)) continue;
      if (m.fast_div) pair_scan<false, true>(m, c, s, sa, (ma ? 1 : 0) | (mb ? 2 : 0), lane, false, N);
      else pair_scan<false, false>(m, c, s, sa, (ma ? 1 : 0) | (mb ? 2 : 0), lane, false, N);
    }
  } else {
    for (int sl = wave; sl < t.slots; sl += n_waves) {
      int e = sl / N, i = sl - e * N;
      if (!((agent_mask[e] >> i) & 1ull)) continue;
      AgentScan r;
      agent_scan_full<false>(m, c, s.path[sl], lane, s.st[sl * 8], s.st[sl * 8 + 1], s.vnew + sl * 10, s.vnew + sl * 10, r);
      if (lane == 0) {
        s.dref[sl] = r.d_ref;
#pragma unroll
        for (int q = 0; q < 5; ++q) { s.dleft[sl * 5 + q] = r.dl[q]; s.dright[sl * 5 + q] = r.dr[q]; }
        s.cp[sl * 3 + 0] = r.cp_ref; s.cp[sl * 3 + 1] = r.cp_l; s.cp[sl * 3 + 2] = r.cp_r;
      }
    }
  }
  __syncthreads();
  TS2(4);
  for (int sl = tid; sl < t.slots; sl += blockDim.x) {  // only the marked agents' derived state is replaced
    int e = sl / N, i = sl - e * N;
    if (!((agent_mask[e] >> i) & 1ull)) continue;
    const size_t gi = t.a0 + sl;
    float mbnd = INFINITY;
    g.dist_ref[gi] = s.dref[sl];
#pragma unroll
    for (int q = 0; q < 5; ++q) { g.dist_left[gi * 5 + q] = s.dleft[sl * 5 + q]; g.dist_right[gi * 5 + q] = s.dright[sl * 5 + q]; }
#pragma unroll
    for (int q = 0; q < 5; ++q) mbnd = fminf(mbnd, s.dleft[sl * 5 + q]);
#pragma unroll
    for (int q = 0; q < 5; ++q) mbnd = fminf(mbnd, s.dright[sl * 5 + q]);
    g.dist_bound[gi] = mbnd;
    g.closest[gi * 3 + 0] = s.cp[sl * 3 + 0]; g.closest[gi * 3 + 1] = s.cp[sl * 3 + 1]; g.closest[gi * 3 + 2] = s.cp[sl * 3 + 2];
    int path = s.path[sl];
    float sp[NS * 2];
    short_term_path(m.center + (size_t)path * m.P * 2, m.n_center[path], m.is_loop[path] != 0, s.cp[sl * 3], sp);
#pragma unroll
    for (int k = 0; k < NS * 2; ++k) { g.short_term[gi * NS * 2 + k] = sp[k]; s.shrt[sl * NS * 2 + k] = sp[k]; }
  }
#undef TS2
  reset_finish_body<false>(c, g, s, t, agent_mask, full, with_obs);
}

// tail of every touched env: mutual distances, collisions cleared, prev_pos := pos, timer (road_traffic.py:902-923); with_obs:
// also a fresh observation of the whole tile (2: every input of it is already in LDS).  Expects the derived state of the marked
// agents in LDS and HBM.
template <bool WAVE, bool VARIANTS>
__device__ inline void reset_finish_body(const sigmaenv_config_t& c, const DevBufs& g, const Smem& s, const Tile& t,
                                         const unsigned long long* agent_mask, const int* full, int with_obs, int g_cap, int* lds_tim, const CfgDerived* dvp) {
  const int N = t.N;
  const int tid = Grp<WAVE>::tid();
#define TS2(k) PROF_TS2(g, tid, k)
  const float diag = dvp ? dvp->diag : sqrtf(c.world_x_dim * c.world_x_dim + c.world_y_dim * c.world_y_dim);
  for (int p = tid; p < t.slots * N; p += Grp<WAVE>::size()) {
    int si = fdiv(p, g.mN), j = p - si * N;
    int e = fdiv(si, g.mN);
    if (agent_mask[e] == 0ull) continue;
    int sj = e * N + j;
    float d = (si == sj) ? diag : pair_distance(c, s.st, s.vnew, si, sj);
    s.dist[si * DIST_STRIDE(N) + j] = d;
    g.dist_agents[t.a0 * N + p] = d;
    g.col_agents[t.a0 * N + p] = 0;
  }
  for (int sl = tid; sl < t.slots; sl += Grp<WAVE>::size()) {
    int e = sl / N, i = sl - e * N;
    if (agent_mask[e] == 0ull) continue;
    const size_t gi = t.a0 + sl;
    reinterpret_cast<uchar4*>(g.col_flags)[gi] = make_uchar4(0, 0, 0, 0);
    reinterpret_cast<float2*>(g.prev_pos)[gi] = make_float2(s.st[sl * 8], s.st[sl * 8 + 1]);
    if (full[e]) reinterpret_cast<float2*>(g.action)[gi] = make_float2(0.f, 0.f);
    if (i == 0) {
      int b = t.env0 + e;
      if (full[e]) {
        g.timer[b * 4] = 0; g.timer[b * 4 + 3] += 1; g.done[b] = 0;
        if (lds_tim) { lds_tim[e * 4] = 0; lds_tim[e * 4 + 3] += 1; }  // the step loop's LDS copy of the timer rows
      }
      g.reset_mask[b] = 0ull;
      g.reset_full[b] = 0;
    }
  }
  if (with_obs) {
    __threadfence_block();
    Grp<WAVE>::sync();
    TS2(5);
    if (with_obs == 2 && t.nenv == 1 && agent_mask[0]) {
      // a one-env tile (the 16 x 1 instantiations know it at compile time): the touched env IS the tile -- no env list, no slot indirection
      observe_tile<WAVE, VARIANTS>(c, s, g, t, -1, nullptr, 0, true, lds_tim, dvp);
    } else if (with_obs == 2) {
      // every input of the observation is in LDS (fused tail / a tile whose envs were all reset): only the touched envs' rows change
      int* env_sel = const_cast<int*>(full) + g_cap + 2;  // after s_full, s_any and the step kernel's pair counter
      if (tid == 0) {
        int cnt = 0;
        for (int e = 0; e < t.nenv; ++e) if (agent_mask[e]) env_sel[1 + cnt++] = e;
        env_sel[0] = cnt;
      }
      Grp<WAVE>::sync();
      observe_tile<WAVE, VARIANTS>(c, s, g, t, -1, env_sel + 1, env_sel[0], true, lds_tim, dvp);
    } else {
      load_tile_for_observation(s, g, t);
      observe_tile<WAVE, VARIANTS>(c, s, g, t);
    }
    TS2(6);
  }
#undef TS2
}

// host-driven resets: only tiles with marked agents do work
__global__ void __launch_bounds__(512) sigmaenv_reset_derive_kernel(sigmaenv_config_t c, DevMap m, DevBufs g, int with_obs, int G) {
  sigma_poison_lds();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const Tile t(c, G);
  const int N = t.N;
  // all LDS lives in the one dynamic region (keeps its base 16-byte aligned): [Smem | masks | full flags | any]
  unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(smem_raw + ((Smem::bytes(G * N, N, t.K, t.DL) + 15) & ~(size_t)15));
  int* s_full = reinterpret_cast<int*>(s_mask + MAX_G);
  int* s_any = s_full + MAX_G;
  if (threadIdx.x == 0) *s_any = 0;
  __syncthreads();
  if (threadIdx.x < t.nenv) {
    unsigned long long mk = g.reset_mask[t.env0 + threadIdx.x];
    s_mask[threadIdx.x] = mk;
    s_full[threadIdx.x] = g.reset_full[t.env0 + threadIdx.x];
    if (mk) *s_any = 1;
  }
  __syncthreads();
  if (!*s_any) return;
  Smem s(smem_raw, G * N, N, t.K, t.DL);
  for (int k = threadIdx.x; k < t.slots * 8; k += blockDim.x) s.st[k] = g.state[t.a0 * 8 + k];
  for (int k = threadIdx.x; k < t.slots * 10; k += blockDim.x) s.vnew[k] = g.vertices[t.a0 * 10 + k];
  for (int k = threadIdx.x; k < t.slots; k += blockDim.x) {
    s.path[k] = g.path[(t.a0 + k) * 4];
    s.fresh[k] = g.fresh ? g.fresh[t.a0 + k] : 0;
    int pt = g.path[(t.a0 + k) * 4 + 3];  // the agent was placed at (or near) this centre-line point: guess for the pruned scan
    s.cp[k * 3 + 0] = pt; s.cp[k * 3 + 1] = pt; s.cp[k * 3 + 2] = pt;
  }
  __syncthreads();
  reset_derive_body(c, m, g, s, t, s_mask, s_full, with_obs);
}

// Start table (DevMap::start_table): one workgroup derives 64 consecutive (path, point) rows with the same code the resets use
// (rect_vertices, scan_mask_task, pair_scan, short_term_path), so that copying a row is bit-identical to deriving it in place.
__global__ void __launch_bounds__(256) sigmaenv_start_table_kernel(sigmaenv_config_t c, DevMap m, float* table) {
  sigma_poison_lds();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int S = 64;
  Smem s(smem_raw, S, 1, 0, 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
  __shared__ unsigned long long s_valid;
  __shared__ float s_cosv[64], s_sinv[64];
  const int total = m.n_paths * m.P;
  const int idx = blockIdx.x * S + lane;
  const int path = idx < total ? idx / m.P : 0, pt = idx < total ? idx - (idx / m.P) * m.P : 0;
  const bool valid = idx < total && pt < m.n_center[path];
  if (wave == 0) {
    const unsigned long long vb = __ballot(valid);
    if (lane == 0) s_valid = vb;
    const int sl = lane;
    float x = 0.f, y = 0.f, rot = 0.f;
    if (valid) {
      x = m.center[((size_t)path * m.P + pt) * 2];
      y = m.center[((size_t)path * m.P + pt) * 2 + 1];
      rot = m.yaw[(size_t)path * m.yaw_stride + (pt < m.yaw_stride ? pt : m.yaw_stride - 1)];
    }
    s.st[sl * 8] = x; s.st[sl * 8 + 1] = y; s.st[sl * 8 + 2] = rot;
    float v[10];
    rect_vertices(c, x, y, rot, v, &s.cs[sl * 2]);
#pragma unroll
    for (int k = 0; k < 10; ++k) s.vnew[sl * 10 + k] = v[k];
    s.path[sl] = path;
    s.cp[sl * 3] = pt; s.cp[sl * 3 + 1] = pt; s.cp[sl * 3 + 2] = pt;  // the scan guesses of a reset (sigmaenv_auto_reset)
    s_cosv[sl] = cr_cos(0.0f + rot);
    s_sinv[sl] = cr_sin(0.0f + rot);
  }
  __syncthreads();
  const unsigned long long vm = s_valid;
  if (m.nch > 0) {
    for (int task = tid; task < S * 3; task += blockDim.x) {
      if ((vm >> (task / 3)) & 1ull) scan_mask_task<false>(m, s, task, false, 1);
    }
    __syncthreads();
    for (int pr = wave; 2 * pr < S; pr += n_waves) {
      const int va = (int)((vm >> (2 * pr)) & 1ull), vb = (int)((vm >> (2 * pr + 1)) & 1ull);
      if (!(va || vb)) continue;
      if (m.fast_div) pair_scan<false, true>(m, c, s, 2 * pr, va | (vb << 1), lane, false, 1);
      else pair_scan<false, false>(m, c, s, 2 * pr, va | (vb << 1), lane, false, 1);
    }
  } else {
    for (int sl = wave; sl < S; sl += n_waves) {
      if (!((vm >> sl) & 1ull)) continue;
      AgentScan r;
      agent_scan_full<false>(m, c, s.path[sl], lane, s.st[sl * 8], s.st[sl * 8 + 1], s.vnew + sl * 10, s.vnew + sl * 10, r);
      if (lane == 0) {
        s.dref[sl] = r.d_ref;
#pragma unroll
        for (int q = 0; q < 5; ++q) { s.dleft[sl * 5 + q] = r.dl[q]; s.dright[sl * 5 + q] = r.dr[q]; }
        s.cp[sl * 3 + 0] = r.cp_ref; s.cp[sl * 3 + 1] = r.cp_l; s.cp[sl * 3 + 2] = r.cp_r;
      }
    }
  }
  __syncthreads();
  if (wave == 0 && idx < total) {
    const int sl = lane;
    float* row = table + (size_t)idx * START_ROW;
    for (int k = 0; k < START_ROW; ++k) row[k] = 0.0f;
    if (valid) {
      row[START_X] = s.st[sl * 8]; row[START_X + 1] = s.st[sl * 8 + 1]; row[START_X + 2] = s.st[sl * 8 + 2];
      row[START_COSV] = s_cosv[sl]; row[START_COSV + 1] = s_sinv[sl];
      for (int k = 0; k < 10; ++k) row[START_VERT + k] = s.vnew[sl * 10 + k];
      row[START_CS] = s.cs[sl * 2]; row[START_CS + 1] = s.cs[sl * 2 + 1];
      row[START_DREF] = s.dref[sl];
      float mbnd = INFINITY;
      for (int q = 0; q < 5; ++q) { row[START_DLEFT + q] = s.dleft[sl * 5 + q]; row[START_DRIGHT + q] = s.dright[sl * 5 + q]; }
      for (int q = 0; q < 5; ++q) mbnd = fminf(mbnd, s.dleft[sl * 5 + q]);
      for (int q = 0; q < 5; ++q) mbnd = fminf(mbnd, s.dright[sl * 5 + q]);
      row[START_DBOUND] = mbnd;
      float sp[NS * 2];
      short_term_path(m.center + (size_t)path * m.P * 2, m.n_center[path], m.is_loop[path] != 0, s.cp[sl * 3], sp);
      for (int k = 0; k < NS * 2; ++k) row[START_SHORT + k] = sp[k];
      for (int k = 0; k < 3; ++k) row[START_CP + k] = __int_as_float(s.cp[sl * 3 + k]);
    }
  }
}

// place agent slot `sl` on centre-line point `pt` of path `path` with speed `speed`: state and every derived tensor of the agent
// come from the start table (LDS and HBM), as reset + reset_init_distances_and_short_term_ref_path leave them
// (world_state_rt_sim.py:189-213, world_state_rt.py:422-529)
__device__ __forceinline__ void place_from_start_table(const DevMap& m, const DevBufs& g, const Smem& s, const Tile& t, int sl, int path, int pt,
                                                       float speed, int path_first, bool full_env, int scenario_id = 0) {
  const float4* row4 = reinterpret_cast<const float4*>(m.start_table + ((size_t)path * m.P + pt) * START_ROW);
  float r[START_ROW];
#pragma unroll
  for (int k = 0; k < START_ROW / 4; ++k) { float4 q = row4[k]; r[4 * k] = q.x; r[4 * k + 1] = q.y; r[4 * k + 2] = q.z; r[4 * k + 3] = q.w; }
  const size_t gi = t.a0 + sl;
  const float st[8] = {r[START_X], r[START_X + 1], r[START_X + 2], speed, 0.0f, speed * r[START_COSV], speed * r[START_COSV + 1], 0.0f};
  float4* gs = reinterpret_cast<float4*>(g.state + gi * 8);
  gs[0] = make_float4(st[0], st[1], st[2], st[3]);
  gs[1] = make_float4(st[4], st[5], st[6], st[7]);
#pragma unroll
  for (int k = 0; k < 8; ++k) s.st[sl * 8 + k] = st[k];
  float2* gv = reinterpret_cast<float2*>(g.vertices + gi * 10);
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    s.vnew[sl * 10 + 2 * k] = r[START_VERT + 2 * k]; s.vnew[sl * 10 + 2 * k + 1] = r[START_VERT + 2 * k + 1];
    gv[k] = make_float2(r[START_VERT + 2 * k], r[START_VERT + 2 * k + 1]);
  }
  s.cs[sl * 2] = r[START_CS]; s.cs[sl * 2 + 1] = r[START_CS + 1];
  // what the next bicycle step of this launch starts from (phase A): steering 0 -> tan 0, sideslip 0, and cos / sin(yaw + 0): the start table's cos / sin of
  // the yaw (same function, same argument; `+ 0.0f` turns the sine of a yaw of -0.0 into the +0.0 that sin(-0.0 + 0.0) is)
  s.carry[sl * 3] = 0.0f; s.carry[sl * 3 + 1] = r[START_CS]; s.carry[sl * 3 + 2] = r[START_CS + 1] + 0.0f;
  s.dref[sl] = r[START_DREF];
  g.dist_ref[gi] = r[START_DREF];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    s.dleft[sl * 5 + q] = r[START_DLEFT + q]; s.dright[sl * 5 + q] = r[START_DRIGHT + q];
    g.dist_left[gi * 5 + q] = r[START_DLEFT + q]; g.dist_right[gi * 5 + q] = r[START_DRIGHT + q];
  }
  s.dbound[sl] = r[START_DBOUND];
  g.dist_bound[gi] = r[START_DBOUND];
  float2* gso = reinterpret_cast<float2*>(g.short_term + gi * NS * 2);
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    s.shrt[sl * NS * 2 + 2 * k] = r[START_SHORT + 2 * k]; s.shrt[sl * NS * 2 + 2 * k + 1] = r[START_SHORT + 2 * k + 1];
    gso[k] = make_float2(r[START_SHORT + 2 * k], r[START_SHORT + 2 * k + 1]);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) { const int cpk = __float_as_int(r[START_CP + k]); s.cp[sl * 3 + k] = cpk; g.closest[gi * 3 + k] = cpk; }
  s.path[sl] = path;
  s.fresh[sl] = 1;
  if (g.fresh) g.fresh[gi] = 1;
  g.path[gi * 4 + 0] = path;
  if (full_env) g.path[gi * 4 + 1] = scenario_id;  // (kept by a per-agent reset; 0 unless cpm_mixed)
  g.path[gi * 4 + 2] = path - path_first;
  g.path[gi * 4 + 3] = pt;
}

// Device-side reset of the marked envs / agents of one tile (s_mask[e]: agents to re-place, s_full[e]: the env is finished and
// restarts as a whole).  Expects s.st / s.vnew / s.path of the whole tile in LDS.  One wavefront per env runs the rejection
// sampler of world_state_rt_sim.py:215-311 (non-testing mode) from a counter-based RNG: the lanes evaluate 64/N tries of every
// agent up front and the FIRST feasible try wins, which is exactly the sequential loop's result for the same draws (bounded to
// 64 tries; the reference loops without bound).  Then the deterministic reset as in sigmaenv_reset and a fresh observation.
// All threads of the block participate.
template <bool WAVE, bool VARIANTS>
__device__ __forceinline__ void auto_reset_tile(const sigmaenv_config_t& c, const DevMap& m, const DevBufs& g, const Smem& s, const Tile& t,
                                       const unsigned long long* s_mask, const int* s_full, uint64_t seed, uint64_t counter, int path_first,
                                       int path_count, int obs_mode, const ResetPrefetch& pre, int g_cap, int* lds_tim, const CfgDerived* dvp) {
  const int N = t.N;
  const int tid = Grp<WAVE>::tid(), lane = tid & 63, wave = tid >> 6, n_waves = Grp<WAVE>::size() >> 6;
  const bool mixed = path_count < 0;  // SIGMAENV_SCENARIO_LISTS: every env draws from the path list of ITS sub-scenario
  ResetDraw rd{seed, counter, path_first, path_count, c.is_testing_mode, c.env_index_base};
#define TS2(k) PROF_TS2(g, tid, k)
  TS2(1);
  float min_d_sq;  // road_traffic.py:679-684
  if (dvp) {
    min_d_sq = dvp->min_d_sq;
  } else {
    const float min_d = sqrtf((float)((double)c.length * (double)c.length + (double)c.width * (double)c.width)) * 1.5f;
    min_d_sq = min_d * min_d;
  }
  const int TR = 64 / N > 0 ? 64 / N : 1;  // tries per agent evaluated up front (all agents at once, loads in flight together)
  for (int e = wave; e < t.nenv; e += n_waves) {
    const int b = t.env0 + e;
    if (!s_full[e]) {
      // per-agent resets of an unfinished env (agents that left through an entry/exit, or collided in testing mode): agents in
      // index order, each against ALL other agents' current positions (is_reset_single_agent, world_state_rt_sim.py:287-309);
      // the 64 lanes evaluate tries 0..63 (draws 2000 + 2t, 2001 + 2t), first feasible wins, else the last one
      unsigned long long rq = s_mask[e];
      while (rq) {
        const int i = __ffsll((long long)rq) - 1;
        rq &= rq - 1ull;
        const int sl = e * N + i;
        int p2, q2;
        float x2, y2;
        if (mixed) scenario_list(m, g.path[(t.a0 + sl) * 4 + 1], rd.path_first, rd.path_count);  // the agent keeps its env's sub-scenario (:325-328)
        reset_candidate(m, rd, b, i, lane, p2, q2, x2, y2, 2000u);
        bool ok = true;
        for (int j = 0; j < N; ++j) {
          if (j == i) continue;
          float dx = x2 - s.st[(e * N + j) * 8], dy = y2 - s.st[(e * N + j) * 8 + 1];
          float d2 = dx * dx + dy * dy;
          if (!(d2 >= min_d_sq)) ok = false;
        }
        unsigned long long f2 = __ballot(ok);
        const int wl = f2 ? (__ffsll((long long)f2) - 1) : (AUTO_RESET_MAX_TRIES - 1);
        if (lane == wl) {
          float u = (float)(rng_u32(seed, counter, (uint32_t)(c.env_index_base + b), (uint32_t)i, 3000u) >> 8) * (1.0f / 16777216.0f);
          place_from_start_table(m, g, s, t, sl, p2, q2, u * c.max_speed, rd.path_first, false);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      continue;
    }
    int sid = 0;
    if (mixed) {
      sid = draw_scenario(m, seed, counter, c.env_index_base + b);
      scenario_list(m, sid, rd.path_first, rd.path_count);
    }
    const int ca = lane / TR, ctr = lane - ca * TR;  // this lane's (agent, try)
    const bool has_c = ca < N;
    int cpath = 0, cpt = 3;
    float cx = 0.f, cy = 0.f;
    if (pre.have && e == wave) { cpath = pre.path; cpt = pre.pt; cx = pre.x; cy = pre.y; }
    else if (has_c) reset_candidate(m, rd, b, ca, ctr, cpath, cpt, cx, cy);
    bool cok = has_c;  // still feasible w.r.t. every agent accepted so far
    int my_path = 0, my_pt = 3;  // lane i keeps agent i's accepted start (registers only: the common case touches LDS once, after the loop)
    for (int i = 0; i < N; ++i) {
      unsigned long long feas = __ballot(cok && ca == i);
      int path, pt;
      float px, py;
      if (feas) {  // first feasible among the precomputed tries (they are tries 0..TR-1 in lane order)
        const int wl = __builtin_amdgcn_readfirstlane(__ffsll((long long)feas) - 1);
        path = __builtin_amdgcn_readlane(cpath, wl); pt = __builtin_amdgcn_readlane(cpt, wl);
        px = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cx), wl)); py = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cy), wl));
      } else {     // rare: tries TR..63 of this agent, all lanes at once; none feasible -> the last try (as the bounded loop would)
        int p2, q2;
        float x2, y2;
        reset_candidate(m, rd, b, i, lane, p2, q2, x2, y2);
        bool ok = lane >= TR;
        wave_sync();  // the accepted positions of agents 0..i-1 (written below by lane 0)
        for (int j = 0; j < i; ++j) {
          float dx = x2 - s.st[(e * N + j) * 8], dy = y2 - s.st[(e * N + j) * 8 + 1];
          float d2 = dx * dx + dy * dy;
          if (!(d2 >= min_d_sq)) ok = false;
        }
        unsigned long long f2 = __ballot(ok);
        const int wl = __builtin_amdgcn_readfirstlane(f2 ? (__ffsll((long long)f2) - 1) : (AUTO_RESET_MAX_TRIES - 1));
        path = __builtin_amdgcn_readlane(p2, wl); pt = __builtin_amdgcn_readlane(q2, wl);
        px = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x2), wl)); py = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y2), wl));
      }
      if (lane == i) { my_path = path; my_pt = pt; }
      if (lane == 0) { s.st[(e * N + i) * 8] = px; s.st[(e * N + i) * 8 + 1] = py; }
      // later agents must keep the minimum distance to this one (world_state_rt_sim.py:296-309)
      float dx = cx - px, dy = cy - py;
      float d2 = dx * dx + dy * dy;
      if (ca > i && !(d2 >= min_d_sq)) cok = false;
    }
    if (lane < N) {  // finalise the accepted starts, one lane per agent (world_state_rt_sim.py:189-213)
      const int i = lane, sl = e * N + i;
      float u = (float)(rng_u32(seed, counter, (uint32_t)(c.env_index_base + b), (uint32_t)i, 1000u) >> 8) * (1.0f / 16777216.0f);
      place_from_start_table(m, g, s, t, sl, my_path, my_pt, u * c.max_speed, rd.path_first, true, sid);
    }
  }
  __threadfence_block();
  Grp<WAVE>::sync();
  TS2(2);
  reset_finish_body<WAVE, VARIANTS>(c, g, s, t, s_mask, s_full, obs_mode, g_cap, lds_tim, dvp);
#undef TS2
}

#include "sigmaenv_step_wave.inc"

// stand-alone launch: one workgroup per tile; tiles without a finished env / a reset request exit at once
__global__ void __launch_bounds__(512) sigmaenv_auto_reset_kernel(sigmaenv_config_t c, DevMap m, DevBufs g, uint64_t seed, uint64_t counter,
                                                                  int path_first, int path_count, int G) {
  sigma_poison_lds();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const Tile t(c, G);
  const int N = t.N;
  const int tid = threadIdx.x;
  unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(smem_raw + ((Smem::bytes(G * N, N, t.K, t.DL) + 15) & ~(size_t)15));
  int* s_full = reinterpret_cast<int*>(s_mask + MAX_G);
  int* s_any = s_full + MAX_G;
#define TS2(k) PROF_TS2(g, tid, k)
#ifdef SIGMAENV_PROFILE
  if (g.dbg_ts2 && tid < 16) g.dbg_ts2[(size_t)blockIdx.x * 16 + tid] = 0ull;
#endif
  TS2(0);
#undef TS2
  if (tid == 0) *s_any = 0;
  __syncthreads();
  if (tid < t.nenv) {
    const int dn = g.done[t.env0 + tid];
    unsigned long long rq = 0ull;  // per-agent reset requests of an unfinished env (road_traffic.py:1435-1447, :1456-1473)
    if (!dn && (c.is_testing_mode || c.has_entry_exit)) {  // no other configuration ever raises a request
      for (int i = 0; i < N; ++i) rq |= (unsigned long long)(g.col_flags[((size_t)(t.env0 + tid) * N + i) * 4 + 3] != 0) << i;
    }
    s_mask[tid] = dn ? ((N >= 64) ? ~0ull : ((1ull << N) - 1ull)) : rq;
    s_full[tid] = dn;
    if (dn || rq) *s_any = 1;
  }
  __syncthreads();
  if (!*s_any) return;
  Smem s(smem_raw, G * N, N, t.K, t.DL);
  // untouched envs of the tile keep their state (needed for the tile-wide observation)
  for (int k = tid; k < t.slots * 8; k += blockDim.x) s.st[k] = g.state[t.a0 * 8 + k];
  for (int k = tid; k < t.slots * 10; k += blockDim.x) s.vnew[k] = g.vertices[t.a0 * 10 + k];
  for (int k = tid; k < t.slots; k += blockDim.x) { s.path[k] = g.path[(t.a0 + k) * 4]; s.fresh[k] = g.fresh ? g.fresh[t.a0 + k] : 0; }
  __syncthreads();
  ResetPrefetch none;
  none.have = false;
  auto_reset_tile(c, m, g, s, t, s_mask, s_full, seed, counter, path_first, path_count, (G == 1 && s_full[0]) ? 2 : 1, none);
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
  int reset_block = 256;
  int G = 1;      // environments per workgroup (G * N <= 64 agent slots) of the block kernels (reset / observe)
  int wave_G = 1, wave_wpb = 1, wave_grid = 1;  // step kernel: environments per wavefront tile, wavefronts per workgroup, workgroups
  size_t wave_tile_lds = 0;
  int wave_spec = 0;  // agents * 256 + envs per wavefront of a fixed-shape instantiation of the step kernel (2 observed neighbours), else 0
  sigmaenv_config_t* d_cfg = nullptr;  // device copy of cfg (the step kernel reads it through scalar loads): a DevConfig -- the config, then the derived block
  DevConfig cfg_derived;
  uint32_t observe_calls = 0;      // stand-alone sigmaenv_observe calls so far (the salt of their sensor noise)
  size_t rollout_slab_stride = 0;  // floats between the record blocks of consecutive steps of sigmaenv_rollout* (0: B * W)
  int grid = 1;
  void* bufs[SIGMAENV_BUF_COUNT] = {nullptr};
  size_t buf_bytes[SIGMAENV_BUF_COUNT] = {0};
  // reset staging
  int32_t *d_env_idx = nullptr, *d_agent_idx = nullptr, *d_path_ids = nullptr;
  float* d_state8 = nullptr;
  size_t staging_cap = 0;
  // kernel timing (HIP events on the handle's stream around a sample of the launches), one pool per timed kernel (SIGMAENV_KERNEL_*)
  struct KernelTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    std::vector<int> used;
    unsigned long long launch_count = 0;
  };
  KernelTimer timers[SIGMAENV_KERNEL_COUNT];
  bool timing = false;
  int timing_stride = 8;
  // QP-free CBF margin reward (sigmaenv_cbf.inc)
  sigmaenv_cbf_config_t cbf_cfg{};
  void *cbf_seg4 = nullptr, *cbf_segl = nullptr, *cbf_cxy = nullptr, *cbf_u = nullptr, *cbf_kin = nullptr, *cbf_clf = nullptr, *cbf_safe = nullptr;
  int cbf_seg_stride = 0;
  const float* cbf_centers_inject = nullptr;  // test hook (sigmaenv_cbf_inject_centers): device [B,N,C,2] circle centres that replace the computed ones
  float* lanelet_centers = nullptr;           // [n_lanelets, lanelet_pts, 2] zero-padded centre lines (sigmaenv_set_lanelets)
  unsigned long long* lanelet_neigh = nullptr;  // [n_lanelets] neighbour bit masks
  int n_lanelets = 0, lanelet_pts = 0;
  int DL = 0;                     // floats per agent slot of the kernels' LDS staging of the observation rows (= D unless SIGMAENV_OBS_FULL, see ObsLayout)
  void* cbf_defer = nullptr;      // [B] u8: envs the CBF-QP's lean launch left to the full-layout launch (up to 32 vehicles; allocated by the first sigmaenv_cbf_qp)
  void* cbf_cand_big = nullptr;   // [B][N (N - 1) C^2] u32: the CBF-QP's candidate pair rows for 33 .. 64 vehicles (allocated by the first such sigmaenv_cbf_qp)
  int32_t* cbf_groups = nullptr;  // [B,N] group index of every vehicle (grouped CBF-QPs), formed by the first sigmaenv_cbf_qp call
  bool cbf_groups_valid = false;
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

// Device memory of the handle, zero-filled (state buffers rely on that).  scratch: a buffer every consumer of which is preceded by its producer in the same call
// (intermediates between kernels, staging areas) -- zero in the product build too, 0xFF bytes in the poison build (sigmaenv_device.h: sigma_poison_lds).
static int dev_alloc(sigmaenv* h, void** p, size_t bytes, bool scratch = false) {
  if (bytes == 0) bytes = 16;
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) { h->err = std::string("hipMalloc: ") + hipGetErrorString(e); return SIGMAENV_ENOMEM; }
  h->allocs.push_back(*p);
#ifdef SIGMAENV_POISON
  e = hipMemsetAsync(*p, scratch ? 0xFF : 0, bytes, h->stream);
#else
  (void)scratch;
  e = hipMemsetAsync(*p, 0, bytes, h->stream);
#endif
  if (e != hipSuccess) { h->err = std::string("hipMemsetAsync: ") + hipGetErrorString(e); return SIGMAENV_EHIP; }
  return SIGMAENV_OK;
}

static void dev_free(sigmaenv* h, void* p) {
  if (!p) return;
  for (size_t k = 0; k < h->allocs.size(); ++k)
    if (h->allocs[k] == p) { h->allocs.erase(h->allocs.begin() + (long)k); break; }
  (void)hipFree(p);
}

extern "C" int sigmaenv_n_short_term(void) { return NS; }
#ifndef SIGMAENV_BUILD_ID
#define SIGMAENV_BUILD_ID "unknown"
#endif
extern "C" const char* sigmaenv_build_id(void) { return SIGMAENV_BUILD_ID; }
extern "C" int sigmaenv_obs_dim(int32_t n_nearing) { return 1 + 2 * NS + 3 + n_nearing * 11; }
extern "C" int sigmaenv_obs_dim_ex(int32_t n_nearing, int32_t f) {  // observation_provider_rt.py:803-925
  const int s = (f & SIGMAENV_OBS_STEERING) ? 1 : 0, r = (f & SIGMAENV_OBS_REF_OTHERS) ? 1 : 0;
  const int own = 1 + s + 2 * NS + ((f & SIGMAENV_OBS_NO_DIST_CENTER) ? 0 : 1) + ((f & SIGMAENV_OBS_BOUNDARY_POINTS) ? 20 : 2) + ((f & SIGMAENV_OBS_BIRD_VIEW) ? 4 : 0);
  const int other = ((f & SIGMAENV_OBS_NO_VERTICES) ? 5 : 8) + 2 + s + ((f & SIGMAENV_OBS_NO_DIST_AGENTS) ? 0 : 1) + r * 2 * NS;
  return own + n_nearing * other + ((f & SIGMAENV_OBS_OPPONENT_PAD) ? 2 * n_nearing : 0);
}

extern "C" int sigmaenv_obs_dim_full(int32_t n_agents, int32_t n_nearing, int32_t f) {  // observation_provider_rt.py:622-851 (SIGMAENV_OBS_FULL: the `else` at :756)
  if (!(f & SIGMAENV_OBS_FULL)) return sigmaenv_obs_dim_ex(n_nearing, f);
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
    if ((N * wid[q]) % K != 0) return SIGMAENV_EINVAL;  // torch.reshape(B, n_nearing_agents, -1) raises
    others += N * wid[q];
  }
  return sigmaenv_obs_dim_ex(0, f & ~SIGMAENV_OBS_OPPONENT_PAD) + others + ((f & SIGMAENV_OBS_OPPONENT_PAD) ? 2 * K : 0);
}

extern "C" const char* sigmaenv_last_error(const sigmaenv_t* h) { return h ? h->err.c_str() : "null handle"; }

extern "C" void sigmaenv_destroy(sigmaenv_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  for (void* p : h->allocs) (void)hipFree(p);
  for (auto& tm : h->timers)
    for (auto& ev : tm.pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  delete h;
}

// Environments per workgroup: a wavefront's worth of agents (G * N <= 64) so that the per-agent phases fill all 64 lanes,
// but never so many that the grid drops below ~4 workgroups per CU (small batches keep G small instead).
static int pick_envs_per_group(int N, int B, int n_cu) {
  int G = 64 / N;
  if (G < 1) G = 1;
  if (G > MAX_G) G = MAX_G;
  while (G > 1 && (B + G - 1) / G < 4 * n_cu) G >>= 1;
  return G;
}

#ifdef SIGMAENV_POISON
// poison build: how many bytes one unit of LDS_ALLOC.LDS_SIZE stands for, measured once per process with a launch of known size (the fill of sigma_poison_lds must
// cover the allocation and nothing beyond it)
__global__ void sigma_poison_probe_kernel(unsigned* out) {
  extern __shared__ char probe_smem[];
  if (threadIdx.x == 0) { probe_smem[0] = 1; out[0] = sigma_lds_alloc_units() + (probe_smem[0] ? 0u : 1u); }
}
static int poison_probe(hipStream_t stream) {
  static bool done = false;
  if (done) return SIGMAENV_OK;
  unsigned* d = nullptr;
  unsigned units[2] = {0u, 0u};
  if (hipMalloc((void**)&d, 8) != hipSuccess) return SIGMAENV_ENOMEM;
  hipLaunchKernelGGL(sigma_poison_probe_kernel, dim3(1), dim3(64), 32768, stream, d);
  hipLaunchKernelGGL(sigma_poison_probe_kernel, dim3(1), dim3(64), 16384, stream, d + 1);
  if (hipMemcpyAsync(units, d, 8, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { (void)hipFree(d); return SIGMAENV_EHIP; }
  (void)hipFree(d);
  // (gfx950 answers 130 / 65: units of 256 bytes, allocations rounded up to 1280 bytes -- 160 KB / 128; the rounding belongs to the workgroup too)
  unsigned granule = 0u;
  for (unsigned g : {64u, 128u, 256u, 512u})
    if (units[0] * g >= 32768u && units[0] * g < 32768u + 2048u && units[1] * g >= 16384u && units[1] * g < 16384u + 2048u) granule = g;
  if (granule == 0u) {
    fprintf(stderr, "sigmaenv (poison build): unexpected LDS_ALLOC.LDS_SIZE readings %u / %u for 32768 / 16384 bytes\n", units[0], units[1]);
    return SIGMAENV_EHIP;
  }
  if (hipMemcpyToSymbol(HIP_SYMBOL(sigmadev::sigma_lds_granule), &granule, sizeof(unsigned)) != hipSuccess) return SIGMAENV_EHIP;
  fprintf(stderr, "sigmaenv (poison build): LDS and scratch buffers are poisoned (LDS_SIZE unit = %u bytes)\n", granule);
  done = true;
  return SIGMAENV_OK;
}
#endif

extern "C" int sigmaenv_create(const sigmaenv_config_t* cfg, const sigmaenv_map_t* map, int device_id, void* hip_stream, sigmaenv_t** out) {
  if (!cfg || !map || !out) return SIGMAENV_EINVAL;
  *out = nullptr;
  if (cfg->abi_version != SIGMAENV_ABI_VERSION) return SIGMAENV_EINVAL;
  for (int k = 0; k < 3; ++k) if (cfg->reserved[k] != 0) return SIGMAENV_EINVAL;
  if (cfg->n_points_short_term != 0 && cfg->n_points_short_term != NS) {
    fprintf(stderr, "sigmaenv_create: n_points_short_term = %d, but this library is built for %d (make -C sigmarl_amd/csrc NS=%d)\n", cfg->n_points_short_term, NS, cfg->n_points_short_term);
    return SIGMAENV_EINVAL;
  }
  if (cfg->env_index_base < 0 || !(cfg->obs_noise_level >= 0.0f)) return SIGMAENV_EINVAL;
  if (cfg->n_envs < 1 || cfg->n_agents < 1 || cfg->n_agents > SIGMAENV_MAX_AGENTS) return SIGMAENV_EINVAL;
  if (cfg->n_nearing < 0 || cfg->n_nearing > SIGMAENV_MAX_NEARING || cfg->n_nearing > cfg->n_agents - 1) return SIGMAENV_EINVAL;
  if (cfg->distance_type != SIGMAENV_DIST_C2C && cfg->distance_type != SIGMAENV_DIST_MTV) return SIGMAENV_EINVAL;
  if (map->n_paths < 1 || map->stride_points < 2) return SIGMAENV_EINVAL;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1 || device_id < 0 || device_id >= n_dev) return SIGMAENV_ENODEV;
  if (hipSetDevice(device_id) != hipSuccess) return SIGMAENV_ENODEV;
#ifdef SIGMAENV_POISON
  { const int rcp = poison_probe(reinterpret_cast<hipStream_t>(hip_stream)); if (rcp != SIGMAENV_OK) return rcp; }
#endif
  sigmaenv* h = new sigmaenv();
  h->cfg = *cfg;
  h->device = device_id;
  h->stream = reinterpret_cast<hipStream_t>(hip_stream);
  const int B = h->B = cfg->n_envs, N = h->N = cfg->n_agents, K = h->K = cfg->n_nearing;
  h->D = sigmaenv_obs_dim_full(N, K, cfg->obs_flags);  // the configured observation row: assembled by every kernel that refreshes observations (observe_tile)
  if (h->D < 0) { delete h; return SIGMAENV_EINVAL; }
  h->DL = h->D;
  if (cfg->obs_flags != 0) {
    const ObsLayout L(cfg->obs_flags, N, K);
    if (L.D != h->D) { delete h; return SIGMAENV_EINVAL; }  // (the two statements of the layout agree by construction)
    h->DL = L.DL;
  }
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
  // centre / left / right tables in ONE allocation each (polyline pl of path p at d_c + pl * poly_stride + p * P * 2, its point
  // count at d_nc[pl * np + p]): a per-lane polyline choice is then an address offset, not a select between kernel arguments
  ALLOC(d_c, hc.size() * 4 * 3); d_l = d_c + hc.size(); d_r = d_l + hc.size();
  ALLOC(d_y, (size_t)np * S * 4);
  ALLOC(d_nc, (size_t)np * 4 * 3); d_nl = d_nc + np; d_nr = d_nl + np;
  ALLOC(d_loop, (size_t)np);
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
  bool prune = nch <= 64;  // the candidate masks are 64-bit; longer polylines fall back to the full scan
  if (const char* e = getenv("SIGMAENV_PRUNE")) prune = prune && atoi(e) != 0;
  float4 *d_box = nullptr, *d_gbox = nullptr;
  std::vector<float4> hb, hg;  // must outlive the asynchronous uploads below
  std::vector<ulonglong4> hn;
  ulonglong4* d_neigh = nullptr;
  const float neigh_radius_far = 9.0f * (sqrtf((float)((double)cfg->length / 2.0) * (float)((double)cfg->length / 2.0) +
                                               (float)((double)cfg->width / 2.0) * (float)((double)cfg->width / 2.0)) * 1.00001f + 1e-5f);
  const float neigh_radius_tight = 3.6f * (sqrtf((float)((double)cfg->length / 2.0) * (float)((double)cfg->length / 2.0) +
                                                 (float)((double)cfg->width / 2.0) * (float)((double)cfg->width / 2.0)) * 1.00001f + 1e-5f);
  const float neigh_radius = 6.0f * (sqrtf((float)((double)cfg->length / 2.0) * (float)((double)cfg->length / 2.0) +
                                           (float)((double)cfg->width / 2.0) * (float)((double)cfg->width / 2.0)) * 1.00001f + 1e-5f);
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
    hg.assign((size_t)np * 3 * 8, make_float4(1e30f, 1e30f, -1e30f, -1e30f));
    for (size_t pq = 0; pq < (size_t)np * 3; ++pq) {
      for (int cidx = 0; cidx < nch; ++cidx) {
        const float4& bx = hb[pq * nch + cidx];
        if (bx.x > bx.z) continue;  // chunk without real segments
        float4& gx = hg[pq * 8 + cidx / 8];
        gx.x = bx.x < gx.x ? bx.x : gx.x; gx.y = bx.y < gx.y ? bx.y : gx.y; gx.z = bx.z > gx.z ? bx.z : gx.z; gx.w = bx.w > gx.w ? bx.w : gx.w;
      }
    }
    ALLOC(d_gbox, hg.size() * sizeof(float4));
    H2D(d_gbox, hg.data(), hg.size() * sizeof(float4));
    // neighbour masks: chunks whose box is within neigh_radius of a chunk's box (double arithmetic, inclusive with slack)
    hn.assign((size_t)np * 3 * nch, make_ulonglong4(0ull, 0ull, 0ull, 0ull));
    for (size_t pq = 0; pq < (size_t)np * 3; ++pq) {
      for (int a = 0; a < nch; ++a) {
        const float4& A = hb[pq * nch + a];
        if (A.x > A.z) continue;
        unsigned long long bits = 0ull, bits_far = 0ull, bits_tight = 0ull;
        for (int b2 = 0; b2 < nch; ++b2) {
          const float4& Bx = hb[pq * nch + b2];
          if (Bx.x > Bx.z) continue;
          double dx = std::max(std::max((double)A.x - Bx.z, (double)Bx.x - A.z), 0.0);
          double dy = std::max(std::max((double)A.y - Bx.w, (double)Bx.y - A.w), 0.0);
          const double dd = std::sqrt(dx * dx + dy * dy);
          if (dd <= (double)neigh_radius_tight + 1e-5) bits_tight |= 1ull << b2;
          if (dd <= (double)neigh_radius + 1e-5) bits |= 1ull << b2;
          if (dd <= (double)neigh_radius_far + 1e-5) bits_far |= 1ull << b2;
        }
        hn[pq * nch + a] = make_ulonglong4(bits_tight, bits, bits_far, 0ull);
      }
    }
    ALLOC(d_neigh, hn.size() * sizeof(ulonglong4));
    H2D(d_neigh, hn.data(), hn.size() * sizeof(ulonglong4));
  }
  const float lh = (float)((double)cfg->length / 2.0), wh = (float)((double)cfg->width / 2.0);
  const float rect_radius = sqrtf(lh * lh + wh * wh) * 1.00001f + 1e-5f;
  // the shared-reciprocal division (div_shared) is exact when no segment needs the scaling of the general IEEE sequence
  int fast_div = 1;
  for (int p = 0; p < np && fast_div; ++p) {
    const float* polys[3] = {hc.data() + (size_t)p * P * 2, hl.data() + (size_t)p * P * 2, hr.data() + (size_t)p * P * 2};
    const int cnt[3] = {map->n_center[p], map->n_left[p], map->n_right[p]};
    for (int q = 0; q < 3 && fast_div; ++q) {
      for (int k = 0; k + 1 < cnt[q]; ++k) {
        const float lx = polys[q][2 * (k + 1)] - polys[q][2 * k], ly = polys[q][2 * (k + 1) + 1] - polys[q][2 * k + 1];
        const float len2 = lx * lx + ly * ly;  // the kernel's operation order (no contraction)
        if (!(len2 >= 0x1p-60f && len2 <= 0x1p60f)) { fast_div = 0; break; }
      }
    }
  }
  if (const char* e = getenv("SIGMAENV_FASTDIV")) fast_div = fast_div && atoi(e) != 0;
  h->map = DevMap{d_c, d_l, d_r, d_y, d_nc, d_nl, d_nr, d_loop, P, np, S, d_box, d_gbox, d_neigh, neigh_radius, neigh_radius_far, prune ? nch : 0, nullptr, fast_div, rect_radius, (int32_t)hc.size()};
  h->map.neigh_radius_tight = neigh_radius_tight;
  {  // start table: derived state of an agent placed on any centre-line point, by the kernels' own scan code
    float* d_tab = nullptr;
    ALLOC(d_tab, (size_t)np * P * START_ROW * sizeof(float));
    const size_t tab_smem = Smem::bytes(64, 1, 0, 1) + 16;
    if (tab_smem > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_start_table_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tab_smem);
    hipLaunchKernelGGL(sigmaenv_start_table_kernel, dim3((np * P + 63) / 64), dim3(256), tab_smem, h->stream, h->cfg, h->map, d_tab);
    if (hipGetLastError() != hipSuccess) { sigmaenv_destroy(h); return SIGMAENV_EHIP; }
    h->map.start_table = d_tab;
  }
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
      {SIGMAENV_BUF_CBF_NOMINAL, (void**)&g.cbf_nominal, BN * 8},
  };
  for (auto& sp : specs) {
    ALLOC(*sp.p, sp.bytes);
    h->bufs[sp.id] = *sp.p;
    h->buf_bytes[sp.id] = sp.bytes;
  }
  ALLOC(g.reset_mask, (size_t)B * 8);
  ALLOC(g.reset_full, (size_t)B);
  g.slab = nullptr;
  g.fresh = nullptr;
  if (cfg->obs_flags & SIGMAENV_OBS_BOUNDARY_POINTS) {
    ALLOC(g.fresh, BN);
    if (hipMemsetAsync(g.fresh, 0, BN, h->stream) != hipSuccess) { sigmaenv_destroy(h); return SIGMAENV_EHIP; }
  }
  {
    auto magic = [](unsigned d) { return d <= 1u ? 0u : (uint32_t)(((1ull << 32) + d - 1ull) / d); };
    g.mN = magic((unsigned)N);
    g.mT1 = magic((unsigned)(NS + 4 * K));
    g.mT2 = magic((unsigned)(K + 1));
    g.mTP = magic((unsigned)(N * (N - 1) / 2));
    g.mW = magic((unsigned)(N * (h->D + 1) + 1));
    g.mVT1 = cfg->obs_flags != 0 ? magic((unsigned)ObsLayout(cfg->obs_flags, N, K).T1) : 0u;
  }
  g.lanelet_centers = nullptr; g.lanelet_neigh = nullptr; g.n_lanelets = 0; g.lanelet_pts = 0;
  g.bnd_left = h->map.left; g.bnd_n_center = h->map.n_center; g.bnd_is_loop = h->map.is_loop; g.bnd_poly_stride = h->map.poly_stride; g.bnd_P = h->map.P;
  g.dbg_ts = nullptr;
  g.dbg_ts2 = nullptr;
  g.dbg_skip = 0;
  g.obs_salt = 0u;
#ifdef SIGMAENV_PROFILE
  if (const char* e = getenv("SIGMAENV_DEBUG_SKIP")) g.dbg_skip = atoi(e);
#endif
#ifdef SIGMAENV_PROFILE
  if (const char* e = getenv("SIGMAENV_TIMESTAMPS")) {
    if (atoi(e) == 1) { ALLOC(g.dbg_ts, (size_t)B * 16 * sizeof(unsigned long long)); (void)hipMemsetAsync(g.dbg_ts, 0, (size_t)B * 128, h->stream); }
    if (atoi(e) == 2) { ALLOC(g.dbg_ts2, (size_t)B * 16 * sizeof(unsigned long long)); (void)hipMemsetAsync(g.dbg_ts2, 0, (size_t)B * 128, h->stream); }
  }
#endif
#undef ALLOC
#undef H2D
  hipDeviceProp_t prop;
  int n_cu = 256;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) n_cu = prop.multiProcessorCount;
  h->G = pick_envs_per_group(N, B, n_cu);
  if (cfg->envs_per_group >= 1 && cfg->envs_per_group * N <= 64) h->G = cfg->envs_per_group;
  if (const char* e = getenv("SIGMAENV_G")) {
    int v = atoi(e);
    if (v >= 1 && v * N <= 64) h->G = v;
  }
  h->block = 256;
  if (const char* e = getenv("SIGMAENV_BLOCK")) {
    int v = atoi(e);
    if (v >= 64 && v <= 256 && v % 64 == 0) h->block = v;
  }
  if (const char* e = getenv("SIGMAENV_TIMING_STRIDE")) { int v = atoi(e); if (v >= 1) h->timing_stride = v; }
  h->grid = (B + h->G - 1) / h->G;
  h->reset_block = 256;  // measured at 16 x 4096: 512-thread reset workgroups are slower (more early-exit launch cost, lower occupancy)
  if (const char* e = getenv("SIGMAENV_RESET_BLOCK")) {
    int v = atoi(e);
    if (v >= 64 && v <= 512 && v % 64 == 0) h->reset_block = v;
  }
  h->smem_bytes = ((Smem::bytes(h->G * N, N, K, h->DL) + 15) & ~(size_t)15) + MAX_G * 8 + MAX_G * 4 + 16 + (MAX_G + 1) * 4;  // masks, flags, counters, env list
  if (h->smem_bytes > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_observe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem_bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_reset_derive_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem_bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_auto_reset_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->smem_bytes);
  }
  {  // the step kernel's tiling: one wavefront per tile of wave_G environments; as many tiles as give every SIMD four wavefronts
    // about 16 agent slots per wavefront (measured at 16 agents: one env per wavefront beats two and four at every batch size from 256
    // to 32768 envs -- a bigger tile is a longer serial chain per wavefront and more LDS per wavefront, i.e. fewer of them resident)
    int wg = 16 / N;
    if (wg < 1) wg = 1;
    while (wg > 1 && (B + wg - 1) / wg < 16 * n_cu) wg >>= 1;
    if (cfg->envs_per_group >= 1 && cfg->envs_per_group * N <= 64) wg = cfg->envs_per_group;
    if (const char* e = getenv("SIGMAENV_WAVE_G")) { int v = atoi(e); if (v >= 1 && v * N <= 64) wg = v; }
    h->wave_G = wg;
    h->cfg_derived = DevConfig{h->cfg, derive_config(h->cfg)};  // (a member: the asynchronous copy reads it after this function returns)
    if (dev_alloc(h, (void**)&h->d_cfg, sizeof(DevConfig)) != SIGMAENV_OK || hipMemcpyAsync(h->d_cfg, &h->cfg_derived, sizeof(DevConfig), hipMemcpyHostToDevice, h->stream) != hipSuccess) {
      sigmaenv_destroy(h);
      return SIGMAENV_EHIP;
    }
    { const unsigned d = (unsigned)(wg * N); h->buf.mSG = d <= 1u ? 0u : (uint32_t)(((1ull << 32) + d - 1ull) / d); }
    h->wave_spec = ((cfg->obs_flags == 0 || (N == 16 && wg == 1)) && K == 2 && ((N == 16 && wg == 1) || (N == 32 && wg == 1) || (N == 8 && wg == 2) || (N == 4 && wg == 4))) ? N * 256 + wg : 0;
    if (const char* e = getenv("SIGMAENV_WAVE_SPEC")) { if (atoi(e) == 0) h->wave_spec = 0; }  // A/B: the generic instantiation
    h->wave_wpb = 1;
    if (const char* e = getenv("SIGMAENV_WPB")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4) h->wave_wpb = v; }
    h->wave_tile_lds = (((Smem::bytes(wg * N, N, K, h->DL, true) + 15) & ~(size_t)15) + 16 * (size_t)wg + 16 + 16 * (size_t)wg + 15) & ~(size_t)15;  // tile | masks, flags, env list | timers
    const int tiles = (B + wg - 1) / wg;
    h->wave_grid = (tiles + h->wave_wpb - 1) / h->wave_wpb;
    if (h->wave_tile_lds * h->wave_wpb > 64 * 1024) {
      const int lds = (int)(h->wave_tile_lds * h->wave_wpb);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_step_wave_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_step_wave_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_step_wave_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_step_wave_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_step_wave_kernel<true, true, 16, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_step_wave_kernel<true, true, 0, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_step_wave_kernel<true, false, 0, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_step_wave_kernel<false, true, 0, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_step_wave_kernel<false, false, 0, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_step_wave_kernel<true, true, 16, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_step_wave_kernel<true, true, 16, 1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_step_wave_kernel<true, true, 16, 1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_step_wave_kernel<true, true, 32, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_step_wave_kernel<true, true, 8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sigmaenv_step_wave_kernel<true, true, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
  }
  if (hipStreamSynchronize(h->stream) != hipSuccess) { sigmaenv_destroy(h); return SIGMAENV_EHIP; }
  *out = h;
  return SIGMAENV_OK;
}

#include "sigmaenv_opponent_fill.inc"

extern "C" int sigmaenv_set_lanelets(sigmaenv_t* h, int32_t n_lanelets, int32_t max_points, const float* centers, const uint64_t* neighbors) {
  if (!h) return SIGMAENV_EINVAL;
  if (n_lanelets < 1 || n_lanelets > 64 || max_points < 1 || !centers || !neighbors) { h->err = "set_lanelets: 1..64 lanelets with their centre lines and neighbour masks"; return SIGMAENV_EINVAL; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->lanelet_centers) {
    dev_free(h, h->lanelet_centers); dev_free(h, h->lanelet_neigh);
    h->lanelet_centers = nullptr; h->lanelet_neigh = nullptr; h->n_lanelets = 0;
    h->buf.lanelet_centers = nullptr; h->buf.lanelet_neigh = nullptr; h->buf.n_lanelets = 0;
  }
  int rc;
  if ((rc = dev_alloc(h, (void**)&h->lanelet_centers, (size_t)n_lanelets * max_points * 2 * sizeof(float))) != SIGMAENV_OK) return rc;
  if ((rc = dev_alloc(h, (void**)&h->lanelet_neigh, (size_t)n_lanelets * sizeof(uint64_t))) != SIGMAENV_OK) return rc;
  HIPCHK(h, hipMemcpyAsync(h->lanelet_centers, centers, (size_t)n_lanelets * max_points * 2 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->lanelet_neigh, neighbors, (size_t)n_lanelets * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->n_lanelets = n_lanelets;
  h->lanelet_pts = max_points;
  h->buf.lanelet_centers = h->lanelet_centers; h->buf.lanelet_neigh = h->lanelet_neigh; h->buf.n_lanelets = n_lanelets; h->buf.lanelet_pts = max_points;
  return SIGMAENV_OK;
}

extern "C" int sigmaenv_opponent_fill(sigmaenv_t* h, const float* actions) {
  if (!h || !actions) return SIGMAENV_EINVAL;
  if (!(h->cfg.obs_flags & SIGMAENV_OBS_OPPONENT_PAD)) { h->err = "opponent_fill: the configuration has no placeholder columns (SIGMAENV_OBS_OPPONENT_PAD)"; return SIGMAENV_EINVAL; }
  HIPCHK(h, hipSetDevice(h->device));
  const size_t n = (size_t)h->B * h->N * h->cfg.n_nearing;
  if (n == 0) return SIGMAENV_OK;
  hipLaunchKernelGGL(obsvar::sigmaenv_opponent_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->buf.obs, h->D, h->buf.nearing, actions, h->B, h->N,
                     h->cfg.n_nearing);
  HIPCHK(h, hipGetLastError());
  return SIGMAENV_OK;
}

// the path range of a device-side reset: [path_first, path_first + path_count) of the table, or the handle's sub-scenario lists
static bool paths_ok(const sigmaenv* h, int32_t path_first, int32_t path_count) {
  if (path_count == SIGMAENV_SCENARIO_LISTS) return h->map.n_lists > 0;
  return path_first >= 0 && path_count >= 1 && path_first + path_count <= h->n_paths;
}

extern "C" int sigmaenv_set_scenario_lists(sigmaenv_t* h, int32_t n_lists, const int32_t* first, const int32_t* count, const float* probabilities) {
  if (!h) return SIGMAENV_EINVAL;
  if (n_lists < 1 || n_lists > 4 || !first || !count || !probabilities) { h->err = "set_scenario_lists: 1..4 lists with their first path, path count and probability"; return SIGMAENV_EINVAL; }
  double tot = 0.0;
  for (int k = 0; k < n_lists; ++k) {
    if (first[k] < 0 || count[k] < 1 || first[k] + count[k] > h->n_paths || first[k] + count[k] > 65535 || !(probabilities[k] >= 0.0f)) { h->err = "set_scenario_lists: list outside the path table, or a negative probability"; return SIGMAENV_EINVAL; }
    tot += (double)probabilities[k];
  }
  if (!(tot > 0.0)) { h->err = "set_scenario_lists: the probabilities sum to zero"; return SIGMAENV_EINVAL; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  int32_t lf[4], lc[4];
  float cdf[4];
  double acc = 0.0;
  for (int k = 0; k < 4; ++k) {
    lf[k] = k < n_lists ? first[k] : 0;
    lc[k] = k < n_lists ? count[k] : 1;
    if (k < n_lists) acc += (double)probabilities[k] / tot;
    cdf[k] = k + 1 >= n_lists ? 1.0f : (float)acc;  // torch.multinomial normalises the weights as well
  }
  DevMap& m = h->map;
  m.list_first16 = 0ull; m.list_count16 = 0ull;
  for (int k = 0; k < 4; ++k) { m.list_first16 |= (unsigned long long)lf[k] << (16 * k); m.list_count16 |= (unsigned long long)lc[k] << (16 * k); }
  m.cdf0 = cdf[0]; m.cdf1 = cdf[1]; m.cdf2 = cdf[2];
  m.n_lists = n_lists;
  return SIGMAENV_OK;
}

static int launch_derive(sigmaenv* h, int with_obs) {
  // resets touch few envs: one env per workgroup (G = 1) keeps the untouched ones out of the way
  hipLaunchKernelGGL(sigmaenv_reset_derive_kernel, dim3(h->B), dim3(h->reset_block), h->smem_bytes, h->stream, h->cfg, h->map, h->buf, with_obs, 1);
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
    if ((rc = dev_alloc(h, &p, cap * 4, true)) != 0) return rc;
    h->d_env_idx = (int32_t*)p;
    if ((rc = dev_alloc(h, &p, cap * 4, true)) != 0) return rc;
    h->d_agent_idx = (int32_t*)p;
    if ((rc = dev_alloc(h, &p, cap * 16, true)) != 0) return rc;
    h->d_path_ids = (int32_t*)p;
    if ((rc = dev_alloc(h, &p, cap * 32, true)) != 0) return rc;
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

// HIP-event bracketing of a SAMPLE of the launches of kernel `id` (every timing_stride-th): every event pair costs a few microseconds of queue
// time, bracketing all launches would slow down the very region it measures.  timer_begin returns the slot to close with timer_end, or -1.
static int timer_begin(sigmaenv* h, int id) {
  if (!h->timing) return -1;
  sigmaenv::KernelTimer& tm = h->timers[id];
  if ((tm.launch_count++ % (unsigned long long)h->timing_stride) != 0) return -1;
  const int slot = (int)tm.used.size();
  if (slot >= (int)tm.pool.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
    tm.pool.emplace_back(a, b);
  }
  if (hipEventRecord(tm.pool[slot].first, h->stream) != hipSuccess) return -1;
  tm.used.push_back(slot);
  return slot;
}
static void timer_end(sigmaenv* h, int id, int slot) {
  if (slot >= 0) (void)hipEventRecord(h->timers[id].pool[slot].second, h->stream);
}

// n_steps launches' worth of fused steps in ONE launch when n_steps > 1 (sigmaenv_step_autoreset_n): step t reads actions + t * act_stride, draws its
// resets from counter + t and records into slab + t * slab_stride (floats)
static int launch_step(sigmaenv* h, const float* actions, uint64_t seed, uint64_t counter, int path_first, int path_count, int n_steps = 1, size_t act_stride = 0,
                       float* slab = nullptr, size_t slab_stride = 0, bool slab_from_handle = true) {
  if (!h || !actions) return SIGMAENV_EINVAL;
  if (slab_from_handle) slab = h->buf.slab;
  HIPCHK(h, hipSetDevice(h->device));  // handles on several GPUs may live in one process: every entry point that enqueues work selects its device
  const int slot = timer_begin(h, SIGMAENV_KERNEL_STEP);
  {
    // instantiations: exact shared-reciprocal division or plain `/` in the scan (DevMap::fast_div), lane pair per agent in the dynamics or not
    const bool par = 2 * h->wave_G * h->N <= 64;
    auto kern = h->map.fast_div ? (par ? sigmaenv_step_wave_kernel<true, true> : sigmaenv_step_wave_kernel<true, false>)
                                : (par ? sigmaenv_step_wave_kernel<false, true> : sigmaenv_step_wave_kernel<false, false>);
    if (h->cfg.obs_flags != 0 && h->map.fast_div && h->wave_spec == 16 * 256 + 1) kern = sigmaenv_step_wave_kernel<true, true, 16, 1, true>;
    else if (h->cfg.obs_flags != 0)  // the instantiations that carry the non-default observation rows
      kern = h->map.fast_div ? (par ? sigmaenv_step_wave_kernel<true, true, 0, 0, true> : sigmaenv_step_wave_kernel<true, false, 0, 0, true>)
                             : (par ? sigmaenv_step_wave_kernel<false, true, 0, 0, true> : sigmaenv_step_wave_kernel<false, false, 0, 0, true>);
    if (h->map.fast_div && h->cfg.obs_flags == 0) {  // fixed-shape instantiations (the plain-division variant of a map with degenerate segments stays generic)
      switch (h->wave_spec) {
        case 16 * 256 + 1: kern = sigmaenv_step_wave_kernel<true, true, 16, 1>; break;
        case 32 * 256 + 1: kern = sigmaenv_step_wave_kernel<true, true, 32, 1>; break;
        case 8 * 256 + 2: kern = sigmaenv_step_wave_kernel<true, true, 8, 2>; break;
        case 4 * 256 + 4: kern = sigmaenv_step_wave_kernel<true, true, 4, 4>; break;
        default: break;
      }
    }
    if (h->cfg.distance_type != SIGMAENV_DIST_C2C && h->map.fast_div && h->wave_spec == 16 * 256 + 1)  // mtv at the metric's shape: per-rectangle records staged per agent
      kern = h->cfg.obs_flags != 0 ? sigmaenv_step_wave_kernel<true, true, 16, 1, true, true> : sigmaenv_step_wave_kernel<true, true, 16, 1, false, true>;
    const StepKernArgs ka{h->map, h->buf, actions, slab, act_stride, slab_stride, seed, counter, h->wave_G, (int)h->wave_tile_lds, path_first, path_count, n_steps};
    hipLaunchKernelGGL(kern, dim3(h->wave_grid), dim3(64 * h->wave_wpb), h->wave_tile_lds * h->wave_wpb, h->stream, (const sigmaenv_config_t*)h->d_cfg, ka);
  }
  HIPCHK(h, hipGetLastError());
  timer_end(h, SIGMAENV_KERNEL_STEP, slot);
  return SIGMAENV_OK;
}

extern "C" int sigmaenv_step(sigmaenv_t* h, const float* actions) { return launch_step(h, actions, 0, 0, 0, 0); }

extern "C" int sigmaenv_step_autoreset(sigmaenv_t* h, const float* actions, uint64_t seed, uint64_t counter, int32_t path_first, int32_t path_count) {
  if (!h || !paths_ok(h, path_first, path_count)) return SIGMAENV_EINVAL;
  return launch_step(h, actions, seed, counter, path_first, path_count);
}

extern "C" int sigmaenv_cbf_rewards(sigmaenv_t* h, const float* actions, double* margins);  // sigmaenv_cbf.inc
extern "C" int sigmaenv_cbf_qp(sigmaenv_t* h, const float* actions, float* actions_safe, double* u_opt, int32_t* info);

// n_steps fused steps (step + rollout record + device-side resets) of every env in ONE launch: the reference's rollout loop over a chunk of steps
// (helper_training.py:687-788: policy -> env.step -> step_mdp, T times) for actions that are already on the device.  Same end state, same record
// rows and same reset draws as n_steps calls of sigmaenv_step_autoreset(h, actions + t * action_stride, seed, counter0 + t, ...) with the slab set to
// slab + t * slab_stride -- bit for bit (tests/test_gpu_nstep.py); envs are independent, so every wavefront walks its own env through the n_steps steps
// without waiting for any other.  actions: device f32, step t at actions + t * action_stride floats ([B, N, 2] each; action_stride 0 repeats one
// action block); slab: device f32 or NULL, row block t ([B, N (D + 1) + 1]) at slab + t * slab_stride floats.
extern "C" int sigmaenv_step_autoreset_n(sigmaenv_t* h, const float* actions, int32_t n_steps, int64_t action_stride, float* slab, int64_t slab_stride,
                                         uint64_t seed, uint64_t counter0, int32_t path_first, int32_t path_count) {
  if (!h) return SIGMAENV_EINVAL;
  if (!actions || n_steps < 1 || action_stride < 0 || slab_stride < 0 || !paths_ok(h, path_first, path_count)) {
    h->err = "step_autoreset_n: bad argument";
    return SIGMAENV_EINVAL;
  }
  if (n_steps > 1 && (h->cfg.rew_flags & (SIGMAENV_REW_CBF | SIGMAENV_REW_CBF_QP))) {
    // rew_method with "cbf": the reward channels / the safe action of step t come from the CBF launch on step t's actions (CBFQP.update_qp between policy and
    // env.step, helper_training.py:1616-1627).  The chunk is then n_steps x (CBF launch, fused step + record + resets) enqueued back to back -- the same calls,
    // hence the same bits, as sigmaenv_cbf_rewards / sigmaenv_cbf_qp + sigmaenv_step_autoreset per step (tests/test_gpu_nstep.py), without returning to the caller.
    if (!h->cbf_seg4) {
      h->err = "step_autoreset_n: rew_method with \"cbf\" needs sigmaenv_cbf_attach before a chunk of steps";
      return SIGMAENV_EINVAL;
    }
    for (int t = 0; t < n_steps; ++t) {
      const float* act = actions + (size_t)t * (size_t)action_stride;
      const float* step_act = act;
      int rc;
      if (h->cfg.rew_flags & SIGMAENV_REW_CBF) {
        rc = sigmaenv_cbf_rewards(h, act, nullptr);
      } else {
        rc = sigmaenv_cbf_qp(h, act, (float*)h->cbf_safe, nullptr, nullptr);
        if (h->cbf_cfg.is_apply_cbf_action || h->cbf_cfg.is_grouping) step_act = (const float*)h->cbf_safe;  // (as sigmaenv_rollout: cbf_qp.py:2211-2222)
      }
      if (rc) return rc;
      rc = launch_step(h, step_act, seed, counter0 + (uint64_t)t, path_first, path_count, 1, 0, slab ? slab + (size_t)t * (size_t)slab_stride : nullptr, 0, false);
      if (rc) return rc;
    }
    return SIGMAENV_OK;
  }
  return launch_step(h, actions, seed, counter0, path_first, path_count, n_steps, (size_t)action_stride, slab, (size_t)slab_stride, false);
}

// One call for several handles (env shards of one GPU on their own streams): slab_ptrs[k] (may be NULL) becomes handle k's record
// target, then its fused step + reset is enqueued.  Saves the per-call host overhead of a binding that steps shards one by one.
extern "C" int sigmaenv_step_autoreset_many(sigmaenv_t** hs, int32_t n, const float* const* actions, float* const* slab_ptrs, const uint64_t* seeds,
                                            uint64_t counter, int32_t path_first, int32_t path_count) {
  if (!hs || !actions || !seeds || n < 1) return SIGMAENV_EINVAL;
  for (int k = 0; k < n; ++k) {
    sigmaenv* h = hs[k];
    if (!h || !paths_ok(h, path_first, path_count)) return SIGMAENV_EINVAL;
    if (slab_ptrs) h->buf.slab = slab_ptrs[k];
    const int rc = launch_step(h, actions[k], seeds[k], counter, path_first, path_count);
    if (rc) return rc;
  }
  return SIGMAENV_OK;
}

extern "C" int sigmaenv_observe(sigmaenv_t* h) {
  if (!h) return SIGMAENV_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  DevBufs b = h->buf;
  b.obs_salt = ++h->observe_calls;  // an observation taken AGAIN draws its own sensor noise (obs_noise; shared specification with the oracle's observe)
  hipLaunchKernelGGL(sigmaenv_observe_kernel, dim3(h->grid), dim3(h->block), h->smem_bytes, h->stream, h->cfg, b, h->G);
  HIPCHK(h, hipGetLastError());
  return SIGMAENV_OK;
}

extern "C" int sigmaenv_auto_reset(sigmaenv_t* h, uint64_t seed, uint64_t counter, int32_t path_first, int32_t path_count) {
  if (!h || !paths_ok(h, path_first, path_count)) return SIGMAENV_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(sigmaenv_auto_reset_kernel, dim3(h->B), dim3(h->reset_block), h->smem_bytes, h->stream, h->cfg, h->map, h->buf, seed, counter,
                     (int)path_first, (int)path_count, 1);
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
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return SIGMAENV_OK;
}

// Rollout slab: when set, every subsequent step also writes one contiguous fp32 row per env, [N*D observation | N reward | done],
// to dev_ptr ([B, N*(D+1)+1]); the caller rotates the pointer through its rollout buffer.  NULL disables it.
extern "C" int sigmaenv_set_slab(sigmaenv_t* h, void* dev_ptr) {
  if (!h) return SIGMAENV_EINVAL;
  h->buf.slab = reinterpret_cast<float*>(dev_ptr);
  return SIGMAENV_OK;
}

// Stride between the record blocks of consecutive steps of sigmaenv_rollout* (0: the handle's own [T, B, W] layout); see include/sigmaenv.h
extern "C" int sigmaenv_set_rollout_slab_stride(sigmaenv_t* h, int64_t stride_floats) {
  if (!h) return SIGMAENV_EINVAL;
  const int64_t own = (int64_t)h->B * ((int64_t)h->N * (h->D + 1) + 1);
  if (stride_floats != 0 && stride_floats < own) { h->err = "set_rollout_slab_stride: stride below the handle's own record block B * (N * (D + 1) + 1)"; return SIGMAENV_EINVAL; }
  h->rollout_slab_stride = (size_t)stride_floats;
  return SIGMAENV_OK;
}

// Diagnostics: copies the per-workgroup phase timestamps of the LAST step launch to `out` ([n_groups][8] u64); returns the number of
// workgroups, 0 when SIGMAENV_TIMESTAMPS was not set at create.
extern "C" int sigmaenv_debug_timestamps(sigmaenv_t* h, unsigned long long* out, int32_t max_groups) {
  if (!h || !out) return SIGMAENV_EINVAL;
  if (!h->buf.dbg_ts && h->buf.dbg_ts2) {
    int n2 = max_groups < h->B ? max_groups : h->B;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, h->buf.dbg_ts2, (size_t)n2 * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return n2;
  }
  if (!h->buf.dbg_ts) return 0;
  int n = h->B < max_groups ? h->B : max_groups;  // one row per tile; unused rows stay zero
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out, h->buf.dbg_ts, (size_t)n * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return n;
}

// Enables HIP-event bracketing of the subsequent launches of the timed kernels (first call) and reports / clears the collected launches of kernel
// `kernel_id` (SIGMAENV_KERNEL_*).
extern "C" int sigmaenv_kernel_time_ms(sigmaenv_t* h, int32_t kernel_id, double* avg_ms, int32_t* n_launches) {
  if (!h || !avg_ms || !n_launches || kernel_id < 0 || kernel_id >= SIGMAENV_KERNEL_COUNT) return SIGMAENV_EINVAL;
  *avg_ms = 0.0;
  *n_launches = 0;
  if (!h->timing) { h->timing = true; return SIGMAENV_OK; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  sigmaenv::KernelTimer& tm = h->timers[kernel_id];
  double total = 0.0;
  for (int slot : tm.used) {
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, tm.pool[slot].first, tm.pool[slot].second));
    total += ms;
  }
  *n_launches = (int32_t)tm.used.size();
  if (*n_launches) *avg_ms = total / *n_launches;
  tm.used.clear();
  return SIGMAENV_OK;
}
extern "C" int sigmaenv_step_time_ms(sigmaenv_t* h, double* avg_ms, int32_t* n_launches) { return sigmaenv_kernel_time_ms(h, SIGMAENV_KERNEL_STEP, avg_ms, n_launches); }

__global__ void sigmaenv_trig_selftest_kernel(int kind, int n, const float* __restrict__ in, float* __restrict__ out) {
  sigma_poison_lds();
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const float x = in[k];
  out[k] = kind == 0 ? cr_sin(x) : (kind == 1 ? cr_cos(x) : (kind == 2 ? cr_tan(x) : cr_atan(x)));
}
extern "C" int sigmaenv_trig_selftest(sigmaenv_t* h, int32_t kind, int32_t n, const float* in, float* out) {
  if (!h || !in || !out || n < 0 || kind < 0 || kind > 3) return SIGMAENV_EINVAL;
  if (n == 0) return SIGMAENV_OK;
  HIPCHK(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(sigmaenv_trig_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, (int)kind, (int)n, in, out);
  HIPCHK(h, hipGetLastError());
  return SIGMAENV_OK;
}

#include "sigmaenv_actor.inc"
#include "sigmaenv_mlp32.inc"
#include "sigmaenv_cbf.inc"
