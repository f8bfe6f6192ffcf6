// sigmaenv_device.h -- gfx950 device functions of the fused environment step.
//
// Arithmetic contract (DESIGN.md "Arithmetic contract"): fp32, one IEEE operation per torch op of the reference,
// compiled with -ffp-contract=off so nothing is fused implicitly; the only fused operation is the explicit fmaf in
// norm2 (PyTorch-CPU evaluates torch.norm over a length-2 dim as sqrt(fma(y,y,x*x))).  Division and sqrt are the
// correctly rounded forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).  sin/cos/tan/atan are the correctly
// rounded fp32 value through the fp64 algorithm the oracle shares (include/sigma_trig_f32.h): identical bits on both sides.
//
// Reference citations are relative to /root/reference/sigmarl.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sigmaenv.h"
#include "../../include/sigma_trig_f32.h"

#define NS SIGMAENV_N_SHORT_TERM
#define PI32 3.14159274101257324f
#define TWO_PI32 6.28318548202514648f

namespace sigmadev {

// `make poison` (libsigmaenv_poison.so, -DSIGMAENV_POISON; never loaded by the package unless SIGMAENV_LIB points at it): every kernel starts by filling its whole LDS
// allocation -- static + dynamic, read from the wavefront's LDS_ALLOC hardware register -- with 0xFF bytes (NaN as float / double, huge as an index), and every SCRATCH
// buffer in HBM is allocated as 0xFF instead of zero (dev_alloc).  A kernel that reads what "the previous workgroup left" then fails deterministically instead of
// from run to run (the packed Hessian's last word of round 4: profiles/r06_poison_suite.txt).  The product build compiles this to nothing.
#ifdef SIGMAENV_POISON
__device__ unsigned sigma_lds_granule = 0u;  // bytes per unit of LDS_ALLOC.LDS_SIZE, measured by sigmaenv_create's probe launch (sigma_poison_probe_kernel)
__device__ __forceinline__ unsigned sigma_lds_alloc_units() { return (unsigned)__builtin_amdgcn_s_getreg(6 | (12 << 6) | ((9 - 1) << 11)); }  // HW_REG_LDS_ALLOC, LDS_SIZE = bits [20:12]
__device__ __forceinline__ void sigma_poison_lds() {
  const unsigned words = sigma_lds_alloc_units() * sigma_lds_granule / 4u;
  auto* lds = (__attribute__((address_space(3))) unsigned*)0;
  const unsigned tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z), nt = blockDim.x * blockDim.y * blockDim.z;
  for (unsigned k = tid; k < words; k += nt) lds[k] = 0xFFFFFFFFu;
  __syncthreads();
}
#else
__device__ __forceinline__ void sigma_poison_lds() {}
#endif

struct DevMap {
  const float* center;  // [n_paths][P][2] padded (world_state_rt.py:313-420)
  const float* left;
  const float* right;
  const float* yaw;     // [n_paths][yaw_stride]
  const int32_t* n_center;
  const int32_t* n_left;
  const int32_t* n_right;
  const uint8_t* is_loop;
  int32_t P, n_paths, yaw_stride;
  // pruning table: axis-aligned boxes of runs of CHUNK consecutive real segments, [n_paths][3 (centre,left,right)][nch]
  const float4* chunk_box;  // (min_x, min_y, max_x, max_y)
  const float4* group_box;  // union boxes of groups of 8 consecutive chunks, [n_paths][3][8]
  const ulonglong4* chunk_neigh;  // [n_paths][3][nch]: bit c set iff the box of chunk c is within neigh_radius_tight (.x) / neigh_radius (.y) / neigh_radius_far (.z) of this chunk's box
  float neigh_radius, neigh_radius_far;
  int32_t nch;              // boxes per polyline (stride); 0 disables pruning (brute-force scan)
  const float* start_table; // derived state of an agent freshly placed on centre-line point pt of path p, [n_paths][P][START_ROW]
  int32_t fast_div;         // every real segment has 2^-60 <= |l|^2 <= 2^60: the shared-reciprocal division is exact (div_shared)
  float rect_radius;        // upper bound of |vertex - centre| of a vehicle rectangle (half diagonal + slack)
  int32_t poly_stride;      // floats between the centre / left / right tables (they share one allocation, as do the point counts)
  // cpm_mixed (world_state_rt_sim.py:313-358): the path lists of the sub-scenarios 1..n_lists and the cumulative distribution a finished env draws its
  // sub-scenario from (sigmaenv_set_scenario_lists); used by the device-side resets when the launch passes path_count = SIGMAENV_SCENARIO_LISTS
  int32_t n_lists;
  float cdf0, cdf1, cdf2;
  float neigh_radius_tight;  // the narrowest neighbour mask: the common case (an agent near its lane) tests the own chunk's +-4 neighbours in ONE pass of eight boxes
  unsigned long long list_first16, list_count16;  // first path / path count of list k in bits [16 k, 16 k + 16): selected by a shift (a select between
                                                  // members becomes a select between their ADDRESSES and puts the whole struct on the stack)
};
#define SIGMAENV_CHUNK 4
// Kernel-uniform quantities the step kernel used to derive from the config per lane and step (float64 conversions, a float64 division, a correctly rounded root:
// ~70 vector instructions per tile-step).  derive_config evaluates the SAME IEEE expressions once on the host (-ffp-contract=off on both sides; sqrtf is correctly
// rounded on both): the same bits.  The device copy of the config is a DevConfig; kernels that take `const sigmaenv_config_t*` read the derived block behind it.
struct CfgDerived {
  float l_wb;       // (float)((double)l_f + (double)l_r)                                   dynamics.py:62-192
  float k_beta;     // (float)((double)l_r / ((double)l_f + (double)l_r))
  float lh, wh;     // (float)((double)length / 2.0), (float)((double)width / 2.0)          helper_scenario.py:695-826
  float diag;       // sqrtf(world_x_dim * world_x_dim + world_y_dim * world_y_dim)         helper_scenario.py:1140-1143
  float rew_denom;  // (float)((double)max_speed * (double)dt)                               road_traffic.py:986-991
  float r_pos, r_v, r_dl;  // reciprocals of the default observation row's normalisers: 1 / (float)(length * 10), 1 / max_speed, 1 / (float)(lane_width * 3)
  float r_wx, r_wy;  // 1 / world_x_dim, 1 / world_y_dim (bird view's position normalisers)
  float min_d_sq;   // (sqrtf((float)(length^2 + width^2)) * 1.5f)^2: the minimum start distance of the resets, squared   road_traffic.py:679-684
};
struct DevConfig {
  sigmaenv_config_t c;
  CfgDerived d;
};
inline CfgDerived derive_config(const sigmaenv_config_t& c) {
  CfgDerived d;
  d.l_wb = (float)((double)c.l_f + (double)c.l_r);
  d.k_beta = (float)((double)c.l_r / ((double)c.l_f + (double)c.l_r));
  d.lh = (float)((double)c.length / 2.0);
  d.wh = (float)((double)c.width / 2.0);
  const float xx = c.world_x_dim * c.world_x_dim, yy = c.world_y_dim * c.world_y_dim;
  d.diag = sqrtf(xx + yy);
  d.rew_denom = (float)((double)c.max_speed * (double)c.dt);
  const float n_pos = (float)((double)c.length * 10.0), n_v = c.max_speed, n_dl = (float)((double)c.lane_width * 3.0);
  d.r_pos = 1.0f / n_pos; d.r_v = 1.0f / n_v; d.r_dl = 1.0f / n_dl;
  d.r_wx = 1.0f / c.world_x_dim; d.r_wy = 1.0f / c.world_y_dim;
  const float min_d = sqrtf((float)((double)c.length * (double)c.length + (double)c.width * (double)c.width)) * 1.5f;
  d.min_d_sq = min_d * min_d;
  return d;
}

// row of the start table (floats): everything reset_init_distances_and_short_term_ref_path derives for an agent standing on a
// centre-line point with the map's yaw there -- a pure function of (path, point), computed once at sigmaenv_create by the very
// kernels' own scan code and copied by the device-side resets
#define START_X 0        /* x, y, yaw */
#define START_COSV 3     /* cr_cos(0 + yaw), cr_sin(0 + yaw): velocity direction (world_state_rt_sim.py:203-207) */
#define START_VERT 5     /* 10 floats */
#define START_CS 15      /* cos / sin of the yaw as rect_vertices produces them */
#define START_DREF 17
#define START_DLEFT 18   /* 5 */
#define START_DRIGHT 23  /* 5 */
#define START_DBOUND 28
#define START_SHORT 29   /* 2 * NS */
#define START_CP (START_SHORT + 2 * NS)             /* 3 ints */
#define START_ROW ((START_CP + 3 + 3) & ~3)         /* whole float4s */

struct DevBufs {
  float *state, *prev_pos, *vertices, *short_term, *dist_ref, *dist_left, *dist_right, *dist_bound, *dist_agents;
  float *reward, *reward_info, *obs, *action, *cbf_nominal;
  int32_t *path, *closest, *nearing, *timer;
  uint8_t *col_agents, *col_flags, *done;
  unsigned long long* reset_mask;  // [B] bit i: agent i needs its derived state rebuilt
  uint8_t* reset_full;             // [B] full-env reset pending
  uint8_t* fresh;                  // optional [B,N] (SIGMAENV_OBS_BOUNDARY_POINTS only): 1 = (re)placed and not stepped since -- the observation's boundary
                                   // points use another index shift then (world_state_rt.py:531-576 vs :686-724); nullptr otherwise
  float* slab;                     // optional rollout record of this step: [B][N*D obs | N reward | 1 done] fp32 (sigmaenv_set_slab)
  // non-default observation switches (sigmaenv_config_t.obs_flags != 0; observe_tile): what the row assembly reads beyond the tile itself
  const float* lanelet_centers;           // [n_lanelets][lanelet_pts][2] zero-padded lanelet centre lines (sigmaenv_set_lanelets), or nullptr
  const unsigned long long* lanelet_neigh;  // [n_lanelets] neighbour bit masks
  int32_t n_lanelets, lanelet_pts;
  const float* bnd_left;                  // the padded boundary tables (DevMap::left; the right table follows at + bnd_poly_stride floats), [n_paths][bnd_P][2]
  const int32_t* bnd_n_center;            // point count of every path's centre line (the loop rule of the boundary points, world_state_rt.py:686-724)
  const uint8_t* bnd_is_loop;
  int32_t bnd_poly_stride, bnd_P;
  uint32_t mVT1;                          // magic multiplier of the variant rows' points per agent (observe_tile)
  uint32_t obs_salt;                      // 0 in every launch that steps or re-places envs; n in the n-th stand-alone sigmaenv_observe of the handle (obs_noise)
  // magic multipliers ceil(2^32 / d) for the divisors the kernels' index arithmetic divides by (fdiv below): agents per env, items per
  // agent of the two observation passes, unordered pairs per env, floats per rollout record row
  uint32_t mN, mT1, mT2, mTP, mW, mSG;  // mSG: slots of a full tile of the step kernel (G * N)
  int dbg_skip;                    // profile build only: phase-ablation mask of the step kernel (SIGMAENV_DEBUG_SKIP; results invalid when non-zero)
  unsigned long long* dbg_ts2;     // same for the auto-reset kernel (SIGMAENV_TIMESTAMPS=2)
  unsigned long long* dbg_ts;      // optional [grid][8] shader-clock timestamps at the phase boundaries (SIGMAENV_TIMESTAMPS=1)
};

// two consecutive polyline points (one segment) in ONE 16-byte load: the points are 8-byte aligned, which a global dwordx4 load
// accepts; halves the address-coalescer work of the scan compared with two 8-byte loads
struct __attribute__((packed, aligned(8))) Seg4 { float ax, ay, bx, by; };
__device__ __forceinline__ Seg4 load_segment(const float2* p, int k) { return *reinterpret_cast<const Seg4*>(p + k); }

// x / d for x * d < 2^32 with m = ceil(2^32 / d) (d = 1: m wraps to 0): two instructions instead of the ~20 of a runtime division
__host__ __device__ constexpr uint32_t magic_u32(unsigned d) { return d <= 1u ? 0u : (uint32_t)(((1ull << 32) + d - 1ull) / d); }
__device__ __forceinline__ int fdiv(int x, uint32_t m) { return m ? (int)__umulhi((uint32_t)x, m) : x; }

// four consecutive floats stored with ONE 16-byte store at 4-byte alignment (global memory accepts it): rows of the rollout record
// have an odd number of floats
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };

// ---- scalar helpers ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float cr_sin(float x) { return sigma_sinf(x); }
__device__ __forceinline__ float cr_cos(float x) { return sigma_cosf(x); }
__device__ __forceinline__ float cr_tan(float x) { return sigma_tanf(x); }
__device__ __forceinline__ float cr_atan(float x) { return sigma_atanf(x); }
__device__ __forceinline__ void cr_sincos(float x, float& s, float& c) { sigma_sincosf(x, &s, &c); }
__device__ __forceinline__ float norm2(float x, float y) { return sqrtf(fmaf(y, y, x * x)); }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
// fminf WITHOUT the canonicalisation the compiler puts in front of it (v_max_f32 x, x, x: quiets a signalling NaN before llvm.minnum): v_min_f32 in IEEE mode already returns the
// other operand for a NaN of either kind -- the same value for every input that is not a SIGNALLING NaN, which no arithmetic result ever is.  One instruction instead of two on the
// running minima of the scan (4 per segment).
__device__ __forceinline__ float fmin_nc(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float remainder_pos(float a, float b) {  // torch.remainder, b > 0
  // 0 <= a < b: fmod returns a itself, exactly (the steering-angle wrap of every step lands here); the general routine -- a loop over the exponent
  // difference -- only runs for the lanes that need it
  if (a >= 0.0f && a < b) return a;
  float m = fmodf(a, b);
  if (m != 0.0f && m < 0.0f) m += b;
  return m;
}
__device__ __forceinline__ float angle_eliminate_two_pi(float a) {  // helper_scenario.py:1276-1289
  float r = remainder_pos(a, TWO_PI32);
  if (r > PI32) r -= TWO_PI32;
  return r;
}
__device__ __forceinline__ float decreasing_lin(float x, float x0, float x1) {  // helper_scenario.py:960-996
  x = clampf(x, x0, x1);
  float denom = x1 - x0;
  return 1.0f - (x - x0) / denom;
}

// ---- K1: WorldCustom.step + KinematicBicycleModel (helper_training.py:797-861, dynamics.py:62-192) ----------------
__device__ inline void bicycle_step(const sigmaenv_config_t& c, float s[8], float u0, float u1, float uc[2]) {
  float a0 = clampf(u0, -c.max_speed, c.max_speed);
  float a1 = clampf(u1, -c.max_steering, c.max_steering);
  uc[0] = a0;
  uc[1] = a1;
  float x = s[0], y = s[1], psi = s[2], v = s[3], delta = s[4];
  float u_acc = (a0 - v) / c.dt;
  float u_sr = (a1 - delta) / c.dt;
  u_acc = clampf(u_acc, c.min_acc, c.max_acc);
  u_sr = clampf(u_sr, c.min_steering_rate, c.max_steering_rate);
  float l_wb = (float)((double)c.l_f + (double)c.l_r);
  float k_beta = (float)((double)c.l_r / ((double)c.l_f + (double)c.l_r));
  float tan_d = cr_tan(delta);
  float beta = cr_atan(k_beta * tan_d);
  float s0, c0;
  cr_sincos(psi + beta, s0, c0);
  float dx0 = v * c0;
  float dx1 = v * s0;
  float dx2 = (v / l_wb) * tan_d * cr_cos(beta);
  float dt = c.dt;
  x = x + dt * dx0;
  y = y + dt * dx1;
  psi = psi + dt * dx2;
  v = v + dt * u_acc;
  delta = delta + dt * u_sr;
  delta = remainder_pos(delta + PI32, TWO_PI32) - PI32;
  float beta1 = cr_atan(k_beta * cr_tan(delta));
  float course = psi + beta1;
  s[0] = x; s[1] = y; s[2] = psi; s[3] = v; s[4] = delta;
  float s1, c1;
  cr_sincos(course, s1, c1);
  s[5] = v * c1;
  s[6] = v * s1;
  s[7] = beta1;
}

// ---- K2: get_rectangle_vertices (helper_scenario.py:695-826) ------------------------------------------------------
__device__ inline void rect_vertices(const sigmaenv_config_t& c, float px, float py, float psi, float* v /*5x2*/, float* cs_out = nullptr) {
  float lh = (float)((double)c.length / 2.0), wh = (float)((double)c.width / 2.0);
  float cs, sn;
  cr_sincos(psi, sn, cs);
  if (cs_out) { cs_out[0] = cs; cs_out[1] = sn; }
  float nsn = -sn;
  const float bx[5] = {lh, lh, -lh, -lh, lh};
  const float by[5] = {wh, -wh, -wh, wh, wh};
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    v[2 * k] = (cs * bx[k] + nsn * by[k]) + px;
    v[2 * k + 1] = (sn * bx[k] + cs * by[k]) + py;
  }
}

// point -> segment distance, helper_scenario.py:858-872
__device__ __forceinline__ float point_segment(float px, float py, float sx, float sy, float lx, float ly, float len2) {
  float vx = px - sx, vy = py - sy;
  float proj = (vx * lx + vy * ly) / len2;
  float t = clampf(proj, 0.0f, 1.0f);
  float cx = sx + lx * t, cy = sy + ly * t;
  return norm2(cx - px, cy - py);
}

// squared form: torch.norm == sqrt(fma(ey,ey,ex*ex)); sqrt is monotone and correctly rounded, so min_k sqrt(s_k) == sqrt(min_k s_k)
// bit-for-bit, which lets the corner queries (value only, no index) defer the sqrt to after the reduction.
__device__ __forceinline__ float point_segment_sq(float px, float py, float sx, float sy, float lx, float ly, float len2) {
  float vx = px - sx, vy = py - sy;
  float proj = (vx * lx + vy * ly) / len2;
  float t = clampf(proj, 0.0f, 1.0f);
  float cx = sx + lx * t, cy = sy + ly * t;
  float ex = cx - px, ey = cy - py;
  return fmaf(ey, ey, ex * ex);
}

// Division by a denominator shared between several numerators (the five queries of one segment divide by the same |l|^2).
// This is the compiler's own IEEE fp32 division sequence (rcp, one Newton step, quotient, two fma corrections) with the
// reciprocal hoisted and WITHOUT the v_div_scale / v_div_fixup wrapping, which only matters when the denominator is denormal or
// above 2^126, the quotient is denormal, the exponents differ by 96 or more, or the numerator is below 2^-103.  The map table is
// checked at sigmaenv_create (2^-60 <= |l|^2 <= 2^60, else fast_div is off and the plain `/` is used); numerators are dot products
// of fp32 coordinate differences, i.e. exactly zero or far above 2^-103.  In that regime the result is the correctly rounded
// quotient, bit-identical to `/`.
__device__ __forceinline__ float shared_rcp(float b) {
  float y0 = __builtin_amdgcn_rcpf(b);
  float e = fmaf(-b, y0, 1.0f);
  return fmaf(e, y0, y0);
}
__device__ __forceinline__ float div_shared(float a, float b, float y) {
  float q = a * y;
  float r = fmaf(-b, q, a);
  q = fmaf(r, y, q);
  r = fmaf(-b, q, a);
  return fmaf(r, y, q);
}
template <bool FAST>
__device__ __forceinline__ float point_segment_t(float px, float py, float sx, float sy, float lx, float ly, float len2, float rcp) {
  float vx = px - sx, vy = py - sy;
  float num = vx * lx + vy * ly;
  float proj = FAST ? div_shared(num, len2, rcp) : num / len2;
  float t = clampf(proj, 0.0f, 1.0f);
  float cx = sx + lx * t, cy = sy + ly * t;
  return norm2(cx - px, cy - py);
}
template <bool FAST>
__device__ __forceinline__ float point_segment_sq_t(float px, float py, float sx, float sy, float lx, float ly, float len2, float rcp) {
  float vx = px - sx, vy = py - sy;
  float num = vx * lx + vy * ly;
  float proj = FAST ? div_shared(num, len2, rcp) : num / len2;
  float t = clampf(proj, 0.0f, 1.0f);
  float cx = sx + lx * t, cy = sy + ly * t;
  float ex = cx - px, ey = cy - py;
  return fmaf(ey, ey, ex * ex);
}

// one rectangle edge against one polyline segment, helper_scenario.py:1165-1196
struct Edge {
  float xa, ya, xb, yb, dx, dy, S;
};
__device__ __forceinline__ Edge make_edge(float xa, float ya, float xb, float yb) {
  Edge e;
  e.xa = xa; e.ya = ya; e.xb = xb; e.yb = yb;
  e.dx = xb - xa;
  e.dy = yb - ya;
  e.S = e.dx * ya - e.dy * xa;
  return e;
}
__device__ __forceinline__ bool edge_hits_segment(const Edge& e, float x2a, float y2a, float x2b, float y2b, float dx2, float dy2, float S2) {
  float ma = e.dx * y2a - e.dy * x2a, mb = e.dx * y2b - e.dy * x2b;
  bool C1 = ((ma - e.S) * (mb - e.S)) < 0.0f;
  float wa = e.ya * dx2 - e.xa * dy2, wb = e.yb * dx2 - e.xb * dy2;
  bool C2 = ((wa - S2) * (wb - S2)) < 0.0f;
  return C1 && C2;
}
// interX of a closed rectangle against one 2-point segment (entry / exit), world_state_rt_sim.py:413-424
__device__ inline bool interx_rect_seg(const float* v, float x2a, float y2a, float x2b, float y2b) {
  float dx2 = x2b - x2a, dy2 = y2b - y2a;
  float S2 = dx2 * y2a - dy2 * x2a;
  bool hit = false;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    Edge e = make_edge(v[2 * i], v[2 * i + 1], v[2 * i + 2], v[2 * i + 3]);
    hit |= edge_hits_segment(e, x2a, y2a, x2b, y2b, dx2, dy2, S2);
  }
  return hit;
}

// ---- K4: mtv distance (helper_scenario.py:1030-1138) ---------------------------------------------------------------
__device__ inline void rect_axes(const float* v, float ax[2][2]) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float ex = v[2 * (k + 1)] - v[2 * k], ey = v[2 * (k + 1) + 1] - v[2 * k + 1];
    float nrm = norm2(ex, ey);
    ax[k][0] = ex / nrm;
    ax[k][1] = ey / nrm;
  }
}
__device__ inline void mtv_half(const float* va, const float* vb, const float axb[2][2], float& pos_min, float& omin_out, bool& any_inside) {
  float maxbb[2], minbb[2], maxab[2], minab[2], pab[4][2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    maxbb[k] = -INFINITY; minbb[k] = INFINITY; maxab[k] = -INFINITY; minab[k] = INFINITY;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float pb = vb[2 * v] * axb[k][0] + vb[2 * v + 1] * axb[k][1];
      float pa = va[2 * v] * axb[k][0] + va[2 * v + 1] * axb[k][1];
      pab[v][k] = pa;
      maxbb[k] = fmaxf(maxbb[k], pb); minbb[k] = fminf(minbb[k], pb);
      maxab[k] = fmaxf(maxab[k], pa); minab[k] = fminf(minab[k], pa);
    }
  }
  float ov0 = fminf(maxbb[0], maxab[0]) - fmaxf(minbb[0], minab[0]);
  float ov1 = fminf(maxbb[1], maxab[1]) - fmaxf(minbb[1], minab[1]);
  float omin = fminf(ov0, ov1);
  omin_out = omin;
  bool inside_any = false;
  float pm = INFINITY;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    float g[2];
    bool inside = true;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float p = pab[v][k];
      g[k] = (p - minbb[k]) * (p <= minbb[k] ? 1.0f : 0.0f) + (maxbb[k] - p) * (p >= maxbb[k] ? 1.0f : 0.0f);
      inside = inside && (p > minbb[k]) && (p < maxbb[k]);
    }
    pm = fminf(pm, norm2(g[0], g[1]));
    float neg = -omin * (inside ? 1.0f : 0.0f);
    if (fabsf(neg) > 0.0f) inside_any = true;
  }
  pos_min = pm;
  any_inside = inside_any;
}
// The same with the per-RECTANGLE quantities taken from a staged record instead of being recomputed for every pair the rectangle is part of: its two unit
// edge axes (a correctly rounded root and two IEEE divisions each) and the extents of its own vertices on them.  rect_mtv_record evaluates them with the very
// operations of rect_axes / mtv_half, in the same order: mtv_pair_staged(vi, vj, rec_i, rec_j) == mtv_pair(vi, vj) bit for bit.
// record: ax[0][0], ax[0][1], ax[1][0], ax[1][1], minbb[0], minbb[1], maxbb[0], maxbb[1]
#define MTV_REC 8
__device__ inline void rect_mtv_record(const float* v, float* rec) {
  float ax[2][2];
  rect_axes(v, ax);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float mx = -INFINITY, mn = INFINITY;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float pb = v[2 * q] * ax[k][0] + v[2 * q + 1] * ax[k][1];
      mx = fmaxf(mx, pb); mn = fminf(mn, pb);
    }
    rec[2 * k] = ax[k][0]; rec[2 * k + 1] = ax[k][1];
    rec[4 + k] = mn; rec[6 + k] = mx;
  }
}
__device__ inline void mtv_half_staged(const float* va, const float* recb, float& pos_min, float& omin_out, bool& any_inside) {
  float maxbb[2], minbb[2], maxab[2], minab[2], pab[4][2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    maxbb[k] = recb[6 + k]; minbb[k] = recb[4 + k]; maxab[k] = -INFINITY; minab[k] = INFINITY;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float pa = va[2 * v] * recb[2 * k] + va[2 * v + 1] * recb[2 * k + 1];
      pab[v][k] = pa;
      maxab[k] = fmaxf(maxab[k], pa); minab[k] = fminf(minab[k], pa);
    }
  }
  float ov0 = fminf(maxbb[0], maxab[0]) - fmaxf(minbb[0], minab[0]);
  float ov1 = fminf(maxbb[1], maxab[1]) - fmaxf(minbb[1], minab[1]);
  float omin = fminf(ov0, ov1);
  omin_out = omin;
  bool inside_any = false;
  float pm = INFINITY;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    float g[2];
    bool inside = true;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float p = pab[v][k];
      g[k] = (p - minbb[k]) * (p <= minbb[k] ? 1.0f : 0.0f) + (maxbb[k] - p) * (p >= maxbb[k] ? 1.0f : 0.0f);
      inside = inside && (p > minbb[k]) && (p < maxbb[k]);
    }
    pm = fminf(pm, norm2(g[0], g[1]));
    float neg = -omin * (inside ? 1.0f : 0.0f);
    if (fabsf(neg) > 0.0f) inside_any = true;
  }
  pos_min = pm;
  any_inside = inside_any;
}
__device__ inline float mtv_pair_staged(const float* vi, const float* vj, const float* reci, const float* recj) {
  float pij, pji, omin_j, omin_i;
  bool neg_ij, neg_ji;
  mtv_half_staged(vi, recj, pij, omin_j, neg_ij);
  mtv_half_staged(vj, reci, pji, omin_i, neg_ji);
  float d = fminf(pij, pji);
  if (neg_ij || neg_ji) d = -fminf(omin_j, omin_i);
  return d;
}
__device__ inline float mtv_pair(const float* vi, const float* vj) {
  float axi[2][2], axj[2][2], pij, pji, omin_j, omin_i;
  bool neg_ij, neg_ji;
  rect_axes(vi, axi);
  rect_axes(vj, axj);
  mtv_half(vi, vj, axj, pij, omin_j, neg_ij);
  mtv_half(vj, vi, axi, pji, omin_i, neg_ji);
  float d = fminf(pij, pji);
  if (neg_ij || neg_ji) d = -fminf(omin_j, omin_i);
  return d;
}

// ---- K6: short-term reference path (helper_scenario.py:892-957) ----------------------------------------------------
__device__ inline void short_term_path(const float* center, int n, bool is_loop, int cp, float* out /*NSx2*/) {
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    int id = k * 2 + cp + 1;
    if (is_loop && id >= n - 1) id = (id + 1) % n;
    out[2 * k] = center[2 * id];
    out[2 * k + 1] = center[2 * id + 1];
  }
}

// ---- wave-level (64 lanes) reductions -------------------------------------------------------------------------------
// lexicographic (distance, index) minimum: torch.min returns the first minimal index (helper_scenario.py:883)
__device__ __forceinline__ void wave_argmin(float& d, int& k) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    float od = __shfl_xor(d, off, 64);
    int ok = __shfl_xor(k, off, 64);
    if (od < d || (od == d && ok < k)) { d = od; k = ok; }
  }
}

// Row-of-16 reductions through DPP lane permutations (no LDS crossbar): quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror,
// row_mirror -- a commutative / associative combine over these four patterns leaves the full 16-lane result in every lane.
// Six row-of-16 minima at once with the DPP operand folded into v_min_f32 (one instruction per step and value instead of
// v_mov_dpp + canonicalise + v_min).  The six chains are interleaved step-major, so that a value is read through DPP five
// instructions after the VALU write that produced it (the hardware needs two wait states there); the leading s_nop covers the
// producers of the inputs.  Inputs are distances (>= 0 or +inf, never NaN for finite states).
#define SIGMA_DPP_ROW6(OP, CTRL) \
  OP " %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t" OP " %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
  OP " %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t" OP " %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
  OP " %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf\n\t" OP " %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void row16_min6(float& a, float& b, float& c, float& d, float& e, float& f) {
  asm volatile("s_nop 1\n\t"
               SIGMA_DPP_ROW6("v_min_f32_dpp", "quad_perm:[1,0,3,2]")
               SIGMA_DPP_ROW6("v_min_f32_dpp", "quad_perm:[2,3,0,1]")
               SIGMA_DPP_ROW6("v_min_f32_dpp", "row_half_mirror")
               SIGMA_DPP_ROW6("v_min_f32_dpp", "row_mirror")
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}
// three minima at once (the same pattern; three chains leave two instructions between a write and its DPP read: one s_nop per step)
#define SIGMA_DPP_ROW3(OP, CTRL) \
  OP " %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t" OP " %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
  OP " %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
__device__ __forceinline__ void row16_min3(float& a, float& b, float& c) {
  asm volatile("s_nop 1\n\t"
               SIGMA_DPP_ROW3("v_min_f32_dpp", "quad_perm:[1,0,3,2]")
               SIGMA_DPP_ROW3("v_min_f32_dpp", "quad_perm:[2,3,0,1]")
               SIGMA_DPP_ROW3("v_min_f32_dpp", "row_half_mirror")
               SIGMA_DPP_ROW3("v_min_f32_dpp", "row_mirror")
               : "+v"(a), "+v"(b), "+v"(c));
}
// ONE minimum (a single chain: two wait states between dependent steps)
__device__ __forceinline__ void row16_min1(float& a) {
  asm volatile("s_nop 1\n\t"
               "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
               : "+v"(a));
}
// two row-of-16 minima of non-negative ints (two interleaved chains, one s_nop between dependent steps)
#define SIGMA_DPP_ROW2(OP, CTRL) \
  OP " %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t" OP " %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
__device__ __forceinline__ void row16_min2_i32(int& a, int& b) {
  asm volatile("s_nop 1\n\t"
               SIGMA_DPP_ROW2("v_min_i32_dpp", "quad_perm:[1,0,3,2]")
               SIGMA_DPP_ROW2("v_min_i32_dpp", "quad_perm:[2,3,0,1]")
               SIGMA_DPP_ROW2("v_min_i32_dpp", "row_half_mirror")
               SIGMA_DPP_ROW2("v_min_i32_dpp", "row_mirror")
               : "+v"(a), "+v"(b));
}

__device__ __forceinline__ void row16_max2(float& a, float& b) {
  asm volatile("s_nop 1\n\t"
               SIGMA_DPP_ROW2("v_max_f32_dpp", "quad_perm:[1,0,3,2]")
               SIGMA_DPP_ROW2("v_max_f32_dpp", "quad_perm:[2,3,0,1]")
               SIGMA_DPP_ROW2("v_max_f32_dpp", "row_half_mirror")
               SIGMA_DPP_ROW2("v_max_f32_dpp", "row_mirror")
               : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void row16_or2_u32(unsigned& a, unsigned& b) {
  asm volatile("s_nop 1\n\t"
               SIGMA_DPP_ROW2("v_or_b32_dpp", "quad_perm:[1,0,3,2]")
               SIGMA_DPP_ROW2("v_or_b32_dpp", "quad_perm:[2,3,0,1]")
               SIGMA_DPP_ROW2("v_or_b32_dpp", "row_half_mirror")
               SIGMA_DPP_ROW2("v_or_b32_dpp", "row_mirror")
               : "+v"(a), "+v"(b));
}

// counter-based RNG (specification shared with the oracle): 32-bit multiplicative mix + murmur3 finalisers over (seed, counter, env, agent, draw)
__device__ __forceinline__ uint32_t rng_u32(uint64_t seed, uint64_t counter, uint32_t env, uint32_t agent, uint32_t draw) {
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

// observation noise of element k of agent `agent` of env `env_global` at the env's counters (episodes_reset, timer.step): level * U[0, 1)
// (observation_provider_rt.py:613-618; specification shared with the oracle, see sigmaenv_config_t.obs_noise_level).  The reference draws rand_like on EVERY
// observation() call: the observations a step or a reset produces are told apart by the env's own counters, and an observation that is TAKEN AGAIN at the same
// counters (sigmaenv_observe) by `salt`, the number of that call on its handle -- so it gets draws of its own, as the reference's second call does.
__device__ __forceinline__ float obs_noise(const sigmaenv_config_t& c, int env_global, int agent, int k, int episodes, int step, uint32_t salt) {
  const uint64_t seed = ((uint64_t)c.obs_noise_seed_hi << 32) | c.obs_noise_seed_lo;
  const uint64_t counter = (uint64_t)(uint32_t)episodes * 65537ull + (uint64_t)(uint32_t)step + (uint64_t)(salt * 0x632BE5ABu);
  const float u = (float)(rng_u32(seed, counter, (uint32_t)env_global, (uint32_t)agent, 9000u + (uint32_t)k) >> 8) * (1.0f / 16777216.0f);
  return c.obs_noise_level * u;
}

}  // namespace sigmadev
