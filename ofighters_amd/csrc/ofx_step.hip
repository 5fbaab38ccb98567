// ofx_step.hip - one lock-step of N arenas: Agent.step bookkeeping, GUI laser
// clean-up, Laser.move for every laser in list order, Ship.move in index order.
//
// Reference semantics (paths under /root/reference/ofighters):
//   lib/battleground.py:146-160  request_actions / generate_frame ordering
//   lib/laser.py:36-62           Laser.move (advance, ordered hit loop, outside)
//   lib/ship.py:303-339          Ship.move   (pointing, thrust, shoot)
//   lib/ship.py:213-222          Ship.thrust (int truncation + clamp)
//   lib/ship.py:134-210          Ship.shoot + aim / trajectory rewards
//   lib/form.py:148-188,298-307  collide / edge / angular_radius
//   agents/agent.py:66-74        score += reward; reward = 0 (dead ships too)
//   lib/ofighters.py:619-625,702-707  destroyed lasers leave the list before the next tick
//
// Mapping: ONE 64-lane wavefront per arena (4 arenas per 256-thread block, no
// inter-wave communication).  Lanes are lasers during the laser phase, ships
// while ships move, and (shooter, enemy) pairs during the reward tests.  The
// reference's sequential "first collider in list order kills, immediately"
// rule becomes: per ship, ballot over the laser lanes and take the lowest set
// bit; the ship leaves the wave-uniform alive mask before later chunks of 64
// lasers are tested.  Laser compaction is a ballot + popcount prefix, stable,
// in place (a chunk's stores land at or below its own load indices).
//
// Arithmetic is fp64 with contraction off (build flag -ffp-contract=off):
// positions are bit-identical to CPython's.  The collide test
// `sqrt(d2) <= R` is evaluated as `d2 <= T` with T the largest double whose
// correctly rounded square root is <= R (host-computed; rn o sqrt is
// monotone, so the two predicates are the same set).
#include "ofx_internal.h"

struct StepParams {
  int N, M, L, W, H;
  int ship_radius, laser_radius, ship_speed, laser_speed;
  int r_death, r_kill, r_aim, r_traj;
  double hit_thresh;
  ofx_state st;
  const ofx_action *actions;   // BOTS = false: the caller's actions [N][M]
  // BOTS = true (ofx_rollout): the scripted bots' action law evaluated in the kernel (agents/agent.py:99-155 on the
  // counter RNG, the same draws as k_bots) for lock-steps tick0 .. tick0 + n_ticks - 1, all inside ONE launch
  const int32_t *beh;          // device [M] OFX_BOT_*
  uint32_t k0, k1, tick0;
  int n_ticks, arena_base;
};

// CPython float_rem: fmod, then the result takes the divisor's sign (b > 0 here)
__device__ inline double py_fmod_pos(double a, double b) {
  double m = fmod(a, b);
  if (m != 0.0) {
    if (m < 0) m += b;
  } else {
    m = 0.0;
  }
  return m;
}

__device__ inline int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// The scripted bots' law for ship `i` of global arena `ga` at lock-step `tick` - the arithmetic of k_bots
// (ofx_api.hip), evaluated by the ship's own lane: `alive` / (ptx, pty) are the ship's state when request_actions
// runs, i.e. before this lock-step's lasers move (battleground.py:146-150).
__device__ inline ofx_action bot_action(int beh, uint32_t ga, uint32_t i, uint32_t tick, uint32_t k0, uint32_t k1, int W,
                                        int H, bool alive, int ptx, int pty) {
  uint32_t r[4];
  ofx_philox4x32_10(ga, i, tick, OFX_STREAM_BOT, k0, k1, r);
  bool shoot = false, thrust = false, repoint = false;
  const double u0 = (double)r[0] * (1.0 / 4294967296.0), u1 = (double)r[1] * (1.0 / 4294967296.0);
  switch (beh) {
    case OFX_BOT_RANDOM: {
      const uint32_t k = (uint32_t)(((uint64_t)r[0] * 3u) >> 32);
      shoot = k == 0; thrust = k == 1; repoint = k == 2;
    } break;
    case OFX_BOT_TURRET: shoot = u0 < 0.8; repoint = u1 < 0.3; break;
    case OFX_BOT_RUNNER: thrust = u0 < 0.9; repoint = u1 < 0.1; break;
    case OFX_BOT_THRUST: thrust = true; break;
    case OFX_BOT_SHOOT: shoot = true; break;
    default: break;
  }
  ofx_action act;
  act.px = ptx; act.py = pty;
  act.shoot = 0; act.thrust = 0; act.valid = alive ? 1 : 0; act._pad = 0;
  if (alive) {
    act.shoot = shoot; act.thrust = thrust;
    if (repoint) { act.px = ofx_draw_int(r[2], W); act.py = ofx_draw_int(r[3], H); }
  }
  return act;
}

template <bool BOTS>
__global__ __launch_bounds__(256) void k_step(StepParams p) {
  const int lane = threadIdx.x & 63;
  const int a = blockIdx.x * OFX_ARENAS_PER_BLOCK + (threadIdx.x >> 6);
  if (a >= p.N) return;  // wave-uniform; the kernel has no block-level barrier
  const int M = p.M, L = p.L;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const bool is_ship = lane < M;
  const size_t si = (size_t)a * M + (is_ship ? lane : 0);
  const int my_beh = BOTS && is_ship ? p.beh[lane] : 0;
  const int n_ticks = BOTS ? p.n_ticks : 1;
#pragma unroll 1
  for (int tk = 0; tk < n_ticks; tk++) {
  // An arena belongs to ONE wave for the whole launch, so the lock-steps of a K-tick launch only need this wave's own
  // stores of lock-step t to be visible to its loads of lock-step t + 1: program order within a wavefront.
  if (BOTS && tk) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

  // ---- ships into registers (lane i == ship i) + Agent.step bookkeeping ----
  int x = 0, y = 0, ptx = 0, pty = 0, hull = 0, rew = 0, sc = 0, killer = -1;
  bool alive = false;
  if (is_ship) {
    x = p.st.ship_x[si];
    y = p.st.ship_y[si];
    ptx = p.st.ship_px[si];
    pty = p.st.ship_py[si];
    hull = p.st.hull[si];
    rew = p.st.reward[si];
    sc = p.st.score[si];
    alive = p.st.alive[si] != 0;
    p.st.obs_reward[si] = rew;  // what obs.reward shows this tick (observation.py:103)
    sc += rew;                  // agent.py:73-74
    rew = 0;
  }
  ofx_action bot_act;           // request_actions comes before generate_frame: the state the lasers have not touched yet
  if (BOTS) bot_act = bot_action(my_beh, (uint32_t)(p.arena_base + a), (uint32_t)lane, p.tick0 + (uint32_t)tk, p.k0, p.k1,
                                 p.W, p.H, alive, ptx, pty);
  unsigned long long alive_mask = __ballot(alive);

  // ---- laser phase ----
  const int n = p.st.n_lasers[a];
  const size_t lbase = (size_t)a * L;
  int out = 0;
  for (int base = 0; base < n; base += 64) {
    const int j = base + lane;
    const bool act = j < n;
    double lx = 0, ly = 0, ldx = 0, ldy = 0;
    int own = 0;
    bool dead = true;
    if (act) {
      lx = p.st.laser_x[lbase + j];
      ly = p.st.laser_y[lbase + j];
      ldx = p.st.laser_dx[lbase + j];
      ldy = p.st.laser_dy[lbase + j];
      own = p.st.laser_owner[lbase + j];
      dead = p.st.laser_dead[lbase + j] != 0;
    }
    const bool keep = act && !dead;  // clear_wreckage of last tick's destroyed lasers
    const unsigned long long km = __ballot(keep);
    const int pos = out + __popcll(km & lt_mask);
    lx += ldx;  // laser.py:45-48 (dx = dy = 0 for a zero-length direction, laser.py:43)
    ly += ldy;
    bool exploded = false;
    unsigned long long am = alive_mask;
    while (am) {  // ships in index order; only the still-playable ones
      const int s = __ffsll((long long)am) - 1;
      am &= am - 1;
      const double ddx = lx - (double)rl(x, s), ddy = ly - (double)rl(y, s);
      const double d2 = ddx * ddx + ddy * ddy;
      const unsigned long long hm = __ballot(keep && d2 <= p.hit_thresh);
      if (hm) {
        const int first = __ffsll((long long)hm) - 1;  // lowest list index wins
        alive_mask &= ~(1ull << s);
        const int kpos = rl(pos, first), kown = rl(own, first);
        if (lane == first) exploded = true;        // no break: it may kill more ships
        if (lane == kown) rew += p.r_kill;          // laser.py:57 (owner may be dead)
        if (lane == s) {                            // Ship.hit -> explode, ship.py:127-131,225-230
          hull -= 1;
          rew += p.r_death;
          killer = kpos;
        }
      }
    }
    if (keep) {
      const bool outside = (lx < 0) || (ly < 0) || (lx >= (double)p.W) || (ly >= (double)p.H);
      p.st.laser_x[lbase + pos] = lx;
      p.st.laser_y[lbase + pos] = ly;
      p.st.laser_dx[lbase + pos] = ldx;
      p.st.laser_dy[lbase + pos] = ldy;
      p.st.laser_owner[lbase + pos] = (uint8_t)own;
      p.st.laser_dead[lbase + pos] = (exploded || outside) ? 1 : 0;
    }
    out += __popcll(km);
  }
  alive = is_ship && ((alive_mask >> lane) & 1ull);

  // ---- ship phase: every ship's own move is independent of the others ----
  const int oldx = x, oldy = y;
  bool shoots = false;
  int ex = 0, ey = 0;
  double ndx = 0, ndy = 0;
  if (is_ship) {
    const ofx_action act = BOTS ? bot_act : p.actions[si];
    if (act.valid && alive) {  // ship.py:308
      ptx = act.px;
      pty = act.py;
      if (act.thrust) {  // ship.py:213-222
        const long long dX = (long long)ptx - x, dY = (long long)pty - y;
        const double dist = sqrt((double)(dX * dX + dY * dY));
        if (dist != 0) {
          const double dx = (double)(dX * p.ship_speed) / dist;
          const double dy = (double)(dY * p.ship_speed) / dist;
          const int nx = (int)((double)x + dx), ny = (int)((double)y + dy);  // int(): toward 0
          x = min(p.W - 1, max(0, nx));
          y = min(p.H - 1, max(0, ny));
        }
      }
      if (act.shoot) {  // ship.py:134-150, form.py:159-188
        const long long dX = (long long)ptx - x, dY = (long long)pty - y;
        const long long d2 = dX * dX + dY * dY;
        const double dist = sqrt((double)d2);
        if (dist != 0) {
          const long long inR = p.ship_radius + p.laser_radius;
          ex = (int)((double)x + (double)(dX * inR) / dist);
          ey = (int)((double)y + (double)(dY * inR) / dist);
          int fx = ex, fy = ey;
          if (d2 <= inR * inR) {  // pointing inside the hit-box: fired = centre (ship.py:147-148)
            fx = x;
            fy = y;
          }
          const long long lX = (long long)ptx - fx, lY = (long long)pty - fy;
          const double ldist = sqrt((double)(lX * lX + lY * lY));
          if (ldist != 0) {  // laser.py:39-46, constant for the laser's life
            ndx = (double)(lX * p.laser_speed) / ldist;
            ndy = (double)(lY * p.laser_speed) / ldist;
          }
          shoots = true;
        }
      }
    }
  }

  // ---- aim / trajectory rewards: (shooter i, enemy k) pairs across the wave.
  // Ships move in index order, so shooter i sees the NEW position of k < i and
  // the OLD position of k > i (ship.py:158-161,171-174).
  const unsigned long long shoot_mask = __ballot(shoots);
  if (shoot_mask) {
    bool aimed_any = false, traj_any = false;
    const int pairs = M * M;
    for (int w0 = 0; w0 < pairs; w0 += 64) {
      const int pi = w0 + lane;
      const int i = pi / M, k = pi - i * M;
      const bool in = pi < pairs;
      const int ii = in ? i : 0, kk = in ? k : 0;
      const int s_x = __shfl(x, ii), s_y = __shfl(y, ii);
      const int s_px = __shfl(ptx, ii), s_py = __shfl(pty, ii);
      const int k_nx = __shfl(x, kk), k_ny = __shfl(y, kk);
      const int k_ox = __shfl(oldx, kk), k_oy = __shfl(oldy, kk);
      const bool valid = in && i != k && ((shoot_mask >> ii) & 1ull) && ((alive_mask >> kk) & 1ull);
      bool aimed = false, traj = false;
      if (valid) {
        const int kx = k < i ? k_nx : k_ox, ky = k < i ? k_ny : k_oy;
        // enemy_aimed: distance(enemy, pointing) <= radius, all ints (ship.py:166-169)
        const long long ax = (long long)kx - s_px, ay = (long long)ky - s_py;
        aimed = ax * ax + ay * ay <= (long long)p.ship_radius * p.ship_radius;
        // enemy_on_trajectory (ship.py:179-210)
        const double kPI = 3.141592653589793, k2PI = 6.283185307179586;
        const double sa = atan2((double)(s_py - s_y), (double)(s_px - s_x)) + kPI;
        const double ta = atan2((double)(ky - s_y), (double)(kx - s_x)) + kPI;
        if (sa != 0.0 && ta != 0.0) {
          const long long ex2 = (long long)s_x - kx, ey2 = (long long)s_y - ky;
          const double dist = sqrt((double)(ex2 * ex2 + ey2 * ey2));
          const double ar = (dist == 0) ? k2PI : atan((double)p.ship_radius / dist);
          const double sup = py_fmod_pos(ta + ar, k2PI);
          const double inf = py_fmod_pos(ta + -ar, k2PI);
          traj = (inf <= sa) && (sa <= sup);
        }
      }
      const unsigned long long bm_a = __ballot(aimed), bm_t = __ballot(traj);
      // lane i (ship i) owns pairs [i*M, i*M+M): pick its bits out of this window
      const int lo = max(lane * M, w0) - w0, hi = min(lane * M + M, w0 + 64) - w0;
      if (is_ship && lo < hi) {
        const unsigned long long mine = (hi - lo >= 64) ? ~0ull : (((1ull << (hi - lo)) - 1ull) << lo);
        aimed_any = aimed_any || (bm_a & mine);
        traj_any = traj_any || (bm_t & mine);
      }
    }
    if (shoots) {
      if (aimed_any) rew += p.r_aim;
      if (traj_any) rew += p.r_traj;
    }
    // ---- append the new lasers in ship order ----
    const int slot = out + __popcll(shoot_mask & lt_mask);
    if (shoots) {
      if (slot < L) {
        p.st.laser_x[lbase + slot] = (double)ex;
        p.st.laser_y[lbase + slot] = (double)ey;
        p.st.laser_dx[lbase + slot] = ndx;
        p.st.laser_dy[lbase + slot] = ndy;
        p.st.laser_owner[lbase + slot] = (uint8_t)lane;
        p.st.laser_dead[lbase + slot] = 0;
      } else {
        atomicAdd(p.st.overflow, 1ull);  // counted, never silent
      }
    }
    out = min(L, out + __popcll(shoot_mask));
  }

  // ---- write back ----
  if (is_ship) {
    p.st.ship_x[si] = x;
    p.st.ship_y[si] = y;
    p.st.ship_px[si] = ptx;
    p.st.ship_py[si] = pty;
    p.st.hull[si] = hull;
    p.st.reward[si] = rew;
    p.st.score[si] = sc;
    p.st.alive[si] = alive ? 1 : 0;
    p.st.killer[si] = (int16_t)killer;
  }
  if (lane == 0) {
    p.st.n_lasers[a] = out;
    p.st.time[a] += 1;
  }
  }  // lock-steps of this launch
}

static StepParams step_params(ofx_handle *h) {
  StepParams p;
  const ofx_config &c = h->cfg;
  p.N = c.n_arenas; p.M = c.n_ships; p.L = c.laser_cap; p.W = c.width; p.H = c.height;
  p.ship_radius = c.ship_radius; p.laser_radius = c.laser_radius;
  p.ship_speed = c.ship_speed; p.laser_speed = c.laser_speed;
  p.r_death = c.reward_death; p.r_kill = c.reward_kill; p.r_aim = c.reward_aim; p.r_traj = c.reward_trajectory;
  p.hit_thresh = h->hit_thresh;
  p.st = h->st;
  p.actions = nullptr; p.beh = nullptr; p.k0 = p.k1 = p.tick0 = 0; p.n_ticks = 1; p.arena_base = c.arena_base;
  return p;
}

int ofx_launch_step(ofx_handle *h, const ofx_action *actions) {
  StepParams p = step_params(h);
  p.actions = actions;
  const int blocks = (h->cfg.n_arenas + OFX_ARENAS_PER_BLOCK - 1) / OFX_ARENAS_PER_BLOCK;
  hipLaunchKernelGGL(k_step<false>, dim3(blocks), dim3(256), 0, h->stream, p);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

// n_ticks lock-steps of bots + step in ONE launch (h->bot_behaviours already holds the behaviours)
int ofx_launch_step_bots(ofx_handle *h, uint64_t seed, uint32_t tick0, int n_ticks) {
  StepParams p = step_params(h);
  p.beh = h->bot_behaviours;
  p.k0 = (uint32_t)seed; p.k1 = (uint32_t)(seed >> 32); p.tick0 = tick0; p.n_ticks = n_ticks;
  const int blocks = (h->cfg.n_arenas + OFX_ARENAS_PER_BLOCK - 1) / OFX_ARENAS_PER_BLOCK;
  hipLaunchKernelGGL(k_step<true>, dim3(blocks), dim3(256), 0, h->stream, p);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}
