/*
 * ofx_oracle.c - CPU restatement of the reference's hot path.  TEST
 * INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg as the checker / reported baseline; never by the
 * product path (ofighters_amd/), which has no CPU fallback.
 *
 * Parity status: PINNED for the step, the rasteriser and the scratch MLP by
 * fixtures captured from the live reference (tests/golden/, generator
 * oracle/gen_golden.py, run in the build container under python3.9 + numpy
 * 1.26.4 + scikit-image 0.18.3).
 *
 * One arena, array-of-structs, sequential loops in the reference's own order;
 * Python float == IEEE double, Python int == long.  Build with
 *   gcc -O2 -ffp-contract=off -fno-builtin -fPIC -shared   (see oracle/Makefile)
 * -fno-builtin matters: CPython evaluates `float ** 2` with libm pow(), which
 * differs from x*x by 1 ulp in ~1e-3 of cases on glibc 2.35; gcc would fold
 * pow(x,2.0) into x*x.
 *
 * Citations are file:line under /root/reference/ofighters.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct orc_cfg {
  int32_t n_ships, width, height;
  int32_t ship_radius, laser_radius, ship_speed, laser_speed;
  int32_t reward_death, reward_kill, reward_aim, reward_trajectory;
} orc_cfg;

typedef struct {
  long x, y;     /* Ship.body  (always Python ints)           ship.py:43,221 */
  long px, py;   /* Ship.pointing                             ship.py:53     */
  long hull;     /*                                           ship.py:45     */
  int playable;  /* state not in destroyed/wreckage           ship.py:108    */
  long reward, score; /* Agent.reward / Agent.score           agent.py:23-25 */
  long obs_reward;    /* obs.reward seen by Agent.step        observation.py:103 */
  long last_score;    /* Agent.scores[-1]                     agent.py:62    */
  long time;
} orc_ship;

typedef struct {
  double x, y;       /* Laser.body                            laser.py:23    */
  long fx, fy;       /* Laser.fired                           laser.py:27, ship.py:148 */
  long tx, ty;       /* Laser.pointing                        laser.py:28    */
  int owner;         /* index of Laser.owner                                 */
  int destroyed;     /* state == "destroyed"                  laser.py:66    */
  long time;
} orc_laser;

typedef struct orc_arena {
  orc_cfg cfg;
  long time;
  orc_ship *ships;
  orc_laser *lasers;
  int n_lasers, cap;
  int *killer; /* per ship: list index of the killing laser this tick or -1 */
} orc_arena;

/* ---------------------------------------------------------------- helpers */
static double py_fmod(double a, double b) {
  /* CPython float_rem: fmod, then the result takes the sign of the divisor */
  double m = fmod(a, b);
  if (m != 0.0) {
    if ((b < 0) != (m < 0)) m += b;
  } else {
    m = copysign(0.0, b);
  }
  return m;
}

/* Forme.distance with both operands Python ints           form.py:42-44 */
static double dist_ii(long ax, long ay, long bx, long by) {
  long dx = ax - bx, dy = ay - by;
  return sqrt((double)(dx * dx + dy * dy));
}

/* Forme.distance with a float operand: `**` is libm pow   form.py:42-44 */
static double dist_fi(double ax, double ay, long bx, long by) {
  return sqrt(pow(ax - (double)bx, 2.0) + pow(ay - (double)by, 2.0));
}

/* ------------------------------------------------------------- lifetime */
orc_arena *orc_create(const orc_cfg *cfg) {
  orc_arena *a = (orc_arena *)calloc(1, sizeof(orc_arena));
  a->cfg = *cfg;
  a->ships = (orc_ship *)calloc((size_t)cfg->n_ships, sizeof(orc_ship));
  a->killer = (int *)calloc((size_t)cfg->n_ships, sizeof(int));
  a->cap = 256;
  a->lasers = (orc_laser *)calloc((size_t)a->cap, sizeof(orc_laser));
  return a;
}

void orc_destroy(orc_arena *a) {
  if (!a) return;
  free(a->ships);
  free(a->lasers);
  free(a->killer);
  free(a);
}

/* Battleground.__init__ -> Ship.__init__      battleground.py:79-81, ship.py:35-58 */
void orc_spawn(orc_arena *a, const int32_t *draws) {
  a->time = 0;
  a->n_lasers = 0;
  for (int i = 0; i < a->cfg.n_ships; i++) {
    orc_ship *s = &a->ships[i];
    memset(s, 0, sizeof(*s));
    s->x = draws[2 * i];
    s->y = draws[2 * i + 1];
    s->px = s->x;
    s->py = s->y;
    s->hull = 1;
    s->playable = 1;
    a->killer[i] = -1;
  }
}

/* crafted starts used by the fixtures (positions / pointing set by hand)   */
void orc_set_ship(orc_arena *a, int i, long x, long y, long px, long py) {
  a->ships[i].x = x;
  a->ships[i].y = y;
  a->ships[i].px = px;
  a->ships[i].py = py;
}

/* Battleground.restart -> Ship.reset -> Agent.reset
 * battleground.py:108-117, ship.py:92-106, agent.py:59-64                   */
void orc_restart(orc_arena *a, const int32_t *draws) {
  a->time = 0;
  a->n_lasers = 0;
  for (int i = 0; i < a->cfg.n_ships; i++) {
    orc_ship *s = &a->ships[i];
    s->last_score = s->score; /* scores.append(score) */
    s->score = 0;             /* reward is NOT cleared */
    s->time = 0;
    s->px = s->x; /* pointing = Point(old x, old y), BEFORE the move */
    s->py = s->y;
    long dx = draws[2 * i], dy = draws[2 * i + 1];
    s->x = dx ? dx : s->x; /* `x or self.body.x` */
    s->y = dy ? dy : s->y;
    s->playable = 1; /* hull is not restored */
    a->killer[i] = -1;
  }
}

/* ---------------------------------------------------------------- physics */
/* Ship.thrust                                               ship.py:213-222 */
void orc_thrust(const orc_cfg *c, long *x, long *y, long px, long py) {
  long dX = px - *x, dY = py - *y;
  double dist = sqrt((double)(dX * dX + dY * dY));
  if (dist != 0) {
    double dx = (double)(dX * c->ship_speed) / dist;
    double dy = (double)(dY * c->ship_speed) / dist;
    long nx = (long)((double)*x + dx); /* int(): truncation toward zero */
    long ny = (long)((double)*y + dy);
    if (nx < 0) nx = 0;
    if (nx > c->width - 1) nx = c->width - 1;
    if (ny < 0) ny = 0;
    if (ny > c->height - 1) ny = c->height - 1;
    *x = nx;
    *y = ny;
  }
}

/* Circle.edge(xn, yn, distance)                             form.py:159-188 */
int orc_edge(long x, long y, long radius, long xn, long yn, long distance, long *ex, long *ey) {
  long inR = radius + distance;
  long dX = xn - x, dY = yn - y;
  double dist = sqrt((double)(dX * dX + dY * dY));
  if (dist == 0) return 0;
  double dx = (double)(dX * inR) / dist;
  double dy = (double)(dY * inR) / dist;
  *ex = (long)((double)x + dx);
  *ey = (long)((double)y + dy);
  return 1;
}

/* Ship.enemy_aimed                                          ship.py:166-169 */
int orc_enemy_aimed(const orc_cfg *c, long px, long py, long ex, long ey) {
  return dist_ii(ex, ey, px, py) <= (double)c->ship_radius;
}

/* Ship.enemy_on_trajectory  ship.py:179-210 ; angle_with form.py:75-83 ;
 * angular_radius form.py:298-307 ; sum_angles form.py:24-31                 */
int orc_enemy_on_trajectory(const orc_cfg *c, long sx, long sy, long px, long py, long ex, long ey) {
  double shooting_angle = atan2((double)(py - sy), (double)(px - sx)) + M_PI;
  if (shooting_angle == 0.0) return 0;
  double target_angle = atan2((double)(ey - sy), (double)(ex - sx)) + M_PI;
  if (target_angle == 0.0) return 0;
  double dist = dist_ii(sx, sy, ex, ey);
  double angular_radius = (dist == 0) ? 2 * M_PI : atan((double)c->ship_radius / dist);
  double sup = py_fmod(target_angle + angular_radius, 2 * M_PI);
  double inf = py_fmod(target_angle + -angular_radius, 2 * M_PI);
  return inf <= shooting_angle && shooting_angle <= sup;
}

static void push_laser(orc_arena *a, const orc_laser *l) {
  if (a->n_lasers == a->cap) {
    a->cap *= 2;
    a->lasers = (orc_laser *)realloc(a->lasers, (size_t)a->cap * sizeof(orc_laser));
  }
  a->lasers[a->n_lasers++] = *l;
}

/* Laser.move                                                laser.py:36-62 */
static void laser_move(orc_arena *a, int j) {
  const orc_cfg *c = &a->cfg;
  orc_laser *l = &a->lasers[j];
  l->time += 1;
  long dX = l->tx - l->fx, dY = l->ty - l->fy;
  double dist = sqrt((double)(dX * dX + dY * dY));
  if (dist != 0) {
    double dx = (double)(dX * c->laser_speed) / dist;
    double dy = (double)(dY * c->laser_speed) / dist;
    l->x += dx;
    l->y += dy;
  }
  int explode = 0;
  for (int s = 0; s < c->n_ships; s++) {
    orc_ship *sh = &a->ships[s];
    /* Circle.collide: distance <= r1 + r2 (inclusive)      form.py:148-149 */
    if (sh->playable && dist_fi(l->x, l->y, sh->x, sh->y) <= (double)(c->laser_radius + c->ship_radius)) {
      a->ships[l->owner].reward += c->reward_kill;
      sh->hull -= 1;                   /* Ship.hit       ship.py:127-131 */
      if (sh->hull <= 0) {
        sh->reward += c->reward_death; /* Ship.explode   ship.py:225-230 */
        sh->playable = 0;
        a->killer[s] = j;
      }
      explode = 1; /* no break, no owner exclusion */
    }
  }
  /* Battleground.outside                            battleground.py:125-126 */
  if (explode || l->x < 0 || l->y < 0 || l->x >= c->width || l->y >= c->height) l->destroyed = 1;
}

/* Ship.shoot                                                ship.py:134-156 */
static void ship_shoot(orc_arena *a, int i) {
  const orc_cfg *c = &a->cfg;
  orc_ship *s = &a->ships[i];
  long ex, ey;
  if (!orc_edge(s->x, s->y, c->ship_radius, s->px, s->py, c->laser_radius, &ex, &ey)) return;
  orc_laser l;
  memset(&l, 0, sizeof(l));
  l.x = (double)ex;
  l.y = (double)ey;
  l.fx = ex;
  l.fy = ey;
  l.tx = s->px;
  l.ty = s->py;
  l.owner = i;
  /* pointing inside the own hit-box: fired = ship centre  ship.py:147-148 */
  if (dist_ii(s->x, s->y, s->px, s->py) <= (double)(c->ship_radius + c->laser_radius)) {
    l.fx = s->x;
    l.fy = s->y;
  }
  push_laser(a, &l);
  int aimed = 0, traj = 0;
  for (int k = 0; k < c->n_ships; k++) {
    if (k == i || !a->ships[k].playable) continue;
    aimed = aimed || orc_enemy_aimed(c, s->px, s->py, a->ships[k].x, a->ships[k].y);
    traj = traj || orc_enemy_on_trajectory(c, s->x, s->y, s->px, s->py, a->ships[k].x, a->ships[k].y);
  }
  if (aimed) s->reward += c->reward_aim;
  if (traj) s->reward += c->reward_trajectory;
}

/* One GUI tick of an arena:
 *   clear_wreckage of lasers destroyed last tick    ofighters.py:619-625,702-707
 *   request_actions: Agent.step bookkeeping for every ship, dead included
 *                                      battleground.py:146-150, agent.py:66-74
 *   generate_frame                                  battleground.py:153-160
 * actions: [M][5] = valid, shoot, thrust, px, py (valid==0 <=> None)        */
void orc_step(orc_arena *a, const int32_t *actions) {
  const orc_cfg *c = &a->cfg;
  int n = 0;
  for (int j = 0; j < a->n_lasers; j++)
    if (!a->lasers[j].destroyed) a->lasers[n++] = a->lasers[j];
  a->n_lasers = n;
  for (int i = 0; i < c->n_ships; i++) {
    orc_ship *s = &a->ships[i];
    s->obs_reward = s->reward;
    s->score += s->reward;
    s->reward = 0;
    a->killer[i] = -1;
  }
  a->time += 1;
  int n0 = a->n_lasers;
  for (int j = 0; j < n0; j++) laser_move(a, j);
  for (int i = 0; i < c->n_ships; i++) {
    orc_ship *s = &a->ships[i];
    const int32_t *act = actions + 5 * i;
    s->time += 1;
    if (!act[0] || !s->playable) continue; /* ship.py:308 */
    s->px = act[3];
    s->py = act[4];
    if (act[2]) orc_thrust(c, &s->x, &s->y, s->px, s->py);
    if (act[1]) ship_shoot(a, i);
  }
}

/* ---------------------------------------------------------------- getters */
int orc_n_lasers(const orc_arena *a) { return a->n_lasers; }

void orc_get_ships(const orc_arena *a, int32_t *xy, int32_t *pt, uint8_t *alive, int64_t *reward,
                   int64_t *score, int32_t *killer, int64_t *last_score, int64_t *hull) {
  for (int i = 0; i < a->cfg.n_ships; i++) {
    const orc_ship *s = &a->ships[i];
    if (xy) { xy[2 * i] = (int32_t)s->x; xy[2 * i + 1] = (int32_t)s->y; }
    if (pt) { pt[2 * i] = (int32_t)s->px; pt[2 * i + 1] = (int32_t)s->py; }
    if (alive) alive[i] = (uint8_t)s->playable;
    if (reward) reward[i] = s->reward;
    if (score) score[i] = s->score;
    if (killer) killer[i] = a->killer[i];
    if (last_score) last_score[i] = s->last_score;
    if (hull) hull[i] = s->hull;
  }
}

void orc_get_lasers(const orc_arena *a, double *x, double *y, int32_t *owner, uint8_t *destroyed) {
  for (int j = 0; j < a->n_lasers; j++) {
    if (x) x[j] = a->lasers[j].x;
    if (y) y[j] = a->lasers[j].y;
    if (owner) owner[j] = a->lasers[j].owner;
    if (destroyed) destroyed[j] = (uint8_t)a->lasers[j].destroyed;
  }
}

/* Observation.analyse_ship + head of toVector     observation.py:101-123 */
void orc_obs_head(const orc_arena *a, double *head, uint8_t *done) {
  for (int i = 0; i < a->cfg.n_ships; i++) {
    const orc_ship *s = &a->ships[i];
    double *v = head + 8 * i;
    v[0] = (double)s->reward;
    v[1] = 1.0; /* can_shoot is constant 1, ship.py:58 */
    v[2] = (double)s->px;
    v[3] = (double)s->py;
    v[4] = (double)a->cfg.width;
    v[5] = (double)a->cfg.height;
    v[6] = (double)s->x;
    v[7] = (double)s->y;
    if (done) done[i] = (uint8_t)!s->playable;
  }
}

/* ------------------------------------------------------------- rasteriser */
/* skimage.draw.disk((r, c), radius, shape=(rows, cols)) restated from
 * scikit-image 0.18.3 (unpinned dependency of the reference: requirements.txt:5)
 * draw.py:11-43 (_ellipse_in_shape), :46-143 (ellipse, rotation = 0), :183-223
 * (disk).  With rotation 0: sin=0, cos=1, so
 *   distances = ((r*1 + c*0)/rad)**2 + ((r*0 - c*1)/rad)**2 , kept iff < 1.
 * numpy evaluates `**2` on arrays as x*x (np.square), NOT pow.               */
void orc_disk(uint8_t *map, int rows, int cols, double r, double c, double radius) {
  double ul_r = ceil(r - radius), ul_c = ceil(c - radius);
  double lr_r = floor(r + radius), lr_c = floor(c + radius);
  long ulr = (long)ul_r, ulc = (long)ul_c, lrr = (long)lr_r, lrc = (long)lr_c;
  if (ulr < 0) ulr = 0;
  if (ulc < 0) ulc = 0;
  if (lrr > rows - 1) lrr = rows - 1;
  if (lrc > cols - 1) lrc = cols - 1;
  double sc_r = r - (double)ulr, sc_c = c - (double)ulc; /* shifted_center */
  long nr = lrr - ulr + 1, nc = lrc - ulc + 1;           /* bounding_shape */
  for (long i = 0; i < nr; i++) {
    for (long j = 0; j < nc; j++) {
      double rr = (double)i - sc_r, cc = (double)j - sc_c;
      double t1 = (rr * 1.0 + cc * 0.0) / radius;
      double t2 = (rr * 0.0 - cc * 1.0) / radius;
      double d = t1 * t1 + t2 * t2;
      if (d < 1) map[(ulr + i) * cols + (ulc + j)] = 1;
    }
  }
}

/* Observation.analyse_battleground                 observation.py:79-95
 * maps are np.zeros((dim.x, dim.y)) indexed [row = y][col = x] (form.py:226) */
void orc_rasterise(const orc_arena *a, uint8_t *ship_map, uint8_t *laser_map) {
  const orc_cfg *c = &a->cfg;
  int rows = c->width, cols = c->height;
  memset(ship_map, 0, (size_t)rows * cols);
  memset(laser_map, 0, (size_t)rows * cols);
  for (int i = 0; i < c->n_ships; i++)
    if (a->ships[i].playable)
      orc_disk(ship_map, rows, cols, (double)a->ships[i].y, (double)a->ships[i].x, (double)c->ship_radius);
  for (int j = 0; j < a->n_lasers; j++) /* destroyed lasers are still listed */
    orc_disk(laser_map, rows, cols, a->lasers[j].y, a->lasers[j].x, (double)c->laser_radius);
}

/* ------------------------------------------------------------- scratch NN */
/* Neural_network.feed: obs = sigmoid(np.dot(W, obs) + b) per layer
 * neural_network.py:396-420, sigmoid :49-53 = 1.0 / (1.0 + exp(-v)).
 * np.dot of a (m,n) f64 matrix with an (n,1) vector goes through BLAS gemv
 * whose summation order is implementation-defined; plain left-to-right
 * accumulation here, compared with a 1e-12 relative tolerance in the tests.  */
void orc_nn_feed(const int32_t *layers, int n_layers, const double *weights, const double *biases,
                 const double *x, double *y) {
  int maxw = 0;
  for (int i = 0; i < n_layers; i++)
    if (layers[i] > maxw) maxw = layers[i];
  double *cur = (double *)malloc(sizeof(double) * (size_t)maxw);
  double *nxt = (double *)malloc(sizeof(double) * (size_t)maxw);
  memcpy(cur, x, sizeof(double) * (size_t)layers[0]);
  const double *W = weights, *B = biases;
  for (int l = 0; l + 1 < n_layers; l++) {
    int nin = layers[l], nout = layers[l + 1];
    for (int o = 0; o < nout; o++) {
      double acc = 0.0;
      for (int k = 0; k < nin; k++) acc += W[(size_t)o * nin + k] * cur[k];
      nxt[o] = 1.0 / (1.0 + exp(-(acc + B[o])));
    }
    W += (size_t)nin * nout;
    B += nout;
    double *t = cur; cur = nxt; nxt = t;
  }
  memcpy(y, cur, sizeof(double) * (size_t)layers[n_layers - 1]);
  free(cur);
  free(nxt);
}

/* --------------------------------------------------- counter RNG + bot law */
/* Philox4x32-10 (Salmon et al., SC'11).  The build's synthetic action law:
 * same distributions as agents/agent.py:99-155, NOT the Mersenne-Twister
 * stream (recorded reference actions are replayed for parity instead).      */
static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                          uint32_t out[4]) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint64_t seed, uint32_t out[4]) {
  philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32), out);
}

static int32_t draw_int(uint32_t r, int32_t n_inclusive) { /* randint(0, n) */
  return (int32_t)(((uint64_t)r * (uint64_t)(n_inclusive + 1)) >> 32);
}

#define ORC_STREAM_BOT 0u
#define ORC_STREAM_RESET 1u

/* actions [M][5] for one arena; dead ships yield None (ship.py:260-262)     */
void orc_bot_actions(const orc_arena *a, const int32_t *behaviours, uint64_t seed, uint32_t global_arena,
                     uint32_t tick, int32_t *actions) {
  const orc_cfg *c = &a->cfg;
  for (int i = 0; i < c->n_ships; i++) {
    const orc_ship *s = &a->ships[i];
    int32_t *act = actions + 5 * i;
    uint32_t r[4];
    philox4x32_10(global_arena, (uint32_t)i, tick, ORC_STREAM_BOT, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    int shoot = 0, thrust = 0, repoint = 0;
    switch (behaviours[i]) {
      case 0: break;                                            /* idle   */
      case 1: {                                                 /* random */
        uint32_t k = (uint32_t)(((uint64_t)r[0] * 3u) >> 32);
        shoot = (k == 0); thrust = (k == 1); repoint = (k == 2);
      } break;
      case 2:                                                   /* turret */
        shoot = ((double)r[0] * (1.0 / 4294967296.0)) < 0.8;
        repoint = ((double)r[1] * (1.0 / 4294967296.0)) < 0.3;
        break;
      case 3:                                                   /* runner */
        thrust = ((double)r[0] * (1.0 / 4294967296.0)) < 0.9;
        repoint = ((double)r[1] * (1.0 / 4294967296.0)) < 0.1;
        break;
      case 4: thrust = 1; break;                                /* thrust */
      case 5: shoot = 1; break;                                 /* shoot  */
      default: break;
    }
    act[0] = s->playable ? 1 : 0;
    act[1] = shoot;
    act[2] = thrust;
    act[3] = repoint ? draw_int(r[2], c->width) : (int32_t)s->px;
    act[4] = repoint ? draw_int(r[3], c->height) : (int32_t)s->py;
    if (!s->playable) { act[1] = act[2] = 0; act[3] = (int32_t)s->px; act[4] = (int32_t)s->py; }
  }
}

#define ORC_STREAM_EXPLORE 2u
/* epsilon branch of Trainer.get_best_action + random_play (agents/qlearnIA_V2.py:199-204,317-321) on the
 * counter RNG: returns 1 and fills (iaction, px, py) when the ship explores this tick, else 0.               */
int orc_policy_explore(const orc_cfg *c, double eps, uint64_t seed, uint32_t global_arena, uint32_t ship,
                       uint32_t tick, int collecting, int32_t *out3) {
  uint32_t r[4];
  philox4x32_10(global_arena, ship, tick, ORC_STREAM_EXPLORE, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  double u = (double)r[0] * (1.0 / 4294967296.0);
  if (!(collecting || u <= eps)) return 0;
  out3[0] = draw_int(r[1], 1);
  out3[1] = draw_int(r[2], c->width - 1);
  out3[2] = draw_int(r[3], c->height - 1);
  return 1;
}

void orc_reset_draws(const orc_cfg *c, uint64_t seed, uint32_t global_arena, uint32_t episode, int32_t *draws) {
  for (int i = 0; i < c->n_ships; i++) {
    uint32_t r[4];
    philox4x32_10(global_arena, (uint32_t)i, episode, ORC_STREAM_RESET, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    draws[2 * i] = draw_int(r[0], c->width);
    draws[2 * i + 1] = draw_int(r[1], c->height);
  }
}

/* Bounded CPU-baseline loop for bench.py: n_arenas arenas x ticks of
 * random-bot actions + step (+ rasterise), single thread.  Returns a checksum
 * so the work cannot be optimised away.                                     */
uint64_t orc_run_random(const orc_cfg *cfg, int n_arenas, int ticks, uint64_t seed, int do_raster,
                        int episode_ticks) {
  uint64_t sum = 0;
  int M = cfg->n_ships;
  int32_t *beh = (int32_t *)malloc(sizeof(int32_t) * (size_t)M);
  int32_t *act = (int32_t *)malloc(sizeof(int32_t) * 5 * (size_t)M);
  int32_t *draws = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)M);
  size_t cells = (size_t)cfg->width * cfg->height;
  uint8_t *m0 = (uint8_t *)malloc(cells), *m1 = (uint8_t *)malloc(cells);
  for (int i = 0; i < M; i++) beh[i] = 1;
  for (int g = 0; g < n_arenas; g++) {
    orc_arena *a = orc_create(cfg);
    orc_reset_draws(cfg, seed, (uint32_t)g, 0, draws);
    orc_spawn(a, draws);
    uint32_t episode = 0;
    for (int t = 0; t < ticks; t++) {
      if (episode_ticks > 0 && t > 0 && t % episode_ticks == 0) {
        episode++;
        orc_reset_draws(cfg, seed, (uint32_t)g, episode, draws);
        orc_restart(a, draws);
      }
      orc_bot_actions(a, beh, seed, (uint32_t)g, (uint32_t)t, act);
      orc_step(a, act);
      if (do_raster) {
        orc_rasterise(a, m0, m1);
        sum += m0[cells / 2] + m1[cells / 3];
      }
      for (int i = 0; i < M; i++) sum += (uint64_t)(a->ships[i].x * 3 + a->ships[i].y + a->ships[i].reward);
      sum += (uint64_t)a->n_lasers;
    }
    orc_destroy(a);
  }
  free(beh); free(act); free(draws); free(m0); free(m1);
  return sum;
}

/* ---- bench.py's cpu_baseline legs as ONE C call (test infrastructure, like the rest of this file) -----------------
 * The same per-arena loop as orc_run_random, optionally with n_pol policy forwards per lock-step (no trunk sharing:
 * one full forward per ship, like the reference's model.predict per QlearnIA, agents/qlearnIA_V2.py:210), spread over
 * n_threads POSIX threads by arena (arena g goes to thread g % n_threads).  weights may be NULL when n_pol == 0.    */
#include <pthread.h>
void orc_policy_forward2(const uint8_t *ship_map, const uint8_t *laser_map, const float *vec8, const float *w,
                         float *act_values, float *heat, int32_t *iaction, int32_t *ipointer, int legacy);

typedef struct {
  const orc_cfg *cfg; const float *w;
  int first, stride, n_arenas, ticks, do_raster, episode_ticks, n_pol;
  uint64_t seed, sum;
} orc_bench_job;

static void *orc_bench_thread(void *arg) {
  orc_bench_job *j = (orc_bench_job *)arg;
  const orc_cfg *cfg = j->cfg;
  int M = cfg->n_ships;
  int32_t *beh = (int32_t *)malloc(sizeof(int32_t) * (size_t)M);
  int32_t *act = (int32_t *)malloc(sizeof(int32_t) * 5 * (size_t)M);
  int32_t *draws = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)M);
  double *head = (double *)malloc(sizeof(double) * 8 * (size_t)M);
  uint8_t *done = (uint8_t *)malloc((size_t)M);
  size_t cells = (size_t)cfg->width * cfg->height;
  uint8_t *m0 = (uint8_t *)malloc(cells), *m1 = (uint8_t *)malloc(cells);
  uint64_t sum = 0;
  for (int i = 0; i < M; i++) beh[i] = 1;
  for (int g = j->first; g < j->n_arenas; g += j->stride) {
    orc_arena *a = orc_create(cfg);
    orc_reset_draws(cfg, j->seed, (uint32_t)g, 0, draws);
    orc_spawn(a, draws);
    uint32_t episode = 0;
    for (int t = 0; t < j->ticks; t++) {
      if (j->episode_ticks > 0 && t > 0 && t % j->episode_ticks == 0) {
        episode++;
        orc_reset_draws(cfg, j->seed, (uint32_t)g, episode, draws);
        orc_restart(a, draws);
      }
      orc_bot_actions(a, beh, j->seed, (uint32_t)g, (uint32_t)t, act);
      if (j->n_pol > 0) {  /* the policy ships see the observation of the previous lock-step */
        orc_rasterise(a, m0, m1);
        orc_obs_head(a, head, done);
        for (int i = 0; i < j->n_pol && i < M; i++) {
          float v8[8], av[2];
          int32_t ia, ip[2];
          for (int k = 0; k < 8; k++) v8[k] = (float)head[8 * i + k];
          orc_policy_forward2(m0, m1, v8, j->w, av, NULL, &ia, ip, 0);
          act[5 * i + 1] = ia == 0; act[5 * i + 2] = ia == 1; act[5 * i + 3] = ip[0]; act[5 * i + 4] = ip[1];
        }
      }
      orc_step(a, act);
      if (j->do_raster) {
        orc_rasterise(a, m0, m1);
        sum += m0[cells / 2] + m1[cells / 3];
      }
      for (int i = 0; i < M; i++) sum += (uint64_t)(a->ships[i].x * 3 + a->ships[i].y + a->ships[i].reward);
      sum += (uint64_t)a->n_lasers;
    }
    orc_destroy(a);
  }
  free(beh); free(act); free(draws); free(head); free(done); free(m0); free(m1);
  j->sum = sum;
  return NULL;
}

uint64_t orc_bench_run(const orc_cfg *cfg, const float *weights, int n_arenas, int ticks, uint64_t seed, int do_raster,
                       int episode_ticks, int n_pol, int n_threads) {
  if (n_threads < 1) n_threads = 1;
  orc_bench_job *jobs = (orc_bench_job *)calloc((size_t)n_threads, sizeof(orc_bench_job));
  pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  uint64_t sum = 0;
  for (int k = 0; k < n_threads; k++) {
    orc_bench_job j = {cfg, weights, k, n_threads, n_arenas, ticks, do_raster, episode_ticks, n_pol, seed, 0};
    jobs[k] = j;
  }
  if (n_threads == 1) orc_bench_thread(&jobs[0]);
  else {
    for (int k = 0; k < n_threads; k++) pthread_create(&th[k], NULL, orc_bench_thread, &jobs[k]);
    for (int k = 0; k < n_threads; k++) pthread_join(th[k], NULL);
  }
  for (int k = 0; k < n_threads; k++) sum += jobs[k].sum;
  free(jobs); free(th);
  return sum;
}
