/* CPU model of the raycast kernel's per-workgroup visit table (csrc/voxelizer_kernels.hip), to count -- for a table
 * organisation, a segment length and a workgroup size -- what the kernel sends to global memory for config 3's cloud A:
 *   atomics            global atomic adds (table flushes + visits that found no place in the table)
 *   line transactions  sum over wave instructions of the distinct 128-byte lines their lanes touch
 * The scattered-atomics microbench (profiles/r4: random cells 27.1 G/s, one 256-byte line per wave instruction 421 G/s)
 * prices them at about 35.7 ps per line transaction + 1.25 ps per atomic, chip-wide.
 *
 * The walk is the kernel's float DDA on a synthetic cloud of the same law as synthetic.raycast_cloud (uniform
 * directions, range uniform in [0.5, 4.0] m, sensor in the middle of a 256^3 grid of 0.02 m voxels, max range 3 m);
 * rays are ordered as the kernel orders them (cube-map face, 64 x 64 bins per face, Morton order inside a face).
 * Races between lanes are not modelled (a slot is claimed at once).
 *
 *   gcc -O2 -o raycast_table_sim raycast_table_sim.c -lm && ./raycast_table_sim [points]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define N 256
static const float kVoxel = 5.12f / N;

typedef struct
{
  uint32_t cell;
  int32_t delta[3];
  uint32_t lim[3];
  uint32_t total;
  float t[3], dt[3];
  int walking;
  int32_t xyz[3], step[3];
} Ray;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double uniform(void)
{
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (double)(rng_state >> 11) / 9007199254740992.0;
}

static uint32_t spread_bits(uint32_t x)
{
  x &= 0xffffu;
  x = (x | (x << 8)) & 0x00ff00ffu;
  x = (x | (x << 4)) & 0x0f0f0f0fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
}

static uint32_t direction_bin(const float d[3])
{
  const float a[3] = {fabsf(d[0]), fabsf(d[1]), fabsf(d[2])};
  int axis = 0;
  if (a[1] > a[axis]) axis = 1;
  if (a[2] > a[axis]) axis = 2;
  const float u = d[(axis + 1) % 3] / a[axis], v = d[(axis + 2) % 3] / a[axis];
  int iu = (int)((u + 1.0f) * 32.0f), iv = (int)((v + 1.0f) * 32.0f);
  if (iu > 63) iu = 63;
  if (iv > 63) iv = 63;
  if (iu < 0) iu = 0;
  if (iv < 0) iv = 0;
  const uint32_t face = (uint32_t)(axis * 2 + (d[axis] < 0.0f));
  return (face << 12) | spread_bits((uint32_t)iu) | (spread_bits((uint32_t)iv) << 1);
}

static float axis_t(float point, float ray, float lo, float hi)
{
  if (ray > 0.0f) return fabsf((hi - point) / ray);
  if (ray < 0.0f) return fabsf((point - lo) / ray);
  return INFINITY;
}

static void init_ray(Ray* r, const float p[3], const float origin[3], float max_range)
{
  const float ray[3] = {p[0] - origin[0], p[1] - origin[1], p[2] - origin[2]};
  const float length = sqrtf(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
  float last[3] = {p[0], p[1], p[2]};
  if (length > max_range)
    for (int a = 0; a < 3; a++) last[a] = origin[a] + ray[a] * (max_range / length);
  const float inv = 1.0f / kVoxel;
  uint64_t remaining = 0;
  const int32_t stride[3] = {N * N, N, 1};
  int inside = 1;
  r->cell = 0;
  for (int a = 0; a < 3; a++)
  {
    const int32_t cur = (int32_t)floorf(origin[a] * inv), end = (int32_t)floorf(last[a] * inv);
    const int32_t diff = end - cur;
    const int32_t step = (diff > 0) - (diff < 0);
    const float centre = ((float)cur + 0.5f) * kVoxel, half = kVoxel * 0.5f;
    r->t[a] = axis_t(origin[a], ray[a], centre - half, centre + half);
    r->dt[a] = fabsf(kVoxel / ray[a]);
    const uint32_t apart = (uint32_t)abs(diff);
    remaining += apart;
    const uint32_t room = (uint32_t)(step > 0 ? N - 1 - cur : cur);
    r->lim[a] = apart < room ? apart : room;
    r->delta[a] = step * stride[a];
    r->cell += (uint32_t)(cur * stride[a]);
    r->xyz[a] = cur;
    r->step[a] = step;
    if (cur < 0 || cur >= N) inside = 0;
  }
  r->total = (uint32_t)remaining;
  r->walking = inside && remaining != 0;
}

/* one DDA step; returns 0 when the ray stops after this visit */
static int step_ray(Ray* r)
{
  const int ax = (r->t[0] <= r->t[1]) && (r->t[0] <= r->t[2]);
  const int ay = !ax && (r->t[1] <= r->t[0]) && (r->t[1] <= r->t[2]);
  const int a = ax ? 0 : (ay ? 1 : 2);
  const uint32_t lim = r->lim[a];
  r->lim[a] -= 1;
  r->t[a] += r->dt[a];
  r->cell += (uint32_t)r->delta[a];
  r->xyz[a] += r->step[a];
  r->total -= 1;
  return !(lim == 0 || r->total == 0);
}

/* ---- table organisations ---- */
enum
{
  kHash2Way,   /* multiplicative hash, two-way sets (the kernel as built) */
  kHashLinear, /* multiplicative hash, linear probing, 6 probes (round 4's table) */
  kTorus,      /* slot = (x mod 2^bx, y mod 2^by, z mod 2^bz), one way */
  kTorus2Way,  /* the same over half as many sets, two ways */
};

typedef struct
{
  int kind, slots, segment, group; /* group = rays per workgroup */
  int bx, by, bz;                  /* torus: log2 of the window's sides (bx + by + bz = log2(slots)) */
  const char* name;
  int interleave; /* 1: wave w of a workgroup takes rays w, w + waves, w + 2 waves, ... of its share */
} Config;

typedef struct
{
  uint64_t atomics, lines, spills, spill_lines, flushed, claims, depth, wave_steps, depth_by_step[16];
} Totals;

static uint32_t* keys;
static uint32_t* counts;
#define EMPTY 0xffffffffu

static int log2i(int v)
{
  int b = 0;
  while ((1 << b) < v) b++;
  return b;
}

static uint32_t torus_slot(const Config* c, const Ray* r, int ways)
{
  const int bz = c->bz - (ways == 2 ? 1 : 0);
  const uint32_t sx = (uint32_t)r->xyz[0] & ((1u << c->bx) - 1), sy = (uint32_t)r->xyz[1] & ((1u << c->by) - 1);
  /* two ways: the set covers a pair of z-neighbours */
  const uint32_t sz = ((uint32_t)r->xyz[2] >> (ways == 2 ? 1 : 0)) & ((1u << bz) - 1);
  return (((sx << c->by) | sy) << bz | sz) * (uint32_t)ways;
}

/* returns 1 when the visit went into the table, 0 when it spills to global memory */
static int table_add(const Config* c, const Ray* r, Totals* tot)
{
  const uint32_t cell = r->cell;
  const int bits = log2i(c->slots);
  if (c->kind == kHashLinear)
  {
    uint32_t slot = (cell * 2654435761u) >> (32 - bits);
    for (int probe = 0; probe < 6; probe++)
    {
      if (keys[slot] == EMPTY)
      {
        keys[slot] = cell;
        tot->claims++;
      }
      if (keys[slot] == cell)
      {
        counts[slot]++;
        return 1;
      }
      slot = (slot + 1) & (uint32_t)(c->slots - 1);
    }
    return 0;
  }
  uint32_t set;
  int ways = 2;
  if (c->kind == kHash2Way)
    set = ((cell * 2654435761u) >> (32 - bits + 1)) << 1;
  else if (c->kind == kTorus2Way)
    set = torus_slot(c, r, 2);
  else
  {
    set = torus_slot(c, r, 1);
    ways = 1;
  }
  for (int w = 0; w < ways; w++)
    if (keys[set + w] == cell)
    {
      counts[set + w]++;
      return 1;
    }
  for (int w = 0; w < ways; w++)
    if (keys[set + w] == EMPTY)
    {
      keys[set + w] = cell;
      counts[set + w] = 1;
      tot->claims++;
      return 1;
    }
  return 0;
}

static int cmp_u32(const void* a, const void* b)
{
  const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
  return (x > y) - (x < y);
}

static int distinct_lines(uint32_t* cells, int n)
{
  /* tracking is int32[cell][2]: 16 cells per 128-byte line */
  for (int i = 0; i < n; i++) cells[i] >>= 4;
  qsort(cells, (size_t)n, sizeof(uint32_t), cmp_u32);
  int d = 0;
  for (int i = 0; i < n; i++)
    if (i == 0 || cells[i] != cells[i - 1]) d++;
  return d;
}

static void flush(const Config* c, Totals* tot)
{
  uint32_t lane_cells[64];
  for (int base = 0; base < c->slots; base += 64)
  {
    int n = 0;
    for (int s = base; s < base + 64 && s < c->slots; s++)
      if (keys[s] != EMPTY)
      {
        lane_cells[n++] = keys[s];
        keys[s] = EMPTY;
        counts[s] = 0;
      }
    if (n)
    {
      tot->atomics += (uint64_t)n;
      tot->flushed += (uint64_t)n;
      tot->lines += (uint64_t)distinct_lines(lane_cells, n);
    }
  }
}

static void run(const Config* c, Ray* rays, const Ray* initial, int num_rays, uint64_t* visits_out)
{
  Totals tot;
  memset(&tot, 0, sizeof tot);
  memcpy(rays, initial, sizeof(Ray) * (size_t)num_rays);
  for (int s = 0; s < c->slots; s++) keys[s] = EMPTY, counts[s] = 0;
  uint64_t visits = 0;
  uint32_t spill_cells[64];
  for (int first = 0; first < num_rays; first += c->group)
  {
    const int last = first + c->group < num_rays ? first + c->group : num_rays;
    int walked = 0;
    for (;;)
    {
      int any = 0;
      for (int i = first; i < last; i++) any |= rays[i].walking;
      if (!any) break;
      for (int s = 0; s < c->segment; s++, walked++)
      {
        const int waves = (last - first + 63) / 64;
        for (int w = 0; w < waves; w++)
        {
          int n = 0, active = 0;
          uint32_t lane_cell[64];
          for (int l = 0; l < 64; l++)
          {
            const int i = c->interleave ? first + l * waves + w : first + w * 64 + l;
            if (i >= last) continue;
            Ray* r = &rays[i];
            if (!r->walking) continue;
            visits++;
            lane_cell[active++] = r->cell;
            if (!table_add(c, r, &tot)) spill_cells[n++] = r->cell;
            r->walking = step_ray(r);
          }
          if (active)
          {
            /* LDS: lanes on one address are served one after the other */
            qsort(lane_cell, (size_t)active, sizeof(uint32_t), cmp_u32);
            int deepest = 1, run = 1;
            for (int l = 1; l < active; l++)
            {
              run = lane_cell[l] == lane_cell[l - 1] ? run + 1 : 1;
              if (run > deepest) deepest = run;
            }
            tot.depth += (uint64_t)deepest;
            tot.wave_steps++;
            tot.depth_by_step[walked / 16 < 15 ? walked / 16 : 15] += (uint64_t)deepest;
          }
          if (n)
          {
            tot.atomics += (uint64_t)n;
            tot.spills += (uint64_t)n;
            const int d = distinct_lines(spill_cells, n);
            tot.lines += (uint64_t)d;
            tot.spill_lines += (uint64_t)d;
          }
        }
      }
      flush(c, &tot);
    }
  }
  *visits_out = visits;
  const double ms = (35.7e-12 * (double)tot.lines + 1.25e-12 * (double)tot.atomics) * 1e3;
  printf("%-34s group %4d seg %3d slots %5d | atomics %6.2f M (spills %6.2f M, flushed %6.2f M) lines %6.2f M -> %.3f ms\n",
         c->name, c->group, c->segment, c->slots, (double)tot.atomics / 1e6, (double)tot.spills / 1e6,
         (double)tot.flushed / 1e6, (double)tot.lines / 1e6, ms);
  printf("    same-address depth per wave step %.2f (%.1f M wave steps); by 16 steps:", (double)tot.depth / (double)tot.wave_steps,
         (double)tot.wave_steps / 1e6);
  for (int k = 0; k < 16; k++) printf(" %.1f", (double)tot.depth_by_step[k] / 1e6);
  printf(" M\n");
  fflush(stdout);
}

typedef struct
{
  uint32_t bin;
  int index;
} Order;
static int cmp_order(const void* a, const void* b)
{
  const Order* x = (const Order*)a;
  const Order* y = (const Order*)b;
  if (x->bin != y->bin) return (x->bin > y->bin) - (x->bin < y->bin);
  return (x->index > y->index) - (x->index < y->index);
}

int main(int argc, char** argv)
{
  const int num_points = argc > 1 ? atoi(argv[1]) : 1000000;
  const float origin[3] = {2.56f, 2.56f, 2.56f};
  Ray* initial = malloc(sizeof(Ray) * (size_t)num_points);
  Ray* rays = malloc(sizeof(Ray) * (size_t)num_points);
  Order* order = malloc(sizeof(Order) * (size_t)num_points);
  float(*pts)[3] = malloc(sizeof(float[3]) * (size_t)num_points);
  for (int i = 0; i < num_points; i++)
  {
    const double zc = 2.0 * uniform() - 1.0, phi = 2.0 * M_PI * uniform(), rg = 0.5 + 3.5 * uniform();
    const double s = sqrt(fmax(0.0, 1.0 - zc * zc));
    const float d[3] = {(float)(s * cos(phi) * rg), (float)(s * sin(phi) * rg), (float)(zc * rg)};
    for (int a = 0; a < 3; a++) pts[i][a] = d[a] + origin[a];
    order[i].bin = direction_bin(d);
    order[i].index = i;
  }
  qsort(order, (size_t)num_points, sizeof(Order), cmp_order);
  for (int i = 0; i < num_points; i++) init_ray(&initial[i], pts[order[i].index], origin, 3.0f);
  keys = malloc(sizeof(uint32_t) * 65536);
  counts = malloc(sizeof(uint32_t) * 65536);

  const Config configs[] = {
      {kHashLinear, 4096, 96, 256, 0, 0, 0, "hash, linear probing (round 4)", 0},
      {kTorus, 4096, 32, 256, 4, 4, 4, "torus 16x16x16", 0},
      {kTorus, 4096, 32, 256, 4, 4, 4, "torus 16x16x16, waves interleaved", 1},
      {kTorus, 4096, 48, 256, 4, 4, 4, "torus 16x16x16, waves interleaved", 1},
      {kTorus, 4096, 32, 512, 4, 4, 4, "torus 16x16x16, waves interleaved", 1},
      {kTorus, 8192, 32, 512, 4, 4, 5, "torus 16x16x32, waves interleaved", 1},
  };
  uint64_t visits = 0;
  for (size_t k = 0; k < sizeof configs / sizeof configs[0]; k++) run(&configs[k], rays, initial, num_points, &visits);
  printf("visits %.1f M\n", (double)visits / 1e6);
  return 0;
}
