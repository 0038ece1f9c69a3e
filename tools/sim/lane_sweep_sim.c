// CPU simulation of the lane-per-line sweep (design study for csrc/edt_sweep_kernels.hip):
// builds the 1024^3 D1/D2 grids, runs Z scan + Y pass exactly, and counts, for waves of 64 adjacent Z lines
// walking the pass axis in lockstep, what a wave would execute: pop tests, pops, pushes, stack depths.
// gcc -O3 -fopenmp -o lane_sweep_sim lane_sweep_sim.c -lm
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define INF32 0x7fffffff
static uint64_t sm_state;
static uint64_t sm_next(void)
{
  sm_state += 0x9E3779B97F4A7C15ull;
  uint64_t z = sm_state;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static double sm_uniform(void) { return (double)(sm_next() >> 11) * (1.0 / 9007199254740992.0); }

typedef struct { int32_t f; int16_t v; int16_t s; } Entry;

// forward sweep of one line; per row: tests, pops, push flag, depth after
static int sweep_line(const int32_t* F, int n, Entry* st, uint8_t* tests, uint8_t* pops, uint8_t* push, int16_t* depth_after)
{
  int depth = 0;
  for (int q = 0; q < n; q++)
  {
    tests[q] = pops[q] = push[q] = 0;
    const int32_t f = F[q] < 0 ? -F[q] : F[q];
    if (f != INF32)
    {
      const int64_t Gq = (int64_t)f + (int64_t)q * q;
      int t = 0, p = 0;
      while (depth > 0)
      {
        const Entry* e = &st[depth - 1];
        t++;
        const int64_t Gt = (int64_t)e->f + (int64_t)e->v * e->v;
        if (Gq - Gt <= 2ll * e->s * (q - e->v)) { depth--; p++; } else break;
      }
      int s = 0;
      if (depth > 0)
      {
        const Entry* e = &st[depth - 1];
        const int64_t Gt = (int64_t)e->f + (int64_t)e->v * e->v;
        const int64_t num = Gq - Gt, den = 2ll * (q - e->v);
        // first row where new <= top: ceil(num/den), num > 0 here
        s = (int)((num + den - 1) / den);
      }
      if (s < n) { st[depth].f = f; st[depth].v = (int16_t)q; st[depth].s = (int16_t)s; depth++; push[q] = 1; }
      tests[q] = t > 255 ? 255 : t; pops[q] = p > 255 ? 255 : p;
    }
    depth_after[q] = (int16_t)depth;
  }
  return depth;
}

int main(int argc, char** argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 1024;
  const int salt = argc > 2 ? atoi(argv[2]) : 0;
  const int64_t N = (int64_t)n * n * n;
  uint8_t* mask = calloc(N, 1);
  if (!salt)
  {
    sm_state = 42;
    for (int i = 0; i < 64; i++)
    {
      int cx = (int)(sm_uniform() * n), cy = (int)(sm_uniform() * n), cz = (int)(sm_uniform() * n);
      double rmax = n / 16.0 > 2.0 ? n / 16.0 : 2.0;
      double r = 2.0 + sm_uniform() * (rmax - 2.0), rr = r * r;
      int ri = (int)ceil(r);
      for (int x = cx - ri > 0 ? cx - ri : 0; x < (cx + ri + 1 < n ? cx + ri + 1 : n); x++)
        for (int y = cy - ri > 0 ? cy - ri : 0; y < (cy + ri + 1 < n ? cy + ri + 1 : n); y++)
          for (int z = cz - ri > 0 ? cz - ri : 0; z < (cz + ri + 1 < n ? cz + ri + 1 : n); z++)
          {
            double d2 = (double)(x - cx) * (x - cx) + (double)(y - cy) * (y - cy) + (double)(z - cz) * (z - cz);
            if (d2 <= rr) mask[((int64_t)x * n + y) * n + z] = 1;
          }
    }
  }
  else
  {
#pragma omp parallel for
    for (int64_t i = 0; i < N; i++)
    {
      uint64_t z = 42ull + (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      mask[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0)) < 0.01;
    }
  }
  int64_t filled = 0;
  for (int64_t i = 0; i < N; i++) filled += mask[i];
  printf("n=%d salt=%d fill=%.4f\n", n, salt, (double)filled / N);
  // Z scan: signed squared distance to the other class along z
  int32_t* A = malloc(N * 4);
#pragma omp parallel for
  for (int64_t line = 0; line < (int64_t)n * n; line++)
  {
    const uint8_t* m = mask + line * n;
    int32_t* a = A + line * n;
    int last[2] = {-1, -1};  // last row of class c
    for (int z = 0; z < n; z++) { last[m[z]] = z; int o = last[!m[z]]; a[z] = o < 0 ? INF32 : (z - o) * (z - o); }
    last[0] = last[1] = -1;
    for (int z = n - 1; z >= 0; z--)
    {
      last[m[z]] = z; int o = last[!m[z]];
      if (o >= 0 && (o - z) * (o - z) < a[z]) a[z] = (o - z) * (o - z);
      if (m[z]) a[z] = -a[z];
    }
  }
  for (int pass = 0; pass < 2; pass++)
  {
    // pass 0: Y (lines along y at fixed x, z; stride n), pass 1: X (stride n*n)
    const int64_t rs = pass == 0 ? n : (int64_t)n * n;
    const int64_t os = pass == 0 ? (int64_t)n * n : n;
    int32_t* B = malloc(N * 4);
    double sum_tests = 0, sum_pops = 0, sum_push = 0, sum_maxtests = 0, sum_maxpops = 0, sum_anypush = 0, sum_evalmax = 0, sum_evalpops = 0;
    double sum_depth = 0, sum_finite = 0, sum_anytrans = 0, sum_trans = 0;
    int64_t maxdepth = 0, wave_rows = 0, refill32 = 0, refill16 = 0, deep_rows32 = 0;
    double sum_resid_spread = 0;
    double sum_look[2] = {0, 0};
#pragma omp parallel
    {
      Entry* st = malloc(sizeof(Entry) * n * 64);
      uint8_t* tests = malloc(n * 64), *pops = malloc(n * 64), *push = malloc(n * 64), *evp = malloc(n * 64);
      int16_t* dep = malloc(2 * n * 64);
      int32_t* F = malloc(4 * n);
      double l_tests = 0, l_pops = 0, l_push = 0, l_maxtests = 0, l_maxpops = 0, l_anypush = 0, l_evalmax = 0, l_evalpops = 0, l_depth = 0, l_finite = 0, l_anytrans = 0, l_trans = 0;
      int64_t l_maxdepth = 0, l_wave_rows = 0, l_deep32 = 0;
      double l_look[2] = {0, 0};
#pragma omp for schedule(dynamic, 4)
      for (int64_t wave = 0; wave < (int64_t)n * n / 64; wave++)
      {
        const int outer = (int)(wave / (n / 64));
        const int z0 = (int)(wave % (n / 64)) * 64;
        uint8_t trans[64 * 0 + 1]; (void)trans;
        static __thread uint8_t tr[2048 * 64];
        for (int lane = 0; lane < 64; lane++)
        {
          const int64_t base = (int64_t)outer * os + z0 + lane;
          for (int q = 0; q < n; q++) F[q] = A[base + q * rs];
          Entry* s = st + lane * n;
          int depth = sweep_line(F, n, s, tests + lane * n, pops + lane * n, push + lane * n, dep + lane * n);
          // eval (forward here, same counts): owner k; bounding rows
          int k = 0;
          int prev_opp = -1;
          memset(evp + lane * n, 0, n);
          for (int q = 0; q < n; q++) tr[lane * n + q] = (q > 0) && ((F[q] < 0) != (F[q - 1] < 0));
          for (int q = 0; q < n; q++)
          {
            int adv = 0;
            while (k + 1 < depth && s[k + 1].s <= q) { k++; adv++; }
            evp[lane * n + q] = adv > 255 ? 255 : adv;
            int64_t best = INF32;
            if (depth > 0) best = (int64_t)(q - s[k].v) * (q - s[k].v) + s[k].f;
            if (q > 0 && (F[q] < 0) != (F[q - 1] < 0)) prev_opp = q - 1;
            if (prev_opp >= 0 && (int64_t)(q - prev_opp) * (q - prev_opp) < best) best = (int64_t)(q - prev_opp) * (q - prev_opp);
            B[base + q * rs] = best >= INF32 ? INF32 : (int32_t)best;
          }
          int next_opp = -1;
          for (int q = n - 1; q >= 0; q--)
          {
            if (q < n - 1 && (F[q] < 0) != (F[q + 1] < 0)) next_opp = q + 1;
            int32_t b = B[base + q * rs];
            if (next_opp >= 0 && (int64_t)(next_opp - q) * (next_opp - q) < b) b = (next_opp - q) * (next_opp - q);
            B[base + q * rs] = F[q] < 0 ? -b : b;
          }
          l_depth += depth;
        }
        if ((wave & 7) != 0) continue;  // statistics on every 8th wave
        // evaluation with look-ahead: a lane evaluates the top K entries of its stack, so it has to pop only when K - 1
        // advances are pending; when any lane has to, every lane with a pending advance pops along
        for (int K = 2; K <= 3; K++)
        {
          int pending[64] = {0};
          for (int q = 0; q < n; q++)
          {
            int must = 0;
            for (int lane = 0; lane < 64; lane++) { pending[lane] += evp[lane * n + q]; if (pending[lane] >= K) must = 1; }
            while (must)
            {
              l_look[K - 2] += 1;
              must = 0;
              for (int lane = 0; lane < 64; lane++) { if (pending[lane] > 0) pending[lane]--; if (pending[lane] >= K) must = 1; }
            }
          }
        }
        for (int q = 0; q < n; q++)
        {
          int mt = 0, mp = 0, ap = 0, me = 0, at = 0;
          for (int lane = 0; lane < 64; lane++)
          {
            const int i = lane * n + q;
            l_tests += tests[i]; l_pops += pops[i]; l_push += push[i]; l_evalpops += evp[i];
            l_finite += (tests[i] || push[i]);
            if (tests[i] > mt) mt = tests[i];
            if (pops[i] > mp) mp = pops[i];
            if (evp[i] > me) me = evp[i];
            ap |= push[i]; at |= tr[i]; l_trans += tr[i];
            if (dep[i] > l_maxdepth) l_maxdepth = dep[i];
            if (pops[i] > 24) l_deep32++;
          }
          l_maxtests += mt; l_maxpops += mp; l_anypush += ap; l_evalmax += me; l_anytrans += at;
          l_wave_rows++;
        }
      }
#pragma omp critical
      {
        sum_tests += l_tests; sum_pops += l_pops; sum_push += l_push; sum_maxtests += l_maxtests; sum_maxpops += l_maxpops;
        sum_anypush += l_anypush; sum_evalmax += l_evalmax; sum_evalpops += l_evalpops; sum_depth += l_depth; sum_finite += l_finite;
        sum_anytrans += l_anytrans; sum_trans += l_trans; deep_rows32 += l_deep32;
        if (l_maxdepth > maxdepth) maxdepth = l_maxdepth;
        sum_look[0] += l_look[0]; sum_look[1] += l_look[1];
        wave_rows += l_wave_rows;
      }
    }
    const double lanes = (double)wave_rows * 64;
    printf("pass %s: per lane-row: finite %.3f tests %.3f pops %.3f push %.3f evaladv %.3f trans %.4f | per wave-row: max tests %.3f max pops %.3f any push %.3f eval max adv %.3f any trans %.3f | final depth avg %.1f max depth %lld | pops>24: %.2e per lane-row\n",
           pass == 0 ? "Y" : "X", sum_finite / lanes, sum_tests / lanes, sum_pops / lanes, sum_push / lanes, sum_evalpops / lanes, sum_trans / lanes,
           sum_maxtests / wave_rows, sum_maxpops / wave_rows, sum_anypush / wave_rows, sum_evalmax / wave_rows, sum_anytrans / wave_rows,
           sum_depth / ((double)n * n), (long long)maxdepth, deep_rows32 / lanes);
    printf("   evaluation pop iterations per wave-row: one entry evaluated %.3f, two %.3f, three %.3f\n",
           sum_evalmax / wave_rows, sum_look[0] / wave_rows, sum_look[1] / wave_rows);
    (void)refill32; (void)refill16; (void)sum_resid_spread;
    free(A);
    A = B;
  }
  // checksum of the final field for cross-checking
  uint64_t h = 0;
  for (int64_t i = 0; i < N; i++) h = h * 1099511628211ull + (uint32_t)A[i];
  printf("checksum %016llx\n", (unsigned long long)h);
  return 0;
}
