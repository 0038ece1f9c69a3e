// CPU design study for the X pass of csrc/edt_sweep_kernels.hip (round 5): how many rows of a 16-row band still need the
// Felzenszwalb-Huttenlocher stack after a branch-free, all-lanes-active LOCAL filter, and how many iterations a wave
// of 64 lines needs when every lane takes ITS next surviving row per iteration (a band-local work list) instead of
// all lanes walking all 16 rows in lockstep.
//
// Filters (all exact: a point on or above a chord between two other points of its line is not a vertex of the lower
// hull of (row, G = F + row^2)):
//   s1      2 G(q) >= G(q-1) + G(q+1)                                  (the kernel's neighbour test, round 4)
//   s2, s4, s8   the same with rows q -+ 2, 4, 8 in addition
//   hull    exact lower hull of the band's rows plus one row on either side (the best any band-local filter can do)
// Builds the 1024^3 D1 / salt grid, runs Z scan + Y pass exactly, then the statistics over the X pass's input.
// gcc -O3 -fopenmp -o band_filter_sim band_filter_sim.c -lm
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define INF32 0x7fffffff
#define NOSITE 0x60000000
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

// exact 1-D squared EDT of one line by brute-force-free two-pass F-H (int64), classes ignored, then class candidates
typedef struct { int64_t G; int r; } Pt;
static void line_transform(const int32_t* F, int n, int32_t* out, Pt* st)
{
  int depth = 0;
  for (int q = 0; q < n; q++)
  {
    const int32_t f = F[q] < 0 ? -F[q] : F[q];
    if (f == INF32) continue;
    const int64_t G = (int64_t)f + (int64_t)q * q;
    while (depth >= 2)
    {
      const Pt a = st[depth - 2], b = st[depth - 1];
      // b on or above chord a..q ?
      if ((b.G - a.G) * (q - b.r) >= (G - b.G) * (b.r - a.r)) depth--; else break;
    }
    st[depth].G = G; st[depth].r = q; depth++;
  }
  int k = 0;
  for (int q = 0; q < n; q++)
  {
    int64_t best = INF32;
    if (depth > 0)
    {
      while (k + 1 < depth && st[k + 1].G - 2ll * q * st[k + 1].r <= st[k].G - 2ll * q * st[k].r) k++;
      best = st[k].G - 2ll * q * st[k].r + (int64_t)q * q;
    }
    out[q] = best >= INF32 ? INF32 : (int32_t)best;
  }
  // nearest row of the other class below / above
  int prev = -1;
  for (int q = 0; q < n; q++)
  {
    if (q > 0 && (F[q] < 0) != (F[q - 1] < 0)) prev = q - 1;
    if (prev >= 0 && (int64_t)(q - prev) * (q - prev) < out[q]) out[q] = (q - prev) * (q - prev);
  }
  int next = -1;
  for (int q = n - 1; q >= 0; q--)
  {
    if (q < n - 1 && (F[q] < 0) != (F[q + 1] < 0)) next = q + 1;
    if (next >= 0 && (int64_t)(next - q) * (next - q) < out[q]) out[q] = (next - q) * (next - q);
  }
  for (int q = 0; q < n; q++) if (F[q] < 0) out[q] = -out[q];
}

#define BAND 16
#define NF 10  // s1, s2, s4, s8, hull, hull32, coarse hull of every 8th / 16th / 32nd / 64th row + s1
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
  int32_t* A = malloc(N * 4);
#pragma omp parallel for
  for (int64_t line = 0; line < (int64_t)n * n; line++)
  {
    const uint8_t* m = mask + line * n;
    int32_t* a = A + line * n;
    int last[2] = {-1, -1};
    for (int z = 0; z < n; z++) { last[m[z]] = z; int o = last[!m[z]]; a[z] = o < 0 ? INF32 : (z - o) * (z - o); }
    last[0] = last[1] = -1;
    for (int z = n - 1; z >= 0; z--)
    {
      last[m[z]] = z; int o = last[!m[z]];
      if (o >= 0 && (o - z) * (o - z) < a[z]) a[z] = (o - z) * (o - z);
      if (m[z]) a[z] = -a[z];
    }
  }
  free(mask);
  // Y pass (exact), in place through a line buffer
#pragma omp parallel
  {
    int32_t* F = malloc(4 * n), *O = malloc(4 * n);
    Pt* st = malloc(sizeof(Pt) * n);
#pragma omp for schedule(dynamic, 64)
    for (int64_t l = 0; l < (int64_t)n * n; l++)
    {
      const int x = (int)(l / n), z = (int)(l % n);
      const int64_t base = (int64_t)x * n * n + z;
      for (int q = 0; q < n; q++) F[q] = A[base + (int64_t)q * n];
      line_transform(F, n, O, st);
      for (int q = 0; q < n; q++) A[base + (int64_t)q * n] = O[q];
    }
    free(F); free(O); free(st);
  }
  printf("n=%d salt=%d: X pass input ready\n", n, salt);
  // X pass statistics
  double surv[NF] = {0}, wave_iter[NF] = {0};
  double hist_iter[NF][2 * BAND + 3];
  memset(hist_iter, 0, sizeof(hist_iter));
  double wave_bands = 0, lane_bands = 0, sites = 0, wave_any_site_rows = 0;
  double full_stack_push = 0;  // rows that the real sweep pushes (after the s1 filter and the push test)
  double coarse_push[4] = {0}, coarse_depth[4] = {0}, anyrow[NF] = {0}, lines = 0;
#pragma omp parallel
  {
    double l_surv[NF] = {0}, l_iter[NF] = {0}, l_hist[NF][2 * BAND + 3];
    memset(l_hist, 0, sizeof(l_hist));
    double l_wb = 0, l_lb = 0, l_sites = 0, l_any = 0, l_push = 0;
    double l_coarse_push[4] = {0}, l_coarse_depth[4] = {0}, l_anyrow[NF] = {0}, l_lines = 0;
    int64_t* G = malloc(8 * (n + 2) * 64);
    uint8_t* keep = malloc((size_t)NF * n * 64);
    Pt* st = malloc(sizeof(Pt) * (n + 4));
#pragma omp for schedule(dynamic, 4)
    for (int64_t wave = 0; wave < (int64_t)n * n / 64; wave += 8)  // every 8th wave
    {
      const int y = (int)(wave / (n / 64));
      const int z0 = (int)(wave % (n / 64)) * 64;
      for (int lane = 0; lane < 64; lane++)
      {
        int64_t* g = G + (int64_t)lane * (n + 2) + 1;
        const int64_t base = (int64_t)y * n + z0 + lane;
        for (int q = 0; q < n; q++)
        {
          int32_t f = A[base + (int64_t)q * n * n];
          f = f < 0 ? -f : f;
          if (f > NOSITE) f = NOSITE;
          g[q] = (int64_t)f + (int64_t)q * q;
          if (f != NOSITE) l_sites++;
        }
        g[-1] = (int64_t)NOSITE + 1;  // row -1: no site (G(-1) = NOSITE + 1)
        g[n] = (int64_t)NOSITE + (int64_t)n * n;
        uint8_t* kp = keep + (size_t)lane * n;
        l_lines++;
        for (int q = 0; q < n; q++)
        {
          const int is_site = g[q] - (int64_t)q * q < NOSITE;
          int k = is_site;
          // s1
          if (k && 2 * g[q] >= g[q - 1] + g[q + 1]) k = 0;
          kp[0 * 64 * n + q] = k;
          for (int lvl = 1, s = 2; lvl <= 3; lvl++, s *= 2)
          {
            if (k && q - s >= 0 && q + s < n && 2 * g[q] >= g[q - s] + g[q + s]) k = 0;
            kp[(size_t)lvl * 64 * n + q] = k;
          }
        }
        // exact band-local hulls (band rows + one row either side), BAND and 2 * BAND rows
        for (int lvl = 4, band = BAND; lvl <= 5; lvl++, band *= 2)
          for (int r0 = 0; r0 < n; r0 += band)
          {
            int depth = 0;
            const int lo = r0 - 1 < 0 ? 0 : r0 - 1, hi = r0 + band < n ? r0 + band : n - 1;
            for (int q = lo; q <= hi; q++)
            {
              if (g[q] - (int64_t)q * q >= NOSITE) continue;
              while (depth >= 2)
              {
                const Pt a = st[depth - 2], b = st[depth - 1];
                if ((b.G - a.G) * (q - b.r) >= (g[q] - b.G) * (b.r - a.r)) depth--; else break;
              }
              st[depth].G = g[q]; st[depth].r = q; depth++;
            }
            for (int q = r0; q < r0 + band && q < n; q++) kp[(size_t)lvl * 64 * n + q] = 0;
            for (int i = 0; i < depth; i++)
              if (st[i].r >= r0 && st[i].r < r0 + band) kp[(size_t)lvl * 64 * n + st[i].r] = 1;
          }
        // A GLOBAL filter: the lower hull of a subsample of the line's rows (every 8th / 16th row) lies on or above the
        // line's hull, so a row strictly above the subsample hull's chord over it is no vertex.  (Rows of the subsample
        // that are vertices of its hull survive; everything is combined with the s1 test.)
        for (int lvl = 6, step = 8; lvl <= 9; lvl++, step *= 2)
        {
          int depth = 0;
          for (int q = 0; q < n; q += step)
          {
            if (g[q] - (int64_t)q * q >= NOSITE) continue;
            while (depth >= 2)
            {
              const Pt a = st[depth - 2], b = st[depth - 1];
              if ((b.G - a.G) * (q - b.r) >= (g[q] - b.G) * (b.r - a.r)) depth--; else break;
            }
            st[depth].G = g[q]; st[depth].r = q; depth++;
            l_coarse_push[lvl - 6]++;
          }
          l_coarse_depth[lvl - 6] += depth;
          int k = 0;
          for (int q = 0; q < n; q++)
          {
            int keepq = kp[q];  // s1
            if (keepq && depth >= 2)
            {
              while (k + 2 < depth && st[k + 1].r <= q) k++;
              const Pt a = st[k], b = st[k + 1];
              if (q > a.r && q < b.r)
              {
                // on or above the chord a..b ?
                if ((g[q] - a.G) * (b.r - a.r) >= (b.G - a.G) * (q - a.r)) keepq = 0;
              }
            }
            kp[(size_t)lvl * 64 * n + q] = keepq;
          }
        }
        // the real sweep's pushes (s1 filter + "beats the top before the last row"), for reference
        {
          int depth = 0;
          for (int q = 0; q < n; q++)
          {
            if (!kp[q]) continue;
            if (depth > 0 && g[q] - st[depth - 1].G >= 2ll * (n - 1) * (q - st[depth - 1].r)) continue;
            while (depth >= 2)
            {
              const Pt a = st[depth - 2], b = st[depth - 1];
              if ((b.G - a.G) * (q - b.r) >= (g[q] - b.G) * (b.r - a.r)) depth--; else break;
            }
            st[depth].G = g[q]; st[depth].r = q; depth++;
            l_push++;
          }
        }
      }
      for (int r0 = 0; r0 < n; r0 += BAND)
      {
        l_wb++;
        for (int f = 0; f < NF; f++)
        {
          const int band = f == 5 ? 2 * BAND : BAND;
          if (f == 5 && (r0 % (2 * BAND)) != 0) continue;
          int mx = 0;
          for (int lane = 0; lane < 64; lane++)
          {
            int c = 0;
            for (int q = r0; q < r0 + band && q < n; q++) c += keep[(size_t)f * 64 * n + (size_t)lane * n + q];
            l_surv[f] += c;
            if (c > mx) mx = c;
          }
          l_iter[f] += mx;
          l_hist[f][mx]++;
        }
        l_lb += 64;
        for (int f = 0; f < NF; f++)
          for (int q = r0; q < r0 + BAND && q < n; q++)
          {
            int any = 0;
            for (int lane = 0; lane < 64; lane++) any |= keep[(size_t)f * 64 * n + (size_t)lane * n + q];
            l_anyrow[f] += any;
          }
        for (int q = r0; q < r0 + BAND && q < n; q++)
        {
          int any = 0;
          for (int lane = 0; lane < 64; lane++) any |= keep[(size_t)lane * n + q];
          l_any += any;
        }
      }
    }
#pragma omp critical
    {
      for (int f = 0; f < NF; f++)
      {
        surv[f] += l_surv[f]; wave_iter[f] += l_iter[f];
        for (int i = 0; i < 2 * BAND + 3; i++) hist_iter[f][i] += l_hist[f][i];
      }
      for (int i = 0; i < 4; i++) { coarse_push[i] += l_coarse_push[i]; coarse_depth[i] += l_coarse_depth[i]; }
      for (int f = 0; f < NF; f++) anyrow[f] += l_anyrow[f];
      lines += l_lines;
      wave_bands += l_wb; lane_bands += l_lb; sites += l_sites; wave_any_site_rows += l_any; full_stack_push += l_push;
    }
    free(G); free(keep); free(st);
  }
  const char* names[NF] = {"s1", "s1+s2", "s1+s2+s4", "s1..s8", "band hull", "32-row hull", "coarse/8 + s1", "coarse/16 + s1", "coarse/32 + s1", "coarse/64 + s1"};
  printf("X pass, per lane and %d-row band: sites %.2f, real pushes %.2f; wave rows with a surviving site (s1): %.2f of %d\n",
         BAND, sites / lane_bands, full_stack_push / lane_bands, wave_any_site_rows / wave_bands, BAND);
  for (int f = 0; f < NF; f++)
  {
    const double bands = f == 5 ? wave_bands / 2 : wave_bands;
    printf("%-12s survivors per lane-band %.2f | work-list iterations per wave-band %.2f (per 16 rows: %.2f) | hist:", names[f],
           surv[f] / (bands * 64), wave_iter[f] / bands, wave_iter[f] / wave_bands);
    for (int i = 0; i < 2 * BAND + 3; i++)
      if (hist_iter[f][i] > 0) printf(" %d:%.3f", i, hist_iter[f][i] / bands);
    printf(" | wave-rows with a survivor: %.2f of 16", anyrow[f] / wave_bands);
    printf("\n");
  }
  for (int i = 0; i < 4; i++)
    printf("coarse hull of every %dth row: %.1f vertices per line on average, %.1f rows fed to its sweep\n", 8 << i,
           coarse_depth[i] / lines, coarse_push[i] / lines);
  return 0;
}
