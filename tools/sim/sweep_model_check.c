// Scalar model of one lane of csrc/edt_sweep_kernels.hip (triple-test stack with two row-0 sentinels, strict pops,
// "never owns a row below n" skip, backward evaluation by comparing the two topmost members, bounding rows of the
// other class as distance counters), checked against a brute-force evaluation on random lines.
// gcc -O2 -o sweep_model_check sweep_model_check.c && ./sweep_model_check
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define NMAX 64
static int64_t brute(const int32_t* F, const uint8_t* neg, int n, int q, int64_t none)
{
  int64_t best = none;
  for (int r = 0; r < n; r++)
  {
    if (F[r] >= 0) { int64_t v = (int64_t)(q - r) * (q - r) + F[r]; if (v < best) best = v; }
    if (neg[r] != neg[q]) { int64_t v = (int64_t)(q - r) * (q - r); if (v < best) best = v; }
  }
  return best;
}

int main(void)
{
  const int64_t kSentinelG = 1 << 22;  // T0 = kSentinelG - 2, V0 = kSentinelG - 1
  srand(12345);
  long checked = 0;
  for (int trial = 0; trial < 2000000; trial++)
  {
    const int n = 1 + rand() % NMAX;
    int32_t F[NMAX]; uint8_t neg[NMAX];
    const int mode = rand() % 6;
    const int pfin = rand() % 101, ptr = rand() % 30;
    int cls = rand() & 1;
    for (int r = 0; r < n; r++)
    {
      if (rand() % 100 < ptr) cls ^= 1;
      neg[r] = cls;
      int32_t v;
      switch (mode) {
        case 0: v = rand() % 4; break;
        case 1: v = rand() % 4000; break;
        case 2: v = (r - n / 2) * (r - n / 2) + rand() % 3; break;
        case 3: v = 1000 - r * r / 4 + rand() % 5; if (v < 0) v = 0; break;
        case 4: v = (rand() % 50) * (rand() % 50); break;
        default: v = 1 + rand() % 2; break;
      }
      F[r] = (rand() % 100 < pfin) ? v : -1;  // -1: no site
    }
    // forward sweep
    int64_t SG[NMAX + 2]; int SR[NMAX + 2]; int depth = 0;
    SG[depth] = kSentinelG - 1; SR[depth++] = 0;
    SG[depth] = kSentinelG - 2; SR[depth++] = 0;
    for (int q = 0; q < n; q++)
    {
      if (F[q] < 0) continue;
      const int64_t G = (int64_t)F[q] + (int64_t)q * q;
      for (;;)
      {
        const int64_t Gt = SG[depth - 1], Gs = SG[depth - 2]; const int rt = SR[depth - 1], rs = SR[depth - 2];
        const int64_t t = (G - Gt) * (rt - rs) + (Gs - Gt) * (q - rt);
        if (t < 0) depth--; else break;
      }
      const int64_t Gt = SG[depth - 1]; const int rt = SR[depth - 1];
      if (G - Gt < 2ll * (n - 1) * (q - rt)) { SG[depth] = G; SR[depth++] = q; }
    }
    // backward evaluation
    int dn = 32768;
    for (int q = n - 1; q >= 0; q--)
    {
      while ((SG[depth - 2] - SG[depth - 1]) + 2ll * q * (SR[depth - 1] - SR[depth - 2]) <= 0) depth--;
      int64_t val = SG[depth - 1] + (int64_t)q * (q - 2 * SR[depth - 1]);
      dn = (q < n - 1 && neg[q] != neg[q + 1]) ? 1 : dn + 1;
      int dp = 32768;
      for (int r = q - 1; r >= 0; r--) if (neg[r] != neg[q]) { dp = q - r; break; }
      const int dm = dp < dn ? dp : dn;
      int64_t best = val < (int64_t)dm * dm ? val : (int64_t)dm * dm;
      if (best >= kSentinelG - 2) best = -1;
      int64_t want = brute(F, neg, n, q, INT64_MAX);
      if (want == INT64_MAX) want = -1;
      if (best != want) { printf("MISMATCH trial %d n %d q %d got %lld want %lld\n", trial, n, q, (long long)best, (long long)want); return 1; }
      checked++;
    }
  }
  printf("ok: %ld rows checked\n", checked);
  return 0;
}
