"""CPU: the on-grid test of the float32 kernels (rank_hist.hpp: grid_key) restated in C and checked exhaustively.
x is accepted iff x == RN32(k / 1000) for k = rint(1000 x), |k| <= 32 767, with the quotient formed by Markstein's
sequence (q0 = k r, rem = fma(-q0, 1000, k), q = fma(rem, r, q0), r = RN32(1/1000)) instead of a division: the sequence must
equal the float32 division for every key, and must reject both float32 neighbours of every grid value — then equal keys
<=> equal samples, and the integer keys order and tie exactly as the floats do."""
import os
import subprocess
import tempfile

SRC = r'''
#include <math.h>
#include <stdio.h>
static int grid_key(float x, int* k) {
  const float t = rintf(x * 1000.0f);
  const float r = 1.0e-3f;
  const float q0 = t * r;
  const float rem = fmaf(-q0, 1000.0f, t);
  const float q = fmaf(rem, r, q0);
  *k = (int)t;
  return q == x && fabsf(t) <= 32767.0f;
}
int main(void) {
  long bad = 0;
  for (int k = -40000; k <= 40000; ++k) {
    const float x = (float)k / 1000.0f;                 /* the canonical float32 of k milli-units */
    int kk; const int ok = grid_key(x, &kk);
    const int want = k >= -32767 && k <= 32767;
    if (ok != want || (ok && kk != k)) { ++bad; if (bad < 10) printf("k=%d ok=%d kk=%d\n", k, ok, kk); }
    if (k != 0) {                                       /* both neighbours are off the grid */
      int k2;
      if (grid_key(nextafterf(x, 1e9f), &k2) || grid_key(nextafterf(x, -1e9f), &k2)) { ++bad; if (bad < 10) printf("neighbour of k=%d accepted\n", k); }
    }
  }
  /* double rounding: a float64 event value k / 1000.0 stored as float32 is the same float */
  for (int k = -32767; k <= 32767; ++k) if ((float)((double)k / 1000.0) != (float)k / 1000.0f) { ++bad; if (bad < 10) printf("double rounding at k=%d\n", k); }
  int k0; if (!grid_key(-0.0f, &k0) || k0 != 0 || grid_key(NAN, &k0) || grid_key(INFINITY, &k0) || grid_key(3.4028234663852886e38f, &k0)) { ++bad; printf("special values\n"); }
  printf("bad=%ld\n", bad);
  return bad != 0;
}
'''


def test_markstein_quotient_equals_division_for_every_key():
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, 'g.c')
        open(c, 'w').write(SRC)
        subprocess.check_call(['gcc', '-O1', '-ffp-contract=off', c, '-o', os.path.join(d, 'g'), '-lm'])
        out = subprocess.run([os.path.join(d, 'g')], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith('bad=0'), out.stdout
