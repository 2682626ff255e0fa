/* TEST INFRASTRUCTURE.  Exhaustive proof-by-enumeration of the uint16 depth normalisation used by the HIP
 * polylines / naive kernels (csrc/ds_stereo_polylines.hip, pl_load_nd):
 *     nd = (depth - min) / (max - min)       stereoimage_generation.py:79-81, uint16 -> float64 true division
 * is computed on the device as  y = 1.0/b;  q0 = a*y;  r = fma(-b, q0, a);  q = fma(r, y, q0)  with a, b integers in
 * [0, 65535].  This program checks q == a/b (IEEE binary64, round to nearest) for EVERY pair: 65535 * 65536 cases.
 * Prints "bad=0 of 4294901760" and exits 0 when the identity holds.   gcc -O2 -fopenmp -ffp-contract=off */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <omp.h>
int main(void){
  long long bad=0, tot=0;
  #pragma omp parallel for schedule(dynamic,64) reduction(+:bad,tot)
  for (int b=1;b<=65535;b++){
    volatile double y = 1.0/(double)b;
    double yy=y, db=(double)b;
    for (int a=0;a<=65535;a++){
      double da=(double)a;
      double q0=da*yy;
      double rem=fma(-db,q0,da);
      double q=fma(rem,yy,q0);
      double ref=da/db;
      if (q!=ref) bad++;
      tot++;
    }
  }
  printf("bad=%lld of %lld\n",bad,tot);
  return bad != 0;
}
