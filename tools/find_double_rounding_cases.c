// find_double_rounding_cases.c -- searches bilinear-sample inputs on which a CONTRACTED (fused multiply-add) evaluation of
// the reference's blend (sample_eigen.h:82-83) rounds to a different float than the reference's own FMA-free build
// (CMakeLists.txt:25: -msse4.1, no -mfma).  About one input in 5e8 does; the cases found go into
// tests/golden/sampler_double_rounding.json (tools/find_double_rounding_cases.py writes it, adding the expected
// values from the oracle) and pin the engine's sampler -- and the oracle -- to the two-rounding form.
//
//   gcc -O2 -march=native -ffp-contract=off -fopenmp tools/find_double_rounding_cases.c -o /tmp/find_cases -lm
//   /tmp/find_cases <seed> <n_million_samples> <mode>      mode 0: u8 texels (single-channel frames)
//                                                          mode 1: float texels (multi-channel descriptors)
//                                                          mode 2: u8 / gradient texels at positions with x >= 4 or y >= 4 (any
//                                                                  float coordinates): checks the claim behind vlerp_u8_interior
//                                                                  (pba_kernels.h) -- no case may be found
//                                                                  (1e11 samples, seed 5: none found, as the proof says)
// Output lines: "case mode a11 a12 a21 a22 kx ky ok fa fb fh" -- texels (ints, or float bit patterns in mode 1),
// dx = kx 2^-23, dy = ky 2^-23, then the float bit patterns of: the reference form; fma(dy, top, omdy*bot);
// fma(omdy, bot, dy*top); (mode 1) the reference vertical blend over FUSED horizontal blends.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t rng_next(uint64_t* s) {   // splitmix64
  uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static inline uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(int argc, char** argv) {
  const uint64_t seed = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
  const long long n = (argc > 2 ? atoll(argv[2]) : 1000) * 1000000ll;
  const int mode = argc > 3 ? atoi(argv[3]) : 0;
  const long long chunk = 1 << 20;
#pragma omp parallel for schedule(dynamic)
  for (long long c0 = 0; c0 < n; c0 += chunk) {
    uint64_t s = seed * 0x100000001b3ull + (uint64_t)c0;
    for (long long i = 0; i < chunk; ++i) {
      const uint64_t r0 = rng_next(&s), r1 = rng_next(&s), r2 = mode == 1 ? rng_next(&s) : 0;
      float a[4];
      uint32_t raw[4];
      if (mode == 2) {
      } else if (!mode) {
        for (int k = 0; k < 4; ++k) { raw[k] = (uint32_t)((r0 >> (8 * k)) & 0xff); a[k] = (float)raw[k]; }
      } else {
        // blurred bit planes: floats in [0, 1) with full 24-bit significands
        a[0] = (float)((r0 >> 8) & 0xffffff) * 0x1p-24f;  a[1] = (float)((r0 >> 36) & 0xffffff) * 0x1p-24f;
        a[2] = (float)((r2 >> 8) & 0xffffff) * 0x1p-24f;  a[3] = (float)((r2 >> 36) & 0xffffff) * 0x1p-24f;
        for (int k = 0; k < 4; ++k) raw[k] = fbits(a[k]);
      }
      uint32_t kx = (uint32_t)(r1 & 0x7fffff), ky = (uint32_t)((r1 >> 23) & 0x7fffff);
      if (!kx || !ky) continue;
      float dx = (float)kx * 0x1p-23f, dy = (float)ky * 0x1p-23f;
      if (mode == 2) {
        // texels: intensity 0..255 or doubled central difference -255..255; position: random float coordinates the way
        // LinearInitAxis sees them (x in [0, 2048), y in [0, 512), any significand), at least one of them >= 4
        for (int k = 0; k < 4; ++k) { const int v = (int)((r0 >> (9 * k)) & 0x1ff); a[k] = (float)((r0 >> 40) & 1 ? v - 255 : (v & 0xff)); raw[k] = (uint32_t)(int)a[k]; }
        const uint64_t r3 = rng_next(&s);
        // log-uniform magnitudes so that small coordinates (long fractions) are well represented
        const float x = ldexpf((float)(kx | 0x800000) * 0x1p-23f, (int)(r3 % 12) - 1);       // [0.5, 2048)
        const float y = ldexpf((float)(ky | 0x800000) * 0x1p-23f, (int)((r3 >> 8) % 10) - 1); // [0.5, 512)
        if (x < 4.0f && y < 4.0f) continue;
        const int ix = (int)x, iy = (int)y;
        dx = (float)(ix + 1) - x; dy = (float)(iy + 1) - y;
        kx = fbits(x); ky = fbits(y);
      }
      const float omdy = 1.0f - dy;
      const double omdx = 1.0 - (double)dx;
      // reference form (this file is compiled with -ffp-contract=off)
      const double top = (double)(dx * a[0]) + omdx * (double)a[1];
      const double bot = (double)(dx * a[2]) + omdx * (double)a[3];
      const double pa = (double)dy * top, pb = (double)omdy * bot;
      const float ok = (float)(pa + pb);
      const float fa = (float)fma((double)dy, top, pb);
      const float fb = (float)fma((double)omdy, bot, pa);
      float fh = ok;
      if (mode == 1) {
        const double topf = fma(omdx, (double)a[1], (double)(dx * a[0]));
        const double botf = fma(omdx, (double)a[3], (double)(dx * a[2]));
        fh = (float)((double)dy * topf + (double)omdy * botf);
      }
      if (ok != fa || ok != fb || ok != fh) {
#pragma omp critical
        {
          printf("case %d %u %u %u %u %u %u %08x %08x %08x %08x\n", mode, raw[0], raw[1], raw[2], raw[3], kx, ky, fbits(ok), fbits(fa),
                 fbits(fb), fbits(fh));
          fflush(stdout);
        }
      }
    }
  }
  return 0;
}
