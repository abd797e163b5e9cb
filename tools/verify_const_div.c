// exhaustive check: for divisor d and every float a, q' = fmaf(fmaf(-q, d, a), r, q) with r = RN(1/d), q = RN(a * r)
// equals RN(a / d)?  (Markstein's sequence; the claim is only used for the four constants below)
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <omp.h>
int main(void)
{
    const float ds[4] = {0.229f, 0.224f, 0.225f, 255.0f};
    for (int k = 0; k < 4; ++k) {
        const float d = ds[k];
        const float r = 1.0f / d;
        unsigned long long bad = 0, bad_normal = 0;
#pragma omp parallel for reduction(+ : bad, bad_normal) schedule(static)
        for (long long i = 0; i < (1ll << 32); ++i) {
            uint32_t u = (uint32_t)i;
            float a;
            memcpy(&a, &u, 4);
            if (isnan(a) || isinf(a)) continue;
            const float want = a / d;
            const float q = a * r;
            const float e = fmaf(-q, d, a);
            const float got = fmaf(e, r, q);
            if (memcmp(&want, &got, 4) != 0) {
                ++bad;
                if (fabsf(a) >= 1e-30f && fabsf(a) <= 1e30f) ++bad_normal;
            }
        }
        printf("d = %.9g  r = %.9g: %llu mismatches over all finite floats, %llu with 1e-30 <= |a| <= 1e30\n", d, r, bad, bad_normal);
    }
    return 0;
}
