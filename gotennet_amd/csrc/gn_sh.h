// gn_sh.h -- real harmonics of a unit vector, degrees 1..LMAX (LMAX <= 8), l = 0 omitted.
//
// Restates TensorInit._calculate_components (reference layers.py:805-1494): the
// degree-1 block is (x, y, z); degree 2 uses the closed polynomials; degrees 3 and 4
// are the e3nn-style recursions on the degree below, and so are degrees 5..8
// (gn_sh_high.h, generated: tools/gen_sh_table.py derives the coupling coefficients).
// The same literal expression order is kept for degrees <= 4 so fp32 rounding stays
// close to the reference.  NOTE the reference's degree 3 mixes the recursion (rows
// 0, 1, 5, 6) with closed polynomials in another normalisation (rows 2, 3, 4), so its
// degrees >= 3 are NOT rotation-covariant harmonics: sum_m Y_3m^2 varies between 2.5 and
// 7 over the sphere.  This file follows the reference, not the textbook.  Templated on
// the scalar type so the backward pass can push dual numbers through the same code.
#pragma once
#include "gn_sh_high.h"

namespace gn {

template <int LMAX, typename T>
__host__ __device__ inline void real_harmonics(const T x, const T y, const T z, T* __restrict__ o) {
    o[0] = x; o[1] = y; o[2] = z;
    if constexpr (LMAX >= 2) {
        const float r3 = 1.7320508075688772f;
        const T y2 = y * y;
        const T x2z2 = x * x + z * z;
        o[3] = r3 * x * z;
        o[4] = r3 * x * y;
        o[5] = y2 - 0.5f * x2z2;
        o[6] = r3 * y * z;
        o[7] = (r3 / 2.0f) * (z * z - x * x);
        if constexpr (LMAX >= 3) {
            const float a = 1.0801234497346435f;   // sqrt(42)/6
            const float b = 2.6457513110645907f;   // sqrt(7)
            const float c = 1.6201851746019651f;   // sqrt(168)/8
            const T* s2 = o + 3;
            T* s3 = o + 8;
            s3[0] = a * (s2[0] * z + s2[4] * x);
            s3[1] = b * s2[0] * y;
            s3[2] = c * (4.0f * y2 - x2z2) * x;
            s3[3] = (0.5f * b) * y * (2.0f * y2 - 3.0f * x2z2);
            s3[4] = c * z * (4.0f * y2 - x2z2);
            s3[5] = b * s2[4] * y;
            s3[6] = a * (s2[4] * z - s2[0] * x);
            if constexpr (LMAX >= 4) {
                const float k0 = 1.0606601717798212f;    // (3/4) sqrt(2)
                const float k1 = 0.75f;
                const float k2 = 0.9185586535436918f;    // (3/8) sqrt(6)
                const float k3 = 0.20044593143431830f;   // (3/56) sqrt(14)
                const float k4 = 0.9819805060619657f;    // (3/14) sqrt(21)
                const float k5 = 0.7763237542601484f;    // (3/56) sqrt(210)
                const float k6 = 0.3471825374147068f;    // (3/56) sqrt(42)
                const float k7 = 1.0978875820671000f;    // (3/28) sqrt(105)
                const float k8 = 0.8964214570007952f;    // (3/28) sqrt(70)
                const float k9 = 0.6943650748294136f;    // (3/28) sqrt(42)
                const float k10 = 1.1338934190276817f;   // (3/7) sqrt(7)
                T* s4 = o + 15;
                s4[0] = k0 * (s3[0] * z + s3[6] * x);
                s4[1] = k1 * s3[0] * y + k2 * s3[1] * z + k2 * s3[5] * x;
                s4[2] = -k3 * s3[0] * z + k4 * s3[1] * y + k5 * s3[2] * z + k5 * s3[4] * x + k3 * s3[6] * x;
                s4[3] = -k6 * s3[1] * z + k7 * s3[2] * y + k8 * s3[3] * x + k6 * s3[5] * x;
                s4[4] = -k9 * s3[2] * x + k10 * s3[3] * y - k9 * s3[4] * z;
                s4[5] = -k6 * s3[1] * x + k8 * s3[3] * z + k7 * s3[4] * y - k6 * s3[5] * z;
                s4[6] = -k3 * s3[0] * x - k5 * s3[2] * x + k5 * s3[4] * z + k4 * s3[5] * y - k3 * s3[6] * z;
                s4[7] = -k2 * s3[1] * x + k2 * s3[5] * z + k1 * s3[6] * y;
                s4[8] = k0 * (-s3[0] * x + s3[6] * z);
                if constexpr (LMAX >= 5) sh_raise_5(o + 15, x, y, z, o + 24);
                if constexpr (LMAX >= 6) sh_raise_6(o + 24, x, y, z, o + 35);
                if constexpr (LMAX >= 7) sh_raise_7(o + 35, x, y, z, o + 48);
                if constexpr (LMAX >= 8) sh_raise_8(o + 48, x, y, z, o + 63);
            }
        }
    }
}

}  // namespace gn
