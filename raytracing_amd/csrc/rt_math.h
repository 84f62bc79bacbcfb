// Device math for the gfx950 integrator kernels.
//
// Every function is the GLSL 4.30 built-in the reference shader uses, written out in IEEE float32
// with a fixed evaluation order (this file is compiled with -ffp-contract=off, correctly rounded
// division and sqrt, denormals kept), so results are reproducible on any IEEE-754 machine:
//   min(x,y) = y<x ? y : x      max(x,y) = x<y ? y : x     (NaN behaviour follows from these selects)
//   dot = x*x' + y*y' + z*z' summed left to right          normalize(v) = v * (1/sqrt(dot(v,v)))
//   reflect(I,N) = I - (2*dot(N,I))*N                      refract per GLSL spec 8.5
// sin/cos/pow are implementation-defined in GLSL; the core uses the portable forms specified in
// DESIGN.md ("Portable transcendentals"): Cody-Waite reduction + Cephes-style polynomials built from
// + - * only, pow(x,5) = ((x*x)*(x*x))*x, pow(x,2) = x*x.
#pragma once
#include <hip/hip_runtime.h>

namespace rt {

struct v3 { float x, y, z; };

__device__ __forceinline__ v3 mk(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, v3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 operator*(float s, v3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ v3 operator/(v3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ v3 operator+(v3 a, float s) { return mk(a.x + s, a.y + s, a.z + s); }
__device__ __forceinline__ v3 operator-(v3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ v3 operator-(float s, v3 a) { return mk(s - a.x, s - a.y, s - a.z); }

__device__ __forceinline__ float gmin(float x, float y) { return y < x ? y : x; }
__device__ __forceinline__ float gmax(float x, float y) { return x < y ? y : x; }
__device__ __forceinline__ float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
__device__ __forceinline__ float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ v3 cross3(v3 a, v3 b) {
    return mk(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
__device__ __forceinline__ float length3(v3 a) { return sqrtf(dot3(a, a)); }
__device__ __forceinline__ v3 normalize3(v3 a) { return a * (1.0f / sqrtf(dot3(a, a))); }
__device__ __forceinline__ float fract1(float x) { return x - floorf(x); }
__device__ __forceinline__ v3 reflect3(v3 I, v3 N) { return I - (2.0f * dot3(N, I)) * N; }
__device__ __forceinline__ v3 refract3(v3 I, v3 N, float eta) {
    float d = dot3(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return mk(0.f, 0.f, 0.f);
    return eta * I - (eta * d + sqrtf(k)) * N;
}

// shader.comp:153-158
__device__ __forceinline__ float hash13(v3 p) {
    p = mk(fract1(p.x * 0.1031f), fract1(p.y * 0.1031f), fract1(p.z * 0.1031f));
    float d = dot3(p, mk(p.y + 33.33f, p.z + 33.33f, p.x + 33.33f));
    p = p + d;
    return fract1((p.x + p.y) * p.z);
}

// sin and cos of phi in [0, 2*pi] (DESIGN.md "Portable transcendentals").
__device__ __forceinline__ void portable_sincos(float phi, float* s_out, float* c_out) {
    int k = (int)(phi * 0.636619772f + 0.5f);
    float kf = (float)k;
    float r = phi - kf * 1.5703125f;
    r = r - kf * 4.837512969970703125e-4f;
    r = r - kf * 7.54978995489188216e-8f;
    float z = r * r;
    float s = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float c = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    int q = k & 3;
    float ss = (q & 1) ? c : s;
    float cc = (q & 1) ? s : c;
    *s_out = (q & 2) ? -ss : ss;            // q: 0 (s,c) 1 (c,-s) 2 (-s,-c) 3 (-c,s)
    *c_out = (q == 1 || q == 2) ? -cc : cc;
}
__device__ __forceinline__ float pow5(float x) { return ((x * x) * (x * x)) * x; }

}  // namespace rt
