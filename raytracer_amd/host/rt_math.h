// Minimal float32 vector / matrix types for the host side of the backend.  They stand in for the GLM types the
// reference's headers use (glm::vec3, glm::mat4) and reproduce GLM 0.9.9.8's scalar operation order where the
// results are consumed bit-for-bit by the GPU path: dot, normalize, cross, mat4 inverse, perspectiveFov, lookAt.
// Citations are relative to the reference's src/vendor/glm.
#pragma once

#include <cmath>

namespace rtm {

struct vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
    vec3() = default;
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};

inline vec3 operator+(const vec3& a, const vec3& b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline vec3 operator-(const vec3& a, const vec3& b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline vec3 operator-(const vec3& a) { return { -a.x, -a.y, -a.z }; }
inline vec3 operator*(const vec3& a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline vec3 operator*(float s, const vec3& a) { return { s * a.x, s * a.y, s * a.z }; }
inline bool operator==(const vec3& a, const vec3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
// component-wise forms the reference's pixel loop and glow code spell (src/main.cpp:197, 313-319)
inline vec3 operator*(const vec3& a, const vec3& b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
inline vec3 operator/(const vec3& a, const vec3& b) { return { a.x / b.x, a.y / b.y, a.z / b.z }; }
inline vec3 operator/(const vec3& a, float s) { return { a.x / s, a.y / s, a.z / s }; }
inline vec3& operator+=(vec3& a, const vec3& b) { a = a + b; return a; }
inline vec3& operator*=(vec3& a, float s) { a = a * s; return a; }
// detail/func_common.inl:20,29,244: min(x, y) = (y < x) ? y : x, max(x, y) = (x < y) ? y : x, clamp = min(max(x, lo), hi)
template <class T> inline T min(T x, T y) { return (y < x) ? y : x; }
template <class T> inline T max(T x, T y) { return (x < y) ? y : x; }
inline vec3 clamp(const vec3& v, const vec3& lo, const vec3& hi)
{
    return { min(max(v.x, lo.x), hi.x), min(max(v.y, lo.y), hi.y), min(max(v.z, lo.z), hi.z) };
}

// detail/func_geometric.inl:48-55: tmp = a * b; tmp.x + tmp.y + tmp.z
inline float dot(const vec3& a, const vec3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// detail/func_geometric.inl:88 with inversesqrt = 1 / sqrt (detail/func_exponential.inl:138)
inline vec3 normalize(const vec3& a) { return a * (1.0f / std::sqrt(dot(a, a))); }
// detail/func_geometric.inl:70-76
inline vec3 cross(const vec3& x, const vec3& y)
{
    return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y };
}

// column-major 4x4 like glm::mat4: m[col][row]
struct mat4 {
    float m[4][4];
    mat4() : mat4(0.0f) {}
    explicit mat4(float diag)
    {
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) m[c][r] = (c == r) ? diag : 0.0f;
    }
    float* operator[](int c) { return m[c]; }
    const float* operator[](int c) const { return m[c]; }
};

// detail/func_trigonometric.inl:13
inline float radians(float degrees) { return degrees * static_cast<float>(0.01745329251994329576923690768489); }

// ext/matrix_clip_space.inl:372-389 (perspectiveFov -> perspectiveFovRH_NO under the default clip-space config)
inline mat4 perspectiveFov(float fov, float width, float height, float zNear, float zFar)
{
    const float rad = fov;
    const float h = std::cos(0.5f * rad) / std::sin(0.5f * rad);
    const float w = h * height / width;
    mat4 R(0.0f);
    R[0][0] = w;
    R[1][1] = h;
    R[2][2] = -(zFar + zNear) / (zFar - zNear);
    R[2][3] = -1.0f;
    R[3][2] = -(2.0f * zFar * zNear) / (zFar - zNear);
    return R;
}

// ext/matrix_transform.inl:99-119 (lookAt -> lookAtRH)
inline mat4 lookAt(const vec3& eye, const vec3& center, const vec3& up)
{
    const vec3 f(normalize(center - eye));
    const vec3 s(normalize(cross(f, up)));
    const vec3 u(cross(s, f));
    mat4 R(1.0f);
    R[0][0] = s.x; R[1][0] = s.y; R[2][0] = s.z;
    R[0][1] = u.x; R[1][1] = u.y; R[2][1] = u.z;
    R[0][2] = -f.x; R[1][2] = -f.y; R[2][2] = -f.z;
    R[3][0] = -dot(s, eye);
    R[3][1] = -dot(u, eye);
    R[3][2] = dot(f, eye);
    return R;
}

// detail/func_matrix.inl:294-351 (cofactor expansion, same sub-expressions and association)
inline mat4 inverse(const mat4& m)
{
    const float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    const float c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    const float c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    const float c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    const float c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    const float c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    const float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    const float c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    const float c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    const float c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    const float c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    const float c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    const float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    const float c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    const float c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    const float c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    const float c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    const float c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];

    const float f0[4] = { c00, c00, c02, c03 }, f1[4] = { c04, c04, c06, c07 }, f2[4] = { c08, c08, c10, c11 };
    const float f3[4] = { c12, c12, c14, c15 }, f4[4] = { c16, c16, c18, c19 }, f5[4] = { c20, c20, c22, c23 };
    const float v0[4] = { m[1][0], m[0][0], m[0][0], m[0][0] }, v1[4] = { m[1][1], m[0][1], m[0][1], m[0][1] };
    const float v2[4] = { m[1][2], m[0][2], m[0][2], m[0][2] }, v3[4] = { m[1][3], m[0][3], m[0][3], m[0][3] };
    const float sa[4] = { +1, -1, +1, -1 }, sb[4] = { -1, +1, -1, +1 };

    mat4 inv(0.0f);
    for (int i = 0; i < 4; ++i) {
        inv[0][i] = (v1[i] * f0[i] - v2[i] * f1[i] + v3[i] * f2[i]) * sa[i];
        inv[1][i] = (v0[i] * f0[i] - v2[i] * f3[i] + v3[i] * f4[i]) * sb[i];
        inv[2][i] = (v0[i] * f1[i] - v1[i] * f3[i] + v3[i] * f5[i]) * sa[i];
        inv[3][i] = (v0[i] * f2[i] - v1[i] * f4[i] + v2[i] * f5[i]) * sb[i];
    }
    const float d0 = m[0][0] * inv[0][0], d1 = m[0][1] * inv[1][0], d2 = m[0][2] * inv[2][0], d3 = m[0][3] * inv[3][0];
    const float det = (d0 + d1) + (d2 + d3);
    const float ood = 1.0f / det;
    mat4 out(0.0f);
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) out[c][r] = inv[c][r] * ood;
    return out;
}

}  // namespace rtm
