// hobbyrt/Math.h -- the few DirectXMath types / functions the path-tracer plugin surface touches, so that
// reference-side code such as `vq.m_Pos = Vector3{...}`, `world._41`, `Matrix m{}` keeps compiling
// (/root/reference/src/pch.h:53-64 aliases them to DirectX::XMFLOAT*). Row-major, row-vector convention.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

namespace hobbyrt {

struct Vector2 { float x = 0, y = 0; Vector2() = default; Vector2(float x_, float y_) : x(x_), y(y_) {} };
struct Vector3 { float x = 0, y = 0, z = 0; Vector3() = default; Vector3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {} };
struct Vector4 { float x = 0, y = 0, z = 0, w = 0; Vector4() = default; Vector4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {} };
using Quaternion = Vector4;
struct Vector2U { uint32_t x = 0, y = 0; };

struct Matrix {   // XMFLOAT4X4
    union {
        struct { float _11, _12, _13, _14, _21, _22, _23, _24, _31, _32, _33, _34, _41, _42, _43, _44; };
        float m[4][4];
    };
    Matrix() { std::memset(m, 0, sizeof m); }
    static Matrix Identity() { Matrix r; r._11 = r._22 = r._33 = r._44 = 1.0f; return r; }
};
static_assert(sizeof(Matrix) == 64 && sizeof(Vector3) == 12 && sizeof(Vector4) == 16, "DirectXMath-compatible layouts");

constexpr float XM_PI = 3.141592654f;
constexpr float XM_PIDIV4 = 0.785398163f;

inline Matrix MatrixMultiply(const Matrix& a, const Matrix& b)
{
    Matrix r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double s = 0; for (int k = 0; k < 4; ++k) s += (double)a.m[i][k] * b.m[k][j];
        r.m[i][j] = (float)s;
    }
    return r;
}
// general 4x4 inverse (Gauss-Jordan in double); returns false when singular
inline bool MatrixInverse(const Matrix& a, Matrix& out)
{
    double w[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { w[i][j] = a.m[i][j]; w[i][j + 4] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int p = c; for (int r = c + 1; r < 4; ++r) if (std::fabs(w[r][c]) > std::fabs(w[p][c])) p = r;
        if (w[p][c] == 0.0) return false;
        if (p != c) for (int j = 0; j < 8; ++j) { double t = w[c][j]; w[c][j] = w[p][j]; w[p][j] = t; }
        double inv = 1.0 / w[c][c];
        for (int j = 0; j < 8; ++j) w[c][j] *= inv;
        for (int r = 0; r < 4; ++r) if (r != c) { double f = w[r][c]; if (f != 0.0) for (int j = 0; j < 8; ++j) w[r][j] -= f * w[c][j]; }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out.m[i][j] = (float)w[i][j + 4];
    return true;
}
// XMMatrixRotationQuaternion(XMQuaternionRotationRollPitchYaw(pitch, yaw, 0)) for roll 0: R = Rx(pitch) * Ry(yaw)
inline Matrix MatrixRotationPitchYaw(float pitch, float yaw)
{
    double cp = std::cos((double)pitch), sp = std::sin((double)pitch), cy = std::cos((double)yaw), sy = std::sin((double)yaw);
    Matrix r = Matrix::Identity();
    r._11 = (float)cy;          r._12 = 0.0f;      r._13 = (float)-sy;
    r._21 = (float)(sp * sy);   r._22 = (float)cp; r._23 = (float)(sp * cy);
    r._31 = (float)(cp * sy);   r._32 = (float)-sp; r._33 = (float)(cp * cy);
    return r;
}
inline Vector3 TransformNormal(const Vector3& v, const Matrix& m)   // XMVector3TransformNormal (row vector, no translation)
{
    return Vector3(v.x * m._11 + v.y * m._21 + v.z * m._31, v.x * m._12 + v.y * m._22 + v.z * m._32, v.x * m._13 + v.y * m._23 + v.z * m._33);
}
inline Vector3 Normalize(const Vector3& v)
{
    float l = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return l > 0.0f ? Vector3(v.x / l, v.y / l, v.z / l) : v;
}
inline Matrix MatrixLookToLH(const Vector3& pos, const Vector3& fwd, const Vector3& upHint)   // XMMatrixLookToLH
{
    Vector3 z = Normalize(fwd);
    Vector3 x = Normalize(Vector3(upHint.y * z.z - upHint.z * z.y, upHint.z * z.x - upHint.x * z.z, upHint.x * z.y - upHint.y * z.x));
    Vector3 y(z.y * x.z - z.z * x.y, z.z * x.x - z.x * x.z, z.x * x.y - z.y * x.x);
    Matrix r = Matrix::Identity();
    r._11 = x.x; r._21 = x.y; r._31 = x.z; r._41 = -(pos.x * x.x + pos.y * x.y + pos.z * x.z);
    r._12 = y.x; r._22 = y.y; r._32 = y.z; r._42 = -(pos.x * y.x + pos.y * y.y + pos.z * y.z);
    r._13 = z.x; r._23 = z.y; r._33 = z.z; r._43 = -(pos.x * z.x + pos.y * z.y + pos.z * z.z);
    return r;
}

} // namespace hobbyrt
