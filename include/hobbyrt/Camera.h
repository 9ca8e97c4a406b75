// hobbyrt/Camera.h -- Camera as PathTracerRenderer and Scene use it (/root/reference/src/Camera.h:7-14,16-90,
// src/Camera.cpp:138-166,204-256): LH view, infinite-far reversed-Z projection, FillPlanarViewConstants (TAA off).
#pragma once

#include "Math.h"
#include "srrhi.h"

namespace hobbyrt {

struct ProjectionParams { float aspectRatio = 16.0f / 9.0f; float fovY = XM_PIDIV4; float nearZ = 0.1f; };

class Camera {
public:
    Vector3 GetPosition() const { return m_Position; }
    float GetYaw() const { return m_Yaw; }
    float GetPitch() const { return m_Pitch; }
    void SetPosition(const Vector3& pos) { m_Position = pos; }
    void SetYaw(float yaw) { m_Yaw = yaw; }
    void SetPitch(float pitch) { m_Pitch = pitch; }
    void SetProjection(const ProjectionParams& proj) { m_Proj = proj; }
    const ProjectionParams& GetProjection() const { return m_Proj; }

    Matrix GetViewMatrix() const;
    Matrix GetProjMatrix() const;
    void FillPlanarViewConstants(srrhi::PlanarViewConstants& constants, float viewportWidth, float viewportHeight) const;

private:
    Vector3 m_Position{ 0.0f, 0.0f, -5.0f };
    float m_Yaw = 0.0f, m_Pitch = 0.0f;
    ProjectionParams m_Proj{};
};

} // namespace hobbyrt
