// Camera.cpp -- /root/reference/src/Camera.cpp:138-166 (view / infinite reversed-Z projection) and :204-256
// (FillPlanarViewConstants with TAA off: the jittered matrices equal the non-jittered ones).
#include "../../../include/hobbyrt/Camera.h"

namespace hobbyrt {

Matrix Camera::GetViewMatrix() const
{
    Matrix rot = MatrixRotationPitchYaw(m_Pitch, m_Yaw);
    Vector3 forward = TransformNormal(Vector3(0, 0, 1), rot), up = TransformNormal(Vector3(0, 1, 0), rot);
    return MatrixLookToLH(m_Position, forward, up);
}

Matrix Camera::GetProjMatrix() const
{
    double yScale = 1.0 / std::tan(0.5 * (double)m_Proj.fovY);
    double xScale = yScale / (double)m_Proj.aspectRatio;
    Matrix m;
    m._11 = (float)xScale; m._22 = (float)yScale; m._33 = 0.0f; m._34 = 1.0f; m._43 = m_Proj.nearZ; m._44 = 0.0f;
    return m;
}

void Camera::FillPlanarViewConstants(srrhi::PlanarViewConstants& c, float viewportWidth, float viewportHeight) const
{
    Matrix view = GetViewMatrix(), proj = GetProjMatrix();
    Matrix viewProj = MatrixMultiply(view, proj);
    Matrix invView, invProj, invViewProj;
    MatrixInverse(view, invView); MatrixInverse(proj, invProj); MatrixInverse(viewProj, invViewProj);
    c.m_MatWorldToView = view; c.m_MatViewToWorld = invView;
    c.m_MatViewToClip = proj; c.m_MatWorldToClip = viewProj; c.m_MatClipToView = invProj; c.m_MatClipToWorld = invViewProj;
    c.m_MatViewToClipNoOffset = proj; c.m_MatWorldToClipNoOffset = viewProj; c.m_MatClipToViewNoOffset = invProj; c.m_MatClipToWorldNoOffset = invViewProj;
    c.m_ViewportOrigin = Vector2(0, 0);
    c.m_ViewportSize = Vector2(viewportWidth, viewportHeight);
    c.m_ViewportSizeInv = Vector2(1.0f / viewportWidth, 1.0f / viewportHeight);
    c.m_PixelOffset = Vector2(0, 0);
    c.m_ClipToWindowScale = Vector2(0.5f * viewportWidth, -0.5f * viewportHeight);
    c.m_ClipToWindowBias = Vector2(0.5f * viewportWidth, 0.5f * viewportHeight);
}

} // namespace hobbyrt
