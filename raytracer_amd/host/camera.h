// Host-side Camera with the reference's interface (reference src/camera.h:8-55): same constructor arguments, same
// five getters, same fixed eye (-3, 6, 10) looking at (0, 3, 0).  Matrices are produced in GLM's operation order
// (rt_math.h) so that the 35 floats handed to the GPU equal the reference's bit for bit.
#pragma once

#include "rt_math.h"

class Camera
{
public:
    Camera(float fov, float width, float height, float nearPlane, float farPlane)
        : m_Position(-3.0f, 6.0f, 10.0f)
    {
        m_ProjectionMatrix = rtm::perspectiveFov(rtm::radians(fov), width, height, nearPlane, farPlane);
        m_invProjectionMatrix = rtm::inverse(m_ProjectionMatrix);

        const rtm::vec3 upDirection(0.0f, 1.0f, 0.0f);
        m_ViewMatrix = rtm::lookAt(m_Position, rtm::vec3(0.0f, 3.0f, 0.0f), upDirection);
        m_invViewMatrix = rtm::inverse(m_ViewMatrix);
    }

    const rtm::vec3& GetPosition() const { return m_Position; }
    const rtm::mat4& GetProjectionMatrix() const { return m_ProjectionMatrix; }
    const rtm::mat4& GetInverseProjectionMatrix() const { return m_invProjectionMatrix; }
    const rtm::mat4& GetViewMatrix() const { return m_ViewMatrix; }
    const rtm::mat4& GetInverseViewMatrix() const { return m_invViewMatrix; }

private:
    rtm::vec3 m_Position;
    rtm::mat4 m_ProjectionMatrix, m_invProjectionMatrix, m_ViewMatrix, m_invViewMatrix;
};
