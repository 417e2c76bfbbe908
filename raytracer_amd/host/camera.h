// Host-side Camera with the reference's public interface (reference src/camera.h:8-55): constructible from
// (fov, width, height, near, far), five getters, fixed eye (-3, 6, 10) looking at (0, 3, 0) with +y up.
// Internally it is a pair of (matrix, inverse) transforms built with rt_math.h, whose functions follow GLM's operation
// order, so the 35 floats the GPU consumes equal the reference's bit for bit (tests/golden/camera_bits.json).
#pragma once

#include "rt_math.h"

class Camera
{
    // a transform and its inverse, computed together
    struct Transform
    {
        rtm::mat4 forward, backward;
        void set(const rtm::mat4& m) { forward = m; backward = rtm::inverse(m); }
    };

    rtm::vec3 eye_;
    Transform lens_;    // view space -> clip space
    Transform pose_;    // world space -> view space

public:
    Camera(float fovDegrees, float frameWidth, float frameHeight, float zNear, float zFar)
        : eye_(-3.0f, 6.0f, 10.0f)                                                      // camera.h:12
    {
        const rtm::vec3 target(0.0f, 3.0f, 0.0f), up(0.0f, 1.0f, 0.0f);                 // camera.h:19-20
        pose_.set(rtm::lookAt(eye_, target, up));
        lens_.set(rtm::perspectiveFov(rtm::radians(fovDegrees), frameWidth, frameHeight, zNear, zFar));   // camera.h:15
    }

    // the reference's getters
    const rtm::vec3& GetPosition() const { return eye_; }
    const rtm::mat4& GetViewMatrix() const { return pose_.forward; }
    const rtm::mat4& GetInverseViewMatrix() const { return pose_.backward; }
    const rtm::mat4& GetProjectionMatrix() const { return lens_.forward; }
    const rtm::mat4& GetInverseProjectionMatrix() const { return lens_.backward; }
};
