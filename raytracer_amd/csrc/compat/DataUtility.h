// Adapter header with the reference's name: the part of DataUtility.h the front end touches (App.cpp:96-98,108-137) —
// Camera, with the same public members — implemented on the renderer's C ABI.  GLM comes from the front end's own
// include path, as it does for App.cpp.
#pragma once
#include <glm.hpp>

#include "../../../include/rt_engine.h"

struct Ray {   // DataUtility.h:43-49 of the reference; World::GetRayColor takes one
    Ray() = default;
    Ray(glm::vec3 p, glm::vec3 d) : pos(p), direction(d) {}
    glm::vec3 pos{0};
    glm::vec3 direction{0};
};

// Camera (reference DataUtility.h:136-154): pos is public and written directly by the front end; the basis is derived
// from the look vector exactly as Camera::SetForwardVector does (DataUtility.cpp:10-15), through rt_camera_look.
struct Camera {
    Camera(glm::vec3 p, const glm::vec3& look) : pos(p) { SetForwardVector(look); }
    Camera() : pos({0, 0, 0}) { SetForwardVector({1, 0, 0}); }

    void SetForwardVector(const glm::vec3& look) {
        const float zero[3] = {0, 0, 0}, l[3] = {look.x, look.y, look.z};
        float cam[12];
        rt_camera_look(zero, l, cam);
        forwardVector = {cam[3], cam[4], cam[5]};
        uVec = {cam[6], cam[7], cam[8]};
        vVec = {cam[9], cam[10], cam[11]};
    }
    const glm::vec3& GetForwardVector() { return forwardVector; }
    const glm::vec3& GetUDirection() { return uVec; }
    const glm::vec3& GetVDirection() { return vVec; }
    void Pack(float cam12[12]) const {
        const glm::vec3 v[4] = {pos, forwardVector, uVec, vVec};
        for (int i = 0; i < 4; i++) { cam12[3 * i] = v[i].x; cam12[3 * i + 1] = v[i].y; cam12[3 * i + 2] = v[i].z; }
    }

    glm::vec3 pos;

private:
    glm::vec3 forwardVector, uVec, vVec;
};
