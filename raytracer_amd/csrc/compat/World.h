// Adapter header with the reference's name (World.h:12-32): the scene and camera owner the front end reaches through
// FrameManager::GetWorld().  The scene lives in an rt_scene, the pixels come from an rt_multi: one engine per GPU named in
// RT_DEVICES (a comma-separated list of HIP device ordinals; default: RT_DEVICE, else device 0), the first being the one
// that holds the frame.
#pragma once
#include <string>

#include "Constants.h"      // the front end's own quality settings (Constants.h:4-22), unchanged
#include "DataUtility.h"

class World {
public:
    // assets are looked up as "<asset root>/src/stb_image/..." like the reference's relative paths (World.cpp:26,49);
    // the root defaults to the current directory and can be set with RT_ASSET_DIR
    explicit World(float time = 0);
    ~World();
    World(const World&) = delete;
    World& operator=(const World&) = delete;

    void CreateWithNewTime(float time);                              // World.cpp:59-66
    glm::u8vec3 CalculateColorForScreenPosition(int x, int y);       // World.cpp:68-102 (one-pixel render window)
    glm::vec3 GetRayColor(const Ray& ray, int bounceAmount = 0);     // World.cpp:104-171
    Camera& GetWorldCamera() { return camera; }

    // adapter internals used by FrameManager
    rt_multi* Devices() { return devices; }
    rt_engine* Engine() { return rt_multi_engine(devices, 0); }   // the display engine: single-pixel and single-ray queries
    void SyncCamera();
    uint32_t frameSeed = 1;      // keys the per-(pixel, sample) RNG streams of the next frame
    uint32_t buildKey = 1;       // keys the BVH axis choices (the reference draws them from std::rand, Object.cpp:191)

private:
    void BuildScene(float time);
    rt_scene* scene = nullptr;
    rt_multi* devices = nullptr;
    Camera camera;
};
