// World and FrameManager of the adapter (see World.h / FrameManager.h in this directory).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "FrameManager.h"

namespace {
void check(int rc, const char* what) {
    // the reference breaks into the debugger on asset errors (DataUtility.cpp:24-27) and rethrows thread-spawn errors
    // (FrameManager.cpp:16-25); the adapter reports and throws
    if (rc < 0) {
        std::cout << what << ": " << rt_last_error() << std::endl;
        throw std::runtime_error(std::string(what) + ": " + rt_last_error());
    }
}
}  // namespace

World::World(float time) : camera({glm::vec3{0, 3, -10}, glm::vec3{0, 0, 1}}) {   // World.cpp:48
    rt_config cfg;
    rt_config_defaults(&cfg);
    cfg.width = WINDOW_WIDTH; cfg.height = WINDOW_HEIGHT;
    cfg.samples_per_axis = SAMPLES_PER_PIXEL_AXIS; cfg.bounces = LIGHT_BOUNCE_AMOUNT;
    cfg.fov = FIELD_OF_VIEW; cfg.skybox_brightness = SKYBOX_BRIGHTNESS;
    cfg.dof_intensity = DEPTH_OF_FIELD_INTENSITY; cfg.focus_distance = FOCUS_DISTANCE;
    // FrameManager.cpp:95-107 spreads a worker's pixels over its range (stride 37) so that a frame in progress shows all
    // over the window.  Here a frame in progress shows the batches that have finished (GetTexturePixelsToPresent), so a
    // frame of many paths is cut into batches of 2^24 paths (3 ms of a 1080p, 64 spp, 8 bounce frame each: every
    // presentation at 24 FPS finds new tiles) instead of the engine's 2^27 (fewest launches; what a caller who wants the
    // finished frame only takes).  RT_PROGRESSIVE=0 keeps the engine's default.
    {
        const long long paths = (long long)WINDOW_WIDTH * WINDOW_HEIGHT * SAMPLES_PER_PIXEL_AXIS * SAMPLES_PER_PIXEL_AXIS;
        const char* v = getenv("RT_PROGRESSIVE");
        if (paths > (1ll << 25) && !(v && atoi(v) == 0)) cfg.max_batch_paths = 1 << 24;
    }
    std::vector<int32_t> ids;
    if (const char* list = getenv("RT_DEVICES")) {
        for (const char* p = list; *p;) {
            char* end = nullptr;
            const long v = strtol(p, &end, 10);
            if (end == p) break;
            ids.push_back((int32_t)v);
            p = (*end == ',') ? end + 1 : end;
        }
    }
    if (ids.empty()) ids.push_back(getenv("RT_DEVICE") ? atoi(getenv("RT_DEVICE")) : 0);
    check(rt_multi_create(&cfg, ids.data(), (int)ids.size(), &devices), "rt_multi_create");
    BuildScene(time);
}
World::~World() {
    rt_multi_destroy(devices);
    rt_scene_destroy(scene);
}
void World::BuildScene(float time) {
    rt_scene* fresh = nullptr;
    check(rt_scene_create(&fresh), "rt_scene_create");
    const char* root = getenv("RT_ASSET_DIR");
    int rc = rt_scene_add_default_objects(fresh, root ? root : ".", time);
    if (rc >= 0) rc = rt_scene_build(fresh, buildKey);
    if (rc < 0) { rt_scene_destroy(fresh); check(rc, "scene"); }
    check(rt_multi_set_scene(devices, fresh), "rt_multi_set_scene");
    rt_scene_destroy(scene);
    scene = fresh;
}
// The reference deletes and rebuilds the whole scene, reloading the OBJ and PNG files, for every frame (World.cpp:59-66).
// The compiled-in scene does not depend on `time` (only the "#else" scene of World.cpp:34-40 does), so the adapter keeps
// the scene it has: same pixels, no per-frame rebuild.
void World::CreateWithNewTime(float) {}
void World::SyncCamera() {
    float cam[12];
    camera.Pack(cam);
    check(rt_multi_set_camera(devices, cam), "rt_multi_set_camera");
}
glm::u8vec3 World::CalculateColorForScreenPosition(int x, int y) {
    SyncCamera();
    rt_engine* engine = Engine();
    check(rt_engine_set_window(engine, x, y, x + 1, y + 1), "rt_engine_set_window");
    check(rt_engine_render(engine, frameSeed, 0, 1, nullptr, nullptr), "rt_engine_render");
    uint8_t p[4];                        // stored [A,B,G,R] (FrameManager.cpp:91-94)
    check(rt_engine_read_rect(engine, x, y, x + 1, y + 1, p), "rt_engine_read_rect");
    check(rt_engine_set_window(engine, 0, 0, WINDOW_WIDTH, WINDOW_HEIGHT), "rt_engine_set_window");
    return glm::u8vec3(p[3], p[2], p[1]);
}
glm::vec3 World::GetRayColor(const Ray& ray, int bounceAmount) {
    const float o[3] = {ray.pos.x, ray.pos.y, ray.pos.z}, d[3] = {ray.direction.x, ray.direction.y, ray.direction.z};
    const uint32_t key = frameSeed;
    float rgb[3];
    check(rt_engine_ray_color(Engine(), 1, o, d, bounceAmount, &key, rgb), "rt_engine_ray_color");
    return {rgb[0], rgb[1], rgb[2]};
}

void FrameManager::StartNewFrame() {
    TerminateAllThreads();                                    // FrameManager.cpp:9: the frame in flight is abandoned
    time += 1.0f;
    world.CreateWithNewTime(time);
    check(rt_multi_clear_frame(world.Devices()), "rt_multi_clear_frame");   // FrameManager.cpp:13
    world.SyncCamera();
    world.frameSeed++;
    check(rt_multi_render(world.Devices(), world.frameSeed), "rt_multi_render");
    state = State::Rendering;
#if LOG_BENCHMARK
    frameStartTime = std::chrono::system_clock::now();
#endif
}
// FrameManager.cpp:31-38: state = Quit, wait for the workers to notice.  Here: the render in flight stops taking rays,
// and the wait covers only the rays already on their way.
void FrameManager::TerminateAllThreads() {
    if (state == State::Rendering) check(rt_multi_cancel(world.Devices()), "rt_multi_cancel");
    check(rt_multi_sync(world.Devices()), "rt_multi_sync");
}
bool FrameManager::NeedUpdatingTexture() {
    if (state == State::AllComplete || state == State::Idle) return false;
    if (rt_multi_poll(world.Devices()) == 1) FinishFrame();
    return true;
}
void FrameManager::FinishFrame() {
    state = State::AllComplete;
#if LOG_BENCHMARK
    std::chrono::duration<double> dur = std::chrono::system_clock::now() - frameStartTime;
    std::cout << "Frame calculating time: " << dur.count() << std::endl;
#endif
}
// FrameManager.cpp:60-71 pauses the workers and hands out the buffer they write into, finished or not.  Here the frame
// buffer lives on the GPU; what is handed out is the engine's page-locked host copy of it: the complete frame once the
// render has finished, a snapshot of the batches finished so far while it is still running (nothing is paused).
void* FrameManager::GetTexturePixelsToPresent() {
    const uint8_t* pixels = nullptr;
    const int finished = state != State::Rendering || rt_multi_poll(world.Devices()) == 1;
    check(rt_multi_map_frame(world.Devices(), finished, &pixels), "rt_multi_map_frame");
    return const_cast<uint8_t*>(pixels);
}
void FrameManager::ContinueWorkingOnImage() {}   // FrameManager.cpp:73-76: nothing was paused
