// Adapter header with the reference's name (FrameManager.h:12-37): same public interface, so that the reference's front
// end (App.h / App.cpp) compiles against it unchanged.  The seven worker threads are replaced by one asynchronous
// rt_engine_render on the GPU.
#pragma once
#include <chrono>
#include <cstdint>

#include "World.h"

class FrameManager {
public:
    void StartNewFrame();               // FrameManager.cpp:7-30
    void TerminateAllThreads();         // :31-38  — cancels the render in flight and waits for the rays already on their way
    bool NeedUpdatingTexture();         // :40-48
    void FinishFrame();                 // :50-58  — prints "Frame calculating time" under LOG_BENCHMARK
    void* GetTexturePixelsToPresent();  // :60-71  — W*H*4 bytes [A,B,G,R], row-major, top row first, pitch W*4; page-locked
                                        //            host memory owned by the engine; a frame in progress shows its finished part
    void ContinueWorkingOnImage();      // :73-76
    World& GetWorld() { return world; }

private:
    enum class State { Idle, Rendering, AllComplete };
    float time = 0;
    World world;
    std::chrono::system_clock::time_point frameStartTime;
    State state = State::Idle;
};
