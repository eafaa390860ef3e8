// Drives the adapter the way the reference's main loop does (App.cpp:71-107: StartNewFrame, then per tick
// NeedUpdatingTexture -> GetTexturePixelsToPresent -> present, ContinueWorkingOnImage), without a window:
//   compat_demo <frames> <out.raw>     writes the last frame's W*H*4 bytes
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "FrameManager.h"

int main(int argc, char** argv) {
    const int frames = argc > 1 ? atoi(argv[1]) : 1;
    const char* out = argc > 2 ? argv[2] : nullptr;
    static FrameManager frameManager;   // the front end holds it by value (App.h:27); static: the pixel array is large
    Camera& camera = frameManager.GetWorld().GetWorldCamera();
    void* pixels = nullptr;
    for (int f = 0; f < frames; f++) {
        if (f > 0) {   // a little mouse motion and a step forward, as App.cpp:96-98,114-121 apply them
            camera.SetForwardVector(camera.GetForwardVector() - 0.001f * (3.0f * camera.GetUDirection() + 1.0f * camera.GetVDirection()));
            glm::vec3 dir = camera.GetForwardVector();
            dir.y = 0;
            camera.pos += dir * 0.3f;
        }
        frameManager.StartNewFrame();
        int ticks = 0;
        while (frameManager.NeedUpdatingTexture()) {
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
            ticks++;
            frameManager.ContinueWorkingOnImage();
        }
        pixels = frameManager.GetTexturePixelsToPresent();
        printf("frame %d done after %d ticks\n", f, ticks);
    }
    frameManager.TerminateAllThreads();
    if (out && pixels) {
        FILE* fh = fopen(out, "wb");
        if (!fh) { perror(out); return 1; }
        fwrite(pixels, 1, (size_t)WINDOW_WIDTH * WINDOW_HEIGHT * 4, fh);
        fclose(fh);
    }
    return 0;
}
