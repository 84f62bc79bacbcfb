// Headless counterpart of the reference's main.cpp (Main::Start + the frame loop, main.cpp:95-141,230-247) and of its
// offline render (`R` key: Renderer::Start / NextFrame, Renderer.cpp:19-193), on the GL-free host classes of
// raytracing_amd/csrc/host/rt_host.hpp over the C ABI of include/rt_core.h.  No window, no GL: what the reference
// draws to the screen is read back with rt_read_image (= glGetTexImage) and the frames go to PPM files.
//
//   headless_main <models_root> [scene=2] [width=800] [height=600] [spp=100] [bounces=10] [frames=1] [out_dir=Renders] [gpus=1]
//
// gpus > 1: the same loop on rt_host::MultiGPU — one context and one host thread per GPU, every frame split into sample
// ranges, one RCCL reduce of the colour image onto GPU 0 (no reference counterpart: the reference is single-GPU).
//
// The defaults are the reference's own hard-coded ones (main.h:35, main.cpp:241-244).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/rt_core.h"
#include "../raytracing_amd/csrc/host/rt_host.hpp"

using namespace rt_host;

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s <models_root> [scene width height spp bounces frames out_dir gpus]\n", argv[0]); return 2; }
    const std::string models = argv[1];
    const int scene = argc > 2 ? std::atoi(argv[2]) : 2;
    const int width = argc > 3 ? std::atoi(argv[3]) : 800, height = argc > 4 ? std::atoi(argv[4]) : 600;
    const int spp = argc > 5 ? std::atoi(argv[5]) : 100, bounces = argc > 6 ? std::atoi(argv[6]) : 10;
    const int frames = argc > 7 ? std::atoi(argv[7]) : 1;
    const std::string outDir = argc > 8 ? argv[8] : "Renders";
    const bool devList = argc > 9 && std::string(argv[9]).find(',') != std::string::npos;
    const int gpus = argc > 9 ? std::atoi(argv[9]) : 1;
    if (gpus > 1 || devList) {
        // (a comma-separated list of ordinals instead of a count — "0,0" — puts several contexts on one GPU: a rehearsal)
        std::vector<int> devs;
        const std::string a = argv[9];
        if (devList) { size_t p = 0; while (p <= a.size()) { size_t q = a.find(',', p); if (q == std::string::npos) q = a.size(); devs.push_back(std::atoi(a.substr(p, q - p).c_str())); p = q + 1; } }
        else for (int d = 0; d < gpus; d++) devs.push_back(d);
        try {
            MultiGPU multi(devs);
            Camera camera(width, height, vec3(0.0f, 3.5f, 14.0f));
            Scene::modelsRoot = models;
            Scene::currSceneIndex = 0;
            Scene::LoadScene(multi.shader(0), scene);
            multi.ReplicateScene();
            multi.setInt("maxBounces", bounces);
            multi.resize((uint32_t)width, (uint32_t)height);
            if (frames > 1) Renderer::Start(width, height, spp, bounces, outDir);
            for (int f = 0; f < frames; f++) {
                const auto t0 = std::chrono::steady_clock::now();
                multi.SetFrameUniforms(&camera, vec2{(float)width, (float)height});
                multi.RenderFrame((uint32_t)spp);
                double rays = 0;
                for (size_t i = 0; i < multi.size(); i++) { rt_stats st; rt_get_stats(multi.shader(i)->ID, &st); rays += (double)(st.rays_primary + st.rays_bounce + st.rays_shadow); }
                const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                std::printf("frame %d: %.1f ms, %.0f Mrays/s on %zu contexts%s\n", f, s * 1e3, rays / s / 1e6, multi.size(), multi.usedRccl() ? " (RCCL reduce)" : "");
                if (Renderer::isRenderering()) Renderer::NextFrame(multi.shader(0), (uint32_t)width, (uint32_t)height, &camera);
            }
            std::vector<float> rgba((size_t)width * height * 4);
            if (rt_read_image(multi.shader(0)->ID, RT_IMAGE_COLOR, rgba.data()) != RT_OK) return 1;
            double sum = 0;
            for (size_t i = 0; i < rgba.size(); i += 4) sum += rgba[i] + rgba[i + 1] + rgba[i + 2];
            std::printf("mean radiance of the last frame: %.6f\n", sum / (3.0 * width * height));
            Scene::UnloadScene();
        } catch (const std::exception& e) { std::fprintf(stderr, "multi-GPU render failed: %s\n", e.what()); return 1; }
        return 0;
    }

    // Main::Start (main.cpp:230-247)
    ComputeShader computeShader("shader.comp");
    if (!computeShader.ID) { std::fprintf(stderr, "no HIP device: %s\n", rt_last_error(nullptr)); return 1; }
    Camera camera(width, height, vec3(0.0f, 3.5f, 14.0f));
    Scene::modelsRoot = models;
    Scene::LoadScene(&computeShader, scene);
    computeShader.Activate();
    computeShader.setInt("samplesPerPixel", spp);
    computeShader.setInt("maxBounces", bounces);
    if (rt_resize(computeShader.ID, (uint32_t)width, (uint32_t)height) != RT_OK) {      // Texture quad/albedo/normal/depth, main.cpp:85-88
        std::fprintf(stderr, "rt_resize: %s\n", rt_last_error(computeShader.ID));
        return 1;
    }
    std::printf("scene %d: BVH %.3f s, upload %.3f s\n", scene, Model::lastBuildSeconds, Model::lastUploadSeconds);

    if (frames > 1) Renderer::Start(width, height, spp, bounces, outDir);               // the `R` key, main.cpp:317-320
    std::vector<float> rgba((size_t)width * height * 4);
    for (int f = 0; f < frames; f++) {
        const auto t0 = std::chrono::steady_clock::now();
        SetFrameUniforms(&computeShader, &camera, vec2{(float)width, (float)height});   // main.cpp:108-124
        if (rt_render_frame(computeShader.ID) != RT_OK) {                              // the tile loop, main.cpp:126-141
            std::fprintf(stderr, "rt_render_frame: %s\n", rt_last_error(computeShader.ID));
            return 1;
        }
        rt_stats st;
        rt_get_stats(computeShader.ID, &st);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("frame %d: %.1f ms, %.0f Mrays/s\n", f, s * 1e3, (double)(st.rays_primary + st.rays_bounce + st.rays_shadow) / s / 1e6);
        if (Renderer::isRenderering()) Renderer::NextFrame(&computeShader, (uint32_t)width, (uint32_t)height, &camera);   // main.cpp:198
    }
    if (rt_read_image(computeShader.ID, RT_IMAGE_COLOR, rgba.data()) != RT_OK) return 1;   // glGetTexImage, main.cpp:157
    double sum = 0;
    for (size_t i = 0; i < rgba.size(); i += 4) sum += rgba[i] + rgba[i + 1] + rgba[i + 2];
    std::printf("mean radiance of the last frame: %.6f\n", sum / (3.0 * width * height));
    Scene::UnloadScene();
    computeShader.Delete();
    return 0;
}
