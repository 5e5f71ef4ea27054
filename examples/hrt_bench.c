/* hrt_bench.c -- the hot path from a plain C host: builds BASELINE config 2's kind of scene (sphere instances) with the
 * host builders of include/hrt_host.h, uploads it through the C ABI of include/hip_raytrace.h and times K frames.
 *   gcc -std=c11 -O2 -Iinclude examples/hrt_bench.c -o hrt_bench -Lilgpu_raytracing_amd/csrc -lhip_raytrace -Wl,-rpath,$PWD/ilgpu_raytracing_amd/csrc -lm
 *   ./hrt_bench [width height spp frames]
 * No Python, no torch: what a C / C++ / C# (P/Invoke) host links against.  Needs an MI355X; without one hrt_create fails. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hip_raytrace.h"
#include "hrt_host.h"

static hrt_sphere sphere(float x, float y, float z, float r, float cr, float cg, float cb, int shading, float ior)
{
    hrt_sphere s;
    memset(&s, 0, sizeof s);
    s.center.X = x; s.center.Y = y; s.center.Z = z; s.radius = r;
    s.albedo.X = cr; s.albedo.Y = cg; s.albedo.Z = cb;
    s.material.Kd = s.albedo; s.material.DiffuseTexIndex = -1; s.material.AlphaTexIndex = -1; s.material.IOR = 1.f; s.material.AlphaCutoff = 0.5f;
    s.shading = shading; s.ior = ior;
    return s;
}

#define CHECK(ctx, call) do { int rc_ = (call); if (rc_ != HRT_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, hrt_last_error(ctx)); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int width = argc > 1 ? atoi(argv[1]) : 1920, height = argc > 2 ? atoi(argv[2]) : 1080;
    const int spp = argc > 3 ? atoi(argv[3]) : 4, frames = argc > 4 ? atoi(argv[4]) : 20;

    /* scene: all spheres first, one instance each (Scene.BuildDefaultScene's order) */
    void* scene = hrth_scene_new();
    const hrt_sphere sph[] = {
        sphere(0.f, -1000.f, 0.f, 1000.f, 0.75f, 0.75f, 0.75f, 0, 1.f),          /* floor */
        sphere(0.f, 1.2f, -1002.f, 1000.f, 0.75f, 0.75f, 0.75f, 0, 1.f),         /* back wall */
        sphere(-1002.f, 1.2f, 0.f, 1000.f, 0.75f, 0.25f, 0.25f, 0, 1.f),         /* left, red */
        sphere(1002.f, 1.2f, 0.f, 1000.f, 0.25f, 0.75f, 0.25f, 0, 1.f),          /* right, green */
        sphere(-0.9f, 0.6f, -0.4f, 0.6f, 0.8f, 0.8f, 0.8f, 0, 1.f),
        sphere(0.9f, 0.6f, 0.2f, 0.6f, 0.95f, 0.95f, 0.95f, 1, 1.f),             /* mirror */
        sphere(0.0f, 0.5f, 1.1f, 0.5f, 1.f, 1.f, 1.f, 2, 1.5f),                  /* glass */
    };
    const int nSph = (int)(sizeof sph / sizeof sph[0]);
    int ids[16];
    for (int i = 0; i < nSph; i++) ids[i] = hrth_scene_add_sphere(scene, &sph[i]);
    const hrt_affine3x4 ident = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (int i = 0; i < nSph; i++)
        if (hrth_scene_build_sphere_instance(scene, &ids[i], 1, &ident) < 0) { fprintf(stderr, "build_sphere_instance failed\n"); return 1; }
    hrth_scene_rebuild_tlas(scene);
    hrt_scene_desc desc;
    hrth_scene_get_desc(scene, &desc);

    hrt_ctx* ctx = NULL;
    const int dev = 0;
    CHECK(NULL, hrt_create(&dev, 1, &ctx));
    CHECK(ctx, hrt_scene_upload(ctx, &desc));

    hrt_frame_params p;
    memset(&p, 0, sizeof p);
    p.width = width; p.height = height; p.frame = 0;
    const float origin[3] = {0.f, 1.5f, 5.5f}, lookAt[3] = {0.f, 1.2f, 0.f}, up[3] = {0.f, 1.f, 0.f};
    hrth_camera_lookat(origin, lookAt, up, 60.f, (float)width / (float)height, 1.f, &p.cam);
    hrth_camera_bake(&p.cam, width, height);
    p.prevCam = p.cam;
    float sun[3];
    hrth_sun_dir(1.5707964f, 0.6f, sun);
    p.dirLightDir.X = sun[0]; p.dirLightDir.Y = sun[1]; p.dirLightDir.Z = sun[2];
    p.dirLightRadiance.X = p.dirLightRadiance.Y = p.dirLightRadiance.Z = 10.f;
    p.skyTintTop.X = 0.5f; p.skyTintTop.Y = 0.7f; p.skyTintTop.Z = 1.f;
    p.skyTintBottom.X = p.skyTintBottom.Y = p.skyTintBottom.Z = 1.f;
    p.spp = spp; p.maxDepth = 3;

    /* one counting frame (rays actually traced), then K production frames enqueued back to back */
    hrt_render_opts opts;
    memset(&opts, 0, sizeof opts);
    opts.strip_n = 1;
    opts.flags = HRT_FLAG_COUNTERS;
    hrt_stats st;
    CHECK(ctx, hrt_render_frame(ctx, &p, &opts, NULL, &st));
    const double rays = (double)st.k[0].rays_closest + (double)st.k[1].rays_closest + (double)st.k[1].rays_shadow;
    opts.flags = HRT_FLAG_NO_SYNC;
    for (int f = 0; f < frames; f++) CHECK(ctx, hrt_render_frame(ctx, &p, &opts, NULL, NULL));
    CHECK(ctx, hrt_synchronize(ctx, &st));
    const double ms = (st.kernel_ms[0] + st.kernel_ms[1]) / (st.frames > 0 ? st.frames : 1);

    /* one blocking frame with the colour image gathered to the host */
    int* color = (int*)malloc((size_t)width * height * sizeof(int));
    hrt_outputs out;
    memset(&out, 0, sizeof out);
    out.color = color;
    opts.flags = 0;
    CHECK(ctx, hrt_render_frame(ctx, &p, &opts, &out, NULL));
    unsigned long long sum = 0;
    for (long long i = 0; i < (long long)width * height; i++) sum += (unsigned)color[i] & 0xFFFFFFu;

    /* BvhManager.BuildOrRefit(scene, ForceRebuild) on the device: nothing moved, the TLAS is rebuilt there; same picture */
    hrt_bvh_update_stats us;
    CHECK(ctx, hrt_scene_update_instances(ctx, NULL, 0, NULL, HRT_REBUILD_FORCE_REBUILD, &us));
    CHECK(ctx, hrt_reset_history(ctx));
    int* color2 = (int*)malloc((size_t)width * height * sizeof(int));
    out.color = color2;
    CHECK(ctx, hrt_render_frame(ctx, &p, &opts, &out, NULL));
    const int same = memcmp(color, color2, (size_t)width * height * sizeof(int)) == 0;
    printf("{\"host\": \"C\", \"width\": %d, \"height\": %d, \"spp\": %d, \"frames\": %d, \"rays_per_frame\": %.0f, \"kernel_ms_per_frame\": %.4f, "
           "\"mrays_per_s\": %.1f, \"color_checksum\": %llu, \"device_tlas_nodes\": %d, \"device_tlas_ms\": %.3f, \"device_tlas_same_picture\": %s}\n",
           width, height, spp, st.frames, rays, ms, rays / ms / 1e3, sum, us.tlas_nodes, us.device_ms, same ? "true" : "false");
    free(color2);
    free(color);
    hrt_destroy(ctx);
    hrth_scene_free(scene);
    return 0;
}
