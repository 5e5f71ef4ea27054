/*
 * hrt_host.h -- host side ABOVE the C ABI (C++ implementation in
 * ilgpu_raytracing_amd/csrc/hrt_host.cpp), exported from libhip_raytrace.so so that
 * hosts without the reference's C# runtime (Python tests, bench.py, the C++ harness) can
 * build scenes and cameras exactly as the reference's host classes do.  A C# host keeps
 * using its own Engine/Scene.cs / Engine/Camera.cs and only calls hip_raytrace.h.
 *
 *   hrth_scene_new/free/clear          Scene ctor / Dispose           Scene.cs:60-64,676-693
 *   hrth_scene_build_default           Scene.BuildDefaultScene        Scene.cs:83-142
 *   hrth_scene_add_texture             AddCheckerTexture's append     Scene.cs:98-109
 *   hrth_scene_add_sphere              Scene.AddSphere                Scene.cs:315-321
 *   hrth_scene_build_sphere_instance   Scene.BuildSphereInstance + append to _hInstances
 *                                                                     Scene.cs:323-356,127-137
 *   hrth_scene_load_mesh_instance      Scene.LoadObjInstance after MeshLoaderOBJ.Load
 *                                      (textures arrive as BGRA, as TextureSrc does)
 *                                                                     Scene.cs:151-256
 *   hrth_mesh_load_obj / _get / _free  MeshLoaderOBJ.Load -> MeshHost (OBJ + MTL + textures)
 *                                                                     MeshLoaderOBJ.cs:67-277,339-443
 *   hrth_image_load / _free            LoadTextureBGRA (TGA raw/RLE; PNG <= 8 bits per sample; BMP) MeshLoaderOBJ.cs:456-593
 *   hrth_scene_load_obj_instance       Scene.LoadObjInstance(objPath, objectToWorld, uniformScale)
 *                                                                     Scene.cs:144-256
 *   hrth_scene_rebuild_tlas            Scene.RebuildTLAS              Scene.cs:358-368
 *   hrth_scene_get_desc                Scene.GetDeviceViews' 15 arrays (host side)
 *                                                                     Scene.cs:281-313
 *   hrth_camera_create                 Camera.CreateCamera            Camera.cs:19-47
 *   hrth_camera_lookat                 Camera(origin,lookAt,up,vfov,aspect,focus)
 *                                                                     Camera.cs:100-119
 *   hrth_camera_translate              Camera.Translate               Camera.cs:121-126
 *   hrth_camera_bake                   RTRenderer.BakeCameraDerived   RTRenderer.cs:241-263
 *   hrth_sun_dir                       sun direction from (azimuth, elevation)
 *                                                                     RTRenderer.cs:174-178
 * Functions returning int give the new index (>= 0) or -1 on invalid arguments
 * (ArgumentException analogue).  The file loaders also return HRTH_ERR_NOT_FOUND
 * (FileNotFoundException, Scene.cs:146-147) and HRTH_ERR_FORMAT (FormatException /
 * InvalidDataException / EndOfStreamException raised while parsing); hrth_last_error() then
 * holds the message (thread-local).  Pointers returned through hrth_scene_get_desc stay
 * valid until the next mutating call on that scene.
 */
#ifndef HRT_HOST_H
#define HRT_HOST_H
#include "hrt_types.h"
#ifdef __cplusplus
extern "C" {
#endif

enum { HRTH_ERR_ARGUMENT = -1, HRTH_ERR_NOT_FOUND = -2, HRTH_ERR_FORMAT = -3 };

/* MeshHost (MeshLoaderOBJ.cs:21-32) as borrowed arrays; textures are BGRA, 4 bytes per texel, concatenated in index order */
typedef struct hrth_mesh_desc {
    const hrt_float3* positions;       int n_positions;
    const hrt_mesh_tri* triangles;     int n_triangles;
    const hrt_float2* texcoords;       int n_texcoords;
    const hrt_mesh_tri_uv* tri_uvs;                               /* n_triangles entries */
    const int* tri_material_index;     int n_tri_material_index;
    const hrt_material* materials;     int n_materials;
    const int* tex_w; const int* tex_h; const uint8_t* tex_bgra; int n_textures;
} hrth_mesh_desc;

void* hrth_scene_new(void);
void  hrth_scene_free(void* scene);
void  hrth_scene_clear(void* scene);
void  hrth_scene_build_default(void* scene);
int   hrth_scene_add_texture(void* scene, int w, int h, const hrt_rgba32* texels);
int   hrth_scene_add_sphere(void* scene, const hrt_sphere* s);
int   hrth_scene_build_sphere_instance(void* scene, const int* sphere_ids, int n, const hrt_affine3x4* objectToWorld);
int   hrth_scene_load_mesh_instance(void* scene,
                                    const hrt_float3* positions, int n_positions,
                                    const hrt_mesh_tri* triangles, int n_triangles,
                                    const hrt_float2* texcoords, int n_texcoords,
                                    const hrt_mesh_tri_uv* tri_uvs,
                                    const int* tri_material_index, int n_tri_material_index,
                                    const hrt_material* materials, int n_materials,
                                    const int* tex_w, const int* tex_h, const uint8_t* tex_bgra, int n_textures,
                                    const hrt_affine3x4* objectToWorld);
int   hrth_mesh_load_obj(const char* path, float scale, int flipWinding, void** mesh_out);
int   hrth_mesh_get(void* mesh, hrth_mesh_desc* out);
const char* hrth_mesh_material_name(void* mesh, int material_index);
const char* hrth_mesh_texture_path(void* mesh, int texture_index);
void  hrth_mesh_free(void* mesh);
int   hrth_image_load(const char* path, int* width, int* height, uint8_t** bgra_out);   /* release with hrth_image_free */
void  hrth_image_free(uint8_t* bgra);
int   hrth_scene_load_obj_instance(void* scene, const char* objPath, const hrt_affine3x4* objectToWorld, float uniformScale);
const char* hrth_last_error(void);
void  hrth_scene_rebuild_tlas(void* scene);
/* moves an instance of the host scene: objectToWorld, worldToObject, uniformScale and the world bounds as the
 * instance builders derive them (Scene.cs:395-402, 236-252); follow with hrth_scene_rebuild_tlas (host rebuild)
 * or mirror the move on the device with hrt_scene_update_instances.  0 ok, -1 bad id. */
int   hrth_scene_set_instance_transform(void* scene, int instance_id, const hrt_affine3x4* objectToWorld);
void  hrth_scene_get_desc(void* scene, hrt_scene_desc* out);

void  hrth_camera_create(int width, int height, float fovDegrees, hrt_camera* out);
void  hrth_camera_lookat(const float origin[3], const float lookAt[3], const float up[3],
                         float vfovDegrees, float aspect, float focusDist, hrt_camera* out);
void  hrth_camera_translate(hrt_camera* cam, const float delta[3]);
void  hrth_camera_bake(hrt_camera* cam, int pixelW, int pixelH);
void  hrth_sun_dir(float azimuth, float elevation, float out[3]);

#ifdef __cplusplus
}
#endif
#endif
