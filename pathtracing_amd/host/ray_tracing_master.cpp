// RayTracingMaster mirror + the C functions of ptmaster.h.
#include "ray_tracing_master.hpp"

#include <cmath>
#include <cstring>
#include <new>

#include "ptmaster.h"

namespace ptm {

RayTracingMaster::RayTracingMaster(int device)
{
    pt_context *c = nullptr;
    int rc = pt_create(device, &c);
    if (rc != PT_OK) { error_ = std::string("pt_create: ") + pt_last_error(nullptr); return; }
    ctx_ = c;
}

RayTracingMaster::~RayTracingMaster()
{
    // OnDisable (RayTracingMaster.cs:154-161): release every GPU buffer
    if (ctx_) pt_destroy(ctx_);
}

int RayTracingMaster::check(int rc, const char *what)
{
    if (rc != PT_OK) error_ = std::string(what) + ": " + pt_last_error(ctx_);
    return rc;
}

int RayTracingMaster::OnEnable(const Scene &scene, int sky_w, int sky_h)
{
    if (!ctx_) return PT_ERR_INVALID;
    currentSample_ = 0;                                   // :145
    frame_ = 0;
    int rc = SetUpScene(scene);                           // :146
    if (rc) return rc;
    // RayTracingObject.OnEnable -> RegisterObject for every mesh of the scene (:296-301)
    meshObjects_ = scene.objects;
    vertices_ = scene.vertices;
    indices_ = scene.indices;
    meshObjectsNeedRebuilding_ = true;
    geometryUploaded_ = false;
    if (sky_w > 0 && sky_h > 0) {                         // SkyboxTexture (:25, :457)
        std::vector<float> sky((size_t)sky_w * sky_h * 4);
        sky_procedural(sky_w, sky_h, sky.data());
        if ((rc = check(pt_set_sky(ctx_, sky.data(), sky_w, sky_h), "pt_set_sky"))) return rc;
    } else {
        if ((rc = check(pt_set_sky(ctx_, nullptr, 0, 0), "pt_set_sky"))) return rc;
    }
    return PT_OK;
}

int RayTracingMaster::SetUpScene(const Scene &scene)
{
    // The sphere list was produced by ptm::sphere_field (the procedure of :317-359);
    // :362-368 upload it, skipping the buffer when the list is empty.
    return check(pt_set_spheres(ctx_, scene.spheres.data(), (int)scene.spheres.size()), "pt_set_spheres");
}

int RayTracingMaster::UpdateObject(int object, const float local_to_world[16])
{
    if (object < 0 || object >= (int)meshObjects_.size()) { error_ = "UpdateObject: no such object"; return PT_ERR_INVALID; }
    std::memcpy(meshObjects_[object].local_to_world, local_to_world, 16 * sizeof(float));
    meshObjectsNeedRebuilding_ = true;                    // :303-307
    return PT_OK;
}

int RayTracingMaster::RebuildMeshObjectBuffers()
{
    if (!meshObjectsNeedRebuilding_) return PT_OK;        // :373-376
    meshObjectsNeedRebuilding_ = false;
    int rc;
    if (!geometryUploaded_) {
        // :410-412 - all three buffers
        rc = check(pt_set_meshes(ctx_, meshObjects_.data(), (int)meshObjects_.size(), vertices_.data(),
                                 (int)(vertices_.size() / 3), indices_.data(), (int)indices_.size()), "pt_set_meshes");
        geometryUploaded_ = rc == PT_OK;
    } else {
        // the reference re-uploads everything; only the matrices can have changed
        rc = check(pt_update_transforms(ctx_, meshObjects_.data(), (int)meshObjects_.size()), "pt_update_transforms");
    }
    return rc;
}

void RayTracingMaster::currentCamera(float e[3], float t[3]) const
{
    for (int k = 0; k < 3; k++) { e[k] = eye[k]; t[k] = target[k]; }
    if (eyeOffset != 0.0f) {
        // shift eye and target along the camera's right axis (parallel stereo axes)
        double f[3] = {t[0] - e[0], t[1] - e[1], t[2] - e[2]};
        double r[3] = {f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0]};
        double l = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        for (int k = 0; k < 3; k++) { float d = (float)(r[k] / l) * eyeOffset; e[k] += d; t[k] += d; }
    }
}

void RayTracingMaster::FillParams(PtFrameParams &p) const
{
    std::memset(&p, 0, sizeof p);
    // _CameraToWorld / _CameraInverseProjection (:489-490; per-eye variants :463-467)
    float e[3], t[3];
    currentCamera(e, t);
    float aspect = RenderHight > 0 ? (float)RenderWidth / (float)RenderHight : 1.0f;
    camera_look_at(e, t, up, fieldOfView, aspect, nearClip, farClip, p.camera_to_world, p.camera_inverse_projection);
    p.pixel_offset[0] = 0.0f; p.pixel_offset[1] = 0.0f;   // :529 (unused by the kernel)
    p.seed = frame_seed(frame_);                          // :530, deterministic stand-in for Random.value
    p.samples_per_pixel = renderMode < 10 ? samplesPerPixel : 1;   // :531
    p.depth = depth ? 1 : 0;                              // :532
    p.directional_light[1] = -1.0f; p.directional_light[3] = 1.0f; // :533-534 (unused by the kernel)
    p.render_mode = renderMode;                           // :735
    p.max_bounces = maxBounces;
    p.accumulate = (depth || renderMode != 1) ? PT_ACCUM_NONE : accumulate;
    p.sample_index = depth ? 0u : currentSample_;         // :805
    p.accumulation_frames = (int32_t)sampleFrames;        // :738
    p.divisions = divisions;                              // :736
    p.counter = divisions > 0 ? counter_ % divisions : 0; // :737 (the increment happens in Render)
    p.left_eye = eyeOffset < 0.0f ? 1 : 0;                // :460-462 (_leftEye; read by renderMode 4 only)
}

int RayTracingMaster::SetShaderParameters(PtFrameParams &p)
{
    FillParams(p);
    // :492-495 - the history lists as they stand, i.e. the cameras of earlier frames.  An empty list changes
    // nothing (SetMatrixArray rejects it); a shorter one leaves the tail of the shader's arrays as it was.
    if (!oldCameras_.empty()) {
        int rc = check(pt_set_camera_history(ctx_, oldCameras_.data(), (int)oldCameras_.size()), "pt_set_camera_history");
        if (rc) return rc;
    }
    // :513-518 - outside the partial-frame modes the lists restart every frame; inside them at the start of a cycle
    if (renderMode < 5 || renderMode > 8 || (divisions > 0 && counter_ % divisions == 0)) oldCameras_.clear();
    // :520-526 - remember this frame's camera
    if (renderMode < 11 && renderMode != 1) {
        float e[3], t[3];
        currentCamera(e, t);
        float aspect = RenderHight > 0 ? (float)RenderWidth / (float)RenderHight : 1.0f;
        PtCameraHistory h;
        camera_history_entry(e, t, up, fieldOfView, aspect, nearClip, farClip, &h);
        oldCameras_.push_back(h);
    }
    return PT_OK;
}

int RayTracingMaster::InitRenderTexture()
{
    if (targetW_ == RenderWidth && targetH_ == RenderHight) return PT_OK;   // :545
    if (RenderWidth == 0 || RenderHight == 0) return PT_OK;                  // :554-556
    int rc = check(pt_resize(ctx_, RenderWidth, RenderHight), "pt_resize");  // :559-576
    if (rc == PT_OK) { targetW_ = RenderWidth; targetH_ = RenderHight; }
    return rc;
}

int RayTracingMaster::Render()
{
    if (!ctx_) return PT_ERR_INVALID;
    int rc;
    if ((rc = RebuildMeshObjectBuffers())) return rc;     // :714
    PtFrameParams p;
    if ((rc = SetShaderParameters(p))) return rc;         // :716
    if ((rc = InitRenderTexture())) return rc;            // :718
    if (targetW_ == 0 || targetH_ == 0) return PT_OK;     // :721
    if ((rc = check(pt_render(ctx_, &p), "pt_render"))) return rc;   // :735-747
    frame_++;
    counter_++;                                           // :737
    // :885-888 - the sample counter saturates at sampleFrames, after which the blend is a moving average
    if (currentSample_ < sampleFrames) currentSample_++; else currentSample_ = sampleFrames;
    return PT_OK;
}

}  // namespace ptm

// ------------------------------------------------------------------ C API

using namespace ptm;

struct ptm_scene { Scene *s; };
struct ptm_master { RayTracingMaster *m; std::string create_error; };

extern "C" {

int ptm_sphere_field(int seed, uint32_t spheres_max, float rmin, float rmax, float placement_radius, PtSphere *out, int capacity)
{
    std::vector<PtSphere> v;
    int n = sphere_field(seed, spheres_max, rmin, rmax, placement_radius, v);
    if (n > capacity) n = capacity;
    if (out && n > 0) std::memcpy(out, v.data(), sizeof(PtSphere) * (size_t)n);
    return n;
}

void ptm_sky_procedural(int w, int h, float *rgba) { sky_procedural(w, h, rgba); }

void ptm_camera_look_at(const float eye[3], const float target[3], const float up[3], float fov, float aspect,
                        float z_near, float z_far, float c2w[16], float ip[16])
{
    camera_look_at(eye, target, up, fov, aspect, z_near, z_far, c2w, ip);
}

void ptm_camera_history_entry(const float eye[3], const float target[3], const float up[3], float fov, float aspect,
                              float z_near, float z_far, PtCameraHistory *out)
{
    camera_history_entry(eye, target, up, fov, aspect, z_near, z_far, out);
}

float ptm_frame_seed(uint32_t k) { return frame_seed(k); }

ptm_scene *ptm_scene_create(const char *name)
{
    Scene *s = Scene::create(name ? name : "");
    if (!s) return nullptr;
    return new ptm_scene{s};
}
void ptm_scene_destroy(ptm_scene *s) { if (s) { delete s->s; delete s; } }
void ptm_scene_counts(const ptm_scene *s, int32_t c[4])
{
    c[0] = (int32_t)s->s->spheres.size(); c[1] = (int32_t)s->s->objects.size();
    c[2] = (int32_t)(s->s->vertices.size() / 3); c[3] = (int32_t)s->s->indices.size();
}
void ptm_scene_copy(const ptm_scene *s, PtSphere *spheres, PtMeshObject *objects, float *vertices, int32_t *indices)
{
    const Scene &sc = *s->s;
    if (spheres && !sc.spheres.empty()) std::memcpy(spheres, sc.spheres.data(), sc.spheres.size() * sizeof(PtSphere));
    if (objects && !sc.objects.empty()) std::memcpy(objects, sc.objects.data(), sc.objects.size() * sizeof(PtMeshObject));
    if (vertices && !sc.vertices.empty()) std::memcpy(vertices, sc.vertices.data(), sc.vertices.size() * sizeof(float));
    if (indices && !sc.indices.empty()) std::memcpy(indices, sc.indices.data(), sc.indices.size() * sizeof(int32_t));
}
void ptm_scene_camera(const ptm_scene *s, float eye[3], float target[3], float *fov)
{
    std::memcpy(eye, s->s->eye, 12); std::memcpy(target, s->s->target, 12); *fov = s->s->fov;
}
int ptm_scene_add_obj(ptm_scene *s, const char *path, const float m[16]) { return s->s->add_obj(path, m); }

int ptm_master_create(int device, ptm_master **out)
{
    if (!out) return PT_ERR_INVALID;
    ptm_master *h = new ptm_master{new RayTracingMaster(device), ""};
    *out = h;
    if (!h->m->ok()) return PT_ERR_HIP;   // handle stays valid so the caller can read the error, then destroy it
    return PT_OK;
}
void ptm_master_destroy(ptm_master *m) { if (m) { delete m->m; delete m; } }
const char *ptm_master_error(const ptm_master *m) { return m ? m->m->error().c_str() : "null master"; }
pt_context *ptm_master_context(ptm_master *m) { return m->m->context(); }
int ptm_master_load_scene(ptm_master *m, const ptm_scene *s, int sky_w, int sky_h)
{
    std::memcpy(m->m->eye, s->s->eye, 12); std::memcpy(m->m->target, s->s->target, 12); m->m->fieldOfView = s->s->fov;
    return m->m->OnEnable(*s->s, sky_w, sky_h);
}
int ptm_master_configure(ptm_master *m, int width, int height, int spp, int max_bounces, int depth, int accumulate, uint32_t sample_frames)
{
    RayTracingMaster &r = *m->m;
    r.RenderWidth = width; r.RenderHight = height; r.samplesPerPixel = spp; r.maxBounces = max_bounces;
    r.depth = depth != 0; r.accumulate = accumulate; r.sampleFrames = sample_frames;
    return PT_OK;
}
int ptm_master_set_render_mode(ptm_master *m, int render_mode) { m->m->renderMode = render_mode; return PT_OK; }
int ptm_master_set_divisions(ptm_master *m, int divisions) { m->m->divisions = divisions; return PT_OK; }
int ptm_master_set_camera(ptm_master *m, const float eye[3], const float target[3], float fov, float eye_offset)
{
    std::memcpy(m->m->eye, eye, 12); std::memcpy(m->m->target, target, 12);
    m->m->fieldOfView = fov; m->m->eyeOffset = eye_offset;
    return PT_OK;
}
int ptm_master_render(ptm_master *m) { return m->m->Render(); }
int ptm_master_peek_params(ptm_master *m, PtFrameParams *out) { m->m->FillParams(*out); return PT_OK; }
void ptm_master_reset_accumulation(ptm_master *m) { m->m->ResetAccumulation(); }
uint32_t ptm_master_current_sample(const ptm_master *m) { return m->m->currentSample(); }
uint32_t ptm_master_frame_index(const ptm_master *m) { return m->m->frameIndex(); }

}  // extern "C"
