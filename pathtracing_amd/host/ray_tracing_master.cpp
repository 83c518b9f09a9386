// RayTracingMaster mirror + the C functions of ptmaster.h.
#include "ray_tracing_master.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <new>
#include <utility>

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
    // RayTracingObject.OnEnable -> RegisterObject for every mesh object of the scene (:296-301).  Objects with the same
    // index range are instances of one mesh.
    rayTracingObjects_.clear();
    for (size_t o = 0; o < scene.objects.size(); o++) {
        const PtMeshObject &mo = scene.objects[o];
        int shared = -1;
        for (size_t q = 0; q < o && shared < 0; q++)
            if (scene.objects[q].indices_offset == mo.indices_offset && scene.objects[q].indices_count == mo.indices_count) shared = (int)q;
        if (shared >= 0) { RegisterInstance(shared, mo.local_to_world); continue; }
        int32_t lo = INT32_MAX, hi = -1;
        for (int32_t i = mo.indices_offset; i < mo.indices_offset + mo.indices_count; i++) { lo = std::min(lo, scene.indices[i]); hi = std::max(hi, scene.indices[i]); }
        std::vector<int32_t> idx((size_t)mo.indices_count);
        for (int32_t i = 0; i < mo.indices_count; i++) idx[(size_t)i] = scene.indices[mo.indices_offset + i] - (hi >= 0 ? lo : 0);
        const float *v = hi >= 0 ? &scene.vertices[(size_t)lo * 3] : nullptr;
        RegisterObject(v, hi >= 0 ? hi - lo + 1 : 0, idx.data(), (int)idx.size(), mo.local_to_world);
    }
    meshObjectsNeedRebuilding_ = true;
    objectListChanged_ = true;
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

// SkyboxTexture (:25), bound to the kernel at :457.
int RayTracingMaster::SetSkyboxTexture(const char *hdr_path, int max_size)
{
    if (!ctx_) return PT_ERR_INVALID;
    std::vector<float> sky; int w = 0, h = 0; std::string err;
    if (!sky_load_hdr(hdr_path, max_size, w, h, sky, err)) { error_ = "SkyboxTexture: " + err; return PT_ERR_INVALID; }
    return check(pt_set_sky(ctx_, sky.data(), w, h), "pt_set_sky");
}

int RayTracingMaster::SetUpScene(const Scene &scene)
{
    // The sphere list was produced by ptm::sphere_field (the procedure of :317-359);
    // :362-368 upload it, skipping the buffer when the list is empty.
    return check(pt_set_spheres(ctx_, scene.spheres.data(), (int)scene.spheres.size()), "pt_set_spheres");
}

int RayTracingMaster::RegisterObject(const float *vertices, int n_vertices, const int32_t *indices, int n_indices, const float local_to_world[16])
{
    if (n_vertices < 0 || n_indices < 0 || (n_vertices > 0 && !vertices) || (n_indices > 0 && !indices)) { error_ = "RegisterObject: bad mesh"; return -1; }
    for (int i = 0; i < n_indices; i++)
        if (indices[i] < 0 || indices[i] >= n_vertices) { error_ = "RegisterObject: index outside the mesh's vertices"; return -1; }
    auto mesh = std::make_shared<Mesh>();
    mesh->vertices.assign(vertices, vertices + (size_t)n_vertices * 3);
    mesh->indices.assign(indices, indices + n_indices);
    RayTracingObject obj;
    obj.mesh = mesh;
    std::memcpy(obj.localToWorld, local_to_world, sizeof obj.localToWorld);
    rayTracingObjects_.push_back(obj);                    // :298
    meshObjectsNeedRebuilding_ = true; objectListChanged_ = true;   // :299
    return (int)rayTracingObjects_.size() - 1;            // :300
}

int RayTracingMaster::RegisterInstance(int of_object, const float local_to_world[16])
{
    if (of_object < 0 || of_object >= (int)rayTracingObjects_.size()) { error_ = "RegisterInstance: no such object"; return -1; }
    RayTracingObject obj;
    obj.mesh = rayTracingObjects_[(size_t)of_object].mesh;
    std::memcpy(obj.localToWorld, local_to_world, sizeof obj.localToWorld);
    rayTracingObjects_.push_back(obj);
    meshObjectsNeedRebuilding_ = true; objectListChanged_ = true;
    return (int)rayTracingObjects_.size() - 1;
}

int RayTracingMaster::UnregisterObject(int object)
{
    if (object < 0 || object >= (int)rayTracingObjects_.size()) { error_ = "UnregisterObject: no such object"; return PT_ERR_INVALID; }
    rayTracingObjects_.erase(rayTracingObjects_.begin() + object);   // :311
    meshObjectsNeedRebuilding_ = true; objectListChanged_ = true;    // :312
    return PT_OK;
}

int RayTracingMaster::UpdateObject(int object, const float local_to_world[16])
{
    if (object < 0 || object >= (int)rayTracingObjects_.size()) { error_ = "UpdateObject: no such object"; return PT_ERR_INVALID; }
    std::memcpy(rayTracingObjects_[(size_t)object].localToWorld, local_to_world, 16 * sizeof(float));
    meshObjectsNeedRebuilding_ = true;                    // :303-307
    return PT_OK;
}

int RayTracingMaster::RebuildMeshObjectBuffers()
{
    if (!meshObjectsNeedRebuilding_) return PT_OK;        // :373-376
    meshObjectsNeedRebuilding_ = false;
    // :381-405 - clear the lists, then every object's vertices, its indices offset by its first vertex, and
    // (matrix, first index, index count).  A mesh shared by several objects is appended once and referenced by all of
    // them (the reference appends a copy per object; the triangles, their order and so the image are the same).
    if (objectListChanged_) {
        vertices_.clear(); indices_.clear();
        std::vector<std::pair<const Mesh *, std::pair<int32_t, int32_t>>> placed;     // mesh -> (first index, index count)
        meshObjects_.assign(rayTracingObjects_.size(), PtMeshObject{});
        for (size_t o = 0; o < rayTracingObjects_.size(); o++) {
            const Mesh *mesh = rayTracingObjects_[o].mesh.get();
            int32_t firstIndex = -1, count = 0;
            for (const auto &pl : placed)
                if (pl.first == mesh) { firstIndex = pl.second.first; count = pl.second.second; }
            if (firstIndex < 0) {
                const int32_t firstVertex = (int32_t)(vertices_.size() / 3);               // :390
                vertices_.insert(vertices_.end(), mesh->vertices.begin(), mesh->vertices.end());
                firstIndex = (int32_t)indices_.size();                                      // :395
                count = (int32_t)mesh->indices.size();
                for (int32_t i : mesh->indices) indices_.push_back(i + firstVertex);        // :397
                placed.push_back({mesh, {firstIndex, count}});
            }
            meshObjects_[o].indices_offset = firstIndex; meshObjects_[o].indices_count = count;   // :400-405
        }
        geometryUploaded_ = false;
        objectListChanged_ = false;
    }
    for (size_t o = 0; o < rayTracingObjects_.size(); o++)
        std::memcpy(meshObjects_[o].local_to_world, rayTracingObjects_[o].localToWorld, sizeof meshObjects_[o].local_to_world);
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
    if (stereoEnabled) {
        // :463-467 - the eye's own view and projection matrices, inverted
        matrix_inverse(stereoView, p.camera_to_world);
        matrix_inverse(stereoProj, p.camera_inverse_projection);
    } else {
        // _CameraToWorld / _CameraInverseProjection (:489-490)
        float e[3], t[3];
        currentCamera(e, t);
        float aspect = RenderHight > 0 ? (float)RenderWidth / (float)RenderHight : 1.0f;
        camera_look_at(e, t, up, fieldOfView, aspect, nearClip, farClip, p.camera_to_world, p.camera_inverse_projection);
    }
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
    // :460-462 - _leftEye is uploaded by the stereo branch only (read by renderMode 4); a mono camera leaves the
    // uniform alone, which for a second view made with eyeOffset is taken to mean "left when shifted left"
    p.left_eye = stereoEnabled ? (stereoLeft ? 1 : 0) : (eyeOffset < 0.0f ? 1 : 0);
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
    if (stereoEnabled) {
        // :473-478 - the stereo branch keeps the lists only in the partial-frame modes 5 and 6, and there only across
        // the frame that STARTS a cycle: every frame with counter % divisions != 0 clears them (the mono branch clears
        // at == 0, and covers modes 5 .. 8)
        if ((renderMode != 5 && renderMode != 6) || (divisions > 0 && counter_ % divisions != 0)) oldCameras_.clear();
        // :480-485 - this eye's matrices: inverse projection, inverse view, view, projection
        if (renderMode < 11 && renderMode != 1) {
            PtCameraHistory h;
            std::memcpy(h.camera_to_world, p.camera_to_world, sizeof h.camera_to_world);
            std::memcpy(h.camera_inverse_projection, p.camera_inverse_projection, sizeof h.camera_inverse_projection);
            std::memcpy(h.world_to_camera, stereoView, sizeof h.world_to_camera);
            std::memcpy(h.camera_projection, stereoProj, sizeof h.camera_projection);
            oldCameras_.push_back(h);
        }
        return PT_OK;
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

void RayTracingMaster::GetRenderScale(int screen_width, int screen_height)
{
    // Mathf.RoundToInt rounds halves to the even neighbour, as nearbyint does in the default rounding mode
    RenderHight = (int)std::nearbyint((double)((float)screen_height * renderScale));
    RenderWidth = (int)std::nearbyint((double)((float)screen_width * renderScale));
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
    // :563-576 gives _converged, denoisedTex and normalTex (int)(RenderWidth / renderScale) texels a side - the screen size -
    // while _target has the render size; every target here has the render size.  The two agree for renderScale 1
    // (TestScene.unity:783) and whenever the division gives the render size back; renderMode 1 reads none of those
    // targets.  Any other combination would show a history of another size than the reference's: refused, not approximated.
    if (renderMode != 1 && ((int)((float)RenderWidth / renderScale) != RenderWidth || (int)((float)RenderHight / renderScale) != RenderHight)) {
        error_ = "Render: renderScale " + std::to_string(renderScale) + " gives the history targets (_converged, denoisedTex: RayTracingMaster.cs:563-571) "
                 "another size than the render target; only renderMode 1 is supported at this scale";
        return PT_ERR_UNSUPPORTED;
    }
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

// No C++ exception leaves this library either: what throws under an entry point (a host allocation that fails) comes back as a code.
template <class F> static int ptm_guarded(F body) noexcept
{
    try { return body(); }
    catch (const std::bad_alloc &) { return PT_ERR_NOMEM; }
    catch (...) { return PT_ERR_HIP; }
}

extern "C" {

int ptm_sphere_field(int seed, uint32_t spheres_max, float rmin, float rmax, float placement_radius, PtSphere *out, int capacity)
{
    return ptm_guarded([&]() -> int {
    std::vector<PtSphere> v;
    int n = sphere_field(seed, spheres_max, rmin, rmax, placement_radius, v);
    if (n > capacity) n = capacity;
    if (out && n > 0) std::memcpy(out, v.data(), sizeof(PtSphere) * (size_t)n);
    return n;
    });
}

void ptm_sky_procedural(int w, int h, float *rgba) { sky_procedural(w, h, rgba); }

static thread_local std::string g_sky_error;
const char *ptm_sky_error(void) { return g_sky_error.c_str(); }
int ptm_sky_load_hdr(const char *path, int max_size, int *w, int *h, float *rgba, int capacity_texels)
{
    return ptm_guarded([&]() -> int {
    g_sky_error.clear();
    if (!path || !w || !h) { g_sky_error = "ptm_sky_load_hdr: NULL argument"; return -1; }
    std::vector<float> img; int iw = 0, ih = 0;
    if (!sky_load_hdr(path, max_size, iw, ih, img, g_sky_error)) return -1;
    *w = iw; *h = ih;
    if (rgba) {
        if ((long long)capacity_texels < (long long)iw * ih) { g_sky_error = "ptm_sky_load_hdr: buffer too small"; return -1; }
        std::memcpy(rgba, img.data(), img.size() * sizeof(float));
    }
    return 0;
    });
}

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

int ptm_matrix_inverse(const float m[16], float out[16]) { return matrix_inverse(m, out) ? 1 : 0; }

void ptm_camera_stereo_eye(const float eye[3], const float target[3], const float up[3], float fov, float aspect,
                           float z_near, float z_far, float offset, float shear, float view[16], float proj[16])
{
    camera_stereo_eye(eye, target, up, fov, aspect, z_near, z_far, offset, shear, view, proj);
}

ptm_scene *ptm_scene_create(const char *name)
{
    try {
        Scene *s = Scene::create(name ? name : "");
        if (!s) return nullptr;
        return new ptm_scene{s};
    } catch (...) { return nullptr; }
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
int ptm_scene_add_obj(ptm_scene *s, const char *path, const float m[16]) { return ptm_guarded([&]() -> int { return s->s->add_obj(path, m); }); }

int ptm_master_create(int device, ptm_master **out)
{
    return ptm_guarded([&]() -> int {
    if (!out) return PT_ERR_INVALID;
    ptm_master *h = new ptm_master{new RayTracingMaster(device), ""};
    *out = h;
    if (!h->m->ok()) return PT_ERR_HIP;   // handle stays valid so the caller can read the error, then destroy it
    return PT_OK;
    });
}
void ptm_master_destroy(ptm_master *m) { if (m) { delete m->m; delete m; } }
const char *ptm_master_error(const ptm_master *m) { return m ? m->m->error().c_str() : "null master"; }
pt_context *ptm_master_context(ptm_master *m) { return m->m->context(); }
int ptm_master_set_sky_hdr(ptm_master *m, const char *path, int max_size)
{
    return ptm_guarded([&]() -> int {
    if (!m || !path) return PT_ERR_INVALID;
    return m->m->SetSkyboxTexture(path, max_size);
    });
}

int ptm_master_load_scene(ptm_master *m, const ptm_scene *s, int sky_w, int sky_h)
{
    return ptm_guarded([&]() -> int {
    std::memcpy(m->m->eye, s->s->eye, 12); std::memcpy(m->m->target, s->s->target, 12); m->m->fieldOfView = s->s->fov;
    return m->m->OnEnable(*s->s, sky_w, sky_h);
    });
}
int ptm_master_configure(ptm_master *m, int width, int height, int spp, int max_bounces, int depth, int accumulate, uint32_t sample_frames)
{
    RayTracingMaster &r = *m->m;
    r.RenderWidth = width; r.RenderHight = height; r.samplesPerPixel = spp; r.maxBounces = max_bounces;
    r.depth = depth != 0; r.accumulate = accumulate; r.sampleFrames = sample_frames;
    return PT_OK;
}
int ptm_master_set_render_scale(ptm_master *m, int screen_width, int screen_height, float render_scale)
{
    if (!(render_scale > 0.0f) || screen_width < 0 || screen_height < 0) return PT_ERR_INVALID;
    m->m->renderScale = render_scale;
    m->m->GetRenderScale(screen_width, screen_height);
    return PT_OK;
}
void ptm_master_render_size(const ptm_master *m, int32_t size[2]) { size[0] = m->m->RenderWidth; size[1] = m->m->RenderHight; }
int ptm_master_set_render_mode(ptm_master *m, int render_mode) { m->m->renderMode = render_mode; return PT_OK; }
int ptm_master_set_divisions(ptm_master *m, int divisions) { m->m->divisions = divisions; return PT_OK; }
int ptm_master_set_camera(ptm_master *m, const float eye[3], const float target[3], float fov, float eye_offset)
{
    std::memcpy(m->m->eye, eye, 12); std::memcpy(m->m->target, target, 12);
    m->m->fieldOfView = fov; m->m->eyeOffset = eye_offset;
    return PT_OK;
}
int ptm_master_set_stereo(ptm_master *m, int enabled, int left_eye, const float view[16], const float proj[16])
{
    RayTracingMaster &r = *m->m;
    r.stereoEnabled = enabled != 0; r.stereoLeft = left_eye != 0;
    if (enabled) {
        if (!view || !proj) return PT_ERR_INVALID;
        std::memcpy(r.stereoView, view, sizeof r.stereoView); std::memcpy(r.stereoProj, proj, sizeof r.stereoProj);
    }
    return PT_OK;
}
int ptm_master_register_object(ptm_master *m, const float *vertices, int n_vertices, const int32_t *indices, int n_indices, const float local_to_world[16])
{
    return ptm_guarded([&]() -> int {
    return m->m->RegisterObject(vertices, n_vertices, indices, n_indices, local_to_world);
    });
}
int ptm_master_register_instance(ptm_master *m, int of_object, const float local_to_world[16]) { return ptm_guarded([&]() -> int { return m->m->RegisterInstance(of_object, local_to_world); }); }
int ptm_master_unregister_object(ptm_master *m, int object) { return ptm_guarded([&]() -> int { return m->m->UnregisterObject(object); }); }
int ptm_master_update_object(ptm_master *m, int object, const float local_to_world[16]) { return ptm_guarded([&]() -> int { return m->m->UpdateObject(object, local_to_world); }); }
int ptm_master_object_count(const ptm_master *m) { return m->m->objectCount(); }
int ptm_master_render(ptm_master *m) { return ptm_guarded([&]() -> int { return m->m->Render(); }); }
int ptm_master_peek_params(ptm_master *m, PtFrameParams *out) { m->m->FillParams(*out); return PT_OK; }
void ptm_master_reset_accumulation(ptm_master *m) { m->m->ResetAccumulation(); }
uint32_t ptm_master_current_sample(const ptm_master *m) { return m->m->currentSample(); }
uint32_t ptm_master_frame_index(const ptm_master *m) { return m->m->frameIndex(); }

}  // extern "C"
