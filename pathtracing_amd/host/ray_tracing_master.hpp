// C++ mirror of the reference's host component
// (Assets/Scripts/RayTracingMaster.cs + RayTracingObject.cs), written against
// the libptrace C ABI only.  Method and field names follow the C# so the two can
// be read side by side; what Unity's engine supplied (camera matrices,
// UnityEngine.Random) is supplied by scene.hpp.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include "../../include/ptrace.h"
#include "scene.hpp"

namespace ptm {

class RayTracingMaster {
public:
    // ---- inspector fields -------------------------------------------------
    int RenderWidth = 0, RenderHight = 0;        // RayTracingMaster.cs:87-88 (spelling as in the reference)
    float renderScale = 1.0f;                     // :86 (TestScene.unity:783: 1)
    // GetRenderScale (:585-597): RenderWidth / RenderHight = Mathf.RoundToInt(screen or eye-texture size * renderScale).
    // The reference then makes _target that size and _converged / denoisedTex (int)(RenderWidth / renderScale) - the
    // screen size again; here every target has the render size, which is the same thing for renderScale == 1.
    void GetRenderScale(int screen_width, int screen_height);
    int samplesPerPixel = 1;                      // :85
    uint32_t sampleFrames = 3;                    // :74
    bool depth = false;                           // :686
    int renderMode = 1;                           // RenderMode.Default, :979
    int divisions = 10;                           // :701
    int maxBounces = 4;                           // hard-coded in the shader, RayTracingShader.compute:867
    int accumulate = PT_ACCUM_NONE;               // the AddMat pass is dormant in the reference (:804-820)

    // ---- camera (Unity supplies these through Camera) -----------------------
    float eye[3] = {0, 5, -10}, target[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    float fieldOfView = 60.0f, nearClip = 0.1f, farClip = 100.0f;  // TestScene.unity:851-853 near/far
    float eyeOffset = 0.0f;                        // signed offset along camera right (a second view of the mono camera)
    // _camera.stereoEnabled / stereoTargetEye and what GetStereoViewMatrix / GetStereoProjectionMatrix return for that
    // eye (:459-467); column-major like Matrix4x4.  With stereoEnabled the branch :459-485 runs instead of :486-527.
    bool stereoEnabled = false;
    bool stereoLeft = true;                        // stereoTargetEye == StereoTargetEyeMask.Left
    float stereoView[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float stereoProj[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

    explicit RayTracingMaster(int device);
    ~RayTracingMaster();
    RayTracingMaster(const RayTracingMaster &) = delete;
    RayTracingMaster &operator=(const RayTracingMaster &) = delete;

    bool ok() const { return ctx_ != nullptr; }
    const std::string &error() const { return error_; }
    pt_context *context() { return ctx_; }

    // OnEnable (RayTracingMaster.cs:139-152): _currentSample = 0; SetUpScene()
    int OnEnable(const Scene &scene, int sky_w, int sky_h);
    int SetSkyboxTexture(const char *hdr_path, int max_size = 2048);   // a Radiance .hdr picture in place of the procedural sky
    // RegisterObject / UpdateObject / UnregisterObject (RayTracingMaster.cs:296-313; called by RayTracingObject.cs:24-38
    // from OnEnable / a changed transform / OnDisable).  An object is a mesh (object-space vertices, sub-mesh 0's
    // indices) and a localToWorld matrix; RegisterObject returns its position in the list, as the reference does, and
    // UnregisterObject moves the later objects up by one.  RegisterInstance adds an object that shares an earlier
    // object's mesh (MeshFilter.sharedMesh).  Each sets _meshObjectsNeedRebuilding; the next Render() rebuilds.
    int RegisterObject(const float *vertices, int n_vertices, const int32_t *indices, int n_indices, const float local_to_world[16]);
    int RegisterInstance(int of_object, const float local_to_world[16]);
    int UnregisterObject(int object);
    int UpdateObject(int object, const float local_to_world[16]);
    int objectCount() const { return (int)rayTracingObjects_.size(); }

    int Render();                                       // :704-747 (+ :885-888)
    int SetShaderParameters(PtFrameParams &p);          // :455-540, :735-738 (uploads and then updates the camera history)
    void FillParams(PtFrameParams &p) const;            // the per-frame uniforms alone, no side effects
    void ResetAccumulation() { currentSample_ = 0; }    // :145, :753
    uint32_t currentSample() const { return currentSample_; }
    uint32_t frameIndex() const { return frame_; }
    int counterValue() const { return counter_; }

private:
    int SetUpScene(const Scene &scene);                 // :315-369 (sphere list comes from the Scene)
    int RebuildMeshObjectBuffers();                     // :371-413
    int InitRenderTexture();                            // :542-583
    int check(int rc, const char *what);

    pt_context *ctx_ = nullptr;
    std::string error_;
    struct Mesh { std::vector<float> vertices; std::vector<int32_t> indices; };   // Mesh.vertices, Mesh.GetIndices(0)
    struct RayTracingObject { std::shared_ptr<const Mesh> mesh; float localToWorld[16]; };
    std::vector<RayTracingObject> rayTracingObjects_;   // _rayTracingObjects :47
    bool objectListChanged_ = false;                    // an object came or went since the last upload (not just a transform)
    std::vector<PtMeshObject> meshObjects_;             // _meshObjects  :50
    std::vector<float> vertices_;                       // _vertices     :51
    std::vector<int32_t> indices_;                      // _indices      :52
    bool meshObjectsNeedRebuilding_ = false;            // :48
    bool geometryUploaded_ = false;
    int targetW_ = 0, targetH_ = 0;                     // size of _target
    uint32_t currentSample_ = 0;                        // _currentSample :45
    uint32_t frame_ = 0;                                // frames dispatched; drives the per-frame _Seed
    int counter_ = 0;                                   // counter :702
    std::vector<PtCameraHistory> oldCameras_;           // oldIPR / oldCTW / oldWTC / oldPRJ :450-453, one struct per entry
    void currentCamera(float e[3], float t[3]) const;   // eye and target with the stereo offset applied
};

}  // namespace ptm
