// C++ side of ptmaster.h: mesh builder, scene container, scene pieces.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/ptrace.h"

namespace ptm {

// Column-major translate * rotate-about-y * uniform scale.
void matrix_trs(float m[16], float tx, float ty, float tz, float yaw_deg, float scale);

struct MeshBuilder {
    std::vector<float> vertices;  // xyz
    std::vector<int> indices;
    int add_vertex(float x, float y, float z);
    void add_tri_outward(int a, int b, int c, const float inside[3]);
    void add_blob(const float c[3], float R, int nu, int nv, float amp, float freq, uint32_t seed);
    void add_box(const float lo[3], const float hi[3]);
    void add_column(float cx, float cz, float y0, float y1, float radius, int sides, int rings);
    void add_arch(float x0, float x1, float y, float cz, float tube, int segs, int sides);
    void add_wall(float x0, float x1, float y0, float y1, float z, float face, int nx, int ny, float bump, uint32_t seed);
};

int sphere_field(int sphere_seed, uint32_t spheres_max, float rmin, float rmax, float placement_radius,
                 std::vector<PtSphere> &out);
void sky_procedural(int w, int h, float *rgba);
// A Radiance .hdr picture as a sky texture (row 0 at v = 0, i.e. the picture's bottom), halved until neither side exceeds max_size (0: as it is).
bool sky_load_hdr(const char *path, int max_size, int &w, int &h, std::vector<float> &rgba, std::string &err);
void camera_look_at(const float eye[3], const float target[3], const float up[3], float fov_y_deg, float aspect,
                    float z_near, float z_far, float c2w[16], float ip[16]);
// The same camera as one element of the shader's four history arrays (cameraToWorld, inverse projection,
// worldToCamera, projection - RayTracingMaster.cs:522-525).
void camera_history_entry(const float eye[3], const float target[3], const float up[3], float fov_y_deg, float aspect,
                          float z_near, float z_far, PtCameraHistory *out);
float frame_seed(uint32_t k);
// Matrix4x4.inverse for a column-major 4x4 (Gauss-Jordan with partial pivoting in double precision); false, and the
// identity in `out`, when the matrix is singular.
bool matrix_inverse(const float m[16], float out[16]);
// One eye of a stereo camera as Unity's Camera.GetStereoViewMatrix / GetStereoProjectionMatrix report it: the view
// (world-to-camera) matrix of a look-at camera moved by `offset` along its right axis, and an off-axis OpenGL
// projection whose frustum is shifted sideways by `shear` (element (0,2); 0 gives the symmetric frustum).
void camera_stereo_eye(const float eye[3], const float target[3], const float up[3], float fov_y_deg, float aspect,
                       float z_near, float z_far, float offset, float shear, float view[16], float proj[16]);

// What the reference keeps in its static lists (RayTracingMaster.cs:47-52) plus the sphere list.
struct Scene {
    std::string name;
    std::vector<PtSphere> spheres;
    std::vector<PtMeshObject> objects;
    std::vector<float> vertices;
    std::vector<int32_t> indices;
    float eye[3] = {0, 5, -10}, target[3] = {0, 0, 0}, fov = 60.0f;

    static Scene *create(const std::string &name);
    int register_object(const MeshBuilder &mesh, const float local_to_world[16]);  // RegisterObject + rebuild
    int add_instance(int object, const float local_to_world[16]);                  // same index range, new matrix
    int add_obj(const std::string &path, const float local_to_world[16]);
};

}  // namespace ptm
